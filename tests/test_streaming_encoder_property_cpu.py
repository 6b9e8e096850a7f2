"""CPU: why the streaming chain re-encodes everything heard so far ("incremental encoder").

The reference builds the streaming speech encoder with FULL self-attention (models/conformer_shaw/builder.py:127-146: the
Shaw window only clamps the relative-position term, no attention mask) and its OfflineWav2VecBertEncoderAgent
(offline_w2v_bert_encoder.py:66-100) encodes the whole utterance again for every segment.  Under full attention the encoder
output of an already-heard position changes when later frames arrive, so an encoder that keeps the outputs of earlier
segments cannot reproduce the reference's numbers - and the text decoder's p_choose reads those numbers.  This test states
the property on the oracle: it is the reason the product keeps the re-encode, not an omission."""
import torch

from oracle import unity as ou
from tests import common


def test_encoder_output_of_heard_positions_changes_when_more_audio_arrives():
    orc = common.make_oracle()
    cfg = orc.cfg
    fb, _ = orc.collate_fbank([common.waves((2.0,))[0]])
    for T in (96, 128, 160):
        more, _ = orc.encode_speech(fb[:, : T + 32], torch.tensor([T + 32]))
        heard, _ = orc.encode_speech(fb[:, :T], torch.tensor([T]))
        n = heard.shape[1]
        diff = (more[:, :n] - heard).abs().amax(-1)[0]
        # every position moves by a sizeable fraction of the output scale (measured 0.3 - 1.9 at a mean magnitude of 0.79)
        assert float(diff.min()) > 0.05 * float(heard.abs().mean()), (T, diff)

    # the Conformer stack alone (no adaptor): the first positions, 60+ frames away from the new audio and far outside the
    # Shaw window of 8 future positions, move as well - the window bounds the positional term, not the attention
    x_more, l_more = ou.speech_frontend(orc.P, cfg, fb[:, :160], torch.tensor([160]))
    x_heard, l_heard = ou.speech_frontend(orc.P, cfg, fb[:, :128], torch.tensor([128]))
    for i in range(cfg.enc_layers):
        x_more = ou.conformer_block(orc.P, cfg, f"speech_encoder.inner.layers.{i}", x_more, l_more)
        x_heard = ou.conformer_block(orc.P, cfg, f"speech_encoder.inner.layers.{i}", x_heard, l_heard)
    diff = (x_more[:, : x_heard.shape[1]] - x_heard).abs().amax(-1)[0]
    assert float(diff[:5].min()) > 0.05


def test_the_causal_parts_alone_would_be_prefix_stable():
    """What does NOT break prefix stability: the causal depthwise convolution and the stride-2 frame stacking - position j
    of the convolution module over T frames equals position j over T + k frames."""
    orc = common.make_oracle()
    cfg = orc.cfg
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 40, cfg.model_dim, generator=g)
    pre = "speech_encoder.inner.layers.0.conv"
    full = ou.conformer_conv(orc.P, cfg, pre, x, torch.tensor([40]))
    part = ou.conformer_conv(orc.P, cfg, pre, x[:, :25], torch.tensor([25]))
    assert torch.allclose(full[:, :25], part, atol=1e-6)
