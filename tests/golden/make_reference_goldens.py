#!/usr/bin/env python
"""Mints the golden fixtures under tests/golden/ from the REFERENCE'S OWN code.

Run in the build container (needs /root/reference; the GPU box has only the
committed .npz files):

    python tests/golden/make_reference_goldens.py

What executes the reference and how
-----------------------------------
* fbank_knf.npz        - the reference's kaldi-native-fbank C++ compiled by
                         oracle/build_ref.sh (oracle/_ref/libknf_ref.so).
* vocoder_ref.npz      - models/vocoder/{hifigan,codehifigan,vocoder}.py imported
                         by file path (torch-only): ``Vocoder.forward`` with
                         ``dur_prediction=False`` on the seeded tiny vocoder.
* unit_tokenizer_ref.npz - models/unity/unit_tokenizer.py (UnitTokenizer,
                         encoder, decoder) imported by file path.
* nar_frontend_ref.npz - models/unity/nar_decoder_frontend.py +
                         length_regulator.py: ``NARDecoderFrontend.forward``
                         (char-length rules, char ids, hard upsampling,
                         VariancePredictor, duration rounding, unit positions).
* fft_layer_ref.npz    - models/unity/fft_decoder_layer.py: Conv1dBlock and the
                         post-LN FeedForwardTransformerLayer (attention supplied
                         by the oracle's MHA, the only piece not in the tree).

The reference modules import fairseq2 names that are not installed here;
tests/golden/_fairseq2_stub.py supplies few-line torch stand-ins for exactly
those primitives (LayerNorm, Linear, PaddingMask, sinusoidal positions).  The
model logic that runs is the reference's, unmodified, read from /root/reference.

Inputs are seeded; weights are the seeded synthetic checkpoints of
seamless_communication_amd.synthetic (deterministic per key), whose checksum is
stored next to each output so a drift of the generator is detected.
"""
from __future__ import annotations

import ctypes
import hashlib
import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))
REF = Path("/root/reference/src/seamless_communication")

from seamless_communication_amd import cards, synthetic as syn  # noqa: E402
from seamless_communication_amd.config import tiny_config  # noqa: E402
from seamless_communication_amd.tokenizer import CharTokenizer, NllbTextTokenizer  # noqa: E402


def sd_checksum(sd) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
    return h.hexdigest()


def load_ref(modname: str, relpath: str):
    spec = importlib.util.spec_from_file_location(modname, REF / relpath)
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


# --------------------------------------------------------------------------- #
def make_fbank():
    lib = ctypes.CDLL(str(ROOT / "oracle" / "_ref" / "libknf_ref.so"))
    lib.knf_ref_num_frames.restype = ctypes.c_int32
    lib.knf_ref_num_frames.argtypes = [ctypes.c_int64]
    lib.knf_ref_fbank.restype = ctypes.c_int32
    lib.knf_ref_fbank.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    out = {}
    waves = {
        "synth0_1s": syn.synthetic_waveform(0, 1.0).numpy(),
        "synth3_0p3s": syn.synthetic_waveform(3, 0.3).numpy(),
        "ramp": (np.linspace(-0.5, 0.5, 3333) * np.sin(np.arange(3333) * 0.05)).astype(np.float32),
    }
    for k, w in waves.items():
        x = np.ascontiguousarray(w.astype(np.float32) * np.float32(2.0**15))
        n = lib.knf_ref_num_frames(len(x))
        fb = np.zeros((n, 80), dtype=np.float32)
        assert lib.knf_ref_fbank(x.ctypes.data, len(x), fb.ctypes.data) == n
        out[k + "_wav"] = w.astype(np.float32)
        out[k + "_fbank"] = fb
    np.savez_compressed(HERE / "fbank_knf.npz", **out)
    print("fbank_knf.npz", {k: v.shape for k, v in out.items()})
    make_fbank_rates(lib)


FBANK_RATES = (8000, 22050, 32000, 44100, 48000)


def make_fbank_rates(lib):
    """fbank_knf_rates.npz: the same compiled library at other sample rates (window 25 ms / shift 10 ms / FFT size / mel banks follow
    the rate; fairseq2n hands the waveform's own rate to kaldi, inference/translator.py:270-292): 0.4 s of synthetic audio per rate."""
    lib.knf_ref_fbank_rate.restype = ctypes.c_int32
    lib.knf_ref_fbank_rate.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]
    out = {}
    for i, rate in enumerate(FBANK_RATES):
        n_samp = int(0.4 * rate) + 37 * i
        w = (syn.synthetic_waveform(20 + i, n_samp / 16000.0 + 0.01).numpy()[:n_samp]).astype(np.float32)
        x = np.ascontiguousarray(w * np.float32(2.0**15))
        fb = np.zeros((n_samp, 80), dtype=np.float32)  # more rows than frames
        n = lib.knf_ref_fbank_rate(x.ctypes.data, len(x), float(rate), fb.ctypes.data)
        out[f"r{rate}_wav"] = w
        out[f"r{rate}_fbank"] = fb[:n].copy()
    np.savez_compressed(HERE / "fbank_knf_rates.npz", **out)
    print("fbank_knf_rates.npz", {k: v.shape for k, v in out.items()})


# --------------------------------------------------------------------------- #
def make_vocoder(cfg, lr):
    hifigan = load_ref("seamless_communication.models.vocoder.hifigan", "models/vocoder/hifigan.py")
    sys.modules["seamless_communication.models.unity"].VariancePredictor = lr.VariancePredictor
    codehifigan = load_ref("seamless_communication.models.vocoder.codehifigan", "models/vocoder/codehifigan.py")
    vocoder = load_ref("seamless_communication.models.vocoder.vocoder", "models/vocoder/vocoder.py")
    v = cfg.vocoder
    gen = codehifigan.CodeGenerator(
        v.upsample_rates, v.upsample_kernel_sizes, v.upsample_initial_channel, v.resblock_kernel_sizes,
        v.resblock_dilation_sizes, v.model_in_dim, v.num_embeddings, v.embedding_dim, {}, v.lang_embedding_dim,
        v.num_langs, v.spkr_embedding_dim, v.num_spkrs,
    )
    voc = vocoder.Vocoder(gen, cards.vocoder_lang_spkr_idx_map())
    sd = syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED)
    missing, unexpected = voc.load_state_dict({k: t.float() for k, t in sd.items()}, strict=True), None
    voc.eval()
    rng = np.random.RandomState(11)
    units = torch.from_numpy(rng.randint(0, v.num_embeddings, size=(2, 23)).astype(np.int64))
    units[1, 19:] = 1  # padded batch: pads are unit 1 (translator.py:407-409)
    with torch.inference_mode():
        wav = voc(units, ["fra", "spa"], [-1, 3], dur_prediction=False)
        wav_single = voc(units[0], "eng", -1, dur_prediction=False)
    np.savez_compressed(
        HERE / "vocoder_ref.npz", units=units.numpy(), langs=np.array(["fra", "spa"]), spkrs=np.array([-1, 3]),
        wav=wav.numpy(), wav_single_eng=wav_single.numpy(), sd_sha256=np.array(sd_checksum(sd)),
    )
    print("vocoder_ref.npz", tuple(wav.shape), float(wav.abs().max()))


# --------------------------------------------------------------------------- #
def make_unit_tokenizer():
    ut = load_ref("seamless_communication.models.unity.unit_tokenizer", "models/unity/unit_tokenizer.py")
    out = {}
    rng = np.random.RandomState(5)
    for arch, tag in (("seamlessM4T_large", "ar"), ("seamlessM4T_large_v2", "nar")):
        tok = ut.UnitTokenizer(num_units=100, langs=["eng", "deu", "fra"], model_arch=arch)
        units = torch.from_numpy(rng.randint(0, 103, size=(4, 9)).astype(np.int64))
        enc = tok.create_encoder(lang="deu", device=torch.device("cpu"))(units.clone())
        tokens = enc.clone()
        tokens[1, 5] = tok.vocab_info.eos_idx
        tokens[2, 7] = tok.vocab_info.pad_idx
        dec = tok.create_decoder()(tokens.clone())
        out.update({f"{tag}_units": units.numpy(), f"{tag}_encoded": enc.numpy(), f"{tag}_tokens": tokens.numpy(),
                    f"{tag}_decoded": dec.numpy(), f"{tag}_vocab_size": np.array(tok.vocab_info.size),
                    f"{tag}_lang_idx": np.array([tok.lang_to_index(l) for l in ("eng", "deu", "fra")])})
    np.savez_compressed(HERE / "unit_tokenizer_ref.npz", **out)
    print("unit_tokenizer_ref.npz", sorted(out))


# --------------------------------------------------------------------------- #
class _Tok:
    """Object with the ``.model`` / ``.vocab_info`` surface NARDecoderFrontend uses."""

    def __init__(self, inner):
        self.model = inner
        self.vocab_info = inner.vocab_info


def make_nar_frontend(cfg, lr, stub):
    naf = load_ref("seamless_communication.models.unity.nar_decoder_frontend", "models/unity/nar_decoder_frontend.py")
    M = cfg.model_dim
    sd = syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED)
    f = "t2u_model.decoder_frontend"
    tt = NllbTextTokenizer(cfg.text_vocab_size, cards.TEXT_LANGS)
    ct = CharTokenizer(cfg.char_vocab_size)
    dp = lr.VariancePredictor(M, cfg.var_pred_hidden_dim, cfg.var_pred_kernel_size, var_pred_dropout=0.5)
    va = lr.VarianceAdaptor(dp, None)
    embed = stub.Embedding(cfg.unit_vocab_size, M)
    embed_char = stub.Embedding(cfg.char_vocab_size, M)
    unit_pos = stub.SinusoidalPositionEncoder(M, cfg.unit_max_seq_len, _legacy_pad_idx=cfg.unit_pad_idx)
    char_pos = stub.SinusoidalPositionEncoder(M, cfg.char_max_seq_len, _legacy_pad_idx=cfg.unit_pad_idx)
    fe = naf.NARDecoderFrontend(embed, embed_char, _Tok(tt), _Tok(ct), unit_pos, char_pos, va, dropout_p=0.0)
    own = {k[len(f) + 1:]: v.float() for k, v in sd.items() if k.startswith(f + ".")}
    fe.load_state_dict(own, strict=True)
    fe.eval()

    # text rows as the generator hands them over: [</s>, lang, w1..wn] with the final EOS
    # trimmed; the shorter row keeps its EOS inside the matrix (generator.py:281-291).
    from tests.common import random_text_seqs

    lens = [13, 9]
    text = torch.from_numpy(random_text_seqs(cfg, tt, 2, lens, seed=21))
    # force a few punctuation / space-prefixed / bare-SPACE pieces so every rule fires
    pieces = {p: tt.token_to_index(p) for p in (",", ".", "▁")}
    text[0, 4] = pieces[","]
    text[0, 8] = pieces["."]
    text[1, 3] = pieces["▁"]
    rng = torch.Generator().manual_seed(77)
    enc_out = torch.randn(2, max(lens), M, generator=rng)
    out = {"text_seqs": text.numpy().copy(), "enc_out": enc_out.numpy().copy()}
    with torch.inference_mode():
        for dfac in (1.0, 1.3):
            t = text.clone()
            seqs, mask, dur = fe(enc_out.clone(), None, t, duration_factor=dfac)
            tag = f"df{dfac}".replace(".", "p")
            out[f"{tag}_seqs"] = seqs.numpy()
            out[f"{tag}_unit_lens"] = mask.seq_lens.numpy()
            out[f"{tag}_durations"] = dur.numpy()
        t = text.clone()
        char_seqs, char_seq_lens, char_lens = fe.text_to_char_seqs(t)
        out.update(char_seqs=char_seqs.numpy(), char_seq_lens=char_seq_lens.numpy(), char_lens=char_lens.numpy(),
                   text_seqs_after=t.numpy())  # in-place EOS->PAD side effect
        # VariancePredictor alone on ragged input
        x = torch.randn(2, 17, M, generator=rng)
        plens = torch.tensor([17, 11])
        logd = dp(x, stub.PaddingMask(plens, 17))
        out.update(vp_in=x.numpy(), vp_lens=plens.numpy(), vp_log_dur=logd.numpy())
        # HardUpsampling alone
        d = torch.tensor([[2, 0, 3, 1], [1, 1, 0, 0]])
        xs = torch.randn(2, 4, 6, generator=rng)
        up, ul = lr.HardUpsampling()(xs, d)
        out.update(hu_in=xs.numpy(), hu_dur=d.numpy(), hu_out=up.numpy(), hu_lens=ul.numpy())
    out["sd_sha256"] = np.array(sd_checksum({k: v for k, v in sd.items() if k.startswith(f)}))
    np.savez_compressed(HERE / "nar_frontend_ref.npz", **out)
    print("nar_frontend_ref.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


def make_vocoder_dur(cfg, lr):
    """vocoder_dur_ref.npz: the reference CodeGenerator WITH its duration predictor (codehifigan.py:46-48, built from
    dur_predictor_params like models/vocoder/builder.py:53-58,109), forward(dur_prediction=True) executed on single
    utterances (the reference concatenates the expanded items of a batch, so a ragged batch raises)."""
    load_ref("seamless_communication.models.vocoder.hifigan", "models/vocoder/hifigan.py")
    sys.modules["seamless_communication.models.unity"].VariancePredictor = lr.VariancePredictor
    codehifigan = load_ref("seamless_communication.models.vocoder.codehifigan", "models/vocoder/codehifigan.py")
    vocoder = load_ref("seamless_communication.models.vocoder.vocoder", "models/vocoder/vocoder.py")
    v = cfg.vocoder
    params = {"encoder_embed_dim": v.embedding_dim, "var_pred_hidden_dim": v.dur_pred_hidden_dim,
              "var_pred_kernel_size": v.dur_pred_kernel_size, "var_pred_dropout": 0.5}
    gen = codehifigan.CodeGenerator(
        v.upsample_rates, v.upsample_kernel_sizes, v.upsample_initial_channel, v.resblock_kernel_sizes,
        v.resblock_dilation_sizes, v.model_in_dim, v.num_embeddings, v.embedding_dim, params, v.lang_embedding_dim,
        v.num_langs, v.spkr_embedding_dim, v.num_spkrs,
    )
    voc = vocoder.Vocoder(gen, cards.vocoder_lang_spkr_idx_map())
    sd = syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED, with_dur_predictor=True)
    voc.load_state_dict({k: t.float() for k, t in sd.items()}, strict=True)
    voc.eval()
    rng = np.random.RandomState(17)
    out = {"sd_sha256": np.array(sd_checksum(sd))}
    for tag, T in (("a", 31), ("b", 9)):
        units = torch.from_numpy(rng.randint(0, v.num_embeddings, size=(1, T)).astype(np.int64))
        with torch.inference_mode():
            x = gen.dict(units)
            log_dur = gen.dur_predictor(x, None)
            dur = torch.clamp(torch.round((torch.exp(log_dur) - 1)).long(), min=1)
            wav = voc(units[0], "fra", -1, dur_prediction=True)
        out.update({f"{tag}_units": units.numpy(), f"{tag}_dur": dur.numpy(), f"{tag}_log_dur": log_dur.numpy(), f"{tag}_wav": wav.numpy()})
        print("vocoder_dur_ref", tag, dur.flatten().tolist()[:12], tuple(wav.shape))
    np.savez_compressed(HERE / "vocoder_dur_ref.npz", **out)


# --------------------------------------------------------------------------- #
def make_fft_layer(cfg, stub):
    fl = load_ref("seamless_communication.models.unity.fft_decoder_layer", "models/unity/fft_decoder_layer.py")
    from oracle import unity as ou

    M = cfg.model_dim
    sd = syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED)
    p = "t2u_model.decoder.layers.0"
    P = ou.Params(sd)

    class OracleMHA(stub.MultiheadAttention):
        """fairseq2's StandardMultiheadAttention is not in the tree; the layer under
        test only needs *an* attention module, the golden pins what surrounds it."""

        def __init__(self):
            super().__init__()
            self.model_dim = M

        def forward(self, seqs, padding_mask, keys, key_padding_mask, values, **kw):
            lens = None if key_padding_mask is None else key_padding_mask.seq_lens
            return ou.mha(P, p + ".self_attn", seqs, keys, cfg.num_heads, key_lens=lens)

    conv = fl.Conv1dBlock(M, cfg.t2u_conv_inner_dim, cfg.t2u_conv_kernel, bias=True)
    layer = fl.FeedForwardTransformerLayer(OracleMHA(), conv, dropout_p=0.0, conv1d_dropout_p=0.0)
    own = {k[len(p) + 1:]: v.float() for k, v in sd.items() if k.startswith(p + ".") and ".self_attn." not in k}
    layer.load_state_dict(own, strict=True)
    layer.eval()
    rng = torch.Generator().manual_seed(3)
    x = torch.randn(2, 19, M, generator=rng)
    lens = torch.tensor([19, 12])
    with torch.inference_mode():
        y, _ = layer(x.clone(), stub.PaddingMask(lens, 19))
        c = conv(x.clone(), stub.PaddingMask(lens, 19))
    np.savez_compressed(HERE / "fft_layer_ref.npz", x=x.numpy(), lens=lens.numpy(), layer_out=y.numpy(), conv_out=c.numpy(),
                        sd_sha256=np.array(sd_checksum({k: v for k, v in sd.items() if k.startswith(p)})))
    print("fft_layer_ref.npz", tuple(y.shape))


def main():
    import _fairseq2_stub as stub

    make_fbank()
    stub.install()
    cfg = tiny_config()
    load_ref("seamless_communication.models.unity.film", "models/unity/film.py")
    lr = load_ref("seamless_communication.models.unity.length_regulator", "models/unity/length_regulator.py")
    make_vocoder(cfg, lr)
    make_vocoder_dur(cfg, lr)
    make_unit_tokenizer()
    make_nar_frontend(cfg, lr, stub)
    make_fft_layer(cfg, stub)


if __name__ == "__main__":
    main()
