"""CPU: pins the oracle (oracle/*.py) against fixtures minted by the reference's
OWN code (tests/golden/make_reference_goldens.py: vocoder, unit tokenizer, NAR
decoder frontend incl. char-length rules / VariancePredictor / HardUpsampling /
duration rounding, Conv1dBlock + post-LN FFT layer).  Integer outputs must be
identical; float outputs within 2e-5 (same fp32 torch ops, different
association only)."""
import hashlib
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import unity as ou
from oracle import vocoder as ov
from seamless_communication_amd import cards, synthetic as syn
from seamless_communication_amd.config import tiny_config
from seamless_communication_amd.tokenizer import CharTokenizer, NllbTextTokenizer, UnitTokenizer

G = Path(__file__).resolve().parent / "golden"
FTOL = 2e-5


def _sha(sd) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def bundle():
    cfg = tiny_config()
    sd = syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED)
    vsd = syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED)
    tt = NllbTextTokenizer(cfg.text_vocab_size, cards.TEXT_LANGS)
    ct = CharTokenizer(cfg.char_vocab_size)
    return cfg, sd, vsd, tt, ct


def test_vocoder_matches_reference_vocoder(bundle):
    cfg, sd, vsd, tt, ct = bundle
    g = np.load(G / "vocoder_ref.npz")
    assert _sha(vsd) == str(g["sd_sha256"]), "synthetic weight generator drifted; re-mint the goldens"
    lang_idx, spkr_idx = ov.resolve_lang_spkr(cards.vocoder_lang_spkr_idx_map(), list(g["langs"]), [int(s) for s in g["spkrs"]])
    assert lang_idx == [11, 25] and spkr_idx == [15, 3]
    wav = ov.vocode(vsd, cfg.vocoder, torch.from_numpy(g["units"]), lang_idx, spkr_idx)
    assert tuple(wav.shape) == g["wav"].shape == (2, 1, 23 * 320)
    assert float((wav - torch.from_numpy(g["wav"])).abs().max()) < FTOL
    li, si = ov.resolve_lang_spkr(cards.vocoder_lang_spkr_idx_map(), ["eng"], [-1])
    w1 = ov.vocode(vsd, cfg.vocoder, torch.from_numpy(g["units"][:1]), li, si)
    assert float((w1 - torch.from_numpy(g["wav_single_eng"])).abs().max()) < FTOL


@pytest.mark.parametrize("tag,arch", [("ar", "seamlessM4T_large"), ("nar", "seamlessM4T_large_v2")])
def test_unit_tokenizer_matches_reference(tag, arch):
    g = np.load(G / "unit_tokenizer_ref.npz")
    tok = UnitTokenizer(100, ["eng", "deu", "fra"], arch)
    assert tok.vocab_info.size == int(g[f"{tag}_vocab_size"])
    assert [tok.lang_to_index(l) for l in ("eng", "deu", "fra")] == g[f"{tag}_lang_idx"].tolist()
    enc = tok.create_encoder("deu")(g[f"{tag}_units"].copy())
    assert np.array_equal(enc, g[f"{tag}_encoded"])
    dec = tok.create_decoder()(g[f"{tag}_tokens"].copy())
    assert np.array_equal(dec, g[f"{tag}_decoded"])


def test_char_rules_match_reference_frontend(bundle):
    cfg, sd, vsd, tt, ct = bundle
    g = np.load(G / "nar_frontend_ref.npz")
    text = torch.from_numpy(g["text_seqs"].copy())
    cs, csl, cl = ou.text_to_char_seqs(text, tt, ct, cfg.pad_idx, cfg.unk_idx, cfg.eos_idx)
    assert np.array_equal(cl.numpy(), g["char_lens"])
    assert np.array_equal(csl.numpy(), g["char_seq_lens"])
    assert np.array_equal(cs.numpy(), g["char_seqs"])
    # the table-driven form the HIP host side uses must agree as well
    tok_len, starts_sp, is_punc, offs, ids = tt.nar_tables(ct)
    for b in range(text.shape[0]):
        toks = [int(t) if int(t) != cfg.eos_idx else cfg.pad_idx for t in g["text_seqs"][b, 2:]]
        n = sum(t != cfg.pad_idx for t in toks)
        flat = []
        for i in range(n):
            t = toks[i]
            if t == cfg.unk_idx:
                want, chars = 1, [cfg.unk_idx]
            else:
                want = int(tok_len[t])
                nxt = i < n - 1 and bool(starts_sp[toks[i + 1]])
                if is_punc[t] and nxt:
                    want += 1
                elif i > 0 and is_punc[toks[i - 1]] and starts_sp[t]:
                    want -= 1
                chars = ids[offs[t]: offs[t + 1]].tolist()
            assert want == int(g["char_lens"][b, 1 + i])
            flat += chars
        assert flat == g["char_seqs"][b, : len(flat)].tolist()


@pytest.mark.parametrize("dfac,tag", [(1.0, "df1p0"), (1.3, "df1p3")])
def test_nar_frontend_matches_reference(bundle, dfac, tag):
    cfg, sd, vsd, tt, ct = bundle
    g = np.load(G / "nar_frontend_ref.npz")
    f = "t2u_model.decoder_frontend"
    assert _sha({k: v for k, v in sd.items() if k.startswith(f)}) == str(g["sd_sha256"])
    P = ou.Params(sd)
    char_pos = ou.sinusoidal_table(cfg.char_max_seq_len, cfg.model_dim, cfg.unit_pad_idx)
    unit_pos = ou.sinusoidal_table(cfg.unit_max_seq_len, cfg.model_dim, cfg.unit_pad_idx)
    seqs, unit_lens, dur, *_ = ou.nar_decoder_frontend(
        P, cfg, torch.from_numpy(g["enc_out"].copy()), torch.from_numpy(g["text_seqs"].copy()), tt, ct, dfac, char_pos, unit_pos)
    assert np.array_equal(dur.numpy(), g[f"{tag}_durations"])
    assert np.array_equal(unit_lens.numpy(), g[f"{tag}_unit_lens"])
    ref = torch.from_numpy(g[f"{tag}_seqs"])
    assert seqs.shape == ref.shape
    for b in range(seqs.shape[0]):
        n = int(unit_lens[b])
        assert float((seqs[b, :n] - ref[b, :n]).abs().max()) < 1e-4  # values O(30): embed_char * sqrt(M)


def test_variance_predictor_and_hard_upsampling(bundle):
    cfg, sd, vsd, tt, ct = bundle
    g = np.load(G / "nar_frontend_ref.npz")
    P = ou.Params(sd)
    logd = ou.variance_predictor(P, "t2u_model.decoder_frontend.variance_adaptor.duration_predictor",
                                 torch.from_numpy(g["vp_in"]), torch.from_numpy(g["vp_lens"]))
    assert float((logd - torch.from_numpy(g["vp_log_dur"])).abs().max()) < FTOL
    up, ul = ou.hard_upsample(torch.from_numpy(g["hu_in"]), torch.from_numpy(g["hu_dur"]))
    assert np.array_equal(ul.numpy(), g["hu_lens"]) and np.array_equal(up.numpy(), g["hu_out"])


def test_fft_layer_matches_reference_layer(bundle):
    cfg, sd, vsd, tt, ct = bundle
    g = np.load(G / "fft_layer_ref.npz")
    P = ou.Params(sd)
    y = ou.fft_layer(P, cfg, "t2u_model.decoder.layers.0", torch.from_numpy(g["x"]), torch.from_numpy(g["lens"]))
    ref = torch.from_numpy(g["layer_out"])
    for b, n in enumerate(g["lens"].tolist()):
        assert float((y[b, :n] - ref[b, :n]).abs().max()) < FTOL


def test_cxx_char_rules_match_reference_frontend(bundle):
    """The C++ restatement of the string rules that sc_t2u_nar runs (csrc/model_t2u.hip: text_to_char_seqs) — called on the
    CPU through its host-only C-ABI entry — against the characters the reference's own nar_decoder_frontend.py produced."""
    import ctypes as C

    from seamless_communication_amd import _lib

    cfg, sd, vsd, tt, ct = bundle
    g = np.load(G / "nar_frontend_ref.npz")
    lib = _lib.load_library()
    tok_len, starts_sp, is_punc, offs, ids = tt.nar_tables(ct)
    arr = [np.ascontiguousarray(tok_len.astype(np.int32)), np.ascontiguousarray(starts_sp.astype(np.uint8)),
           np.ascontiguousarray(is_punc.astype(np.uint8)), np.ascontiguousarray(offs.astype(np.int64)),
           np.ascontiguousarray(ids.astype(np.int32))]
    text = np.ascontiguousarray(g["text_seqs"].astype(np.int32))
    n, s_text = text.shape
    cap = 64
    char_lens = np.zeros((n, s_text), dtype=np.int32)
    char_ids = np.full((n, cap), -1, dtype=np.int32)
    seq_lens = np.zeros(n, dtype=np.int32)
    P = lambda a: C.c_void_p(a.ctypes.data)
    longest = lib.sc_text_to_char_seqs(len(tok_len), *[P(a) for a in arr], cfg.pad_idx, cfg.unk_idx, cfg.eos_idx, P(text), n, s_text,
                                       P(char_lens), P(char_ids), cap, P(seq_lens))
    assert longest == int(g["char_seq_lens"].max()), lib.sc_last_error()
    assert np.array_equal(char_lens, g["char_lens"])
    assert np.array_equal(seq_lens, g["char_seq_lens"])
    for b in range(n):
        assert char_ids[b, : seq_lens[b]].tolist() == g["char_seqs"][b, : seq_lens[b]].tolist()
    # capacity and geometry errors are reported, not written past
    assert lib.sc_text_to_char_seqs(len(tok_len), *[P(a) for a in arr], cfg.pad_idx, cfg.unk_idx, cfg.eos_idx, P(text), n, s_text,
                                    P(char_lens), P(char_ids), 3, P(seq_lens)) < 0
    assert b"capacity" in lib.sc_last_error()


def test_vocoder_duration_predictor_matches_reference():
    """CodeGenerator.forward(dur_prediction=True) (codehifigan.py:79-88), executed by the reference's own classes
    (tests/golden/make_reference_goldens.py: vocoder_dur_ref.npz): durations exact, waveform of the expanded units 2e-5."""
    from oracle import vocoder as ov
    from seamless_communication_amd import cards, synthetic as syn
    from seamless_communication_amd.config import tiny_config
    from tests.golden.make_reference_goldens import sd_checksum

    gold = np.load(G / "vocoder_dur_ref.npz")
    cfg = tiny_config()
    sd = syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED, with_dur_predictor=True)
    assert sd_checksum(sd) == str(gold["sd_sha256"])
    base = syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED)
    assert all(torch.equal(sd[k], v) for k, v in base.items())  # the predictor tensors are additions only
    lang_idx, spkr_idx = ov.resolve_lang_spkr(cards.vocoder_lang_spkr_idx_map(), ["fra"], [-1])
    for tag in ("a", "b"):
        units = torch.from_numpy(gold[f"{tag}_units"])
        dur = ov.vocoder_durations(sd, cfg.vocoder, units)
        assert dur.tolist() == gold[f"{tag}_dur"].tolist()
        assert dur.min() >= 1 and dur.max() > 1
        wav = ov.vocode(sd, cfg.vocoder, units, lang_idx, spkr_idx, dur_prediction=True)
        want = torch.from_numpy(gold[f"{tag}_wav"])
        assert wav.shape == want.shape == (1, 1, int(dur.sum()) * cfg.vocoder.hop)
        assert float((wav - want).abs().max()) < 2e-5
