"""ORACLE-SIDE checker (test infrastructure, never shipped or benchmarked as the product).

ctypes driver of ``oracle/_ref/libggml_ref.so``: the reference's OWN native restatement of the
fairseq2 modules (``/root/reference/ggml/examples/unity/fairseq2.cpp`` on the reference's ggml
fork), compiled from where the sources lie by ``oracle/build_ref.sh`` with the glue
``oracle/ggml_ref_wrap.cc``.  It exists to pin ``oracle/unity.py``: the reference Python path needs
fairseq2 0.2 (not installed, not vendored), but this C++ restatement ships in the reference tree,
is what the reference's own ``ggml/test_unity_cpp.py`` compares against fairseq2, and runs here.

Tensors are handed over in the fairseq2 state-dict naming (the names ``fairseq2.cpp`` looks up) with
the conventions of the reference's converter ``ggml/ggml_convert.py``:
  * 1-D ``.bias`` tensors are stored as (1, n) unless the key contains "adaptor" (:519-522);
  * LayerNorm eps / attention num_heads / layer norm_order live in ``layer_config`` (:404-470);
  * the embedding scale sqrt(model_dim) is baked into ``*.embed.weight`` (:371-380) and the
    sinusoidal table is stored as ``<frontend>.pos_encoder`` (:382-402).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

LIB_PATH = Path(__file__).resolve().parent / "_ref" / "libggml_ref.so"
NORM_ORDER_POST, NORM_ORDER_PRE = 0, 1  # fairseq2.h:262-266

_P = C.c_void_p
_lib: Optional[C.CDLL] = None


def available() -> bool:
    return LIB_PATH.exists()


def _load() -> C.CDLL:
    global _lib
    if _lib is None:
        lib = C.CDLL(str(LIB_PATH))
        lib.gref_new.restype = _P
        lib.gref_new.argtypes = [C.c_int64]
        lib.gref_free.argtypes = [_P]
        lib.gref_add_tensor.argtypes = [_P, C.c_char_p, C.c_int, _P, _P]
        lib.gref_set_int.argtypes = [_P, C.c_char_p, C.c_int64]
        lib.gref_set_double.argtypes = [_P, C.c_char_p, C.c_double]
        lib.gref_add_token.argtypes = [_P, C.c_char_p, C.c_int]
        lib.gref_forward.restype = C.c_int64
        lib.gref_forward.argtypes = [_P, C.c_char_p, C.c_char_p, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, _P, C.c_int64,
                                     _P, _P, C.c_int64, C.c_int]
        lib.gref_embed.restype = C.c_int64
        lib.gref_embed.argtypes = [_P, C.c_char_p, _P, C.c_int, _P, C.c_int64, C.c_int64]
        lib.gref_generate.argtypes = [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                      C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      _P, C.c_int, _P, _P, _P]
        lib.gref_generate_all.argtypes = [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                          C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          _P, C.c_int, _P, _P]
        lib.gref_tweak_lprobs.argtypes = [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        _lib = lib
    return _lib


class GgmlRef:
    """One ``fairseq2_model`` of the reference's C++ restatement, filled from a state dict."""

    def __init__(self, tensor_mem_mb: int = 256) -> None:
        self.lib = _load()
        self.h = self.lib.gref_new(int(tensor_mem_mb) << 20)

    def close(self) -> None:
        if self.h:
            self.lib.gref_free(self.h)
            self.h = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- model construction ------------------------------------------------------------
    def add_tensor(self, name: str, t: torch.Tensor) -> None:
        a = np.ascontiguousarray(t.detach().to(torch.float32).numpy())
        shape = (C.c_int64 * a.ndim)(*a.shape)
        rc = self.lib.gref_add_tensor(self.h, name.encode(), a.ndim, shape, a.ctypes.data_as(_P))
        assert rc == 0, name

    def add_state_dict(self, sd: Dict[str, torch.Tensor], prefix_filter: Sequence[str] = ("",)) -> None:
        """Registers tensors under their fairseq2 names with ggml_convert.py's layout conventions."""
        for k, v in sd.items():
            if not any(k.startswith(p) for p in prefix_filter):
                continue
            v = v.detach().to(torch.float32)
            if k.endswith(".bias") and v.dim() == 1 and "adaptor" not in k:
                v = v.reshape(1, -1)
            if "pointwise_conv" in k:
                v = v.squeeze(-1)
            if "depthwise_conv" in k:
                v = v.squeeze(1)
            self.add_tensor(k, v)

    def set_layer_norm_eps(self, prefix: str, eps: float = 1e-5) -> None:
        self.lib.gref_set_double(self.h, (prefix + ".eps").encode(), float(eps))

    def set_num_heads(self, prefix: str, heads: int) -> None:
        self.lib.gref_set_int(self.h, (prefix + ".num_heads").encode(), int(heads))

    def set_norm_order(self, prefix: str, order: int) -> None:
        self.lib.gref_set_int(self.h, (prefix + ".norm_order").encode(), int(order))

    def configure(self, sd: Dict[str, torch.Tensor], num_heads: int, norm_order: int = NORM_ORDER_PRE, eps: float = 1e-5) -> None:
        """Derives the layer_config entries fairseq2.cpp reads from the key names of ``sd``."""
        for k in sd:
            if k.endswith("layer_norm.weight") or k.endswith("_norm.weight") or k.endswith(".ln1.weight"):
                self.set_layer_norm_eps(k[: -len(".weight")], eps)
            if k.endswith(".q_proj.weight"):
                self.set_num_heads(k[: -len(".q_proj.weight")], num_heads)
            if k.endswith(".self_attn_layer_norm.weight"):
                self.set_norm_order(k[: -len(".self_attn_layer_norm.weight")], norm_order)

    def add_token(self, token: str, idx: int) -> None:
        self.lib.gref_add_token(self.h, token.encode(), int(idx))

    # ---- forward calls -------------------------------------------------------------------
    def forward(self, kind: str, prefix: str, x: torch.Tensor, y: Optional[torch.Tensor] = None, causal: bool = False,
                mem_mb: int = 256, threads: int = 4) -> torch.Tensor:
        xa = np.ascontiguousarray(x.detach().to(torch.float32).numpy())
        xs = (C.c_int64 * xa.ndim)(*xa.shape)
        if y is not None:
            ya = np.ascontiguousarray(y.detach().to(torch.float32).numpy())
            ys = (C.c_int64 * ya.ndim)(*ya.shape)
            yp, ynd = ya.ctypes.data_as(_P), ya.ndim
        else:
            ya, ys, yp, ynd = None, None, None, 0
        cap = max(xa.size, ya.size if ya is not None else 0) * 64 + 4096
        out = np.empty(cap, dtype=np.float32)
        oshape = (C.c_int64 * 4)()
        ond = C.c_int(0)
        n = self.lib.gref_forward(self.h, kind.encode(), prefix.encode(), xa.ctypes.data_as(_P), xs, xa.ndim, yp, ys, ynd,
                                  int(causal), out.ctypes.data_as(_P), cap, oshape, C.byref(ond), mem_mb, threads)
        if n <= 0:
            raise RuntimeError(f"gref_forward({kind}, {prefix}) failed: {n}")
        shape = tuple(oshape[i] for i in range(ond.value))
        return torch.from_numpy(out[:n].copy()).reshape(shape)

    def embed(self, prefix: str, tokens: Sequence[int], model_dim: int, mem_mb: int = 64) -> torch.Tensor:
        tok = np.ascontiguousarray(np.asarray(tokens, dtype=np.int32))
        out = np.empty(tok.size * model_dim, dtype=np.float32)
        n = self.lib.gref_embed(self.h, prefix.encode(), tok.ctypes.data_as(_P), tok.size, out.ctypes.data_as(_P), out.size,
                                mem_mb)
        if n != out.size:
            raise RuntimeError(f"gref_embed failed: {n}")
        return torch.from_numpy(out).reshape(tok.size, model_dim)

    def generate(self, enc: torch.Tensor, prefix: Sequence[int], beam_size: int = 1, soft_max_seq_len=(1, 200),
                 hard_max_seq_len: int = 1024, min_seq_len: int = 1, len_penalty: float = 1.0, unk_penalty: float = 0.0,
                 normalize_scores: bool = True, pad_idx: int = 0, unk_idx: int = 1, bos_idx: int = 2, eos_idx: int = 3,
                 mem_mb: int = 256, threads: int = 4) -> Tuple[list, float, np.ndarray]:
        """``generate_sequence`` for ONE utterance; enc (S_enc, M).  Returns (ids, score, step_scores)."""
        ea = np.ascontiguousarray(enc.detach().to(torch.float32).numpy())
        pre = np.ascontiguousarray(np.asarray(prefix, dtype=np.int32))
        cap = int(hard_max_seq_len) + 8
        ids = np.zeros(cap, dtype=np.int32)
        steps = np.zeros(cap, dtype=np.float32)
        ln, sc = C.c_int(0), C.c_float(0)
        rc = self.lib.gref_generate(self.h, ea.ctypes.data_as(_P), ea.shape[0], ea.shape[1], pre.ctypes.data_as(_P), pre.size,
                                    beam_size, float(soft_max_seq_len[0]), int(soft_max_seq_len[1]), int(hard_max_seq_len),
                                    int(min_seq_len), float(len_penalty), float(unk_penalty), int(normalize_scores), pad_idx,
                                    unk_idx, bos_idx, eos_idx, mem_mb, threads, ids.ctypes.data_as(_P), cap, C.byref(ln),
                                    C.byref(sc), steps.ctypes.data_as(_P))
        if rc != 0:
            raise RuntimeError("gref_generate returned no hypothesis")
        return ids[: ln.value].tolist(), float(sc.value), steps[: ln.value].copy()

    def generate_all(self, enc: torch.Tensor, prefix: Sequence[int], beam_size: int, soft_max_seq_len=(1, 200),
                     hard_max_seq_len: int = 1024, min_seq_len: int = 1, len_penalty: float = 1.0, unk_penalty: float = 0.0,
                     normalize_scores: bool = True, pad_idx: int = 0, unk_idx: int = 1, bos_idx: int = 2, eos_idx: int = 3,
                     mem_mb: int = 256, threads: int = 4):
        """``generate_sequence`` for ONE utterance, every finished hypothesis: [(score, ids)], best first."""
        ea = np.ascontiguousarray(enc.detach().to(torch.float32).numpy())
        pre = np.ascontiguousarray(np.asarray(prefix, dtype=np.int32))
        cap = int(hard_max_seq_len) + 8
        ids = np.zeros((beam_size, cap), dtype=np.int32)
        lens = np.zeros(beam_size, dtype=np.int32)
        scores = np.zeros(beam_size, dtype=np.float32)
        rc = self.lib.gref_generate_all(self.h, ea.ctypes.data_as(_P), ea.shape[0], ea.shape[1], pre.ctypes.data_as(_P), pre.size,
                                        beam_size, float(soft_max_seq_len[0]), int(soft_max_seq_len[1]), int(hard_max_seq_len),
                                        int(min_seq_len), float(len_penalty), float(unk_penalty), int(normalize_scores), pad_idx,
                                        unk_idx, bos_idx, eos_idx, mem_mb, threads, ids.ctypes.data_as(_P), cap,
                                        lens.ctypes.data_as(_P), scores.ctypes.data_as(_P))
        if rc < 0:
            raise RuntimeError("gref_generate_all failed")
        return [(float(scores[b]), ids[b, : lens[b]].tolist()) for b in range(beam_size) if lens[b] > 0]


def tweak_lprobs(lprobs: torch.Tensor, step_nr: int, max_seq_len: int, min_seq_len: int, unk_penalty: float, pad_idx: int,
                 unk_idx: int, eos_idx: int) -> torch.Tensor:
    """The reference's compiled ``_tweak_lprobs`` on a (beam, V) tensor; returns the edited copy."""
    a = np.ascontiguousarray(lprobs.detach().to(torch.float32).numpy()).copy()
    rc = _load().gref_tweak_lprobs(a.ctypes.data_as(_P), a.shape[0], a.shape[1], step_nr, max_seq_len, min_seq_len, float(unk_penalty),
                                   pad_idx, unk_idx, eos_idx)
    if rc != 0:
        raise RuntimeError("gref_tweak_lprobs failed")
    return torch.from_numpy(a)
