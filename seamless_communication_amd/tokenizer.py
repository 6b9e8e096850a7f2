"""Host-side tokenizers of the S2ST path (integer / string logic only).

* :class:`UnitTokenizer` / :class:`UnitTokenEncoder` / :class:`UnitTokenDecoder` restate
  src/seamless_communication/models/unity/unit_tokenizer.py:15-243 (the drop-in surface of row a17) as a vocabulary
  LAYOUT with look-up tables built once per tokenizer; class / attribute names (the ``lang_symbol_repititions`` spelling
  included) and error texts are the reference's, the arithmetic is this package's; pinned against the executed
  reference class (tests/golden/unit_tokenizer_ref.npz).
* :class:`NllbTextTokenizer` exposes what the hot path needs from fairseq2's
  ``NllbTokenizer``: vocabulary info, ``index_to_token`` (used by
  nar_decoder_frontend.py:130-141), the target-mode prefix ``[</s>, __lang__]``
  and id -> text decoding.  It is backed either by a real SentencePiece model
  (``tokenizer.model`` of cards/unity_nllb-100.yaml) or, when no model file is
  reachable, by a deterministic synthetic vocabulary of the same layout.
* :class:`CharTokenizer` is the character SPM used by the NAR T2U frontend
  (``token_to_index`` only, nar_decoder_frontend.py:249-252).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

SPACE = "▁"  # nar_decoder_frontend.py:28


def _is_tensor(x) -> bool:
    return type(x).__module__.startswith("torch")


def _like(ref, arr: np.ndarray):
    import torch

    return torch.from_numpy(np.ascontiguousarray(arr)).to(ref.device)


@dataclass(frozen=True)
class VocabularyInfo:
    size: int
    unk_idx: Optional[int]
    bos_idx: Optional[int]
    eos_idx: Optional[int]
    pad_idx: Optional[int]


# --------------------------------------------------------------------------- #
# Units
# --------------------------------------------------------------------------- #
# The unit vocabulary as a LAYOUT (what models/unity/unit_tokenizer.py:15-117 computes with index arithmetic at every
# call): four control symbols, the speech units, then `blocks` language blocks of len(langs) + 1 symbols each - the v1
# (autoregressive) models carry two blocks and use the second one, the "_v2" (NAR) models one.
#
#     [ <s>=0 <pad>=1 </s>=2 <unk>=3 | unit 0 .. unit U-1 | block 0: langs.., spare | block 1: langs.., spare ]
#
# Encoding and decoding are then table look-ups built once per tokenizer (`encode_lut` over raw unit values, `decode_lut`
# over vocabulary ids) instead of per-call masked arithmetic.  Public names, signatures and error texts are the
# reference's (they are the drop-in surface of row a17); results are pinned against the executed reference class
# (tests/golden/unit_tokenizer_ref.npz, tests/test_unit_tokenizer.py).
_N_CONTROL = 4


class UnitTokenizer:
    """Drop-in for ``UnitTokenizer`` (unit_tokenizer.py:15-117)."""

    def __init__(self, num_units: int, langs: Sequence[str], model_arch: str) -> None:
        self.num_units = num_units
        self.langs = list(langs)
        self.lang_map = {lang: idx for idx, lang in enumerate(langs)}
        self.is_nar_decoder = model_arch.rsplit("_", 1)[-1] == "v2"   # unit_tokenizer.py:38
        blocks = 1 if self.is_nar_decoder else 2
        self.lang_symbol_repititions = blocks                           # the reference's attribute name (and spelling)
        block = len(self.langs) + 1
        self._first_lang = _N_CONTROL + num_units + (blocks - 1) * block  # first language symbol of the block in use
        self.vocab_info = VocabularyInfo(size=_N_CONTROL + num_units + blocks * block, bos_idx=0, pad_idx=1, eos_idx=2, unk_idx=3)
        # raw unit value -> vocabulary id; one extra slot catches everything outside the dictionary (-> <unk>)
        self.encode_lut = np.concatenate([np.arange(num_units, dtype=np.int64) + _N_CONTROL, [self.vocab_info.unk_idx]])
        # vocabulary id -> id with </s> folded into <pad> and <pad> moved out of the way of unit 0 (unit_tokenizer.py:232-239:
        # after the -4 below the pad lands on 1 again, every unit on its raw value)
        self.decode_lut = np.arange(self.vocab_info.size, dtype=np.int64)
        self.decode_lut[[self.vocab_info.eos_idx, self.vocab_info.pad_idx]] = self.vocab_info.pad_idx + _N_CONTROL

    def _unknown_lang(self, lang: str) -> ValueError:
        return ValueError(f"`lang` must be one of the supported languages, but is '{lang}' instead. Supported languages: "
                          f"{', '.join(self.langs)}")

    def lang_to_index(self, lang: str) -> int:
        if lang not in self.lang_map:
            raise self._unknown_lang(lang)
        return self._first_lang + self.lang_map[lang]

    def index_to_lang(self, idx: int) -> str:
        k = idx - self._first_lang
        if not 0 <= k < len(self.langs):
            raise ValueError(f"`idx` must correspond to one of the supported language symbol indices (0 to {len(self.langs) - 1}), "
                             f"but is {idx} instead.")
        return self.langs[k]

    def create_encoder(self, lang: str, device=None) -> "UnitTokenEncoder":
        """``device`` is accepted for signature parity (unit_tokenizer.py:96-107); tensors come back on the input's device."""
        return UnitTokenEncoder(self, lang, self.is_nar_decoder)

    def create_decoder(self) -> "UnitTokenDecoder":
        return UnitTokenDecoder(self, self.is_nar_decoder)


class UnitTokenEncoder:
    """Raw units (N, S) -> model ids; the autoregressive models get the prompt ``[</s>, __lang__]`` in front
    (unit_tokenizer.py:120-206)."""

    def __init__(self, tokenizer: UnitTokenizer, lang: str, is_nar_decoder: bool) -> None:
        if lang not in tokenizer.lang_map:
            raise tokenizer._unknown_lang(lang)
        self.tokenizer = tokenizer
        self.is_nar_decoder = is_nar_decoder
        self.eos_idx = tokenizer.vocab_info.eos_idx
        self.unk_idx = tokenizer.vocab_info.unk_idx
        self.lang_idx = tokenizer.lang_to_index(lang)
        self.prefix_indices = None if is_nar_decoder else np.array([self.eos_idx, self.lang_idx], dtype=np.int64)

    def __call__(self, units):
        """int64 array or tensor (N, S) -> same kind, (N, S [+2])."""
        if _is_tensor(units):
            return _like(units, self(units.detach().cpu().numpy()))
        units = np.asarray(units, dtype=np.int64)
        lut = self.tokenizer.encode_lut
        ids = lut[np.minimum(units, len(lut) - 1)]   # values past the dictionary share the <unk> slot
        if self.prefix_indices is None:
            return ids
        return np.concatenate([np.broadcast_to(self.prefix_indices, (units.shape[0], 2)), ids], axis=1)


class UnitTokenDecoder:
    """Model ids -> raw units with pads (value 1) where the sequence has ended (unit_tokenizer.py:209-243).  The NAR models'
    rows are all units; the autoregressive models' rows start [</s>, __lang__]: the </s> column is dropped and the language
    column comes back as the reference leaves it (un-shifted, only </s> / <pad> folded)."""

    def __init__(self, tokenizer: UnitTokenizer, is_nar_decoder: bool) -> None:
        self.eos_idx = tokenizer.vocab_info.eos_idx
        self.pad_idx = tokenizer.vocab_info.pad_idx
        self.is_nar_decoder = is_nar_decoder
        self._lut = tokenizer.decode_lut

    def __call__(self, token_indices):
        if _is_tensor(token_indices):
            return _like(token_indices, self(token_indices.detach().cpu().numpy()))
        token_indices = np.asarray(token_indices, dtype=np.int64)
        if token_indices.shape[1] == 0:
            return token_indices
        if self.is_nar_decoder:
            return self._lut[token_indices] - _N_CONTROL
        folded = self._lut[token_indices[:, 1:]]
        folded[:, 1:] -= _N_CONTROL
        return folded


# --------------------------------------------------------------------------- #
# Text
# --------------------------------------------------------------------------- #
_PUNCT = [",", ".", "!", "?", ";", ":", "-", "'", '"', "(", ")"]


def _mix(i: int) -> int:
    x = (i * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 29
    x = (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 32
    return x


def synthetic_piece(i: int) -> str:
    """Deterministic pseudo sentence piece for vocabulary slot ``i``."""
    h = _mix(i)
    kind = h % 100
    if kind < 3:
        return _PUNCT[(h >> 8) % len(_PUNCT)]
    if kind < 4:
        return SPACE
    n = 1 + (h >> 8) % 6
    s = "".join(chr(ord("a") + ((h >> (16 + 5 * k)) % 26)) for k in range(n))
    if kind < 60:
        s = SPACE + s
    return s


class NllbTextTokenizer:
    """NLLB-layout text vocabulary.

    Layout (fairseq2 NllbTokenizer: ``<pad>@0`` control symbol in front of the
    SentencePiece model, language + data-source control symbols appended):
    ``<pad>=0 <unk>=1 <s>=2 </s>=3``, pieces, ``__lang__`` x len(langs),
    ``<MINED_DATA> <MMT_BT_DATA> <SMT_BT_DATA>``, padding pieces up to ``size``.
    """

    def __init__(
        self,
        size: int,
        langs: Sequence[str],
        default_lang: str = "eng",
        spm_path: Optional[str] = None,
    ) -> None:
        self.langs = list(langs)
        self.default_lang = default_lang
        self.vocab_info = VocabularyInfo(size=size, unk_idx=1, bos_idx=2, eos_idx=3, pad_idx=0)
        extra = [f"__{l}__" for l in self.langs] + ["<MINED_DATA>", "<MMT_BT_DATA>", "<SMT_BT_DATA>"]
        self._spm = None
        if spm_path is not None:
            import sentencepiece as spm

            self._spm = spm.SentencePieceProcessor(model_file=spm_path)
            n_spm = self._spm.get_piece_size()
            # SPM ids 0..2 are <unk>,<s>,</s>; fairseq2 inserts <pad> at 0.
            pieces = ["<pad>"] + [self._spm.id_to_piece(i) for i in range(n_spm)]
            self._first_lang = len(pieces)
            pieces += extra
        else:
            n_pieces = size - len(extra) - 4
            if n_pieces < 16:
                raise ValueError("text vocabulary too small for the control symbols")
            pieces = ["<pad>", "<unk>", "<s>", "</s>"] + [synthetic_piece(i) for i in range(4, 4 + n_pieces)]
            self._first_lang = len(pieces)
            pieces += extra
        while len(pieces) < size:
            pieces.append(f"<extra_{len(pieces)}>")
        self._pieces: List[str] = pieces[:size]
        self._index: Optional[Dict[str, int]] = None
        self._lang_idx = {l: self._first_lang + i for i, l in enumerate(self.langs)}

    # -- fairseq2 SentencePieceModel surface used by the reference ---------- #
    def index_to_token(self, idx: int) -> str:
        return self._pieces[idx]

    def token_to_index(self, tok: str) -> int:
        if self._index is None:
            self._index = {}
            for i, p in enumerate(self._pieces):
                self._index.setdefault(p, i)
        return self._index.get(tok, self.vocab_info.unk_idx)

    def lang_token_idx(self, lang: str) -> int:
        if lang not in self._lang_idx:
            raise ValueError(
                f"`lang` must be a supported language, but is '{lang}' instead."
            )
        return self._lang_idx[lang]

    # -- fairseq2 NllbTokenizer.create_encoder (call site translator.py:299-302) ---------- #
    def create_encoder(self, *, task: Optional[str] = None, lang: Optional[str] = None, mode: Optional[str] = None,
                       device=None, pin_memory: bool = False) -> "TextTokenEncoder":
        """NLLB conventions (fairseq2 0.2 models/nllb/tokenizer.py, restated): ``source`` (default) =
        ``[__lang__] pieces [</s>]``; ``target`` = ``[</s>, __lang__] pieces [</s>]``; ``source_mining`` /
        ``source_mmt_bt`` / ``source_smt_bt`` insert the data-source tag after the language."""
        if task is not None and task != "translation":
            raise ValueError(f"`task` must be 'translation', but is '{task}' instead.")
        lang = lang or self.default_lang
        lang_idx = self.lang_token_idx(lang)
        eos = self.vocab_info.eos_idx
        tags = {"source_mining": "<MINED_DATA>", "source_mmt_bt": "<MMT_BT_DATA>", "source_smt_bt": "<SMT_BT_DATA>"}
        if mode is None or mode == "source":
            prefix = [lang_idx]
        elif mode in tags:
            prefix = [lang_idx, self.token_to_index(tags[mode])]
        elif mode == "target":
            prefix = [eos, lang_idx]
        else:
            raise ValueError(
                f"`mode` must be 'source', 'source_mining', 'source_mmt_bt', 'source_smt_bt', or 'target', but is '{mode}' instead."
            )
        return TextTokenEncoder(self, prefix, [eos], device)

    def encode_pieces(self, text: str) -> List[int]:
        """text -> piece ids (no control symbols).  With a SentencePiece model: its own segmentation, ids shifted
        by the ``<pad>@0`` slot.  Synthetic vocabulary: SentencePiece-style normalisation (dummy prefix, spaces ->
        ``▁``) and greedy longest match; characters outside the vocabulary -> ``<unk>``."""
        if self._spm is not None:
            return [int(i) + 1 for i in self._spm.encode(text)]
        norm = SPACE + SPACE.join(text.split())
        if norm == SPACE:
            return []
        self.token_to_index(SPACE)  # builds the index
        first_ctrl = self._first_lang
        ids: List[int] = []
        pos = 0
        while pos < len(norm):
            for n in range(min(8, len(norm) - pos), 0, -1):
                i = self._index.get(norm[pos : pos + n])
                if i is not None and 4 <= i < first_ctrl:
                    ids.append(i)
                    pos += n
                    break
            else:
                ids.append(self.vocab_info.unk_idx)
                pos += 1
        return ids

    def target_prefix(self, lang: str) -> List[int]:
        """NLLB "target" mode prefix: ``[</s>, __lang__]``."""
        return [self.vocab_info.eos_idx, self.lang_token_idx(lang)]

    def decode(self, ids: Sequence[int]) -> str:
        """ids -> text; control symbols are skipped (SentencePiece decode).  With a SentencePiece model the pieces go
        through its own decoder (normalisation rules, byte pieces); the synthetic vocabulary joins pieces directly."""
        first, last = self._first_lang, self._first_lang + len(self.langs) + 3
        keep = []
        for i in ids:
            i = int(i)
            if i in (0, 2, 3) or first <= i < last or i >= last:
                continue
            keep.append(i)
        if self._spm is not None:
            return self._spm.decode([i - 1 for i in keep])  # ids are shifted by the <pad>@0 slot
        toks = [" ⁇ " if i == 1 else self._pieces[i] for i in keep]
        return "".join(toks).replace(SPACE, " ").strip()

    # -- per-vocabulary tables for the NAR frontend (built once) ------------ #
    def nar_tables(self, char_tokenizer: "CharTokenizer"):
        """Vectorised form of the per-token string rules of
        nar_decoder_frontend.py:158-259: for every vocabulary entry its length,
        "starts with SPACE and longer than one char", "is punctuation", and the
        char-id sequence (CSR)."""
        if getattr(self, "_nar_tables", None) is None:
            n = len(self._pieces)
            tok_len = np.zeros(n, dtype=np.int64)
            starts_sp = np.zeros(n, dtype=bool)
            is_punc = np.zeros(n, dtype=bool)
            offs = np.zeros(n + 1, dtype=np.int64)
            ids: List[int] = []
            for i, p in enumerate(self._pieces):
                tok_len[i] = len(p)
                starts_sp[i] = len(p) > 1 and p[0] == SPACE
                is_punc[i] = (
                    len(p) == 1 and not p.isalpha() and not p.isnumeric() and p != SPACE
                )
                ids.extend(char_tokenizer.token_to_index(ch) for ch in p)
                offs[i + 1] = len(ids)
            self._nar_tables = (tok_len, starts_sp, is_punc, offs, np.asarray(ids, dtype=np.int64))
        return self._nar_tables


class TextTokenEncoder:
    """Callable ``text -> 1-D int64 tensor`` (fairseq2 ``TextTokenEncoder``): prefix + pieces + suffix."""

    def __init__(self, tokenizer: NllbTextTokenizer, prefix: Sequence[int], suffix: Sequence[int], device=None) -> None:
        self.tokenizer = tokenizer
        self.prefix_indices = list(prefix)
        self.suffix_indices = list(suffix)
        self.device = device

    def __call__(self, text: str):
        import torch

        ids = self.prefix_indices + self.tokenizer.encode_pieces(text) + self.suffix_indices
        return torch.tensor(ids, dtype=torch.int64, device=self.device)


class CharTokenizer:
    """Character vocabulary (fairseq control order bos=0,pad=1,eos=2,unk=3;
    t2u_builder.py:229 ``char_pad_idx=1``)."""

    def __init__(self, size: int, spm_path: Optional[str] = None) -> None:
        self.vocab_info = VocabularyInfo(size=size, bos_idx=0, pad_idx=1, eos_idx=2, unk_idx=3)
        self._spm = None
        if spm_path is not None:
            import sentencepiece as spm

            self._spm = spm.SentencePieceProcessor(model_file=spm_path)
            self._map = None
        else:
            chars = [SPACE] + [chr(ord("a") + k) for k in range(26)] + _PUNCT
            if 4 + len(chars) > size:
                raise ValueError("char vocabulary too small")
            self._map = {c: 4 + i for i, c in enumerate(chars)}

    def pieces(self) -> Optional[List[str]]:
        """The pieces in model order: what ``_get_char_index_mapping`` (models/unity/loader.py:158-176) re-orders the char
        embedding of a fairseq-keyed checkpoint by.  ``None`` without a SentencePiece model: a published fairseq-keyed
        checkpoint loaded from a card that names no ``char_tokenizer_path`` must fail in ``convert_unity_checkpoint`` instead
        of having the first rows of its real ``embed_char`` table permuted by the built-in alphabet."""
        if self._spm is None:
            return None
        return [self._spm.id_to_piece(i) for i in range(self._spm.get_piece_size())]

    def synthetic_pieces(self) -> List[str]:
        """The built-in alphabet as a piece list - only for fairseq-layout renditions of the SYNTHETIC weights
        (scripts/real_layout_check.py and ``synthetic://`` / ``file://...?synthetic_chars=1`` cards)."""
        if self._spm is not None:
            raise ValueError("synthetic_pieces() is the built-in alphabet's list; this tokenizer wraps a SentencePiece model")
        by_index = sorted(self._map.items(), key=lambda kv: kv[1])
        return ["<s>", "<pad>", "</s>", "<unk>"] + [c for c, _ in by_index]

    def token_to_index(self, ch: str) -> int:
        if self._spm is not None:
            return int(self._spm.piece_to_id(ch))
        return self._map.get(ch, self.vocab_info.unk_idx)
