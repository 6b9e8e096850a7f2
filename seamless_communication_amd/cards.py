"""Asset-card data of the models on the S2ST path.

The reference keeps this in YAML cards consumed by fairseq2's asset store
(src/seamless_communication/cards/seamlessM4T_v2_large.yaml:7-51,
cards/unity_nllb-100.yaml, cards/vocoder_v2.yaml:7-201).  Only the fields the
hot path reads are restated here: language lists and the vocoder's
``lang_spkr_idx_map``.  A card dict loaded from a local YAML file (same
schema) can be passed instead, see :func:`load_card_file`.
"""
from __future__ import annotations

from typing import Any, Dict, List

# cards/unity_nllb-100.yaml `langs` (98 text languages, NLLB-100 subset).
TEXT_LANGS: List[str] = (
    "afr amh arb ary arz asm azj bel ben bos bul cat ceb ces ckb cmn cmn_Hant cym dan deu "
    "ell eng est eus fin fra fuv gaz gle glg guj heb hin hrv hun hye ibo ind isl ita jav jpn "
    "kan kat kaz khk khm kir kor lao lit lug luo lvs mai mal mar mkd mlt mni mya nld nno nob "
    "npi nya ory pan pbt pes pol por ron rus sat slk slv sna snd som spa srp swe swh tam tel "
    "tgk tgl tha tur ukr urd uzn vie yor yue zsm zul"
).split()

# cards/seamlessM4T_v2_large.yaml `unit_langs` (38) and `num_units`.
UNIT_LANGS: List[str] = (
    "arb ben cat ces cmn cym dan deu eng est fin fra hin ind ita jpn kan kor mlt nld pes pol "
    "por ron rus slk spa swe swh tam tel tgl tha tur ukr urd uzn vie"
).split()
NUM_UNITS = 10000

# cards/vocoder_v2.yaml model_config.lang_spkr_idx_map
_VOCODER_LANGS: List[str] = (
    "arb ben cat ces cmn cym dan deu eng est fin fra hin ind ita jpn kor mlt nld pes pol por "
    "ron rus slk spa swe swh tel tgl tha tur ukr urd uzn vie"
).split()
_VOCODER_SPKRS: Dict[str, List[int]] = {
    "arb": [0], "ben": [1], "cat": [2], "ces": [3], "cmn": [4, 5], "cym": [6], "dan": [7, 8],
    "deu": [9], "eng": [10], "est": [11, 12, 13], "fin": [14], "fra": [15], "hin": [16],
    "ind": [17, 24, 18, 20, 19, 21, 23, 27, 26, 22, 25], "ita": [29, 28], "jpn": [30],
    "kor": [31], "mlt": [32, 33, 34], "nld": [35, 37, 36], "pes": [38], "pol": [39],
    "por": [40], "ron": [41], "rus": [43, 42], "slk": [44], "spa": [45], "swe": [46, 48, 47],
    "swh": [49, 51, 50], "tel": [52], "tgl": [53], "tha": [54, 57, 58, 55, 56],
    "tur": [61, 60, 59], "ukr": [62], "urd": [63, 64, 65], "uzn": [66, 67, 68],
    "vie": [69, 70, 73, 74, 71, 72],
}


def vocoder_lang_spkr_idx_map() -> Dict[str, Any]:
    return {
        "multilingual": {l: i for i, l in enumerate(_VOCODER_LANGS)},
        "multispkr": {k: list(v) for k, v in _VOCODER_SPKRS.items()},
    }


def load_card_file(path: str) -> Dict[str, Any]:
    """Read one reference-schema YAML card from a local file (no asset store,
    no network); ``base:`` chains are resolved by the caller."""
    import yaml

    with open(path, "r", encoding="utf-8") as fp:
        return yaml.safe_load(fp)
