"""Loader / comparator for tests/golden/fullsize_ref.json (minted by make_fullsize_goldens.py from the CPU oracle on the
BASELINE configuration).  Pure JSON + integer comparison: used by tests/test_fullsize_gpu.py and by bench.py's parity
block (neither the oracle nor the reference is needed at run time).

Bar: text ids, char ids and durations bit-exact; unit ids bit-exact except where the ORACLE's own
arg-max margin (top-1 minus top-2 logit) is below UNIT_MARGIN_TOL - fp32 re-association noise of two correct fp32
evaluations is of that order, a mismatch on a wider margin is a defect.  Every mismatch is reported with its margin.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, List, Optional, Sequence

GOLDEN = Path(__file__).resolve().parent / "fullsize_ref.json"
GOLDEN_MORE = Path(__file__).resolve().parent / "fullsize_more_ref.json"  # make_fullsize_more_goldens.py
UNIT_MARGIN_TOL = 1e-4   # logit units; the oracle's unit logits are O(10)
TEXT_MARGIN_TOL = 1e-4   # log-probability units


def load(path: Path = GOLDEN) -> Dict:
    return json.loads(Path(path).read_text())


def items_by_index(section: Dict) -> Dict[int, Dict]:
    return {int(r["index"]): r for r in section["items"]}


def ragged_items(gold_more: Dict, hard_max_seq_len: int):
    """The oracle's rows of the ragged (`eos_ramp`) 64-utterance batch under a given hard_max_seq_len: ({index: row}, name of
    the fixture) or (None, why not).  64: section b64eos as minted.  Any limit above the longest hypothesis that runs to its own
    EOS (72 tokens) - the reference default 1024 included; the soft rule (1, 200) never binds for 10 s of audio, fairseq2
    applies it to the ~1000 fbank frames -: b64eos with the rows it cut at 64 replaced by their b64long versions (a greedy row
    depends on the limit only through the forced EOS at max_len - 2, so every other row is the same;
    make_fullsize_more_goldens.py asserts that on a cross-check row)."""
    base = items_by_index(gold_more["b64eos"])
    cut_len = int(gold_more["meta"]["eos_text_len"])
    if int(hard_max_seq_len) == cut_len:
        return base, "tests/golden/fullsize_more_ref.json: b64eos"
    long_sec = gold_more.get("b64long")
    if long_sec is None:
        return None, "the golden fixture has no b64long section (tests/golden/make_fullsize_more_goldens.py --sections b64long)"
    longer = items_by_index(long_sec)
    longest = max(len(r["text_ids"]) for r in longer.values())
    if int(hard_max_seq_len) <= longest:
        return None, f"the golden fixture holds hard_max_seq_len {cut_len} and > {longest} (no row cut) only"
    assert set(long_sec["cut_at_64"]) == {i for i, r in base.items() if len(r["text_ids"]) >= cut_len}
    merged = dict(base)
    merged.update({i: longer[i] for i in long_sec["cut_at_64"]})
    return merged, "tests/golden/fullsize_more_ref.json: b64eos + b64long (no row cut)"


def _first_diff(a: Sequence[int], b: Sequence[int]) -> Optional[int]:
    for i, (x, y) in enumerate(zip(a, b)):
        if int(x) != int(y):
            return i
    return None if len(a) == len(b) else min(len(a), len(b))


def compare(golden: Dict, text_ids: Optional[Sequence[int]] = None, units: Optional[Sequence[int]] = None,
            char_ids: Optional[Sequence[int]] = None, durations: Optional[Sequence[int]] = None) -> Dict:
    """One utterance against its golden record.  Returns {"text": bool, "units": bool, "chars": bool, "durations": bool,
    "text_first_diff", "text_margin_at_diff", "unit_mismatches": [(pos, margin)], "unit_positions": n, "unit_len_equal"}
    (keys only for what was passed)."""
    out: Dict = {"index": int(golden["index"])}
    if text_ids is not None:
        d = _first_diff(list(text_ids), golden["text_ids"])
        out["text"] = d is None
        if d is not None:
            out["text_first_diff"] = d
            tm = golden.get("text_margins")
            # margins start at the first generated token (the prompt is echoed: 2 tokens for "</s> __lang__")
            k = d - (len(golden["text_ids"]) - len(tm)) if tm else -1
            out["text_margin_at_diff"] = tm[k] if tm and 0 <= k < len(tm) else None
    if char_ids is not None:
        out["chars"] = [int(c) for c in char_ids] == golden["char_ids"]
    if durations is not None:
        out["durations"] = [int(c) for c in durations] == golden["durations"]
    if units is not None:
        g = golden["units"]
        u = [int(x) for x in units]
        out["unit_len_equal"] = len(u) == len(g)
        out["unit_positions"] = len(g)
        mm = [(i, golden["unit_margins"][i]) for i in range(min(len(u), len(g))) if u[i] != g[i]]
        out["unit_mismatches"] = mm
        out["units"] = out["unit_len_equal"] and not mm
    return out


def summarize(reports: List[Dict]) -> Dict:
    """Match rates over a set of compare() reports + the verdict under the margin rule."""
    n = len(reports)
    s: Dict = {"n_checked": n, "utterances": [r["index"] for r in reports]}
    if any("text" in r for r in reports):
        bad = [r for r in reports if not r.get("text", True)]
        s["text_match"] = f"{n - len(bad)}/{n}"
        s["text_mismatches"] = [{"index": r["index"], "first_diff": r.get("text_first_diff"), "oracle_margin": r.get("text_margin_at_diff")}
                                for r in bad]
    if any("units" in r for r in reports):
        bad = [r for r in reports if not r.get("units", True)]
        pos = sum(r.get("unit_positions", 0) for r in reports)
        mm = [(r["index"], p, m) for r in reports for (p, m) in r.get("unit_mismatches", [])]
        s["unit_match"] = f"{n - len(bad)}/{n}"
        s["unit_positions"] = pos
        s["unit_position_mismatches"] = len(mm)
        s["unit_len_mismatches"] = [r["index"] for r in reports if not r.get("unit_len_equal", True)]
        s["unit_mismatch_list"] = [{"index": i, "pos": p, "oracle_margin": m} for i, p, m in mm[:32]]
        s["max_margin_of_a_unit_mismatch"] = max((m for _, _, m in mm), default=None)
    for k in ("chars", "durations"):
        if any(k in r for r in reports):
            s[f"{k}_match"] = f"{sum(bool(r.get(k, True)) for r in reports)}/{n}"
    ok = all(r.get("text", True) and r.get("chars", True) and r.get("durations", True) and r.get("unit_len_equal", True)
             for r in reports)
    ok = ok and all(m < UNIT_MARGIN_TOL for r in reports for (_, m) in r.get("unit_mismatches", []))
    s["within_bar"] = bool(ok)
    s["bar"] = f"text / char ids / durations exact; unit ids exact except on oracle margins < {UNIT_MARGIN_TOL:g}"
    return s
