"""CPU: the parts of bench.py that do not need a GPU — the command-line contract the driver relies on
(--gpus / --steps / --warmup, defaults that finish within minutes), the `cpu_baseline` leg (the oracle on one utterance
in a child process), the roofline bookkeeping and the PMC lookup."""
import importlib.util
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_argument_contract(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert (a.gpus, a.steps, a.warmup, a.batch, a.arch) == (1, 16, 2, 64, "base_v2")  # ~3 s of timed passes: two per pass worker of the pipeline
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    a = bench.parse_args()
    assert (a.gpus, a.steps, a.warmup) == (8, 5, 2)


def test_cpu_baseline_worker_reports_the_oracle_on_one_utterance():
    """1 warm-up + 3 timed passes (median), per-stage split, and the ids / arg-max margins of the checked utterances."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-worker", "--arch", "tiny_v2", "--text-len", "10",
                        "--cpu-threads", "2", "--workload", "fixed"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["kind"] == "port" and d["unit"] == "utterances/s" and d["cores"] == 2 and d["value"] > 0
    assert len(d["seconds_all"]) == 3 and sorted(d["seconds_all"])[1] == d["seconds"] and "median" in d["sample"]
    assert set(d["stage_ms"]) == {"fbank", "encoder", "text_decoder", "t2u_and_vocoder"}
    assert [c["index"] for c in d["checked"]] == [37, 0]
    for c in d["checked"]:
        assert len(c["text_ids"]) == 10 and c["text_ids"][0] == 3 and len(c["units"]) > 0
        assert c["min_text_margin"] >= 0 and c["min_unit_margin"] >= 0
        assert sum(c["text_margin_hist"].values()) == 8 and sum(c["unit_margin_hist"].values()) >= len(c["units"])


def test_default_workload_is_the_ragged_one(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert a.workload == "ragged" and a.text_len == 0 and a.engine_slots == 256  # text length resolved in main(): 64 tokens


def test_decoder_row_statistics():
    """Two slices of 4 rows: useful row-steps = generated tokens, computed = rows x steps up to the first poll after the last EOS."""
    lens = [10, 6, 3, 12, 8, 8, 8, 8]
    d = bench.decoder_row_stats(lens, 8, 2, {"text_decoder": 2.0})
    assert d["useful_row_steps"] == sum(l - 2 for l in lens) == 47
    assert d["slices"][0]["steps_to_last_eos"] == 10 and d["slices"][1]["steps_to_last_eos"] == 6
    assert d["computed_row_steps"] == 4 * 12 + 4 * 8
    assert d["slices"][0]["rows_alive_at_quartiles"][0] == 4 and d["slices"][0]["rows_alive_at_quartiles"][-1] == 1
    assert d["slice0_decoder_us_per_useful_row_step"] == round(2000.0 / (8 + 4 + 1 + 10), 2)


def test_decoder_row_statistics_with_live_row_compaction(monkeypatch):
    """More than 16 rows per slice: the greedy decoder packs the rows still generating to the front at every look at the
    finished flags (after generated tokens 4, 8, ...), the step kernels stop at their number - `computed_row_steps` follows
    that; SC_GREEDY_COMPACT=0 (and slices of up to 16 rows) report rows x steps as before."""
    lens = [6] * 10 + [10] * 6 + [14] * 4          # generated tokens 4 / 8 / 12 per row
    monkeypatch.delenv("SC_GREEDY_COMPACT", raising=False)
    d = bench.decoder_row_stats(lens, 20, 1, {})
    assert d["live_row_compaction"] is True and d["row_steps_without_compaction"] == 20 * 12
    # steps 0-3 on 20 rows, then the 10 rows that wrote their EOS at token 4 are behind the boundary: 4-7 on 10, 8-11 on 4
    assert d["computed_row_steps"] == 4 * 20 + 4 * 10 + 4 * 4
    assert d["useful_row_steps"] == 10 * 4 + 6 * 8 + 4 * 12
    monkeypatch.setenv("SC_GREEDY_COMPACT", "0")
    d0 = bench.decoder_row_stats(lens, 20, 1, {})
    assert d0["live_row_compaction"] is False and d0["computed_row_steps"] == 20 * 12
    monkeypatch.delenv("SC_GREEDY_COMPACT")
    assert bench.decoder_row_stats(lens[:16], 16, 1, {})["live_row_compaction"] is False


def test_margin_histogram_and_parity_block_without_oracle_output():
    assert bench._margin_hist([5e-6, 2e-4, 0.5, 3e-3, 1e-5]) == {"<1e-5": 1, "<1e-4": 1, "<1e-3": 1, "<1e-2": 1, ">=1e-2": 1}
    p = bench.parity_block({"error": "boom"}, None, None, None, None, None, None, None)
    assert p["n_checked"] == 0 and p["error"] == "boom"


def test_roofline_bookkeeping_and_pmc_lookup():
    fams = {"dec:skinny_m64": {"launches": 100, "ms": 2.0, "flops": 1e9, "bytes": 1.6e9},
            "enc:gemm_128x128_presplit": {"launches": 10, "ms": 1.0, "flops": 4e11, "bytes": 1e9}}
    roof, shares = bench.roofline_of(fams)
    assert roof["kernel"] == "dec:skinny_m64" and roof["bound"] == "hbm" and roof["unit"] == "GB/s"
    assert abs(roof["achieved"] - 800.0) < 1e-6 and abs(roof["frac"] - 0.1) < 1e-9 and roof["peak"] == 8000.0
    assert abs(roof["avg_launch_us"] - 20.0) < 1e-9 and roof["algorithmic_bytes_per_launch"] == 1.6e7
    # traffic is never measured inside bench.py: the value of the newest committed PMC summary, named - or null, with the
    # reason when that capture was made with other kernel sources than this tree's ("stale: ...")
    assert roof["traffic"] is None or isinstance(roof["traffic"], float)
    assert roof["traffic"] is None or roof["traffic_from"].startswith("profiles/")
    assert list(shares)[0] == "dec:skinny_m64" and shares["_profiled_total_ms"] == 3.0
    fams["enc:gemm_128x128_presplit"]["ms"] = 5.0
    roof, _ = bench.roofline_of(fams)
    assert roof["bound"] == "mfma" and abs(roof["achieved"] - 80.0) < 1e-9 and abs(roof["mfma_issue_tflops"] - 160.0) < 1e-9
    assert bench.roofline_of({}) == (None, {})
    val, src = bench.pmc_traffic("enc:no_such_family")
    assert val is None and (src is None or src.startswith("stale: "))
    # a capture stamped with this tree's kernel-source digest is used; one with another digest is reported as stale
    tmp = ROOT / "profiles" / "r99_unit_test_pmc_hbm_traffic.csv"
    rows = ["kernel,launches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean,hbm_bytes_per_launch_corrected",
            '"void sc::(anonymous namespace)::gemm_ps_kernel<128, 128, 2, 2, true, true, false>(sc::GemmPsArgs)",10,1.0,1.0,123456',
            '"void sc::(anonymous namespace)::gemv3_kernel<1, 1, 16, 4, 0, 1>(sc::Gemv3Args)",96,1.0,1.0,20000000',
            '"void sc::(anonymous namespace)::vocab3_kernel<8, true, false, false>(sc::Vocab3Args)",2,1.0,1.0,600000000',
            '"sc::add_i32_kernel(int*, int)",2,0.1,0.1,128']
    try:
        tmp.write_text("# csrc_sha=" + bench.csrc_sha() + "\n" + "\n".join(rows) + "\n")
        assert bench.pmc_traffic("enc:no_such_family") == (None, None)
        val, src = bench.pmc_traffic("enc:gemm_128x128_presplit")
        assert val == 123456.0 and src == "profiles/" + tmp.name
        val, src = bench.pmc_traffic("dec:step_graph")  # sum of the step's kernels / steps (= add_i32 launches)
        assert abs(val - (96 * 2e7 + 2 * 6e8 + 2 * 128) / 2) < 1 and "sum over the kernels of a step" in src
        tmp.write_text("# csrc_sha=0123456789abcdef\n" + "\n".join(rows) + "\n")
        val, src = bench.pmc_traffic("dec:step_graph")
        assert val is None and src.startswith("stale: profiles/" + tmp.name)
    finally:
        tmp.unlink(missing_ok=True)
