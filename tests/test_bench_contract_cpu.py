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
    assert (a.gpus, a.steps, a.warmup, a.batch, a.arch) == (1, 3, 1, 64, "base_v2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    a = bench.parse_args()
    assert (a.gpus, a.steps, a.warmup) == (8, 5, 2)


def test_cpu_baseline_worker_reports_the_oracle_on_one_utterance():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-worker", "--arch", "tiny_v2", "--text-len", "10",
                        "--cpu-threads", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["kind"] == "port" and d["unit"] == "utterances/s" and d["cores"] == 2 and d["value"] > 0
    assert len(d["text_ids"]) == 10 and d["text_ids"][0] == 3 and len(d["units"]) > 0 and "1 utterance" in d["sample"]


def test_roofline_bookkeeping_and_pmc_lookup():
    fams = {"dec:skinny_m64": {"launches": 100, "ms": 2.0, "flops": 1e9, "bytes": 1.6e9},
            "enc:gemm_128x128_presplit": {"launches": 10, "ms": 1.0, "flops": 4e11, "bytes": 1e9}}
    roof, shares = bench.roofline_of(fams)
    assert roof["kernel"] == "dec:skinny_m64" and roof["bound"] == "hbm" and roof["unit"] == "GB/s"
    assert abs(roof["achieved"] - 800.0) < 1e-6 and abs(roof["frac"] - 0.1) < 1e-9 and roof["peak"] == 8000.0
    assert abs(roof["avg_launch_us"] - 20.0) < 1e-9 and roof["algorithmic_bytes_per_launch"] == 1.6e7
    assert roof["traffic"] is None or roof["traffic"] > 0  # from the newest committed PMC summary, when there is one
    assert list(shares)[0] == "dec:skinny_m64" and shares["_profiled_total_ms"] == 3.0
    fams["enc:gemm_128x128_presplit"]["ms"] = 5.0
    roof, _ = bench.roofline_of(fams)
    assert roof["bound"] == "mfma" and abs(roof["achieved"] - 80.0) < 1e-9 and abs(roof["mfma_issue_tflops"] - 160.0) < 1e-9
    assert bench.roofline_of({}) == (None, {})
    # the lookup reads the newest committed PMC summary (not the "early" ones) and matches the kernel by its template arguments
    import csv
    import glob

    newest = sorted(f for f in glob.glob(str(ROOT / "profiles" / "*pmc_hbm_traffic*.csv")) if "early" not in f)[-1]
    rows = [r for r in csv.DictReader(open(newest, newline="")) if "skinny_kernel<2, 1" in r["kernel"]]
    assert rows and bench.pmc_traffic("dec:skinny_m64") == float(rows[0]["hbm_bytes_per_launch_corrected"]) > 1e6
    assert bench.pmc_traffic("enc:no_such_family") is None
