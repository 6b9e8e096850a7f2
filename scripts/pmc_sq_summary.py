"""Summarises rocprofv3 --pmc passes of SQ / TCP / GRBM counters per (kernel, grid size): mean per launch, summed over
the XCDs.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (32 per
v_mfma_f32_32x32x16_f16), MI355X_MICROARCH.md "rocprofv3 PMC slots".  Derived lines: matrix-pipe busy share
(SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs) and the wait split.
    python scripts/pmc_sq_summary.py DIR [DIR ...] [--filter gemm_ps]"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
    dirs = [a for a in sys.argv[1:] if not a.startswith("--") and a != flt]
    by_kernel = "--by-kernel" in sys.argv  # one line per kernel name (all grids together), sorted by total wave cycles
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    if flt and flt not in row["Kernel_Name"]:
                        continue
                    key = (row["Kernel_Name"][:110], "*" if by_kernel else row.get("Grid_Size", "?"))
                    a = acc[key][row["Counter_Name"]]
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
    # by kernel: largest total GPU time first (sum of GRBM_GUI_ACTIVE over the launches)
    order = (lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", [0, 0.0])[1]) if by_kernel else (lambda kv: -max(v[0] for v in kv[1].values()))
    for (k, grid), cs in sorted(acc.items(), key=order):
        m = {c: s / max(n, 1) for c, (n, s) in cs.items()}
        n = max(v[0] for v in cs.values())
        print(f"{k} grid={grid} launches={n}")
        print("   " + "  ".join(f"{c}={v:.3e}" for c, v in sorted(m.items())))
        d = []
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE", 0) > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: kernel cycles = GUI_ACTIVE / 8; 256 CUs x 4 SIMDs issue slots per cycle
            d.append(f"mfma_pipe_busy={m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * m['GRBM_GUI_ACTIVE'] / 8.0):.3f}")
        if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in m:
                    d.append(f"{c}/WAVE_CYCLES={m[c] / m['SQ_WAVE_CYCLES']:.3f}")
        if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            d.append(f"lds_conflict_share={m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.3f}")
        if d:
            print("   derived: " + "  ".join(d))


if __name__ == "__main__":
    main()
