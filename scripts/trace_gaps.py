"""Per-kernel durations AND the idle gaps between consecutive kernels of one stream from a rocprofv3 --kernel-trace
CSV (columns Kernel_Name, Start_Timestamp, End_Timestamp [, Stream_Id / Queue_Id]).
    python scripts/trace_gaps.py <kernel_trace.csv> [--last N]   (N = only the last N records: the steady state)"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
rows = []
with open(path, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
if last:
    rows = rows[-last:]
dur = defaultdict(lambda: [0, 0.0])
gap_after = defaultdict(lambda: [0, 0.0])
total_busy = total_gap = 0.0
for i, (s, e, name, q) in enumerate(rows):
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("sc::", "").split("(")[0][:60]
    d = dur[short]
    d[0] += 1
    d[1] += e - s
    total_busy += e - s
    if i + 1 < len(rows):
        g = rows[i + 1][0] - e
        if 0 <= g < 50_000:  # ignore host-side pauses between calls
            ga = gap_after[short]
            ga[0] += 1
            ga[1] += g
            total_gap += g
print(f"{len(rows)} kernels, busy {total_busy / 1e6:.2f} ms, gaps (<50 us) {total_gap / 1e6:.2f} ms, span {(rows[-1][1] - rows[0][0]) / 1e6:.2f} ms")
print(f"{'kernel':62s} {'calls':>7s} {'avg us':>8s} {'total ms':>9s} {'gap after us':>13s}")
for k, (n, t) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:30]:
    ga = gap_after.get(k, [0, 0.0])
    print(f"{k:62s} {n:7d} {t / n / 1e3:8.2f} {t / 1e6:9.2f} {(ga[1] / ga[0] / 1e3 if ga[0] else 0):13.2f}")
