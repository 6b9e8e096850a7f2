"""How busy is the device over the last bench pass?  From a rocprofv3 --kernel-trace CSV (Kernel_Name, Start_Timestamp,
End_Timestamp): the union of the kernel intervals against the wall-clock span, the concurrency profile (time with 0, 1, 2,
3+ kernels in flight), the idle gaps (count, total, the largest with the kernels either side) and a coarse timeline
(per bin: busy share, mean concurrency, the kernel family that dominates the bin).
    python scripts/trace_cover.py <kernel_trace.csv> [--window-ms W] [--bin-ms B] [--skip-tail-ms S]
W: analyse W ms of the trace (default 260: one 64-utterance pass) ending S ms before the last kernel (default 0; the pipelined
schedule drains at the end of a run: S = a few passes puts the window into its steady state); B: timeline bin (default 5)."""
import csv
import sys
from collections import defaultdict


def arg(name, default):
    return float(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


path = sys.argv[1]
window = arg("--window-ms", 260.0) * 1e6
binw = arg("--bin-ms", 5.0) * 1e6
rows = []
with open(path, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_end = max(e for _, e, _ in rows) - int(arg("--skip-tail-ms", 0.0) * 1e6)
t0 = t_end - window
rows = [(max(s, t0), min(e, t_end), n) for s, e, n in rows if e > t0 and s < t_end]


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").replace("sc::", "").split("(")[0][:44]


# sweep line over start / end events
ev = []
for i, (s, e, n) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, -1, i))
ev.sort(key=lambda x: (x[0], x[1]))
conc_time = defaultdict(float)
gaps = []
cur = 0
last_t = t0
last_end_kernel = None
for t, d, i in ev:
    if t > last_t:
        conc_time[min(cur, 4)] += t - last_t
        if cur == 0 and last_end_kernel is not None:
            gaps.append((t - last_t, last_t, last_end_kernel, None))
    if d == 1:
        if cur == 0 and gaps and gaps[-1][3] is None and gaps[-1][1] == last_t:
            g = gaps[-1]
            gaps[-1] = (g[0], g[1], g[2], short(rows[i][2]))
        cur += 1
    else:
        cur -= 1
        last_end_kernel = short(rows[i][2])
    last_t = t
span = t_end - t0
busy = span - conc_time[0]
print(f"window {span / 1e6:.1f} ms: {len(rows)} kernels, sum of durations {sum(e - s for s, e, _ in rows) / 1e6:.1f} ms, "
      f"device busy (union) {busy / 1e6:.1f} ms = {busy / span:.3f}, idle {conc_time[0] / 1e6:.1f} ms")
print("time with k kernels in flight: " + "  ".join(f"k={'4+' if k == 4 else k}: {conc_time[k] / 1e6:.1f} ms" for k in range(5)))
gaps.sort(reverse=True)
print(f"idle gaps: {len(gaps)}; > 10 us: {sum(1 for g in gaps if g[0] > 1e4)} totalling {sum(g[0] for g in gaps if g[0] > 1e4) / 1e6:.2f} ms; "
      f"> 100 us: {sum(1 for g in gaps if g[0] > 1e5)} totalling {sum(g[0] for g in gaps if g[0] > 1e5) / 1e6:.2f} ms")
for g in gaps[:12]:
    print(f"   {g[0] / 1e3:9.1f} us at +{(g[1] - t0) / 1e6:7.2f} ms   after {g[2]}   before {g[3]}")
# timeline
nb = int(span // binw) + 1
bb = [0.0] * nb
fam = [defaultdict(float) for _ in range(nb)]
for s, e, n in rows:
    b0, b1 = int((s - t0) // binw), int((e - t0) // binw)
    for b in range(b0, min(b1, nb - 1) + 1):
        lo, hi = max(s, t0 + b * binw), min(e, t0 + (b + 1) * binw)
        if hi > lo:
            bb[b] += hi - lo
            fam[b][short(n)] += hi - lo
print(f"timeline ({binw / 1e6:g} ms bins): mean kernels in flight | dominant kernels")
for b in range(nb):
    top = sorted(fam[b].items(), key=lambda kv: -kv[1])[:3]
    print(f"  +{b * binw / 1e6:6.1f} ms  {bb[b] / binw:5.2f} | " + ", ".join(f"{k} {v / binw:.2f}" for k, v in top))
