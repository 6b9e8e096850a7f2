#!/usr/bin/env python3
"""Static scan of the generated gfx950 ISA for the pattern that made the decoder-step kernels latency-bound
(profiles/r1_skinny_isa_notes.txt): loops in which global / buffer loads sit behind exec-masked branches and are drained
with `s_waitcnt vmcnt(0)` before their consumers.  Cross-compiles every kernel file (no GPU needed) and prints, per kernel
and loop, the instruction counts that matter.

    python scripts/isa_scan.py [file.hip ...] > profiles/rN_isa_scan.txt
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "seamless_communication_amd" / "csrc"


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def scan(src: Path):
    with tempfile.TemporaryDirectory() as d:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-c", str(src), "--save-temps",
                        "-o", str(Path(d) / "k.o")], cwd=d, check=True, capture_output=True)
        isa = next(Path(d).glob("*gfx950.s")).read_text().splitlines()
    starts = [i for i, l in enumerate(isa) if re.match(r"^_Z\w+:", l)]
    rows = []
    for s in starts:
        name = isa[s].split(":")[0]
        e = next(i for i in range(s, len(isa)) if "s_endpgm" in isa[i])
        fn = isa[s: e + 1]
        labels = {l.split(":")[0]: i for i, l in enumerate(fn) if re.match(r"^\.LBB\d+_\d+:", l)}
        extent = {}
        for i, l in enumerate(fn):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                a = labels[m.group(1)]
                extent[a] = max(extent.get(a, a), i)  # one loop per header: up to its last back-edge
        loops = sorted(extent.items())
        for a, b in loops:
            if any(a2 > a and b2 <= b for a2, b2 in loops):
                continue  # not innermost
            body = fn[a: b + 1]
            t = "\n".join(body)
            loads = len(re.findall(r"\b(global_load|buffer_load)_", t))
            if loads == 0:
                continue
            rows.append((name, b - a + 1, loads, t.count("vmcnt(0)"), len(re.findall(r"s_cbranch_exec", t)), len(re.findall(r"v_mfma", t)),
                         t.count("s_barrier"), t.count("v_accvgpr")))
    names = demangle(sorted({r[0] for r in rows}))
    return [(names[r[0]],) + r[1:] for r in rows]


def main():
    files = [Path(a) for a in sys.argv[1:]] or sorted(CSRC.glob("k_*.hip")) + sorted(CSRC.glob("model_*.hip"))
    print("# innermost loops that contain global / buffer loads, per kernel (profiles/r1_skinny_isa_notes.txt explains the columns)")
    print(f"{'kernel (loop)':90s} {'lines':>6s} {'loads':>6s} {'vmcnt0':>7s} {'execbr':>7s} {'mfma':>5s} {'barr':>5s} {'accmov':>7s}")
    for f in files:
        rows = scan(f)
        if rows:
            print(f"# {f.name}")
        for name, n, loads, v0, eb, mf, ba, am in sorted(rows, key=lambda r: -r[3]):
            flag = "  <-- loads drained inside the loop" if v0 and loads else ""
            print(f"{name[:90]:90s} {n:6d} {loads:6d} {v0:7d} {eb:7d} {mf:5d} {ba:5d} {am:7d}{flag}")


if __name__ == "__main__":
    main()
