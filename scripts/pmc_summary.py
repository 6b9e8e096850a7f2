"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel name:
launches, mean FETCH_SIZE / WRITE_SIZE (KiB as reported) and the HBM traffic per
launch in bytes with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE
reports half the bytes of a wide coalesced read: doubled here)."""
import csv
import glob
import sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                a = acc[row["Kernel_Name"]][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


def csrc_sha() -> str:
    """Digest of the kernel sources the capture was made with: bench.py compares it with the tree it runs from and reports
    `traffic: null, traffic_from: "stale ..."` when a kernel source changed since the capture."""
    import hashlib
    from pathlib import Path

    root = Path(__file__).resolve().parents[1] / "seamless_communication_amd" / "csrc"
    h = hashlib.sha256()
    for f in sorted(list(root.glob("*.hip")) + list(root.glob("*.h")) + list(root.glob("*.cpp"))):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def main():
    print(f"# csrc_sha={csrc_sha()}")
    merged = defaultdict(dict)
    for d in sys.argv[1:]:
        for k, cs in load(d).items():
            for c, (n, s) in cs.items():
                merged[k][c] = (n, s / max(n, 1))
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean", "hbm_bytes_per_launch_corrected"])
    rows = []
    for k, cs in merged.items():
        n = max(v[0] for v in cs.values())
        fe = cs.get("FETCH_SIZE", (0, 0.0))[1]
        wr = cs.get("WRITE_SIZE", (0, 0.0))[1]
        rows.append((k, n, fe, wr, (2.0 * fe + wr) * 1024.0))
    rows.sort(key=lambda r: -r[1] * r[4])
    for r in rows:
        w.writerow([r[0][:120], r[1], f"{r[2]:.1f}", f"{r[3]:.1f}", f"{r[4]:.0f}"])


if __name__ == "__main__":
    main()
