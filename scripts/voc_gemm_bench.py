"""Vocoder ResBlock convolutions of the wide stages (C = 256 / 128) alone on the DMA GEMM's implicit-convolution mode, through
the C ABI (sc_op_conv1d_presplit), timed with the library's per-launch HIP events: one length bucket of a 32-utterance slice,
one of a whole 64-utterance batch, and all rows of the batch in one launch (what a packed, bucket-free vocoder would run).
Tile thresholds are process-wide knobs (SC_PS_MIN256 / SC_PS_MIN128 / SC_PS_TILE): run once per setting."""
import ctypes as C
import math
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seamless_communication_amd import _lib  # noqa: E402

lib = _lib.load_library()


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def report():
    n = lib.sc_prof_report(None, 0)
    buf = C.create_string_buffer(int(n) + 16)
    lib.sc_prof_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, launches, ms, flops, byts = line.split()
        out[name] = (int(launches), float(ms), float(flops), float(byts))
    return out


SHAPES = []  # (label, nb, T, C, k, dil)
for label, nb in (("bucket of a 32-utt slice", 4), ("bucket of a 64-utt batch", 8), ("whole batch, one launch", 64)):
    SHAPES += [(label, nb, 2630, 256, 3, 1), (label, nb, 2630, 256, 7, 3), (label, nb, 2630, 256, 11, 5),
               (label, nb, 10520, 128, 3, 1), (label, nb, 10520, 128, 11, 5)]
knobs = " ".join(f"{k}={os.environ[k]}" for k in ("SC_PS_TILE", "SC_PS_MIN256", "SC_PS_MIN128") if k in os.environ) or "defaults"
print(f"# {knobs}")
for label, nb, T, Cc, k, dil in SHAPES:
    x = torch.randn(nb, T, Cc, device="cuda")
    wp = (torch.randn(Cc, Cc * k, device="cuda") / math.sqrt(Cc * k)).half()
    b = torch.randn(Cc, device="cuda")
    res = torch.randn(nb, T, Cc, device="cuda")
    y = torch.empty(nb, T, Cc, device="cuda")
    yh = torch.empty(nb, T, Cc, device="cuda", dtype=torch.float16)
    yl = torch.empty(nb, T, Cc, device="cuda", dtype=torch.float16)
    torch.cuda.synchronize()
    pad = dil * (k - 1) // 2

    def run():
        lib.sc_op_conv1d_presplit(P(x), P(wp), P(b), P(res), P(y), P(yh), P(yl), nb, T, Cc, Cc, k, pad, dil, None, 0)

    run()
    lib.sc_prof_reset()
    lib.sc_prof_enable(1)
    for _ in range(4):
        run()
    lib.sc_prof_enable(0)
    rep = {kk: v for kk, v in report().items() if "presplit" in kk}
    name = max(rep, key=lambda kk: rep[kk][1])
    launches, ms, flops, byts = rep[name]
    us = 1e3 * ms / launches
    print(f"{label:26s} rows {nb * T:7d} C {Cc:3d} k {k:2d} d {dil}  {name:28s} {us:9.1f} us  {flops / launches / us / 1e6:7.1f} TFLOP/s  "
          f"{byts / launches / us / 1e3:7.1f} GB/s", flush=True)
