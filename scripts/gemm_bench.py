"""A/B timing of the two dense-product kernels (general k_gemm.hip vs fast k_gemm2.hip) on the
shapes the S2ST path launches at batch 16, through the C ABI (sc_op_linear / sc_op_conv1d) with
the library's own per-launch HIP-event profiler.  Prints one line per shape and path."""
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


def timed(fn, reps=6):
    res = {}
    for general in ((0,) if QUICK else (1, 0)):
        lib.sc_op_force_general_gemm(general)
        fn()  # warm-up
        lib.sc_prof_reset()
        lib.sc_prof_enable(1)
        for _ in range(reps):
            fn()
        lib.sc_prof_enable(0)
        rep = {k: v for k, v in report().items() if "gemm" in k}
        name = max(rep, key=lambda k: rep[k][1])
        launches, ms, flops, byts = rep[name]
        res["general" if general else "fast"] = (name, ms / launches, flops / launches, byts / launches)
    lib.sc_op_force_general_gemm(0)
    return res


def show(label, res):
    for path, (name, ms, flops, byts) in res.items():
        print(f"gm={os.environ.get('SC_GEMM_GROUP_M', '-'):3s} {label:44s} {path:8s} {name:34s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TFLOP/s  {byts/ms/1e6:8.1f} GB/s", flush=True)


LINEAR = [  # (M, N, K) at batch 16: S=499 rows per utterance
    (7984, 4096, 1024), (7984, 1024, 4096), (7984, 3072, 1024), (7984, 2048, 1024), (7984, 1024, 1024),
    (1008, 8192, 1024), (8320, 8192, 1024), (8320, 1024, 8192), (8320, 10082, 1024), (640, 8192, 1024),
]
CONV = [  # (nb, T, cin, cout, k, stride, pad, dil, in_act)
    (16, 520, 1024, 1024, 7, 1, 3, 1, 0),      # NAR decoder conv k7
    (16, 520, 1792, 512, 7, 1, 3, 1, 0),       # vocoder conv_pre
    (16, 2600, 256, 256, 3, 1, 1, 1, 1),       # resblock stage 1
    (16, 2600, 256, 256, 11, 1, 25, 5, 1),
    (16, 10400, 128, 128, 7, 1, 9, 3, 1),      # stage 2
    (16, 41600, 64, 64, 7, 1, 9, 3, 1),        # stage 3
    (16, 83200, 32, 32, 11, 1, 5, 1, 1),       # stage 4
    (16, 83200, 32, 32, 3, 1, 1, 1, 1),
]

QUICK = "--quick" in sys.argv
PS_ONLY = "--presplit-only" in sys.argv
if QUICK:
    # the encoder's products for a 32-utterance slice (15968 rows) and for 64 utterances
    LINEAR = [(15968, 4096, 1024), (15968, 1024, 4096), (15968, 3072, 1024), (15968, 2048, 1024), (15968, 1024, 1024),
              (31936, 4096, 1024), (31936, 1024, 4096), (31936, 1024, 1024)]
    CONV = [(16, 520, 1024, 1024, 7, 1, 3, 1, 0), (16, 2600, 256, 256, 11, 1, 25, 5, 1)]
if "--decoder-shapes" in sys.argv:
    # the decoder step's products at the row counts of the decode engine / of beam search, as the DMA GEMM sees them (what a
    # tiled product could reach there; K = 8192 runs as ONE sequential K loop here, no split-K)
    PS_ONLY = True
    LINEAR = [(M, N, K) for M in (128, 192, 256, 320) for N, K in ((8192, 1024), (1024, 8192), (1024, 1024), (3072, 1024), (2048, 1024))]
    CONV = []


if any(a.startswith("--shapes=") for a in sys.argv):
    # --shapes=M,N,K;M,N,K: any list of products on the DMA GEMM (e.g. L2-resident operands: 2048,1024,1024 with SC_PS_MIN256=1)
    PS_ONLY = True
    LINEAR = [tuple(int(v) for v in t.split(",")) for a in sys.argv if a.startswith("--shapes=") for t in a[len("--shapes="):].split(";")]
    CONV = []


def timed_presplit(fn, reps=6):
    fn()
    lib.sc_prof_reset()
    lib.sc_prof_enable(1)
    for _ in range(reps):
        fn()
    lib.sc_prof_enable(0)
    rep = {k: v for k, v in report().items() if "presplit" in k}
    name = max(rep, key=lambda k: rep[k][1])
    launches, ms, flops, byts = rep[name]
    return {"presplit": (name, ms / launches, flops / launches, byts / launches)}


ZEROS = "--zeros" in sys.argv  # all-zero operands: the same instruction stream without toggling data (power management)
for M, N, K in LINEAR:
    x = torch.zeros(M, K, device="cuda") if ZEROS else torch.randn(M, K, device="cuda")
    w = torch.zeros(N, K, device="cuda", dtype=torch.float16) if ZEROS else (torch.randn(N, K, device="cuda") / math.sqrt(K)).half()
    b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    torch.cuda.synchronize()
    if not PS_ONLY:
        show(f"linear M={M} N={N} K={K}", timed(lambda: lib.sc_op_linear(P(x), P(w), P(b), None, P(y), M, N, K, 0, 1.0, 1, 0)))
    show(f"linear M={M} N={N} K={K}", timed_presplit(lambda: lib.sc_op_linear_presplit(P(x), P(w), P(b), None, P(y), None, None, M, N, K, 0, 1.0)))
for nb, T, cin, cout, k, stride, pad, dil, in_act in ([] if PS_ONLY else CONV):
    x = torch.randn(nb, T, cin, device="cuda")
    wp = (torch.randn(cout, cin * k, device="cuda") / math.sqrt(cin * k)).half()
    b = torch.randn(cout, device="cuda")
    t_out = (T + 2 * pad - dil * (k - 1) - 1) // stride + 1
    y = torch.empty(nb, t_out, cout, device="cuda")
    show(f"conv nb={nb} T={T} {cin}->{cout} k{k} d{dil}",
         timed(lambda: lib.sc_op_conv1d(P(x), P(wp), P(b), None, P(y), nb, T, cin, cout, k, stride, pad, dil, None, in_act, 0)))
