"""Per-shape timing of the decoder-step skinny products (k_skinny.hip) through the C ABI + HIP-event profiler.
`--variant 1` times the experimental software-pipelined kernel (k_skinny2.hip), `--variant both` prints the two side by side."""
import ctypes as C
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seamless_communication_amd import _lib  # noqa: E402

lib = _lib.load_library()
VARIANTS = {"0": (0,), "1": (3,), "both": (0, 3)}  # 3 = KV_SKINNY | KV_REDUCE[sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else "0"]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def report():
    n = lib.sc_prof_report(None, 0)
    buf = C.create_string_buffer(int(n) + 16)
    lib.sc_prof_report(buf, len(buf))
    return {l.split()[0]: (int(l.split()[1]), float(l.split()[2]), float(l.split()[4])) for l in buf.value.decode().splitlines()}


for M in (1, 32, 64):
    for N, K in ((3072, 1024), (1024, 1024), (8192, 1024), (1024, 8192), (256102, 1024)):
        x = torch.randn(M, K, device="cuda")
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).half()
        y = torch.empty(M, N, device="cuda")
        torch.cuda.synchronize()
        for mode in ("linear", "res_ln"):
            if mode == "res_ln" and N != 1024:
                continue
            g = torch.ones(N, device="cuda")
            xin = torch.zeros(M, N, device="cuda")
            h = torch.empty(M, N, device="cuda")
            fn = (lambda: lib.sc_op_skinny_linear(P(x), P(w), None, None, P(y), M, N, K, 0, 1.0)) if mode == "linear" else \
                 (lambda: lib.sc_op_skinny_res_ln(P(x), P(w), None, P(xin), P(g), P(g), P(h), M, N, K, 0))
            for variant in VARIANTS:
                lib.sc_op_set_skinny_variant(variant)
                fn()
                lib.sc_prof_reset(); lib.sc_prof_enable(1)
                for _ in range(20):
                    fn()
                lib.sc_prof_enable(0)
                for name, (launches, ms, byts) in report().items():
                    print(f"M={M:3d} N={N:6d} K={K:5d} {mode:7s} v{variant} {name:16s} {1e3*ms/launches:8.2f} us  "
                          f"{byts/launches/(ms/launches)/1e6:8.1f} GB/s", flush=True)
lib.sc_op_set_skinny_variant(0)
