"""LayerNorm launches of the encoder at the benchmark shape (15 968 rows x 1024), through the C ABI op hooks, for a
`rocprofv3 --kernel-trace --stats` run (scripts/gpu.sh TAG pyprof with PYPROF=scripts/ln_bench.py): kernel time per launch of
layernorm_kernel<4> (fp32 out) and layernorm2_kernel (fp32 + split planes of a second LayerNorm)."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seamless_communication_amd import _lib  # noqa: E402

lib = _lib.load_library()


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


rows, Cc = 15968, 1024
x = torch.randn(rows, Cc, device="cuda")
g = torch.rand(Cc, device="cuda") + 0.5
b = torch.randn(Cc, device="cuda") * 0.1
y = torch.empty_like(x)
yh = torch.empty(rows, Cc, dtype=torch.float16, device="cuda")
yl = torch.empty(rows, Cc, dtype=torch.float16, device="cuda")
for _ in range(30):
    assert lib.sc_op_layernorm(P(x), P(g), P(b), P(y), rows, Cc, 0) == 0
    assert lib.sc_op_layernorm2(P(x), P(g), P(b), P(g), P(b), P(y), P(yh), P(yl), rows, Cc, 1) == 0  # layernorm2_kernel
    assert lib.sc_op_layernorm2(P(x), P(g), P(b), P(g), P(b), P(y), P(yh), P(yl), rows, Cc, 0) == 0  # layernorm + layernorm_split
torch.cuda.synchronize()
print("done")
