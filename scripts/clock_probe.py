#!/usr/bin/env python3
"""What clock the shader engines run at WHILE a kernel runs: a one-wave probe kernel on its own stream samples s_memtime (a
counter that follows the shader clock: 24.6 ticks per v_mfma_f32_32x32x16_f16 whatever the clock, scripts/micro/mfma_rate.hip)
and s_memrealtime (constant 100 MHz) every few microseconds while the main stream runs (a) nothing, (b) the DMA GEMM on the
encoder shapes, (c) the same on zero operands.  Prints ticks per microsecond per phase - relative clock, idle = 1.

    python scripts/clock_probe.py
"""
import ctypes as C
import math
import subprocess
import sys
import tempfile
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seamless_communication_amd import _lib  # noqa: E402

SRC = r"""
#include <hip/hip_runtime.h>
__global__ void probe(unsigned long long* out, int n, int spin) {
    for (int i = 0; i < n; ++i) {
        out[2 * i] = __builtin_amdgcn_s_memtime();
        out[2 * i + 1] = __builtin_amdgcn_s_memrealtime();
        for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(8);
    }
}
static hipStream_t g_s = nullptr;
extern "C" int probe_launch(unsigned long long* out, int n, int spin) {
    if (!g_s && hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking) != hipSuccess) return -1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, g_s, out, n, spin);
    return (int)hipGetLastError();
}
extern "C" int probe_wait() { return (int)hipStreamSynchronize(g_s); }
"""


def main():
    d = Path(tempfile.mkdtemp())
    (d / "probe.hip").write_text(SRC)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", str(d / "probe.hip"), "-o", str(d / "probe.so")], check=True)
    lib = _lib.load_library()
    probe = C.CDLL(str(d / "probe.so"))
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
    M, N, K = 15968, 4096, 1024
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).half()
    y = torch.empty(M, N, device="cuda")
    n = 40000
    for name, x in (("idle", None), ("gemm, random operands", torch.randn(M, K, device="cuda")), ("gemm, zero operands", torch.zeros(M, K, device="cuda"))):
        ww = torch.zeros_like(w) if name.endswith("zero operands") else w
        out = torch.zeros(n, 2, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        assert probe.probe_launch(P(out), n, 4) == 0
        if x is not None:
            for _ in range(12):
                lib.sc_op_linear_presplit(P(x), P(ww), None, None, P(y), None, None, M, N, K, 0, C.c_float(1.0))
        assert probe.probe_wait() == 0
        torch.cuda.synchronize()
        s = out.cpu().numpy()
        dt = (s[1:, 0] - s[:-1, 0]).astype(float)
        dr = (s[1:, 1] - s[:-1, 1]).astype(float)
        tot_us = (s[-1, 1] - s[0, 1]) / 100.0
        rate = dt.sum() / (dr.sum() / 100.0)
        # ticks per microsecond over windows of 20 samples (~20 us): the launches (~260 us each) stand out as runs of low windows
        import numpy as np

        win = 20
        k = (len(dt) // win) * win
        rates = dt[:k].reshape(-1, win).sum(1) / (dr[:k].reshape(-1, win).sum(1) / 100.0)
        q = np.percentile(rates, [0, 1, 5, 10, 25, 50])
        print(f"{name:24s}: {tot_us:8.0f} us sampled, s_memtime {rate:7.1f} ticks/us on average; 20-sample windows: min {q[0]:.0f}  p1 {q[1]:.0f}  p5 {q[2]:.0f}  p10 {q[3]:.0f}  "
              f"p25 {q[4]:.0f}  median {q[5]:.0f}", flush=True)


if __name__ == "__main__":
    main()
