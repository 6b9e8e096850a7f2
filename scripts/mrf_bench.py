"""The narrow vocoder stages' multi-receptive-field block (C = 32 / 16): the fused kernel (sc_op_mrf_fused) against the nine
pair launches it replaces (sc_op_resblock_pair, the last one averaging), alone on the device, timed with the library's
per-launch HIP events.  Shapes: one length bucket (8 utterances) of the bench batch at each stage's rate."""
import ctypes as C
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seamless_communication_amd import _lib  # noqa: E402

lib = _lib.load_library()
KS, DILS = (3, 7, 11), (1, 3, 5)


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


def timed(fn, reps=3):
    fn()
    lib.sc_prof_reset()
    lib.sc_prof_enable(1)
    for _ in range(reps):
        fn()
    lib.sc_prof_enable(0)
    rep = report()
    return sum(v[1] for v in rep.values()) / reps, sum(v[2] for v in rep.values()) / reps


for nb, T, Cc in ((8, 84160, 32), (8, 168320, 16), (2, 84160, 32), (2, 168320, 16)):
    x = torch.randn(nb, T, Cc, device="cuda")
    w1, w2, b1, b2 = [], [], [], []
    for k in KS:
        kpad = (Cc * k + 31) // 32 * 32
        for _ in DILS:
            for lst in (w1, w2):
                w = torch.zeros(Cc, kpad, device="cuda", dtype=torch.float16)
                w[:, : Cc * k] = (torch.randn(Cc, Cc * k, device="cuda") / math.sqrt(Cc * k)).half()
                lst.append(w)
            b1.append(torch.randn(Cc, device="cuda") * 0.1)
            b2.append(torch.randn(Cc, device="cuda") * 0.1)
    bufs = [torch.empty_like(x) for _ in range(5)]
    out_pairs, out_fused = torch.empty_like(x), torch.empty_like(x)

    def pairs():
        outs = []
        for j, k in enumerate(KS):
            cur = x
            for d, dil in enumerate(DILS):
                q = 3 * j + d
                last = j == 2 and d == 2
                dst = out_pairs if last else (bufs[2 + j] if d == 2 else bufs[d & 1])
                lib.sc_op_resblock_pair(P(cur), P(w1[q]), P(b1[q]), P(w2[q]), P(b2[q]), P(dst), nb, T, Cc, k, dil, 0.1,
                                        P(outs[0]) if last else None, P(outs[1]) if last else None)
                cur = dst
            outs.append(cur)

    arr = lambda ts: (C.c_void_p * 9)(*[t.data_ptr() for t in ts])
    kk = (C.c_int32 * 3)(*KS)
    dd = (C.c_int32 * 9)(*[dil for _ in KS for dil in DILS])

    def fused():
        rc = lib.sc_op_mrf_fused(P(x), arr(w1), arr(b1), arr(w2), arr(b2), P(out_fused), nb, T, Cc, kk, dd, 0.1)
        assert rc == 0

    mp, fl = timed(pairs)
    mf, _ = timed(fused)
    same = bool(torch.equal(out_pairs, out_fused))
    print(f"nb={nb} T={T} C={Cc}: nine pairs {mp:.3f} ms ({fl / mp / 1e9:.1f} TFLOP/s)   fused {mf:.3f} ms ({fl / mf / 1e9:.1f} TFLOP/s)"
          f"   fused/pairs {mf / mp:.2f}   identical bits {same}")
