"""The vocoder's shipped ResBlock kernels: products on the hi fp16 plane of the (LeakyReLU'd) activations only - one matrix
instruction per fragment, no lo plane produced, staged or read (k_resblock.hip: resblock_pair_kernel<C, true>,
mrf_fused_kernel<C, true>; k_gemm_ps.hip: SPLIT = false with hi-only output planes).  The north star puts the vocoder under
a waveform tolerance, not bit-exactness (reference: models/vocoder/hifigan.py:114-121, 180-196; its own GPU path runs these
layers in fp16 altogether).

What has to hold at op level:
  * against float64 with the SAME rounding points (activations rounded to fp16 where a convolution reads them, everything
    else exact) the result agrees to accumulation noise - i.e. the kernels do what they say and nothing else got lost with
    the lo plane (halo rows, tile edges, the averaging epilogue);
  * against the un-rounded float64 block the error is that of fp16 activations (about 2^-11 of the activations' scale);
  * the fused multi-receptive-field kernel equals nine single-plane pair launches bit for bit, like its two-plane twin.
Model-level error (waveform against the CPU oracle, bar 2e-3) is reported by tests/test_fullsize_gpu.py and bench.py."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from tests.test_ops_gpu import P, _log, check, dev, lib, rel_err  # noqa: F401  (lib is a fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture()
def single(lib):
    check(lib, lib.sc_op_single_plane(1))
    yield lib
    check(lib, lib.sc_op_single_plane(0))


def _h(x):
    """what a convolution reads: LeakyReLU, rounded to fp16 once"""
    return F.leaky_relu(x, 0.1).to(torch.float32).half().double()


def _pair_refs(x, w1, b1, w2, b2, k, dil):
    xt = x.double().transpose(1, 2)
    # same rounding points as the kernel: the fp32 value is rounded to fp16 when the next convolution reads it
    c1 = F.conv1d(_h(xt), w1.double(), b1.double(), padding=dil * (k - 1) // 2, dilation=dil)
    emu = F.conv1d(_h(c1), w2.double(), b2.double(), padding=(k - 1) // 2).transpose(1, 2) + x.double()
    c1x = F.conv1d(F.leaky_relu(xt, 0.1), w1.double(), b1.double(), padding=dil * (k - 1) // 2, dilation=dil)
    exact = F.conv1d(F.leaky_relu(c1x, 0.1), w2.double(), b2.double(), padding=(k - 1) // 2).transpose(1, 2) + x.double()
    return emu, exact


def _weights(g, C_, k):
    w = (torch.randn(C_, C_, k, generator=g) / math.sqrt(C_ * k)).half()
    b = torch.randn(C_, generator=g) * 0.1
    return w, b


def _pack(lib, w, C_, k, kpad):
    wp = torch.zeros(C_, kpad, dtype=torch.float16, device="cuda")
    check(lib, lib.sc_op_pack_conv_weight(P(dev(w)), P(wp), C_, C_, k))
    return wp


@pytest.mark.parametrize("nb,T,C_,k,dil,avg", [(2, 1000, 32, 11, 5, True), (1, 257, 32, 7, 3, False), (3, 777, 16, 11, 5, False), (1, 100, 16, 7, 1, False),
                                               (2, 900, 64, 11, 5, True), (1, 1300, 64, 7, 3, False), (2, 118, 64, 3, 1, False)])
def test_narrow_pair_on_one_plane(single, report_dir, nb, T, C_, k, dil, avg):
    lib = single
    g = torch.Generator().manual_seed(T * 13 + C_ * 5 + k + dil)
    x = torch.randn(nb, T, C_, generator=g)
    (w1, b1), (w2, b2) = _weights(g, C_, k), _weights(g, C_, k)
    ra, rb = torch.randn(nb, T, C_, generator=g), torch.randn(nb, T, C_, generator=g)
    kpad = (C_ * k + 31) // 32 * 32
    wp1, wp2 = _pack(lib, w1, C_, k, kpad), _pack(lib, w2, C_, k, kpad)
    got = torch.full((nb, T, C_), float("nan"), device="cuda")
    check(lib, lib.sc_op_resblock_pair(P(dev(x)), P(wp1), P(dev(b1)), P(wp2), P(dev(b2)), P(got), nb, T, C_, k, dil, 0.1,
                                       P(dev(ra)) if avg else None, P(dev(rb)) if avg else None))
    got = got.cpu()
    assert not torch.isnan(got).any()
    emu, exact = _pair_refs(x, w1, b1, w2, b2, k, dil)
    if avg:
        emu, exact = (ra.double() + rb.double() + emu) / 3.0, (ra.double() + rb.double() + exact) / 3.0
    e_emu, e_exact = rel_err(got, emu), rel_err(got, exact)
    _log(report_dir, "single_plane_pair", nb=nb, T=T, C=C_, k=k, dil=dil, avg=avg, err_vs_same_rounding=e_emu, err_vs_float64=e_exact)
    # a rounding point can fall the other way where fp32 and float64 sums straddle a tie: a handful of 1-ulp flips
    assert e_emu < 5e-5, e_emu
    assert e_exact < 1.5e-3, e_exact


@pytest.mark.parametrize("nb,T,C_,k,dil", [(2, 2500, 256, 3, 1), (2, 2500, 256, 11, 5), (1, 10000, 128, 7, 3), (3, 333, 128, 11, 1), (1, 40, 256, 7, 5)])
def test_wide_pair_on_one_plane(single, report_dir, nb, T, C_, k, dil):
    """the wide stages: both convolutions on the DMA-fed GEMM without a lo plane (neither fetched nor produced)"""
    lib = single
    g = torch.Generator().manual_seed(T * 13 + C_ * 5 + k + dil)
    x = torch.randn(nb, T, C_, generator=g)
    (w1, b1), (w2, b2) = _weights(g, C_, k), _weights(g, C_, k)
    wp1, wp2 = _pack(lib, w1, C_, k, C_ * k), _pack(lib, w2, C_, k, C_ * k)
    got = torch.full((nb, T, C_), float("nan"), device="cuda")
    check(lib, lib.sc_op_resblock_pair_ps(P(dev(x)), P(wp1), P(dev(b1)), P(wp2), P(dev(b2)), P(got), nb, T, C_, k, dil))
    got = got.cpu()
    assert not torch.isnan(got).any()
    emu, exact = _pair_refs(x, w1, b1, w2, b2, k, dil)
    e_emu, e_exact = rel_err(got, emu), rel_err(got, exact)
    _log(report_dir, "single_plane_pair_ps", nb=nb, T=T, C=C_, k=k, dil=dil, err_vs_same_rounding=e_emu, err_vs_float64=e_exact)
    assert e_emu < 5e-5, e_emu
    assert e_exact < 1.5e-3, e_exact


@pytest.mark.parametrize("nb,T,C_,ks,dils", [(2, 1000, 32, (3, 7, 11), (1, 3, 5)), (3, 393, 16, (3, 7, 11), (1, 3, 5)), (1, 37, 16, (3, 7, 11), (1, 3, 5)),
                                             (2, 901, 32, (11, 3, 7), (5, 1, 3))])
def test_fused_mrf_on_one_plane_equals_nine_one_plane_pairs(single, report_dir, nb, T, C_, ks, dils):
    lib = single
    g = torch.Generator().manual_seed(T * 7 + C_ * 3 + ks[0])
    x = torch.randn(nb, T, C_, generator=g)
    W1, W2, B1, B2, P1, P2, D1, D2 = [], [], [], [], [], [], [], []
    for k in ks:
        kpad = (C_ * k + 31) // 32 * 32
        for _ in dils:
            (w1, b1), (w2, b2) = _weights(g, C_, k), _weights(g, C_, k)
            W1.append(w1), W2.append(w2), B1.append(b1), B2.append(b2)
            P1.append(_pack(lib, w1, C_, k, kpad)), P2.append(_pack(lib, w2, C_, k, kpad))
            D1.append(dev(b1)), D2.append(dev(b2))
    dx = dev(x)
    outs = []
    for j, k in enumerate(ks):
        cur = dx
        for d, dil in enumerate(dils):
            q = 3 * j + d
            last = j == 2 and d == 2
            nxt = torch.full((nb, T, C_), float("nan"), device="cuda")
            check(lib, lib.sc_op_resblock_pair(P(cur), P(P1[q]), P(D1[q]), P(P2[q]), P(D2[q]), P(nxt), nb, T, C_, k, dil, 0.1,
                                               P(outs[0]) if last else None, P(outs[1]) if last else None))
            cur = nxt
        outs.append(cur)
    want = outs[2].cpu()
    arr = lambda ts: (C.c_void_p * 9)(*[t.data_ptr() for t in ts])  # noqa: E731
    got = torch.full((nb, T, C_), float("nan"), device="cuda")
    check(lib, lib.sc_op_mrf_fused(P(dx), arr(P1), arr(D1), arr(P2), arr(D2), P(got), nb, T, C_, (C.c_int32 * 3)(*ks),
                                   (C.c_int32 * 9)(*[dil for _ in ks for dil in dils]), 0.1))
    got = got.cpu()
    assert not torch.isnan(got).any()
    ref = torch.zeros(nb, T, C_, dtype=torch.float64)
    for j, k in enumerate(ks):
        cur = x.double().transpose(1, 2)
        for d, dil in enumerate(dils):
            q = 3 * j + d
            h = F.conv1d(F.leaky_relu(cur, 0.1), W1[q].double(), B1[q].double(), padding=dil * (k - 1) // 2, dilation=dil)
            cur = F.conv1d(F.leaky_relu(h, 0.1), W2[q].double(), B2[q].double(), padding=(k - 1) // 2) + cur
        ref += cur.transpose(1, 2)
    ref /= 3.0
    err = rel_err(got, ref)
    _log(report_dir, "single_plane_mrf", nb=nb, T=T, C=C_, ks=list(ks), err_vs_float64=err, bit_identical=bool(torch.equal(got, want)))
    assert torch.equal(got, want)
    assert err < 3e-3, err
