"""Kernel-level parity: every HIP op of libseamless_hip against a plain PyTorch
fp32 CPU restatement of the same op (called through the C ABI, sc_op_*).

Tolerances are stated per test.  The dense products use an fp32-activation x
fp16-weight split-MFMA scheme whose error is ~2^-22 relative per product, so
GEMM results are expected to agree with fp32 to ~1e-5 relative.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from seamless_communication_amd import _lib

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return _lib.load_library()


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_KEEP = []


def dev(t):
    """Device copy that stays alive until the end of the test: the raw pointer
    handed to the C ABI must not be recycled by the caching allocator."""
    d = t.contiguous().cuda()
    _KEEP.append(d)
    return d


@pytest.fixture(autouse=True)
def _release_device_copies():
    yield
    _KEEP.clear()


def check(lib, st):
    assert st == 0, lib.sc_last_error().decode()


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _log(report_dir, name, **kw):
    with open(report_dir / "ops_report.txt", "a") as f:
        f.write(name + " " + " ".join(f"{k}={v}" for k, v in kw.items()) + "\n")


@pytest.mark.parametrize("rows,C_", [(7, 160), (499, 1024), (33, 256), (5, 2048), (3, 128)])
@pytest.mark.parametrize("act", [0, 2])
def test_layernorm(lib, report_dir, rows, C_, act):
    g = torch.Generator().manual_seed(rows * 1000 + C_)
    x = torch.randn(rows, C_, generator=g) * 3 + 0.5
    w = torch.rand(C_, generator=g) + 0.5
    b = torch.randn(C_, generator=g) * 0.1
    ref = F.layer_norm(x, (C_,), w, b, 1e-5)
    if act == 2:
        ref = F.silu(ref)
    y = torch.empty(rows, C_, device="cuda")
    check(lib, lib.sc_op_layernorm(P(dev(x)), P(dev(w)), P(dev(b)), P(y), rows, C_, act))
    err = float((y.cpu() - ref).abs().max())
    _log(report_dir, "layernorm", rows=rows, C=C_, act=act, err=err)
    assert err < 2e-5


GEMM_SHAPES = [
    (499, 4096, 1024),  # encoder FFN inner, 64x64 tiles
    (998, 1024, 4096),
    (5, 1024, 1024),  # skinny tile (M <= 32)
    (40, 3072, 1024),
    (63, 2048, 1024),
    (2048, 2048, 1024),  # 128x128 tiles
    (130, 100, 160),  # ragged N, K = 160
    (257, 1, 352),  # N = 1
    (31, 10082, 1024),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_linear_mfma_split(lib, report_dir, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, generator=g) * 2.0
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half()
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g)
    ref = 0.5 * F.silu(x.double() @ w.double().t() + b.double()) + r.double()
    y = torch.empty(M, N, device="cuda")
    check(lib, lib.sc_op_linear(P(dev(x)), P(dev(w)), P(dev(b)), P(dev(r)), P(y), M, N, K, 2, 0.5, 1, 0))
    err = rel_err(y.cpu(), ref)
    # transpose-detecting: asymmetric random operands; fp32-class accuracy expected
    ref32 = 0.5 * F.silu(x @ w.float().t() + b) + r
    err32 = rel_err(ref32, ref)
    _log(report_dir, "linear_split", M=M, N=N, K=K, err=err, fp32_cpu_err=err32)
    assert err < 2e-6, (err, err32)


def test_linear_mfma_nosplit_is_fp16_class(lib, report_dir):
    M, N, K = 300, 512, 1024
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half()
    ref = x.double() @ w.double().t()
    y = torch.empty(M, N, device="cuda")
    check(lib, lib.sc_op_linear(P(dev(x)), P(dev(w)), None, None, P(y), M, N, K, 0, 1.0, 0, 0))
    err = rel_err(y.cpu(), ref)
    _log(report_dir, "linear_nosplit", err=err)
    assert 1e-6 < err < 2e-3  # fp16 rounding of the activations is visible, as designed


@pytest.mark.parametrize("M", [1, 2, 3, 8])
@pytest.mark.parametrize("N,K", [(1024, 1024), (8192, 1024), (1024, 8192), (1003, 1024)])
def test_gemv(lib, report_dir, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half()
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g)
    ref = F.relu(x.double() @ w.double().t() + b.double()) + r.double()
    y = torch.empty(M, N, device="cuda")
    check(lib, lib.sc_op_linear(P(dev(x)), P(dev(w)), P(dev(b)), P(dev(r)), P(y), M, N, K, 1, 1.0, 1, 1))
    err = rel_err(y.cpu(), ref)
    _log(report_dir, "gemv", M=M, N=N, K=K, err=err)
    assert err < 2e-6


CONV_CASES = [
    # nb, T, cin, cout, k, stride, pad, dil, in_act, act, lens
    (2, 50, 1024, 256, 3, 1, 1, 1, 0, 1, [50, 37]),  # duration predictor conv1 (+ReLU, masked)
    (2, 61, 128, 128, 7, 1, 3, 1, 0, 0, [61, 20]),  # NAR conv k7
    (2, 499, 128, 256, 8, 8, 4, 1, 0, 0, None),  # adaptor strided conv
    (1, 300, 64, 64, 11, 1, 25, 5, 1, 0, None),  # resblock dilated conv, LeakyReLU(0.1) on input
    (2, 200, 16, 16, 3, 1, 3, 3, 1, 0, None),  # narrow channels (generic A path)
    (1, 333, 16, 1, 7, 1, 3, 1, 2, 3, None),  # conv_post: lrelu(0.01) in, tanh out
    (1, 40, 1792, 512, 7, 1, 3, 1, 0, 0, None),  # conv_pre
    (3, 17, 8, 4, 3, 1, 1, 1, 0, 0, None),  # tiny
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv1d(lib, report_dir, case):
    nb, T, cin, cout, k, stride, pad, dil, in_act, act, lens = case
    g = torch.Generator().manual_seed(T * 31 + cin)
    x = torch.randn(nb, T, cin, generator=g)
    w = (torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)).half()
    b = torch.randn(cout, generator=g) * 0.1
    xin = x.clone()
    if lens is not None:
        for i, l in enumerate(lens):
            xin[i, l:] = 0
    if in_act == 1:
        xin = F.leaky_relu(xin, 0.1)
    elif in_act == 2:
        xin = F.leaky_relu(xin, 0.01)
    ref = F.conv1d(xin.transpose(1, 2).double(), w.double(), b.double(), stride=stride, padding=pad, dilation=dil).transpose(1, 2)
    t_out = ref.shape[1]
    res = torch.randn(nb, t_out, cout, generator=g)
    if act == 1:
        ref = F.relu(ref)
    elif act == 3:
        ref = torch.tanh(ref)
    ref = ref + res.double()
    kpad = (cin * k + 31) // 32 * 32
    wp = torch.zeros(cout, kpad, dtype=torch.float16, device="cuda")
    check(lib, lib.sc_op_pack_conv_weight(P(dev(w)), P(wp), cout, cin, k))
    d_lens = dev(torch.tensor(lens, dtype=torch.int32)) if lens is not None else None
    y = torch.full((nb, t_out, cout), float("nan"), device="cuda")
    check(lib, lib.sc_op_conv1d(P(dev(x)), P(wp), P(dev(b)), P(dev(res)), P(y), nb, T, cin, cout, k, stride, pad, dil,
                                P(d_lens), in_act, act))
    err = rel_err(y.cpu(), ref)
    _log(report_dir, "conv1d", case=case, err=err)
    assert err < 3e-6


FAST_LINEAR_SHAPES = [
    (7984, 4096, 1024),  # encoder FFN inner at batch 16: 128x128 tiles, XCD-ordered 1-D grid
    (2048, 1024, 4096),
    (499, 1024, 1024),  # 64x64 tiles
    (70000, 32, 96),  # narrow output: 256x32 tiles
    (3000, 32, 160),  # 128x32 tiles
    (70000, 64, 192),  # 128x64 tiles
    (333, 48, 64),  # ragged N below one tile
    (129, 130, 32),  # a single K slab
]


@pytest.mark.parametrize("M,N,K", FAST_LINEAR_SHAPES)
@pytest.mark.parametrize("act,with_res", [(0, False), (2, True)])
def test_fast_gemm_bit_identical_to_general_linear(lib, report_dir, M, N, K, act, with_res):
    """k_gemm2.hip (double-buffered, XCD-ordered) must reproduce the general kernel bit for bit:
    same hi/lo split, same K order, same epilogue arithmetic."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    x = dev(torch.randn(M, K, generator=g) * 2.0)
    w = dev((torch.randn(N, K, generator=g) / math.sqrt(K)).half())
    b = dev(torch.randn(N, generator=g) * 0.1)
    r = dev(torch.randn(M, N, generator=g)) if with_res else None
    out = []
    for general in (1, 0):
        check(lib, lib.sc_op_force_general_gemm(general))
        y = torch.full((M, N), float("nan"), device="cuda")
        check(lib, lib.sc_op_linear(P(x), P(w), P(b), P(r), P(y), M, N, K, act, 0.5, 1, 0))
        out.append(y.cpu())
    check(lib, lib.sc_op_force_general_gemm(0))
    ref = x.cpu().double() @ w.cpu().double().t() + b.cpu().double()
    ref = 0.5 * (F.silu(ref) if act == 2 else ref) + (r.cpu().double() if with_res else 0)
    err = rel_err(out[1], ref)
    _log(report_dir, "fast_vs_general_linear", M=M, N=N, K=K, act=act, err=err)
    assert torch.equal(out[0], out[1])
    assert err < 2e-6


FAST_CONV_CASES = [
    # nb, T, cin, cout, k, stride, pad, dil, in_act, act, lens
    (2, 50, 1024, 256, 3, 1, 1, 1, 0, 1, [50, 37]),
    (3, 520, 128, 128, 7, 1, 3, 1, 0, 0, [520, 20, 333]),
    (2, 499, 128, 256, 8, 8, 4, 1, 0, 0, None),
    (1, 3000, 64, 64, 11, 1, 25, 5, 1, 0, None),
    (2, 9000, 32, 32, 3, 1, 3, 3, 1, 0, None),
    (1, 4000, 32, 1, 7, 1, 3, 1, 2, 3, None),
    (1, 40, 1792, 512, 7, 1, 3, 1, 0, 0, None),
]


@pytest.mark.parametrize("case", FAST_CONV_CASES)
def test_fast_gemm_bit_identical_to_general_conv(lib, report_dir, case):
    nb, T, cin, cout, k, stride, pad, dil, in_act, act, lens = case
    g = torch.Generator().manual_seed(T * 31 + cin + k)
    x = dev(torch.randn(nb, T, cin, generator=g))
    w = (torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)).half()
    b = dev(torch.randn(cout, generator=g) * 0.1)
    t_out = (T + 2 * pad - dil * (k - 1) - 1) // stride + 1
    res = dev(torch.randn(nb, t_out, cout, generator=g))
    wp = torch.zeros(cout, cin * k, dtype=torch.float16, device="cuda")
    check(lib, lib.sc_op_pack_conv_weight(P(dev(w)), P(wp), cout, cin, k))
    d_lens = dev(torch.tensor(lens, dtype=torch.int32)) if lens is not None else None
    out = []
    for general in (1, 0):
        check(lib, lib.sc_op_force_general_gemm(general))
        y = torch.full((nb, t_out, cout), float("nan"), device="cuda")
        check(lib, lib.sc_op_conv1d(P(x), P(wp), P(b), P(res), P(y), nb, T, cin, cout, k, stride, pad, dil, P(d_lens), in_act, act))
        out.append(y.cpu())
    check(lib, lib.sc_op_force_general_gemm(0))
    _log(report_dir, "fast_vs_general_conv", case=case)
    assert not torch.isnan(out[1]).any()
    assert torch.equal(out[0], out[1])


@pytest.mark.parametrize("nb,T,act", [(1, 4000, 3), (3, 1000, 3), (2, 255, 0), (1, 7, 3), (2, 257, 3)])
def test_conv_to_one_channel_matches_torch_and_the_gemm_path(lib, report_dir, nb, T, act):
    """The vocoder's conv_post shape (16 -> 1 channel, k = 7, LeakyReLU(0.01) in front, tanh behind) runs on a direct fp32
    kernel (k_misc.hip: conv_to_mono_kernel) instead of an MFMA tile with one live column: against a float64 reference, and
    against the GEMM path (which the general-kernel switch still selects) - two fp32-accurate evaluations of the same sum."""
    cin, cout, k = 16, 1, 7
    g = torch.Generator().manual_seed(T + nb)
    x = torch.randn(nb, T, cin, generator=g) * 1.5
    w = (torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)).half()
    b = torch.randn(cout, generator=g) * 0.1
    wp = torch.zeros(cout, 128, dtype=torch.float16, device="cuda")  # packed row, padded to a multiple of 32
    check(lib, lib.sc_op_pack_conv_weight(P(dev(w)), P(wp), cout, cin, k))
    ref = F.conv1d(F.leaky_relu(x.double(), 0.01).transpose(1, 2), w.double(), b.double(), padding=3).transpose(1, 2)
    if act == 3:
        ref = torch.tanh(ref)
    out = []
    for general in (1, 0):
        check(lib, lib.sc_op_force_general_gemm(general))
        y = torch.full((nb, T, cout), float("nan"), device="cuda")
        check(lib, lib.sc_op_conv1d(P(dev(x)), P(wp), P(dev(b)), P(None), P(y), nb, T, cin, cout, k, 1, 3, 1, P(None), 2, act))
        out.append(y.cpu())
    check(lib, lib.sc_op_force_general_gemm(0))
    err = float((out[1].double() - ref).abs().max())
    err_g = float((out[0].double() - ref).abs().max())
    _log(report_dir, "conv_to_mono", nb=nb, T=T, act=act, err=err, err_gemm_path=err_g)
    assert err < 2e-6 and err_g < 2e-6


@pytest.mark.parametrize("nb,T,cin,cout,k,s", [(2, 250, 512, 256, 11, 5), (1, 4000, 64, 32, 4, 2), (3, 77, 32, 16, 8, 4)])
def test_fast_gemm_bit_identical_to_general_conv_transpose(lib, report_dir, nb, T, cin, cout, k, s):
    g = torch.Generator().manual_seed(T + cin + k)
    x = dev(torch.randn(nb, T, cin, generator=g))
    v = dev((torch.randn(cin, cout, k, generator=g) / math.sqrt(cin * k)).half())
    gg = dev((torch.rand(cin, 1, 1, generator=g) + 0.5).half())
    b = dev(torch.randn(cout, generator=g) * 0.1)
    pad = (k - s) // 2
    t_out = (T - 1) * s - 2 * pad + k
    out = []
    for general in (1, 0):
        check(lib, lib.sc_op_force_general_gemm(general))
        y = torch.full((nb, t_out, cout), float("nan"), device="cuda")
        check(lib, lib.sc_op_conv_transpose1d(P(x), P(v), P(gg), P(b), P(y), nb, T, cin, cout, k, s, pad, 1))
        out.append(y.cpu())
    check(lib, lib.sc_op_force_general_gemm(0))
    assert not torch.isnan(out[1]).any()
    assert torch.equal(out[0], out[1])


PRESPLIT_SHAPES = [
    (7984, 4096, 1024),  # 256x256 tiles (8 waves), 32 slabs
    (15968, 1024, 1024),  # 256x256 tiles, exactly one round of 252 tiles (the 32-utterance encoder slice)
    (20011, 2000, 96),   # 256x256 tiles, ragged M and N, 3 slabs
    (4100, 1024, 4096),  # 128x128 tiles, ragged M
    (20000, 2048, 96),   # 3 slabs
    (20000, 2048, 64),   # 2 slabs
    (20000, 2048, 32),   # 1 slab
    (499, 1024, 1024),   # 64x64 tiles
    (333, 100, 160),     # ragged N, 5 slabs
    (40, 3072, 1024),
    (1, 64, 32),
]


@pytest.mark.parametrize("M,N,K", PRESPLIT_SHAPES)
@pytest.mark.parametrize("act,with_res", [(0, False), (2, True)])
def test_presplit_gemm_bit_identical_to_split_gemm(lib, report_dir, M, N, K, act, with_res):
    """k_gemm_ps.hip (activation pre-split into two fp16 planes, operands DMA'd global -> LDS, XOR-swizzled tiles,
    three stages) against the on-the-fly split product: identical bits; the split-plane output of the epilogue is
    hi = fp16(y), lo = fp16(y - hi)."""
    g = torch.Generator().manual_seed(M + 5 * N + 11 * K)
    x = dev(torch.randn(M, K, generator=g) * 2.0)
    w = dev((torch.randn(N, K, generator=g) / math.sqrt(K)).half())
    b = dev(torch.randn(N, generator=g) * 0.1)
    r = dev(torch.randn(M, N, generator=g)) if with_res else None
    want = torch.full((M, N), float("nan"), device="cuda")
    check(lib, lib.sc_op_linear(P(x), P(w), P(b), P(r), P(want), M, N, K, act, 0.5, 1, 0))
    got = torch.full((M, N), float("nan"), device="cuda")
    gh = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16)
    gl = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16)
    check(lib, lib.sc_op_linear_presplit(P(x), P(w), P(b), P(r), P(got), P(gh), P(gl), M, N, K, act, 0.5))
    got, want = got.cpu(), want.cpu()
    _log(report_dir, "presplit_gemm", M=M, N=N, K=K, act=act, equal=bool(torch.equal(got, want)),
         maxdiff=float((got - want).abs().max()))
    assert not torch.isnan(got).any()
    assert torch.equal(got, want)
    hi = got.half()
    assert torch.equal(gh.cpu(), hi)
    assert torch.equal(gl.cpu(), (got - hi.float()).half())
    # split planes only (no fp32 output)
    gh2 = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16)
    gl2 = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16)
    check(lib, lib.sc_op_linear_presplit(P(x), P(w), P(b), P(r), None, P(gh2), P(gl2), M, N, K, act, 0.5))
    assert torch.equal(gh2.cpu(), hi)


@pytest.mark.parametrize("M,N,K", [(3001, 10082, 1024), (4100, 1000, 128), (499, 1030, 64), (70, 10082, 1024), (1, 64, 32)])
@pytest.mark.parametrize("ties", ["none", "adjacent", "far"])
def test_presplit_gemm_fused_argmax(lib, report_dir, M, N, K, ties):
    """The arg-max riding in the epilogue of the DMA GEMM (the unit projection of the NAR T2U; reference
    models/unity/model.py:438-441 + inference/generator.py:346) against the arg-max of the logits the same kernel writes out:
    equal, lowest index among equal values - `adjacent`: every odd column repeats the even one before it (ties inside a
    wave's chunk), `far`: the second half of the columns repeats the first (ties across chunks and tiles).  Shapes: 256 x 256,
    128 x 128 and 64 x 64 tiles, N not a multiple of the tile or of 4."""
    g = torch.Generator().manual_seed(3 * M + 5 * N + 11 * K)
    x = torch.randn(M, K, generator=g) * 2.0
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half()
    b = torch.randn(N, generator=g) * 0.1
    if ties == "adjacent":
        n2 = N // 2
        w[1 : 2 * n2 : 2] = w[0 : 2 * n2 : 2]
        b[1 : 2 * n2 : 2] = b[0 : 2 * n2 : 2]
    elif ties == "far":
        h = N // 2
        w[N - h :] = w[:h]
        b[N - h :] = b[:h]
    x, w, b = dev(x), dev(w), dev(b)
    logits = torch.full((M, N), float("nan"), device="cuda")
    check(lib, lib.sc_op_linear_presplit(P(x), P(w), P(b), None, P(logits), None, None, M, N, K, 0, 1.0))
    want = np.argmax(logits.cpu().numpy(), axis=1)  # first occurrence of the maximum
    got = torch.full((M,), -7, device="cuda", dtype=torch.int32)
    check(lib, lib.sc_op_linear_presplit_argmax(P(x), P(w), P(b), P(got), M, N, K))
    got = got.cpu().numpy()
    _log(report_dir, "presplit_gemm_fused_argmax", M=M, N=N, K=K, ties=ties, equal=int((got == want).sum()), rows=M)
    assert np.array_equal(got, want)
    if ties == "adjacent":
        assert (got % 2 == 0).all() or N % 2 == 1
    if ties == "far" and N % 2 == 0:
        assert (got < N // 2).all()
    # without a bias
    check(lib, lib.sc_op_linear_presplit(P(x), P(w), None, None, P(logits), None, None, M, N, K, 0, 1.0))
    got2 = torch.full((M,), -7, device="cuda", dtype=torch.int32)
    check(lib, lib.sc_op_linear_presplit_argmax(P(x), P(w), None, P(got2), M, N, K))
    assert np.array_equal(got2.cpu().numpy(), np.argmax(logits.cpu().numpy(), axis=1))


CONV_PS_CASES = [
    # nb, T, cin, cout, k, dil, act, with_res, masked
    (3, 700, 1024, 1024, 7, 1, 1, False, True),   # the T2U FFT-decoder convolution (64 x 64 / 128 x 128 tiles), masked tail rows
    (2, 333, 256, 512, 3, 1, 0, True, False),     # ragged item length, residual
    (32, 500, 128, 1024, 3, 2, 0, False, False),  # 256 x 256 tiles (252 of them), dilation 2, K slabs cross taps
    (1, 5, 64, 96, 11, 1, 0, True, False),        # shorter than the kernel
]


@pytest.mark.parametrize("case", CONV_PS_CASES)
def test_presplit_conv_bit_identical_to_conv1d(lib, report_dir, case):
    """Implicit-convolution mode of the DMA GEMM (k_gemm_ps.hip: per-slab tap offset on the row address, rows of another
    item as out-of-range offsets) against sc_op_conv1d on the same values: identical bits; rows flagged dead are exact
    zeros in the fp32 output and in both planes."""
    nb, T, cin, cout, k, dil, act, with_res, masked = case
    pad = dil * (k - 1) // 2
    g = torch.Generator().manual_seed(T * 7 + cin + k)
    x = dev(torch.randn(nb, T, cin, generator=g) * 1.5)
    w = (torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)).half()
    b = dev(torch.randn(cout, generator=g) * 0.1)
    r = dev(torch.randn(nb, T, cout, generator=g)) if with_res else None
    kpad = cin * k
    assert kpad % 32 == 0
    wp = torch.zeros(cout, kpad, dtype=torch.float16, device="cuda")
    check(lib, lib.sc_op_pack_conv_weight(P(dev(w)), P(wp), cout, cin, k))
    want = torch.full((nb, T, cout), float("nan"), device="cuda")
    check(lib, lib.sc_op_conv1d(P(x), P(wp), P(b), P(r), P(want), nb, T, cin, cout, k, 1, pad, dil, None, 0, act))
    valid = None
    if masked:
        valid = torch.ones(nb, T, dtype=torch.uint8)
        valid[0, T - 37:] = 0
        valid[2, 100:] = 0
        valid = dev(valid)
    got = torch.full((nb, T, cout), float("nan"), device="cuda")
    gh = torch.full((nb, T, cout), float("nan"), device="cuda", dtype=torch.float16)
    gl = torch.full((nb, T, cout), float("nan"), device="cuda", dtype=torch.float16)
    check(lib, lib.sc_op_conv1d_presplit(P(x), P(wp), P(b), P(r), P(got), P(gh), P(gl), nb, T, cin, cout, k, pad, dil, P(valid), act))
    got, want = got.cpu(), want.cpu()
    if masked:
        dead = valid.cpu() == 0
        assert float(got[dead].abs().max()) == 0.0 and float(gh.cpu()[dead].abs().max()) == 0.0 and float(gl.cpu()[dead].abs().max()) == 0.0
        want[dead] = 0
    _log(report_dir, "presplit_conv", case=case, equal=bool(torch.equal(got, want)), maxdiff=float((got - want).abs().max()))
    assert not torch.isnan(got).any()
    assert torch.equal(got, want)
    hi = got.half()
    assert torch.equal(gh.cpu(), hi)
    assert torch.equal(gl.cpu(), (got - hi.float()).half())


RESPAIR_CASES = [
    # nb, T, C, k, dil, avg
    (2, 1000, 32, 3, 1, False),
    (2, 1000, 32, 11, 5, True),
    (1, 257, 32, 7, 3, False),
    (3, 777, 16, 11, 5, False),
    (2, 4100, 16, 3, 3, True),
    (1, 100, 16, 7, 1, False),  # shorter than one tile
    (2, 900, 64, 11, 5, True),
    (1, 1300, 64, 7, 3, False),
    (2, 118, 64, 3, 1, False),
]


@pytest.mark.parametrize("nb,T,C_,k,dil,avg", RESPAIR_CASES)
def test_resblock_pair_bit_identical_to_two_convs(lib, report_dir, nb, T, C_, k, dil, avg):
    """k_resblock.hip (one ResBlock dilation pair, intermediate in LDS) against the two implicit-GEMM launches
    it replaces: identical bits; and against F.conv1d in float64 within the split-product accuracy."""
    g = torch.Generator().manual_seed(T * 13 + C_ * 5 + k + dil)
    x = torch.randn(nb, T, C_, generator=g)
    w1 = (torch.randn(C_, C_, k, generator=g) / math.sqrt(C_ * k)).half()
    w2 = (torch.randn(C_, C_, k, generator=g) / math.sqrt(C_ * k)).half()
    b1, b2 = torch.randn(C_, generator=g) * 0.1, torch.randn(C_, generator=g) * 0.1
    ra, rb = torch.randn(nb, T, C_, generator=g), torch.randn(nb, T, C_, generator=g)
    kpad = (C_ * k + 31) // 32 * 32
    wp1 = torch.zeros(C_, kpad, dtype=torch.float16, device="cuda")
    wp2 = torch.zeros(C_, kpad, dtype=torch.float16, device="cuda")
    check(lib, lib.sc_op_pack_conv_weight(P(dev(w1)), P(wp1), C_, C_, k))
    check(lib, lib.sc_op_pack_conv_weight(P(dev(w2)), P(wp2), C_, C_, k))
    dx, db1, db2 = dev(x), dev(b1), dev(b2)
    tmp = torch.full((nb, T, C_), float("nan"), device="cuda")
    two = torch.full((nb, T, C_), float("nan"), device="cuda")
    check(lib, lib.sc_op_conv1d(P(dx), P(wp1), P(db1), None, P(tmp), nb, T, C_, C_, k, 1, dil * (k - 1) // 2, dil, None, 1, 0))
    check(lib, lib.sc_op_conv1d(P(tmp), P(wp2), P(db2), P(dx), P(two), nb, T, C_, C_, k, 1, (k - 1) // 2, 1, None, 1, 0))
    want = two.cpu()
    if avg:
        want = torch.from_numpy(((ra.numpy() + rb.numpy()) + want.numpy()) / np.float32(3.0))
    got = torch.full((nb, T, C_), float("nan"), device="cuda")
    check(lib, lib.sc_op_resblock_pair(P(dx), P(wp1), P(db1), P(wp2), P(db2), P(got), nb, T, C_, k, dil, 0.1,
                                       P(dev(ra)) if avg else None, P(dev(rb)) if avg else None))
    got = got.cpu()
    assert not torch.isnan(got).any()
    xt = F.leaky_relu(x, 0.1).transpose(1, 2).double()
    h = F.conv1d(xt, w1.double(), b1.double(), padding=dil * (k - 1) // 2, dilation=dil)
    ref = F.conv1d(F.leaky_relu(h, 0.1), w2.double(), b2.double(), padding=(k - 1) // 2).transpose(1, 2) + x.double()
    if avg:
        ref = (ra.double() + rb.double() + ref) / 3.0
    err = rel_err(got, ref)
    _log(report_dir, "resblock_pair", nb=nb, T=T, C=C_, k=k, dil=dil, avg=avg, err=err, bit_identical=bool(torch.equal(got, want)))
    assert err < 3e-6
    assert torch.equal(got, want)


MRF_CASES = [
    # nb, T, C, kernel sizes, dilations per block
    (2, 1000, 32, (3, 7, 11), (1, 3, 5)),
    (1, 392, 32, (3, 7, 11), (1, 3, 5)),    # exactly one tile
    (3, 393, 16, (3, 7, 11), (1, 3, 5)),    # one row into the second tile
    (2, 4100, 16, (3, 7, 11), (1, 3, 5)),
    (1, 37, 16, (3, 7, 11), (1, 3, 5)),     # shorter than the halo
    (2, 901, 32, (11, 3, 7), (5, 1, 3)),    # other orders: the halo is the widest block's, whichever it is
    (1, 650, 16, (5, 5, 9), (2, 1, 4)),
]


@pytest.mark.parametrize("nb,T,C_,ks,dils", MRF_CASES)
def test_mrf_fused_bit_identical_to_nine_pairs(lib, report_dir, nb, T, C_, ks, dils):
    """k_resblock.hip: mrf_fused_kernel (three ResBlocks x three dilation pairs + the average in one kernel, residual stream
    in registers, intermediates in LDS) against the nine pair launches it replaces: identical bits; and against the
    float64 HiFi-GAN block (hifigan.py:37-127, 186-191)."""
    import ctypes as C

    g = torch.Generator().manual_seed(T * 7 + C_ * 3 + ks[0])
    x = torch.randn(nb, T, C_, generator=g)
    W1, W2, B1, B2, P1, P2, D1, D2 = [], [], [], [], [], [], [], []
    for k in ks:
        kpad = (C_ * k + 31) // 32 * 32
        for _ in dils:
            w1 = (torch.randn(C_, C_, k, generator=g) / math.sqrt(C_ * k)).half()
            w2 = (torch.randn(C_, C_, k, generator=g) / math.sqrt(C_ * k)).half()
            b1, b2 = torch.randn(C_, generator=g) * 0.1, torch.randn(C_, generator=g) * 0.1
            wp1 = torch.zeros(C_, kpad, dtype=torch.float16, device="cuda")
            wp2 = torch.zeros(C_, kpad, dtype=torch.float16, device="cuda")
            check(lib, lib.sc_op_pack_conv_weight(P(dev(w1)), P(wp1), C_, C_, k))
            check(lib, lib.sc_op_pack_conv_weight(P(dev(w2)), P(wp2), C_, C_, k))
            W1.append(w1), W2.append(w2), B1.append(b1), B2.append(b2), P1.append(wp1), P2.append(wp2)
            D1.append(dev(b1)), D2.append(dev(b2))
    dx = dev(x)
    # the nine-launch chain
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
    # one launch
    arr = lambda ts: (C.c_void_p * 9)(*[t.data_ptr() for t in ts])
    kk = (C.c_int32 * 3)(*ks)
    dd = (C.c_int32 * 9)(*[dil for _ in ks for dil in dils])
    got = torch.full((nb, T, C_), float("nan"), device="cuda")
    check(lib, lib.sc_op_mrf_fused(P(dx), arr(P1), arr(D1), arr(P2), arr(D2), P(got), nb, T, C_, kk, dd, 0.1))
    got = got.cpu()
    assert not torch.isnan(got).any()
    # float64 reference
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
    _log(report_dir, "mrf_fused", nb=nb, T=T, C=C_, ks=list(ks), dils=list(dils), err=err, bit_identical=bool(torch.equal(got, want)),
         maxdiff=float((got - want).abs().max()))
    assert err < 5e-6
    assert torch.equal(got, want)


CONVT_CASES = [(2, 25, 64, 32, 11, 5), (1, 100, 32, 16, 8, 4), (2, 77, 16, 8, 4, 2), (1, 13, 512, 256, 11, 5)]


@pytest.mark.parametrize("nb,T,cin,cout,k,s", CONVT_CASES)
def test_conv_transpose1d_weight_norm(lib, report_dir, nb, T, cin, cout, k, s):
    g = torch.Generator().manual_seed(T + cin + k)
    x = torch.randn(nb, T, cin, generator=g)
    v = (torch.randn(cin, cout, k, generator=g) / math.sqrt(cin * k)).half()
    gg = (torch.rand(cin, 1, 1, generator=g) + 0.5).half()
    b = torch.randn(cout, generator=g) * 0.1
    pad = (k - s) // 2
    wn = gg.double() * v.double() / v.double().reshape(cin, -1).norm(dim=1).reshape(cin, 1, 1)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1).transpose(1, 2).double(), wn, b.double(), stride=s, padding=pad).transpose(1, 2)
    y = torch.full((nb, T * s, cout), float("nan"), device="cuda")
    check(lib, lib.sc_op_conv_transpose1d(P(dev(x)), P(dev(v)), P(dev(gg)), P(dev(b)), P(y), nb, T, cin, cout, k, s, pad, 1))
    err = rel_err(y.cpu(), ref)
    _log(report_dir, "conv_transpose1d", nb=nb, T=T, cin=cin, cout=cout, k=k, s=s, err=err)
    # folded weights are rounded to fp16 once (2^-11 relative per weight)
    assert err < 1e-3
    assert not torch.isnan(y).any()


def _attn_ref(q, k, v, lens, causal, rel, left, right):
    # q (nb,H,Sq,64) ...
    nb, H, Sq, D = q.shape
    Skv = k.shape[2]
    w = (q.double() @ k.double().transpose(-1, -2)) * D ** -0.5
    if rel is not None:
        idx = torch.arange(Skv)[None, :] - torch.arange(Skv)[:, None]
        idx = idx.clamp(-left, right) + left
        rk = rel.double()[idx][-Sq:]
        w = w + torch.einsum("nhsm,stm->nhst", q.double(), rk) * D ** -0.5
    if causal:
        cm = torch.ones(Sq, Skv, dtype=torch.bool).tril(diagonal=Skv - Sq)
        w = w.masked_fill(~cm, float("-inf"))
    if lens is not None:
        km = torch.arange(Skv)[None, :] < torch.tensor(lens)[:, None]
        w = w.masked_fill(~km[:, None, None, :], float("-inf"))
    return torch.softmax(w, -1) @ v.double()


ATTN_CASES = [
    # nb, H, Sq, Skv, lens, causal, shaw
    (2, 4, 499, 499, [499, 310], False, True),
    (1, 16, 63, 63, None, False, False),
    (2, 2, 40, 40, [40, 17], True, False),
    (1, 2, 130, 130, None, False, True),
    (2, 2, 5, 70, [70, 3], False, False),
    (1, 3, 200, 200, [129], True, False),
]


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention(lib, report_dir, case):
    nb, H, Sq, Skv, lens, causal, shaw = case
    g = torch.Generator().manual_seed(Sq * 13 + Skv)
    M = H * 64
    q = torch.randn(nb, Sq, M, generator=g)
    k = torch.randn(nb, Skv, M, generator=g)
    v = torch.randn(nb, Skv, M, generator=g)
    rel = torch.randn(73, 64, generator=g) * 0.3 if shaw else None

    def heads(t, S):
        return t.view(nb, S, H, 64).transpose(1, 2)

    ref = _attn_ref(heads(q, Sq), heads(k, Skv), heads(v, Skv), lens, causal, rel, 64, 8).transpose(1, 2).reshape(nb, Sq, M)
    out = torch.full((nb, Sq, M), float("nan"), device="cuda")
    d_lens = dev(torch.tensor(lens, dtype=torch.int32)) if lens is not None else None
    check(lib, lib.sc_op_attention(P(dev(q)), P(dev(k)), P(dev(v)), P(out), nb, H, Sq, Skv, M, M, M, M, P(d_lens),
                                   int(causal), P(dev(rel)) if shaw else None, 64 if shaw else 0, 8 if shaw else 0))
    err = float((out.cpu().double() - ref).abs().max())
    _log(report_dir, "attention", case=case, err=err)
    assert err < 2e-5


@pytest.mark.parametrize("nb,T,C_,lens", [(2, 70, 256, [70, 33]), (1, 499, 1024, None), (3, 5, 128, [5, 1, 3])])
def test_glu_dwconv(lib, report_dir, nb, T, C_, lens):
    g = torch.Generator().manual_seed(T + C_)
    x = torch.randn(nb, T, 2 * C_, generator=g)
    w = torch.randn(C_, 31, generator=g) * 0.2
    gl = F.glu(x.double(), dim=-1)
    if lens is not None:
        for i, l in enumerate(lens):
            gl[i, l:] = 0
    ref = F.conv1d(F.pad(gl.transpose(1, 2), (30, 0)), w.double().unsqueeze(1), groups=C_).transpose(1, 2)
    y = torch.full((nb, T, C_), float("nan"), device="cuda")
    d_lens = dev(torch.tensor(lens, dtype=torch.int32)) if lens is not None else None
    check(lib, lib.sc_op_glu_dwconv(P(dev(x)), P(dev(w)), P(y), nb, T, C_, 31, P(d_lens)))
    err = float((y.cpu().double() - ref).abs().max())
    _log(report_dir, "glu_dwconv", nb=nb, T=T, C=C_, err=err)
    assert err < 2e-5


@pytest.mark.parametrize("rows,V", [(1, 256102), (7, 10082), (64, 1200)])
def test_argmax_lprob(lib, report_dir, rows, V):
    g = torch.Generator().manual_seed(V)
    x = torch.randn(rows, V, generator=g) * 3
    x[0, 5] = x[0].max() + 1.0
    x[0, 9] = x[0, 5]  # tie -> lowest index wins
    idx = torch.empty(rows, dtype=torch.int32, device="cuda")
    lp = torch.empty(rows, device="cuda")
    check(lib, lib.sc_op_argmax(P(dev(x)), rows, V, P(idx), P(lp)))
    ref_lp = torch.log_softmax(x.double(), -1).max(-1).values
    assert idx.cpu().tolist() == x.argmax(-1).tolist()
    assert int(idx[0]) == 5
    err = float((lp.cpu().double() - ref_lp).abs().max())
    _log(report_dir, "argmax", rows=rows, V=V, err=err)
    assert err < 1e-5


@pytest.mark.parametrize("rows,V", [(1, 256102), (7, 10082), (64, 1200), (5, 1201), (1030, 10082)])
def test_argmax_without_lprob_takes_the_plain_kernel(lib, rows, V):
    """No log-probability asked for (the NAR T2U unit projection): one wave per row, 8-byte loads, no exp (k_misc.hip:
    argmax_plain_rows_kernel; odd V falls back to the general kernel).  Ties -> lowest index; a row of -inf -> index 0."""
    g = torch.Generator().manual_seed(V + rows)
    x = torch.randn(rows, V, generator=g) * 3
    x[0, V - 1] = x[0].max() + 1.0
    x[0, 9] = x[0, V - 1]  # tie -> lowest index wins
    if rows > 2:
        x[2] = float("-inf")
        x[1, 1::2] = 7.0  # ties across lanes and inside a load
        x[1, 0::2] = -7.0
    idx = torch.full((rows,), -5, dtype=torch.int32, device="cuda")
    check(lib, lib.sc_op_argmax(P(dev(x)), rows, V, P(idx), P(None)))
    want = x.argmax(-1).tolist()
    want[0] = 9
    if rows > 2:
        want[1], want[2] = 1, 0
    assert idx.cpu().tolist() == want


SKINNY_SHAPES = [
    (1, 1024, 1024), (16, 3072, 1024), (16, 8192, 1024), (16, 1024, 8192), (32, 1024, 1024), (33, 1024, 1024),
    (64, 3072, 1024), (7, 100, 128), (16, 256102, 1024), (5, 1200, 256), (16, 70, 64),
]


@pytest.mark.parametrize("M,N,K", SKINNY_SHAPES)
def test_skinny_linear(lib, report_dir, M, N, K):
    """Decoder-step product (1..64 rows): same fp32-class accuracy bar as the big GEMM."""
    g = torch.Generator().manual_seed(M * 11 + N * 5 + K)
    x = torch.randn(M, K, generator=g) * 2.0
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half()
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g)
    ref = 0.5 * torch.relu(x.double() @ w.double().t() + b.double()) + r.double()
    y = torch.full((M, N), float("nan"), device="cuda")
    check(lib, lib.sc_op_skinny_linear(P(dev(x)), P(dev(w)), P(dev(b)), P(dev(r)), P(y), M, N, K, 1, 0.5))
    err = rel_err(y.cpu(), ref)
    _log(report_dir, "skinny_linear", M=M, N=N, K=K, err=err)
    assert err < 2e-6, err
    # through the generic entry point (rows <= 64 dispatches to the same kernel inside the stages)
    y0 = torch.empty(M, N, device="cuda")
    check(lib, lib.sc_op_skinny_linear(P(dev(x)), P(dev(w)), P(None), P(None), P(y0), M, N, K, 0, 1.0))
    assert rel_err(y0.cpu(), x.double() @ w.double().t()) < 2e-6


@pytest.mark.parametrize("M,N,K,splits", [(16, 1024, 1024, 0), (16, 1024, 8192, 0), (16, 1024, 8192, 4), (3, 128, 256, 0),
                                          (64, 1024, 1024, 2), (16, 1024, 1024, 1)])
def test_skinny_split_k_residual_layernorm(lib, report_dir, M, N, K, splits):
    """x += in.W^T + b via K-range partials, then LayerNorm; repeated calls must be bit-identical."""
    g = torch.Generator().manual_seed(M + N + K + splits)
    inp = torch.randn(M, K, generator=g)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half()
    b = torch.randn(N, generator=g) * 0.1
    x0 = torch.randn(M, N, generator=g)
    gam = torch.rand(N, generator=g) + 0.5
    bet = torch.randn(N, generator=g) * 0.1
    xr = x0.double() + inp.double() @ w.double().t() + b.double()
    hr = F.layer_norm(xr, (N,), gam.double(), bet.double(), 1e-5)
    outs = []
    for _ in range(2):
        x = dev(x0.clone())
        h = torch.empty(M, N, device="cuda")
        check(lib, lib.sc_op_skinny_res_ln(P(dev(inp)), P(dev(w)), P(dev(b)), P(x), P(dev(gam)), P(dev(bet)), P(h), M, N, K, splits))
        outs.append((x.cpu(), h.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ex, eh = rel_err(outs[0][0], xr), float((outs[0][1].double() - hr).abs().max())
    _log(report_dir, "skinny_res_ln", M=M, N=N, K=K, splits=splits, err_x=ex, err_h=eh)
    assert ex < 2e-6 and eh < 2e-5


@pytest.mark.parametrize("M,N,K", [(16, 256102, 1024), (1, 256102, 1024), (5, 1200, 128), (40, 10082, 1024)])
@pytest.mark.parametrize("mode", ["plain", "no_eos", "force_eos", "unk_pen"])
def test_skinny_fused_argmax(lib, report_dir, M, N, K, mode):
    """Vocabulary projection with the arg-max / log-softmax folded into the epilogue vs torch."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half()
    pad, unk, eos = 0, 1, 3
    logits = x.double() @ w.double().t()
    # make the rule under test matter: push the special columns to the top of some rows
    boost = logits.max(dim=1).values + 1.0
    step, min_eos, force, pen = 5, 0, -1, 0.0
    wq = w.clone()
    if mode == "no_eos":
        min_eos = 10
    elif mode == "force_eos":
        force = 5
    elif mode == "unk_pen":
        pen = 1e9
    logits = x.double() @ wq.double().t()
    lsm = torch.log_softmax(logits, dim=1)
    t = logits.clone()
    t[:, unk] -= pen
    t[:, pad] = -float("inf")
    if step < min_eos:
        t[:, eos] = -float("inf")
    if force == step:
        keep = t[:, eos].clone()
        t[:] = -float("inf")
        t[:, eos] = keep
    ref_idx = t.argmax(dim=1)
    ref_lp = (t.gather(1, ref_idx[:, None])[:, 0] - torch.logsumexp(logits, dim=1))
    idx = torch.empty(M, dtype=torch.int32, device="cuda")
    lp = torch.empty(M, device="cuda")
    check(lib, lib.sc_op_skinny_argmax(P(dev(x)), P(dev(wq)), M, N, K, step, min_eos, force, pad, eos, unk, pen, P(idx), P(lp)))
    assert idx.cpu().tolist() == ref_idx.tolist()
    err = float((lp.cpu().double() - ref_lp).abs().max()) if mode != "unk_pen" else 0.0
    _log(report_dir, "skinny_argmax", M=M, N=N, K=K, mode=mode, err=err)
    assert err < 1e-4
