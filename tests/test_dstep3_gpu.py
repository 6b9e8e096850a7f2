"""Kernel-level parity of the third-generation decoder-step kernels (csrc/k_dstep3.hip: row-group products that apply
the preceding LayerNorm and the bias / residual / ReLU themselves; LDS-staged vocabulary projection) against plain
PyTorch fp64 restatements, through the C ABI (sc_op_dstep3_*).

Accuracy bars as for generation 2 (tests/test_dstep_gpu.py): products ~2e-6 relative on un-normalised inputs, 2e-5
absolute behind a LayerNorm, arg-max indices exact.  Unused row slots of the k-group-major buffers are filled with NaN
by the op entry points: a kernel that reads them fails these tests.  Results must not depend on the rows per row group.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.test_ops_gpu import P, check, dev, lib, rel_err, _release_device_copies  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def _log(report_dir, name, **kw):
    with open(report_dir / "ops_report.txt", "a") as f:
        f.write(name + " " + " ".join(f"{k}={v}" for k, v in kw.items()) + "\n")


def _case(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g) * 1.5 + 0.3  # a LayerNorm input with a mean
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half()
    b = torch.randn(N, generator=g) * 0.1
    gam = torch.rand(K, generator=g) + 0.5
    bet = torch.randn(K, generator=g) * 0.1
    return x, w, b, gam, bet


# QKV (N = 3 M), q projection, tiny-model shapes, ragged row counts; rg = rows per row group
@pytest.mark.parametrize("M,N,K,rg", [
    (1, 3072, 1024, 16), (64, 3072, 1024, 32), (64, 3072, 1024, 16), (33, 1024, 1024, 8), (20, 1024, 1024, 16), (64, 1024, 1024, 16),
    (5, 384, 128, 16), (40, 128, 128, 32), (17, 104, 64, 8),
])
def test_gemv3_layernorm_rows(lib, report_dir, M, N, K, rg):
    """y = LayerNorm(x) . W^T + b with the LayerNorm computed inside the product's workgroups."""
    x, w, b, gam, bet = _case(M, N, K, M + N + K)
    ref = F.layer_norm(x.double(), (K,), gam.double(), bet.double(), 1e-5) @ w.double().t() + b.double()
    outs = []
    for r in (rg, 32 if rg != 32 else 16):
        y = torch.full((M, N), float("nan"), device="cuda")
        check(lib, lib.sc_op_dstep3_gemv(0, P(dev(x)), P(dev(w)), P(dev(b)), P(dev(gam)), P(dev(bet)), P(None), P(y), P(None), M, N, K, 0, r, 0))
        outs.append(y.cpu())
    assert torch.equal(outs[0], outs[1]), "the result depends on the row grouping"
    err = float((outs[0].double() - ref).abs().max())
    _log(report_dir, "gemv3_ln_rows", M=M, N=N, K=K, rg=rg, err=err)
    assert err < 2e-5, err


@pytest.mark.parametrize("M,N,K,rg", [(1, 1024, 1024, 16), (64, 1024, 1024, 16), (64, 1024, 1024, 8), (33, 1024, 1024, 32), (7, 128, 128, 16),
                                      (40, 128, 128, 16)])
def test_gemv3_residual(lib, report_dir, M, N, K, rg):
    """x += in . W^T + b finished inside the product (out-projections): no partial sums, no reduce launch."""
    x, w, b, _, _ = _case(M, N, K, 3 * M + N + K)
    res = torch.randn(M, N, generator=torch.Generator().manual_seed(M)) * 2
    ref = res.double() + x.double() @ w.double().t() + b.double()
    y = torch.full((M, N), float("nan"), device="cuda")
    check(lib, lib.sc_op_dstep3_gemv(1, P(dev(x)), P(dev(w)), P(dev(b)), P(None), P(None), P(dev(res)), P(y), P(None), M, N, K, 0, rg, 0))
    err = rel_err(y.cpu(), ref)
    _log(report_dir, "gemv3_resid", M=M, N=N, K=K, rg=rg, err=err)
    assert err < 2e-6, err


@pytest.mark.parametrize("shape", [0, 1])
@pytest.mark.parametrize("M,N,K,act,rg", [(1, 8192, 1024, 1, 32), (64, 8192, 1024, 1, 32), (64, 8192, 1024, 1, 16), (33, 256, 128, 1, 32),
                                          (9, 8192, 1024, 0, 16), (40, 224, 128, 1, 32)])
def test_gemv3_layernorm_planes(lib, report_dir, M, N, K, act, rg, shape):
    """FFN inner projection: act(LayerNorm(x) . W^T + b) leaving the kernel as split fp16 planes."""
    x, w, b, gam, bet = _case(M, N, K, 5 * M + N + K)
    ref = F.layer_norm(x.double(), (K,), gam.double(), bet.double(), 1e-5) @ w.double().t() + b.double()
    if act:
        ref = torch.relu(ref)
    y = torch.full((M, N), float("nan"), device="cuda")
    check(lib, lib.sc_op_dstep3_gemv(2, P(dev(x)), P(dev(w)), P(dev(b)), P(dev(gam)), P(dev(bet)), P(None), P(y), P(None), M, N, K, act, rg, shape))
    err = float((y.cpu().double() - ref).abs().max())
    _log(report_dir, "gemv3_ln_planes", M=M, N=N, K=K, act=act, rg=rg, shape=shape, err=err)
    assert err < 2e-5, err


@pytest.mark.parametrize("shape", [0, 2])
@pytest.mark.parametrize("M,N,K,ln", [(16, 1024, 8192, 0), (64, 1024, 8192, 1), (33, 1024, 8192, 1), (5, 128, 256, 1), (40, 128, 256, 0),
                                      (1, 1024, 8192, 1), (40, 96, 256, 1)])
def test_gemv3_partials_and_reduce(lib, report_dir, M, N, K, ln, shape):
    """FFN output projection: K-slice partial sums + the reduce kernel (bias, residual; optionally the final LayerNorm)."""
    x, w, b, _, _ = _case(M, N, K, 7 * M + N + K)
    g = torch.Generator().manual_seed(M + 1)
    res = torch.randn(M, N, generator=g) * 2
    gam = torch.rand(N, generator=g) + 0.5
    bet = torch.randn(N, generator=g) * 0.1
    ref = res.double() + x.double() @ w.double().t() + b.double()
    href = F.layer_norm(ref, (N,), gam.double(), bet.double(), 1e-5)
    y = torch.full((M, N), float("nan"), device="cuda")
    h = torch.full((M, N), float("nan"), device="cuda")
    check(lib, lib.sc_op_dstep3_gemv(3, P(dev(x)), P(dev(w)), P(dev(b)), P(dev(gam) if ln else None), P(dev(bet) if ln else None),
                                     P(dev(res)), P(y), P(h), M, N, K, 0, 0, shape))
    err = rel_err(y.cpu(), ref)
    eh = float((h.cpu().double() - href).abs().max()) if ln else 0.0
    _log(report_dir, "gemv3_partial_reduce", M=M, N=N, K=K, ln=ln, shape=shape, err=err, err_h=eh)
    assert err < 2e-6 and eh < 2e-5


# Wide steps (decode engine, beam search): the FFN products with the weights stationary - a workgroup keeps its tiles'
# fragments in registers and walks the row groups (gemv3s_kernel).  `shape` bits 4..7 of the op entry: 15 = one workgroup
# per row group (gemv3_kernel), k = stationary with k workgroups per tile.  Same bits whatever the walk.
@pytest.mark.parametrize("M", [192, 150, 97, 320, 64, 65, 512])
def test_gemv3_stationary_bit_identical(lib, report_dir, M):
    # FFN-in: act(LayerNorm(x) . W1^T + b) as planes, 2 tiles x 8 waves x 8 k-steps
    N, K = 8192, 1024
    x, w, b, gam, bet = _case(M, N, K, 11 * M + N + K)
    ref = torch.relu(F.layer_norm(x.double(), (K,), gam.double(), bet.double(), 1e-5) @ w.double().t() + b.double())
    outs = {}
    for walk in (15, 1, 2, 3, 0):
        y = torch.full((M, N), float("nan"), device="cuda")
        check(lib, lib.sc_op_dstep3_gemv(2, P(dev(x)), P(dev(w)), P(dev(b)), P(dev(gam)), P(dev(bet)), P(None), P(y), P(None), M, N, K, 1, 32,
                                         1 | (walk << 4)))
        outs[walk] = y.cpu()
    err_in = float((outs[15].double() - ref).abs().max())
    assert err_in < 2e-5, err_in
    for walk in (1, 2, 3, 0):
        assert torch.equal(outs[15], outs[walk]), f"FFN-in: stationary walk {walk} differs from one workgroup per row group"
    # FFN-out: K-slice partial sums (64-row groups, 2 x 2 tiles x 8 waves x 4 k-steps) + the reduce kernel
    N, K = 1024, 8192
    x, w, b, _, _ = _case(M, N, K, 13 * M + N + K)
    res = torch.randn(M, N, generator=torch.Generator().manual_seed(M + 2)) * 2
    ref = res.double() + x.double() @ w.double().t() + b.double()
    outs = {}
    for walk in (15, 1, 2, 14, 0):  # 14: tile-owning waves (gemv3t_kernel), 0: whatever the launcher picks
        y = torch.full((M, N), float("nan"), device="cuda")
        check(lib, lib.sc_op_dstep3_gemv(3, P(dev(x)), P(dev(w)), P(dev(b)), P(None), P(None), P(dev(res)), P(y), P(None), M, N, K, 0, 0,
                                         2 | (walk << 4)))
        outs[walk] = y.cpu()
    err_out = rel_err(outs[15], ref)
    assert err_out < 2e-6, err_out
    for walk in (1, 2, 14, 0):
        assert torch.equal(outs[15], outs[walk]), f"FFN-out: walk {walk} differs from one workgroup per row group"
    _log(report_dir, "gemv3_stationary", M=M, err_in=err_in, err_out=err_out)


@pytest.mark.parametrize("M,live", [(192, 150), (192, 31), (320, 257), (128, 128)])
def test_gemv3_wide_kernels_stop_at_the_live_rows(lib, report_dir, M, live):
    """The decode engine / the beam search hand the step a device-side count of live rows (packed to the front): every
    variant of the wide FFN products gives the live rows the bits of the full launch and leaves the rows behind alone
    (`rg` bits 8.. of the op entry = live rows)."""
    N, K = 8192, 1024
    x, w, b, gam, bet = _case(M, N, K, 17 * M + live)
    full = torch.full((M, N), float("nan"), device="cuda")
    check(lib, lib.sc_op_dstep3_gemv(2, P(dev(x)), P(dev(w)), P(dev(b)), P(dev(gam)), P(dev(bet)), P(None), P(full), P(None), M, N, K, 1, 32,
                                     1 | (15 << 4)))
    full = full.cpu()
    for walk in (15, 1, 2, 0):
        y = torch.full((M, N), float("nan"), device="cuda")
        check(lib, lib.sc_op_dstep3_gemv(2, P(dev(x)), P(dev(w)), P(dev(b)), P(dev(gam)), P(dev(bet)), P(None), P(y), P(None), M, N, K, 1,
                                         32 | (live << 8), 1 | (walk << 4)))
        assert torch.equal(y.cpu()[:live], full[:live]), f"FFN-in walk {walk}: live rows differ"
    N, K = 1024, 8192
    x, w, b, _, _ = _case(M, N, K, 19 * M + live)
    res = torch.randn(M, N, generator=torch.Generator().manual_seed(M + live)) * 2
    full = torch.full((M, N), float("nan"), device="cuda")
    check(lib, lib.sc_op_dstep3_gemv(3, P(dev(x)), P(dev(w)), P(dev(b)), P(None), P(None), P(dev(res)), P(full), P(None), M, N, K, 0, 0,
                                     2 | (15 << 4)))
    full = full.cpu()
    for walk in (15, 1, 14, 0):
        y = torch.full((M, N), float("nan"), device="cuda")
        check(lib, lib.sc_op_dstep3_gemv(3, P(dev(x)), P(dev(w)), P(dev(b)), P(None), P(None), P(dev(res)), P(y), P(None), M, N, K, 0,
                                         live << 8, 2 | (walk << 4)))
        y = y.cpu()
        assert torch.equal(y[:live], full[:live]), f"FFN-out walk {walk}: live rows differ"
        assert torch.equal(y[live:], res[live:]), f"FFN-out walk {walk}: a row behind the live rows was touched"
    _log(report_dir, "gemv3_wide_live_rows", M=M, live=live)


@pytest.mark.parametrize("M,N,K", [(16, 256102, 1024), (1, 256102, 1024), (64, 256102, 1024), (33, 256102, 1024), (5, 1200, 128), (40, 10082, 1024),
                                   (33, 1200, 128)])
@pytest.mark.parametrize("mode", ["plain", "no_eos", "force_eos", "unk_pen"])
def test_vocab3_fused_argmax(lib, report_dir, M, N, K, mode):
    """Vocabulary projection (planes staged in LDS, tiles streamed per wave) with the arg-max / log-softmax fused."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half()
    pad, unk, eos = 0, 1, 3
    step, min_eos, force, pen = 5, 0, -1, 0.0
    if mode == "no_eos":
        min_eos = 10
    elif mode == "force_eos":
        force = 5
    elif mode == "unk_pen":
        pen = 1e9
    logits = x.double() @ w.double().t()
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
    check(lib, lib.sc_op_dstep3_argmax(P(dev(x)), P(dev(w)), M, N, K, step, min_eos, force, pad, eos, unk, pen, P(idx), P(lp)))
    assert idx.cpu().tolist() == ref_idx.tolist()
    err = float((lp.cpu().double() - ref_lp).abs().max()) if mode != "unk_pen" else 0.0
    _log(report_dir, "vocab3_argmax", M=M, N=N, K=K, mode=mode, err=err)
    assert err < 1e-4

