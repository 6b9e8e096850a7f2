"""Kernel-level parity of the second-generation decoder-step kernels (csrc/k_dstep.hip: weights packed into MFMA
fragment order, activations as split fp16 planes) against plain PyTorch fp64 restatements, through the C ABI (sc_op_dstep_*).

Same accuracy bars as the first-generation kernels in tests/test_ops_gpu.py: products ~2e-6 relative, LayerNorm output
2e-5 absolute, arg-max indices exact.  Unused row slots of the activation planes are filled with NaN by the op entry
points: a kernel that reads them fails these tests.
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


# every product shape of the decoder step at full size (M rows: 1, a partial tile, 32 / 33 / 64) and of the tiny model
@pytest.mark.parametrize("M,N,K,splits", [
    (1, 1024, 1024, 4), (32, 1024, 1024, 4), (33, 1024, 1024, 4),            # output / query projections
    (16, 1024, 8192, 8), (64, 1024, 8192, 8), (5, 1024, 8192, 3),            # FFN output projection
    (3, 128, 128, 4), (40, 128, 256, 8), (2, 384, 128, 2), (7, 104, 64, 1),  # tiny model / ragged N
    (16, 1024, 1024, 1), (16, 1024, 1024, 2),
])
def test_dstep_split_k_residual_layernorm(lib, report_dir, M, N, K, splits):
    """x += in.W^T + b via K-range partials of the packed-weight product, then LayerNorm; deterministic."""
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
        h = torch.full((M, N), float("nan"), device="cuda")
        check(lib, lib.sc_op_dstep_res_ln(P(dev(inp)), P(dev(w)), P(dev(b)), P(x), P(dev(gam)), P(dev(bet)), P(h), M, N, K, splits))
        outs.append((x.cpu(), h.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ex, eh = rel_err(outs[0][0], xr), float((outs[0][1].double() - hr).abs().max())
    _log(report_dir, "dstep_res_ln", M=M, N=N, K=K, splits=splits, err_x=ex, err_h=eh)
    assert ex < 2e-6 and eh < 2e-5


@pytest.mark.parametrize("M,N,K", [(1, 3072, 1024), (64, 3072, 1024), (20, 3072, 1024)])
def test_dstep_wide_projection_partials(lib, report_dir, M, N, K):
    """N = 3 x 1024 (q | k | v): checked in three 1024-wide column blocks through the reduce kernel (bias / LN off)."""
    g = torch.Generator().manual_seed(M + N + K)
    inp = torch.randn(M, K, generator=g)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half()
    ref = inp.double() @ w.double().t()
    one, zero = torch.ones(1024), torch.zeros(1024)
    for blk in range(3):
        x = dev(torch.zeros(M, 1024))
        h = torch.empty(M, 1024, device="cuda")
        wb = w[1024 * blk: 1024 * (blk + 1)].contiguous()
        check(lib, lib.sc_op_dstep_res_ln(P(dev(inp)), P(dev(wb)), P(None), P(x), P(dev(one)), P(dev(zero)), P(h), M, 1024, K, 2))
        err = rel_err(x.cpu(), ref[:, 1024 * blk: 1024 * (blk + 1)])
        _log(report_dir, "dstep_qkv_block", M=M, blk=blk, err=err)
        assert err < 2e-6


@pytest.mark.parametrize("M,N,K,act", [(1, 8192, 1024, 1), (16, 8192, 1024, 1), (64, 8192, 1024, 1), (33, 256, 128, 1), (5, 1024, 256, 0),
                                       (9, 8192, 1024, 0)])
def test_dstep_linear_planes(lib, report_dir, M, N, K, act):
    """FFN inner projection: act(x.W^T + b) leaving the kernel as split fp16 planes (hi + lo re-summed here)."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x = torch.randn(M, K, generator=g) * 2.0
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half()
    b = torch.randn(N, generator=g) * 0.1
    ref = x.double() @ w.double().t() + b.double()
    if act:
        ref = torch.relu(ref)
    y = torch.full((M, N), float("nan"), device="cuda")
    check(lib, lib.sc_op_dstep_linear_planes(P(dev(x)), P(dev(w)), P(dev(b)), P(y), M, N, K, act))
    err = rel_err(y.cpu(), ref)
    _log(report_dir, "dstep_linear_planes", M=M, N=N, K=K, act=act, err=err)
    assert err < 2e-6, err  # the hi + lo planes hold the value to 2^-22 relative


@pytest.mark.parametrize("M,N,K,ntl", [(16, 256102, 1024, 4), (1, 256102, 1024, 4), (64, 256102, 1024, 4), (5, 1200, 128, 4), (40, 10082, 1024, 1),
                                       (33, 1200, 128, 3)])
@pytest.mark.parametrize("mode", ["plain", "no_eos", "force_eos", "unk_pen"])
def test_dstep_fused_argmax(lib, report_dir, M, N, K, ntl, mode):
    """Vocabulary projection with the arg-max / log-softmax folded into the epilogue vs torch."""
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
    check(lib, lib.sc_op_dstep_argmax(P(dev(x)), P(dev(w)), M, N, K, step, min_eos, force, pad, eos, unk, pen, ntl, P(idx), P(lp)))
    assert idx.cpu().tolist() == ref_idx.tolist()
    err = float((lp.cpu().double() - ref_lp).abs().max()) if mode != "unk_pen" else 0.0
    _log(report_dir, "dstep_argmax", M=M, N=N, K=K, mode=mode, err=err)
    assert err < 1e-4


@pytest.mark.parametrize("nb,heads,cap,pos,S", [(1, 16, 42, 0, 2), (1, 16, 42, 41, 2), (5, 16, 42, 17, 2), (64, 16, 42, 40, 2), (33, 2, 20, 7, 1),
                                                (2, 16, 200, 150, 3), (3, 4, 64, 63, 2), (3, 4, 70, 64, 2)])
def test_dstep_self_attention(lib, report_dir, nb, heads, cap, pos, S):
    """Appends the new key / value row at `pos` and attends over keys 0..pos; q/k/v arrive as S partial sums + bias."""
    g = torch.Generator().manual_seed(nb * 7 + heads + cap + pos)
    M = heads * 64
    proj = torch.randn(S, nb, 3 * M, generator=g)
    bias = torch.randn(3 * M, generator=g) * 0.1
    kc = torch.randn(nb, cap, M, generator=g)
    vc = torch.randn(nb, cap, M, generator=g)
    qkv = proj.double().sum(0) + bias.double()
    q, kn, vn = qkv[:, :M], qkv[:, M: 2 * M], qkv[:, 2 * M:]
    kr, vr = kc.double().clone(), vc.double().clone()
    kr[:, pos], vr[:, pos] = kn, vn

    def hd(t):  # (nb, L, M) -> (nb, H, L, 64)
        return t.view(nb, -1, heads, 64).transpose(1, 2)

    w = (hd(q[:, None]) @ hd(kr[:, : pos + 1]).transpose(-1, -2)) * 0.125
    ref = (torch.softmax(w, -1) @ hd(vr[:, : pos + 1])).transpose(1, 2).reshape(nb, M)
    # rows from `pos` on are uninitialised memory in the product (the step appends row `pos` itself): NaN there must
    # not reach the result (regression: the clamped re-read of row `pos` once entered the value sum as 0 * NaN)
    kc_in, vc_in = kc.clone(), vc.clone()
    kc_in[:, pos:] = float("nan")
    vc_in[:, pos:] = float("nan")
    d_k, d_v = dev(kc_in), dev(vc_in)
    out = torch.full((nb, M), float("nan"), device="cuda")
    check(lib, lib.sc_op_dstep_attention(P(dev(proj)), S, P(dev(bias)), P(d_k), P(d_v), cap, pos, P(None), 0, nb, heads, P(out)))
    assert not torch.isnan(out).any()
    err = float((out.cpu().double() - ref).abs().max())
    # the cache rows: `pos` holds the new row (fp32 of the partial sums), every other row is untouched
    ek = float((d_k.cpu().double()[:, : pos + 1] - kr[:, : pos + 1]).abs().max())
    ev = float((d_v.cpu().double()[:, : pos + 1] - vr[:, : pos + 1]).abs().max())
    _log(report_dir, "dstep_self_attention", nb=nb, heads=heads, cap=cap, pos=pos, err=err, ek=ek, ev=ev)
    assert err < 2e-5 and ek < 1e-5 and ev < 1e-5


@pytest.mark.parametrize("nb,heads,s_enc,lens,S", [(1, 16, 63, [63], 4), (4, 16, 63, [63, 1, 40, 62], 4), (64, 16, 63, None, 4), (2, 2, 13, [13, 5], 1),
                                                    (2, 4, 200, [200, 131], 2), (3, 16, 64, [64, 64, 33], 4), (2, 16, 65, [65, 2], 4)])
def test_dstep_cross_attention(lib, report_dir, nb, heads, s_enc, lens, S):
    g = torch.Generator().manual_seed(nb * 5 + heads + s_enc)
    M = heads * 64
    if lens is None:
        lens = [1 + (i * 7) % s_enc for i in range(nb)]
    proj = torch.randn(S, nb, M, generator=g)
    bias = torch.randn(M, generator=g) * 0.1
    kv = torch.randn(nb, s_enc, 2 * M, generator=g)
    q = proj.double().sum(0) + bias.double()
    k, v = kv.double()[..., :M], kv.double()[..., M:]

    def hd(t):
        return t.reshape(nb, -1, heads, 64).transpose(1, 2)

    w = (hd(q[:, None]) @ hd(k).transpose(-1, -2)) * 0.125
    mask = torch.arange(s_enc)[None, :] < torch.tensor(lens)[:, None]
    w = w.masked_fill(~mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(w, -1) @ hd(v)).transpose(1, 2).reshape(nb, M)
    kv_in = kv.clone()
    for b, l in enumerate(lens):  # keys behind a row's length are masked: whatever they hold must not matter
        kv_in[b, l:] = float("nan")
    out = torch.full((nb, M), float("nan"), device="cuda")
    check(lib, lib.sc_op_dstep_attention(P(dev(proj)), S, P(dev(bias)), P(dev(kv_in)), P(None), s_enc, 0,
                                         P(dev(torch.tensor(lens, dtype=torch.int32))), 1, nb, heads, P(out)))
    err = float((out.cpu().double() - ref).abs().max())
    _log(report_dir, "dstep_cross_attention", nb=nb, heads=heads, s_enc=s_enc, err=err)
    assert err < 2e-5
