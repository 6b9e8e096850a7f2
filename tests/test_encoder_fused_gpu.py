"""Fused element-wise passes of the Conformer stack (csrc/k_norm.hip): GLU -> causal depthwise conv (31) -> LayerNorm ->
SiLU -> split planes in one kernel, and a layer's closing LayerNorm together with the next layer's first one.  Both must
be BIT-IDENTICAL to the separate launches they replace (the encoder's 2e-4 parity against the oracle is then unchanged) and
are also checked against a PyTorch fp64 restatement (conformer_shaw/builder.py:148-156)."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_ops_gpu import P, check, dev, lib, _release_device_copies  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def _log(report_dir, name, **kw):
    with open(report_dir / "ops_report.txt", "a") as f:
        f.write(name + " " + " ".join(f"{k}={v}" for k, v in kw.items()) + "\n")


@pytest.mark.parametrize("nb,T,C_,lens", [(2, 499, 1024, None), (3, 499, 1024, [499, 154, 319]), (32, 499, 1024, None), (1, 37, 1024, None),
                                          (5, 70, 128, [70, 3, 31, 64, 9]), (2, 8, 128, None), (64, 499, 1024, None), (300, 40, 128, None)])
def test_glu_dwconv_ln_fused_equals_separate_launches(lib, report_dir, nb, T, C_, lens):
    g = torch.Generator().manual_seed(T + C_ + nb)
    x = torch.randn(nb, T, 2 * C_, generator=g)
    w = torch.randn(C_, 31, generator=g) * 0.2
    gam = torch.rand(C_, generator=g) + 0.5
    bet = torch.randn(C_, generator=g) * 0.1
    d_lens = dev(torch.tensor(lens, dtype=torch.int32)) if lens is not None else None
    outs = []
    for fused in (1, 0):
        yh = torch.full((nb, T, C_), float("nan"), dtype=torch.float16, device="cuda")
        yl = torch.full((nb, T, C_), float("nan"), dtype=torch.float16, device="cuda")
        check(lib, lib.sc_op_glu_dwconv_ln(P(dev(x)), P(dev(w)), P(dev(gam)), P(dev(bet)), 2, P(yh), P(yl), nb, T, C_, 31, P(d_lens), fused))
        outs.append((yh.cpu(), yl.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "fused kernel differs from the separate launches"
    if nb * T * C_ <= 4 * 499 * 1024:  # fp64 restatement on the smaller cases
        gl = F.glu(x.double(), dim=-1)
        if lens is not None:
            for i, l in enumerate(lens):
                gl[i, l:] = 0
        conv = F.conv1d(F.pad(gl.transpose(1, 2), (30, 0)), w.double().unsqueeze(1), groups=C_).transpose(1, 2)
        ref = F.silu(F.layer_norm(conv, (C_,), gam.double(), bet.double(), 1e-5))
        got = outs[0][0].double() + outs[0][1].double()
        err = float((got - ref).abs().max())
        _log(report_dir, "glu_dwconv_ln_fused", nb=nb, T=T, C=C_, err=err)
        assert err < 5e-5, err


@pytest.mark.parametrize("rows,C_", [(15968, 1024), (7, 1024), (33, 128), (1, 64)])
def test_layernorm2_fused_equals_two_launches(lib, report_dir, rows, C_):
    g = torch.Generator().manual_seed(rows + C_)
    x = torch.randn(rows, C_, generator=g) * 2 + 0.5
    ga, gb = torch.rand(C_, generator=g) + 0.5, torch.rand(C_, generator=g) + 0.5
    ba, bb = torch.randn(C_, generator=g) * 0.1, torch.randn(C_, generator=g) * 0.1
    outs = []
    for fused in (1, 0):
        y = torch.full((rows, C_), float("nan"), device="cuda")
        yh = torch.full((rows, C_), float("nan"), dtype=torch.float16, device="cuda")
        yl = torch.full((rows, C_), float("nan"), dtype=torch.float16, device="cuda")
        check(lib, lib.sc_op_layernorm2(P(dev(x)), P(dev(ga)), P(dev(ba)), P(dev(gb)), P(dev(bb)), P(y), P(yh), P(yl), rows, C_, fused))
        outs.append((y.cpu(), yh.cpu(), yl.cpu()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    y_ref = F.layer_norm(x.double(), (C_,), ga.double(), ba.double(), 1e-5)
    z_ref = F.layer_norm(y_ref, (C_,), gb.double(), bb.double(), 1e-5)
    err_y = float((outs[0][0].double() - y_ref).abs().max())
    err_z = float((outs[0][1].double() + outs[0][2].double() - z_ref).abs().max())
    _log(report_dir, "layernorm2_fused", rows=rows, C=C_, err_y=err_y, err_z=err_z)
    assert err_y < 2e-5 and err_z < 2e-5


@pytest.mark.parametrize("nb,T,C_,k,dil", [(2, 2500, 256, 3, 1), (2, 2500, 256, 11, 5), (1, 10000, 128, 7, 3), (3, 333, 128, 11, 1), (1, 40, 256, 7, 5),
                                           (32, 2500, 256, 7, 1)])
def test_resblock_pair_on_dma_gemm(lib, report_dir, nb, T, C_, k, dil):
    """One HiFi-GAN dilation pair the way the wide vocoder stages (C >= 128) run it since round 3 - LeakyReLU'd split planes,
    both convolutions on the DMA-fed GEMM in implicit-convolution mode (hifigan.py:130-196) - against the two
    register-staged implicit-GEMM launches it replaces and against F.conv1d in float64."""
    import math

    import numpy as np  # noqa: F401

    from tests.test_ops_gpu import rel_err

    g = torch.Generator().manual_seed(T * 13 + C_ * 5 + k + dil)
    x = torch.randn(nb, T, C_, generator=g)
    w1 = (torch.randn(C_, C_, k, generator=g) / math.sqrt(C_ * k)).half()
    w2 = (torch.randn(C_, C_, k, generator=g) / math.sqrt(C_ * k)).half()
    b1, b2 = torch.randn(C_, generator=g) * 0.1, torch.randn(C_, generator=g) * 0.1
    kpad = C_ * k
    wp1 = torch.zeros(C_, kpad, dtype=torch.float16, device="cuda")
    wp2 = torch.zeros(C_, kpad, dtype=torch.float16, device="cuda")
    check(lib, lib.sc_op_pack_conv_weight(P(dev(w1)), P(wp1), C_, C_, k))
    check(lib, lib.sc_op_pack_conv_weight(P(dev(w2)), P(wp2), C_, C_, k))
    dx, db1, db2 = dev(x), dev(b1), dev(b2)
    tmp = torch.full((nb, T, C_), float("nan"), device="cuda")
    two = torch.full((nb, T, C_), float("nan"), device="cuda")
    check(lib, lib.sc_op_conv1d(P(dx), P(wp1), P(db1), None, P(tmp), nb, T, C_, C_, k, 1, dil * (k - 1) // 2, dil, None, 1, 0))
    check(lib, lib.sc_op_conv1d(P(tmp), P(wp2), P(db2), P(dx), P(two), nb, T, C_, C_, k, 1, (k - 1) // 2, 1, None, 1, 0))
    got = torch.full((nb, T, C_), float("nan"), device="cuda")
    check(lib, lib.sc_op_resblock_pair_ps(P(dx), P(wp1), P(db1), P(wp2), P(db2), P(got), nb, T, C_, k, dil))
    got, want = got.cpu(), two.cpu()
    assert not torch.isnan(got).any()
    same = bool(torch.equal(got, want))
    err = None
    if nb * T * C_ <= 2 * 2500 * 256 * 2:
        xt = F.leaky_relu(x, 0.1).transpose(1, 2).double()
        h = F.conv1d(xt, w1.double(), b1.double(), padding=dil * (k - 1) // 2, dilation=dil)
        ref = F.conv1d(F.leaky_relu(h, 0.1), w2.double(), b2.double(), padding=(k - 1) // 2).transpose(1, 2) + x.double()
        err = rel_err(got, ref)
        assert err < 3e-6, err
    dmax = float((got - want).abs().max())
    _log(report_dir, "resblock_pair_ps", nb=nb, T=T, C=C_, k=k, dil=dil, err=err, bit_identical_to_two_convs=same, max_abs_diff=dmax)
    assert dmax < 2e-5
