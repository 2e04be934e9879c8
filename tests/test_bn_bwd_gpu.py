"""BatchNorm backward with the column sums out of the PRODUCING product (ns_gemm stat_z -> ns_bn_bwd sum_dy / sum_dyxh,
modules.py:198 backward) against a float64 restatement, on every kernel family that forms a dy in the training step;
the stand-alone reduction fallback; bitwise repeatability (no float atomics); the vector form of ns_cast2d."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2


def _bn_bwd_ref(dy, z, mean, istd, gamma, act, valid, count):
    dy = dy * valid[:, None]
    xh = (z - mean) * istd
    s1 = (dy).sum(0)
    s2 = (dy * xh * valid[:, None]).sum(0)
    dz = gamma * istd * (dy - s1 / count - xh * s2 / count) * valid[:, None]
    if act == ACT_RELU:
        dz = dz * (z > 0)
    elif act == ACT_TANH:
        dz = dz * (1 - z * z)
    return dz, s1, s2


# (operand dtype, f32 passes): bf16 -> 128- or 256-tile bf16 kernels; fp32 + 3 passes -> split-bf16 kernel; fp32 + 0 -> exact
@pytest.mark.parametrize("dtype,passes,M", [(torch.bfloat16, 0, 600), (torch.bfloat16, 0, 24 * 1028), (torch.float32, 3, 600),
                                            (torch.float32, 0, 300)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_bn_backward_sums_out_of_the_producing_product(dev, dtype, passes, M, accumulate):
    from nspeech_amd import ops
    g = torch.Generator().manual_seed(M + passes + accumulate)
    C, K = 256, 128
    period, lo, hi = (1028, 2, 1026) if M > 1000 else (30, 2, 27)
    A = (torch.randn(M, K, generator=g) * 0.3).to(dtype).to(dev)
    B = (torch.randn(C, K, generator=g) * 0.3).to(dtype).to(dev)         # b_mode 0: [N, K]
    zdt = torch.bfloat16 if dtype == torch.bfloat16 else torch.float32
    z = torch.tanh(torch.randn(M, C, generator=g)).to(zdt).to(dev)
    mean = (torch.randn(C, generator=g) * 0.1).to(dev)
    istd = (1.0 + 0.2 * torch.rand(C, generator=g)).to(dev)
    gamma = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    base = (torch.randn(M, C, generator=g) * 0.1).to(dev)
    dy = base.clone() if accumulate else torch.full((M, C), float("nan"), device=dev)
    sums = torch.zeros(2 * C, device=dev)
    # as the conv data gradient: output row 0 lands on buffer row 2 (c_off), the product covers rows 2 .. M-1 and is
    # masked in the frame of the buffer rows (row_shift 2); rows 0, 1 are pad rows
    if accumulate:
        dy[:2] = 0
        base[:2] = 0
    else:
        dy[:2] = 0
    ops.F32_PASSES = passes
    ops.gemm(A, B, dy, M - 2, C, K, K, K, C, a_mode=0, b_mode=0, a_off=2 * K, c_off=2 * C, accumulate=accumulate,
             row_mask=(period, lo, hi, 2), col_sum=sums, col_sumsq=sums[C:], stat_z=z, ld_stat_z=C, stat_z_off=2 * C,
             stat_mean=mean, stat_istd=istd)
    ops.F32_PASSES = 0
    m = torch.arange(M)
    valid = (((m % period) >= lo) & ((m % period) < hi)).double()
    count = float(valid.sum())
    dy64 = dy.double().cpu()
    assert torch.equal(dy64 * (1 - valid[:, None]), torch.zeros_like(dy64)) or accumulate   # masked rows of a plain store are 0
    want_dy = (A.double().cpu() @ B.double().cpu().t()) * valid[:, None] + (base.double().cpu() if accumulate else 0)
    assert (dy64 - want_dy).abs().max().item() < (2e-2 if dtype == torch.bfloat16 or passes else 1e-4)
    dz, s1, s2 = _bn_bwd_ref(dy64, z.double().cpu(), mean.double().cpu(), istd.double().cpu(), gamma.double().cpu(), ACT_TANH,
                             valid, count)
    sc = max(1.0, s1.abs().max().item(), s2.abs().max().item())
    assert (sums[:C].double().cpu() - s1).abs().max().item() < 2e-5 * sc * np.sqrt(M), "sum dy"
    assert (sums[C:].double().cpu() - s2).abs().max().item() < 2e-5 * sc * np.sqrt(M), "sum dy*xhat"
    # the apply pass on those sums vs the stand-alone reduction vs float64
    outs = []
    for use in (True, False):
        dpre = torch.full((M, C), float("nan"), dtype=zdt, device=dev)
        gr = torch.zeros(3 * C, device=dev)
        work = torch.zeros(200 * 1024, device=dev)
        ops.bn_bwd(dy, z, dpre, M, C, mean, istd, gamma, gr, gr, gr, work, count, ACT_TANH, row_mask=(period, lo, hi),
                   dgamma_off=0, dbeta_off=C, dbias_off=2 * C, sums=(sums[:C], sums[C:]) if use else None)
        torch.cuda.synchronize()
        outs.append((dpre.clone(), gr.clone()))
        tol = 2e-2 if zdt == torch.bfloat16 else 1e-5
        assert (dpre.double().cpu() - dz).abs().max().item() < tol * max(1.0, dz.abs().max().item()), use
        assert (gr[:C].double().cpu() - s2).abs().max().item() < 2e-5 * sc * np.sqrt(M)
        assert (gr[C:2 * C].double().cpu() - s1).abs().max().item() < 2e-5 * sc * np.sqrt(M)
        want_db = dpre.double().cpu().sum(0)
        assert (gr[2 * C:].double().cpu() - want_db).abs().max().item() < 1e-4 * max(1.0, want_db.abs().max().item()) * np.sqrt(M / 300)
    # repeatability: a second call gives the same bits (fixed-order partial sums everywhere)
    dpre2 = torch.zeros((M, C), dtype=zdt, device=dev)
    gr2 = torch.zeros(3 * C, device=dev)
    ops.bn_bwd(dy, z, dpre2, M, C, mean, istd, gamma, gr2, gr2, gr2, work, count, ACT_TANH, row_mask=(period, lo, hi),
               dgamma_off=0, dbeta_off=C, dbias_off=2 * C, sums=None)
    assert torch.equal(dpre2, outs[1][0]) and torch.equal(gr2, outs[1][1])


@pytest.mark.parametrize("rows,cols", [(256, 1024), (1024, 4096), (2560, 512), (68, 132), (64, 64), (7, 20)])
@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("dst", ["f32", "bf16", "pair", "bf16+pair"])
def test_cast2d_forms(dev, rows, cols, transpose, dst):
    from nspeech_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    src = torch.randn(rows + 3, cols, generator=g).to(dev)
    ld_dst = (rows if transpose else cols)
    n = (cols if transpose else rows) * ld_dst
    d = None if dst == "pair" else torch.full((n,), float("nan"), dtype=torch.float32 if dst == "f32" else torch.bfloat16, device=dev)
    hi = lo = None
    if "pair" in dst:
        hi = torch.full((n,), float("nan"), dtype=torch.bfloat16, device=dev)
        lo = torch.full((n,), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.cast2d(src, rows, cols, cols, d, ld_dst, transpose, src_off=cols, dst_hi=hi, dst_lo=lo)
    want = src[1:1 + rows]
    want = (want.t() if transpose else want).contiguous().reshape(-1)
    if d is not None:
        assert torch.equal(d, want.to(d.dtype))
    if hi is not None:
        h = want.to(torch.bfloat16)
        assert torch.equal(hi, h) and torch.equal(lo, (want - h.float()).to(torch.bfloat16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [128, 6])          # the vector kernels (C % 4 == 0) and the scalar ones
def test_batchnorm_on_a_column_block_of_a_wider_activation(dev, dtype, C):
    """ns_bn_fwd_params.ld_y / ns_bn_bwd_params.ld_dy (round 5): the output written as a column block of a [rows, LD]
    array, the backward pass reading dy from the same block of the wide gradient - bit for bit what the contiguous
    call gives, and nothing outside the block touched (the CBHG convolution bank without its concatenation copies)."""
    from nspeech_amd import ops
    g = torch.Generator().manual_seed(C)
    rows, LD, col = 75, (3 * C + 8) // 4 * 4, C + 4
    period, lo, hi = 25, 2, 22
    z = torch.randn(rows, C, generator=g).to(dtype).to(dev)
    gamma = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    zf = z.float()
    m = torch.arange(rows, device=dev)
    valid = ((m % period) >= lo) & ((m % period) < hi)
    count = int(valid.sum())
    st = torch.zeros(4 * C, device=dev)
    st[:C] = (zf * valid[:, None]).sum(0)
    st[C:2 * C] = (zf * zf * valid[:, None]).sum(0)
    y0 = torch.full((rows, C), 7.0, dtype=dtype, device=dev)
    wide = torch.full((rows, LD), 7.0, dtype=dtype, device=dev)
    for y, kw in ((y0, {}), (wide, dict(y_off=col, ld_y=LD))):
        ops.bn_fwd(z, y, rows, C, st, st[C:], count, gamma, beta, mm.clone(), mv.clone(), st[2 * C:], st[3 * C:], True,
                   row_mask=(period, lo, hi), **kw)
    assert torch.equal(wide[:, col:col + C], y0)
    out = wide.clone()
    out[:, col:col + C] = 7.0
    assert bool((out == 7.0).all())                                   # nothing outside the block was written
    dy = (torch.randn(rows, C, generator=g) * 0.3).to(dev)
    dwide = torch.full((rows, LD), float("nan"), device=dev)
    dwide[:, col:col + C] = dy
    res = []
    for d, kw in ((dy, {}), (dwide, dict(dy_off=col, ld_dy=LD))):
        dpre = torch.zeros(rows, C, dtype=dtype, device=dev)
        grads = torch.zeros(3 * C, device=dev)
        work = torch.zeros(200 * max(1024, C), device=dev)
        ops.bn_bwd(d, z, dpre, rows, C, st[2 * C:], st[3 * C:], gamma, grads, grads, grads, work, count, ACT_RELU,
                   row_mask=(period, lo, hi), dgamma_off=0, dbeta_off=C, dbias_off=2 * C, **kw)
        res.append((dpre.clone(), grads.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert float(res[0][1].abs().max()) > 0
