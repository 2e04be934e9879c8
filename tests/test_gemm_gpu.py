"""GPU parity of ns_gemm (all operand modes, masks, stats, split-K) against a float64 matmul."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, dtype, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(*shape, generator=g, dtype=torch.float32)
    return x.to(dtype).to(dev)


def _ref(A, B, a_mode, b_mode):
    a = A.double().cpu()
    b = B.double().cpu()
    if a_mode == 1:
        a = a.t()
    if b_mode == 0:
        b = b.t()
    return a @ b


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("a_mode,b_mode", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 320), (32, 256, 512), (8, 64, 96), (257, 72, 1000)])
def test_gemm_modes(dev, dtype, a_mode, b_mode, M, N, K):
    from nspeech_amd import ops
    A = _mk((M, K) if a_mode == 0 else (K, M), dtype, dev, 1)
    B = _mk((N, K) if b_mode == 0 else (K, N), dtype, dev, 2)
    Cm = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    ops.gemm(A, B, Cm, M, N, K, A.shape[1], B.shape[1], N, a_mode=a_mode, b_mode=b_mode)
    torch.cuda.synchronize()
    ref = _ref(A, B, a_mode, b_mode)
    err = (Cm.double().cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-5 * scale + 1e-4, (err, scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogue_mask_stats(dev, dtype):
    from nspeech_amd import ops
    M, N, K = 300, 192, 160
    A = _mk((M, K), dtype, dev, 3)
    B = _mk((K, N), dtype, dev, 4)
    bias = _mk((N,), torch.float32, dev, 5)
    out = torch.full((M + 2, N), 7.0, dtype=dtype, device=dev)
    s1 = torch.zeros(N, device=dev)
    s2 = torch.zeros(N, device=dev)
    period, lo, hi, shift = 30, 2, 27, 2
    ops.gemm(A, B, out, M, N, K, K, N, N, b_mode=1, c_off=2 * N, bias=bias, act=1,
             row_mask=(period, lo, hi, shift), col_sum=s1, col_sumsq=s2)
    torch.cuda.synchronize()
    ref = torch.relu(A.double().cpu() @ B.double().cpu() + bias.double().cpu())
    m = torch.arange(M)
    valid = (((m + shift) % period) >= lo) & (((m + shift) % period) < hi)
    ref = ref * valid[:, None]
    got = out[2:].double().cpu()
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    assert (out[:2].float().cpu() == 7.0).all()
    # stats are over the stored values
    assert torch.allclose(s1.double().cpu(), got.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(s2.double().cpu(), (got * got).sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_splitk_atomic_and_segments(dev, dtype):
    from nspeech_amd import ops
    # wgrad-like: contraction over a long dim with both operands k-slow, split-K atomics
    Kc, M, N = 2000, 160, 136
    A = _mk((Kc, M), dtype, dev, 6)
    B = _mk((Kc, N), dtype, dev, 7)
    Cm = torch.ones((M, N), dtype=torch.float32, device=dev)
    ops.gemm(A, B, Cm, M, N, Kc, M, N, N, a_mode=1, b_mode=1, accumulate=2, split_k=4)
    torch.cuda.synchronize()
    ref = 1.0 + A.double().cpu().t() @ B.double().cpu()
    assert (Cm.double().cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-3
    # segmented B (conv data-gradient pattern): 3 taps walked backwards
    taps, Cin, Cout, Mr = 3, 64, 128, 100
    W = _mk((taps, Cin, Cout), dtype, dev, 8)
    dY = _mk((Mr + taps - 1, Cout), dtype, dev, 9)
    dX = torch.zeros((Mr, Cin), dtype=torch.float32, device=dev)
    ops.gemm(dY, W, dX, Mr, Cin, taps * Cout, Cout, Cout, Cin, a_mode=0, b_mode=0,
             b_off=(taps - 1) * Cin * Cout, b_seg=(Cout, -Cin * Cout))
    torch.cuda.synchronize()
    dy = dY.double().cpu()
    w = W.double().cpu()
    ref = torch.zeros(Mr, Cin, dtype=torch.float64)
    for j in range(taps):
        ref += dy[j:j + Mr] @ w[taps - 1 - j].t()
    assert (dX.double().cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-3


def test_gemm_splitk_many_workgroups_takes_the_32_deep_tiles(dev):
    """A k-slow x k-slow bf16 product with more workgroups than the 64-deep form keeps resident (>= 600) runs on the
    BK = 32 instantiation of gemm_mfma_kernel (four workgroups per CU); odd K, ragged M and N, split-K atomics."""
    from nspeech_amd import ops, profiling
    Kc, M, N = 4099, 520, 1288                     # 5 x 11 tiles x 12 splits = 660 workgroups
    A = _mk((Kc, M), torch.bfloat16, dev, 16)
    B = _mk((Kc, N), torch.bfloat16, dev, 17)
    Cm = torch.full((M, N), 0.5, dtype=torch.float32, device=dev)
    ops.gemm(A, B, Cm, M, N, Kc, M, N, N, a_mode=1, b_mode=1, accumulate=2, split_k=12)
    torch.cuda.synchronize()
    assert profiling._last_kernel().endswith("32, 128>"), profiling._last_kernel()
    ref = 0.5 + A.double().cpu().t() @ B.double().cpu()
    assert (Cm.double().cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-3
    # and without split-K (plain stores through the vector epilogue), still >= 600 workgroups
    M2, N2, K2 = 3208, 3080, 96                    # 26 x 25 tiles
    A2 = _mk((K2, M2), torch.bfloat16, dev, 18)
    B2 = _mk((K2, N2), torch.bfloat16, dev, 19)
    C2 = torch.zeros((M2, N2), dtype=torch.float32, device=dev)
    ops.gemm(A2, B2, C2, M2, N2, K2, M2, N2, N2, a_mode=1, b_mode=1)
    torch.cuda.synchronize()
    assert profiling._last_kernel().endswith("32, 128>"), profiling._last_kernel()
    ref2 = A2.double().cpu().t() @ B2.double().cpu()
    assert (C2.double().cpu() - ref2).abs().max().item() <= 2e-5 * ref2.abs().max().item() + 1e-3


@pytest.mark.parametrize("passes,tol", [(3, 3e-5), (1, 2e-2)])
@pytest.mark.parametrize("a_mode,b_mode", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 320), (32, 256, 512), (260, 72, 1000)])
def test_gemm_fp32_split_bf16(dev, passes, tol, a_mode, b_mode, M, N, K):
    """fp32 operands on the bf16 MFMA: x = hi + lo, three products (or hi*hi only)."""
    from nspeech_amd import ops
    A = _mk((M, K) if a_mode == 0 else (K, M), torch.float32, dev, 11)
    B = _mk((N, K) if b_mode == 0 else (K, N), torch.float32, dev, 12)
    Cm = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    ops.gemm(A, B, Cm, M, N, K, A.shape[1], B.shape[1], N, a_mode=a_mode, b_mode=b_mode, f32_passes=passes)
    torch.cuda.synchronize()
    ref = _ref(A, B, a_mode, b_mode)
    # error model: sum of K products each off by ~2^-17 (3 passes) or ~2^-8 (1 pass) relative
    bound = tol * (A.double().abs().cpu().max() * B.double().abs().cpu().max() * K ** 0.5).item()
    err = (Cm.double().cpu() - ref).abs().max().item()
    assert err <= bound, (err, bound)
    if passes == 3:   # and it is clearly better than single-pass bf16
        assert err <= 1e-4 * ref.abs().max().item()


def test_gemm_256_tile_kernel(dev):
    """Large k-contiguous bf16 products take the 256 x 256 eight-phase kernel (global_load_lds staging, counted
    vmcnt, staggered wave groups): ragged M, strided (im2col) A, bias + ReLU + row mask + BatchNorm sums + bf16
    output, the segmented-B data-gradient walk, and the three-segment product of pre-split fp32 values."""
    import os
    from nspeech_amd import ops
    from nspeech_amd import _lib as L
    bf = torch.bfloat16
    # im2col view: rows overlap (lda = Cin < K = taps * Cin)
    taps, Cin, Cout, rows = 5, 128, 256, 12300
    X = _mk((rows + taps - 1, Cin), bf, dev, 21)
    Wt = _mk((Cout, taps * Cin), bf, dev, 22)                 # k-contiguous weights [N, K]
    bias = _mk((Cout,), torch.float32, dev, 23)
    out = torch.full((rows, Cout), 7.0, dtype=bf, device=dev)
    st = torch.zeros(2 * Cout, device=dev)
    ops.gemm(X, Wt, out, rows, Cout, taps * Cin, Cin, taps * Cin, Cout, a_mode=0, b_mode=0, bias=bias, act=1,
             row_mask=(100, 2, 98, 0), col_sum=st, col_sumsq=st[Cout:])
    torch.cuda.synchronize()
    a = torch.as_strided(X.double().cpu(), (rows, taps * Cin), (Cin, 1))
    ref = torch.relu(a @ Wt.double().cpu().t() + bias.double().cpu())
    t = torch.arange(rows) % 100
    ref[(t < 2) | (t >= 98)] = 0
    got = out.double().cpu()
    assert (got - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()
    assert torch.allclose(st[:Cout].double().cpu(), got.sum(0), rtol=1e-4, atol=1e-1)
    assert torch.allclose(st[Cout:].double().cpu(), (got * got).sum(0), rtol=1e-4, atol=1e-1)
    # the same call through the 128-tile kernel gives the same values up to summation order
    os.environ["NS_GEMM_NO256"] = "1"
    try:
        out2 = torch.zeros_like(out)
        ops.gemm(X, Wt, out2, rows, Cout, taps * Cin, Cin, taps * Cin, Cout, a_mode=0, b_mode=0, bias=bias, act=1,
                 row_mask=(100, 2, 98, 0))
        torch.cuda.synchronize()
    finally:
        del os.environ["NS_GEMM_NO256"]
    assert (out2.float() - out.float()).abs().max().item() <= 2e-2 * ref.abs().max().item()
    # data gradient: taps walked backwards through segments of B
    W = _mk((taps, Cin, Cout), bf, dev, 24)
    dY = _mk((rows + taps - 1, Cout), bf, dev, 25)
    dX = torch.zeros((rows, Cin), dtype=torch.float32, device=dev)
    ops.gemm(dY, W, dX, rows, Cin, taps * Cout, Cout, Cout, Cin, a_mode=0, b_mode=0,
             b_off=(taps - 1) * Cin * Cout, b_seg=(Cout, -Cin * Cout))
    torch.cuda.synchronize()
    dy, w = dY.double().cpu(), W.double().cpu()
    ref = torch.zeros(rows, Cin, dtype=torch.float64)
    for j in range(taps):
        ref += dy[j:j + rows] @ w[taps - 1 - j].t()
    assert (dX.double().cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-3
    # odd number of K tiles, one K tile short of the pipeline depth, N not a multiple of 256
    for M, N, K in ((25000, 384, 192), (33000, 256, 128), (1100, 6144, 320)):
        A = _mk((M, K), bf, dev, M)
        B = _mk((N, K), bf, dev, N)
        Cm = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
        ops.gemm(A, B, Cm, M, N, K, K, K, N, a_mode=0, b_mode=0)
        torch.cuda.synchronize()
        ref = A.double().cpu() @ B.double().cpu().t()
        assert (Cm.double().cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-4, (M, N, K)
    # three segments: fp32 values pre-split into bf16 hi + lo
    M, N, K = 13000, 512, 256
    a32, b32 = _mk((M, K), torch.float32, dev, 31), _mk((N, K), torch.float32, dev, 32)
    ah, bh = a32.to(bf), b32.to(bf)
    al, bl = (a32 - ah.float()).to(bf), (b32 - bh.float()).to(bf)
    Cm = torch.zeros((M, N), dtype=torch.float32, device=dev)
    ops.gemm(ah, bh, Cm, M, N, K, K, K, N, a_mode=0, b_mode=0, a_lo=al, b_lo=bl)
    torch.cuda.synchronize()
    ref = a32.double().cpu() @ b32.double().cpu().t()
    assert (Cm.double().cpu() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()
    # pre-split operands on a shape the 256-tile kernel does not take: refused, not silently dropped
    with pytest.raises(L.NSError):
        ops.gemm(ah, bh, Cm, 64, N, K, K, K, N, a_mode=0, b_mode=0, a_lo=al, b_lo=bl)


@pytest.mark.parametrize("dtype,passes,tol", [(torch.float32, 0, 2e-5), (torch.float32, 3, 3e-5), (torch.bfloat16, 0, 2e-5)])
@pytest.mark.parametrize("a_mode,b_mode", [(0, 1), (1, 1), (0, 0)])
def test_gemm_batched(dev, dtype, passes, tol, a_mode, b_mode):
    """batch > 1: independent products in one launch (the per-utterance attention products), with padded strides,
    accumulate = 1 into distinct outputs, and refusal of per-call extras."""
    from nspeech_amd import ops
    from nspeech_amd import _lib as L
    nb, M, N, K = 5, 41, 72, 88
    sa, sb, sc = (M + 3) * K + 8, (N + 1) * K + 16, M * N + 24
    A = _mk((nb * sa,), dtype, dev, 41)
    B = _mk((nb * sb,), dtype, dev, 42)
    Cm = torch.ones(nb * sc, dtype=torch.float32, device=dev)
    lda = K if a_mode == 0 else M
    ldb = K if b_mode == 0 else N
    ops.gemm(A, B, Cm, M, N, K, lda, ldb, N, a_mode=a_mode, b_mode=b_mode, accumulate=1, batch=nb,
             batch_strides=(sa, sb, sc), f32_passes=passes)
    torch.cuda.synchronize()
    for z in range(nb):
        a = A[z * sa:z * sa + M * K].double().cpu().view((M, K) if a_mode == 0 else (K, M))
        b = B[z * sb:z * sb + N * K].double().cpu().view((N, K) if b_mode == 0 else (K, N))
        ref = 1.0 + _ref(a, b, a_mode, b_mode)
        got = Cm[z * sc:z * sc + M * N].double().cpu().view(M, N)
        assert (got - ref).abs().max().item() <= tol * ref.abs().max().item() + 1e-4, z
        assert (Cm[z * sc + M * N:(z + 1) * sc] == 1.0).all()          # the padding between items is untouched
    with pytest.raises(L.NSError):
        ops.gemm(A, B, Cm, M, N, K, lda, ldb, N, a_mode=a_mode, b_mode=b_mode, batch=nb, batch_strides=(sa, sb, sc),
                 bias=Cm)


def test_gemm_256_tile_kernel_is_bit_identical_to_the_128_tile_kernel(dev):
    """Both kernels add the same 32-deep MFMA products in the same order, so their fp32 results must be bit-identical.
    Many shapes and repeats: a mis-ordered LDS-DMA wait in the eight-phase schedule would show up as rare wrong tiles
    that a tolerance-based check against a reference can miss."""
    import os
    from nspeech_amd import ops
    bf = torch.bfloat16
    rs = __import__("random").Random(5)
    shapes = [(rs.randrange(12300, 40000), 128 * rs.randrange(2, 9), 64 * rs.randrange(2, 24)) for _ in range(10)]
    shapes += [(rs.randrange(1024, 3000), 128 * rs.randrange(24, 64), 64 * rs.randrange(2, 12)) for _ in range(6)]
    for M, N, K in shapes:
        A = _mk((M, K), bf, dev, M)
        B = _mk((N, K), bf, dev, N + 1)
        c128 = torch.zeros((M, N), dtype=torch.float32, device=dev)
        os.environ["NS_GEMM_NO256"] = "1"
        try:
            ops.gemm(A, B, c128, M, N, K, K, K, N, a_mode=0, b_mode=0)
            torch.cuda.synchronize()
        finally:
            del os.environ["NS_GEMM_NO256"]
        for rep in range(3):
            c256 = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
            ops.gemm(A, B, c256, M, N, K, K, K, N, a_mode=0, b_mode=0)
            torch.cuda.synchronize()
            assert torch.equal(c256, c128), (M, N, K, rep, (c256 - c128).abs().max().item())
