"""ns_rows32 (csrc/rows32.hip): <= 32 rows against weights packed once as split-bf16 MFMA fragments - the step products of
batched free-running synthesis - against float64 on the same operands, with fp32 activation rows and with activations
kept in the packed (fragment) layout."""
import numpy as np
import pytest
import torch

from nspeech_amd import ops
from nspeech_amd._lib import ACT_NONE, ACT_RELU

pytestmark = pytest.mark.gpu


def _rand(rng, *shape):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda()


def _unpack_rows(rows, K, N):
    """Host-side reading of packed rows: [row tile][chunk][lane = g * 16 + r16][plane][8 bf16] -> hi + lo as float64 [N, K]."""
    nkc = (K + 31) // 32
    raw = rows.view(torch.bfloat16).float().cpu().numpy().astype(np.float64).reshape(2, nkc, 4, 16, 2, 8)
    val = raw[:, :, :, :, 0, :] + raw[:, :, :, :, 1, :]            # [tile, kc, g, r16, j]
    full = val.transpose(0, 3, 1, 2, 4).reshape(32, nkc * 32)      # [tile * 16 + r16, kc * 32 + g * 8 + j]
    return full[:N, :K]


@pytest.mark.parametrize("N,K,C", [(32, 1024, 256), (5, 256, 128), (17, 392, 72), (1, 2048, 16)])
@pytest.mark.parametrize("passes", [3, 1])
def test_dense_rows(dev, N, K, C, passes):
    rng = np.random.default_rng(N + K)
    lda, ldo = K + 8, C + 4
    a, w = _rand(rng, N, lda), _rand(rng, K, C) * 0.05
    bias, add = _rand(rng, C), _rand(rng, N, ldo)
    out = torch.full((N, ldo), 7.0, device="cuda")
    out2 = torch.full((N, ldo), 7.0, device="cuda")
    packed = ops.rows32_pack(w, K, C)
    tol = 1.5e-5 if passes == 3 else 5e-3
    ops.rows32(a, lda, packed, N, K, C, out, ldo, bias=bias, add=add, add_sn=ldo, act=ACT_RELU, out2=out2, out2_sn=ldo,
               f32_passes=passes)
    ref = np.maximum(a[:, :K].double().cpu().numpy() @ w.double().cpu().numpy() + bias.double().cpu().numpy()
                     + add[:, :C].double().cpu().numpy(), 0.0)
    got = out[:, :C].cpu().numpy()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print("dense N %d K %d C %d passes %d: rel max %.2e" % (N, K, C, passes, err))
    assert err < tol, err
    assert torch.equal(out[:, :C], out2[:, :C])
    assert (out[:, C:] == 7.0).all()                 # nothing written past the last column
    ops.rows32(a, lda, packed, N, K, C, out, ldo, act=ACT_NONE, f32_passes=passes)
    ref = a[:, :K].double().cpu().numpy() @ w.double().cpu().numpy()
    assert np.abs(out[:, :C].cpu().numpy() - ref).max() / np.abs(ref).max() < tol


@pytest.mark.parametrize("N,K,C,col0,wide", [(32, 1024, 256, 1024, 2048), (7, 256, 128, 0, 256), (19, 384, 80, 64, 456)])
def test_packed_activation_rows(dev, N, K, C, col0, wide):
    """The operand as a column range of packed rows (written by ns_rows32_pack_rows), the result into a column range of
    other packed rows AND as fp32: both destinations hold the same values (the packed one as hi + lo, to 2^-16), and the
    fp32 one equals what fp32 operand rows give bit for bit (the same fragments reach the matrix core either way)."""
    rng = np.random.default_rng(K + C)
    a, w, bias = _rand(rng, N, K), _rand(rng, K, C) * 0.05, _rand(rng, C)
    packed = ops.rows32_pack(w, K, C)
    rows = ops.rows32_rows(wide, "cuda")
    ops.rows32_pack_rows(a, K, N, K, rows, wide, col0)
    back = _unpack_rows(rows, wide, N)[:, col0:col0 + K]
    assert np.abs(back - a.double().cpu().numpy()).max() < 2e-5 * 5     # hi + lo keeps ~16 bits
    dst_w, dst_c = 300, 40
    dst = ops.rows32_rows(dst_w, "cuda")
    out = torch.zeros(N, C, device="cuda")
    ref32 = torch.zeros(N, C, device="cuda")
    ops.rows32(None, 0, packed, N, K, C, out, C, bias=bias, act=ACT_RELU, a_rows=(rows, wide, col0), rows_out=(dst, dst_w, dst_c))
    ops.rows32(a, K, packed, N, K, C, ref32, C, bias=bias, act=ACT_RELU)
    assert torch.equal(out, ref32)
    got = _unpack_rows(dst, dst_w, N)
    assert np.abs(got[:, dst_c:dst_c + C] - out.double().cpu().numpy()).max() < 1e-4
    assert np.abs(got[:, :dst_c]).max() == 0 and np.abs(got[:, dst_c + C:]).max() == 0     # only its columns are written


@pytest.mark.parametrize("N,K,H", [(32, 1792, 1024), (5, 1792, 1024), (16, 2048, 1024), (5, 392, 256), (3, 392, 256), (20, 128, 64)])
@pytest.mark.parametrize("zone", [0.0, 0.1])
def test_cell_rows(dev, N, K, H, zone):
    """LSTMBlockCell on [input | h_prev] rows (gates i, j, f, o; forget bias 1), optionally with the zoneout expectation."""
    rng = np.random.default_rng(N + K + H)
    lda = K + 4
    a, w = _rand(rng, N, lda), _rand(rng, K, 4 * H) * 0.02
    bias, cp = _rand(rng, 4 * H) * 0.1, _rand(rng, N, H)
    h = torch.zeros(N, H + 8, device="cuda")
    h2 = torch.zeros(N, H, device="cuda")
    c = torch.zeros(N, H, device="cuda")
    hprev = a[:, K - H:K].contiguous()
    packed = ops.rows32_pack(w, K, 4 * H, cell_units=H)
    rows = ops.rows32_rows(H + 32, "cuda")
    ops.rows32(a, lda, packed, N, K, 4 * H, h, H + 8, bias=bias, out2=h2, out2_sn=H, cell_units=H, c_prev=cp, c_sn=H, c_out=c,
               co_sn=H, zoneout=zone, h_prev=hprev, hp_sn=H, rows_out=(rows, H + 32, 32))
    z = a[:, :K].double().cpu().numpy() @ w.double().cpu().numpy() + bias.double().cpu().numpy()
    sg = lambda x: 1.0 / (1.0 + np.exp(-x))
    i, j, f, o = sg(z[:, :H]), np.tanh(z[:, H:2 * H]), sg(z[:, 2 * H:3 * H] + 1.0), sg(z[:, 3 * H:])
    cpd = cp.double().cpu().numpy()
    cr = f * cpd + i * j
    hr = o * np.tanh(cr)
    if zone > 0:
        cr = zone * cpd + (1 - zone) * cr
        hr = zone * hprev.double().cpu().numpy() + (1 - zone) * hr
    eh = np.abs(h[:, :H].cpu().numpy() - hr).max()
    ec = np.abs(c.cpu().numpy() - cr).max()
    print("cell N %d K %d H %d zoneout %.1f: |dh| %.2e |dc| %.2e" % (N, K, H, zone, eh, ec))
    assert eh < 2e-5 and ec < 4e-5
    assert torch.equal(h[:, :H], h2)
    assert np.abs(_unpack_rows(rows, H + 32, N)[:, 32:] - h2.double().cpu().numpy()).max() < 1e-4
    # the same operand as packed rows: the same bits
    arows = ops.rows32_rows(K, "cuda")
    ops.rows32_pack_rows(a, lda, N, K, arows, K, 0)
    h3 = torch.zeros(N, H, device="cuda")
    c3 = torch.zeros(N, H, device="cuda")
    ops.rows32(None, 0, packed, N, K, 4 * H, h3, H, bias=bias, cell_units=H, c_prev=cp, c_sn=H, c_out=c3, co_sn=H, zoneout=zone,
               h_prev=hprev, hp_sn=H, a_rows=(arows, K, 0))
    bad = (h3 != h2).any(dim=1).nonzero().flatten().tolist()
    assert torch.equal(h3, h2) and torch.equal(c3, c), ("rows that differ", bad)
    # no c_prev = zeros
    ops.rows32(a, lda, packed, N, K, 4 * H, h, H + 8, bias=bias, cell_units=H, c_out=c, co_sn=H)
    assert np.abs(c.cpu().numpy() - i * j).max() < 4e-5
