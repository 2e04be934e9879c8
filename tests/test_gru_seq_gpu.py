"""The persistent whole-sequence GRU kernels (csrc/gru.hip: ns_gru_seq_fwd / _bwd) against a float64 restatement of
tf.contrib.rnn.GRUCell under (bidirectional_)dynamic_rnn (modules.py:172-181, tacotron.py:69-76; the cell as
oracle/taco1_oracle.py: gru_cell, lengths as its bigru): forward history and saved gates, backward gate gradients and the
gradient wrt the initial state, through the C ABI.  H = 128: one workgroup per chain; H = 256: clusters of four with the
granule exchange and the K split.  Per-row lengths, ragged row groups (N not a multiple of 16), one and two directions,
an initial state, T = 1, repeated launches on the same work buffer."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(d, H, T, N, lengths, h0):
    """float64 autograd: returns h [N,T,H] (zeros past the length), ru, c, rh and d(sum h * dh)/d(xg, xc, h0)."""
    xg = d["xg"].double().clone().requires_grad_(True)
    xc = d["xc"].double().clone().requires_grad_(True)
    h0t = None if h0 is None else h0.double().clone().requires_grad_(True)
    wg, wc = d["wg"].double(), d["wc"].double()
    h = torch.zeros(N, H, dtype=torch.float64) if h0t is None else h0t
    L = torch.full((N,), T) if lengths is None else lengths.long()
    hs, rus, cs, rhs = [None] * T, [None] * T, [None] * T, [None] * T
    for t in (range(T - 1, -1, -1) if d["reverse"] else range(T)):
        ru = torch.sigmoid(xg[:, t] + h @ wg)
        r, u = ru[:, :H], ru[:, H:]
        rh = r * h
        c = torch.tanh(xc[:, t] + rh @ wc)
        h2 = u * h + (1 - u) * c
        m = (t < L).double()[:, None]
        hs[t] = m * h2
        h = m * h2 + (1 - m) * h
        rus[t], cs[t], rhs[t] = ru, c, rh
    hh = torch.stack(hs, 1)
    (hh * d["dh"].double()).sum().backward()
    return dict(h=hh.detach(), ru=torch.stack(rus, 1).detach(), c=torch.stack(cs, 1).detach(), rh=torch.stack(rhs, 1).detach(),
                dzg=xg.grad, dzc=xc.grad, dh0=None if h0t is None else h0t.grad, valid=(torch.arange(T)[None, :] < L[:, None]))


def _case(dev, N, T, H, ndir, masked, with_h0, dtype, passes, seed):
    from nspeech_amd import ops
    g = torch.Generator().manual_seed(seed)
    P, padl = T + 5, 3
    rows = N * P
    Tt = torch.bfloat16 if dtype == "bf16" else torch.float32
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    lengths = None
    if masked:
        lengths = torch.randint(1, T + 1, (N,), generator=g, dtype=torch.int32)
        lengths[0] = T
    h0 = rnd(N, H, sc=0.5) if with_h0 else None
    dirs = []
    for di in range(ndir):
        wg, wc = rnd(H, 2 * H, sc=1.0 / H ** 0.5), rnd(H, H, sc=1.0 / H ** 0.5)
        if Tt == torch.bfloat16:            # the kernel sees bf16 weights: so does the reference
            wg, wc = wg.to(Tt).float(), wc.to(Tt).float()
        dirs.append(dict(reverse=di == 1, xg=rnd(N, T, 2 * H), xc=rnd(N, T, H), wg=wg, wc=wc, dh=rnd(N, T, H, sc=0.3)))
    srows = (N + 15) // 16 * 16 * T          # the saved gates cover whole 16-row groups (ns_gru_seq_params.ru)
    ldh = ndir * H + 8                       # the history is a column block of a wider buffer
    hb = torch.zeros(rows * ldh, dtype=Tt, device=dev)
    dh = torch.zeros(rows * ldh, dtype=torch.float32, device=dev)
    lens_d = None if lengths is None else lengths.to(dev)
    h0_d = None if h0 is None else h0.to(dev).contiguous()
    ops.F32_PASSES = passes
    pf, pb, bufs = [], [], []
    for di, d in enumerate(dirs):
        def padded(x, C):
            full = torch.zeros(N, P, C)
            full[:, padl:padl + T] = x
            return full.reshape(-1).to(dev)
        b = dict(xg=padded(d["xg"], 2 * H), xc=padded(d["xc"], H),
                 wg=d["wg"].to(Tt).to(dev).contiguous(), wc=d["wc"].to(Tt).to(dev).contiguous(),
                 wgT=d["wg"].t().contiguous().to(Tt).to(dev), wcT=d["wc"].t().contiguous().to(Tt).to(dev),
                 ru=torch.full((srows * 2 * H,), 7.0, device=dev), c=torch.full((srows * H,), 7.0, device=dev),
                 rh=torch.zeros(rows * H, dtype=Tt, device=dev), dzg=torch.zeros(rows * 2 * H, dtype=Tt, device=dev),
                 dzc=torch.zeros(rows * H, dtype=Tt, device=dev), dh0=torch.zeros(N * H, device=dev))
        dview = dh.view(N, P, ldh)
        dview[:, padl:padl + T, di * H:(di + 1) * H] = d["dh"].to(dev)
        bufs.append(b)
        common = (hb, N, T, H, P, padl, d["reverse"], lens_d, b["xg"], b["xc"], b["wgT"], b["wcT"], b["wg"], 2 * H, b["wc"], H,
                  (hb, di * H), ldh, b["ru"], b["c"], b["rh"])
        pf.append(ops.gru_seq_params(*common, h_init=h0_d, ld_hi=H))
        pb.append(ops.gru_seq_params(*common, h_init=h0_d, ld_hi=H, dh=(dh, di * H), ld_dh=ldh, dzg=b["dzg"], dzc=b["dzc"],
                                     dh_init=b["dh0"] if with_h0 else None, ld_dhi=H))
    p1f, p1b = (pf[1], pb[1]) if ndir == 2 else (None, None)
    assert ops.gru_seq_supported(pf[0], p1f, backward=False) and ops.gru_seq_supported(pb[0], p1b, backward=True)
    work = torch.zeros(ops.gru_seq_work_floats(pf[0]), device=dev)
    for _ in range(2):                       # the second launch re-initialises the exchange state itself
        ops.gru_seq("fwd", pf[0], p1f, work)
    torch.cuda.synchronize()
    assert int(work[:1].view(torch.int32).item()) == 0
    for _ in range(2):
        ops.gru_seq("bwd", pb[0], p1b, work)
    torch.cuda.synchronize()
    assert int(work[:1].view(torch.int32).item()) == 0
    ops.F32_PASSES = 0
    out = []
    hv = hb.float().cpu().view(N, P, ldh)
    assert float(hv[:, :padl].abs().max()) == 0.0 and float(hv[:, padl + T:].abs().max()) == 0.0       # pad rows untouched
    assert float(hv[:, :, ndir * H:].abs().max()) == 0.0                                               # other columns too
    for di, (d, b) in enumerate(zip(dirs, bufs)):
        ref = _reference(d, H, T, N, lengths, h0)
        cut = lambda x, C: x.float().cpu().view(N, P, C)[:, padl:padl + T]

        def saved(x, nsec):
            """ns_gru_seq's own order [row group][t][workgroup][section][chunk][row][4] -> [N, T, nsec * H]"""
            G = 1 if H == 128 else 4
            U = H // G
            v = x.float().cpu().view((N + 15) // 16, T, G, nsec, U // 4, 16, 4)
            v = v.permute(0, 5, 1, 3, 2, 4, 6).reshape((N + 15) // 16 * 16, T, nsec * H)       # row group, row | t | section, workgroup, chunk, 4
            return v[:N]
        got = dict(h=hv[:, padl:padl + T, di * H:(di + 1) * H], ru=saved(b["ru"], 2), c=saved(b["c"], 1), rh=cut(b["rh"], H),
                   dzg=cut(b["dzg"], 2 * H), dzc=cut(b["dzc"], H), dh0=b["dh0"].cpu().view(N, H) if with_h0 else None)
        out.append((got, ref))
    return out


def _compare(pairs, tol_fwd, tol_bwd):
    for got, ref in pairs:
        v = ref["valid"][:, :, None]
        for k in ("h", "ru", "c", "rh", "dzg", "dzc"):
            a, b = got[k].double(), ref[k]
            if k in ("ru", "c", "rh"):       # saved gates of steps past the length are never used (their gradient is zero)
                a, b = a * v, b * v
            tol = tol_fwd if k in ("h", "ru", "c", "rh") else tol_bwd
            scale = float(b.abs().max()) + 1e-9
            err = float((a - b).abs().max())
            assert err <= tol * scale, (k, err, scale)
        if ref["dh0"] is not None:
            a, b = got["dh0"].double(), ref["dh0"]
            assert float((a - b).abs().max()) <= tol_bwd * (float(b.abs().max()) + 1e-9), "dh0"


SHAPES = [  # N, T, H, directions, lengths, initial state
    (3, 7, 128, 2, True, False), (16, 1, 128, 2, False, False), (32, 40, 128, 2, True, True), (33, 12, 128, 1, True, False),
    (5, 9, 256, 1, False, False), (32, 25, 256, 1, False, False), (20, 13, 256, 2, True, True), (16, 2, 256, 2, True, False),
]


@pytest.mark.parametrize("shape", SHAPES)
def test_three_pass_kernels_match_float64(dev, shape):
    """fp32 storage, three split-bf16 passes: ~fp32 accuracy (the forward pass of `mixed`, both passes of `bf16x3`)."""
    N, T, H, ndir, masked, h0 = shape
    _compare(_case(dev, N, T, H, ndir, masked, h0, "fp32", 3, seed=N + T), 2e-5 * max(1, T) ** 0.5, 5e-5 * max(1, T) ** 0.5)


@pytest.mark.parametrize("shape", SHAPES[2:3] + SHAPES[5:7])
@pytest.mark.parametrize("dtype,passes", [("fp32", 1), ("bf16", 0)])
def test_single_pass_kernels_match_float64(dev, shape, dtype, passes):
    """Operands rounded to bf16 (the backward pass of `mixed`; everything in `bf16`): bf16-level agreement."""
    N, T, H, ndir, masked, h0 = shape
    _compare(_case(dev, N, T, H, ndir, masked, h0, dtype, passes, seed=N + T + 1), 4e-2, 8e-2)
