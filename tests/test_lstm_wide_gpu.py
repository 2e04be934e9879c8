"""The persistent wide-cell LSTM kernels (csrc/lstm_wide.hip: register-resident W_h, state exchanged through the
sentinel-filled history arrays themselves) against the one-launch-per-step kernels on the same operands,
forward and backward: bf16 storage, and fp32 storage in the `mixed` arrangement (forward three split-bf16 passes over
pre-split weights, backward one bf16 pass over bf16 copies).  Partial row groups, per-row lengths, repeated launches,
the status word."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev, N, T, H, seed, masked, f32):
    g = torch.Generator().manual_seed(seed)
    P, padl = T + 1, 1                       # the decoder's slot layout: slot 0 = zero initial state
    rows = N * P
    bf = torch.bfloat16
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    d = dict(N=N, T=T, H=H, P=P, padl=padl, f32=f32)
    d["lengths"] = torch.randint(1, T + 1, (N,), generator=g, dtype=torch.int32).to(dev) if masked else None
    d["xg"] = mk(rows, 4 * H).to(dev)
    w = mk(H, 4 * H, sc=1.0 / H ** 0.5)
    d["dh"] = mk(rows, H, sc=0.1).to(dev)
    if f32:
        d["wh"] = w.to(dev).contiguous()
        d["whT"] = w.t().contiguous().to(dev)
        hi = d["whT"].to(bf)
        d["whT_hi"], d["whT_lo"] = hi, (d["whT"] - hi.float()).to(bf)
        d["wh_bf16"] = d["wh"].to(bf)
    else:
        d["wh"] = w.to(bf).to(dev).contiguous()
        d["whT"] = w.t().contiguous().to(bf).to(dev)
    return d


def _run(dev, d, wide):
    from nspeech_amd import ops
    N, T, H, P, padl, f32 = (d[k] for k in ("N", "T", "H", "P", "padl", "f32"))
    rows = N * P
    D = torch.float32 if f32 else torch.bfloat16
    out = dict(h=torch.zeros(rows * H, dtype=D, device=dev), c=torch.zeros(rows * H, device=dev),
               g=torch.zeros(rows * 4 * H, dtype=D, device=dev), dg=torch.zeros(rows * 4 * H, dtype=D, device=dev))
    if f32:
        out["dgb"] = torch.zeros(rows * 4 * H, dtype=torch.bfloat16, device=dev)
    work = torch.zeros(N * H + 64, device=dev)
    ops.F32_PASSES = 3 if f32 else 0
    fp = ops.lstm_seq_params(N, T, H, P, padl, d["xg"], 4 * H, d["whT"], None, d["lengths"], False, out["h"], H, out["c"],
                             out["g"], whT_hi=d.get("whT_hi"), whT_lo=d.get("whT_lo"))
    ops.F32_PASSES = 1 if f32 else 0
    bp = ops.lstm_seq_params(N, T, H, P, padl, d["xg"], 4 * H, None, d["wh"], d["lengths"], False, out["h"], H, out["c"],
                             out["g"], dh=d["dh"], ld_dh=H, dgates=out["dg"], work=work, wh_bf16=d.get("wh_bf16"),
                             dgates_bf16=out.get("dgb"))
    ops.F32_PASSES = 0
    if wide:
        assert ops.lstm_wide_supported(bp, True)
        w = torch.zeros(ops.lstm_wide_work_floats(fp), device=dev)
        if ops.lstm_wide_supported(fp, False):
            for _ in range(2):      # a second launch re-fills the sentinel itself
                ops.lstm_wide("fwd", fp, w)
            torch.cuda.synchronize()
            assert int(w[:1].view(torch.int32).item()) == 0
        else:                       # beyond 32 rows the forward pass keeps the step launches; the backward kernels take 64
            assert N > 32
            ops.lstm_seq_call("fwd", fp)
        for _ in range(2):
            ops.lstm_wide("bwd", bp, w)
        torch.cuda.synchronize()
        assert int(w[:1].view(torch.int32).item()) == 0
    else:
        ops.lstm_seq_call("fwd", fp)
        ops.lstm_seq_call("bwd", bp)
        torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("N,T,H,masked", [(32, 25, 1024, False), (16, 9, 256, False), (20, 14, 512, True), (5, 7, 1024, True),
                                              (60, 6, 1024, True)])
def test_wide_matches_per_step_kernels(dev, N, T, H, masked, f32):
    d = _setup(dev, N, T, H, seed=N + T, masked=masked, f32=f32)
    ref = _run(dev, d, wide=False)
    got = _run(dev, d, wide=True)
    for k in ref:
        a, b = got[k].float(), ref[k].float()
        scale = b.abs().max().item() + 1e-6
        err = (a - b).abs().max().item()
        if f32 and k in ("h", "c", "g"):
            tol = 2e-5             # both are three-pass split-bf16 products of the same operands: summation order only
        elif f32:
            tol = 2e-2             # one bf16 pass; "dgb" is a bf16 copy (one ulp = 8e-3)
        else:
            tol = 3e-2             # bf16 storage: the states are rounded every step
        assert err <= tol * scale, (k, err, scale)
        assert (a - b).abs().mean().item() <= 0.1 * tol * scale, k


def test_wide_refuses_what_it_cannot_hold(dev):
    from nspeech_amd import _lib as L
    from nspeech_amd import ops
    z = torch.zeros(64, device="cuda")
    zb = z.bfloat16()
    p = ops.lstm_seq_params(4, 3, 96, 4, 1, z, 384, zb, None, None, False, zb, 96, z, zb)           # H not in {256, 512, 1024}
    assert not ops.lstm_wide_supported(p, False)
    with pytest.raises(L.NSError, match="unsupported"):
        ops.lstm_wide("fwd", p, torch.zeros(1024, device="cuda"))
    p = ops.lstm_seq_params(48, 3, 1024, 4, 1, z, 4096, zb, None, None, False, zb, 1024, z, zb)     # 3 row groups x 128 > 256
    assert not ops.lstm_wide_supported(p, False)
