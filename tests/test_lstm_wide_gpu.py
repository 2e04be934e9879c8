"""The persistent wide-cell LSTM kernels (register-resident W_h slices, flag + write-through exchange through
the history arrays) against the one-launch-per-step kernels on the same inputs: bf16 storage, and fp32 storage
with the 3-pass forward / bf16-side-copy backward used by the `mixed` model mode.  Slot layout as in the decoder
(P = T + 1, slot 0 = zero state), per-row lengths, partial row groups, repeated launches."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(dev, N, T, H, f32, masked, seed=0):
    from nspeech_amd import ops
    g = torch.Generator().manual_seed(seed)
    P, padl = T + 1, 1
    rows = N * P
    bf = torch.bfloat16
    st = torch.float32 if f32 else bf
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    d = dict(N=N, T=T, H=H, P=P, padl=padl, st=st)
    d["lengths"] = torch.randint(1, T + 1, (N,), generator=g, dtype=torch.int32).to(dev) if masked else None
    d["xg"] = mk(rows, 4 * H).to(dev)
    w = mk(H, 4 * H, sc=1.0 / H ** 0.5)
    d["wh"] = w.to(st).to(dev).contiguous()
    d["whT"] = w.t().contiguous().to(st).to(dev)
    d["dh"] = mk(rows, H, sc=0.1).to(dev)
    if f32:
        d["whT_hi"] = torch.empty(4 * H * H, dtype=bf, device=dev)
        d["whT_lo"] = torch.empty(4 * H * H, dtype=bf, device=dev)
        ops.split_hi_lo(d["whT"], d["whT_hi"], d["whT_lo"], 4 * H * H)
        d["wh_bf16"] = d["wh"].to(bf).contiguous()
    return d


def _run(dev, d, wide):
    from nspeech_amd import ops
    N, T, H, P, padl, st = (d[k] for k in ("N", "T", "H", "P", "padl", "st"))
    rows = N * P
    f32 = st == torch.float32
    out = dict(h=torch.zeros(rows * H, dtype=st, device=dev), c=torch.zeros(rows * H, device=dev),
               g=torch.zeros(rows * 4 * H, dtype=st, device=dev), dg=torch.zeros(rows * 4 * H, dtype=st, device=dev))
    dgb = torch.zeros(rows * 4 * H, dtype=torch.bfloat16, device=dev) if f32 else None
    work = torch.zeros(2 * N * H + 64, device=dev)
    ops.F32_PASSES = 3
    fp = ops.lstm_seq_params(N, T, H, P, padl, d["xg"], 4 * H, d["whT"], None, d["lengths"], False, out["h"], H, out["c"],
                             out["g"], whT_hi=d.get("whT_hi"), whT_lo=d.get("whT_lo"))
    ops.F32_PASSES = 1
    bp = ops.lstm_seq_params(N, T, H, P, padl, d["xg"], 4 * H, None, d["wh"], d["lengths"], False, out["h"], H, out["c"],
                             out["g"], dh=d["dh"], ld_dh=H, dgates=out["dg"], work=work, wh_bf16=d.get("wh_bf16"),
                             dgates_bf16=dgb)
    ops.F32_PASSES = 0
    if wide:
        assert ops.lstm_wide_supported(fp, False) and ops.lstm_wide_supported(bp, True)
        w = torch.zeros(ops.lstm_wide_work_floats(fp), device=dev)
        for _ in range(2):      # the second launch re-initialises the flags itself
            ops.lstm_wide("fwd", fp, w)
        torch.cuda.synchronize()
        assert int(w[:1].view(torch.int32).item()) == 0
        for _ in range(2):
            ops.lstm_wide("bwd", bp, w)
        torch.cuda.synchronize()
        assert int(w[:1].view(torch.int32).item()) == 0
    else:
        L = __import__("nspeech_amd._lib", fromlist=["x"])
        L.call("ns_lstm_seq_fwd", fp, ops.stream())
        L.call("ns_lstm_seq_bwd", bp, ops.stream())
        torch.cuda.synchronize()
    if dgb is not None:
        out["dgb"] = dgb
    return out


@pytest.mark.parametrize("N,T,H,f32,masked", [(32, 12, 256, False, False), (20, 9, 128, False, True), (32, 6, 1024, False, False),
                                              (32, 12, 256, True, False), (20, 9, 128, True, True), (32, 6, 1024, True, False),
                                              (8, 7, 128, True, False), (40, 5, 256, False, True)])
def test_wide_matches_per_step_kernels(dev, N, T, H, f32, masked):
    d = _case(dev, N, T, H, f32, masked)
    ref = _run(dev, d, wide=False)
    got = _run(dev, d, wide=True)
    for k in ("h", "c", "g", "dg", "dgb"):
        if k not in ref:
            continue
        a, b = got[k].float(), ref[k].float()
        scale = b.abs().max().item() + 1e-6
        err = (a - b).abs().max().item()
        # same operands, different fp32 summation order (plus bf16 rounding of stored states in bf16 storage);
        # fp32 backward: the recurrent operand is the bf16 side copy in both paths
        tol = ((8e-3 if k == "dgb" else 2e-3 if k == "dg" else 2e-4) if f32 else 3e-2)   # dgb: one bf16 ulp
        assert err <= tol * scale, (k, err, scale)
