"""The persistent whole-sequence BiLSTM kernels (cluster of workgroups, register-resident W_h,
tagged-granule exchange) against the one-launch-per-step kernels on the same bf16 inputs,
forward and backward, with per-row lengths, partial row groups and repeated launches."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev, N, T, H, seed, masked):
    from nspeech_amd import ops
    g = torch.Generator().manual_seed(seed)
    P, padl = T + 4, 2
    rows = N * P
    bf = torch.bfloat16
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    data = dict(N=N, T=T, H=H, P=P, padl=padl)
    data["lengths"] = (torch.randint(1, T + 1, (N,), generator=g, dtype=torch.int32).to(dev) if masked else None)
    if masked:
        data["lengths"][0] = T
    for d in ("fw", "bw"):
        data["xg_" + d] = mk(rows, 4 * H).to(dev)
        w = mk(H, 4 * H, sc=1.0 / H ** 0.5)
        data["wh_" + d] = w.to(bf).to(dev).contiguous()
        data["whT_" + d] = w.t().contiguous().to(bf).to(dev)
        data["dh"] = mk(rows, 2 * H, sc=0.1).to(dev)
    return data


def _run(dev, data, cluster):
    from nspeech_amd import ops
    N, T, H, P, padl = (data[k] for k in ("N", "T", "H", "P", "padl"))
    rows = N * P
    bf = torch.bfloat16
    out = dict(h=torch.zeros(rows * 2 * H, dtype=bf, device=dev))
    fp, bp = [], []
    for di, d in enumerate(("fw", "bw")):
        out["c_" + d] = torch.zeros(rows * H, device=dev)
        out["g_" + d] = torch.zeros(rows * 4 * H, dtype=bf, device=dev)
        out["dg_" + d] = torch.zeros(rows * 4 * H, dtype=bf, device=dev)
        work = torch.zeros(N * H + 64, device=dev)
        common = dict(h_off=di * H)
        fp.append(ops.lstm_seq_params(N, T, H, P, padl, data["xg_" + d], 4 * H, data["whT_" + d], None, data["lengths"],
                                      d == "bw", out["h"], 2 * H, out["c_" + d], out["g_" + d], **common))
        bp.append(ops.lstm_seq_params(N, T, H, P, padl, data["xg_" + d], 4 * H, None, data["wh_" + d], data["lengths"],
                                      d == "bw", out["h"], 2 * H, out["c_" + d], out["g_" + d], dh=data["dh"],
                                      ld_dh=2 * H, dgates=out["dg_" + d], work=work, dh_off=di * H, **common))
        out["_w" + d] = work
    if cluster:
        assert ops.lstm_cluster_supported(fp[0])
        w = torch.zeros(ops.lstm_cluster_work_floats(fp[0]), device=dev)
        for _ in range(2):      # second launch re-initialises the exchange state itself
            ops.lstm_cluster("fwd", fp[0], fp[1], w)
        torch.cuda.synchronize()
        assert int(w[:1].view(torch.int32).item()) == 0
        for _ in range(2):
            ops.lstm_cluster("bwd", bp[0], bp[1], w)
        torch.cuda.synchronize()
        assert int(w[:1].view(torch.int32).item()) == 0
    else:
        ops.lstm_seq2("fwd", fp[0], fp[1])
        ops.lstm_seq2("bwd", bp[0], bp[1])
        torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("N,T,H,masked", [(16, 9, 64, False), (20, 33, 256, True), (32, 61, 256, False), (5, 12, 128, True),
                                              (40, 21, 256, True), (48, 17, 192, False)])
def test_cluster_matches_per_step_kernels(dev, N, T, H, masked):
    data = _setup(dev, N, T, H, seed=N + T, masked=masked)
    ref = _run(dev, data, cluster=False)
    got = _run(dev, data, cluster=True)
    for k in ("h", "c_fw", "c_bw", "g_fw", "g_bw", "dg_fw", "dg_bw"):
        a, b = got[k].float(), ref[k].float()
        scale = b.abs().max().item() + 1e-6
        err = (a - b).abs().max().item()
        # same bf16 operands, different fp32 summation order; bf16 rounding of stored states
        assert err <= 3e-2 * scale, (k, err, scale)
        assert (a - b).abs().mean().item() <= 2e-3 * scale, k


@pytest.mark.parametrize("N,T,H,masked", [(40, 21, 256, True), (32, 17, 192, False)])
def test_interleaved_row_groups_form_still_matches(dev, monkeypatch, N, T, H, masked):
    """Two row groups interleaved per workgroup (R = 2) is no longer any shape's default (re-measured slower at the end
    of round 3); NS_CLUSTER_DBG bits 512 / 1024 / 2048 force it, and it has to stay correct."""
    monkeypatch.setenv("NS_CLUSTER_DBG", str(512 + 1024 + 2048))
    data = _setup(dev, N, T, H, seed=N + T, masked=masked)
    ref = _run(dev, data, cluster=False)
    got = _run(dev, data, cluster=True)
    for k in ("h", "c_fw", "c_bw", "g_fw", "g_bw", "dg_fw", "dg_bw"):
        a, b = got[k].float(), ref[k].float()
        scale = b.abs().max().item() + 1e-6
        assert (a - b).abs().max().item() <= 3e-2 * scale, k
        assert (a - b).abs().mean().item() <= 2e-3 * scale, k


@pytest.mark.parametrize("N,T,H,masked", [(32, 41, 256, True), (20, 33, 256, True), (5, 12, 128, True), (16, 9, 64, False),
                                              (40, 21, 192, False), (1, 7, 256, True)])
def test_fp32_state_cluster_forward_matches_per_step_kernels(dev, N, T, H, masked):
    """lstm_cluster3_fwd_kernel (fp32 state, three split-bf16 passes, the encoder BiLSTM of `mixed`) against the
    per-step fp32 kernels in the same arithmetic: h and c agree to summation order, the gates it saves as bf16 to bf16
    rounding, the bf16 copy of h is the rounded h; a second launch re-initialises its exchange state; and the bf16
    backward cluster kernel accepts what it saved."""
    from nspeech_amd import ops
    g = torch.Generator().manual_seed(N * 7 + T)
    P, padl = T + 4, 2
    rows = N * P
    bf = torch.bfloat16
    lengths = None
    if masked:
        lengths = torch.randint(1, T + 1, (N,), generator=g, dtype=torch.int32)
        lengths[0] = T
        lengths = lengths.to(dev)
    xg, whT, hi, lo, wh16 = {}, {}, {}, {}, {}
    for d in ("fw", "bw"):
        xg[d] = torch.randn(rows, 4 * H, generator=g).to(dev)
        w = torch.randn(H, 4 * H, generator=g) / H ** 0.5
        whT[d] = w.t().contiguous().to(dev)
        hi[d] = whT[d].to(bf)
        lo[d] = (whT[d] - hi[d].float()).to(bf)
        wh16[d] = w.to(bf).to(dev).contiguous()
    dh = (torch.randn(rows, 2 * H, generator=g) * 0.1).to(dev)

    def run(cluster):
        out = dict(h=torch.zeros(rows * 2 * H, device=dev), hb=torch.zeros(rows * 2 * H, dtype=bf, device=dev))
        pair = []
        ops.F32_PASSES = 3
        for di, d in enumerate(("fw", "bw")):
            out["c_" + d] = torch.zeros(rows * H, device=dev)
            out["g_" + d] = torch.zeros(rows * 4 * H, dtype=bf if cluster else torch.float32, device=dev)
            pair.append(ops.lstm_seq_params(N, T, H, P, padl, xg[d], 4 * H, whT[d], None, lengths, d == "bw", out["h"], 2 * H,
                                            out["c_" + d], out["g_" + d], h_off=di * H, whT_hi=hi[d], whT_lo=lo[d],
                                            h_bf16=out["hb"] if cluster else None, h_bf16_off=di * H, ld_h_bf16=2 * H))
        ops.F32_PASSES = 0
        if cluster:
            assert ops.lstm_cluster_supported(pair[0], pair[1], False)
            w = torch.zeros(ops.lstm_cluster_work_floats(pair[0]), device=dev)
            for _ in range(2):
                ops.lstm_cluster("fwd", pair[0], pair[1], w)
            torch.cuda.synchronize()
            assert int(w[:1].view(torch.int32).item()) == 0
            out["work"] = w
        else:
            ops.lstm_seq2("fwd", pair[0], pair[1])
            torch.cuda.synchronize()
        return out

    ref, got = run(False), run(True)
    for k, tol in (("h", 2e-5), ("c_fw", 2e-5), ("c_bw", 2e-5), ("g_fw", 6e-3), ("g_bw", 6e-3)):
        a, b = got[k].float(), ref[k].float()
        err = (a - b).abs().max().item()
        assert err <= tol * (b.abs().max().item() + 1e-6), (k, err)
    assert torch.equal(got["hb"], got["h"].to(bf))
    # the bf16 backward cluster kernel on what the fp32 forward saved (gates bf16, c fp32) against the per-step bf16
    # backward on the same operands
    res = []
    for cluster in (False, True):
        bp, dgs = [], []
        for di, d in enumerate(("fw", "bw")):
            dg = torch.zeros(rows * 4 * H, dtype=bf, device=dev)
            work = torch.zeros(N * H + 64, device=dev)
            dgs.append(dg)
            bp.append(ops.lstm_seq_params(N, T, H, P, padl, xg[d], 4 * H, None, wh16[d], lengths, d == "bw", got["hb"], 2 * H,
                                          got["c_" + d], got["g_" + d], dh=dh, ld_dh=2 * H, dgates=dg, work=work,
                                          h_off=di * H, dh_off=di * H))
        if cluster:
            assert ops.lstm_cluster_supported(bp[0], bp[1], True)
            ops.lstm_cluster("bwd", bp[0], bp[1], got["work"])
            torch.cuda.synchronize()
            assert int(got["work"][:1].view(torch.int32).item()) == 0
        else:
            ops.lstm_seq2("bwd", bp[0], bp[1])
            torch.cuda.synchronize()
        res.append(dgs)
    for a, b in zip(res[1], res[0]):
        sc = b.float().abs().max().item() + 1e-6
        assert (a.float() - b.float()).abs().max().item() <= 3e-2 * sc
