"""Time ns_gru_seq_fwd / _bwd at the Tacotron-1 benchmark shapes (HIP events on the launch stream):
post-CBHG BiGRU(128) N 32 x T 1000, encoder BiGRU(128) T 160, decoder GRU(256) T 200."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nspeech_amd import ops

dev = torch.device("cuda:0")


def run(N, T, H, ndir, dtype, passes, reps=5):
    g = torch.Generator().manual_seed(1)
    P, padl = T + 16, 8
    rows = N * P
    Tt = torch.bfloat16 if dtype == "bf16" else torch.float32
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    ldh = ndir * H
    hb = torch.zeros(rows * ldh, dtype=Tt, device=dev)
    dh = rnd(rows * ldh, sc=0.1)
    ops.F32_PASSES = passes
    pf, pb = [], []
    keep = []
    for di in range(ndir):
        b = dict(xg=rnd(rows * 2 * H), xc=rnd(rows * H), wg=rnd(H * 2 * H, sc=H ** -0.5).to(Tt), wc=rnd(H * H, sc=H ** -0.5).to(Tt),
                 ru=torch.zeros(rows * 2 * H, device=dev), c=torch.zeros(rows * H, device=dev),
                 rh=torch.zeros(rows * H, dtype=Tt, device=dev), dzg=torch.zeros(rows * 2 * H, dtype=Tt, device=dev),
                 dzc=torch.zeros(rows * H, dtype=Tt, device=dev))
        keep.append(b)
        common = (hb, N, T, H, P, padl, di == 1, None, b["xg"], b["xc"], b["wg"], b["wc"], b["wg"], 2 * H, b["wc"], H,
                  (hb, di * H), ldh, b["ru"], b["c"], b["rh"])
        pf.append(ops.gru_seq_params(*common))
        pb.append(ops.gru_seq_params(*common, dh=(dh, di * H), ld_dh=ldh, dzg=b["dzg"], dzc=b["dzc"]))
    p1f, p1b = (pf[1], pb[1]) if ndir == 2 else (None, None)
    work = torch.zeros(ops.gru_seq_work_floats(pf[0]), device=dev)
    res = []
    for name, a, b in (("fwd", pf[0], p1f), ("bwd", pb[0], p1b)):
        ops.gru_seq(name, a, b, work)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gru_seq(name, a, b, work)
        e1.record()
        torch.cuda.synchronize()
        assert int(work[:1].view(torch.int32).item()) == 0
        ms = e0.elapsed_time(e1) / reps
        res.append("%s %.3f ms = %.2f us/step" % (name, ms, ms * 1e3 / T))
    print("N %d T %4d H %d dirs %d %s passes %d: %s" % (N, T, H, ndir, dtype, passes, "; ".join(res)), flush=True)
    if os.environ.get("NS_GRU_TRACE") and passes == 3:
        ops.gru_seq("fwd", pf[0], p1f, work)
        torch.cuda.synchronize()
        tr = work.view(torch.int64)[-16 * 8:].view(16, 8).cpu().numpy()
        names = ["B0 wait", "phase-1 reads + MFMA", "gates, r*h, stores", "B1 wait", "phase-2 reads + MFMA", "tail"]
        for w in (range(8) if os.environ.get("NS_GRU_TRACE_ALL") else (0, 3, 7)):
            print("   wave %d, us per step: %s" % (w, ", ".join("%s %.2f" % (nm, tr[w, k] * 0.01 / T) for k, nm in enumerate(names))))


for dtype, passes in (("fp32", 3), ("fp32", 1), ("bf16", 0)):
    run(32, 1000, 128, 2, dtype, passes)
    run(32, 160, 128, 2, dtype, passes)
    run(32, 200, 256, 1, dtype, passes)
