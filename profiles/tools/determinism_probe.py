#!/usr/bin/env python3
"""Runs forward + backward of one Tacotron-2 batch several times in one process and lists every named model buffer
whose contents differ from the first run by more than summation-order rounding: finds the first kernel of the backward
chain whose result is not repeatable.  Usage: python profiles/tools/determinism_probe.py [mode] [N Ti To] [runs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd.models import create_model  # noqa: E402
from util import make_batch  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    N, Ti, To = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (8, 48, 100)
    runs = int(sys.argv[5]) if len(sys.argv) > 5 else 12
    hp = hparams_mod.load("taco2")
    m = create_model("taco2", hp, device="cuda:0", dtype=mode, seed=5)
    for off in os.environ.get("PROBE_OFF", "").split(","):      # attn / wide / cluster: fall back to the per-step kernels
        if off == "attn":
            m.use_attn_cluster = False
        elif off == "wide":
            m.use_wide = False
        elif off == "cluster":
            m.use_cluster = False
    m.add_optimizer(0)
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=3)
    ref = None
    use_nccl = os.environ.get("PROBE_NCCL", "0") == "1"     # runs 1.. go through a one-rank RCCL GradReducer
    if use_nccl:
        import torch.distributed as dist
        from nspeech_amd import parallel
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29672")
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    for r in range(runs):
        if use_nccl and r == 1:
            m.reducer = parallel.make_reducer(m, force=True)
        m.initialize(inputs, lengths, None, mel, lin)
        m.backward()
        if m.reducer is not None:
            m.reducer.wait()
        torch.cuda.synchronize()
        m.check_status()
        snap = {k: v.clone() for k, v in m._bufs.items() if torch.is_tensor(v) and v.is_floating_point()}
        snap["flat_g"] = m.flat_g.clone()
        if ref is None:
            ref = snap
            continue
        bad = []
        for k, v in snap.items():
            a, b = ref[k].float(), v.float()
            if a.shape != b.shape:
                continue
            d = (a - b).abs().max().item()
            sc = a.abs().max().item()
            if d > float(os.environ.get("PROBE_TOL", "2e-5")) * sc or d != d:
                bad.append((k, d, sc))
        if len(bad) > 1 or os.environ.get("PROBE_VERBOSE"):
            print("run %d: %d buffers differ" % (r, len(bad)), flush=True)
        outcomes = locals().setdefault("outcomes", [])
        outcomes.append(len(bad) > 1)
        for k, d, sc in sorted(bad):
            print("    %-28s diff %.3e  scale %.3e" % (k, d, sc))
        if len(bad) > 3 and os.environ.get("PROBE_WHERE", "0") == "1":
            S1 = To // hp.outputs_per_step + 1
            if not getattr(main, "_dumped", False):
                main._dumped = True
                Tia = ref["d_energy"].numel() // (N * S1)
                al = m.alignments.permute(0, 2, 1)            # [N, S, Ti] (slot s+1 = row s)
                for n in range(N):
                    A_, B_ = ref["d_energy"].view(N, S1, Tia)[n, S1 - 1].double().cpu(), snap["d_energy"].view(N, S1, Tia)[n, S1 - 1].double().cpu()
                    if not (A_ != B_).any():
                        continue
                    L_ = int(lengths[n])
                    a_ = al[n, S1 - 2].double().cpu()[:L_]
                    d0 = snap["d_a0"].view(N, S1, Tia)[n, S1 - 1].double().cpu()[:L_]
                    want = a_ * (d0 - (a_ * d0).sum())
                    print("      n=%d length %d  sum(align) %.6f" % (n, L_, float(a_.sum())))
                    print("      expected %s" % [float("%.4e" % v) for v in want[:10]])
                    print("      run 0    %s" % [float("%.4e" % v) for v in A_[:10]])
                    print("      this run %s" % [float("%.4e" % v) for v in B_[:10]])
                    print("      |run0 - expected| max %.3e   |this - expected| max %.3e   scale %.3e" % (
                        float((A_[:L_] - want).abs().max()), float((B_[:L_] - want).abs().max()), float(want.abs().max())))
                    ts = max(1, (L_ + 7) // 8)
                    print("      positions 12..20: align %s" % [float("%.5e" % v) for v in a_[12:21]])
                    print("      da0 %s" % [float("%.5e" % v) for v in d0[12:21]])
                    print("      run 0 de %s" % [float("%.5e" % v) for v in A_[12:21]])
                    print("      this  de %s" % [float("%.5e" % v) for v in B_[12:21]])
                    print("      run 0 de/align %s" % [float("%.5e" % v) for v in (A_[12:21] / a_[12:21])])
                    print("      this  de/align %s" % [float("%.5e" % v) for v in (B_[12:21] / a_[12:21])])
                    for nm, X in (("run 0", A_), ("this run", B_)):
                        imp = (d0 - X[:L_] / a_)              # implied dot per position
                        print("      %s implied dot per workgroup slice: %s" % (nm, [float("%.5e" % imp[g * ts:(g + 1) * ts].mean()) for g in range(8) if g * ts < L_]))
            for name, width in (("d_energy", None), ("d_q", hp.attention_dim), ("d_ga", 4 * hp.attention_dim),
                                ("d_p2", 128), ("d_f1", 256)):
                if name not in snap:
                    continue
                a, b = ref[name].float(), snap[name].float()
                w = width or (a.numel() // (N * S1))
                dd = (a - b)[:N * S1 * w].view(N, S1, w).abs()
                for n in range(N):
                    slots = [int(x) for x in torch.nonzero(dd[n].amax(1) > 0).flatten().tolist()]
                    if slots:
                        top = slots[-1]
                        cols = [int(x) for x in torch.nonzero(dd[n, top] > 0).flatten().tolist()]
                        print("      %-9s n=%d slots %d..%d (%d of them); at slot %d cols %s max %.3e" % (
                            name, n, slots[0], top, len(slots), top, cols[:12], dd[n, top].max().item()))
    print("done: %d of %d runs differ from run 0" % (sum(outcomes), len(outcomes)))


if __name__ == "__main__":
    main()
