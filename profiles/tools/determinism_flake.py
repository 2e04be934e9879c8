import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from util import make_batch
from nspeech_amd import hparams as H
from nspeech_amd.models import create_model
hp = H.load("taco2")
N, Ti, To = 2, 160, 10
m = create_model("taco2", hp, device="cuda:0", dtype="mixed", seed=3)
inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=N)
m.initialize(inputs, lengths, None, mel, lin)
names = ("d_wcl", "d_energy", "d_keys", "d_q", "d_ga")
ref = None
worst = {}
for it in range(60):
    for cluster in (False, True):
        m._attn_cluster_fwd = cluster
        m.backward()
        torch.cuda.synchronize()
        out = {k: m._bufs[k].float().clone() for k in names}
        key = cluster
        if ref is None:
            ref = {}
        if key not in ref:
            ref[key] = out
        else:
            for k in names:
                d = float((out[k] - ref[key][k]).abs().max()); sc = float(ref[key][k].abs().max())
                if d > 0:
                    worst[(key, k)] = max(worst.get((key, k), 0.0), d / sc)
print("overlap", m.overlap_wgrads, "worst relative run-to-run differences (path, buffer):", worst)
