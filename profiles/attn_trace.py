#!/usr/bin/env python3
"""Phase timings inside the persistent attention-RNN kernels (csrc/attn_cluster.hip) at the benchmark shape:
NS_ATTN_TRACE=1 makes workgroup 0 stamp the 100 MHz clock at every phase boundary of every step; this prints the
mean duration of each phase.  Usage: python profiles/attn_trace.py > profiles/r02_attn_trace.txt"""
import os
import sys

os.environ["NS_ATTN_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd.models import create_model  # noqa: E402

FWD = ["p2", "gates+cell", "q partial", "X2 wait", "q sum", "energies", "softmax", "ctx partial", "X3 wait", "combine"]
BWD = ["history -> LDS", "dalign + E1 publish", "energy pass (MFMA) + E1 wait", "de, dq partials, G, carry", "E2 wait",
       "dq sum + cell", "input grads + publish", "E3 wait", "dp2 / carry", "dp1"]


def report(name, work, labels, S):
    tr = work[-(256 * 16 * 2):].view(torch.int64).view(256, 16).cpu().numpy().astype(np.float64)
    n = len(labels)
    rows = tr[4:min(S, 256) - 1]
    nxt = tr[5:min(S, 256)]
    d = np.concatenate([rows[:, 1:n] - rows[:, 0:n - 1], (nxt[:, 0] - rows[:, n - 1])[:, None]], axis=1) * 10.0   # ns
    print("%s: %.2f us per step" % (name, (nxt[:, 0] - rows[:, 0]).mean() * 1e-2))
    for i, lab in enumerate(labels):
        print("  %-24s %7.2f us" % (lab, d[:, i].mean() * 1e-3))


def main():
    hp = hparams_mod.load("taco2")
    m = create_model("taco2", hp, device="cuda:0", dtype="mixed", seed=1234)
    inputs, lengths, mel, lin = bench.synthetic_batch(hp, 32, 160, 1000, 1234)
    m.add_optimizer(0)
    for _ in range(3):
        m.initialize(inputs, lengths, None, mel, lin)
        m.backward()
    torch.cuda.synchronize()
    m.check_status()
    report("forward", m._bufs["attn_cluster_work"], FWD, 200)
    report("backward", m._bufs["attn_cluster_work_b"], BWD, 200)


if __name__ == "__main__":
    main()
