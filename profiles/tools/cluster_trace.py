#!/usr/bin/env python3
"""In-kernel slot timings of the persistent BiLSTM forward kernel (NS_CLUSTER_DBG=16: 100 MHz stamps taken by
workgroup 0's compute wave 0 and first poller) at the benchmark shape's expand BiLSTM.
Per slot q (= step * 2 + row group): c0 compute start (after the barrier), c1 cell math done, c2 publish issued;
p4 poller starts waiting for the slot's operands, p5 has them (spins in [6])."""
import os
import sys

os.environ["NS_CLUSTER_DBG"] = "16"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd.models import create_model  # noqa: E402


def main():
    hp = hparams_mod.load("taco2")
    m = create_model("taco2", hp, device="cuda:0", dtype="mixed", seed=1234)
    inputs, lengths, mel, lin = bench.synthetic_batch(hp, 32, 160, 1000, 1234)
    for _ in range(2):
        m.initialize(inputs, lengths, None, mel, lin)
    torch.cuda.synchronize()
    w = m._bufs["lstm_cluster_work_expl_fwd"]
    N, H = 32, hp.expand_lstm_units
    chains = 2 * ((N + 15) // 16 + 1)
    off = 256 + 4096 + chains * 2 * 16 * (4 * H // 2) * 8
    tr = w.view(torch.uint8)[off:off + 512 * 8 * 8].view(torch.int64).view(512, 8).cpu().numpy().astype(np.float64) * 0.01   # us
    q0, q1 = 100, 500
    c0, c1, c2, p4, p5, sp = tr[q0:q1, 0], tr[q0:q1, 1], tr[q0:q1, 2], tr[q0:q1, 4], tr[q0:q1, 5], tr[q0:q1, 6] * 100
    print("slots %d..%d (two row groups per step): slot period %.2f us" % (q0, q1, (c0[-1] - c0[0]) / (q1 - q0 - 1)))
    print("compute: barrier -> cell math done %.2f us, -> publish issued %.2f us" % ((c1 - c0).mean(), (c2 - c1).mean()))
    print("poller: wait for the slot's operands %.2f us (%.1f polls), operands of slot q ready %.2f us after compute of slot q-1 started"
          % ((p5 - p4).mean(), sp.mean(), (p5[1:] - c0[:-1]).mean()))
    print("publish of slot q (c2) -> poller of slot q+2 (same row group, next step) has everything: %.2f us" % (p5[2:] - c2[:-2]).mean())
    print("poll done of slot q -> compute start of slot q: %.2f us" % (c0 - p5).mean())


if __name__ == "__main__":
    main()
