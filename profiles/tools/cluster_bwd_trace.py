#!/usr/bin/env python3
"""In-kernel slot timings of lstm_cluster2p_bwd_kernel (the partial-sum backward BiLSTM) at the benchmark shape's expand
BiLSTM: NS_CLUSTER_DBG=16 makes workgroup 0's compute wave 0 take 100 MHz stamps.
Per slot: [0] slot start (straight behind the publish of the slot before), [4] the peers' blocks of the step before
have arrived ([6] poll passes beyond the first), [1] cell update done + operand image written, [2] behind the slot's
barrier, [3] MFMAs issued and done, [7] partial sums published."""
import os
import sys

os.environ["NS_CLUSTER_DBG"] = os.environ.get("NS_CLUSTER_DBG", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd.models import create_model  # noqa: E402

hp = hparams_mod.load("taco2")
m = create_model("taco2", hp, device="cuda:0", dtype="mixed", seed=1234)
inputs, lengths, mel, lin = bench.synthetic_batch(hp, 32, 160, 1000, 1234)
for _ in range(2):
    m.initialize(inputs, lengths, None, mel, lin)
    m.backward()
torch.cuda.synchronize()
w = m._bufs["lstm_cluster_work_%s_bwd" % os.environ.get("NS_TRACE_TAG", "expl")]
N, H = 32, (hp.expand_lstm_units if os.environ.get("NS_TRACE_TAG", "expl") == "expl" else hp.encoder_lstm_units)
chains = 2 * ((N + 15) // 16 + 1)
off = 256 + 4096 + chains * 2 * 16 * (4 * H // 2) * 8
tr = w.view(torch.uint8)[off:off + 512 * 8 * 8].view(torch.int64).view(512, 8).cpu().numpy().astype(np.float64) * 0.01   # us
q0, q1 = (100, 500) if os.environ.get("NS_TRACE_TAG", "expl") == "expl" else (20, 150)
c = [tr[q0:q1, i] for i in range(8)]
print("slots %d..%d: slot period %.2f us" % (q0, q1, (c[0][-1] - c[0][0]) / (q1 - q0 - 1)))
print("compute wave: start -> peers' sums in %.2f (%.1f extra poll passes) | -> cell update + image %.2f | barrier %.2f | MFMA %.2f | publish %.2f | -> next slot start %.2f"
      % ((c[4] - c[0]).mean(), (c[6] * 100).mean(), (c[1] - c[4]).mean(), (c[2] - c[1]).mean(), (c[3] - c[2]).mean(), (c[7] - c[3]).mean(),
         (c[0][1:] - c[7][:-1]).mean()))
print("publish(q-1) -> peers' sums in(q): %.2f us" % (c[4][1:] - c[7][:-1]).mean())
