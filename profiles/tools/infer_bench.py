#!/usr/bin/env python3
"""Free-running synthesis alone (batch 1 by default, 300 decoder steps, shipped widths), for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 profiles/tools/infer_bench.py [N] [mode]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd.models import create_model  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = sys.argv[2] if len(sys.argv) > 2 else "mixed"
hp = hparams_mod.load("taco2")
hp.max_iters = 300
m = create_model("taco2", hp, device="cuda:0", dtype=mode, seed=1234)
m.use_graph = os.environ.get("NS_INFER_GRAPH", "1") != "0"
inputs, lengths, _, _ = bench.synthetic_batch(hp, N, 160, 10, 1234)
for _ in range(3):
    m.initialize(inputs, lengths)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    m.initialize(inputs, lengths)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("batch %d, %s: %.3f ms per synthesis pass (300 decoder steps), decode path %s" % (N, mode, dt * 1e3, m.last_paths.get("decode")))
