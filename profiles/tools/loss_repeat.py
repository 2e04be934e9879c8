#!/usr/bin/env python3
"""Loss of the first training steps at the benchmark shape, two fresh models per mode: run-to-run spread."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd.models import create_model  # noqa: E402

hp = hparams_mod.load("taco2")
inputs, lengths, mel, lin = bench.synthetic_batch(hp, 32, 160, 1000, 1234)
for mode in sys.argv[1:] or ["mixed"]:
    for rep in range(2):
        m = create_model("taco2", hp, device="cuda:0", dtype=mode, seed=1234)
        m.add_optimizer(0)
        m.initialize(inputs, lengths, None, mel, lin)
        out = []
        for _ in range(9):
            m.forward_train()
            m.backward()
            m.apply_gradients()
            out.append(m.read_losses())
        print(mode, rep, " ".join("%.5f" % (x if isinstance(x, float) else x[0]) for x in out))
        del m
        torch.cuda.empty_cache()
