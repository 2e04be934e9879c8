#!/usr/bin/env python3
"""Griffin-Lim alone (bench.griffin_lim_bench without the CPU leg): 10 s clip and the batch of 32."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from nspeech_amd import hparams as hparams_mod  # noqa: E402

hp = hparams_mod.load("taco2")
hparams_mod.set_hparams(hp) if hasattr(hparams_mod, "set_hparams") else None
print(json.dumps(bench.griffin_lim_bench(hp, False), indent=1))
