"""Griffin-Lim alone for rocprofv3: bench.py's griffin_lim leg (one 10 s clip x 7 calls, the batch of 32 x 4 calls, 60
iterations each) and nothing else."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from nspeech_amd import hparams as hparams_mod  # noqa: E402

print(json.dumps(bench.griffin_lim_bench(hparams_mod.load("taco2"), False)))
