"""Measures (does not assert) the distance between the GPU training pass and the float64 oracle at the SHIPPED layer
widths, per precision mode: the numbers behind the bounds of tests/test_taco2_fullwidth_gpu.py.
    python profiles/tools/parity_probe.py > profiles/r04_parity_fullwidth.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from util import make_batch, oracle_report, stabilise_targets  # noqa: E402
from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd.models import create_model  # noqa: E402

hp = hparams_mod.load("taco2")
shapes = [(2, 24, 40), (4, 32, 50), (32, 24, 25), (3, 40, 60)]
modes = [a for a in sys.argv[1:] if a != "--full"] or ["fp32", "bf16x3", "mixed", "bf16"]
if "--full" in sys.argv:        # the benchmarked launch at its own lengths (mixed): several minutes of host time
    shapes, modes = [(32, 160, 1000)], ["mixed"]
for mode in modes:
    for (N, Ti, To) in shapes:
        m = create_model("taco2", hp, device="cuda:0", dtype=mode, seed=5)
        if To >= 500:
            inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=52)
            rep = oracle_report(m, hp, inputs, lengths, mel, lin, stabilise=2e-3)
        else:
            inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=N + 20)
            mel, lin = stabilise_targets(hp, m.numpy_params(), m.numpy_stats(), inputs, lengths, mel, lin)
            rep = oracle_report(m, hp, inputs, lengths, mel, lin)
        print("== mode %s  N %d T_in %d T_out %d  flips %d  paths %s" % (mode, N, Ti, To, rep["flips"], rep["paths"]))
        print("   ReLU branch differences per family (count, elements, largest |oracle pre-activation| / site rms): %s" % (
            {k: (v[0], v[1], float("%.2e" % v[2])) for k, v in rep["flip_families"].items()}))
        print("   loss got/want %.6f %.6f   mel %.6f %.6f   lin %.6f %.6f" % (rep["loss"] + rep["mel_loss"] + rep["linear_loss"]))
        for k, v in rep["out"].items():
            print("   out  %-18s relL2 %.2e  relmax %.2e  L1 %.2e" % ((k,) + v))
        worst = sorted(rep["grad"].items(), key=lambda kv: -kv[1][0])
        print("   grad worst relL2: " + ", ".join("%s %.2e" % (k.split("inference/")[-1], v[0]) for k, v in worst[:6]))
        print("   grad relL2 median %.2e  max %.2e ; relmax median %.2e max %.2e" % (
            np.median([v[0] for v in rep["grad"].values()]), max(v[0] for v in rep["grad"].values()),
            np.median([v[1] for v in rep["grad"].values()]), max(v[1] for v in rep["grad"].values())))
        for k, v in worst:
            print("      %-60s relL2 %.2e relmax %.2e" % (k, v[0], v[1]))
        sys.stdout.flush()
        del m
