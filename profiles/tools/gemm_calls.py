#!/usr/bin/env python3
"""One line per ns_gemm call of a Tacotron-2 training step at the benchmark shape: shape, operand modes, the kernel
that ran, its HIP-event time and rate.  Usage: python profiles/tools/gemm_calls.py [mixed|bf16|fp32] [another build of
the library, for an A/B in one box]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from nspeech_amd import _lib, hparams as hparams_mod, ops, profiling  # noqa: E402

if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from nspeech_amd.models import create_model  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "mixed"
    hp = hparams_mod.load("taco2")
    m = create_model("taco2", hp, device="cuda:0", dtype=mode, seed=1234)
    inputs, lengths, mel, lin = bench.synthetic_batch(hp, 32, 160, 1000, 1234)
    m.add_optimizer(0)
    m.overlap_wgrads = False        # every product timed running alone

    def step():
        m.initialize(inputs, lengths, None, mel, lin)
        m.backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    rec = []
    orig = ops.gemm

    def timed(A, B, Cm, M, N, K, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(A, B, Cm, M, N, K, *a, **kw)
        e1.record()
        rec.append((M, N, K, kw.get("a_mode", 0), kw.get("b_mode", 0), kw.get("batch", 1), kw.get("split_k", 1),
                    str(A.dtype)[6:], str(Cm.dtype)[6:], kw.get("accumulate", 0), profiling._last_kernel(), e0, e1,
                    m._tick_name if hasattr(m, "_tick_name") else ""))
    ops.gemm = timed
    try:
        step()
        torch.cuda.synchronize()
    finally:
        ops.gemm = orig
    tot = 0.0
    for r in rec:
        ms = r[11].elapsed_time(r[12])
        tot += ms
        fl = 2.0 * r[0] * r[1] * r[2] * r[5]
        print("M %6d N %5d K %6d a%d b%d batch %3d splitk %2d %-8s->%-8s acc %d  %-30s %8.1f us %7.1f TF/s" % (
            r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], ms * 1e3, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0))
    print("total %.3f ms over %d calls" % (tot, len(rec)))


if __name__ == "__main__":
    main()
