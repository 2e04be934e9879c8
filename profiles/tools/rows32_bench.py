#!/usr/bin/env python3
"""ns_rows32 alone (csrc/rows32.hip): the LSTMBlockCell step of batched free-running synthesis on packed weights, 100 calls
captured in a HIP graph (the way the synthesis pass issues them; a Python call alone costs ~10 us of host time), operand as
packed rows and as fp32 rows.  Usage: python profiles/tools/rows32_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nspeech_amd import ops  # noqa: E402

rng = np.random.default_rng(0)


def R(*s):
    return torch.from_numpy(rng.standard_normal(s).astype(np.float32)).cuda()


def bench(N, K, H, passes, packed_rows, reps=100, nbuf=2):
    a = R(32, K)
    rows = ops.rows32_rows(K, "cuda")
    ops.rows32_pack_rows(a, K, N, K, rows, K, 0)
    pk = [ops.rows32_pack(R(K, 4 * H) * 0.02, K, 4 * H, cell_units=H) for _ in range(nbuf)]      # alternate: no reuse from L2
    bias = R(4 * H)
    h = torch.zeros(32, H, device="cuda")
    c = torch.zeros(32, H, device="cuda")

    def run(i):
        if packed_rows:
            ops.rows32(None, 0, pk[i % nbuf], N, K, 4 * H, h, H, bias=bias, cell_units=H, c_out=c, co_sn=H, f32_passes=passes,
                       a_rows=(rows, K, 0))
        else:
            ops.rows32(a, K, pk[i % nbuf], N, K, 4 * H, h, H, bias=bias, cell_units=H, c_out=c, co_sn=H, f32_passes=passes)
    for i in range(4):
        run(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            run(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    mb = K * 4 * H * (4 if passes == 3 else 2) / 1e6
    print("rows %2d K %4d H %4d passes %d %-11s: %6.2f us per call, %5.2f TB/s of weights" % (
        N, K, H, passes, "packed rows" if packed_rows else "fp32 rows", us, mb / us))


for pr in (True, False):
    for N in (32, 16):
        for passes in (3, 1):
            bench(N, 1792, 1024, passes, pr)
    bench(32, 2048, 1024, 3, pr)
    bench(32, 384, 256, 3, pr)
