#!/usr/bin/env python3
"""gemm_x256_kernel<1> / <3> alone: throughput on uniform random operands and a race screen (the kernel's LDS hand-offs
are ordered by counted waits and barriers only, so a schedule edit is screened over many launches at several sizes
against torch.matmul - every launch checked, not the first).
Usage: python profiles/tools/x256_bench.py [reps] [other build of the library, e.g. one compiled with -DNS_X256_EARLY_WAIT]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from nspeech_amd import _lib, ops, profiling  # noqa: E402

if len(sys.argv) > 2:          # A/B: the same tool over another build of the library (never the product's loader path)
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])

SHAPES = [("4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192), ("conv data gradient", 32124, 512, 2560),
          ("decoder input product", 6432, 4096, 1024), ("K = 128", 4096, 2048, 128), ("K = 192, ragged M", 6404, 1024, 192),
          ("K = 320, ragged M, N", 5000, 1152, 320)]


def run(label, M, N, K, nseg, reps, dev):
    g = torch.Generator().manual_seed(M + K)
    A = torch.rand(M, K, generator=g) * 2 - 1
    B = torch.rand(N, K, generator=g) * 2 - 1
    if nseg == 1:
        Ah, Bh = A.to(torch.bfloat16), B.to(torch.bfloat16)
        ref = (Ah.to(dev).float() @ Bh.to(dev).float().t())
        args = dict()
        Ad, Bd = Ah.to(dev), Bh.to(dev)
        tol = 2e-6 * K ** 0.5
    else:
        Ah, Bh = A.to(torch.bfloat16), B.to(torch.bfloat16)
        Al, Bl = (A - Ah.float()).to(torch.bfloat16), (B - Bh.float()).to(torch.bfloat16)
        ref = (A.to(dev).double() @ B.to(dev).double().t()).float()
        Ad, Bd = Ah.to(dev), Bh.to(dev)
        args = dict(a_lo=Al.to(dev), b_lo=Bl.to(dev), f32_passes=3)
        tol = 3e-5
    C = torch.zeros(M, N, device=dev)
    scale = float(ref.abs().max())
    worst = 0.0
    bad = 0
    for i in range(reps):
        C.fill_(7.0)
        ops.gemm(Ad, Bd, C, M, N, K, K, K, N, **args)
        err = float((C - ref).abs().max()) / scale
        worst = max(worst, err)
        bad += err > tol
    kern = profiling._last_kernel() if hasattr(profiling, "_last_kernel") else ""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    ops.gemm(Ad, Bd, C, M, N, K, K, K, N, **args)
    e0.record()
    for _ in range(n):
        ops.gemm(Ad, Bd, C, M, N, K, K, K, N, **args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print("%-24s nseg %d M %6d N %5d K %5d %-22s %9.1f us %7.1f TF/s algorithmic  worst rel err %.1e  wrong launches %d / %d" % (
        label, nseg, M, N, K, kern, us, 2.0 * M * N * K / us / 1e6, worst, bad, reps), flush=True)
    return bad


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = "cuda:0"
    bad = 0
    for nseg in (1, 3):
        for label, M, N, K in SHAPES:
            if nseg == 3 and M * N > 5e7:
                continue
            bad += run(label, M, N, K, nseg, reps, dev)
    print("wrong launches in all:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
