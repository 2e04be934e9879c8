#!/usr/bin/env python3
"""The 128 x 128-tile ns_gemm kernels alone at shapes of the benchmark step (random operands, 30 launches back to back
between two HIP events, result checked against torch.matmul on the first launch).
Usage: python profiles/tools/gemm_shapes.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from nspeech_amd import ops, profiling  # noqa: E402

# (label, M, N, K, a_mode, b_mode, dtype, c dtype, f32_passes, split_k)
SHAPES = [
    ("encoder conv data gradient", 5244, 512, 2560, 0, 0, "bf16", "f32", 0, 1),
    ("decoder input product bwd", 5248, 512, 1024, 0, 0, "bf16", "f32", 0, 1),
    ("postnet data gradient (x256 shape on 128 tiles)", 32124, 512, 2560, 0, 0, "bf16", "f32", 0, 1),
    ("conv weight gradient", 512, 2560, 32124, 1, 1, "bf16", "f32", 0, 16),
    ("encoder conv weight gradient", 512, 2560, 5244, 1, 1, "bf16", "f32", 0, 8),
    ("lstm input weight gradient", 1024, 4096, 6432, 1, 1, "bf16", "f32", 0, 4),
    ("encoder conv forward, 3 passes", 5244, 512, 2560, 0, 1, "f32", "f32", 3, 1),
    ("attention input product, 3 passes", 5248, 1024, 512, 0, 1, "f32", "f32", 3, 1),
    ("expand dense, 3 passes", 32124, 512, 400, 0, 1, "f32", "f32", 3, 1),
    ("fp32 weight gradient, 1 pass", 1024, 400, 6432, 1, 1, "f32", "f32", 1, 12),
    ("fp32 data gradient, 1 pass", 6432, 1024, 400, 0, 0, "f32", "f32", 1, 1),
    # (the same shape again: round 3's log showed 2 356 us for the LAST row - VERDICT r3 weak #15; if that was the row
    # and not its position, both rows show it)
    ("fp32 data gradient, 1 pass (again)", 6432, 1024, 400, 0, 0, "f32", "f32", 1, 1),
    ("fp32 weight gradient, 1 pass (again)", 1024, 400, 6432, 1, 1, "f32", "f32", 1, 12),
]


def main():
    dev = "cuda:0"
    g = torch.Generator().manual_seed(1)
    for label, M, N, K, am, bm, dtn, cdn, passes, sk in SHAPES:
        D = torch.bfloat16 if dtn == "bf16" else torch.float32
        A = (torch.randn(M, K, generator=g) * 0.5).to(D)
        B = (torch.randn(N, K, generator=g) * 0.5).to(D)
        ref = A.double() @ B.double().t()
        Ad = (A.t().contiguous() if am else A).to(dev)          # a_mode 1: [K][M]
        Bd = (B.t().contiguous() if bm else B).to(dev)          # b_mode 1: [K][N]
        C = torch.zeros(M, N, device=dev)
        lda = M if am else K
        ldb = N if bm else K
        acc = 2 if sk > 1 else 0
        ops.gemm(Ad, Bd, C, M, N, K, lda, ldb, N, a_mode=am, b_mode=bm, f32_passes=passes, split_k=sk, accumulate=acc)
        torch.cuda.synchronize()
        kern = profiling._last_kernel() if hasattr(profiling, "_last_kernel") else ""
        err = (C.double().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-9)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps):
            ops.gemm(Ad, Bd, C, M, N, K, lda, ldb, N, a_mode=am, b_mode=bm, f32_passes=passes, split_k=sk, accumulate=acc)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("%-50s M %6d N %5d K %6d a%d b%d %-4s p%d sk %2d  %-32s %8.1f us %7.1f TF/s  rel err %.1e" % (
            label, M, N, K, am, bm, dtn, passes, sk, kern, us, 2.0 * M * N * K / us / 1e6, err), flush=True)


if __name__ == "__main__":
    main()
