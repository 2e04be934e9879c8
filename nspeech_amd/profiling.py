"""Live roofline measurement for bench.py: wraps every ns_gemm call of one training step in
HIP events (torch.cuda.Event on the stream the kernels are launched on) and reports the
achieved bf16-MFMA rate of the GEMM family against the dense peak."""
import torch

from . import ops

MFMA_BF16_PEAK_TFLOPS = 2500.0


def _last_kernel():
    from . import _lib as L
    fn = L.lib().ns_gemm_last_kernel
    fn.restype = L.C.c_char_p
    return fn().decode()


PROFILE_ROUND = "r05"


def pmc_traffic(key):
    """Memory-side bytes per launch from THIS round's PMC passes (profiles/<round>_pmc_traffic.json: FETCH_SIZE x2 +
    WRITE_SIZE as MI355X_MICROARCH.md prescribes, collected offline - rocprofv3 refuses --pmc together with the trace
    domains a live run would need).  None when the round has no such file: a stale number is worse than none."""
    import json
    import os
    f = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "%s_pmc_traffic.json" % PROFILE_ROUND)
    try:
        return float(json.load(open(f))[key]["traffic_bytes_per_launch"])
    except Exception:
        return None


def pmc_traffic_source():
    return "profiles/%s_pmc_traffic.json" % PROFILE_ROUND


def roofline(model, one_step):
    rec = []
    orig = ops.gemm

    def timed(A, B, Cm, M, N, K, *a, **kw):
        big = M > 32 and (A.dtype == torch.bfloat16 or ops.F32_PASSES > 0)
        if not big:
            return orig(A, B, Cm, M, N, K, *a, **kw)
        passes = 1 if A.dtype == torch.bfloat16 else max(1, ops.F32_PASSES)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(A, B, Cm, M, N, K, *a, **kw)
        e1.record()
        esz = A.element_size()
        nb = kw.get("batch", 1)
        # unique bytes: a conv1d 'same' product reads its [rows, C_in] input once although the im2col view is K = k * C_in
        # wide (lda < K: overlapping rows)
        lda = a[0] if a else kw.get("lda", K)
        a_unique = float(M + (K // lda - 1 if lda and K > lda else 0)) * min(lda, K) if kw.get("a_mode", 0) == 0 else float(M * K)
        rec.append((2.0 * M * N * K * nb, e0, e1, passes,
                    (a_unique + float(K * N)) * esz * nb + float(M * N) * Cm.element_size() * nb, _last_kernel()))

    # every product is timed running alone: the training step proper runs some weight gradients on a second stream
    # beside the encoder BiLSTM (Tacotron2.overlap_wgrads), where a launch's duration is not the kernel's own
    overlap, model.overlap_wgrads = getattr(model, "overlap_wgrads", False), False
    ops.gemm = timed
    try:
        one_step()
        torch.cuda.synchronize()
    finally:
        ops.gemm = orig
        model.overlap_wgrads = overlap
    if not rec:
        return None
    flops = sum(r[0] for r in rec)                 # algorithmic (one product per multiply-add)
    issued = sum(r[0] * r[3] for r in rec)         # MFMA work actually issued (x3 for split-bf16)
    ms = sum(r[1].elapsed_time(r[2]) for r in rec)
    ach = flops / (ms * 1e-3) / 1e12
    traffic = pmc_traffic("gemm_family")
    # the same timings per kernel variant, named as rocprofv3 names them (profiles/r01_v6_kernel_stats.csv)
    by = {}
    for r in rec:
        b = by.setdefault(r[5], [0, 0.0, 0.0, 0.0])
        b[0] += 1
        b[1] += r[1].elapsed_time(r[2])
        b[2] += r[0]
        b[3] += r[0] * r[3]
    by_kernel = [{"kernel": k, "launches": v[0], "avg_launch_us": v[1] * 1e3 / v[0], "ms_per_step": v[1],
                  "achieved_TFLOPs": v[2] / (v[1] * 1e-3) / 1e12, "issued_mfma_TFLOPs": v[3] / (v[1] * 1e-3) / 1e12}
                 for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])]
    return {"bound": "mfma", "kernel": "ns_gemm kernels (conv1d / dense / LSTM input + all weight and data gradients); "
                                       "dominant: %s" % by_kernel[0]["kernel"],
            "by_kernel": by_kernel,
            "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS,
            "launches": len(rec), "avg_launch_us": ms * 1e3 / len(rec), "gflop_per_step": flops / 1e9,
            "issued_mfma_tflops": issued / (ms * 1e-3) / 1e12,
            "algorithmic_bytes_per_launch": sum(r[4] for r in rec) / len(rec),
            "traffic": traffic,
            "traffic_note": "bytes per launch at the L2's fabric side (FETCH_SIZE x2 + WRITE_SIZE, Infinity-Cache hits "
                            "included), rocprofv3 --pmc passes summarised in profiles/%s_pmc_traffic.json; null = not "
                            "collected this round" % PROFILE_ROUND}
