"""Thin tensor-level wrappers over the C ABI (include/nspeech_hip.h).

Every function takes torch CUDA tensors purely as (device pointer, dtype) carriers and
enqueues HIP kernels on torch's current stream.  No torch math happens here.
"""
import ctypes as C

import torch

from . import _lib as L
from ._lib import NS_BF16, NS_F32


def stream():
    return torch.cuda.current_stream().cuda_stream


def dt(t):
    if t.dtype == torch.float32:
        return NS_F32
    if t.dtype == torch.bfloat16:
        return NS_BF16
    raise TypeError("unsupported dtype %s" % t.dtype)


def ptr(t, off=0):
    """Device address of element `off` (in elements) of tensor `t`; None -> NULL."""
    if t is None:
        return None
    assert t.is_cuda, "hot-path tensors must live on the GPU"
    return t.data_ptr() + off * t.element_size()


def gemm(A, B, Cm, M, N, K, lda, ldb, ldc, a_mode=0, b_mode=0, a_off=0, b_off=0, c_off=0,
         bias=None, bias_off=0, act=0, alpha=1.0, accumulate=0, row_mask=None, col_sum=None,
         col_sumsq=None, split_k=1, b_seg=None):
    """C = act(alpha * A.B + bias); see ns_gemm in include/nspeech_hip.h."""
    assert A.dtype == B.dtype
    p = L.GemmParams()
    p.dtype = dt(A)
    p.M, p.N, p.K = M, N, K
    p.A, p.lda, p.a_mode = ptr(A, a_off), lda, a_mode
    p.B, p.ldb, p.b_mode = ptr(B, b_off), ldb, b_mode
    if b_seg is not None:
        p.b_seg_len, p.b_seg_stride = b_seg
    p.C, p.ldc, p.c_dtype = ptr(Cm, c_off), ldc, dt(Cm)
    p.accumulate = accumulate
    p.bias = ptr(bias, bias_off)
    p.act = act
    p.alpha = alpha
    if row_mask is not None:
        p.row_period, p.row_lo, p.row_hi, p.row_shift = row_mask
    p.col_sum = ptr(col_sum)
    p.col_sumsq = ptr(col_sumsq)
    p.split_k = split_k
    L.call("ns_gemm", p, stream())
