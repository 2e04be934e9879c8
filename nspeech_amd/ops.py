"""Thin tensor-level wrappers over the C ABI (include/nspeech_hip.h).

Every function takes torch CUDA tensors purely as (device pointer, dtype) carriers and
enqueues HIP kernels on torch's current stream.  No torch math happens here.
"""
import ctypes as C
import ctypes as C_          # (the name C is also a parameter of colsum)

import torch

from . import _lib as L
from ._lib import NS_BF16, NS_F32


# fp32-operand GEMM precision (see ns_gemm_params.f32_passes); models set it per call
F32_PASSES = 0
# LSTMBlockCell's cell_clip for every LSTM parameter block built below (hparam lstm_cell_clip; 0 = off = the reference's
# cells); the model sets it at the top of a pass, like F32_PASSES
CELL_CLIP = 0.0


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
         col_sumsq=None, split_k=1, b_seg=None, addend=None, addend_off=0, ld_add=0, f32_passes=None,
         gate=None, gate_off=0, ld_gate=0, a_lo=None, b_lo=None, batch=1, batch_strides=(0, 0, 0),
         stat_z=None, ld_stat_z=0, stat_mean=None, stat_istd=None, stat_z_off=0):
    """C = act(alpha * A.B + bias); see ns_gemm in include/nspeech_hip.h.
    stat_z (+ ld_stat_z, stat_mean, stat_istd): col_sum / col_sumsq become the BatchNorm-backward sums of C as `dy`
    against the saved BatchNorm input stat_z (sum dy, sum dy * xhat); stat_z_off: element offset of the row that pairs
    with output row 0 (as c_off for C)."""
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
    if col_sum is not None:      # scratch of the deterministic two-stage statistics (one per device and stream)
        p.stat_part = ptr(_stat_part(col_sum.device, M, N))
    if stat_z is not None:
        p.stat_z, p.ld_stat_z, p.stat_z_dtype = ptr(stat_z, stat_z_off), ld_stat_z, dt(stat_z)
        p.stat_mean, p.stat_istd = ptr(stat_mean), ptr(stat_istd)
    p.split_k = split_k
    if addend is not None:
        p.addend, p.ld_add, p.addend_dtype = ptr(addend, addend_off), ld_add, dt(addend)
    if gate is not None:
        p.gate, p.ld_gate = ptr(gate, gate_off), ld_gate
    p.f32_passes = F32_PASSES if f32_passes is None else f32_passes
    if batch > 1:
        p.batch = batch
        p.batch_stride_a, p.batch_stride_b, p.batch_stride_c = batch_strides
    if a_lo is not None:        # pre-split fp32 values: A / B hold the high parts (same offsets and strides)
        p.A_lo, p.B_lo = ptr(a_lo, a_off), ptr(b_lo, b_off)
    if split_k > 1 and batch == 1 and DETERMINISTIC_SPLITK:
        work, count = _splitk_scratch(Cm.device, M, N, split_k)
        p.splitk_work, p.splitk_count = ptr(work), ptr(count)
    L.call("ns_gemm", p, stream())


# Split-K products can add their k slices in a FIXED order (ns_gemm_params.splitk_work): with every other sum of the
# backward pass already ordered, two runs of a training step then give the same gradient bits.  It costs the partial tiles'
# trip through memory (measured at the benchmark shape: +14 us per weight-gradient product, +0.4 ms = 2 % of the step), so
# it is a switch: models set it from hparams.deterministic_gradients (default off), NS_DETERMINISTIC=1 / 0 forces it.
DETERMINISTIC_SPLITK = __import__("os").environ.get("NS_DETERMINISTIC", "0") == "1"
_SPLITK = {}


def _splitk_scratch(device, M, N, split_k):
    """(partial-tile scratch, zeroed tile counters) of the deterministic split-K, one pair per (device, stream): a
    product and the next one on the same stream run back to back, products on different streams must not share."""
    lib = L.lib()
    fb, fc = lib.ns_gemm_splitk_work_bytes, lib.ns_gemm_splitk_counters
    fb.restype = fc.restype = L.C.c_size_t
    need, cnt = int(fb(int(M), int(N), int(split_k))), int(fc(int(M), int(N)))
    key = (device, stream())
    cur = _SPLITK.get(key)
    if cur is None or cur[0].numel() * 4 < need or cur[1].numel() < cnt:
        w = torch.empty(max(need // 4, cur[0].numel() if cur else 0), dtype=torch.float32, device=device)
        c = torch.zeros(max(cnt, cur[1].numel() if cur else 0, 4096), dtype=torch.int32, device=device)
        cur = _SPLITK[key] = (w, c)
    return cur


_STAT_PART = {}


def _stat_part(device, M, N):
    """Partial-sum scratch of the GEMM statistics.  One per (device, stream): the producing product and its finalize
    kernel run back to back on one stream, products on different streams (the weight-gradient stream, a feeder thread,
    a second model) must not share it."""
    fn = L.lib().ns_gemm_stat_part_floats
    fn.restype = L.C.c_size_t
    need = int(fn(int(M), int(N)))
    key = (device, stream())
    buf = _STAT_PART.get(key)
    if buf is None or buf.numel() < need:
        buf = _STAT_PART[key] = torch.empty(need, dtype=torch.float32, device=device)
    return buf


def _fill(_st, **kw):
    for k, v in kw.items():
        setattr(_st, k, v)
    return _st


def copy3d(src, dst, I, J, Cc, src_strides, dst_strides, src_off=0, dst_off=0, accumulate=0):
    p = L.struct("ns_copy3d_params")
    _fill(p, src=ptr(src, src_off), src_dtype=dt(src), src_si=src_strides[0], src_sj=src_strides[1],
          dst=ptr(dst, dst_off), dst_dtype=dt(dst), dst_si=dst_strides[0], dst_sj=dst_strides[1],
          I=I, J=J, Cc=Cc, accumulate=accumulate)
    L.call("ns_copy3d", p, stream())


def embedding_fwd(ids, table, out, N, T, P, padl, D, V, table_off=0):
    p = L.struct("ns_embedding_params")
    _fill(p, ids=ptr(ids), table=ptr(table, table_off), out=ptr(out), out_dtype=dt(out), N=N, T=T, P=P,
          padl=padl, D=D, V=V)
    L.call("ns_embedding_fwd", p, stream())


_EMB_WORK = {}


def embedding_bwd(ids, dout, dtable, N, T, P, padl, D, V, dtable_off=0):
    p = L.struct("ns_embedding_bwd_params")
    fn = L.lib().ns_embedding_bwd_work_floats
    fn.restype = C.c_size_t
    work = _scratch(_EMB_WORK, dout.device, int(fn(int(N), int(D), int(V))))
    _fill(p, ids=ptr(ids), dout=ptr(dout), dtable=ptr(dtable, dtable_off), N=N, T=T, P=P, padl=padl, D=D, V=V, work=ptr(work))
    L.call("ns_embedding_bwd", p, stream())


def bn_fwd(z, y, rows, C, col_sum, col_sumsq, count, gamma, beta, moving_mean, moving_var, mean_out, istd_out,
           training, row_mask=None, eps=1e-3, momentum=0.99, gamma_off=0, beta_off=0, mm_off=0, mv_off=0,
           y_hi=None, y_lo=None, y_off=0, ld_y=0):
    """y_off / ld_y: y as a column block of a wider [rows, ld_y] array (element offset of its first column)."""
    p = L.struct("ns_bn_fwd_params")
    if y_hi is not None:
        p.y_hi, p.y_lo = ptr(y_hi), ptr(y_lo)
    p.ld_y = ld_y
    _fill(p, z=ptr(z), y=ptr(y, y_off), dtype=dt(z), rows=rows, C=C, col_sum=ptr(col_sum), col_sumsq=ptr(col_sumsq),
          count=float(count), gamma=ptr(gamma, gamma_off), beta=ptr(beta, beta_off),
          moving_mean=ptr(moving_mean, mm_off), moving_var=ptr(moving_var, mv_off),
          mean_out=ptr(mean_out), istd_out=ptr(istd_out), eps=eps, momentum=momentum, training=int(training))
    if row_mask is not None:
        p.row_period, p.row_lo, p.row_hi = row_mask
    L.call("ns_bn_fwd", p, stream())


def bn_bwd(dy, z, dpre, rows, C, mean, istd, gamma, dgamma, dbeta, dbias, work, count, act, row_mask=None,
           gamma_off=0, dgamma_off=0, dbeta_off=0, dbias_off=0, sums=None, dy_off=0, ld_dy=0):
    """sums = (sum dy, sum dy * xhat) per column as left by the product that formed dy (gemm(..., stat_z=...)).
    dy_off / ld_dy: dy as a column block of a wider [rows, ld_dy] gradient."""
    p = L.struct("ns_bn_bwd_params")
    p.ld_dy = ld_dy
    if sums is not None:
        p.sum_dy, p.sum_dyxh = ptr(sums[0]), ptr(sums[1])
    assert work.numel() >= 200 * C, "ns_bn_bwd: work needs 200 * C floats"
    _fill(p, dy=ptr(dy, dy_off), z=ptr(z), dpre=ptr(dpre), dtype=dt(z), rows=rows, C=C, mean=ptr(mean), istd=ptr(istd),
          gamma=ptr(gamma, gamma_off), dgamma=ptr(dgamma, dgamma_off), dbeta=ptr(dbeta, dbeta_off),
          dbias=ptr(dbias, dbias_off), work=ptr(work), count=float(count), act=act)
    if dpre.dtype != z.dtype:
        p.dpre_dtype = dt(dpre)
    if row_mask is not None:
        p.row_period, p.row_lo, p.row_hi = row_mask
    L.call("ns_bn_bwd", p, stream())


_COLSUM_WORK = {}


def _scratch(store, key_extra, floats, zero_head=0):
    """Grow-only fp32 scratch per (device, stream); the first `zero_head` words are zero when it is handed out first (the
    kernels that count arrivals there leave them zero)."""
    key = (key_extra, stream())
    buf = store.get(key)
    if buf is None or buf.numel() < floats:
        buf = store[key] = torch.zeros(int(floats), dtype=torch.float32, device=key_extra)
    return buf


def colsum(x, ld, rows, C, out, x_off=0, out_off=0):
    """out[c] += column sums of x, added in a fixed order (ns_colsum_params.work): bias gradients are bit-reproducible."""
    p = L.struct("ns_colsum_params")
    fn = L.lib().ns_colsum_work_floats
    fn.restype = C_.c_size_t
    work = _scratch(_COLSUM_WORK, x.device, int(fn(int(C))))
    _fill(p, x=ptr(x, x_off), dtype=dt(x), ld=ld, rows=rows, C=C, out=ptr(out, out_off), work=ptr(work))
    L.call("ns_colsum", p, stream())


def l1_loss(pred, ldp, target, dpred, ldd, N, T, P, padl, F, n_prio, w_all, w_prio, loss_acc, acc_off=0):
    p = L.struct("ns_l1_loss_params")
    _fill(p, pred=ptr(pred), ldp=ldp, target=ptr(target), dpred=ptr(dpred),
          dpred_dtype=dt(dpred) if dpred is not None else 0, ldd=ldd, N=N, T=T, P=P, padl=padl, F=F,
          n_prio=n_prio, w_all=w_all, w_prio=w_prio, loss_acc=ptr(loss_acc, acc_off))
    L.call("ns_l1_loss", p, stream())


def sumsq(x, n, out, out_off=0, work=None):
    p = L.struct("ns_sumsq_params")
    _fill(p, x=ptr(x), n=n, out=ptr(out, out_off), work=ptr(work))
    L.call("ns_sumsq", p, stream())


def segment_stats(x, offsets, nseg, out):
    """out[4 s + (sum, sum of squares, min, max)] over x[offsets[s] : offsets[s + 1]]; offsets = device int64 tensor."""
    p = L.struct("ns_segment_stats_params")
    _fill(p, x=ptr(x), offsets=ptr(offsets), nseg=nseg, out=ptr(out))
    L.call("ns_segment_stats", p, stream())


def adam(pw, g, m, v, n, gnorm_sq, clip, grad_scale, lr_t, beta1, beta2, eps, shadow=None, status=(), skipped=None):
    """status: work buffers of this step's persistent recurrences (their first int is the status word); if one is
    non-zero the kernel updates nothing and sets skipped[0] = 1."""
    p = L.struct("ns_adam_params")
    _fill(p, p=ptr(pw), g=ptr(g), m=ptr(m), v=ptr(v), n=n, gnorm_sq=ptr(gnorm_sq), clip=clip,
          grad_scale=grad_scale, lr_t=lr_t, beta1=beta1, beta2=beta2, eps=eps, shadow_bf16=ptr(shadow), skipped=ptr(skipped))
    assert len(status) <= 12
    for i, w in enumerate(status):
        p.status[i] = ptr(w)
    L.call("ns_adam", p, stream())


def zero(t):
    """Clear a tensor with the library's fill kernel (16-byte granularity)."""
    nbytes = t.numel() * t.element_size()
    assert nbytes % 16 == 0 and t.data_ptr() % 16 == 0, "ops.zero: 16-byte granularity"
    L.check(L.lib().ns_zero(C.c_void_p(ptr(t)), C.c_size_t(nbytes), C.c_void_p(stream())), "ns_zero")


def streams_concurrent(a, b):
    """True when launches on torch streams a and b can run side by side (ns_streams_concurrent: they do not share a
    hardware queue).  Synchronises both."""
    work = torch.zeros(4, dtype=torch.int32, device=a.device)
    rc = L.lib().ns_streams_concurrent(C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream), C.c_void_p(ptr(work)))
    if rc < 0:
        L.check(rc, "ns_streams_concurrent")
    return rc == 1


def concurrent_streams(device, n=1, tries=16):
    """n new streams that run beside the CURRENT one and beside one another.  HIP hands its hardware queues to streams in
    turn; a stream that lands on another one's queue (every queue-count-th one of a process) serialises behind it and an
    overlap planned on the pair is silently lost - so candidates are probed against the current stream and against the
    ones already accepted, and the rejected ones are kept alive until the set is complete (a released queue slot would
    be dealt out again).  With fewer free queues than asked for, the last candidates fill the set."""
    cur = torch.cuda.current_stream(device)
    chosen, rejected = [], []
    for _ in range(tries):
        if len(chosen) == n:
            break
        s = torch.cuda.Stream(device=device)
        if streams_concurrent(cur, s) and all(streams_concurrent(c, s) for c in chosen):
            chosen.append(s)
        else:
            rejected.append(s)
    while len(chosen) < n:
        chosen.append(rejected.pop() if rejected else torch.cuda.Stream(device=device))
    return chosen


def concurrent_stream(device, tries=8):
    return concurrent_streams(device, 1, tries)[0]


def zero_many(tensors):
    """Clear several tensors with one launch per 24 of them (ns_zero_many)."""
    ts = [t for t in tensors if t is not None and t.numel()]
    if not ts:
        return
    for t in ts:
        assert (t.numel() * t.element_size()) % 16 == 0 and t.data_ptr() % 16 == 0, "ops.zero_many: 16-byte granularity"
    n = len(ts)
    ptrs = (C.c_void_p * n)(*[ptr(t) for t in ts])
    nbytes = (C.c_size_t * n)(*[t.numel() * t.element_size() for t in ts])
    L.check(L.lib().ns_zero_many(ptrs, nbytes, n, C.c_void_p(stream())), "ns_zero_many")


CAST_RECORD = None        # a list: cast2d() appends its parameter block instead of launching (CastBatch)


def cast2d(src, rows, cols, ld_src, dst, ld_dst, transpose, src_off=0, dst_off=0, dst_hi=None, dst_lo=None):
    """dst may be None when the pre-split pair (dst_hi, dst_lo) is all that is wanted."""
    p = L.struct("ns_cast2d_params")
    _fill(p, src=ptr(src, src_off), rows=rows, cols=cols, ld_src=ld_src, dst=ptr(dst, dst_off) if dst is not None else None,
          dst_dtype=dt(dst) if dst is not None else NS_F32, ld_dst=ld_dst, transpose=int(transpose))
    if dst_hi is not None:
        p.dst_hi, p.dst_lo = ptr(dst_hi, dst_off), ptr(dst_lo, dst_off)
    if CAST_RECORD is not None and L.lib().ns_cast2d_batchable(C.byref(p)) > 0:
        CAST_RECORD.append(p)
        return
    L.call("ns_cast2d", p, stream())


class CastBatch(object):
    """A fixed set of cast2d calls replayed as ONE launch (ns_cast2d_batch).  record(fn): run fn() with cast2d()
    collecting its eligible parameter blocks (the others launch as usual); the table goes to the device once - every
    pointer in it must stay valid, which holds for a model's flat parameter buffer and its shadow copies."""

    def __init__(self, device):
        self.device = device
        self.table = self.ends = None
        self.n = self.total = 0

    def record(self, fn):
        global CAST_RECORD
        CAST_RECORD = rec = []
        try:
            fn()
        finally:
            CAST_RECORD = None
        if not rec:
            return self
        lib = L.lib()
        ends, tot = [], 0
        for p in rec:
            tot += int(lib.ns_cast2d_batchable(C.byref(p)))
            ends.append(tot)
        raw = b"".join(bytes(p) for p in rec)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self.ends = torch.tensor(ends, dtype=torch.int32, device=self.device)
        self.n, self.total = len(rec), tot
        return self

    def run(self):
        if self.n:
            L.check(L.lib().ns_cast2d_batch(C.c_void_p(self.table.data_ptr()), C.c_void_p(self.ends.data_ptr()), self.n,
                                            self.total, C.c_void_p(stream())), "ns_cast2d_batch")


def lstm_seq_params(N, T, H, P, padl, xg, ld_xg, whT, wh, lengths, reverse, h, ld_h, c, gates,
                    dh=None, ld_dh=0, dgates=None, work=None, xg_off=0, whT_off=0, wh_off=0, h_off=0, dh_off=0,
                    forget_bias=1.0, whT_hi=None, whT_lo=None, wh_bf16=None, wh_bf16_off=0, dgates_bf16=None,
                    h_bf16=None, h_bf16_off=0, ld_h_bf16=0, dtype=None, zoneout=None):
    """zoneout = (thr_cell, thr_output, seed_cell, seed_output) or None (plain cell), see zoneout_threshold()."""
    p = L.struct("ns_lstm_seq_params")
    if zoneout is not None:
        p.zoneout_thr_cell, p.zoneout_thr_output, p.zoneout_seed_cell, p.zoneout_seed_output = [int(x) for x in zoneout]
    if h_bf16 is not None:
        p.h_bf16, p.ld_h_bf16 = ptr(h_bf16, h_bf16_off), ld_h_bf16
    _fill(p, dtype=dt(h) if dtype is None else dtype, N=N, T=T, H=H, P=P, padl=padl, xg=ptr(xg, xg_off), ld_xg=ld_xg,
          whT=ptr(whT, whT_off), wh=ptr(wh, wh_off), lengths=ptr(lengths), reverse=int(reverse),
          forget_bias=forget_bias, h=ptr(h, h_off), ld_h=ld_h, c=ptr(c), gates=ptr(gates),
          dh=ptr(dh, dh_off), ld_dh=ld_dh, dgates=ptr(dgates), work=ptr(work), f32_passes=F32_PASSES,
          whT_hi=ptr(whT_hi), whT_lo=ptr(whT_lo), wh_bf16=ptr(wh_bf16, wh_bf16_off), dgates_bf16=ptr(dgates_bf16),
          cell_clip=float(CELL_CLIP))
    return p


def zoneout_threshold(rate):
    """A zoneout rate as the 24-bit threshold the kernels compare the mixed counter with (ns_lstm_seq_params)."""
    return int(float(rate) * 16777216.0)


def _fmix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def zoneout_seed(base, step, stream_id):
    """The 32-bit seed of one mask stream (layer, cell | output) of training step `step`: every step draws new masks,
    every data-parallel rank draws its own (the caller folds the rank into `base`)."""
    return _fmix32(_fmix32((base & 0xFFFFFFFF) ^ ((step * 0x9E3779B9) & 0xFFFFFFFF)) ^ ((stream_id * 0x7FEB352D) & 0xFFFFFFFF))


def lstm_seq(direction, dtype_t, *a, **kw):
    """direction: 'fwd' or 'bwd' (through-time gradient)."""
    p = lstm_seq_params(*a, **kw)
    L.call("ns_lstm_seq_fwd" if direction == "fwd" else "ns_lstm_seq_bwd", p, stream())


def lstm_wide_supported(p, backward):
    return bool(L.lib().ns_lstm_wide_supported(C.byref(p), int(backward)))


def lstm_wide_work_floats(p):
    fn = L.lib().ns_lstm_wide_work_bytes
    fn.restype = C.c_size_t
    return (fn(C.byref(p)) + 3) // 4


def lstm_wide(direction, p, wide_work):
    """Persistent whole-sequence recurrence for wide cells (one launch); wide_work[0] is the status word."""
    fn = getattr(L.lib(), "ns_lstm_wide_fwd" if direction == "fwd" else "ns_lstm_wide_bwd")
    L.check(fn(C.byref(p), C.c_void_p(ptr(wide_work)), C.c_void_p(stream())), "ns_lstm_wide_" + direction)


def lstm_seq_call(direction, p):
    L.call("ns_lstm_seq_fwd" if direction == "fwd" else "ns_lstm_seq_bwd", p, stream())


def lstm_seq2(direction, p0, p1):
    """Two independent recurrences (BiLSTM directions) advanced together, one launch per step."""
    fn = getattr(L.lib(), "ns_lstm_seq2_fwd" if direction == "fwd" else "ns_lstm_seq2_bwd")
    L.check(fn(C.byref(p0), C.byref(p1), C.c_void_p(stream())), "ns_lstm_seq2_" + direction)


def _attn_params(kw):
    p = L.struct("ns_taco2_attn_params")
    for k, v in kw.items():
        if v is None:
            continue
        if isinstance(v, tuple):        # (tensor, offset)
            v = ptr(v[0], v[1])
        elif hasattr(v, "data_ptr"):
            v = ptr(v)
        setattr(p, k, v)
    p.f32_passes = F32_PASSES
    p.cell_clip = float(CELL_CLIP)
    return p


def taco2_attn(direction, **kw):
    L.call("ns_taco2_attn_fwd" if direction == "fwd" else "ns_taco2_attn_bwd", _attn_params(kw), stream())


def taco2_attn_cluster_supported(**kw):
    return bool(L.lib().ns_taco2_attn_cluster_supported(C.byref(_attn_params(kw))))


def taco2_attn_cluster_work_floats(**kw):
    fn = L.lib().ns_taco2_attn_cluster_work_bytes
    fn.restype = C.c_size_t
    return (fn(C.byref(_attn_params(kw))) + 3) // 4


def taco2_attn_cluster(direction, cluster_work, **kw):
    """The attention RNN of all decoder steps as one persistent launch; cluster_work[0] is the status word (kw may
    hold the per-step path's own `work` scratch, which the backward post-pass uses)."""
    fn = getattr(L.lib(), "ns_taco2_attn_cluster_fwd" if direction == "fwd" else "ns_taco2_attn_cluster_bwd")
    L.check(fn(C.byref(_attn_params(kw)), C.c_void_p(ptr(cluster_work)), C.c_void_p(stream())),
            "ns_taco2_attn_cluster_" + direction)


def lstm_cluster_supported(p0, p1=None, backward=False):
    """Whether the persistent BiLSTM kernels take this pair of parameter blocks (bf16; forward also the fp32-state form)."""
    if p1 is None:
        return p0.dtype == NS_BF16 and p0.H % 64 == 0 and p0.H <= 512 and p0.T >= 2
    return bool(L.lib().ns_lstm_cluster_supported(C.byref(p0), C.byref(p1), int(backward)))


def lstm_cluster_work_floats(p0):
    fn = L.lib().ns_lstm_cluster_work_bytes
    fn.restype = C.c_size_t
    return (fn(C.byref(p0)) + 3) // 4


def lstm_cluster(direction, p0, p1, work):
    """Persistent whole-sequence BiLSTM (one launch); work[0] is the status word."""
    fn = getattr(L.lib(), "ns_lstm_cluster_fwd" if direction == "fwd" else "ns_lstm_cluster_bwd")
    L.check(fn(C.byref(p0), C.byref(p1), C.c_void_p(ptr(work)), C.c_void_p(stream())), "ns_lstm_cluster_" + direction)


def split_hi_lo(src, hi, lo, n):
    """hi = bf16(src), lo = bf16(src - hi)  (pre-split operands of the 3-pass products)."""
    p = L.struct("ns_split_params")
    _fill(p, src=ptr(src), hi=ptr(hi), lo=ptr(lo), n=n)
    L.call("ns_split_hi_lo", p, stream())


def act_bwd(dy, y, dpre, rows, Cc, act, row_mask=None):
    p = L.struct("ns_act_bwd_params")
    _fill(p, dy=ptr(dy), y=ptr(y), dpre=ptr(dpre), dtype=dt(y), rows=rows, C=Cc, act=act)
    if row_mask is not None:
        p.row_period, p.row_lo, p.row_hi = row_mask
    L.call("ns_act_bwd", p, stream())


def highway(h, t, x, n, y=None, dy=None, dhpre=None, dtpre=None, dx=None):
    p = L.struct("ns_highway_params")
    _fill(p, backward=int(dy is not None), dtype=dt(h), n=n, h=ptr(h), t=ptr(t), x=ptr(x), y=ptr(y), dy=ptr(dy),
          dhpre=ptr(dhpre), dtpre=ptr(dtpre), dx=ptr(dx))
    L.call("ns_highway", p, stream())


def gru_pointwise(mode, like, N, H, t, lengths, ru=None, ru_sn=0, c=None, c_sn=0, h_prev=None, hp_sn=0, out=None,
                  out_sn=0, out2=None, out2_sn=0, dzg=None, dzg_sn=0, dh=None, dh_sn=0, carry=None, carry_sn=0,
                  dh_add=None, dha_sn=0, h_init=None, hi_sn=0, reverse=False, T=0):
    """Pointers are (tensor, element offset) pairs or None."""
    def P(x):
        return None if x is None else ptr(x[0], x[1])
    p = L.struct("ns_gru_pointwise_params")
    _fill(p, mode=mode, dtype=dt(like), N=N, H=H, t=t, lengths=ptr(lengths), ru=P(ru), ru_sn=ru_sn, c=P(c), c_sn=c_sn,
          h_prev=P(h_prev), hp_sn=hp_sn, out=P(out), out_sn=out_sn, out2=P(out2), out2_sn=out2_sn, dzg=P(dzg),
          dzg_sn=dzg_sn, dh=P(dh), dh_sn=dh_sn, carry=P(carry), carry_sn=carry_sn, dh_add=P(dh_add), dha_sn=dha_sn,
          h_init=P(h_init), hi_sn=hi_sn, reverse=int(bool(reverse)), T=T)
    L.call("ns_gru_pointwise", p, stream())


def _taco1_attn_params(kw):
    p = L.struct("ns_taco1_attn_params")
    for k, v in kw.items():
        if v is None:
            continue
        if isinstance(v, tuple):        # (tensor, offset)
            v = ptr(v[0], v[1])
        elif hasattr(v, "data_ptr"):
            v = ptr(v)
        setattr(p, k, v)
    return p


def taco1_attn_cluster_supported(**kw):
    return bool(L.lib().ns_taco1_attn_cluster_supported(C.byref(_taco1_attn_params(kw))))


def taco1_attn_cluster_work_floats(**kw):
    fn = L.lib().ns_taco1_attn_cluster_work_bytes
    fn.restype = C.c_size_t
    return (fn(C.byref(_taco1_attn_params(kw))) + 3) // 4


def taco1_attn_cluster(direction, work, **kw):
    """Tacotron-1's attention RNN (prenet -> GRU -> Bahdanau) over all decoder steps, one persistent launch; work[0] is the
    status word."""
    fn = getattr(L.lib(), "ns_taco1_attn_cluster_fwd" if direction == "fwd" else "ns_taco1_attn_cluster_bwd")
    L.check(fn(C.byref(_taco1_attn_params(kw)), C.c_void_p(ptr(work)), C.c_void_p(stream())), "ns_taco1_attn_cluster_" + direction)


def gru_seq_params(like, N, T, H, P, padl, reverse, lengths, xg, xc, wgT, wcT, wg, ld_wg, wc, ld_wc, h, ld_h, ru, c, rh,
                   h_init=None, ld_hi=0, dh=None, ld_dh=0, dzg=None, dzc=None, dh_init=None, ld_dhi=0):
    """One direction of ns_gru_seq_*; pointer arguments are tensors, (tensor, element offset) pairs or None."""
    p = L.struct("ns_gru_seq_params")
    _fill(p, dtype=dt(like), N=N, T=T, H=H, P=P, padl=padl, reverse=int(bool(reverse)), f32_passes=F32_PASSES or 0,
          lengths=ptr(lengths), xg=_pp(xg), ld_xg=2 * H, xc=_pp(xc), ld_xc=H, wgT=_pp(wgT), wcT=_pp(wcT), wg=_pp(wg),
          ld_wg=ld_wg, wc=_pp(wc), ld_wc=ld_wc, h=_pp(h), ld_h=ld_h, ru=_pp(ru), c=_pp(c), rh=_pp(rh), h_init=_pp(h_init),
          ld_hi=ld_hi, dh=_pp(dh), ld_dh=ld_dh, dzg=_pp(dzg), dzc=_pp(dzc), dh_init=_pp(dh_init), ld_dhi=ld_dhi)
    return p


def gru_seq_supported(p0, p1=None, backward=False):
    return bool(L.lib().ns_gru_seq_supported(C.byref(p0), C.byref(p1) if p1 is not None else None, int(backward)))


def gru_seq_work_floats(p0):
    fn = L.lib().ns_gru_seq_work_bytes
    fn.restype = C.c_size_t
    return (fn(C.byref(p0)) + 3) // 4


def gru_seq(direction, p0, p1, work):
    """Persistent whole-sequence GRU recurrence, one or two directions in one launch; work[0] is the status word."""
    fn = getattr(L.lib(), "ns_gru_seq_fwd" if direction == "fwd" else "ns_gru_seq_bwd")
    L.check(fn(C.byref(p0), C.byref(p1) if p1 is not None else None, C.c_void_p(ptr(work)), C.c_void_p(stream())),
            "ns_gru_seq_" + direction)


def _pp(x):
    """(tensor, offset) pair, tensor or None -> device address."""
    if x is None:
        return None
    if isinstance(x, tuple):
        return ptr(x[0], x[1])
    return ptr(x)


def keys_transpose(keys, keys_t, N, Ti, Tia, Pi, padl, A):
    L.check(L.lib().ns_taco2_keys_transpose(C.c_void_p(ptr(keys)), C.c_void_p(ptr(keys_t)), N, Ti, Tia, Pi, padl, A,
                                            C.c_void_p(stream())), "ns_taco2_keys_transpose")


def keys_transpose_add(keys, keys_t, N, Ti, Tia, Pi, padl, A):
    L.check(L.lib().ns_taco2_keys_transpose_add(C.c_void_p(ptr(keys)), C.c_void_p(ptr(keys_t)), N, Ti, Tia, Pi, padl, A,
                                                C.c_void_p(stream())), "ns_taco2_keys_transpose_add")


def attention_step(like, N, Ti, Pi, padl, Tia, A, E, kw, lengths, keys_t, values, q, q_sn, aprev, aout, al_sn, ctx_out,
                   ctx_sn, ctx_out2, ctx2_sn, wcl, v, e_raw, pv=None, E2=0, pv_out=None, pv_out_sn=0, ctx_rows=None):
    p = L.struct("ns_attention_step_params")
    if pv is not None:
        _fill(p, pv=ptr(pv), E2=E2, pv_out=_pp(pv_out), pv_out_sn=pv_out_sn)
    if ctx_rows is not None:
        _fill(p, ctx_rows=ptr(ctx_rows[0]), ctx_rows_K=ctx_rows[1], ctx_rows_col=ctx_rows[2])
    _fill(p, dtype=dt(like), N=N, Ti=Ti, Pi=Pi, padl_i=padl, Tia=Tia, A=A, E=E, kw=kw, lengths=ptr(lengths),
          keys_t=ptr(keys_t), values=ptr(values), q=_pp(q), q_sn=q_sn, aprev=_pp(aprev), aout=_pp(aout), al_sn=al_sn,
          ctx_out=_pp(ctx_out), ctx_sn=ctx_sn, ctx_out2=_pp(ctx_out2), ctx2_sn=ctx2_sn, wcl=_pp(wcl), v=_pp(v),
          e_raw=ptr(e_raw))
    L.call("ns_attention_step", p, stream())


def rows32_packed_floats(K, Cc):
    fn = L.lib().ns_rows32_packed_bytes
    fn.restype = C.c_size_t
    return int(fn(K, Cc)) // 4


def rows32_pack(w, K, Cc, ldw=None, cell_units=0, out=None):
    """The [K, Cc] fp32 matrix at `w` (tensor or (tensor, offset)) as ns_rows32's packed split-bf16 fragments."""
    t = w[0] if isinstance(w, tuple) else w
    packed = out if out is not None else torch.empty(rows32_packed_floats(K, Cc), dtype=torch.float32, device=t.device)
    L.check(L.lib().ns_rows32_pack(C.c_void_p(_pp(w)), C.c_int64(ldw if ldw is not None else Cc), K, Cc, cell_units,
                                   C.c_void_p(ptr(packed)), C.c_void_p(stream())), "ns_rows32_pack")
    return packed


def rows32_rows(K, device):
    """Zeroed packed activation rows (32 rows x K columns) for ns_rows32 / ns_attention_step."""
    fn = L.lib().ns_rows32_rows_bytes
    fn.restype = C.c_size_t
    return torch.zeros(int(fn(K)) // 4, dtype=torch.float32, device=device)


def rows32_rows_floats(K):
    fn = L.lib().ns_rows32_rows_bytes
    fn.restype = C.c_size_t
    return int(fn(K)) // 4


def rows32_pack_rows(a, a_sn, N, K, rows, rows_K, col0=0):
    """fp32 rows `a` (tensor or (tensor, offset)) into columns [col0, col0 + K) of packed rows of width rows_K."""
    L.check(L.lib().ns_rows32_pack_rows(C.c_void_p(_pp(a)), C.c_int64(a_sn), N, K, C.c_void_p(ptr(rows)), rows_K, col0,
                                        C.c_void_p(stream())), "ns_rows32_pack_rows")


def rows32(a, a_sn, packed, N, K, Cc, out=None, out_sn=0, bias=None, add=None, add_sn=0, act=0, out2=None, out2_sn=0,
           cell_units=0, c_prev=None, c_sn=0, c_out=None, co_sn=0, forget_bias=1.0, zoneout=0.0, h_prev=None, hp_sn=0,
           f32_passes=3, a_rows=None, rows_out=None, rows_out2=None):
    """ns_rows32: <= 32 rows against packed weights; dense (+bias, +add, act) or, with cell_units = H, an LSTMBlockCell on
    [input | h_prev] rows (destinations receive h).  The operand is fp32 rows (a, a_sn) or a_rows = (packed rows, their
    width, first column); rows_out / rows_out2 = (packed rows, width, first column) destinations.  Tensor arguments may
    be (tensor, element offset) pairs."""
    p = L.struct("ns_rows32_params")
    _fill(p, N=N, K=K, C=Cc, a=_pp(a), a_sn=a_sn, packed=ptr(packed), f32_passes=f32_passes, bias=_pp(bias), add=_pp(add),
          add_sn=add_sn, act=act, out=_pp(out), out_sn=out_sn, out2=_pp(out2), out2_sn=out2_sn, cell_units=cell_units,
          c_prev=_pp(c_prev), c_sn=c_sn, c_out=_pp(c_out), co_sn=co_sn, forget_bias=forget_bias,
          zoneout_cell=float(zoneout), zoneout_output=float(zoneout), h_prev=_pp(h_prev), hp_sn=hp_sn,
          cell_clip=float(CELL_CLIP))
    if a_rows is not None:
        _fill(p, a_rows=ptr(a_rows[0]), a_rows_K=a_rows[1], a_rows_col=a_rows[2])
    if rows_out is not None:
        _fill(p, rows_out=ptr(rows_out[0]), rows_out_K=rows_out[1], rows_out_col=rows_out[2])
    if rows_out2 is not None:
        _fill(p, rows_out2=ptr(rows_out2[0]), rows_out2_K=rows_out2[1], rows_out2_col=rows_out2[2])
    L.call("ns_rows32", p, stream())


def attention_step_bwd(like, N, Ti, Pi, padl, Tia, A, E, kw, lengths, keys, keys_t, values, q, q_sn, acur, aprev, al_sn,
                       dctx_ext, dce_sn, dctx_carry, gk, da, has_carry, dq_out, dq_sn, de_out, dctx_out, dco_sn, wcl, v):
    p = L.struct("ns_attention_step_bwd_params")
    _fill(p, dtype=dt(like), N=N, Ti=Ti, Pi=Pi, padl_i=padl, Tia=Tia, A=A, E=E, kw=kw, lengths=ptr(lengths),
          keys=ptr(keys), keys_t=ptr(keys_t), values=ptr(values), q=_pp(q), q_sn=q_sn, acur=_pp(acur), aprev=_pp(aprev),
          al_sn=al_sn, dctx_ext=_pp(dctx_ext), dce_sn=dce_sn, dctx_carry=_pp(dctx_carry), gk=ptr(gk), da=ptr(da),
          has_carry=has_carry, dq_out=_pp(dq_out), dq_sn=dq_sn, de_out=_pp(de_out), dctx_out=_pp(dctx_out), dco_sn=dco_sn,
          wcl=_pp(wcl), v=_pp(v))
    L.call("ns_attention_step_bwd", p, stream())


_POST_PART = {}


def attention_post_part(device, N, Tia, A):
    """Scratch of the fixed-order dv / dwcl sums (ns_attention_post_bwd_params.part, ns_taco2_attn_params.post_part)."""
    fn = L.lib().ns_attention_post_part_floats
    fn.restype = C.c_size_t
    return _scratch(_POST_PART, device, int(fn(int(N), int(Tia), int(A))))


def attention_post_bwd(N, S, Ti, Tia, A, kw, lengths, keys_t, q, align, de, wcl, v, dkeys_t, dv, dwcl):
    p = L.struct("ns_attention_post_bwd_params")
    p.part = ptr(attention_post_part(keys_t.device, N, Tia, A))
    _fill(p, N=N, S=S, Ti=Ti, Tia=Tia, A=A, kw=kw, lengths=ptr(lengths), keys_t=ptr(keys_t), q=ptr(q), align=ptr(align),
          de=ptr(de), wcl=_pp(wcl), v=_pp(v), dkeys_t=ptr(dkeys_t), dv=_pp(dv), dwcl=ptr(dwcl))
    L.call("ns_attention_post_bwd", p, stream())


def wavenet_post_floats(B):
    fn = L.lib().ns_wavenet_post_bytes
    fn.restype = C.c_size_t
    return (int(fn(int(B))) + 3) // 4


def wavenet_input(ids, w, x, N, T, C, Q, w_off=0, dx=None, dw=None, dw_off=0, start=0):
    p = L.struct("ns_wavenet_input_params")
    _fill(p, ids=ptr(ids), w=ptr(w, w_off), x=ptr(x), dtype=dt(x) if x is not None else 0, dx=ptr(dx),
          dx_dtype=dt(dx) if dx is not None else 0, dw=ptr(dw, dw_off), start=start, N=N, T=T, C=C, Q=Q)
    L.call("ns_wavenet_input", p, stream())


def wavenet_gate(z, rows, C, T, start, out=None, out_off=0, ld_out=0, dout=None, dout_off=0, ld_dout=0, dz=None):
    p = L.struct("ns_wavenet_gate_params")
    ref = out if out is not None else dz
    _fill(p, z=ptr(z), rows=rows, C=C, T=T, start=start, out=ptr(out, out_off), ld_out=ld_out, dtype=dt(ref),
          dout=ptr(dout, dout_off), ld_dout=ld_dout, dz=ptr(dz))
    L.call("ns_wavenet_gate", p, stream())


def wavenet_softmax_ce(logits, ld, targets, rows, Q, scale, loss_acc, acc_off=0, dlogits=None, ld_d=0):
    p = L.struct("ns_wavenet_ce_params")
    _fill(p, logits=ptr(logits), ld=ld, targets=ptr(targets), rows=rows, Q=Q, scale=scale, loss_acc=ptr(loss_acc, acc_off),
          dlogits=ptr(dlogits), ld_d=ld_d, d_dtype=dt(dlogits) if dlogits is not None else 0)
    L.call("ns_wavenet_softmax_ce", p, stream())


def wavenet_softmax(logits, ld, rows, Q, probs, logits_off=0):
    p = L.struct("ns_wavenet_softmax_params")
    _fill(p, logits=ptr(logits, logits_off), ld=ld, rows=rows, Q=Q, probs=ptr(probs))
    L.call("ns_wavenet_softmax", p, stream())


def wavenet_generate(weights, offs, dilations, L_, R, Dc, S, Q, B, n_seed, total, queue_rows, ids, uniform, queues, probs=None,
                     fgT=None, deT=None, engine=0, cond=None, dense_bias=None, skip_bias=None, post1_bias=None, post2_bias=None,
                     post_x=None, helper_stream=None):
    p = L.struct("ns_wavenet_generate_params")
    if post_x is not None:
        p.post_x, p.helper_stream = ptr(post_x), helper_stream.cuda_stream
    _fill(p, cond=ptr(cond), dense_bias=ptr(dense_bias), skip_bias=ptr(skip_bias), post1_bias=ptr(post1_bias),
          post2_bias=ptr(post2_bias))
    _fill(p, weights=ptr(weights), w_dtype=dt(weights), off_causal=offs["causal"], off_layer0=offs["layer0"],
          layer_stride=offs["layer_stride"], off_dense_in_layer=offs["dense_in_layer"], off_skip=offs["skip"],
          off_post1=offs["post1"], off_post2=offs["post2"], dilations=ptr(dilations), L=L_, R=R, Dc=Dc, S=S, Q=Q, B=B,
          n_seed=n_seed, total=total, queue_rows=queue_rows, ids=ptr(ids), uniform=ptr(uniform), queues=ptr(queues),
          probs=ptr(probs), fgT=ptr(fgT), deT=ptr(deT), engine=engine)
    L.call("ns_wavenet_generate", p, stream())
