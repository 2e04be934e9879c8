/*
 * libnspeech_hip.so — C ABI of the MI355X (gfx950) Tacotron hot path.
 *
 * The reference (MLCogUP/nspeech) has no FFI of its own: every operator below
 * replaces a stock TensorFlow-1.7 / librosa call made by the reference's Python
 * (cited per entry as file:line under /root/reference).  A maintainer binds these
 * with ctypes; no torch types cross this boundary.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (workspace included);
 *    the library never allocates, frees or synchronises;
 *  - all work is enqueued on the caller's `stream`;
 *  - return 0 on success, a negative NS_ERR_* otherwise; ns_last_error() gives the
 *    thread-local message; nothing throws or exits;
 *  - "padded layout": a [N,T,C] time series lives in a [N,P,C] buffer, P = T+padl+padr,
 *    valid rows at n*P + padl + t; pad rows are kept at exactly zero by every writer
 *    (this is what lets conv1d run as ONE strided GEMM without an im2col copy).
 */
#ifndef NSPEECH_HIP_H
#define NSPEECH_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ns_stream_t; /* hipStream_t */

enum { NS_OK = 0, NS_ERR_BAD_ARG = -1, NS_ERR_UNSUPPORTED_SHAPE = -2, NS_ERR_LAUNCH = -3,
       NS_ERR_SHORT_BUFFER = -4 /* a caller-owned output buffer is too small; retry with a larger one */ };
enum { NS_F32 = 0, NS_BF16 = 1 };
enum { NS_ACT_NONE = 0, NS_ACT_RELU = 1, NS_ACT_TANH = 2, NS_ACT_SIGMOID = 3,
       NS_ACT_SOFTSIGN = 4 /* x / (1 + |x|): the speaker projections, rnn_wrappers.py:29, modules.py:159,167 */ };

int ns_version(void);
const char* ns_device_arch(void); /* "gfx950" */
const char* ns_last_error(void);

/* Diagnostics for the coexistence tests (tests/test_coexist_gpu.py); no reference counterpart, the reference
 * is single-GPU (train.py:157).
 * ns_occupy: `blocks` workgroups of `threads` threads, each with `lds_bytes` of LDS, that stay resident for about
 *   `usec` microseconds and then exit - the footprint of a collective's channel kernel (RCCL launches a few dozen
 *   256..512-thread workgroups that sit on their CUs until the peers have answered).  heavy != 0: 128 VGPRs per lane
 *   instead of a handful.  `started` (nullable, device int, zeroed by the caller): every workgroup adds 1 once it runs.
 * ns_wait_counter: one thread that returns once *counter >= target or after timeout_usec - put on the stream in front
 *   of the kernel under test so that the occupier is resident when that kernel is dispatched. */
int ns_occupy(int blocks, int threads, int lds_bytes, int heavy, double usec, int* started, ns_stream_t stream);
int ns_wait_counter(const int* counter, int target, double timeout_usec, ns_stream_t stream);

/* ------------------------------------------------------------------ GEMM / conv1d
 * C[M,N] (=|+=) alpha * act( A·B + bias ), fp32 accumulate.
 * Replaces tf.layers.dense / tf.layers.conv1d / the matmuls inside LSTMBlockCell and
 * GRUCell (modules.py:25,58,188-198; tacotron2.py:73,107; rnn_wrappers.py:29) and
 * their gradients (tacotron2.py:153 compute_gradients).
 *   a_mode 0: A(m,k) = A[m*lda + k]      a_mode 1: A(m,k) = A[k*lda + m]
 *   b_mode 0: B(k,n) = B[n*ldb + k]      b_mode 1: B(k,n) = B[k*ldb + n]
 *   b_seg_len>0: K is cut into segments of b_seg_len; segment s of B starts at
 *                B + s*b_seg_stride (conv data-gradient walks the taps backwards).
 * conv1d 'same' over the padded layout is a_mode 0 with lda = C_in, K = k*C_in.
 *   row mask: if row_period>0, output row m is written as 0 unless
 *             lo <= (m + row_shift) % row_period < hi.
 *   col_sum/col_sumsq: optional fp32[N]; = sum / sum of squares of the stored values
 *             over unmasked rows (BatchNorm batch statistics, modules.py:198).  Deterministic:
 *             every 64-row slot of the output writes its own partial sums into `stat_part`
 *             (caller-owned scratch, ns_gemm_stat_part_floats(M, N) floats, no need to clear it),
 *             and a fixed-order second stage adds the slots up, so two calls on the same operands
 *             give the same bits (no float atomics).
 *   accumulate 0: store   1: C += (fp32 C, split_k must be 1)   2: atomic C += (fp32)
 */
typedef struct {
  int dtype; /* element type of A and B: NS_F32 | NS_BF16 */
  int M, N, K;
  const void* A; int64_t lda; int a_mode;
  const void* B; int64_t ldb; int b_mode;
  int b_seg_len; int64_t b_seg_stride;
  void* C; int64_t ldc; int c_dtype;
  int accumulate;
  const float* bias;
  int act;
  float alpha;
  int row_period, row_lo, row_hi, row_shift;
  float* col_sum; float* col_sumsq;
  int split_k;
  const void* addend; int64_t ld_add;    /* optional [M,N] (fp32, or bf16 when addend_dtype = NS_BF16) added before the activation */
  const void* gate; int64_t ld_gate;     /* optional (dtype of A) [M,N]: result *= (gate > 0)  (ReLU backward) */
  int f32_passes;   /* fp32 operands only: 0 = exact fp32 FMA kernel; 3 = split-bf16 on MFMA
                       (x = hi + lo, hi*hi + hi*lo + lo*hi, ~2^-17 relative); 1 = hi*hi only */
  int addend_dtype; /* 0 / NS_F32: addend is fp32; NS_BF16: addend is bf16 */
  /* optional pre-split low parts of fp32 values (dtype NS_BF16, a_mode 0, b_mode 0, same strides as A / B): with
   * A = hi(a), A_lo = lo(a), B = hi(b), B_lo = lo(b) the call computes hi.hi + hi.lo + lo.hi on the matrix cores
   * (the f32_passes = 3 product without the in-kernel split).  Both or neither.  With f32_passes = 2 the hi.lo term
   * is dropped (B, the weights, rounded to bf16; A exact to ~16 bits): two thirds of the matrix-core work. */
  const void* A_lo; const void* B_lo;
  /* batch > 1: `batch` independent products in one launch; item z uses A + z*batch_stride_a, B + z*batch_stride_b,
   * C + z*batch_stride_c (elements).  The per-utterance products of the attention loop (align[n] . memory[n],
   * attention.py:48 and its gradients).  No bias / addend / gate / statistics in batched calls. */
  int batch; int64_t batch_stride_a, batch_stride_b, batch_stride_c;
  float* stat_part;  /* scratch for col_sum / col_sumsq (required with them), ns_gemm_stat_part_floats(M, N) floats */
  int stat_slots;    /* set by the library */
  /* BatchNorm-BACKWARD statistics out of the product that forms dy (the data gradient of the layer above, or the
   * BiLSTM's input gradient): with stat_z set - the saved BatchNorm input of the layer below, [M,N] in the layout of C -
   * col_sumsq receives sum_m C[m,n] * (stat_z[m,n] - stat_mean[n]) * stat_istd[n] over the unmasked rows instead of the
   * sum of squares, so col_sum / col_sumsq are the two column sums of modules.py:198's backward (sum dy, sum dy*xhat)
   * and ns_bn_bwd needs no reduction pass of its own (ns_bn_bwd_params.sum_dy / sum_dyxh).  With accumulate = 1 the
   * sums are taken on the accumulated value. */
  const void* stat_z; int64_t ld_stat_z; int stat_z_dtype;
  const float* stat_mean; const float* stat_istd;
  /* Deterministic split-K (round 4).  With split_k > 1 the k slices of an output tile used to meet in fp32 atomic adds,
   * in whatever order they finished: the weight gradients (tacotron2.py:153) differed in their last bits from run to
   * run.  With splitk_work set (caller-owned scratch of ns_gemm_splitk_work_bytes(M, N, split_k) bytes, no need to clear
   * it) and splitk_count (int[ns_gemm_splitk_counters(M, N)], ZERO before the first call; every call leaves it zero)
   * each slice stores its partial tile in the scratch and raises the tile's counter; the slice that arrives LAST adds
   * the partial tiles up in slice order 0, 1, ... - whichever slice it is - and runs the epilogue once: a fixed
   * summation order, no waiting, no float atomics.  Both NULL: the atomic form.  batch must be 1. */
  float* splitk_work; int* splitk_count;
} ns_gemm_params;
int ns_gemm(const ns_gemm_params* p, ns_stream_t stream);
size_t ns_gemm_stat_part_floats(int M, int N);
size_t ns_gemm_splitk_work_bytes(int M, int N, int split_k);
size_t ns_gemm_splitk_counters(int M, int N);
/* name of the kernel the calling thread's last ns_gemm call launched (e.g. "gemm_mfma_f32_kernel<0, 1, 3>"): lets a
 * caller attribute its own event timings to the names a profiler reports. */
const char* ns_gemm_last_kernel(void);


/* ------------------------------------------------------------------ element-wise / reductions */

/* dst[i,j,c] (=|+=) src[i,j,c] with independent strides and dtypes (pack / pad / concat /
 * cast).  Replaces tf.reshape / tf.concat / slicing glue (tacotron2.py:86, helpers.py:53,
 * rnn_wrappers.py:58-64). */
typedef struct {
  const void* src; int src_dtype; int64_t src_si, src_sj;
  void* dst; int dst_dtype; int64_t dst_si, dst_sj;
  int I, J, Cc;
  int accumulate;
} ns_copy3d_params;
int ns_copy3d(const ns_copy3d_params* p, ns_stream_t stream);

/* tf.nn.embedding_lookup (modules.py:8-18) into the padded layout, and its scatter-add
 * gradient. */
typedef struct {
  const int* ids;          /* [N,T] int32 */
  const float* table;      /* [V,D] */
  void* out; int out_dtype; /* [N,P,D] */
  int N, T, P, padl, D, V;
} ns_embedding_params;
int ns_embedding_fwd(const ns_embedding_params* p, ns_stream_t stream);
typedef struct {
  const int* ids;
  const float* dout;       /* fp32 [N,P,D] */
  float* dtable;           /* [V,D] += : workgroup (row, utterance) gathers that utterance's positions holding the row, in
                              position order; the utterances' shares are added in utterance order (a fixed summation
                              order, no float atomics: tacotron2.py:153's gradient is bit-reproducible) */
  int N, T, P, padl, D, V;
  float* work;             /* fp32 scratch of ns_embedding_bwd_work_floats(N, D, V) floats whose FIRST 1024 words are zero
                              before the first call; every call leaves them zero */
} ns_embedding_bwd_params;
int ns_embedding_bwd(const ns_embedding_bwd_params* p, ns_stream_t stream);
size_t ns_embedding_bwd_work_floats(int N, int D, int V);

/* tf.layers.batch_normalization (modules.py:198), axis -1, eps 1e-3, momentum 0.99.
 * Training: statistics come from col_sum / col_sumsq (accumulated by the producing GEMM)
 * over `count` rows; moving stats are updated in place.  Inference: moving stats.
 * Rows failing the (period, lo, hi) test are written as zero. */
typedef struct {
  const void* z; void* y; int dtype;        /* [rows,C] both */
  int rows, C;
  const float* col_sum; const float* col_sumsq; float count;
  const float* gamma; const float* beta;
  float* moving_mean; float* moving_var;
  float* mean_out; float* istd_out;         /* saved for backward, [C] each */
  float eps, momentum;
  int training;
  int row_period, row_lo, row_hi;
  /* optional (dtype NS_F32): the output also / instead as a pre-split bf16 pair y_hi = bf16(y), y_lo = bf16(y - y_hi),
   * the operand form of the three-segment 256-tile product (ns_gemm A_lo / B_lo); with them `y` may be NULL */
  void* y_hi; void* y_lo;
  int64_t ld_y;     /* row stride of y in elements; 0 = C.  A wider one writes the output as a column block of a concatenated
                       activation (the CBHG convolution bank, modules.py:121-128, without the copy) */
} ns_bn_fwd_params;
int ns_bn_fwd(const ns_bn_fwd_params* p, ns_stream_t stream);

/* BatchNorm + activation + bias backward of modules.py:194-198.
 * dy fp32 [rows,C] (grad wrt BN output), z = activated pre-BN value saved by forward.
 * Outputs: dpre (grad wrt conv output before the activation, operand dtype, pad rows 0),
 * dgamma/dbeta/dbias += .  work = fp32[200*C] scratch (per-row-block partial sums, added up in a fixed order: every
 * output is bitwise repeatable; no need to clear it).
 * sum_dy / sum_dyxh (both or neither, fp32[C]): the column sums of dy and dy*xhat over the unmasked rows, as left by
 * the product that formed dy (ns_gemm_params.stat_z); without them a reduction pass over dy and z computes them. */
typedef struct {
  const float* dy; const void* z; void* dpre; int dtype;
  int rows, C;
  const float* mean; const float* istd; const float* gamma;
  float* dgamma; float* dbeta; float* dbias;
  float* work;
  float count;
  int act;
  int row_period, row_lo, row_hi;
  int dpre_dtype;   /* 0: dpre has `dtype`; NS_BF16 with dtype NS_F32: dpre is written as bf16 (single-pass backward
                       products read it at half the bytes; the bias gradient is summed on the rounded values) */
  const float* sum_dy; const float* sum_dyxh;
  int64_t ld_dy;    /* row stride of dy in elements; 0 = C (dy as a column block of a concatenated activation's gradient) */
} ns_bn_bwd_params;
int ns_bn_bwd(const ns_bn_bwd_params* p, ns_stream_t stream);

/* out[c] += sum over rows of x[row,c]  (bias gradients).  With `work` (caller-owned fp32 scratch of
 * ns_colsum_work_floats(C) floats whose FIRST 1024 words are zero before the first call; every call leaves them zero) the
 * row blocks park their partial sums there and the last one to arrive adds them in block order: a fixed summation
 * order.  work NULL: the row blocks meet in float atomics (order varies from run to run). */
typedef struct {
  const void* x; int dtype; int64_t ld;
  int rows, C;
  float* out;
  float* work;
} ns_colsum_params;
int ns_colsum(const ns_colsum_params* p, ns_stream_t stream);
size_t ns_colsum_work_floats(int C);

/* L1 losses of tacotron2.py:130-139 and their gradient in one pass.
 * pred fp32 padded [N,P,ldp] (valid rows padl..padl+T), target fp32 [N,T,F].
 * loss_acc[0] += sum|d| over all bins, loss_acc[1] += sum|d| over bins < n_prio.
 * dpred[n,padl+t,f] = sign(pred-target) * (w_all + (f<n_prio ? w_prio : 0)). */
typedef struct {
  const float* pred; int64_t ldp;
  const float* target;
  void* dpred; int dpred_dtype; int64_t ldd;
  int N, T, P, padl, F;
  int n_prio; float w_all, w_prio;
  float* loss_acc;
  int vec4;                /* set by the library (four columns per thread where the layout allows) */
} ns_l1_loss_params;
int ns_l1_loss(const ns_l1_loss_params* p, ns_stream_t stream);

/* ------------------------------------------------------------------ optimizer
 * tf.clip_by_global_norm + tf.train.AdamOptimizer (tacotron2.py:150-161).
 * ns_sumsq: out[0] += sum g^2.   ns_adam: scale = clip / max(sqrt(gnorm_sq[0]), clip);
 * m,v,p updated in place with lr_t (bias-corrected on the host); optional bf16 shadow copy.
 * With `work` (fp32[1032], zeroed once by the caller) the sum is formed in a FIXED order (per-block partials, then the
 * last block to arrive adds them up in index order), so identical inputs give bit-identical norms on every
 * data-parallel rank; without it the block sums meet in one float atomic (order varies from run to run). */
typedef struct { const float* x; int64_t n; float* out; float* work; } ns_sumsq_params;
int ns_sumsq(const ns_sumsq_params* p, ns_stream_t stream);
typedef struct {
  float* p; const float* g; float* m; float* v; int64_t n;
  const float* gnorm_sq; float clip; float grad_scale;
  float lr_t, beta1, beta2, eps;
  void* shadow_bf16;
  /* status words of the step's persistent recurrences (ns_lstm_wide_* / ns_lstm_cluster_* / ns_taco2_attn_cluster_*:
   * the first int of their work buffers; unused entries NULL).  If any is non-zero - an exchange timed out, that pass's
   * outputs and therefore this gradient are invalid - NOTHING is updated and *skipped (nullable) is set to 1, on the
   * device, without a host round trip in front of the optimiser. */
  const int* status[12];
  float* skipped;
} ns_adam_params;
int ns_adam(const ns_adam_params* p, ns_stream_t stream);

/* Moments of `nseg` segments of a flat fp32 buffer: out[4 s + {0,1,2,3}] = sum, sum of squares, min, max over
 * x[offsets[s] .. offsets[s+1]) (offsets: DEVICE int64[nseg + 1]).  The training summaries of tacotron2.py:163-188
 * (tf.summary.histogram of outputs / targets, per-variable tf.norm of the gradients, max_gradient_norm), reduced on the
 * device so that a summary costs one small read-back. */
typedef struct { const float* x; const int64_t* offsets; int nseg; float* out; } ns_segment_stats_params;
int ns_segment_stats(const ns_segment_stats_params* p, ns_stream_t stream);

/* Zero `bytes` (a multiple of 16, p 16-byte aligned) with a kernel on the stream.  Kernel, not hipMemsetAsync: a memset
 * NODE of a captured HIP graph was seen to replay wrongly on ROCm 7.2 (see csrc/core.hip). */
int ns_zero(void* p, size_t bytes, ns_stream_t stream);
/* The same for n buffers in one launch per NS_ZERO_MANY_MAX of them (ptrs / bytes: host arrays, read before the call returns). */
#define NS_ZERO_MANY_MAX 24
int ns_zero_many(void* const* ptrs, const size_t* bytes, int n, ns_stream_t stream);
/* Do two streams run side by side?  HIP deals its hardware queues to streams in turn, so a second stream shares the
 * first one's queue whenever the process has created a multiple of the queue count in between - its launches then
 * queue BEHIND the first stream's and an overlap planned on it is silently lost (measured: the seventh stream of a
 * process).  The probe: a one-thread kernel on `a` waits up to 200 us for a word that a one-thread kernel launched on
 * `b` right behind it sets.  Returns 1 (concurrent), 0 (b ran only after a's kernel gave up) or a negative error;
 * synchronises both streams.  work: 16 bytes of device memory. */
int ns_streams_concurrent(ns_stream_t a, ns_stream_t b, void* work);

/* hi[i] = bf16(src[i]), lo[i] = bf16(src[i] - hi[i]): pre-split operands for f32_passes = 3. */
typedef struct { const float* src; void* hi; void* lo; int64_t n; } ns_split_params;
int ns_split_hi_lo(const ns_split_params* p, ns_stream_t stream);

/* dst[c,r] = (T)src[r,c]  (k-contiguous shadow copies of recurrent weights).  dst_hi / dst_lo (bf16, same layout as
 * dst, both or neither): the pre-split pair of ns_split_hi_lo from the same pass; dst may then be NULL. */
typedef struct {
  const float* src; int rows, cols; int64_t ld_src;
  void* dst; int dst_dtype; int64_t ld_dst;
  int transpose;
  void* dst_hi; void* dst_lo;
} ns_cast2d_params;
int ns_cast2d(const ns_cast2d_params* p, ns_stream_t stream);
/* Many ns_cast2d calls as ONE launch (the per-step refresh of the weight shadows after tf.train.AdamOptimizer's update,
 * tacotron2.py:159-161).  table_dev: `n` parameter blocks in DEVICE memory, each of which ns_cast2d_batchable() accepted
 * (it returns the block's number of 64 x 64 tiles, 0 = not eligible: ragged or unaligned); tile_end_dev: DEVICE int[n],
 * the running sums of those tile counts; total_tiles = the last of them. */
int ns_cast2d_batchable(const ns_cast2d_params* p);
int ns_cast2d_batch(const ns_cast2d_params* table_dev, const int* tile_end_dev, int n, int total_tiles, ns_stream_t stream);

/* ------------------------------------------------------------------ LSTM over time
 * tf.contrib.rnn.LSTMBlockCell inside dynamic_rnn / bidirectional_dynamic_rnn / dynamic_decode
 * (modules.py:40-49, tacotron2.py:67-83): gates [i,j,f,o] = xg[t] + h[t-1].Wh,
 * c' = sig(f+forget_bias)*c + sig(i)*tanh(j), h' = sig(o)*tanh(c').
 * xg = x.Wx + b is hoisted by the caller into one big ns_gemm.  All per-row buffers use the
 * padded layout row(n,t) = n*P + padl + t; h and c of step t-1 (t+1 when reverse) are read
 * back from hist buffers, whose pad rows must be zero (= zero initial state).
 * lengths (optional): rows with t >= lengths[n] emit h = 0, c = 0 (state frozen at zero
 * initial state for the reverse direction, unused afterwards for the forward one). */
typedef struct {
  int dtype;
  int N, T, H, P, padl;
  const float* xg; int64_t ld_xg;      /* [N*P, >=4H] fp32 */
  const void* whT;                      /* [4H, H] (dtype) k-contiguous */
  const void* wh;                       /* [H, 4H] (dtype) natural layout (backward) */
  const int* lengths;
  int reverse;
  float forget_bias;
  void* h; int64_t ld_h;                /* (dtype) [N*P, ld_h], this direction's H columns */
  float* c;                             /* fp32 [N*P, H] */
  void* gates;                          /* (dtype) [N*P, 4H] post-activation i,j,f,o */
  /* backward only */
  const float* dh; int64_t ld_dh;       /* fp32 [N*P, ld_dh] grad wrt h outputs */
  void* dgates;                         /* (dtype) [N*P, 4H] out */
  float* work;                          /* fp32 [N*H] scratch (cell-state gradient carry) */
  int f32_passes;                       /* as in ns_gemm_params, for the recurrent product */
  /* optional pre-split bf16 copies of fp32 weights (dtype NS_F32 only): whT = whT_hi + whT_lo
   * spares the in-kernel split for f32_passes = 3; wh_bf16 serves f32_passes = 1 at half the bytes */
  const void* whT_hi; const void* whT_lo; const void* wh_bf16;
  void* dgates_bf16;                    /* optional [N*P, 4H] bf16 copy of dgates written and re-read by the
                                           backward recurrence (with wh_bf16: pure bf16 operand loads) */
  void* h_bf16; int64_t ld_h_bf16;      /* optional bf16 copy of h, written by the fp32 form of ns_lstm_cluster_fwd */
  /* Zoneout (Krueger et al. 2017; BASELINE north_star: "2-layer Zoneout-LSTM decoder").  The reference builds plain
   * LSTMBlockCells (tacotron2.py:69-70), so both thresholds 0 - the default - IS the reference and takes the code path
   * it always took.  With a threshold > 0, training step t keeps the OLD value of a unit with probability rate:
   *   c[t] = m_c ? c[t-1] : c'[t],   h[t] = m_h ? h[t-1] : h'[t],   c' = f c[t-1] + i j,  h' = o tanh(c')
   * where m(t, n, u) = (mix(seed, t, n, u) >> 8) < thr with thr = floor(rate * 2^24) and mix = three rounds of the
   * 32-bit murmur3 finaliser over seed ^ t * 0x9E3779B9, ^ n * 0x7FEB352D, ^ u * 0x846CA68B (csrc/common.h:
   * ns_zone_keep; counter-based, so the backward pass regenerates the masks instead of storing them).  The saved gates
   * are those of the plain cell; `c` holds the zoned state.  The backward calls apply the matching gradient (the carry
   * of dh through kept units needs a second [N*H] row of `work`: ns_lstm_seq_work_bytes()).  Supported by
   * ns_lstm_seq_fwd / _bwd, ns_lstm_wide_fwd and the partial-sum form of ns_lstm_wide_bwd; the other persistent forms
   * report "unsupported" and the caller falls back to the step launches. */
  uint32_t zoneout_thr_cell, zoneout_thr_output;
  uint32_t zoneout_seed_cell, zoneout_seed_output;
  /* LSTMBlockCell's cell_clip attribute: > 0 clips the new cell state to [-cell_clip, cell_clip] in every forward form
   * (the clipped state is what c holds); TF's gradient kernel applies no mask for clipped values, so the backward forms
   * are unchanged.  0 (the default) = no clipping = the reference's cells as this build reads them (modules.py:41-42,
   * tacotron2.py:69-70 pass no cell_clip; hparam lstm_cell_clip). */
  float cell_clip;
} ns_lstm_seq_params;
int ns_lstm_seq_fwd(const ns_lstm_seq_params* p, ns_stream_t stream);
int ns_lstm_seq_bwd(const ns_lstm_seq_params* p, ns_stream_t stream);
size_t ns_lstm_seq_work_bytes(const ns_lstm_seq_params* p);
/* Two independent recurrences with equal N/T/H (the fw and bw halves of
 * tf.nn.bidirectional_dynamic_rnn, modules.py:40-46) advanced together: one launch per step. */
int ns_lstm_seq2_fwd(const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1, ns_stream_t stream);
int ns_lstm_seq2_bwd(const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1, ns_stream_t stream);


/* Persistent variant of ns_lstm_seq2_*: ONE launch for the whole sequence.  Each (direction,
 * 16-row group) recurrence runs on a cluster of H/64 workgroups that keep their W_h slice and the
 * cell state in registers and exchange h (backward: the gate gradients) through `work` with
 * tagged 8-byte granules.  bf16: H %% 64 == 0, H <= 512.  The forward call also takes dtype NS_F32 (H <= 256,
 * f32_passes 3 with whT_hi / whT_lo): fp32 state, the recurrent product as three split-bf16 passes, h written as fp32
 * (+ the optional h_bf16 copy), c as fp32 and the saved gates as BF16 - the backward pass of that arrangement is the
 * bf16 call on wh = the bf16 weight copy.  ns_lstm_cluster_supported() says whether a pair of parameter blocks
 * qualifies.  work[0] (int) is a status word: non-zero after the call completes = an exchange timed out and the
 * outputs are invalid. */
int ns_lstm_cluster_supported(const ns_lstm_seq_params* fw, const ns_lstm_seq_params* bw, int backward);
size_t ns_lstm_cluster_work_bytes(const ns_lstm_seq_params* p);
int ns_lstm_cluster_fwd(const ns_lstm_seq_params* fw, const ns_lstm_seq_params* bw, void* work, ns_stream_t stream);
int ns_lstm_cluster_bwd(const ns_lstm_seq_params* fw, const ns_lstm_seq_params* bw, void* work, ns_stream_t stream);

/* Persistent variant of ns_lstm_seq_* for WIDE cells at small batch (the decoder LSTMs of tacotron2.py:67-73: 1024
 * units, 32 rows): ONE launch for the whole sequence.  A workgroup keeps its slice of W_h in registers (forward: 8
 * units x 4 gates; backward: 16 units) and only the state travels, through the history arrays themselves: the call
 * first fills h[:, 0..T) (backward: the bf16 gate gradients) with an all-ones NaN sentinel, the producers store
 * h[t] write-through and every workgroup of the row group polls and fetches h[t-1] (backward: the gate gradients of
 * step t+1) with L1/L2-bypassing loads until no element is the sentinel - the data is its own flag.
 * Forward: dtype NS_BF16, or NS_F32 with whT_hi / whT_lo and f32_passes == 3.  Backward: NS_BF16, or NS_F32 with
 * wh_bf16 + dgates_bf16 and f32_passes == 1.  H in {256, 512, 1024}, reverse == 0, and the grid (16-row groups x H/8
 * forward, x H/16 backward) must fit the device at one workgroup per CU (<= 256).  ns_lstm_wide_supported() says
 * whether a parameter block qualifies.  work: ns_lstm_wide_work_bytes(); work[0] (int) is a status word, non-zero
 * after the call completes = a wait timed out and the outputs are invalid (they may then hold the sentinel). */
int ns_lstm_wide_supported(const ns_lstm_seq_params* p, int backward);
size_t ns_lstm_wide_work_bytes(const ns_lstm_seq_params* p);
int ns_lstm_wide_fwd(const ns_lstm_seq_params* p, void* work, ns_stream_t stream);
int ns_lstm_wide_bwd(const ns_lstm_seq_params* p, void* work, ns_stream_t stream);

/* One LSTMBlockCell step on an explicit input row: gates = [a].W^T + xg + bias with a = the
 * concatenated [input | h_prev] rows (the free-running decoder of tacotron2.py:67-83 with
 * TacoTestHelper feedback, helpers.py:32-38, where nothing can be hoisted).
 * wT is the whole TF kernel transposed, [4H, K] k-contiguous. */
typedef struct {
  int dtype, N, H, K;
  const void* a; int64_t a_sn;
  const void* wT;
  const float* xg; int64_t xg_sn;
  const float* bias;
  const float* c_prev; int64_t c_sn;
  void* h_out; int64_t h_sn;
  void* h_out2; int64_t h2_sn;
  float* c_out; int64_t co_sn;
  float forget_bias;
  int f32_passes;
  const void* wT_hi; const void* wT_lo;   /* optional pre-split weights (dtype NS_F32) */
  /* zoneout at inference = the expectation of the training masks (ns_lstm_seq_params): with a rate > 0,
   * c = zc c_prev + (1 - zc) c', h = zh h_prev + (1 - zh) h'; h_prev = this cell's previous output rows (dtype, row
   * stride hp_sn; NULL = zeros).  Both rates 0 (the default, = the reference's plain cells) reads nothing. */
  float zoneout_cell, zoneout_output;
  const void* h_prev; int64_t hp_sn;
  float cell_clip;                        /* as in ns_lstm_seq_params */
} ns_lstm_step_params;
int ns_lstm_step(const ns_lstm_step_params* p, ns_stream_t stream);

/* rows32: at most 32 activation rows against a weight matrix that stays FIXED over many calls - the step products
 * of batched free-running synthesis (tacotron2.py:55-83 under TacoTestHelper, helpers.py:7-38: a dense layer or an
 * LSTMBlockCell on [input | h_prev] rows per decoder step, nothing hoistable), which are bound by streaming the weights
 * (62 MB per step for the two decoder LSTMs).  Both operands can live in memory as the split-bf16 (hi, lo) MFMA
 * fragments the kernel consumes, so that every wave instruction loads consecutive bytes and nothing is converted:
 *   weights: ns_rows32_pack writes them ONCE per synthesis call.  w: [K, C] row-major (the TF kernel as stored), row
 *     stride ldw.  cell_units = 0: dense, column tile t = columns 16 t .. 16 t + 15; cell_units = H > 0 (C = 4 H): tile
 *     t = units 4 t .. 4 t + 3 x the gates i, j, f, o.  K % 8 == 0.  packed: ns_rows32_packed_bytes(K, C) bytes.
 *   activations ("packed rows"): a buffer of ns_rows32_rows_bytes(K) bytes holds 32 rows x K columns (K rounded up to
 *     32); the caller ZEROES it once (rows past N and columns past K must stay finite), ns_rows32 / ns_attention_step
 *     write into column ranges of it (rows_out, ctx_rows) and ns_rows32 reads its operand out of it (a_rows; the operand
 *     may be a column range [a_rows_col, a_rows_col + K) of a wider buffer, a_rows_col % 32 == 0).  ns_rows32_pack_rows
 *     converts fp32 rows into columns [col0, col0 + K) of such a buffer (rows_K = its width).
 *   fp32 activation rows (a, a_sn) are accepted too and split in registers. */
size_t ns_rows32_packed_bytes(int K, int C);
int ns_rows32_pack(const float* w, int64_t ldw, int K, int C, int cell_units, void* packed, ns_stream_t stream);
size_t ns_rows32_rows_bytes(int K);
int ns_rows32_pack_rows(const float* a, int64_t a_sn, int N, int K, void* rows, int rows_K, int col0, ns_stream_t stream);
typedef struct {
  int N, K, C;                     /* rows (<= 32), contraction, output columns (4 H in the cell form) */
  const float* a; int64_t a_sn;    /* fp32 activation rows, K floats each (16-byte aligned rows), or NULL with a_rows */
  const void* a_rows; int a_rows_K, a_rows_col;   /* packed rows of width a_rows_K; the operand starts at column a_rows_col */
  const void* packed;              /* ns_rows32_pack of the [K, C] weights with the same cell_units */
  int f32_passes;                  /* 1 (bf16 operands) or 3 (split-bf16: hi.hi + hi.lo + lo.hi) */
  const float* bias;               /* [C], nullable */
  /* dense form (cell_units = 0): y[n, c] = act(a[n] . w[:, c] + bias[c] + add[n, c]) */
  const float* add; int64_t add_sn;
  int act;
  /* destinations of y (dense) / h (cell), each optional, at least one: two fp32 row sets and two column ranges of
   * packed rows (width rows_out_K, first column rows_out_col) */
  float* out; int64_t out_sn;
  float* out2; int64_t out2_sn;
  void* rows_out; int rows_out_K, rows_out_col;
  void* rows_out2; int rows_out2_K, rows_out2_col;
  /* cell form (cell_units = H): the arguments of ns_lstm_step_params with the same meaning */
  int cell_units;
  const float* c_prev; int64_t c_sn;   /* NULL = zeros */
  float* c_out; int64_t co_sn;
  float forget_bias;
  float zoneout_cell, zoneout_output;
  const float* h_prev; int64_t hp_sn;
  float cell_clip;                     /* as in ns_lstm_seq_params */
} ns_rows32_params;
int ns_rows32(const ns_rows32_params* p, ns_stream_t stream);

/* One location-sensitive attention step (attention.py:30-60): energies, masked softmax, context. */
typedef struct {
  int dtype, N, Ti, Pi, padl_i, Tia, A, E, kw;
  const int* lengths;
  const float* keys_t;              /* fp32 [N,A,Tia] (see ns_taco2_keys_transpose) */
  const void* values;               /* (dtype) [N*Pi, E] */
  const float* q; int64_t q_sn;     /* fp32 [N, A] rows */
  const float* aprev; float* aout; int64_t al_sn;   /* fp32 [N, Tia] rows */
  void* ctx_out; int64_t ctx_sn;    /* (dtype) [N, E] rows */
  void* ctx_out2; int64_t ctx2_sn;  /* optional */
  const float* wcl; const float* v;
  float* e_raw;                     /* fp32 [N,Tia] scratch */
  /* optional, same launch: a second value matrix pv (dtype) [N*Pi, E2] - free-running synthesis hands in the projected
   * memory values . W_prenet1[context rows], whose weighted sum pv_out[n, 0:E2] is the context term of the NEXT step's
   * first prenet layer (no context -> prenet product inside the loop) */
  const void* pv; int E2;
  void* pv_out; int64_t pv_out_sn;
  /* optional: the context also as columns [ctx_rows_col, ctx_rows_col + E) of ns_rows32's packed rows (width ctx_rows_K) */
  void* ctx_rows; int ctx_rows_K, ctx_rows_col;
} ns_attention_step_params;
int ns_attention_step(const ns_attention_step_params* p, ns_stream_t stream);
/* keys_t[n,u,t] = keys[n, padl+t, u] */
int ns_taco2_keys_transpose(const float* keys, float* keys_t, int N, int Ti, int Tia, int Pi, int padl, int A,
                            ns_stream_t stream);

/* Backward of ns_attention_step for one decoder step (the three kernels of the training loop):
 * da = dctx.values + carry, softmax backward, gradient wrt the previous alignments (as per-tap
 * terms gk) and dq.  de_out / dctx_out feed ns_attention_post_bwd. */
typedef struct {
  int dtype, N, Ti, Pi, padl_i, Tia, A, E, kw;
  const int* lengths;
  const float* keys; const float* keys_t; const void* values;
  const float* q; int64_t q_sn;
  const float* acur; const float* aprev; int64_t al_sn;
  const float* dctx_ext; int64_t dce_sn;
  const float* dctx_carry;          /* fp32 [N,E] or NULL */
  float* gk; float* da; int has_carry;   /* fp32 [N,Tia,8] in/out, fp32 [N,Tia] scratch */
  void* dq_out; int64_t dq_sn;      /* (dtype) [N, A] rows */
  float* de_out;                    /* fp32 rows, stride al_sn */
  void* dctx_out; int64_t dco_sn;   /* (dtype) [N, E] rows */
  const float* wcl; const float* v;
} ns_attention_step_bwd_params;
int ns_attention_step_bwd(const ns_attention_step_bwd_params* p, ns_stream_t stream);
size_t ns_attention_post_part_floats(int N, int Tia, int A);
/* Sums over all S decoder steps that no recurrence needs: dkeys_t (plain store), dv +=, dwcl +=. */
typedef struct {
  int N, S, Ti, Tia, A, kw;
  const int* lengths;
  const float* keys_t; const float* q; const float* align; const float* de;   /* [N,S+1,...] slot layout */
  const float* wcl; const float* v;
  float* dkeys_t; float* dv; float* dwcl;
  /* optional scratch of ns_attention_post_part_floats(N, Tia, A) floats: the (utterance, position block) partial sums of
   * dv / dwcl are parked there and added in a fixed order by a second launch instead of meeting in float atomics */
  float* part;
} ns_attention_post_bwd_params;
int ns_attention_post_bwd(const ns_attention_post_bwd_params* p, ns_stream_t stream);
/* keys[n, padl+t, u] += keys_t[n,u,t] */
int ns_taco2_keys_transpose_add(float* keys, const float* keys_t, int N, int Ti, int Tia, int Pi, int padl, int A,
                                ns_stream_t stream);

/* ------------------------------------------------------------------ GRU / highway element-wise
 * tf.contrib.rnn.GRUCell pieces (modules.py:92,172-181; tacotron.py:69-76) around the gate GEMMs:
 *  mode 0: rh = r * h_prev                         (ru = [r | u] fp32 [N,2H])
 *  mode 1: h = u*h_prev + (1-u)*c  (0 past length) -> h_out (dtype) (+ h_out2)
 *  mode 2: backward A: dzc = dh*(1-u)*(1-c^2); dzu = dh*(h_prev-c)*u*(1-u) -> dzg[:,H:]; carry = dh*u
 *  mode 3: backward B: dzr = drh*h_prev*r*(1-r) -> dzg[:,:H]; carry += drh*r
 * rows with t >= lengths[n] produce zeros (modes 1-3) and leave `carry` as it is (mode 2): the state, and so its
 * gradient, passes a step beyond the length unchanged (tf.nn.dynamic_rnn).
 * Non-zero initial state (modules.py:165-181, the speaker projection as initial_state_fw / _bw): with h_init set,
 * h_prev is h_init[n] at a sequence's first step - t == 0, or with `reverse` t == len(n) - 1 where len(n) =
 * lengths ? lengths[n] : T - instead of what h_prev points at. */
typedef struct {
  int mode, dtype, N, H, t;
  const int* lengths;
  const float* ru; int64_t ru_sn;
  const float* c; int64_t c_sn;
  const void* h_prev; int64_t hp_sn;      /* (dtype), NULL = zeros */
  void* out; int64_t out_sn;              /* mode 0: rh; mode 1: h_out; mode 2: dzc (dtype) */
  void* out2; int64_t out2_sn;            /* mode 1: optional second h destination */
  void* dzg; int64_t dzg_sn;              /* (dtype) [N,2H] rows (modes 2, 3) */
  const float* dh; int64_t dh_sn;         /* modes 2: total grad wrt h (fp32); mode 3: drh */
  float* carry; int64_t carry_sn;         /* fp32 [N,H] */
  const float* dh_add; int64_t dha_sn;    /* mode 2, optional: added to dh where the row is valid (the gradient wrt this
                                             step's output; dh itself then only carries the recurrent part) */
  const float* h_init; int64_t hi_sn;     /* optional fp32 [N,H] initial state */
  int reverse, T;
} ns_gru_pointwise_params;
int ns_gru_pointwise(const ns_gru_pointwise_params* p, ns_stream_t stream);

/* Persistent whole-sequence GRU recurrence: ONE launch instead of four per time step (two gate products + two
 * element-wise kernels) - the BiGRU(128) of both CBHGs (modules.py:172-181 under bidirectional_dynamic_rnn) and the
 * residual GRU(256) decoder cells (tacotron.py:69-76).  tf.contrib.rnn.GRUCell with the input halves hoisted:
 *   [r | u] = sigmoid(xg[t] + h . Wg_h),  c = tanh(xc[t] + (r * h) . Wc_h),  h' = u * h + (1 - u) * c.
 * A chain = (direction, 16 batch rows).  H = 128: one workgroup per chain, nothing leaves the CU - the recurrent
 * matrices (128 x 384) live in registers as MFMA fragments (hi and lo planes for three passes), the state in LDS.
 * H = 256: a chain is a cluster of 4 workgroups with 64 units each that exchange r * h and h through `work` as
 * {step tag, fp32} granules (two hops per step).  dtype NS_BF16: single-pass bf16 products; NS_F32: f32_passes 1
 * (operands rounded to bf16 when loaded) or 3 (split-bf16, ~fp32).  Rows with t >= lengths[n] output zeros and carry
 * the state unchanged; h_init (optional) is the initial state (modules.py:165-181).
 * Forward writes the history h (dtype), ru = [r | u] and c (fp32) and rh = r * h_prev (dtype) for the backward pass and
 * the hoisted weight gradients.  Backward reads them plus dh (gradient wrt the outputs) and writes the gate gradients
 * dzg = [dzr | dzu], dzc (dtype) and, optionally, the gradient wrt the initial state.
 * fw / bw: one or two directions with equal N, T, H, P, dtype (bw may be NULL).  work: ns_gru_seq_work_bytes();
 * work[0] (int) is a status word, non-zero after the call completes = an exchange timed out, outputs invalid. */
typedef struct {
  int dtype, N, T, H, P, padl;      /* rows of every [N*P, .] array: n * P + padl + t */
  int reverse;                      /* 1: walk t = T-1 .. 0 */
  int f32_passes;                   /* dtype NS_F32: 1 or 3 */
  const int* lengths;               /* nullable [N] */
  const float* xg; int ld_xg;       /* [N*P, 2H]: x . Wg_x + bg */
  const float* xc; int ld_xc;       /* [N*P, H]:  x . Wc_x + bc */
  const void* wgT;                  /* (dtype) [2H][H]: recurrent rows of gates/kernel, transposed (k contiguous)    fwd */
  const void* wcT;                  /* (dtype) [H][H]:  recurrent rows of candidate/kernel, transposed               fwd */
  const void* wg; int ld_wg;        /* (dtype) [H][2H] row-major recurrent rows of gates/kernel                      bwd */
  const void* wc; int ld_wc;        /* (dtype) [H][H]                                                                bwd */
  void* h; int ld_h;                /* (dtype) history, this cell's H columns of row n * P + padl + t */
  /* saved gates, PRIVATE to this pair of calls (the forward call writes them in the order the backward call's LDS stage
   * wants, 1 KB of consecutive bytes per wave): [row group][t][workgroup][r | u sections][16-byte chunk][16 rows][4],
   * ceil(N / 16) * 16 * T * 2H floats (c: ... * H).  Every row group is written whole. */
  float* ru;
  float* c;
  void* rh;                         /* (dtype) [N*P, H] */
  const float* h_init; int ld_hi;   /* nullable [N, H] */
  const float* dh; int ld_dh;       /* bwd: gradient wrt h, addressed as h */
  void* dzg;                        /* bwd out: (dtype) [N*P, 2H] */
  void* dzc;                        /* bwd out: (dtype) [N*P, H] */
  float* dh_init; int ld_dhi;       /* bwd out, nullable: [N, H] gradient wrt the initial state (overwritten) */
} ns_gru_seq_params;
int ns_gru_seq_supported(const ns_gru_seq_params* fw, const ns_gru_seq_params* bw, int backward);
size_t ns_gru_seq_work_bytes(const ns_gru_seq_params* p);
int ns_gru_seq_fwd(const ns_gru_seq_params* fw, const ns_gru_seq_params* bw, void* work, ns_stream_t stream);
int ns_gru_seq_bwd(const ns_gru_seq_params* fw, const ns_gru_seq_params* bw, void* work, ns_stream_t stream);

/* dpre = dy * act'(y) for a dense layer whose output y = act(pre) was stored (ReLU / tanh /
 * sigmoid / none); rows failing the (period, lo, hi) test give 0. */
typedef struct {
  const float* dy; const void* y; void* dpre; int dtype;
  int rows, C; int act;
  int row_period, row_lo, row_hi;
} ns_act_bwd_params;
int ns_act_bwd(const ns_act_bwd_params* p, ns_stream_t stream);

/* modules.py:185-191 highway combine y = H*T + x*(1-T) and its backward
 * (dHpre = dy*T*(H>0), dTpre = dy*(H-x)*T*(1-T), dx = dy*(1-T)). */
typedef struct {
  int backward, dtype; int64_t n;
  const void* h; const void* t; const void* x;
  void* y;                               /* forward out (dtype) */
  const float* dy; void* dhpre; void* dtpre; float* dx;   /* backward */
} ns_highway_params;
int ns_highway(const ns_highway_params* p, ns_stream_t stream);

/* ------------------------------------------------------------------ Tacotron-2 attention RNN
 * The part of the decoder loop that is recurrent through the attention state
 * (tacotron2.py:63-83 with AttentionWrapper(PrenetWrapper(LSTMBlockCell(256)),
 * LocationSensitiveAttention), modules.py:83-102, attention.py:30-60, rnn_wrappers.py:25-31):
 *   p1 = relu(F1[s] + ctx[s-1].W1c)       (F1 = frame part of prenet dense_1 + bias, hoisted)
 *   p2 = relu(p1.W2 + b2)
 *   (c,h) = LSTM([p2, h[s-1]].Watt + b)
 *   e[t] = sum_u v[u] tanh(keys[t,u] + (h.Wq)[u] + sum_k align[s-1][t+k-kw/2] Wcl[k,u])
 *   align[s] = softmax over t < length;  ctx[s] = sum_t align[s][t] values[t]
 * where Wcl = location_conv . location_layer folded into one [kw, A] filter.
 * In training (teacher forcing, helpers.py:73-77) this chain does not depend on the two
 * decoder LSTMs, so it runs first for all S steps and the big LSTMs get hoisted inputs.
 * All per-step buffers are [N, S+1, X]: step s lives in slot s+1, slot 0 is the zero initial
 * state and must be zero on entry.
 */
typedef struct {
  int dtype;
  int N, S, Ti, Pi, padl_i, Tia;   /* memory row(n,t) = n*Pi + padl_i + t; Tia = ld of align */
  int A, E, D1, D2, kw;
  const int* lengths;
  const float* keys;               /* fp32 [N*Pi, A] */
  const void* values;              /* (dtype) [N*Pi, E] */
  const float* f1;                 /* fp32 [N,S+1,D1] */
  const void* w1cT; const void* w2T; const void* wattT; const void* wqT;   /* k-contiguous (dtype) */
  const float* b2; const float* batt; const float* wcl; const float* v;
  void* p1; void* xa; void* hc;    /* (dtype) [N,S+1,D1], [N,S+1,D2+A], [N,S+1,A+E] */
  float* ca; void* ga;             /* fp32 [N,S+1,A], (dtype) [N,S+1,4A] */
  float* q; float* align;          /* fp32 [N,S+1,A], fp32 [N,S+1,Tia] */
  float* keys_t;                   /* fp32 [N,A,Tia] scratch: keys transposed by fwd, re-read by bwd */
  void* align_t;                   /* (dtype) [N,S+1,Tia] copy of align written by fwd (GEMM operand in bwd) */
  float* de; void* dctx_t;         /* bwd scratch: fp32 [N,S+1,Tia] energy grads, (dtype) [N,S+1,E] context grads */
  /* backward */
  const void* w1c; const void* w2; const void* watt; const void* wq;       /* natural layouts (dtype) */
  const float* dhc;                /* fp32 [N,S+1,A+E] grad wrt hc from downstream */
  void* df1; void* dp2; void* dga; void* dq;    /* (dtype) [N,S+1,D1|D2|4A|A] out */
  float* dkeys; float* dvalues;    /* fp32 [N*Pi,A], [N*Pi,E] += */
  float* dv; float* dwcl;          /* fp32 [A], [kw*A] += */
  float* work;                     /* fp32 scratch, ns_taco2_attn_work_bytes() */
  int f32_passes;                  /* as in ns_gemm_params, for the in-loop products */
  const void* wattT_hi; const void* wattT_lo; const void* watt_bf16;   /* optional, as in ns_lstm_seq_params */
  void* dga_bf16;                  /* optional [N,S+1,4A] bf16 copy of dga for the backward recurrence */
  /* Projected-memory form (optional): pv (dtype) [N*Pi, D1] = values . W1c in the row layout of `values`.  Forward:
   * the context kernel writes the NEXT step's prenet layer relu(align . pv + f1) directly (no per-step p1 product)
   * and the contexts hc[:, :, A:] are formed after the loop by one product per batch item.  Backward additionally
   * takes da0 fp32 [N,S+1,Tia] = dhc[:, :, A:] . values^T (hoisted by the caller); the per-step dctx product
   * disappears and dctx_t is formed after the loop (df1 then needs one extra zero row behind its N*(S+1) rows).
   * Results equal the plain form up to rounding. */
  const void* pv; const float* da0;
  /* Multi-speaker (tacotron2.py:40-49, rnn_wrappers.py:28-30): Dsp > 0 widens the attention LSTM input to
   * [prenet(D2) | speaker projection(Dsp) | h(A)]; xa is then [N,S+1,D2+Dsp+A] with the per-utterance projection
   * written into columns D2..D2+Dsp of every slot by the caller, wattT / watt hold all D2+Dsp+A input rows, and the
   * caller forms the projection's gradient from dga (sum over the slots) after the backward call. */
  int Dsp;
  /* backward, optional: scratch for the fixed-order sums of dv / dwcl (ns_attention_post_part_floats(N, Tia, A) floats,
   * see ns_attention_post_bwd_params.part); NULL = float atomics */
  float* post_part;
  float cell_clip;                 /* the attention LSTM's (and, in ns_taco2_decode, the decoder LSTMs') cell_clip, as in ns_lstm_seq_params */
} ns_taco2_attn_params;
int ns_taco2_attn_fwd(const ns_taco2_attn_params* p, ns_stream_t stream);
int ns_taco2_attn_bwd(const ns_taco2_attn_params* p, ns_stream_t stream);
size_t ns_taco2_attn_work_bytes(const ns_taco2_attn_params* p);

/* The same recurrence as ONE persistent launch (per 32 utterances) instead of ~12 dependent launches per step pair:
 * an utterance runs on a cluster of 8 workgroups that keep the prenet-2 / attention-LSTM / query weights in registers
 * (exact fp32 products), the utterance's keys and memory.W1c rows in LDS, and exchange two small vectors per step
 * through `work` with tagged 8-byte granules.  Needs the projected-memory form (pv), D1 = 256, D2 = 128, A in
 * {64, 256}, T_in <= 256; reads the natural-layout weights w2 / watt / wq.  Same history outputs as
 * ns_taco2_attn_fwd / _bwd.  work: ns_taco2_attn_cluster_work_bytes(); work[0] (int) is a status word, non-zero
 * after the call completes = an exchange timed out and the outputs are invalid. */
int ns_taco2_attn_cluster_supported(const ns_taco2_attn_params* p);
size_t ns_taco2_attn_cluster_work_bytes(const ns_taco2_attn_params* p);
int ns_taco2_attn_cluster_fwd(const ns_taco2_attn_params* p, void* work, ns_stream_t stream);
/* Backward through time of the same recurrence, one persistent launch; takes what ns_taco2_attn_bwd takes (da0
 * included), writes df1 / dp2 / dga (/ dga_bf16) / dq / de and then runs the same
 * hoisted post-pass (dkeys, dv, dWcl, dctx_t, dvalues) as ns_taco2_attn_bwd. */
int ns_taco2_attn_cluster_bwd(const ns_taco2_attn_params* p, void* work, ns_stream_t stream);

/* ------------------------------------------------------------------ Tacotron-1 attention RNN, persistent
 * The recurrent part of Tacotron-1's teacher-forced decoder (tacotron.py:64-76 with AttentionWrapper(PrenetWrapper(
 * GRUCell(256)), BahdanauAttention(256)), modules.py:76-102, rnn_wrappers.py:25-31) as ONE launch per direction:
 *   p1 = relu(F1[s] + ctx[s-1].W1c)   p2 = relu(p1.W2 + b2)
 *   [r | u] = sigmoid([p2, h[s-1]].Wg + bg)   c = tanh([p2, r * h[s-1]].Wc + bc)   h = u * h[s-1] + (1 - u) * c
 *   e[t] = sum_u v[u] tanh(keys[t,u] + (h.Wq)[u])   align[s] = softmax over t < length   ctx[s] = sum_t align[s][t] values[t]
 * in the projected-memory form of ns_taco2_attn_params (pv = values . W1c: the loop forms p1[s+1] = relu(align . pv +
 * F1[s+1]) directly; the caller forms the contexts hc[:, :, A:] = align . values after the loop, and for the backward
 * pass da0 = dhc[:, :, A:] . values^T before it).  An utterance runs on a cluster of 8 workgroups that keep the prenet-2 /
 * GRU / query weights in registers as fp32 (matrix-vector products: exact FMAs), the utterance's keys and pv rows in LDS,
 * and exchange three (backward: four) small vectors per step through `work` as tagged 8-byte granules.
 * Shipped widths only: A = 256 units, D1 = 256, D2 = 128, T_in <= 256 (any memory width E, any speaker width Dsp).
 * Per-step buffers are [N, S+1, X]: step s in slot s+1, slot 0 = the zero initial state (zero on entry).
 * work: ns_taco1_attn_cluster_work_bytes(); work[0] (int) is a status word (non-zero after the call completes = an
 * exchange timed out, outputs invalid). */
typedef struct {
  int dtype;
  int N, S, Ti, Pi, padl_i, Tia;     /* memory row(n,t) = n*Pi + padl_i + t; Tia = ld of align */
  int A, E, D1, D2;
  const int* lengths;
  const float* keys;                 /* fp32 [N*Pi, A] */
  const void* pv;                    /* (dtype) [N*Pi, D1] */
  const float* f1;                   /* fp32 [N,S+1,D1] */
  const void* w2; const void* wg; const void* wc; const void* wq;   /* (dtype) [D1][D2], [D2+A][2A], [D2+A][A], [A][A] */
  const float* b2; const float* bg; const float* bc; const float* v;
  void* p1; void* xa; void* xc; void* hc;   /* (dtype) [N,S+1,D1], [N,S+1,D2+A] = [p2 | h_prev], same = [p2 | r*h_prev], [N,S+1,A+E] (h columns) */
  float* ru; float* cc;              /* fp32 [N,S+1,2A], [N,S+1,A] */
  float* q; float* align;            /* fp32 [N,S+1,A], [N,S+1,Tia] */
  void* align_t;                     /* (dtype) [N,S+1,Tia] copy of align */
  /* backward */
  const float* dhc;                  /* fp32 [N,S+1,A+E]: the h columns are read */
  const float* da0;                  /* fp32 [N,S+1,Tia] */
  void* df1; void* dp2; void* dzg; void* dzc; void* dq;   /* (dtype) [N,S+1,D1 | D2 | 2A | A | A] out */
  float* de;                         /* fp32 [N,S+1,Tia] out: energy gradients */
  /* Multi-speaker (rnn_wrappers.py:28-30): Dsp > 0 widens the GRU input to [prenet(D2) | speaker projection(Dsp) | h(A)]: xa / xc
   * rows are D2+Dsp+A wide with the per-utterance projection written into columns D2..D2+Dsp of every xa slot by the caller,
   * wg / wc hold all D2+Dsp+A rows, and the caller forms the projection's gradient from dzg / dzc after the backward call. */
  int Dsp;
} ns_taco1_attn_params;
int ns_taco1_attn_cluster_supported(const ns_taco1_attn_params* p);
size_t ns_taco1_attn_cluster_work_bytes(const ns_taco1_attn_params* p);
int ns_taco1_attn_cluster_fwd(const ns_taco1_attn_params* p, void* work, ns_stream_t stream);
int ns_taco1_attn_cluster_bwd(const ns_taco1_attn_params* p, void* work, ns_stream_t stream);

/* Free-running synthesis loop (tacotron2.py:78-83 with TacoTestHelper, helpers.py:7-38: the last predicted frame is
 * the next step's input) as ONE persistent launch: the attention-RNN clusters of ns_taco2_attn_cluster_fwd plus
 * workgroups that keep the two decoder LSTMs (tacotron2.py:67-70) register-resident as fp32, 12 units each, exchanging
 * h_att / alignment / h1 / h2 / the next frame term as tagged 8-byte granules.  The frame feedback is folded:
 * f1[s+1] = h2[s] . wpf + bpf with wpf = W_proj[:, last frame] . W_prenet1[frame rows] ([D, D1]) and
 * bpf = b_proj[last frame] . W_prenet1[frame rows] + b_prenet1, so the output projection (tacotron2.py:73) leaves the
 * loop: the caller forms decoder_outputs = h2 . W_proj + b_proj over the whole history afterwards.
 * att: the attention block as for ns_taco2_attn_cluster_fwd with S = decoder steps; f1 slot 1 must hold the <GO>
 * frame's term (= b_prenet1), later slots are not read.  Writes att.align / p1 / xa / q / ca / ga / hc[:, :, :A] and
 * h2 fp32 [N, S+1, D] (step s in slot s+1).  N <= 2; (A, E, D) = (256, 512, 1024) or (64, 64, 64).
 * work: ns_taco2_decode_work_bytes(); work[0] (int) is the status word (non-zero = an exchange timed out). */
typedef struct {
  ns_taco2_attn_params att;
  int D;
  const float* w_l1; const float* b_l1;   /* decoder/lstm_1 kernel [(A+E+D), 4D], bias [4D], fp32 */
  const float* w_l2; const float* b_l2;   /* decoder/lstm_2 kernel [2D, 4D], bias */
  const float* wpf; const float* bpf;     /* folded feedback [D, D1], [D1] */
  float* h2;
} ns_taco2_decode_params;
int ns_taco2_decode_supported(const ns_taco2_decode_params* p);
size_t ns_taco2_decode_work_bytes(const ns_taco2_decode_params* p);
int ns_taco2_decode(const ns_taco2_decode_params* p, void* work, ns_stream_t stream);


/* ------------------------------------------------------------------ simple WaveNet (models/wavenet_simple.py)
 * Row r of every series buffer is (n, t) = (r / T, r %% T) on ONE time grid of T = (clip length - 1) rows per item;
 * the VALID causal convolutions only shift the first valid row (`start`) to the right.  The convolutions are
 * ns_gemm launches; these are the pieces around them. */
/* One-hot causal layer (wavenet_simple.py:246-252, 385-397) as table look-ups: x[n,t,:] = w[0][ids[n,t-1]] +
 * w[1][ids[n,t]] (t >= 1), w = [2, Q, C] fp32.  With dx != NULL the call is the backward pass instead:
 * dw[0][ids[t-1]] += dx[t], dw[1][ids[t]] += dx[t] for t >= max(1, start) (dx (dx_dtype) [N*T, C]). */
typedef struct {
  const int* ids;            /* [N, T] */
  const float* w; void* x; int dtype;
  const void* dx; int dx_dtype; float* dw; int start;
  int N, T, C, Q;
} ns_wavenet_input_params;
int ns_wavenet_input(const ns_wavenet_input_params* p, ns_stream_t stream);
/* Gated unit (wavenet_simple.py:325): z fp32 [rows, 2C] = [filter | gate]; forward out = tanh * sigmoid into
 * out[row * ld_out + c] (dtype); with dout != NULL backward: dz (dtype) [rows, 2C].  Rows with t < start give 0. */
typedef struct {
  const float* z; int rows, C, T, start;
  void* out; int64_t ld_out; int dtype;
  const void* dout; int64_t ld_dout; void* dz;
} ns_wavenet_gate_params;
int ns_wavenet_gate(const ns_wavenet_gate_params* p, ns_stream_t stream);
/* tf.nn.softmax_cross_entropy_with_logits + reduce_mean (wavenet_simple.py:479-502) against integer targets:
 * loss_acc[0] += scale * sum_rows (logsumexp - logit[target]); dlogits (optional) = scale * (softmax - onehot). */
typedef struct {
  const float* logits; int64_t ld; const int* targets; int rows, Q;
  float scale; float* loss_acc;
  void* dlogits; int64_t ld_d; int d_dtype;
} ns_wavenet_ce_params;
int ns_wavenet_softmax_ce(const ns_wavenet_ce_params* p, ns_stream_t stream);
/* predict_proba (wavenet_simple.py:436-453): probs fp32 [rows, Q] = softmax of logits [rows, ld] evaluated in float64. */
typedef struct {
  const float* logits; int64_t ld; int rows, Q;
  float* probs;
} ns_wavenet_softmax_params;
int ns_wavenet_softmax(const ns_wavenet_softmax_params* p, ns_stream_t stream);

/* Sample-by-sample generation (generate_wavenet.py:56-142): B waveforms, one workgroup each.  ids [B, total] holds
 * n_seed seed codes per waveform (n_seed >= receptive field for results equal to the full network) and receives the
 * total - n_seed drawn ones; uniform [B, total - n_seed] in [0,1) drives the inverse-CDF draws (float64 softmax as in
 * predict_proba, wavenet_simple.py:436-453); queues fp32 [B, sum(dilations), R] zeroed by the caller; weights = one
 * flat buffer (w_dtype fp32 or bf16) with element offsets: causal [2,Q,R]; per layer l at off_layer0 + l*layer_stride
 * the [2,R,2Dc] filter|gate block and, off_dense_in_layer further, the [Dc,R] dense kernel; skip [L,Dc,S];
 * post1 [S,S]; post2 [S,Q].  probs (optional) fp32 [B,Q]: distribution of the last drawn sample.
 * fgT / deT (optional, bf16, with w_dtype = NS_BF16 and R == Dc in {16, 32}): column-major shadows [L][2Dc][2R] and
 * [L][R][Dc] of the layer kernels; with them the residual chain runs inside one wavefront (no barriers, weights
 * fetched three layers ahead), about 7x faster per sample. */
typedef struct {
  const void* weights; int w_dtype;
  int64_t off_causal, off_layer0, layer_stride, off_dense_in_layer, off_skip, off_post1, off_post2;
  const int* dilations; int L, R, Dc, S, Q;
  int B, n_seed, total; int64_t queue_rows;
  int* ids; const float* uniform; float* queues; float* probs;
  const void* fgT; const void* deT;
  int engine;   /* with fgT / deT: 0 / 1 = single-wave VALU chain, 2 = MFMA chain + concurrent skip waves (R == Dc == 32) */
  /* The full WaveNetModel's incremental generator (neural_speech/models/wavenet.py:398-437, 487-557), all optional and
   * fp32; with any of them the per-layer kernel runs (fgT / deT must be NULL):
   *   cond [B, L, 2Dc]  added to a layer's [filter | gate] pre-activations: the global condition's 1x1 convolution
   *                     (:409-419) and filter_bias | gate_bias (:421-423), formed once per call by the caller;
   *   dense_bias [L, R] (:428-429);  skip_bias [S] = the sum of the layers' skip biases (:432-434, summed :543);
   *   post1_bias [S], post2_bias [Q] (:546-553). */
  const float* cond; const float* dense_bias; const float* skip_bias; const float* post1_bias; const float* post2_bias;
  /* engine 3 (round 5) = engine 2 with the post-processing products of a drawn sample - relu -> post1 -> relu -> post2,
   * 768 KB of weights that ONE workgroup streamed per sample (16 - 18 of 41 us) - on NS_WN_HELPERS helper workgroups per
   * waveform that keep their quarter of both kernels in REGISTERS for the whole call: the chain workgroup hands over the
   * 512 skip sums, helper h forms hidden units 128 h .. 128 h + 127 (its columns of post1) and their share of all 256
   * logits (its rows of post2), the chain workgroup adds the four partial vectors in a fixed order - two hand-offs of
   * self-flagging {tag, value} granules through post_x instead of two matrix-vector products from L2.
   * post_x: ns_wavenet_post_bytes(B) of device memory, zeroed by the call; helper_stream: a stream that runs BESIDE the
   * call's own (ns_streams_concurrent) - the helper kernel is launched on it first; post_x[0] (int) is a status word
   * (non-zero after the call = a wait timed out, the samples are invalid).  S = 512, Q = 256, B * (1 + 4) <= the CUs. */
  void* post_x; void* helper_stream;   /* helper_stream: an ns_stream_t */
} ns_wavenet_generate_params;
#define NS_WN_HELPERS 4
size_t ns_wavenet_post_bytes(int B);
int ns_wavenet_generate(const ns_wavenet_generate_params* p, ns_stream_t stream);

/* ------------------------------------------------------------------ audio DSP (utils/audio.py)
 * Radix-2 Stockham FFTs of n_fft points run inside LDS, one workgroup per frame; no MFMA,
 * the kernels are bandwidth / latency bound.  window = periodic Hann(win) and
 * twiddle[m] = (cos, -sin)(2*pi*m/n_fft), m < n_fft/2, are supplied by the caller (immutable). */

/* audio.spectrogram + audio.melspectrogram in ONE pass over the waveform (audio.py:39-42,61-64
 * with librosa.stft(center=True, reflect) framing, _amp_to_db, _normalize; preemphasis fused).
 * lin_out [T, n_fft/2+1], mel_out [T, n_mels], both normalised, T = 1 + L/hop. */
typedef struct {
  const float* wav; int L;
  float preemph;
  int n_fft, hop, win, T;
  const float* window; const float* twiddle;
  const float* mel_basis; int n_mels;
  float ref_level_db, min_level_db;
  float* lin_out; float* mel_out;
  float* stft_out;   /* optional: the complex transform itself (audio._stft, audio.py:106-108), [T, n_fft/2+1] (re, im) */
} ns_spectrogram_params;
int ns_spectrogram(const ns_spectrogram_params* p, ns_stream_t stream);

/* audio.inv_spectrogram_tensorflow (audio.py:51-58,90-123): denormalise -> dB to amplitude ->
 * ^power -> Griffin-Lim with zero initial phase and TF-style (un-centred, un-normalised)
 * STFT / inverse STFT.  spec [N, T, F] normalised, wav [N, (T-1)*hop + win]. */
typedef struct {
  const float* spec; int N, T;
  int n_fft, hop, win, iters;
  float power, ref_level_db, min_level_db;
  const float* window; const float* twiddle;
  float* wav;
  float* work;   /* ns_griffin_lim_work_bytes() */
  int raw_magnitude;   /* 1: spec already holds the magnitudes S^power (audio._griffin_lim_tensorflow, audio.py:90-103) */
} ns_griffin_lim_params;
int ns_griffin_lim(const ns_griffin_lim_params* p, ns_stream_t stream);
size_t ns_griffin_lim_work_bytes(const ns_griffin_lim_params* p);

/* The transforms behind the features and the vocoder as calls of their own (audio.py:106-123).
 * ns_stft_tf: tf.contrib.signal.stft(signals, win, hop, n_fft, pad_end=False) of ONE signal: frames of `win` samples
 *   every `hop`, periodic Hann, zero-padded at the end to n_fft; out [T, n_fft/2+1] (re, im), T = 1 + (L - win) / hop.
 * ns_istft: the inverse of a one-sided spectrum spec [T, n_fft/2+1] (re, im).
 *   center == 0: tf.contrib.signal.inverse_stft(stfts, win, hop, n_fft): irfft[:win] x Hann, overlap-add, no
 *                window-sum normalisation; wav [(T-1) hop + win].
 *   center == 1: librosa 0.6.0 istft(hop, win): the window centred in the n_fft frame, overlap-add, divided by the
 *                summed squared window where that exceeds FLT_MIN, n_fft/2 trimmed at both ends; wav [(T-1) hop].
 *   work: T * win floats. */
typedef struct {
  const float* wav; int L;
  int n_fft, hop, win, T;
  const float* window; const float* twiddle;
  float* out;
} ns_stft_tf_params;
int ns_stft_tf(const ns_stft_tf_params* p, ns_stream_t stream);
typedef struct {
  const float* spec; int T;
  int n_fft, hop, win, center;
  const float* window; const float* twiddle;
  float* wav; float* work;
} ns_istft_params;
int ns_istft(const ns_istft_params* p, ns_stream_t stream);

/* Element-wise conversions of audio.py:150-171 on device arrays: mode 0 _amp_to_db 20 log10(max(1e-5, x)); 1 _db_to_amp
 * 10^(x / 20); 2 _normalize clip((x - min_level_db) / -min_level_db, 0, 1); 3 _denormalize clip(x, 0, 1) * -min_level_db
 * + min_level_db. */
typedef struct { const float* x; float* y; int64_t n; int mode; float min_level_db; } ns_audio_pointwise_params;
int ns_audio_pointwise(const ns_audio_pointwise_params* p, ns_stream_t stream);

/* audio.preemphasis (inverse=0: y[n] = x[n] - c x[n-1]) and audio.inv_preemphasis
 * (inverse=1: y[n] = x[n] + c y[n-1]), scipy.signal.lfilter with zero initial state. */
typedef struct { const float* x; float* y; int64_t n; float coef; int inverse; } ns_preemphasis_params;
int ns_preemphasis(const ns_preemphasis_params* p, ns_stream_t stream);

/* ---------------------------------------------------------------- FLAC input (host code, flac.hip)
 * The reference reads LibriSpeech's .flac files through librosa.core.load (datasets/corpus/ljspeech.py:17,
 * utils/audio.py:13-14).  data = the whole file in HOST memory.  ns_flac_info: stream parameters (total_samples per
 * channel, 0 when the encoder left it out; md5_16 = STREAMINFO's MD5 of the decoded PCM, may be NULL).
 * ns_flac_decode: out[sample * channels + channel] int32 in HOST memory, capacity / *decoded in samples per channel;
 * every frame is checked against its CRC-8 and CRC-16.  NS_ERR_SHORT_BUFFER: the stream holds more than `capacity`
 * samples (only possible when STREAMINFO carries no total). */
int ns_flac_info(const uint8_t* data, size_t n, int* sample_rate, int* channels, int* bits_per_sample,
                 int64_t* total_samples, uint8_t* md5_16);
int ns_flac_decode(const uint8_t* data, size_t n, int32_t* out, int64_t capacity, int64_t* decoded);

#ifdef __cplusplus
}
#endif
#endif
