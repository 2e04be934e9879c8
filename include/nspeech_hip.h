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

enum { NS_OK = 0, NS_ERR_BAD_ARG = -1, NS_ERR_UNSUPPORTED_SHAPE = -2, NS_ERR_LAUNCH = -3 };
enum { NS_F32 = 0, NS_BF16 = 1 };
enum { NS_ACT_NONE = 0, NS_ACT_RELU = 1, NS_ACT_TANH = 2, NS_ACT_SIGMOID = 3 };

int ns_version(void);
const char* ns_device_arch(void); /* "gfx950" */
const char* ns_last_error(void);

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
 *   col_sum/col_sumsq: optional fp32[N]; += sum / sum of squares of the stored values
 *             over unmasked rows (BatchNorm batch statistics, modules.py:198).
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
} ns_gemm_params;
int ns_gemm(const ns_gemm_params* p, ns_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
