// Element-wise pieces of the simple WaveNet (neural_speech/models/wavenet_simple.py); the convolutions
// themselves are ns_gemm launches (a dilated width-2 VALID convolution is two accumulated GEMMs whose A
// operands are the same [N*T, C] buffer shifted by `dilation` rows).  All series live on ONE time grid of T
// rows per batch item, right-aligned: layer l's outputs are valid from row t >= start_l, rows before it are
// never read by a valid output.
//   ns_wavenet_input      one-hot causal layer as two table look-ups           (wavenet_simple.py:246-252, 385-397)
//   ns_wavenet_gate       tanh(filter) * sigmoid(gate) and its gradient         (:325)
//   ns_wavenet_softmax_ce mean softmax cross-entropy against integer targets    (:479-502) and d/dlogits
//   ns_wavenet_generate   incremental sample-by-sample generation (persistent)  (generate_wavenet.py:56-142)
#include "common.h"

// ------------------------------------------------------------------ input layer
// x0[n,t,:] = W[0][ids[n,t-1]] + W[1][ids[n,t]]  for 1 <= t < T   (row t = 0 is written as zero)
template <typename T>
__global__ void wn_input_fwd_kernel(ns_wavenet_input_params p) {
  const long total = (long)p.N * p.T * p.C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % p.C;
    const long row = i / p.C;
    const int t = row % p.T;
    float v = 0.f;
    if (t >= 1) {
      const int a = p.ids[row - 1], b = p.ids[row];
      v = p.w[(long)a * p.C + c] + p.w[((long)p.Q + b) * p.C + c];
    }
    stf((T*)p.x + i, v);
  }
}
// dW[0][ids[t-1]] += dx[t], dW[1][ids[t]] += dx[t]   for start <= t < T
__global__ void wn_input_bwd_kernel(ns_wavenet_input_params p) {
  const long total = (long)p.N * p.T * p.C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % p.C;
    const long row = i / p.C;
    const int t = row % p.T;
    if (t < 1 || t < p.start) continue;
    const float g = p.dx_dtype == NS_BF16 ? (float)((const bf16_t*)p.dx)[i] : ((const float*)p.dx)[i];
    atomicAdd(p.dw + (long)p.ids[row - 1] * p.C + c, g);
    atomicAdd(p.dw + ((long)p.Q + p.ids[row]) * p.C + c, g);
  }
}
extern "C" int ns_wavenet_input(const ns_wavenet_input_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->ids && p->N > 0 && p->T > 1 && p->C > 0 && p->Q > 0, "ns_wavenet_input: bad arguments");
  const long total = (long)p->N * p->T * p->C;
  const int grid = (int)min((long)8192, (total + 255) / 256);
  if (p->dx) {
    NS_CHECK_ARG(p->dw, "ns_wavenet_input: backward needs dw");
    hipLaunchKernelGGL(wn_input_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  } else {
    NS_CHECK_ARG(p->w && p->x && (p->dtype == NS_F32 || p->dtype == NS_BF16), "ns_wavenet_input: forward needs w, x");
    if (p->dtype == NS_BF16) hipLaunchKernelGGL(wn_input_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
    else hipLaunchKernelGGL(wn_input_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  }
  NS_CHECK_LAUNCH("wavenet_input");
  return NS_OK;
}

// ------------------------------------------------------------------ gated activation
// z [rows, 2C] fp32 = [filter | gate] pre-activations.  forward: out[rows, C] (ld_out) = tanh(zf) * sigmoid(zg);
// backward (dout given): dz[rows, 2C] = [dout * sg * (1 - th^2) | dout * th * sg * (1 - sg)].
// Rows with t < start (t = row % T) are written as zero so that later GEMMs over the whole buffer stay finite.
template <typename T, typename TD>
__global__ void wn_gate_kernel(ns_wavenet_gate_params p) {
  const long total = (long)p.rows * p.C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % p.C;
    const long row = i / p.C;
    const bool valid = (row % p.T) >= p.start;
    const float zf = p.z[row * 2 * p.C + c], zg = p.z[row * 2 * p.C + p.C + c];
    const float th = tanhf(zf), sg = 1.f / (1.f + expf(-zg));
    if (p.dout) {
      const float g = valid ? ldf((const TD*)p.dout + row * p.ld_dout + c) : 0.f;
      stf((TD*)p.dz + row * 2 * p.C + c, g * sg * (1.f - th * th));
      stf((TD*)p.dz + row * 2 * p.C + p.C + c, g * th * sg * (1.f - sg));
    } else {
      stf((T*)p.out + row * p.ld_out + c, valid ? th * sg : 0.f);
    }
  }
}
extern "C" int ns_wavenet_gate(const ns_wavenet_gate_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->z && p->rows > 0 && p->C > 0 && p->T > 0, "ns_wavenet_gate: bad arguments");
  NS_CHECK_ARG(p->dtype == NS_F32 || p->dtype == NS_BF16, "ns_wavenet_gate: bad dtype");
  const long total = (long)p->rows * p->C;
  const int grid = (int)min((long)16384, (total + 255) / 256);
  if (p->dout) {
    NS_CHECK_ARG(p->dz, "ns_wavenet_gate: backward needs dz");
    if (p->dtype == NS_BF16) hipLaunchKernelGGL((wn_gate_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
    else hipLaunchKernelGGL((wn_gate_kernel<float, float>), dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  } else {
    NS_CHECK_ARG(p->out, "ns_wavenet_gate: forward needs out");
    if (p->dtype == NS_BF16) hipLaunchKernelGGL((wn_gate_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
    else hipLaunchKernelGGL((wn_gate_kernel<float, float>), dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  }
  NS_CHECK_LAUNCH("wavenet_gate");
  return NS_OK;
}

// ------------------------------------------------------------------ softmax cross-entropy
// One wave per row: loss_acc += (logsumexp(logits) - logits[target]) * scale; dlogits = (softmax - onehot) * scale.
__global__ __launch_bounds__(256) void wn_softmax_ce_kernel(ns_wavenet_ce_params p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const float* lg = p.logits + row * p.ld;
  float m = -3.0e38f;
  for (int c = lane; c < p.Q; c += 64) m = fmaxf(m, lg[c]);
  m = wave_max(m);
  float se = 0.f;
  for (int c = lane; c < p.Q; c += 64) se += expf(lg[c] - m);
  se = wave_sum(se);
  const int tgt = p.targets[row];
  if (lane == 0) atomicAdd(p.loss_acc, (logf(se) + m - lg[tgt]) * p.scale);
  if (p.dlogits) {
    const float inv = 1.f / se;
    for (int c = lane; c < p.Q; c += 64) {
      const float pr = expf(lg[c] - m) * inv;
      const float g = (pr - (c == tgt ? 1.f : 0.f)) * p.scale;
      if (p.d_dtype == NS_BF16) ((bf16_t*)p.dlogits)[row * p.ld_d + c] = (bf16_t)g;
      else ((float*)p.dlogits)[row * p.ld_d + c] = g;
    }
  }
}
extern "C" int ns_wavenet_softmax_ce(const ns_wavenet_ce_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->logits && p->targets && p->loss_acc && p->rows > 0 && p->Q > 0, "ns_wavenet_softmax_ce: bad arguments");
  hipLaunchKernelGGL(wn_softmax_ce_kernel, dim3((unsigned)((p->rows + 3) / 4)), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("wavenet_softmax_ce");
  return NS_OK;
}
