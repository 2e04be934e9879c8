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

// ------------------------------------------------------------------ incremental generation
// One workgroup per waveform walks the samples one by one (generate_wavenet.py:56-142 with per-layer queues): for
// sample t the causal layer, the L dilated layers (each reads the value its own input had `dilation` steps ago from
// a ring in global memory, L2 resident), the skip sum and the two post-processing layers, then - once the seed is
// used up - a draw from the softmax by inverse CDF on a caller-supplied uniform number.  The state after the seed
// equals what the full network computes on the same history, so the samples equal a sliding-window predict_proba.
constexpr int GEN_THREADS = 512;
template <typename W>
__global__ __launch_bounds__(GEN_THREADS) void wn_generate_kernel(ns_wavenet_generate_params p) {
  extern __shared__ float gsm[];
  const int R = p.R, Dc = p.Dc, S = p.S, Q = p.Q;
  float* xin = gsm;                 // [2R]: x[t-d] | x[t]
  float* z = xin + 2 * R;           // [2Dc]
  float* out = z + 2 * Dc;          // [Dc]
  float* skip = out + Dc;           // [S]
  float* h1 = skip + S;             // [S]
  float* lg = h1 + S;               // [Q]
  const int tid = threadIdx.x, b = blockIdx.x;
  const W* wb = (const W*)p.weights;
  int* ids = p.ids + (long)b * p.total;
  float* queues = p.queues + (long)b * p.queue_rows * R;
  const float* un = p.uniform + (long)b * (p.total - p.n_seed);
  for (int t = 1; t < p.total; ++t) {
    const bool emit = t + 1 >= p.n_seed && t + 1 < p.total;      // sample t+1 must be drawn
    // causal layer
    if (tid < R) {
      const int a = ids[t - 1], c = ids[t];
      xin[R + tid] = ldf(wb + p.off_causal + (long)a * R + tid) + ldf(wb + p.off_causal + ((long)Q + c) * R + tid);
    }
    for (int j = tid; j < S; j += GEN_THREADS) skip[j] = 0.f;
    long qrow = 0;
    for (int l = 0; l < p.L; ++l) {
      const int d = p.dilations[l];
      float* ring = queues + (qrow + (t % d)) * R;
      qrow += d;
      __syncthreads();                                         // xin[R..2R) (this layer's input) is complete
      if (tid < R) {
        xin[tid] = ring[tid];                                    // the input of d steps ago
        ring[tid] = xin[R + tid];                                // and the current one takes its slot
      }
      __syncthreads();
      const W* fg = wb + p.off_layer0 + (long)l * p.layer_stride;     // [2][R][2Dc]
      if (tid < 2 * Dc) {
        float acc = 0.f;
        for (int k = 0; k < 2 * R; ++k) acc = fmaf(xin[k], ldf(fg + (long)k * 2 * Dc + tid), acc);
        z[tid] = acc;
      }
      __syncthreads();
      if (tid < Dc) out[tid] = tanhf(z[tid]) * (1.f / (1.f + expf(-z[Dc + tid])));
      __syncthreads();
      if (emit) {                                                // the skip path only matters when a sample is drawn
        const W* sk = wb + p.off_skip + (long)l * Dc * S;
        for (int j = tid; j < S; j += GEN_THREADS) {
          float acc = skip[j];
          for (int k = 0; k < Dc; ++k) acc = fmaf(out[k], ldf(sk + (long)k * S + j), acc);
          skip[j] = acc;
        }
      }
      float xn = 0.f;
      if (tid < R) {
        const W* de = fg + p.off_dense_in_layer;                 // [Dc][R]
        xn = xin[R + tid];
        for (int k = 0; k < Dc; ++k) xn = fmaf(out[k], ldf(de + (long)k * R + tid), xn);
      }
      __syncthreads();                                           // every reader of xin[R..2R) and out is done
      if (tid < R) xin[R + tid] = xn;                            // input of the next layer
    }
    if (!emit) continue;
    __syncthreads();
    for (int j = tid; j < S; j += GEN_THREADS) skip[j] = fmaxf(skip[j], 0.f);
    __syncthreads();
    for (int j = tid; j < S; j += GEN_THREADS) {
      float acc = 0.f;
      for (int k = 0; k < S; ++k) acc = fmaf(skip[k], ldf(wb + p.off_post1 + (long)k * S + j), acc);
      h1[j] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int j = tid; j < Q; j += GEN_THREADS) {
      float acc = 0.f;
      for (int k = 0; k < S; ++k) acc = fmaf(h1[k], ldf(wb + p.off_post2 + (long)k * Q + j), acc);
      lg[j] = acc;
    }
    __syncthreads();
    if (tid == 0) {
      float m = lg[0];
      for (int j = 1; j < Q; ++j) m = fmaxf(m, lg[j]);
      double se = 0.0;
      for (int j = 0; j < Q; ++j) se += exp((double)lg[j] - (double)m);      // float64 softmax as predict_proba
      const double u = (double)un[t + 1 - p.n_seed] * se;
      double c = 0.0;
      int pick = Q - 1;
      for (int j = 0; j < Q; ++j) {
        c += exp((double)lg[j] - (double)m);
        if (u < c) { pick = j; break; }
      }
      ids[t + 1] = pick;
      if (p.probs) {                                             // optional: the distribution of the LAST drawn sample
        for (int j = 0; j < Q; ++j) p.probs[(long)b * Q + j] = (float)(exp((double)lg[j] - (double)m) / se);
      }
    }
    __syncthreads();
  }
}

extern "C" int ns_wavenet_generate(const ns_wavenet_generate_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->weights && p->ids && p->queues && p->uniform && p->dilations, "ns_wavenet_generate: null");
  NS_CHECK_ARG(p->B > 0 && p->n_seed >= 2 && p->total > p->n_seed && p->L > 0, "ns_wavenet_generate: bad sizes");
  NS_CHECK_ARG(p->R <= 256 && 2 * p->Dc <= GEN_THREADS && p->Q <= 1024, "ns_wavenet_generate: layer too wide");
  const size_t lds = sizeof(float) * (2 * p->R + 3 * p->Dc + 2 * p->S + p->Q);
  NS_CHECK_ARG(lds <= 60 * 1024, "ns_wavenet_generate: state does not fit in LDS");
  if (p->w_dtype == NS_BF16) hipLaunchKernelGGL(wn_generate_kernel<bf16_t>, dim3(p->B), dim3(GEN_THREADS), lds, (hipStream_t)s, *p);
  else hipLaunchKernelGGL(wn_generate_kernel<float>, dim3(p->B), dim3(GEN_THREADS), lds, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("wavenet_generate");
  return NS_OK;
}
