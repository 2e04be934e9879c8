// Element-wise pieces of the simple WaveNet (neural_speech/models/wavenet_simple.py); the convolutions
// themselves are ns_gemm launches (a dilated width-2 VALID convolution is two accumulated GEMMs whose A
// operands are the same [N*T, C] buffer shifted by `dilation` rows).  All series live on ONE time grid of T
// rows per batch item, right-aligned: layer l's outputs are valid from row t >= start_l, rows before it are
// never read by a valid output.
//   ns_wavenet_input      one-hot causal layer as two table look-ups           (wavenet_simple.py:246-252, 385-397)
//   ns_wavenet_gate       tanh(filter) * sigmoid(gate) and its gradient         (:325)
//   ns_wavenet_softmax    float64 softmax of logit rows (predict_proba)                  (:436-453)
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
// ~10^5 rows land on a 2Q x C table: global float atomics on so few addresses run at a fraction of their rate
// (MI355X_MICROARCH.md), so every block first sums its share into a private copy of the table in LDS (when it fits)
// and only then adds the touched entries to the global one.
__global__ __launch_bounds__(256) void wn_input_bwd_kernel(ns_wavenet_input_params p, int use_lds) {
  extern __shared__ float tab[];                       // [2Q * C] when use_lds
  const int tsize = 2 * p.Q * p.C;
  if (use_lds) {
    for (int i = threadIdx.x; i < tsize; i += blockDim.x) tab[i] = 0.f;
    __syncthreads();
  }
  const long total = (long)p.N * p.T * p.C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % p.C;
    const long row = i / p.C;
    const int t = row % p.T;
    if (t < 1 || t < p.start) continue;
    const float g = p.dx_dtype == NS_BF16 ? (float)((const bf16_t*)p.dx)[i] : ((const float*)p.dx)[i];
    const long o0 = (long)p.ids[row - 1] * p.C + c, o1 = ((long)p.Q + p.ids[row]) * p.C + c;
    if (use_lds) {
      atomicAdd(tab + o0, g);
      atomicAdd(tab + o1, g);
    } else {
      atomicAdd(p.dw + o0, g);
      atomicAdd(p.dw + o1, g);
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < tsize; i += blockDim.x) {
      const float v = tab[i];
      if (v != 0.f) atomicAdd(p.dw + i, v);
    }
  }
}
extern "C" int ns_wavenet_input(const ns_wavenet_input_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->ids && p->N > 0 && p->T > 1 && p->C > 0 && p->Q > 0, "ns_wavenet_input: bad arguments");
  const long total = (long)p->N * p->T * p->C;
  const int grid = (int)min((long)8192, (total + 255) / 256);
  if (p->dx) {
    NS_CHECK_ARG(p->dw, "ns_wavenet_input: backward needs dw");
    const size_t tbytes = sizeof(float) * 2 * (size_t)p->Q * p->C;
    if (tbytes <= 64 * 1024) {        // 32 blocks: few adders per address at the final flush
      hipLaunchKernelGGL(wn_input_bwd_kernel, dim3(min(grid, 32)), dim3(256), tbytes, (hipStream_t)s, *p, 1);
    } else {
      hipLaunchKernelGGL(wn_input_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, *p, 0);
    }
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
// One wave per row, rows strided over the grid: loss_acc += (logsumexp(logits) - logits[target]) * scale;
// dlogits = (softmax - onehot) * scale.  The loss is summed per wave and per block first: one atomic per ROW on a
// single address serialises at the memory side (64 000 rows took 0.8 ms for 130 MB of traffic).
__global__ __launch_bounds__(256) void wn_softmax_ce_kernel(ns_wavenet_ce_params p) {
  __shared__ float part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float loss = 0.f;
  for (long row = (long)blockIdx.x * 4 + wave; row < p.rows; row += (long)gridDim.x * 4) {
  const float* lg = p.logits + row * p.ld;
  float m = -3.0e38f;
  for (int c = lane; c < p.Q; c += 64) m = fmaxf(m, lg[c]);
  m = wave_max(m);
  float se = 0.f;
  for (int c = lane; c < p.Q; c += 64) se += expf(lg[c] - m);
  se = wave_sum(se);
  const int tgt = p.targets[row];
  loss += (logf(se) + m - lg[tgt]) * p.scale;            // the same value in every lane
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
  if (lane == 0) part[wave] = loss;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(p.loss_acc, part[0] + part[1] + part[2] + part[3]);
}
extern "C" int ns_wavenet_softmax_ce(const ns_wavenet_ce_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->logits && p->targets && p->loss_acc && p->rows > 0 && p->Q > 0, "ns_wavenet_softmax_ce: bad arguments");
  const unsigned blocks = (unsigned)min((long)1024, (long)((p->rows + 3) / 4));
  hipLaunchKernelGGL(wn_softmax_ce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("wavenet_softmax_ce");
  return NS_OK;
}

// ------------------------------------------------------------------ float64 softmax of logit rows (predict_proba)
// wavenet_simple.py:436-453 casts the last position's logits to float64 before the softmax; one wave per row.
__global__ __launch_bounds__(64) void wn_softmax_f64_kernel(ns_wavenet_softmax_params p) {
  const int lane = threadIdx.x;
  const float* lg = p.logits + (long)blockIdx.x * p.ld;
  float m = -3.0e38f;
  for (int c = lane; c < p.Q; c += 64) m = fmaxf(m, lg[c]);
  m = wave_max(m);
  double se = 0.0;
  for (int c = lane; c < p.Q; c += 64) se += exp((double)lg[c] - (double)m);
  for (int o = 32; o; o >>= 1) se += __shfl_xor(se, o, 64);
  for (int c = lane; c < p.Q; c += 64) p.probs[(long)blockIdx.x * p.Q + c] = (float)(exp((double)lg[c] - (double)m) / se);
}
extern "C" int ns_wavenet_softmax(const ns_wavenet_softmax_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->logits && p->probs && p->rows > 0 && p->Q > 0 && p->ld >= p->Q, "ns_wavenet_softmax: bad arguments");
  hipLaunchKernelGGL(wn_softmax_f64_kernel, dim3(p->rows), dim3(64), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("wavenet_softmax");
  return NS_OK;
}

// ------------------------------------------------------------------ incremental generation
// One workgroup per waveform walks the samples one by one (generate_wavenet.py:56-142 with per-layer queues): for
// sample t the causal layer, the L dilated layers (each reads the value its own input had `dilation` steps ago from
// a ring in global memory, L2 resident), the skip sum and the two post-processing layers, then - once the seed is
// used up - a draw from the softmax by inverse CDF on a caller-supplied uniform number.  The state after the seed
// equals what the full network computes on the same history, so the samples equal a sliding-window predict_proba.
constexpr int GEN_THREADS = 512;
constexpr int GEN_FG = 8, GEN_DE = 2, GEN_SK = 32;     // per-thread weight registers: filter|gate, dense, skip

// this thread's share of one layer's weights; loaded one layer ahead so that the L2 / Infinity-Cache latency of a
// layer's 10 KB sits under the previous layer's barriers instead of in front of its own products
template <typename W>
struct GenLayerW { W fg[GEN_FG]; W de[GEN_DE]; W sk[GEN_SK]; };

template <typename W>
__device__ __forceinline__ void gen_load(GenLayerW<W>& w, const ns_wavenet_generate_params& p, const W* wb, int l, int tid,
                                         bool with_skip) {
  const int R = p.R, Dc = p.Dc, S = p.S;
  const W* fg = wb + p.off_layer0 + (long)l * p.layer_stride;
  const W* de = fg + p.off_dense_in_layer;
  const W* sk = wb + p.off_skip + (long)l * Dc * S;
  const int nfg = 2 * R * 2 * Dc, nde = Dc * R;
#pragma unroll
  for (int q = 0; q < GEN_FG; ++q) {
    const int e = q * GEN_THREADS + tid;
    w.fg[q] = e < nfg ? fg[e] : (W)0.f;
  }
#pragma unroll
  for (int q = 0; q < GEN_DE; ++q) {
    const int e = q * GEN_THREADS + tid;
    w.de[q] = e < nde ? de[e] : (W)0.f;
  }
  if (with_skip) {
#pragma unroll
    for (int k = 0; k < GEN_SK; ++k) w.sk[k] = (k < Dc && tid < S) ? sk[(long)k * S + tid] : (W)0.f;
  }
}

template <typename W>
__global__ __launch_bounds__(GEN_THREADS) void wn_generate_kernel(ns_wavenet_generate_params p) {
  extern __shared__ float gsm[];
  const int R = p.R, Dc = p.Dc, S = p.S, Q = p.Q;
  float* xin = gsm;                       // [2R]: x[t-d] | x[t]
  float* z = xin + 2 * R;                 // [2Dc]
  float* out = z + 2 * Dc;                // [Dc]
  float* part = out + Dc;                 // [GEN_THREADS] partial sums of the layer products
  float* h0 = part + GEN_THREADS;         // [S] relu(skip sum)
  float* h1 = h0 + S;                     // [S]
  float* lg = h1 + S;                     // [Q]
  double* ex = (double*)(lg + ((Q + 1) & ~1));   // [Q]
  const int tid = threadIdx.x, b = blockIdx.x;
  const W* wb = (const W*)p.weights;
  int* ids = p.ids + (long)b * p.total;
  float* queues = p.queues + (long)b * p.queue_rows * R;
  const float* un = p.uniform + (long)b * (p.total - p.n_seed);
  const int nfg = 2 * R * 2 * Dc, nde = Dc * R;
  const int jz = tid % (2 * Dc), kz0 = tid / (2 * Dc), kzs = GEN_THREADS / (2 * Dc);   // fg element q: k = q*kzs + kz0
  const int jd = tid % R, kd0 = tid / R, kds = GEN_THREADS / R;
  __shared__ int dil[128];
  for (int i = tid; i < p.L && i < 128; i += GEN_THREADS) dil[i] = p.dilations[i];
  __syncthreads();
  GenLayerW<W> wa, wbuf;
  for (int t = 1; t < p.total; ++t) {
    const bool emit = t + 1 >= p.n_seed && t + 1 < p.total;      // sample t+1 must be drawn
    gen_load(wa, p, wb, 0, tid, emit);
    if (tid < R) {                                               // causal layer
      const int a = ids[t - 1], c = ids[t];
      xin[R + tid] = ldf(wb + p.off_causal + (long)a * R + tid) + ldf(wb + p.off_causal + ((long)Q + c) * R + tid);
    }
    float skip = 0.f;                                            // thread j < S owns skip[j]
    long qrow = 0;
    // ring slot of layer l at time t: holds that layer's input of d steps ago; read one layer ahead like the weights
    float rcur = (tid < R) ? queues[(0 + (t % dil[0])) * R + tid] : 0.f, rnext = 0.f;
    // two layers per trip so that the current / next weight registers are statically known (a run-time choice
    // between the two structs would push both into scratch memory)
    auto layer = [&](int l, GenLayerW<W>& wc, GenLayerW<W>& wn) {
      const int d = dil[l];
      float* ring = queues + (qrow + (t % d)) * R;
      qrow += d;
      if (l + 1 < p.L) {
        if (tid < R) rnext = queues[(qrow + (t % dil[l + 1])) * R + tid];
        gen_load(wn, p, wb, l + 1, tid, emit);
      }
      __syncthreads();                                           // xin[R..2R) (this layer's input) is complete
      if (tid < R) {
        xin[tid] = rcur;                                         // the input of d steps ago
        ring[tid] = xin[R + tid];                                // and the current one takes its slot
      }
      rcur = rnext;
      __syncthreads();
      {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < GEN_FG; ++q)
          if (q * GEN_THREADS + tid < nfg) acc = fmaf(xin[q * kzs + kz0], (float)wc.fg[q], acc);
        part[tid] = acc;
      }
      __syncthreads();
      if (tid < 2 * Dc) {
        float v = p.cond ? p.cond[((long)b * p.L + l) * 2 * Dc + tid] : 0.f;      // condition + filter | gate bias
        for (int i = 0; i < kzs; ++i) v += part[i * 2 * Dc + tid];
        z[tid] = v;
      }
      __syncthreads();
      if (tid < Dc) out[tid] = tanhf(z[tid]) * (1.f / (1.f + expf(-z[Dc + tid])));
      __syncthreads();
      if (emit) {
#pragma unroll
        for (int k = 0; k < GEN_SK; ++k)
          if (k < Dc) skip = fmaf(out[k], (float)wc.sk[k], skip);
      }
      {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < GEN_DE; ++q)
          if (q * GEN_THREADS + tid < nde) acc = fmaf(out[q * kds + kd0], (float)wc.de[q], acc);
        part[tid] = acc;
      }
      __syncthreads();
      float xn = 0.f;
      if (tid < R) {
        xn = xin[R + tid] + (p.dense_bias ? p.dense_bias[(long)l * R + tid] : 0.f);
        for (int i = 0; i < kds; ++i) xn += part[i * R + tid];
      }
      __syncthreads();                                           // every reader of xin[R..2R), out and part is done
      if (tid < R) xin[R + tid] = xn;                            // input of the next layer
    };
    for (int l = 0; l < p.L; l += 2) {
      layer(l, wa, wbuf);
      if (l + 1 < p.L) layer(l + 1, wbuf, wa);
    }
    if (!emit) continue;
    if (tid < S) h0[tid] = fmaxf(skip + (p.skip_bias ? p.skip_bias[tid] : 0.f), 0.f);
    __syncthreads();
    for (int j = tid; j < S; j += GEN_THREADS) {
      float acc = p.post1_bias ? p.post1_bias[j] : 0.f;
      for (int k = 0; k < S; ++k) acc = fmaf(h0[k], ldf(wb + p.off_post1 + (long)k * S + j), acc);
      h1[j] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int j = tid; j < Q; j += GEN_THREADS) {
      float acc = p.post2_bias ? p.post2_bias[j] : 0.f;
      for (int k = 0; k < S; ++k) acc = fmaf(h1[k], ldf(wb + p.off_post2 + (long)k * Q + j), acc);
      lg[j] = acc;
    }
    __syncthreads();
    float m = -3.0e38f;
    for (int j = tid; j < Q; j += GEN_THREADS) m = fmaxf(m, lg[j]);
    m = block_max(m, part);
    for (int j = tid; j < Q; j += GEN_THREADS) ex[j] = exp((double)lg[j] - (double)m);   // float64 softmax as predict_proba
    __syncthreads();
    if (tid == 0) {
      double se = 0.0;
      for (int j = 0; j < Q; ++j) se += ex[j];
      const double u = (double)un[t + 1 - p.n_seed] * se;
      double c = 0.0;
      int pick = Q - 1;
      for (int j = 0; j < Q; ++j) {
        c += ex[j];
        if (u < c) { pick = j; break; }
      }
      ids[t + 1] = pick;
      if (p.probs)                                               // the distribution of the LAST drawn sample
        for (int j = 0; j < Q; ++j) p.probs[(long)b * Q + j] = (float)(ex[j] / se);
    }
    __syncthreads();
  }
}

// ---- fast variant (bf16 weights, R == Dc == C): the residual chain of a sample - 50 dependent layers of tiny
// matrix-vector products - runs inside ONE wavefront with no barrier at all: lane j owns column j, the input vector is
// broadcast lane by lane (v_readlane), and the layer's weights come from a transposed bf16 shadow ([column][k], so a
// lane's whole column is 8 + 4 sixteen-byte loads) fetched THREE layers ahead into packed registers, which covers the
// L2 latency.  The other seven waves only join for the skip / post-processing products of a drawn sample.
template <int C>
struct GenPk { unsigned fg[C]; unsigned de[C / 2]; float ring; };   // 2C + C bf16 values, packed in pairs

template <int C>
__device__ __forceinline__ void genpk_load(GenPk<C>& w, const ns_wavenet_generate_params& p, int l, int lane, const float* queues,
                                           long qrow, int t, int d) {
  const uint4* fg = (const uint4*)((const bf16_t*)p.fgT + ((long)l * 2 * C + (lane < 2 * C ? lane : 0)) * 2 * C);
#pragma unroll
  for (int i = 0; i < C / 4; ++i) {
    const uint4 v = fg[i];
    w.fg[4 * i] = v.x; w.fg[4 * i + 1] = v.y; w.fg[4 * i + 2] = v.z; w.fg[4 * i + 3] = v.w;
  }
  const uint4* de = (const uint4*)((const bf16_t*)p.deT + ((long)l * C + (lane < C ? lane : 0)) * C);
#pragma unroll
  for (int i = 0; i < C / 8; ++i) {
    const uint4 v = de[i];
    w.de[4 * i] = v.x; w.de[4 * i + 1] = v.y; w.de[4 * i + 2] = v.z; w.de[4 * i + 3] = v.w;
  }
  w.ring = lane < C ? queues[(qrow + (t % d)) * C + lane] : 0.f;
}
__device__ __forceinline__ float pk_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float pk_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float lane_bcast(float v, int k) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}

// y[0..N) = act(sum_k x[k] * Wm[k*N + n]) for a bf16 [K, N] matrix streamed from L2 once: a thread owns 8 adjacent
// columns (one 16-byte load per row) and the workgroup splits K, 8 rows in flight per thread; the partial sums meet
// in LDS.  x, y and part are LDS; part needs (GEN_THREADS / (N/8)) * N floats.  N % 8 == 0, N/8 <= GEN_THREADS.
template <int RIF = 8>      // weight rows in flight per thread (16 measured no faster for the post-processing products: 43.5 vs 43.2 us per sample)
__device__ __forceinline__ void gen_matvec8(const bf16_t* Wm, int K, int N, const float* x, float* y, float* part, bool relu,
                                            int tid) {
  const int ncg = N >> 3, nks = GEN_THREADS / ncg;
  const int cg = tid % ncg, ks = tid / ncg;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (ks < nks) {
    const uint4* col = (const uint4*)(Wm + cg * 8);
    for (int k0 = ks; k0 < K; k0 += nks * RIF) {
      uint4 w[RIF];
#pragma unroll
      for (int q = 0; q < RIF; ++q) {
        const int k = k0 + q * nks;
        w[q] = k < K ? col[(long)k * ncg] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int q = 0; q < RIF; ++q) {
        const int k = k0 + q * nks;
        const float xv = k < K ? x[k] : 0.f;
        acc[0] = fmaf(xv, pk_lo(w[q].x), acc[0]); acc[1] = fmaf(xv, pk_hi(w[q].x), acc[1]);
        acc[2] = fmaf(xv, pk_lo(w[q].y), acc[2]); acc[3] = fmaf(xv, pk_hi(w[q].y), acc[3]);
        acc[4] = fmaf(xv, pk_lo(w[q].z), acc[4]); acc[5] = fmaf(xv, pk_hi(w[q].z), acc[5]);
        acc[6] = fmaf(xv, pk_lo(w[q].w), acc[6]); acc[7] = fmaf(xv, pk_hi(w[q].w), acc[7]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) part[ks * N + cg * 8 + i] = acc[i];
  }
  __syncthreads();
  for (int n = tid; n < N; n += GEN_THREADS) {
    float v = 0.f;
    for (int i = 0; i < nks; ++i) v += part[i * N + n];
    y[n] = relu ? fmaxf(v, 0.f) : v;
  }
  __syncthreads();
}

template <int C>
__global__ __launch_bounds__(GEN_THREADS) void wn_generate_fast_kernel(ns_wavenet_generate_params p) {
  extern __shared__ float gsm[];
  const int S = p.S, Q = p.Q, L = p.L;
  float* outs = gsm;                       // [L][C] gated outputs of this sample (skip path)
  float* h0 = outs + L * C;                // [S]
  float* h1 = h0 + S;                      // [S]
  float* lg = h1 + S;                      // [Q]
  float* red = lg + ((Q + 1) & ~1);        // [16]
  float* part = red + 16;                  // [GEN_THREADS * 8]: K-split partial sums of the big products
  double* ex = (double*)(part + GEN_THREADS * 8);   // [Q]
  __shared__ int dil[128];
  __shared__ long qoff[128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const bf16_t* wb = (const bf16_t*)p.weights;
  int* ids = p.ids + (long)b * p.total;
  float* queues = p.queues + (long)b * p.queue_rows * C;
  const float* un = p.uniform + (long)b * (p.total - p.n_seed);
  if (tid == 0) {
    long q = 0;
    for (int i = 0; i < L; ++i) { dil[i] = p.dilations[i]; qoff[i] = q; q += p.dilations[i]; }
  }
  __syncthreads();
  for (int t = 1; t < p.total; ++t) {
    const bool emit = t + 1 >= p.n_seed && t + 1 < p.total;
    if (wave == 0) {
      float xcur = 0.f;
      if (lane < C) {
        const int a = ids[t - 1], c = ids[t];
        xcur = (float)wb[p.off_causal + (long)a * C + lane] + (float)wb[p.off_causal + ((long)Q + c) * C + lane];
      }
      GenPk<C> w0, w1, w2;
      genpk_load(w0, p, 0, lane, queues, qoff[0], t, dil[0]);
      if (L > 1) genpk_load(w1, p, 1, lane, queues, qoff[1], t, dil[1]);
      if (L > 2) genpk_load(w2, p, 2, lane, queues, qoff[2], t, dil[2]);
      auto layer = [&](int l, GenPk<C>& w) {
        const float xold = w.ring;
        float z = 0.f;
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const unsigned pw = w.fg[k >> 1];
          z = fmaf(lane_bcast(xold, k), (k & 1) ? pk_hi(pw) : pk_lo(pw), z);
        }
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const unsigned pw = w.fg[(C + k) >> 1];
          z = fmaf(lane_bcast(xcur, k), (k & 1) ? pk_hi(pw) : pk_lo(pw), z);
        }
        const float zg = __shfl(z, (lane + C) & 63, 64);
        const float o = tanhf(z) * (1.f / (1.f + expf(-zg)));          // meaningful in lanes < C
        if (emit && lane < C) outs[l * C + lane] = o;
        float xn = xcur;
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const unsigned pw = w.de[k >> 1];
          xn = fmaf(lane_bcast(o, k), (k & 1) ? pk_hi(pw) : pk_lo(pw), xn);
        }
        if (lane < C) queues[(qoff[l] + (t % dil[l])) * C + lane] = xcur;   // the current input replaces the one just used
        xcur = xn;
        if (l + 3 < L) genpk_load(w, p, l + 3, lane, queues, qoff[l + 3], t, dil[l + 3]);
      };
      for (int l = 0; l < L; l += 3) {
        layer(l, w0);
        if (l + 1 < L) layer(l + 1, w1);
        if (l + 2 < L) layer(l + 2, w2);
      }
    }
    __syncthreads();
    if (!emit) continue;
    // skip sum over all layers as ONE [L*C, S] product, relu, post1, relu, post2
    gen_matvec8(wb + p.off_skip, L * C, S, outs, h0, part, true, tid);
    gen_matvec8(wb + p.off_post1, S, S, h0, h1, part, true, tid);
    gen_matvec8(wb + p.off_post2, S, Q, h1, lg, part, false, tid);
    float m = -3.0e38f;
    for (int j = tid; j < Q; j += GEN_THREADS) m = fmaxf(m, lg[j]);
    m = block_max(m, red);
    for (int j = tid; j < Q; j += GEN_THREADS) ex[j] = exp((double)lg[j] - (double)m);
    __syncthreads();
    if (tid == 0) {
      double se = 0.0;
      for (int j = 0; j < Q; ++j) se += ex[j];
      const double u = (double)un[t + 1 - p.n_seed] * se;
      double c = 0.0;
      int pick = Q - 1;
      for (int j = 0; j < Q; ++j) {
        c += ex[j];
        if (u < c) { pick = j; break; }
      }
      ids[t + 1] = pick;
      if (p.probs)
        for (int j = 0; j < Q; ++j) p.probs[(long)b * Q + j] = (float)(ex[j] / se);
    }
    __syncthreads();
  }
}

// ---- MFMA variant (R == Dc == 32): the residual chain of a sample on the matrix cores of ONE wavefront, the skip
// products of the same sample on the other seven waves while the chain is still running.
//   chain wave : the products are taken TRANSPOSED - the layer's weights are the A operand of
//                v_mfma_f32_16x16x32_bf16 (M = 16 output channels per tile), the activation vector its B operand in
//                column 0 (lanes 0, 16, 32, 48 hold 8 channels each) - so the result comes back in the SAME four lanes
//                (column 0 of D: lane 16 g holds rows 4 g .. 4 g + 3 of every tile).  With the K axis of the weight
//                shadows stored in the order k-slot (g, j) <-> channel 4 g + (j & 3) + 16 (j >> 2) (the host permutes
//                the transposed shadows once), a lane's 8 results of the two tiles of a product ARE its 8 operand
//                slots of the next product: z = [x[t-d] | x[t]] . W (2 k-steps x 4 tiles), gate, 1x1 dense (2 tiles)
//                and the residual run from registers to registers - no LDS round trip in the 50-layer dependency
//                chain (round 2 gathered the operand through an LDS line twice per layer: 1.1 us per layer); the only
//                cross-lane moves are DPP row shifts that deal the 8 gate inputs of a lane to 8 lanes of its row and back,
//                so the transcendental-rate gate math is issued once per layer instead of 8 times.  Weights (10 sixteen-byte loads per lane per layer) and the ring line are fetched two
//                layers ahead; the gated outputs go to LDS for the skip waves as fire-and-forget writes.
//   skip waves : the 32 rows of every layer's [32, S] skip kernel are dealt 5 5 5 5 4 4 4 over the seven waves, a lane
//                owns 8 adjacent columns; they follow the chain through an LDS progress counter and keep three layers
//                of weights in flight, so the skip sum is finished when the chain is.
constexpr int MF_AHEAD = 2;
// DPP moves inside a row of 16 lanes: row_shr:n = 0x110 + n (lane i takes lane i - n; a lane without a source keeps
// `old`), row_shl:n = 0x100 + n (lane i takes lane i + n; without a source: 0)
#define MF_DPP_KEEP(old_, src_, ctrl_) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old_)), __builtin_bit_cast(int, (float)(src_)), (ctrl_), 0xF, 0xF, false))
#define MF_DPP_ZERO(src_, ctrl_) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (float)(src_)), (ctrl_), 0xF, 0xF, true))
// ------------------------------------------------------------------ engine 3: post-processing helpers
// See ns_wavenet_generate_params.post_x.  Per waveform b the exchange region holds, as 8-byte {tag, value} granules,
// h0 [512] (chain workgroup -> helpers) and the helpers' PARTIAL logits [4][256] (helper h -> chain workgroup): helper h
// owns hidden units 128 h .. 128 h + 127 - it forms them from all of h0 (post1, its 128 columns) and, without handing
// them to anyone, their contribution to ALL 256 logits (post2, its 128 rows); the chain workgroup adds the four partial
// vectors in a fixed order.  Two hand-offs per sample.  The tag is the index of the drawn sample (1, 2, ...), so a
// granule is its own flag and nothing is ever reset; one buffer per array is enough: h0 of sample e + 1 is published
// only after every partial logit of sample e has come back.
typedef unsigned long long wn_u64;
constexpr int WN_S = 512, WN_Q = 256, WN_XW = WN_S + NS_WN_HELPERS * WN_Q;       // granules per waveform
__device__ __forceinline__ wn_u64 wn_pack(unsigned tag, float v) { return ((wn_u64)tag << 32) | (wn_u64)__float_as_uint(v); }
__device__ __forceinline__ void wn_put(wn_u64* p, unsigned tag, float v) {
  __hip_atomic_store((NS_GLOBAL wn_u64*)p, wn_pack(tag, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// polls one granule until it carries `tag`; false = gave up (status raised, or `ticks` of the 100 MHz clock went by)
__device__ __forceinline__ bool wn_get(const wn_u64* p, unsigned tag, float& v, int* status, unsigned ticks) {
  unsigned spins = 0, t0 = 0;
  for (;;) {
    const wn_u64 g = __hip_atomic_load((const NS_GLOBAL wn_u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(g >> 32) == tag) { v = __uint_as_float((unsigned)g); return true; }
    if ((++spins & 255u) == 0) {
      if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
      const unsigned now = (unsigned)wall_clock64() | 1u;
      if (t0 == 0u) t0 = now;
      else if (now - t0 > ticks) { atomicExch(status, 1); return false; }
    }
  }
}

// helper h of waveform b = block b * NS_WN_HELPERS + h.  512 threads.  post1: thread (n = tid % 128, kq = tid / 128) keeps
// W1[128 kq .. +128][128 h + n] (64 registers of bf16 pairs); post2: thread (q = tid % 256, kh = tid / 256) keeps
// W2[128 h + 64 kh .. +64][q] (32 registers).
__global__ __launch_bounds__(512) void wn_post_helper_kernel(ns_wavenet_generate_params p) {
  __shared__ float hv[WN_S];
  __shared__ float h1v[128];
  __shared__ float part[512];
  __shared__ int abortf;
  const int tid = threadIdx.x, b = blockIdx.x / NS_WN_HELPERS, h = blockIdx.x % NS_WN_HELPERS;
  int* status = (int*)p.post_x;
  wn_u64* x = (wn_u64*)((char*)p.post_x + 256) + (size_t)b * WN_XW;
  wn_u64 *xh0 = x, *xlg = x + WN_S + h * WN_Q;
  const bf16_t* wb = (const bf16_t*)p.weights;
  const int n1 = tid & 127, kq = tid >> 7, q2 = tid & 255, kh = tid >> 8;
  unsigned w1[64], w2[32];
  {
    const bf16_t* W1 = wb + p.off_post1 + 128 * h + n1;          // [k][n], row stride S
    const bf16_t* W2 = wb + p.off_post2 + q2;                    // [k][q], row stride Q
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      const int k = 128 * kq + 2 * i;
      w1[i] = (unsigned)*(const unsigned short*)(W1 + (long)k * WN_S) | ((unsigned)*(const unsigned short*)(W1 + (long)(k + 1) * WN_S) << 16);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int k = 128 * h + 64 * kh + 2 * i;
      w2[i] = (unsigned)*(const unsigned short*)(W2 + (long)k * WN_Q) | ((unsigned)*(const unsigned short*)(W2 + (long)(k + 1) * WN_Q) << 16);
    }
  }
  if (tid == 0) abortf = 0;
  __syncthreads();
  const int n_emit = p.total - p.n_seed;
  // the first hand-over arrives behind the seed walk (~23 us per seed sample): its bound grows with the seed
  const unsigned first_ticks = NS_SPIN_TICKS + (unsigned)min((long)p.n_seed * 10000L, 3000000000L);
  for (int e = 1; e <= n_emit; ++e) {
    float v = 0.f;
    if (!wn_get(xh0 + tid, (unsigned)e, v, status, e == 1 ? first_ticks : NS_SPIN_TICKS)) abortf = 1;
    hv[tid] = v;
    __syncthreads();
    if (abortf) return;
    {   // post1, own 128 columns: relu(h0 . W1[:, 128 h + n])
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        acc = fmaf(hv[128 * kq + 2 * i], pk_lo(w1[i]), acc);
        acc = fmaf(hv[128 * kq + 2 * i + 1], pk_hi(w1[i]), acc);
      }
      part[tid] = acc;
    }
    __syncthreads();
    if (tid < 128) h1v[tid] = fmaxf((part[tid] + part[128 + tid]) + (part[256 + tid] + part[384 + tid]), 0.f);
    __syncthreads();
    {   // post2, own 128 rows: partial logits of all 256 outputs
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        acc = fmaf(h1v[64 * kh + 2 * i], pk_lo(w2[i]), acc);
        acc = fmaf(h1v[64 * kh + 2 * i + 1], pk_hi(w2[i]), acc);
      }
      part[tid] = acc;
    }
    __syncthreads();
    if (tid < 256) wn_put(xlg + tid, (unsigned)e, part[tid] + part[256 + tid]);
    __syncthreads();                        // part, hv and h1v are free for the next sample
  }
}
extern "C" size_t ns_wavenet_post_bytes(int B) { return 256 + (size_t)(B > 0 ? B : 0) * WN_XW * sizeof(wn_u64); }

struct GenMf { uint4 fg[8]; uint4 de[2]; float4 r0, r1; };

__device__ __forceinline__ bf16x8 mf_pack(float4 a, float4 b) {
  bf16x8 v;
  v[0] = (bf16_t)a.x; v[1] = (bf16_t)a.y; v[2] = (bf16_t)a.z; v[3] = (bf16_t)a.w;
  v[4] = (bf16_t)b.x; v[5] = (bf16_t)b.y; v[6] = (bf16_t)b.z; v[7] = (bf16_t)b.w;
  return v;
}
__device__ __forceinline__ bf16x8 mf_bits(uint4 w) {
  union { uint4 u; bf16x8 b; } c;
  c.u = w;
  return c.b;
}

__global__ __launch_bounds__(GEN_THREADS) void wn_generate_mfma_kernel(ns_wavenet_generate_params p) {
  extern __shared__ float gsm[];
  constexpr int C = 32;
  const int S = p.S, Q = p.Q, L = p.L;
  float* outs = gsm;                       // [L][C] gated outputs of this sample
  float* xg = outs + L * C;                // [C] operand gather line of the chain wave
  float* h0 = xg + C;                      // [S]
  float* h1 = h0 + S;                      // [S]
  float* lg = h1 + S;                      // [Q]
  float* red = lg + ((Q + 1) & ~1);        // [16]
  float* part = red + 16;                  // [GEN_THREADS * 8]
  double* ex = (double*)(part + GEN_THREADS * 8);   // [Q]
  __shared__ int dil[128];
  __shared__ long qoff[128];
  __shared__ int chain_pos;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const bf16_t* wb = (const bf16_t*)p.weights;
  const bf16_t* fgT = (const bf16_t*)p.fgT;
  const bf16_t* deT = (const bf16_t*)p.deT;
  int* ids = p.ids + (long)b * p.total;
  float* queues = p.queues + (long)b * p.queue_rows * C;
  const float* un = p.uniform + (long)b * (p.total - p.n_seed);
  if (tid == 0) {
    long q = 0;
    for (int i = 0; i < L; ++i) { dil[i] = p.dilations[i]; qoff[i] = q; q += p.dilations[i]; }
  }
  __syncthreads();
  const int l15 = lane & 15, kq = lane >> 4;
  const bool afl = l15 == 0;               // lanes that carry row 0 of an A fragment
  // the sample loop is written out once per role: the chain wave and the skip waves then get register allocations of
  // their own (one loop with a role branch inside made the allocator spill ~300 registers, and every scratch access
  // forces vmcnt(0) behind the prefetches); both loops execute the same sequence of workgroup barriers
  auto gen_post = [&](int t) -> bool {
        for (int j = tid; j < S; j += GEN_THREADS) {
          float v = 0.f;
    #pragma unroll
          for (int w = 0; w < 7; ++w) v += part[w * S + j];
          h0[j] = fmaxf(v, 0.f);
        }
        __syncthreads();
        if (p.post_x) {        // engine 3: the helpers of this waveform hold post1 / post2 in registers
          int* status = (int*)p.post_x;
          wn_u64* x = (wn_u64*)((char*)p.post_x + 256) + (size_t)b * WN_XW;
          const unsigned e = (unsigned)(t + 2 - p.n_seed);       // index of the sample being drawn
          wn_put(x + tid, e, h0[tid]);                           // (S == GEN_THREADS == 512)
          {   // thread (q = tid % 256, pair = tid / 256) takes helpers 2 pair and 2 pair + 1: the four partial vectors add up
              // as (h0 + h1) + (h2 + h3), always in that order
            const int q = tid & 255, pr = tid >> 8;
            float v0 = 0.f, v1 = 0.f;
            if (!wn_get(x + WN_S + (2 * pr) * WN_Q + q, e, v0, status, NS_SPIN_TICKS) ||
                !wn_get(x + WN_S + (2 * pr + 1) * WN_Q + q, e, v1, status, NS_SPIN_TICKS)) ns_lds_poke(&chain_pos, -1);
            h1[tid] = v0 + v1;                                   // (h1 [S] is free in this engine)
          }
          __syncthreads();
          if (ns_lds_peek(&chain_pos) < 0) return false;
          if (tid < Q) lg[tid] = h1[tid] + h1[256 + tid];
          __syncthreads();
        } else {
          gen_matvec8(wb + p.off_post1, S, S, h0, h1, part, true, tid);
          gen_matvec8(wb + p.off_post2, S, Q, h1, lg, part, false, tid);
        }
        float m = -3.0e38f;
        for (int j = tid; j < Q; j += GEN_THREADS) m = fmaxf(m, lg[j]);
        m = block_max(m, red);
        for (int j = tid; j < Q; j += GEN_THREADS) ex[j] = exp((double)lg[j] - (double)m);
        __syncthreads();
        if (Q <= GEN_THREADS) {
          // inverse CDF in parallel: inclusive scan of the float64 weights, pick = #{j : cdf[j] <= u} (the first j with
          // u < cdf[j]); `part` is free again after the post-processing products
          double* wsum = (double*)part;                      // [8] wave totals
          int* wcnt = (int*)(wsum + GEN_THREADS / 64);       // [8] per-wave counts
          const double v = tid < Q ? ex[tid] : 0.0;
          double sc = v;
    #pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const double nb = __shfl_up(sc, o, 64);
            if (lane >= o) sc += nb;
          }
          if (lane == 63) wsum[wave] = sc;
          __syncthreads();
          double off = 0.0, se = 0.0;
    #pragma unroll
          for (int w = 0; w < GEN_THREADS / 64; ++w) {
            const double tw = wsum[w];
            if (w < wave) off += tw;
            se += tw;
          }
          const double u = (double)un[t + 1 - p.n_seed] * se;
          const unsigned long long bal = __ballot(tid < Q && off + sc <= u);
          if (lane == 0) wcnt[wave] = __popcll(bal);
          if (p.probs && t + 2 == p.total && tid < Q)        // the distribution of the LAST drawn sample
            p.probs[(long)b * Q + tid] = (float)(v / se);
          __syncthreads();
          if (tid == 0) {
            int cnt = 0;
    #pragma unroll
            for (int w = 0; w < GEN_THREADS / 64; ++w) cnt += wcnt[w];
            ids[t + 1] = min(cnt, Q - 1);
          }
        } else if (tid == 0) {
          double se = 0.0;
          for (int j = 0; j < Q; ++j) se += ex[j];
          const double u = (double)un[t + 1 - p.n_seed] * se;
          double c = 0.0;
          int pick = Q - 1;
          for (int j = 0; j < Q; ++j) {
            c += ex[j];
            if (u < c) { pick = j; break; }
          }
          ids[t + 1] = pick;
          if (p.probs)
            for (int j = 0; j < Q; ++j) p.probs[(long)b * Q + j] = (float)(ex[j] / se);
        }
        __syncthreads();
        return true;
  };
  if (wave == 0) {
    for (int t = 1; t < p.total; ++t) {
      const bool emit = t + 1 >= p.n_seed && t + 1 < p.total;
      if (tid == 0) chain_pos = 0;
      __syncthreads();
      // ================================================================ chain wave
      // ring rows of this sample for every layer, one modulo per lane instead of two scalar division sequences (+ two LDS
      // reads and their waits) per layer inside the chain: lane i holds the row of layers i and i + 64
      const int rr0 = lane < L ? (int)qoff[lane] + t % dil[lane] : 0;
      const int rr1 = lane + 64 < L ? (int)qoff[(lane + 64) & 127] + t % dil[(lane + 64) & 127] : 0;
#define MF_RROW(l_) ((l_) < 64 ? __builtin_amdgcn_readlane(rr0, (l_)) : __builtin_amdgcn_readlane(rr1, (l_) - 64))
      // lane 16 g (the only lanes that carry data; the others hold zeros throughout) keeps channels
      // 4 g + (j & 3) + 16 (j >> 2), j < 8, of the layer input
      float xv[8];
      {
        const int a = ids[t - 1], c = ids[t];
        const bf16_t* w0 = wb + p.off_causal + (long)a * C, *w1 = wb + p.off_causal + ((long)Q + c) * C;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ch = 4 * kq + (j & 3) + 16 * (j >> 2);
          xv[j] = afl ? (float)w0[ch] + (float)w1[ch] : 0.f;
        }
      }
      // (macros, not lambdas over struct references: the weight sets must stay in named registers - an address
      // taken struct goes to scratch memory, and every scratch access drags a vmcnt(0) wait behind the prefetches)
#define MF_LOAD_FG(W_, l_)                                                                                               \
  do {                                                                                                                   \
    const int ll_ = (l_);                                                                                                \
    _Pragma("unroll") for (int sidx = 0; sidx < 2; ++sidx)                                                                \
      _Pragma("unroll") for (int tt = 0; tt < 4; ++tt)                                                                    \
        W_##_fg[sidx * 4 + tt] = *(const uint4*)(fgT + (((long)ll_ * 2 * C + 16 * tt + l15) * 2 * C + 32 * sidx + 8 * kq)); \
  } while (0)
#define MF_LOAD_DE(W_, l_)                                                                                               \
  do {                                                                                                                   \
    const int ll_ = (l_);                                                                                                \
    _Pragma("unroll") for (int tt = 0; tt < 2; ++tt)                                                                      \
      W_##_de[tt] = *(const uint4*)(deT + (((long)ll_ * C + 16 * tt + l15) * C + 8 * kq));                                 \
  } while (0)
#define MF_LOAD_RING(W_, l_)                                                                                             \
  do {                                                                                                                   \
    const float* ring_ = queues + (long)MF_RROW(l_) * C + 8 * kq;                                                         \
    W_##_r0 = afl ? *(const float4*)ring_ : make_float4(0.f, 0.f, 0.f, 0.f);                                              \
    W_##_r1 = afl ? *(const float4*)(ring_ + 4) : make_float4(0.f, 0.f, 0.f, 0.f);                                        \
  } while (0)
#define MF_LOAD(W_, l_) do { MF_LOAD_FG(W_, l_); MF_LOAD_DE(W_, l_); MF_LOAD_RING(W_, l_); } while (0)
#define MF_DECL(W_) uint4 W_##_fg[8], W_##_de[2]; float4 W_##_r0, W_##_r1
      MF_DECL(w0); MF_DECL(w1);
      MF_LOAD(w0, 0);
      if (L > 1) MF_LOAD(w1, 1);
#define MF_LAYER(l_, W_)                                                                                                 \
  do {                                                                                                                   \
    const int l = (l_);                                                                                                  \
    const bf16x8 b_cur = mf_pack(make_float4(xv[0], xv[1], xv[2], xv[3]), make_float4(xv[4], xv[5], xv[6], xv[7]));        \
    const bf16x8 b_old = mf_pack(W_##_r0, W_##_r1);                                                                       \
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;                                             \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_bits(W_##_fg[0]), b_old, acc0, 0, 0, 0);                            \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_bits(W_##_fg[1]), b_old, acc1, 0, 0, 0);                            \
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_bits(W_##_fg[2]), b_old, acc2, 0, 0, 0);                            \
    acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_bits(W_##_fg[3]), b_old, acc3, 0, 0, 0);                            \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_bits(W_##_fg[4]), b_cur, acc0, 0, 0, 0);                            \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_bits(W_##_fg[5]), b_cur, acc1, 0, 0, 0);                            \
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_bits(W_##_fg[6]), b_cur, acc2, 0, 0, 0);                            \
    acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_bits(W_##_fg[7]), b_cur, acc3, 0, 0, 0);                            \
    /* this layer's eight products are issued: their weight registers take the fragments of layer l + 2 now, half a    */  \
    /* layer earlier than the rest of the set (the prefetch distance was one layer's time, about an L2 round trip)    */  \
    if (l + MF_AHEAD < L) MF_LOAD_FG(W_, l + MF_AHEAD);                                                                   \
    /* column 0 of the results: lane 16 g, register r = row 4 g + r of the tile (tiles: filter 0..15, 16..31, gate ..): */  \
    /* its 8 (filter, gate) pairs are dealt to lanes 16 g + s, s < 8, of the DPP row (row_shr:s reaches exactly lane s */    \
    /* from lane 0; a lane below s keeps what it has), so the transcendental-rate gate math runs ONCE per layer on    */    \
    /* 32 lanes instead of 8 times on 4, and row_shl:j brings the 8 results back to lane 16 g                         */    \
    float zf = acc0[0], zg = acc2[0];                                                                                     \
    zf = MF_DPP_KEEP(zf, acc0[1], 0x111); zg = MF_DPP_KEEP(zg, acc2[1], 0x111);                                           \
    zf = MF_DPP_KEEP(zf, acc0[2], 0x112); zg = MF_DPP_KEEP(zg, acc2[2], 0x112);                                           \
    zf = MF_DPP_KEEP(zf, acc0[3], 0x113); zg = MF_DPP_KEEP(zg, acc2[3], 0x113);                                           \
    zf = MF_DPP_KEEP(zf, acc1[0], 0x114); zg = MF_DPP_KEEP(zg, acc3[0], 0x114);                                           \
    zf = MF_DPP_KEEP(zf, acc1[1], 0x115); zg = MF_DPP_KEEP(zg, acc3[1], 0x115);                                           \
    zf = MF_DPP_KEEP(zf, acc1[2], 0x116); zg = MF_DPP_KEEP(zg, acc3[2], 0x116);                                           \
    zf = MF_DPP_KEEP(zf, acc1[3], 0x117); zg = MF_DPP_KEEP(zg, acc3[3], 0x117);                                           \
    const float og = tanhf_(zf) * sigmoidf_(zg);          /* lane 16 g + s: slot s = channel 4 g + (s & 3) + 16 (s >> 2) */ \
    /* for the skip waves (natural channel order); LDS is in order inside a wave: the counter lands behind the data */    \
    if (emit && l15 < 8) outs[l * C + 4 * kq + (l15 & 3) + 16 * (l15 >> 2)] = og;                                         \
    /* an LDS store (a volatile access through a generic pointer compiled to flat_store + s_waitcnt vmcnt(0): every  */     \
    /* layer of a drawn sample then waited for the weight prefetches of the next two)                                 */     \
    if (emit && lane == 0) __hip_atomic_store(&chain_pos, l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);         \
    float4 oa, ob;                                                                                                        \
    oa.x = og;                       oa.y = MF_DPP_ZERO(og, 0x101); oa.z = MF_DPP_ZERO(og, 0x102); oa.w = MF_DPP_ZERO(og, 0x103); \
    ob.x = MF_DPP_ZERO(og, 0x104); ob.y = MF_DPP_ZERO(og, 0x105); ob.z = MF_DPP_ZERO(og, 0x106); ob.w = MF_DPP_ZERO(og, 0x107); \
    const bf16x8 b_out = mf_pack(oa, ob);                                                                                 \
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;                                                                             \
    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_bits(W_##_de[0]), b_out, d0, 0, 0, 0);                                \
    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_bits(W_##_de[1]), b_out, d1, 0, 0, 0);                                \
    if (afl) { /* the current input replaces the one just used (in this lane's slot order) */                             \
      float* ringw = queues + (long)MF_RROW(l) * C + 8 * kq;                                                              \
      *(float4*)ringw = make_float4(xv[0], xv[1], xv[2], xv[3]);                                                          \
      *(float4*)(ringw + 4) = make_float4(xv[4], xv[5], xv[6], xv[7]);                                                    \
    }                                                                                                                     \
    xv[0] += d0[0]; xv[1] += d0[1]; xv[2] += d0[2]; xv[3] += d0[3];                                                       \
    xv[4] += d1[0]; xv[5] += d1[1]; xv[6] += d1[2]; xv[7] += d1[3];                                                       \
    /* (the ring line issued at the top of the layer as well measured slower: 43.2 against 40.8 us per drawn sample)  */  \
    if (l + MF_AHEAD < L) { MF_LOAD_DE(W_, l + MF_AHEAD); MF_LOAD_RING(W_, l + MF_AHEAD); }                              \
  } while (0)
      for (int l4 = 0; l4 < L; l4 += MF_AHEAD) {
        MF_LAYER(l4, w0);
        if (l4 + 1 < L) MF_LAYER(l4 + 1, w1);
      }
#undef MF_LAYER
#undef MF_LOAD
#undef MF_LOAD_FG
#undef MF_LOAD_DE
#undef MF_LOAD_RING
#undef MF_DECL
#undef MF_RROW
      __syncthreads();
      if (!emit) continue;
      if (!gen_post(t)) return;                              // (engine 3: a hand-over timed out)
    }
  } else {
    for (int t = 1; t < p.total; ++t) {
      const bool emit = t + 1 >= p.n_seed && t + 1 < p.total;
      __syncthreads();
      if (emit) {
      // ================================================================ skip waves
      // the 32 rows of a layer's skip kernel over the seven waves: waves 1 - 4 take 5 consecutive rows, waves 5 - 7 take 4
      // (the slowest skip wave bounds a drawn sample once the chain runs from registers: with 8 rows on wave 1 the skip
      // waves took 0.75 us per layer); a lane owns columns 8*lane .. 8*lane+7
      const int nr = wave <= 4 ? 5 : 4;
      const int row0 = wave <= 4 ? 5 * (wave - 1) : 20 + 4 * (wave - 5);
      const bool colok = 8 * lane < S;
#define SK_ROW(i_) (row0 + (i_))
#define SK_LOAD(W_, l_)                                                                                              \
  do {                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 5; ++i)                                                                     \
      W_[i] = (i < nr && colok) ? *(const uint4*)(wb + p.off_skip + ((long)(l_) * C + SK_ROW(i)) * S + 8 * lane)      \
                                : make_uint4(0u, 0u, 0u, 0u);                                                        \
  } while (0)
#define SK_LAYER(l_, W_)                                                                                             \
  do {                                                                                                               \
    const int l = (l_);                                                                                              \
    unsigned spins = 0;                                                                                              \
    while (__hip_atomic_load(&chain_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < l + 1) {                  \
      __builtin_amdgcn_s_sleep(1);                                                                                    \
      if (++spins > (1u << 26)) break;                                                                                \
    }                                                                                                                 \
    const float* ol = outs + l * C;                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                                                   \
      if (i < nr) {                                                                                                   \
        const float xv = ol[SK_ROW(i)];                                                                               \
        a0 = fmaf(xv, pk_lo(W_[i].x), a0); a1 = fmaf(xv, pk_hi(W_[i].x), a1);                                         \
        a2 = fmaf(xv, pk_lo(W_[i].y), a2); a3 = fmaf(xv, pk_hi(W_[i].y), a3);                                         \
        a4 = fmaf(xv, pk_lo(W_[i].z), a4); a5 = fmaf(xv, pk_hi(W_[i].z), a5);                                         \
        a6 = fmaf(xv, pk_lo(W_[i].w), a6); a7 = fmaf(xv, pk_hi(W_[i].w), a7);                                         \
      }                                                                                                               \
    }                                                                                                                 \
    if (l + 3 < L) SK_LOAD(W_, l + 3);                                                                                \
  } while (0)
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
      uint4 s0[5], s1[5], s2[5];
      SK_LOAD(s0, 0);
      if (L > 1) SK_LOAD(s1, 1);
      if (L > 2) SK_LOAD(s2, 2);
      for (int l3 = 0; l3 < L; l3 += 3) {
        SK_LAYER(l3, s0);
        if (l3 + 1 < L) SK_LAYER(l3 + 1, s1);
        if (l3 + 2 < L) SK_LAYER(l3 + 2, s2);
      }
#undef SK_LAYER
#undef SK_LOAD
#undef SK_ROW
      const float acc[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
      if (colok) {
#pragma unroll
        for (int i = 0; i < 8; ++i) part[(wave - 1) * S + 8 * lane + i] = acc[i];
      }
      }
      __syncthreads();
      if (!emit) continue;
      if (!gen_post(t)) return;                              // (engine 3: a hand-over timed out)
    }
  }
}

extern "C" int ns_wavenet_generate(const ns_wavenet_generate_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->weights && p->ids && p->queues && p->uniform && p->dilations, "ns_wavenet_generate: null");
  NS_CHECK_ARG(p->B > 0 && p->n_seed >= 2 && p->total > p->n_seed && p->L > 0 && p->L <= 128, "ns_wavenet_generate: bad sizes");
  NS_CHECK_ARG(GEN_THREADS % (2 * p->Dc) == 0 && GEN_THREADS % p->R == 0 && 2 * p->R * 2 * p->Dc <= GEN_FG * GEN_THREADS &&
                   p->Dc * p->R <= GEN_DE * GEN_THREADS && p->Dc <= GEN_SK && p->S <= GEN_THREADS && p->Q <= 1024,
               "ns_wavenet_generate: channel counts outside the kernel's register plan");
  const size_t lds = sizeof(float) * (2 * p->R + 3 * p->Dc + GEN_THREADS + 2 * p->S + ((p->Q + 1) & ~1)) + sizeof(double) * p->Q;
  NS_CHECK_ARG(lds <= 60 * 1024, "ns_wavenet_generate: state does not fit in LDS");
  const bool full = p->cond || p->dense_bias || p->skip_bias || p->post1_bias || p->post2_bias;
  NS_CHECK_ARG(!full || (!p->fgT && !p->deT), "ns_wavenet_generate: conditions / biases run on the per-layer kernel (fgT, deT = NULL)");
  if (p->fgT && p->deT) {
    NS_CHECK_ARG(p->w_dtype == NS_BF16 && p->R == p->Dc && (p->R == 32 || p->R == 16) && p->S % 8 == 0 && p->Q % 8 == 0 && p->S / 8 <= GEN_THREADS &&
                     GEN_THREADS % (p->S / 8) == 0 && GEN_THREADS % (p->Q / 8) == 0,
                 "ns_wavenet_generate: the single-wave chain needs bf16 weights and R == Dc in {16, 32}");
    const size_t lds2 = sizeof(float) * ((size_t)p->L * p->R + 2 * p->S + ((p->Q + 1) & ~1) + 16 + GEN_THREADS * 8) + sizeof(double) * p->Q;
    NS_CHECK_ARG(lds2 <= 60 * 1024 && ((size_t)p->L * p->R) % 2 == 0, "ns_wavenet_generate: state does not fit in LDS");
    NS_CHECK_ARG(p->engine == 3 || !p->post_x, "ns_wavenet_generate: post_x belongs to engine 3");
    if (p->engine == 2 || p->engine == 3) {
      NS_CHECK_ARG(p->R == 32 && p->S / 8 <= 64 && p->S * 7 <= GEN_THREADS * 8, "ns_wavenet_generate: the MFMA chain needs R == Dc == 32, S <= 512");
      if (p->engine == 3) {
        NS_CHECK_ARG(p->post_x && p->helper_stream && p->helper_stream != s && p->S == WN_S && p->Q == WN_Q &&
                         (((uintptr_t)p->post_x) & 15) == 0 && (long)p->B * (1 + NS_WN_HELPERS) <= ns_device_cus(),
                     "ns_wavenet_generate: engine 3 needs post_x, a helper stream of its own, S = 512, Q = 256 and B * 5 workgroups resident");
        const int zrc = ns_zero_async(p->post_x, (ns_wavenet_post_bytes(p->B) + 15) & ~(size_t)15, (hipStream_t)s);
        if (zrc) return zrc;
        // the helpers start behind the clearing of the exchange region and run beside the chain kernel
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, (hipStream_t)s) != hipSuccess ||
            hipStreamWaitEvent((hipStream_t)p->helper_stream, ev, 0) != hipSuccess) {
          ns_set_error("ns_wavenet_generate: could not order the helper stream behind the call's");
          return NS_ERR_LAUNCH;
        }
        (void)hipEventDestroy(ev);
        hipLaunchKernelGGL(wn_post_helper_kernel, dim3(p->B * NS_WN_HELPERS), dim3(512), 0, (hipStream_t)p->helper_stream, *p);
        NS_CHECK_LAUNCH("wavenet_post_helper");
      }
      const size_t lds3 = lds2 + sizeof(float) * 32;
      hipLaunchKernelGGL(wn_generate_mfma_kernel, dim3(p->B), dim3(GEN_THREADS), lds3, (hipStream_t)s, *p);
      NS_CHECK_LAUNCH("wavenet_generate_mfma");
      return NS_OK;
    }
    if (p->R == 32) hipLaunchKernelGGL(wn_generate_fast_kernel<32>, dim3(p->B), dim3(GEN_THREADS), lds2, (hipStream_t)s, *p);
    else hipLaunchKernelGGL(wn_generate_fast_kernel<16>, dim3(p->B), dim3(GEN_THREADS), lds2, (hipStream_t)s, *p);
    NS_CHECK_LAUNCH("wavenet_generate_fast");
    return NS_OK;
  }
  if (p->w_dtype == NS_BF16) hipLaunchKernelGGL(wn_generate_kernel<bf16_t>, dim3(p->B), dim3(GEN_THREADS), lds, (hipStream_t)s, *p);
  else hipLaunchKernelGGL(wn_generate_kernel<float>, dim3(p->B), dim3(GEN_THREADS), lds, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("wavenet_generate");
  return NS_OK;
}
