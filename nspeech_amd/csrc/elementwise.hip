// Bandwidth-bound kernels around the GEMMs: strided copy/cast, embedding, BatchNorm forward /
// backward (+ activation and bias gradient), column sums, L1 loss + gradient, Adam with global
// norm clipping, transposing casts.  All grid-stride, vectorised where layouts allow.
#include "common.h"
#include <stdlib.h>

__device__ __forceinline__ float ld_dyn(const void* p, int dtype, long i) {
  return dtype == NS_BF16 ? (float)((const bf16_t*)p)[i] : ((const float*)p)[i];
}
__device__ __forceinline__ void st_dyn(void* p, int dtype, long i, float v) {
  if (dtype == NS_BF16) ((bf16_t*)p)[i] = (bf16_t)v;
  else ((float*)p)[i] = v;
}

// ------------------------------------------------------------------ copy3d
__global__ void copy3d_kernel(ns_copy3d_params p) {
  const long total = (long)p.I * p.J * p.Cc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = idx % p.Cc;
    const long ij = idx / p.Cc;
    const int j = ij % p.J;
    const int i = ij / p.J;
    const float v = ld_dyn(p.src, p.src_dtype, i * p.src_si + j * p.src_sj + c);
    const long d = i * p.dst_si + j * p.dst_sj + c;
    if (p.accumulate) st_dyn(p.dst, p.dst_dtype, d, ld_dyn(p.dst, p.dst_dtype, d) + v);
    else st_dyn(p.dst, p.dst_dtype, d, v);
  }
}
extern "C" int ns_copy3d(const ns_copy3d_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->src && p->dst, "ns_copy3d: null");
  const long total = (long)p->I * p->J * p->Cc;
  if (total <= 0) return NS_OK;
  int grid = (int)min((long)4096, (total + 255) / 256);
  hipLaunchKernelGGL(copy3d_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("copy3d");
  return NS_OK;
}

// ------------------------------------------------------------------ embedding
__global__ void embedding_fwd_kernel(ns_embedding_params p) {
  const int row = blockIdx.x;  // n*T + t
  const int n = row / p.T, t = row % p.T;
  int id = p.ids[row];
  id = id < 0 ? 0 : (id >= p.V ? p.V - 1 : id);
  const float* src = p.table + (long)id * p.D;
  const long dst = ((long)n * p.P + p.padl + t) * p.D;
  for (int d = threadIdx.x; d < p.D; d += blockDim.x) st_dyn(p.out, p.out_dtype, dst + d, src[d]);
}
extern "C" int ns_embedding_fwd(const ns_embedding_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->ids && p->table && p->out, "ns_embedding_fwd: null");
  if (p->N * p->T == 0) return NS_OK;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(p->N * p->T), dim3(64), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("embedding_fwd");
  return NS_OK;
}
// Fixed summation order, no float atomics (round 4; the scatter form added rows in whatever order their workgroups ran).
// Workgroup (v, n) gathers utterance n's positions that hold table row v - 256 positions at a time, listed in position
// order by ballot + prefix - and adds its columns of those rows in list order (four row loads in flight).  Its share is
// parked in `work`; the last of the N workgroups of row v to arrive adds the shares in utterance order into dtable.
// (One workgroup per row alone was 0.4 ms at the benchmark shape: the padding id holds a quarter of all positions and
// its workgroup walked them as one dependent chain.)
constexpr int EMB_CNT = 1024;                    // ns_embedding_bwd_params.work: [0, 1024) arrival counters, then [V][N][D]
__global__ __launch_bounds__(256) void embedding_bwd_kernel(ns_embedding_bwd_params p) {
  __shared__ int list[256];
  __shared__ int wcnt[4];
  __shared__ int last;
  const int v = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int MAXD = 4;                       // columns per thread: D <= 1024
  float acc[MAXD] = {0.f, 0.f, 0.f, 0.f};
  const float* rows = p.dout + ((long)n * p.P + p.padl) * p.D;
  for (int base = 0; base < p.T; base += 256) {
    const int t = base + tid;
    bool hit = false;
    if (t < p.T) {
      int id = p.ids[n * p.T + t];
      id = id < 0 ? 0 : (id >= p.V ? p.V - 1 : id);
      hit = id == v;
    }
    const unsigned long long b = __ballot(hit);
    if (lane == 0) wcnt[wave] = __popcll(b);
    __syncthreads();
    int off = __popcll(b & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) off += wcnt[w];
    const int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (hit) list[off] = t;
    __syncthreads();
    for (int e = 0; e < total; e += 4) {
      float x[4][MAXD];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* src = rows + (long)list[min(e + q, total - 1)] * p.D;
#pragma unroll
        for (int i = 0; i < MAXD; ++i) x[q][i] = (e + q < total && tid + 256 * i < p.D) ? src[tid + 256 * i] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < MAXD; ++i) acc[i] += x[q][i];
    }
    __syncthreads();
  }
  float* share = p.work + EMB_CNT + ((long)v * p.N + n) * p.D;
#pragma unroll
  for (int i = 0; i < MAXD; ++i)
    if (tid + 256 * i < p.D) ns_st_sc1(share + tid + 256 * i, acc[i]);
  ns_drain_stores();
  __syncthreads();
  int* counter = (int*)p.work + v;
  if (tid == 0) last = atomicAdd(counter, 1) == p.N - 1;
  __syncthreads();
  if (!last) return;
#pragma unroll
  for (int i = 0; i < MAXD; ++i) {
    if (tid + 256 * i < p.D) {
      float a = 0.f;
      for (int m = 0; m < p.N; ++m) a += ns_ld_sc1(p.work + EMB_CNT + ((long)v * p.N + m) * p.D + tid + 256 * i);
      p.dtable[(long)v * p.D + tid + 256 * i] += a;
    }
  }
  if (tid == 0) *counter = 0;
}
extern "C" size_t ns_embedding_bwd_work_floats(int N, int D, int V) {
  return (size_t)EMB_CNT + (size_t)(V > 0 ? V : 0) * (size_t)(N > 0 ? N : 0) * (size_t)(D > 0 ? D : 0);
}
extern "C" int ns_embedding_bwd(const ns_embedding_bwd_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->ids && p->dout && p->dtable && p->work, "ns_embedding_bwd: null (work: ns_embedding_bwd_work_floats)");
  NS_CHECK_ARG(p->D <= 1024 && p->V >= 1 && p->V <= EMB_CNT, "ns_embedding_bwd: D <= 1024, V <= 1024");
  if (p->N * p->T == 0) return NS_OK;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(p->V, p->N), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("embedding_bwd");
  return NS_OK;
}

// ------------------------------------------------------------------ BatchNorm forward
__global__ void bn_finalize_kernel(ns_bn_fwd_params p) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.C) return;
  float mean, var;
  if (p.training) {
    mean = p.col_sum[c] / p.count;
    var = p.col_sumsq[c] / p.count - mean * mean;
    var = var < 0.f ? 0.f : var;
    if (p.moving_mean) {
      p.moving_mean[c] = p.moving_mean[c] * p.momentum + mean * (1.f - p.momentum);
      p.moving_var[c] = p.moving_var[c] * p.momentum + var * (1.f - p.momentum);
    }
  } else {
    mean = p.moving_mean[c];
    var = p.moving_var[c];
  }
  p.mean_out[c] = mean;
  p.istd_out[c] = rsqrtf(var + p.eps);
}

template <typename T>
__global__ void bn_apply_kernel(ns_bn_fwd_params p) {
  const T* z = (const T*)p.z;
  T* y = (T*)p.y;
  const long total = (long)p.rows * p.C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = idx % p.C;
    const int row = idx / p.C;
    bool valid = true;
    if (p.row_period > 0) {
      const int t = row % p.row_period;
      valid = t >= p.row_lo && t < p.row_hi;
    }
    float v = 0.f;
    if (valid) v = (ldf(z + idx) - p.mean_out[c]) * p.istd_out[c] * p.gamma[c] + p.beta[c];
    stf(y + (p.ld_y ? (long)row * p.ld_y + c : idx), v);
  }
}
// vector form: 4 adjacent channels per thread (16-byte accesses), optional pre-split bf16 outputs
__device__ __forceinline__ float4 ld4f(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ float4 ld4f(const bf16_t* p) {
  const bf16x4 v = *(const bf16x4*)p;
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <typename T>
__global__ __launch_bounds__(256) void bn_apply4_kernel(ns_bn_fwd_params p) {
  const T* z = (const T*)p.z;
  const int C4 = p.C >> 2;
  const long total4 = (long)p.rows * C4;
  for (long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i4 / C4), c = (int)(i4 - (long)row * C4) * 4;
    bool valid = true;
    if (p.row_period > 0) {
      const int t = row % p.row_period;
      valid = t >= p.row_lo && t < p.row_hi;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
      const float4 x = ld4f(z + i4 * 4), m = *(const float4*)(p.mean_out + c), is = *(const float4*)(p.istd_out + c);
      const float4 g = *(const float4*)(p.gamma + c), b = *(const float4*)(p.beta + c);
      v.x = (x.x - m.x) * is.x * g.x + b.x; v.y = (x.y - m.y) * is.y * g.y + b.y;
      v.z = (x.z - m.z) * is.z * g.z + b.z; v.w = (x.w - m.w) * is.w * g.w + b.w;
    }
    if (p.y) {
      const long yo = p.ld_y ? (long)row * p.ld_y + c : i4 * 4;
      if constexpr (sizeof(T) == 4) *(float4*)((float*)p.y + yo) = v;
      else {
        bf16x4 o; o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
        *(bf16x4*)((bf16_t*)p.y + yo) = o;
      }
    }
    if (p.y_hi) {
      const float f[4] = {v.x, v.y, v.z, v.w};
      bf16x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) { hi[j] = (bf16_t)f[j]; lo[j] = (bf16_t)(f[j] - (float)hi[j]); }
      *(bf16x4*)((bf16_t*)p.y_hi + i4 * 4) = hi;
      *(bf16x4*)((bf16_t*)p.y_lo + i4 * 4) = lo;
    }
  }
}

extern "C" int ns_bn_fwd(const ns_bn_fwd_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->z && (p->y || p->y_hi) && p->gamma && p->beta && p->mean_out && p->istd_out, "ns_bn_fwd: null");
  NS_CHECK_ARG(!p->training || (p->col_sum && p->col_sumsq && p->count > 0), "ns_bn_fwd: training needs stats");
  NS_CHECK_ARG(p->training || (p->moving_mean && p->moving_var), "ns_bn_fwd: inference needs moving stats");
  NS_CHECK_ARG((p->y_hi != nullptr) == (p->y_lo != nullptr), "ns_bn_fwd: y_hi and y_lo come together");
  NS_CHECK_ARG(p->ld_y == 0 || (p->ld_y >= p->C && p->y), "ns_bn_fwd: ld_y < C, or no y");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(p->C, 256)), dim3(256), 0, (hipStream_t)s, *p);
  const long total = (long)p->rows * p->C;
  auto al = [](const void* q, int b) { return ((uintptr_t)q % b) == 0; };
  const int esz = p->dtype == NS_BF16 ? 2 : 4;
  const bool vec = p->C % 4 == 0 && p->ld_y % 4 == 0 && al(p->z, 4 * esz) && al(p->y, 4 * esz) && al(p->mean_out, 16) && al(p->istd_out, 16) &&
                   al(p->gamma, 16) && al(p->beta, 16) && al(p->y_hi, 8) && al(p->y_lo, 8);
  if (vec) {
    const int grid = (int)min((long)4096, (total / 4 + 255) / 256);
    if (p->dtype == NS_BF16) hipLaunchKernelGGL(bn_apply4_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
    else hipLaunchKernelGGL(bn_apply4_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
    NS_CHECK_LAUNCH("bn_fwd");
    return NS_OK;
  }
  NS_CHECK_ARG(p->y && !p->y_hi, "ns_bn_fwd: the pre-split outputs need C %% 4 == 0 and 16-byte aligned operands");
  int grid = (int)min((long)8192, (total + 255) / 256);
  if (p->dtype == NS_BF16) hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  else hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("bn_fwd");
  return NS_OK;
}

// ------------------------------------------------------------------ BatchNorm backward
// pass 1: per row block b (gridDim.x <= 32 blocks): work[(2b)*C + c] = sum dy, work[(2b+1)*C + c] = sum dy*xhat over the
// block's valid rows - plain stores, added up in block order by pass 2, so the result does not depend on timing.
constexpr int BN_MAX_BLOCKS = 32;
constexpr int BN_ROWS_PER_BLOCK = 64;

template <typename T>
__global__ void bn_bwd_reduce_kernel(ns_bn_bwd_params p) {
  const T* z = (const T*)p.z;
  const int rpb = (p.rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rpb, r1 = min(p.rows, r0 + rpb);
  for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
    const float mean = p.mean[c], istd = p.istd[c];
    float s1 = 0.f, s2 = 0.f;
    for (int row = r0; row < r1; ++row) {
      if (p.row_period > 0) {
        const int t = row % p.row_period;
        if (t < p.row_lo || t >= p.row_hi) continue;
      }
      const long idx = (long)row * p.C + c;
      const float dy = p.dy[p.ld_dy ? (long)row * p.ld_dy + c : idx];
      const float xh = (ldf(z + idx) - mean) * istd;
      s1 += dy;
      s2 += dy * xh;
    }
    p.work[(long)(2 * blockIdx.x) * p.C + c] = s1;
    p.work[(long)(2 * blockIdx.x + 1) * p.C + c] = s2;
  }
}

// pass 2: dz = gamma*istd*(dy - s1/M - xhat*s2/M); dpre = dz*act'(z); dbias += colsum(dpre)
template <typename T>
__global__ void bn_bwd_apply_kernel(ns_bn_bwd_params p) {
  const T* z = (const T*)p.z;
  T* dpre = (T*)p.dpre;
  const int rpb = (p.rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rpb, r1 = min(p.rows, r0 + rpb);
  const float invM = 1.f / p.count;
  for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
    const float mean = p.mean[c], istd = p.istd[c], g = p.gamma[c];
    float t1 = 0.f, t2 = 0.f;
    for (int b = 0; b < (int)gridDim.x; ++b) { t1 += p.work[(long)(2 * b) * p.C + c]; t2 += p.work[(long)(2 * b + 1) * p.C + c]; }
    const float m1 = t1 * invM, m2 = t2 * invM;
    float sb = 0.f;
    for (int row = r0; row < r1; ++row) {
      const long idx = (long)row * p.C + c;
      bool valid = true;
      if (p.row_period > 0) {
        const int t = row % p.row_period;
        valid = t >= p.row_lo && t < p.row_hi;
      }
      float d = 0.f;
      if (valid) {
        const float zv = ldf(z + idx);
        const float xh = (zv - mean) * istd;
        d = g * istd * (p.dy[p.ld_dy ? (long)row * p.ld_dy + c : idx] - m1 - xh * m2);
        if (p.act == NS_ACT_RELU) d = zv > 0.f ? d : 0.f;
        else if (p.act == NS_ACT_TANH) d *= (1.f - zv * zv);
        else if (p.act == NS_ACT_SIGMOID) d *= zv * (1.f - zv);
        else if (p.act == NS_ACT_SOFTSIGN) d *= (1.f - fabsf(zv)) * (1.f - fabsf(zv));
      }
      // bias gradient is taken on the value the weight-gradient GEMM will read
      if (sizeof(T) == 4 && p.dpre_dtype == NS_BF16) {
        const bf16_t db = (bf16_t)d;
        ((bf16_t*)p.dpre)[idx] = db;
        sb += (float)db;
      } else {
        stf(dpre + idx, d);
        sb += (float)(T)d;
      }
    }
    if (p.dbias) atomicAdd(p.dbias + c, sb);
    if (blockIdx.x == 0) {
      if (p.dgamma) p.dgamma[c] += t2;
      if (p.dbeta) p.dbeta[c] += t1;
    }
  }
}
// Vector forms of the two passes for C % 4 == 0.  A block owns rows/32 rows x 64 channels and reduces through LDS; its
// sums go to its own slot of `work` (no float atomics: they are slow when contended - MI355X_MICROARCH.md, Global
// float atomics - and their arrival order would make the gradient differ from run to run).  A thread owns 4 adjacent
// channels (16-byte loads) of every 16th row of the block and keeps four rows in flight.
__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  const bf16x4 v = *(const bf16x4*)p;
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st4(float* p, float4 v) { *(float4*)p = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
  bf16x4 o;
  o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
  *(bf16x4*)p = o;
}
constexpr int BN4_QUADS = 16;      // channel quads per block (64 channels)
constexpr int BN4_LANES = 16;      // row lanes per block: 16 x 16 = 256 threads
constexpr int BN4_UNROLL = 4;
constexpr int BN4_MAX_RB = 64;     // row blocks: 64 x (C / 64) workgroups = two per CU at C = 512, ~64 KB of loads in flight per CU

// Fallback reduction (no producer statistics): per row block b, work[b*C + c] = sum dy, work[(RB + b)*C + c] = sum dy*xhat
// over the block's valid rows; ns_stats_finalize adds the blocks in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce4_kernel(ns_bn_bwd_params p) {
  __shared__ float red[256][8];
  const T* z = (const T*)p.z;
  const int tid = threadIdx.x, ql = tid % BN4_QUADS, rl = tid / BN4_QUADS;
  const int q = blockIdx.y * BN4_QUADS + ql;
  const bool active = 4 * q < p.C;
  const int rpb = (p.rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rpb, r1 = min(p.rows, r0 + rpb);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    const float4 mean = *(const float4*)(p.mean + 4 * q), istd = *(const float4*)(p.istd + 4 * q);
    for (int base = r0 + rl; base < r1; base += BN4_UNROLL * BN4_LANES) {
      float4 d[BN4_UNROLL], zz[BN4_UNROLL];
      bool ok[BN4_UNROLL];
#pragma unroll
      for (int u = 0; u < BN4_UNROLL; ++u) {
        const int row = base + u * BN4_LANES;
        ok[u] = row < r1;
        if (ok[u] && p.row_period > 0) {
          const int t = row % p.row_period;
          ok[u] = t >= p.row_lo && t < p.row_hi;
        }
        if (ok[u]) {
          const long idx = (long)row * p.C + 4 * q;
          d[u] = *(const float4*)(p.dy + (p.ld_dy ? (long)row * p.ld_dy + 4 * q : idx));
          zz[u] = ld4(z + idx);
        }
      }
#pragma unroll
      for (int u = 0; u < BN4_UNROLL; ++u) {
        if (ok[u]) {
          s1[0] += d[u].x; s2[0] = fmaf(d[u].x, (zz[u].x - mean.x) * istd.x, s2[0]);
          s1[1] += d[u].y; s2[1] = fmaf(d[u].y, (zz[u].y - mean.y) * istd.y, s2[1]);
          s1[2] += d[u].z; s2[2] = fmaf(d[u].z, (zz[u].z - mean.z) * istd.z, s2[2]);
          s1[3] += d[u].w; s2[3] = fmaf(d[u].w, (zz[u].w - mean.w) * istd.w, s2[3]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[tid][i] = s1[i]; red[tid][4 + i] = s2[i]; }
  __syncthreads();
  if (rl == 0 && active) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = 0.f, b = 0.f;
      for (int r = 0; r < BN4_LANES; ++r) { a += red[r * BN4_QUADS + ql][i]; b += red[r * BN4_QUADS + ql][4 + i]; }
      p.work[(long)blockIdx.x * p.C + 4 * q + i] = a;
      p.work[((long)gridDim.x + blockIdx.x) * p.C + 4 * q + i] = b;
    }
  }
}

// dz = gamma*istd*(dy - s1/M - xhat*s2/M); dpre = dz*act'(z) with s1 = sum dy, s2 = sum dy*xhat given per column (by the
// product that formed dy, or by the reduction above).  Block = rows/RB rows x 64 channels; its column sums of dpre (the
// bias gradient, taken on the values the weight-gradient product will read) go to part[blockIdx.x][C], added up in block
// order by the finalize kernel: no float atomics, every output repeats bit for bit.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(ns_bn_bwd_params p, const float* sum_dy, const float* sum_dyxh,
                                                            float* part) {
  __shared__ float red[256][4];
  const T* z = (const T*)p.z;
  const int tid = threadIdx.x, ql = tid % BN4_QUADS, rl = tid / BN4_QUADS;
  const int q = blockIdx.y * BN4_QUADS + ql;
  const bool active = 4 * q < p.C;
  const int rpb = (p.rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rpb, r1 = min(p.rows, r0 + rpb);
  const bool d16 = sizeof(T) == 4 && p.dpre_dtype == NS_BF16;
  const float invM = 1.f / p.count;
  float sb[4] = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    const float4 mean = *(const float4*)(p.mean + 4 * q), istd = *(const float4*)(p.istd + 4 * q);
    const float4 g = *(const float4*)(p.gamma + 4 * q);
    const float4 w1 = *(const float4*)(sum_dy + 4 * q), w2 = *(const float4*)(sum_dyxh + 4 * q);
    const float mu[4] = {mean.x, mean.y, mean.z, mean.w}, is[4] = {istd.x, istd.y, istd.z, istd.w};
    const float gi[4] = {g.x * istd.x, g.y * istd.y, g.z * istd.z, g.w * istd.w};
    const float m1[4] = {w1.x * invM, w1.y * invM, w1.z * invM, w1.w * invM};
    const float m2[4] = {w2.x * invM, w2.y * invM, w2.z * invM, w2.w * invM};
    for (int base = r0 + rl; base < r1; base += BN4_UNROLL * BN4_LANES) {
      float4 d[BN4_UNROLL], zz[BN4_UNROLL];
      bool inr[BN4_UNROLL], ok[BN4_UNROLL];
#pragma unroll
      for (int u = 0; u < BN4_UNROLL; ++u) {
        const int row = base + u * BN4_LANES;
        inr[u] = row < r1;
        ok[u] = inr[u];
        if (ok[u] && p.row_period > 0) {
          const int t = row % p.row_period;
          ok[u] = t >= p.row_lo && t < p.row_hi;
        }
        if (ok[u]) {
          const long idx = (long)row * p.C + 4 * q;
          d[u] = *(const float4*)(p.dy + (p.ld_dy ? (long)row * p.ld_dy + 4 * q : idx));
          zz[u] = ld4(z + idx);
        }
      }
#pragma unroll
      for (int u = 0; u < BN4_UNROLL; ++u) {
        if (!inr[u]) continue;
        const long idx = (long)(base + u * BN4_LANES) * p.C + 4 * q;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (ok[u]) {
          const float dv[4] = {d[u].x, d[u].y, d[u].z, d[u].w}, zv[4] = {zz[u].x, zz[u].y, zz[u].z, zz[u].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float xh = (zv[i] - mu[i]) * is[i];
            float v = gi[i] * (dv[i] - m1[i] - xh * m2[i]);
            if (p.act == NS_ACT_RELU) v = zv[i] > 0.f ? v : 0.f;
            else if (p.act == NS_ACT_TANH) v *= (1.f - zv[i] * zv[i]);
            else if (p.act == NS_ACT_SIGMOID) v *= zv[i] * (1.f - zv[i]);
            else if (p.act == NS_ACT_SOFTSIGN) v *= (1.f - fabsf(zv[i])) * (1.f - fabsf(zv[i]));
            o[i] = v;
          }
        }
        const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
        if (d16) {
          st4((bf16_t*)p.dpre + idx, ov);
#pragma unroll
          for (int i = 0; i < 4; ++i) sb[i] += (float)(bf16_t)o[i];
        } else {
          st4((T*)p.dpre + idx, ov);
#pragma unroll
          for (int i = 0; i < 4; ++i) sb[i] += (float)(T)o[i];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[tid][i] = sb[i];
  __syncthreads();
  if (rl == 0 && active) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = 0.f;
      for (int r = 0; r < BN4_LANES; ++r) a += red[r * BN4_QUADS + ql][i];
      part[(long)blockIdx.x * p.C + 4 * q + i] = a;
    }
  }
}
// 32 columns x 8 row-block lanes per workgroup: lane q adds blocks q, q + 8, ... in order, then the 8 partial sums in a
// fixed tree (a single thread walking 64 dependent-latency loads took 20 us per layer)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(ns_bn_bwd_params p, const float* sum_dy, const float* sum_dyxh,
                                                              const float* part, int nb) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float a = 0.f;
  if (c < p.C && p.dbias)
    for (int b = q; b < nb; b += 8) a += part[(long)b * p.C + c];
  red[q][cl] = a;
  __syncthreads();
  if (q == 0 && c < p.C) {
    if (p.dbias) {
      const float s01 = red[0][cl] + red[1][cl], s23 = red[2][cl] + red[3][cl];
      const float s45 = red[4][cl] + red[5][cl], s67 = red[6][cl] + red[7][cl];
      p.dbias[c] += (s01 + s23) + (s45 + s67);
    }
    if (p.dgamma) p.dgamma[c] += sum_dyxh[c];
    if (p.dbeta) p.dbeta[c] += sum_dy[c];
  }
}

extern "C" int ns_bn_bwd(const ns_bn_bwd_params* p, ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  NS_CHECK_ARG(p && p->dy && p->z && p->dpre && p->mean && p->istd && p->gamma && p->work, "ns_bn_bwd: null");
  NS_CHECK_ARG(!p->sum_dy == !p->sum_dyxh, "ns_bn_bwd: sum_dy and sum_dyxh come together");
  const bool al16 = ((uintptr_t)p->dy % 16 == 0) && ((uintptr_t)p->z % 8 == 0) && ((uintptr_t)p->dpre % 8 == 0) &&
                    ((uintptr_t)p->mean % 16 == 0) && ((uintptr_t)p->istd % 16 == 0) && ((uintptr_t)p->gamma % 16 == 0) &&
                    ((uintptr_t)p->work % 16 == 0) && (p->dtype == NS_BF16 || ((uintptr_t)p->z % 16 == 0 &&
                    (uintptr_t)p->dpre % (p->dpre_dtype == NS_BF16 ? 8 : 16) == 0)) &&
                    (!p->sum_dy || ((uintptr_t)p->sum_dy % 16 == 0 && (uintptr_t)p->sum_dyxh % 16 == 0));
  NS_CHECK_ARG(p->ld_dy == 0 || p->ld_dy >= p->C, "ns_bn_bwd: ld_dy < C");
  if (p->C % 4 == 0 && p->ld_dy % 4 == 0 && al16) {
    // work: [0, C) sum dy | [C, 2C) sum dy*xhat (fallback only) | [2C, 2C + RB*C) bias-gradient partials |
    //       [2C + 64C, 2C + 64C + 2*RB*C) the fallback reduction's partials
    const int rb = max(1, min(BN4_MAX_RB, ceil_div(p->rows, 128)));
    const dim3 grid4(rb, ceil_div(p->C / 4, BN4_QUADS));
    const float* s1 = p->sum_dy;
    const float* s2 = p->sum_dyxh;
    float* part = p->work + 2L * p->C;
    if (!s1) {
      ns_bn_bwd_params q = *p;
      q.work = p->work + (2L + BN4_MAX_RB) * p->C;
      if (p->dtype == NS_BF16) hipLaunchKernelGGL(bn_bwd_reduce4_kernel<bf16_t>, grid4, dim3(256), 0, s, q);
      else hipLaunchKernelGGL(bn_bwd_reduce4_kernel<float>, grid4, dim3(256), 0, s, q);
      NS_CHECK_LAUNCH("bn_bwd_reduce");
      const int rc = ns_stats_finalize(q.work, rb, p->C, p->work, p->work + p->C, s);
      if (rc) return rc;
      s1 = p->work;
      s2 = p->work + p->C;
    }
    if (p->dtype == NS_BF16) hipLaunchKernelGGL(bn_bwd_apply4_kernel<bf16_t>, grid4, dim3(256), 0, s, *p, s1, s2, part);
    else hipLaunchKernelGGL(bn_bwd_apply4_kernel<float>, grid4, dim3(256), 0, s, *p, s1, s2, part);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(p->C, 32)), dim3(256), 0, s, *p, s1, s2, part, rb);
    NS_CHECK_LAUNCH("bn_bwd");
    return NS_OK;
  }
  const int grid = max(1, min(BN_MAX_BLOCKS, ceil_div(p->rows, BN_ROWS_PER_BLOCK)));
  if (p->dtype == NS_BF16) {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, *p);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, *p);
  } else {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(grid), dim3(256), 0, s, *p);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, s, *p);
  }
  NS_CHECK_LAUNCH("bn_bwd");
  return NS_OK;
}

// ------------------------------------------------------------------ column sums
constexpr int CS_ROWS = 32;
constexpr int COLSUM_MAX_BLOCKS = 64;     // row blocks of a launch = partial sums per column
constexpr int COLSUM_CNT = 1024;          // ns_colsum_params.work: [0, 1024) arrival counters (as int), then the partial sums
__global__ void colsum_kernel(ns_colsum_params p) {
  const int r0 = blockIdx.x * CS_ROWS;
  for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
    float v[CS_ROWS];
#pragma unroll
    for (int j = 0; j < CS_ROWS; ++j) v[j] = (r0 + j < p.rows) ? ld_dyn(p.x, p.dtype, (long)(r0 + j) * p.ld + c) : 0.f;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CS_ROWS; ++j) s += v[j];
    atomicAdd(p.out + c, s);
  }
}
// vector form: block = 16 channel quads x 16 row lanes over rows/32 rows, LDS reduction, then at most 32 adders per
// address (contended float atomics collapse, see the BatchNorm backward kernels)
template <typename T>
__global__ __launch_bounds__(256) void colsum4_kernel(ns_colsum_params p) {
  __shared__ float red[256][4];
  const T* x = (const T*)p.x;
  const int tid = threadIdx.x, ql = tid % BN4_QUADS, rl = tid / BN4_QUADS;
  const int q = blockIdx.y * BN4_QUADS + ql;
  const bool active = 4 * q < p.C;
  const int rpb = (p.rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rpb, r1 = min(p.rows, r0 + rpb);
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int CU = 8;          // rows in flight per thread (a bf16 row quad is only 8 bytes)
  if (active) {
    for (int base = r0 + rl; base < r1; base += CU * BN4_LANES) {
      float4 v[CU];
#pragma unroll
      for (int u = 0; u < CU; ++u) {
        const int row = base + u * BN4_LANES;
        v[u] = row < r1 ? ld4(x + (long)row * p.ld + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < CU; ++u) { s4[0] += v[u].x; s4[1] += v[u].y; s4[2] += v[u].z; s4[3] += v[u].w; }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[tid][i] = s4[i];
  __syncthreads();
  if (rl == 0 && active) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = 0.f;
      for (int r = 0; r < BN4_LANES; ++r) a += red[r * BN4_QUADS + ql][i];
      if (4 * q + i < p.C) {
        if (p.work) ns_st_sc1(p.work + COLSUM_CNT + (long)blockIdx.x * p.C + 4 * q + i, a);      // parked: the last row block adds them in order
        else atomicAdd(p.out + 4 * q + i, a);
      }
    }
  }
  if (!p.work) return;
  // fixed-order finish: write-through partials, drained -> this column block's counter; the last one adds partials 0, 1, ...
  __shared__ int last;
  ns_drain_stores();
  __syncthreads();
  int* counter = (int*)p.work + blockIdx.y;
  if (tid == 0) last = atomicAdd(counter, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  // 64 columns x 4 groups of 16 row blocks: every thread has its 16 loads in flight at once and adds them in block order,
  // the four group sums are added in group order (the serial 64-load chain of one thread per column cost 20 us a call)
  const int c0 = blockIdx.y * BN4_QUADS * 4;
  constexpr int GB = COLSUM_MAX_BLOCKS / 4;
  const int cc = tid & 63, grp = tid >> 6;
  float pv[GB];
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    const int b = grp * GB + i;
    pv[i] = (b < (int)gridDim.x && c0 + cc < p.C) ? ns_ld_sc1(p.work + COLSUM_CNT + (long)b * p.C + c0 + cc) : 0.f;
  }
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < GB; ++i) a += pv[i];
  red[tid][0] = a;
  __syncthreads();
  if (tid < 64 && c0 + tid < p.C) p.out[c0 + tid] += ((red[tid][0] + red[64 + tid][0]) + red[128 + tid][0]) + red[192 + tid][0];
  if (tid == 0) *counter = 0;
}
extern "C" size_t ns_colsum_work_floats(int C) { return (size_t)COLSUM_CNT + (size_t)COLSUM_MAX_BLOCKS * (size_t)(C > 0 ? C : 0); }

// the deterministic form for rows the vector kernel cannot take (ragged C or unaligned rows): thread = column, a row
// block per blockIdx.x, same parked partial sums and fixed-order finish
__global__ __launch_bounds__(256) void colsum_det_kernel(ns_colsum_params p) {
  __shared__ int last;
  const int c = blockIdx.y * 256 + threadIdx.x;
  const int rpb = (p.rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rpb, r1 = min(p.rows, r0 + rpb);
  if (c < p.C) {
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a += ld_dyn(p.x, p.dtype, (long)r * p.ld + c);
    ns_st_sc1(p.work + COLSUM_CNT + (long)blockIdx.x * p.C + c, a);
  }
  ns_drain_stores();
  __syncthreads();
  int* counter = (int*)p.work + blockIdx.y;
  if (threadIdx.x == 0) last = atomicAdd(counter, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  if (c < p.C) {
    float a = 0.f;
    for (int b = 0; b < (int)gridDim.x; ++b) a += ns_ld_sc1(p.work + COLSUM_CNT + (long)b * p.C + c);
    p.out[c] += a;
  }
  if (threadIdx.x == 0) *counter = 0;
}

extern "C" int ns_colsum(const ns_colsum_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->x && p->out, "ns_colsum: null");
  if (p->rows <= 0 || p->C <= 0) return NS_OK;
  const int esz = p->dtype == NS_BF16 ? 2 : 4;
  NS_CHECK_ARG(!p->work || p->C <= 64 * COLSUM_CNT, "ns_colsum: the deterministic form takes C <= 65536");
  // a ragged last quad reads into the row's padding (C rounded up to 4 <= ld) and adds only its valid columns
  if ((p->C + 3) / 4 * 4 <= p->ld && p->ld % 4 == 0 && ((uintptr_t)p->x % (4 * esz)) == 0) {
    const dim3 grid(max(1, min(64, ceil_div(p->rows, 128))), ceil_div((p->C + 3) / 4, BN4_QUADS));
    if (p->dtype == NS_BF16) hipLaunchKernelGGL(colsum4_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, *p);
    else hipLaunchKernelGGL(colsum4_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, *p);
    NS_CHECK_LAUNCH("colsum");
    return NS_OK;
  }
  if (p->work) {
    ns_colsum_params q = *p;
    hipLaunchKernelGGL(colsum_det_kernel, dim3(COLSUM_MAX_BLOCKS, ceil_div(p->C, 256)), dim3(256), 0, (hipStream_t)s, q);
    NS_CHECK_LAUNCH("colsum");
    return NS_OK;
  }
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(p->rows, CS_ROWS)), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("colsum");
  return NS_OK;
}

// ------------------------------------------------------------------ L1 loss + gradient
// Wide rows (the linear spectrogram, F = 1025): one wave per row (n, t), rows strided over the grid; a lane walks the
// row's F columns 64 apart (coalesced, no per-element division), eight loads in flight per operand.  Narrow rows (mel,
// F = 80) would leave most of such a wave idle: they keep the flat element loop.
__device__ __forceinline__ void l1_point(const ns_l1_loss_params& p, float pv, float tv, long prow, int f, float& s_all,
                                         float& s_pr) {
  const float d = pv - tv;
  const float ab = fabsf(d);
  s_all += ab;
  const bool pr = f < p.n_prio;
  if (pr) s_pr += ab;
  if (p.dpred) {
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    st_dyn(p.dpred, p.dpred_dtype, prow * p.ldd + f, sg * (p.w_all + (pr ? p.w_prio : 0.f)));
  }
}
template <bool WIDE>
__global__ __launch_bounds__(256) void l1_loss_kernel(ns_l1_loss_params p) {
  __shared__ float red[32];
  float s_all = 0.f, s_pr = 0.f;
  if (WIDE && p.vec4) {
    // wide rows, four columns per lane: prediction rows are 16-byte aligned (padded leading dimension), target rows are
    // not (F = 1025 floats) - their quads come through 4-byte-aligned 16-byte loads, which global memory takes
    struct __attribute__((packed, aligned(4))) f4u { float v[4]; };
    const int lane = threadIdx.x & 63;
    const long rows = (long)p.N * p.T;
    const int nq = p.F >> 2;                       // whole quads; the F % 4 tail columns go to the first lanes
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
      const int t = (int)(row % p.T);
      const long n = row / p.T;
      const long prow = n * p.P + p.padl + t;
      const float* pr_ = p.pred + prow * p.ldp;
      const float* tg_ = p.target + row * p.F;
      for (int q0 = lane; q0 < nq; q0 += 64 * 4) {
        float4 a[4];
        f4u b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = q0 + 64 * j;
          if (q < nq) { a[j] = *(const float4*)(pr_ + 4 * q); b[j] = *(const f4u*)(tg_ + 4 * q); }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = q0 + 64 * j;
          if (q >= nq) continue;
          const float av[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
          float g[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float d = av[i] - b[j].v[i], ab = fabsf(d);
            s_all += ab;
            const bool pr = 4 * q + i < p.n_prio;
            if (pr) s_pr += ab;
            g[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (p.w_all + (pr ? p.w_prio : 0.f));
          }
          if (p.dpred) {
            if (p.dpred_dtype == NS_BF16) {
              bf16x4 o;
#pragma unroll
              for (int i = 0; i < 4; ++i) o[i] = (bf16_t)g[i];
              *(bf16x4*)((bf16_t*)p.dpred + prow * p.ldd + 4 * q) = o;
            } else {
              *(float4*)((float*)p.dpred + prow * p.ldd + 4 * q) = make_float4(g[0], g[1], g[2], g[3]);
            }
          }
        }
      }
      const int f = 4 * nq + lane;
      if (f < p.F) l1_point(p, pr_[f], tg_[f], prow, f, s_all, s_pr);
    }
  } else if (WIDE) {
    const int lane = threadIdx.x & 63;
    const long rows = (long)p.N * p.T;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
      const int t = (int)(row % p.T);
      const long n = row / p.T;
      const long prow = n * p.P + p.padl + t;
      const float* pr_ = p.pred + prow * p.ldp;
      const float* tg_ = p.target + row * p.F;
      for (int f0 = lane; f0 < p.F; f0 += 64 * 8) {
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int f = f0 + 64 * j;
          a[j] = f < p.F ? pr_[f] : 0.f;
          b[j] = f < p.F ? tg_[f] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int f = f0 + 64 * j;
          if (f < p.F) l1_point(p, a[j], b[j], prow, f, s_all, s_pr);
        }
      }
    }
  } else if (p.vec4) {
    // narrow rows, four columns per thread (F % 4 == 0, 16-byte aligned rows, fp32 gradient): 32-bit index arithmetic and
    // 16-byte accesses (the flat loop below divided a 64-bit index twice per ELEMENT: 59 us for the 10 MB mel spectrogram)
    const int q4 = p.F >> 2;
    const unsigned total4 = (unsigned)p.N * (unsigned)p.T * (unsigned)q4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
      const unsigned nt = i / (unsigned)q4, f = (i - nt * (unsigned)q4) * 4u;
      const unsigned n = nt / (unsigned)p.T, t = nt - n * (unsigned)p.T;
      const long prow = (long)n * p.P + p.padl + t;
      const float4 a = *(const float4*)(p.pred + prow * p.ldp + f);
      const float4 b = *(const float4*)(p.target + (long)nt * p.F + f);
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
      float g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = av[j] - bv[j], ab = fabsf(d);
        s_all += ab;
        const bool pr = (int)f + j < p.n_prio;
        if (pr) s_pr += ab;
        g[j] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (p.w_all + (pr ? p.w_prio : 0.f));
      }
      if (p.dpred) *(float4*)((float*)p.dpred + prow * p.ldd + f) = make_float4(g[0], g[1], g[2], g[3]);
    }
  } else {
    const long total = (long)p.N * p.T * p.F;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
      const int f = idx % p.F;
      const long nt = idx / p.F;
      const int t = nt % p.T;
      const long n = nt / p.T;
      const long prow = n * p.P + p.padl + t;
      l1_point(p, p.pred[prow * p.ldp + f], p.target[idx], prow, f, s_all, s_pr);
    }
  }
  s_all = block_sum(s_all, red);
  s_pr = block_sum(s_pr, red);
  if (threadIdx.x == 0) {
    atomicAdd(p.loss_acc, s_all);
    atomicAdd(p.loss_acc + 1, s_pr);
  }
}
extern "C" int ns_l1_loss(const ns_l1_loss_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->pred && p->target && p->loss_acc, "ns_l1_loss: null");
  const long rows = (long)p->N * p->T, total = rows * p->F;
  if (total <= 0) return NS_OK;
  if (p->F >= 256) {
    ns_l1_loss_params q = *p;
    auto al = [](const void* x, int b) { return (((uintptr_t)x) % b) == 0; };
    q.vec4 = p->ldp % 4 == 0 && al(p->pred, 16) && al(p->target, 4) &&
             (!p->dpred || (p->ldd % 4 == 0 && al(p->dpred, p->dpred_dtype == NS_BF16 ? 8 : 16)));
    // every block ends in two float atomics on the SAME two words: 2048 blocks spent ~60 us queueing there (the L2 takes
    // same-address atomics one at a time) - 512 blocks keep the loads in flight that the memory system needs
    hipLaunchKernelGGL(l1_loss_kernel<true>, dim3((int)min((long)512, (rows + 3) / 4)), dim3(256), 0, (hipStream_t)s, q);
  } else {
    ns_l1_loss_params q = *p;
    auto al16 = [](const void* x) { return (((uintptr_t)x) & 15) == 0; };
    q.vec4 = p->F % 4 == 0 && p->ldp % 4 == 0 && al16(p->pred) && al16(p->target) && total / 4 < 0x7fffffffL &&
             (!p->dpred || (p->dpred_dtype == NS_F32 && p->ldd % 4 == 0 && al16(p->dpred)));
    const long items = q.vec4 ? total / 4 : total;
    hipLaunchKernelGGL(l1_loss_kernel<false>, dim3((int)min((long)256, (items + 255) / 256)), dim3(256), 0, (hipStream_t)s, q);
  }
  NS_CHECK_LAUNCH("l1_loss");
  return NS_OK;
}

// ------------------------------------------------------------------ sum of squares / Adam
__global__ void sumsq_kernel(ns_sumsq_params p) {
  __shared__ float red[32];
  __shared__ int last;
  float s = 0.f;
  const long n4 = p.n / 4;
  const float4* x4 = (const float4*)p.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 8 * stride) {      // eight 16-byte loads in flight
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = i + u * stride < n4 ? x4[i + u * stride] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
  }
  for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long)gridDim.x * blockDim.x)
    s += p.x[i] * p.x[i];
  s = block_sum(s, red);
  if (!p.work) {
    if (threadIdx.x == 0) atomicAdd(p.out, s);
    return;
  }
  // deterministic: partials in work[0..grid), arrival counter in work[1024] (as unsigned); the last block sums in order
  unsigned* counter = (unsigned*)(p.work + 1024);
  if (threadIdx.x == 0) {
    __hip_atomic_store(p.work + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = (atomicAdd(counter, 1u) == gridDim.x - 1);
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float t = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x)
    t += __hip_atomic_load(p.work + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t = block_sum(t, red);          // fixed tree: same lanes, same order on every run
  if (threadIdx.x == 0) {
    *p.out += t;
    *counter = 0u;                 // ready for the next call on this stream
  }
}
extern "C" int ns_sumsq(const ns_sumsq_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->x && p->out, "ns_sumsq: null");
  NS_CHECK_ARG((((uintptr_t)p->x) & 15) == 0, "ns_sumsq: x must be 16-byte aligned");
  if (p->n <= 0) return NS_OK;
  // (512 blocks: every block ends in an atomic on one counter, and the L2 takes same-address atomics one at a time)
  int grid = (int)min((long)512, (p->n / 4 + 255) / 256 + 1);
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("sumsq");
  return NS_OK;
}

// ------------------------------------------------------------------ per-segment moments (training summaries)
// One workgroup per segment [offsets[s], offsets[s+1]) of a flat fp32 buffer: sum, sum of squares, min, max.  Fixed
// summation order (lane-strided partial sums, then the block tree), no atomics.  Not on the step's path: the training
// loop calls it every --summary-interval steps (tacotron2.py:163-188: gradient norms and value histograms).
__global__ __launch_bounds__(1024) void segment_stats_kernel(ns_segment_stats_params p) {
  __shared__ float red[32];
  const int sgm = blockIdx.x;
  const long lo = p.offsets[sgm], hi = p.offsets[sgm + 1];
  float s = 0.f, q = 0.f, mn = INFINITY, mx = -INFINITY;
  for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float v = p.x[i];
    s += v; q += v * v; mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
  s = block_sum(s, red);
  q = block_sum(q, red);
  mx = block_max(mx, red);
  mn = -block_max(-mn, red);
  if (threadIdx.x == 0) {
    float* o = p.out + 4 * (long)sgm;
    o[0] = s; o[1] = q; o[2] = mn; o[3] = mx;
  }
}
extern "C" int ns_segment_stats(const ns_segment_stats_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->x && p->offsets && p->out && p->nseg >= 0, "ns_segment_stats: null");
  if (p->nseg == 0) return NS_OK;
  hipLaunchKernelGGL(segment_stats_kernel, dim3(p->nseg), dim3(1024), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("segment_stats");
  return NS_OK;
}

__global__ void adam_kernel(ns_adam_params p) {
  // a timed-out persistent recurrence left an invalid gradient: update nothing (uniform over the grid: every thread
  // reads the same words, which no kernel of this step writes any more)
  bool bad = false;
#pragma unroll
  for (int i = 0; i < 12; ++i)
    if (p.status[i] && __hip_atomic_load(p.status[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) bad = true;
  if (bad) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.skipped) *p.skipped = 1.f;
    return;
  }
  float scale = p.grad_scale;
  if (p.gnorm_sq) {
    const float gn = sqrtf(p.gnorm_sq[0]) * p.grad_scale;
    scale *= p.clip / fmaxf(gn, p.clip);
  }
  bf16_t* sh = (bf16_t*)p.shadow_bf16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long)gridDim.x * blockDim.x) {
    const float g = p.g[i] * scale;
    const float m = p.beta1 * p.m[i] + (1.f - p.beta1) * g;
    const float v = p.beta2 * p.v[i] + (1.f - p.beta2) * g * g;
    const float w = p.p[i] - p.lr_t * m / (sqrtf(v) + p.eps);
    p.m[i] = m;
    p.v[i] = v;
    p.p[i] = w;
    if (sh) sh[i] = (bf16_t)w;
  }
}
extern "C" int ns_adam(const ns_adam_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->p && p->g && p->m && p->v, "ns_adam: null");
  if (p->n <= 0) return NS_OK;
  int grid = (int)min((long)4096, (p->n + 255) / 256);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("adam");
  return NS_OK;
}

// ------------------------------------------------------------------ cast / transpose 2-D
__device__ __forceinline__ void cast2d_put(const ns_cast2d_params& p, long i, float v) {
  if (p.dst) st_dyn(p.dst, p.dst_dtype, i, v);
  if (p.dst_hi) {
    const bf16_t h = (bf16_t)v;
    ((bf16_t*)p.dst_hi)[i] = h;
    ((bf16_t*)p.dst_lo)[i] = (bf16_t)(v - (float)h);
  }
}
__global__ void cast2d_kernel(ns_cast2d_params p) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx over cols, by over rows of src
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    tile[j][tx] = (r < p.rows && c < p.cols) ? p.src[(long)r * p.ld_src + c] : 0.f;
  }
  __syncthreads();
  if (p.transpose) {
    for (int j = ty; j < 32; j += 8) {
      const int c = bx + j, r = by + tx;  // dst[c][r]
      if (r < p.rows && c < p.cols) cast2d_put(p, (long)c * p.ld_dst + r, tile[tx][j]);
    }
  } else {
    for (int j = ty; j < 32; j += 8) {
      const int r = by + j, c = bx + tx;
      if (r < p.rows && c < p.cols) cast2d_put(p, (long)r * p.ld_dst + c, tile[j][tx]);
    }
  }
}
// Vector form: 64 x 64 tiles, 16-byte loads, 4 destination elements per store (16 B fp32 / 8 B bf16).  The weight
// shadows refreshed after every optimiser step are 28 M elements (tacotron2.py refresh_shadows); the scalar form above
// moved them at a fraction of the HBM rate (0.27 ms per step).
__device__ __forceinline__ void cast2d_put4(const ns_cast2d_params& p, long i, const float (&v)[4]) {
  if (p.dst) {
    if (p.dst_dtype == NS_BF16) {
      bf16x4 o; o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
      *(bf16x4*)((bf16_t*)p.dst + i) = o;
    } else {
      *(float4*)((float*)p.dst + i) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  if (p.dst_hi) {
    bf16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) { hi[j] = (bf16_t)v[j]; lo[j] = (bf16_t)(v[j] - (float)hi[j]); }
    *(bf16x4*)((bf16_t*)p.dst_hi + i) = hi;
    *(bf16x4*)((bf16_t*)p.dst_lo + i) = lo;
  }
}
template <bool TRANSPOSE>
__global__ __launch_bounds__(256) void cast2d_vec_kernel(ns_cast2d_params p) {
  __shared__ float tile[TRANSPOSE ? 64 : 1][65];
  const int bx = blockIdx.x * 64, by = blockIdx.y * 64;      // bx over cols, by over rows of src
  const int q = threadIdx.x & 15, l = threadIdx.x >> 4;
  float4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = by + l + 16 * i, c = bx + q * 4;
    v[i] = (r < p.rows && c < p.cols) ? *(const float4*)(p.src + (long)r * p.ld_src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (!TRANSPOSE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = by + l + 16 * i, c = bx + q * 4;
      const float f[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
      if (r < p.rows && c < p.cols) cast2d_put4(p, (long)r * p.ld_dst + c, f);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float* t = &tile[l + 16 * i][q * 4];
    t[0] = v[i].x; t[1] = v[i].y; t[2] = v[i].z; t[3] = v[i].w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = l + 16 * i, r4 = q * 4;               // dst row bx + c holds src rows by + r4 .. + 3
    const float f[4] = {tile[r4][c], tile[r4 + 1][c], tile[r4 + 2][c], tile[r4 + 3][c]};
    if (bx + c < p.cols && by + r4 < p.rows) cast2d_put4(p, (long)(bx + c) * p.ld_dst + by + r4, f);
  }
}
// Many casts in ONE launch (the per-step refresh of the operand-dtype weight shadows: ~25 small transposes that cost a
// launch each): block b of the grid looks its descriptor up in a DEVICE table by the prefix sums of the tile counts and
// runs the body of cast2d_vec_kernel on it.  Every descriptor must qualify for the vector kernel (ns_cast2d_batch checks
// what it can on the host: the table itself lives on the device, so the caller vouches for it via ns_cast2d_batchable).
__global__ __launch_bounds__(256) void cast2d_batch_kernel(const ns_cast2d_params* tab, const int* tile_end, int n) {
  __shared__ float tile[64][65];
  __shared__ int which, first;
  if (threadIdx.x == 0) {
    int lo = 0, hi = n - 1;                     // smallest d with tile_end[d] > blockIdx.x
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (tile_end[mid] > (int)blockIdx.x) hi = mid; else lo = mid + 1; }
    which = lo;
    first = lo ? tile_end[lo - 1] : 0;
  }
  __syncthreads();
  const ns_cast2d_params p = tab[which];
  const int t = (int)blockIdx.x - first, tx = (p.cols + 63) / 64;
  const int bx = (t % tx) * 64, by = (t / tx) * 64;
  const int q = threadIdx.x & 15, l = threadIdx.x >> 4;
  float4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = by + l + 16 * i, c = bx + q * 4;
    v[i] = (r < p.rows && c < p.cols) ? *(const float4*)(p.src + (long)r * p.ld_src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (!p.transpose) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = by + l + 16 * i, c = bx + q * 4;
      const float f[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
      if (r < p.rows && c < p.cols) cast2d_put4(p, (long)r * p.ld_dst + c, f);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float* tt = &tile[l + 16 * i][q * 4];
    tt[0] = v[i].x; tt[1] = v[i].y; tt[2] = v[i].z; tt[3] = v[i].w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = l + 16 * i, r4 = q * 4;
    const float f[4] = {tile[r4][c], tile[r4 + 1][c], tile[r4 + 2][c], tile[r4 + 3][c]};
    if (bx + c < p.cols && by + r4 < p.rows) cast2d_put4(p, (long)(bx + c) * p.ld_dst + by + r4, f);
  }
}
static bool cast2d_vec_ok(const ns_cast2d_params* p) {
  auto al = [](const void* q, int b) { return ((uintptr_t)q % b) == 0; };
  const int dsz = p->dst_dtype == NS_BF16 ? 8 : 16;
  return p->src && (p->dst || p->dst_hi) && !p->dst_hi == !p->dst_lo && p->rows > 0 && p->cols > 0 && p->cols % 4 == 0 &&
         p->ld_src % 4 == 0 && p->ld_dst % 4 == 0 && (!p->transpose || p->rows % 4 == 0) && al(p->src, 16) && al(p->dst, dsz) &&
         al(p->dst_hi, 8) && al(p->dst_lo, 8);
}
extern "C" int ns_cast2d_batchable(const ns_cast2d_params* p) { return p && cast2d_vec_ok(p) ? (p->cols + 63) / 64 * ((p->rows + 63) / 64) : 0; }
extern "C" int ns_cast2d_batch(const ns_cast2d_params* table_dev, const int* tile_end_dev, int n, int total_tiles, ns_stream_t s) {
  NS_CHECK_ARG(table_dev && tile_end_dev && n >= 1 && total_tiles >= 1, "ns_cast2d_batch: null / empty");
  hipLaunchKernelGGL(cast2d_batch_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)s, table_dev, tile_end_dev, n);
  NS_CHECK_LAUNCH("cast2d_batch");
  return NS_OK;
}

extern "C" int ns_cast2d(const ns_cast2d_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->src && (p->dst || p->dst_hi), "ns_cast2d: null");
  NS_CHECK_ARG(!p->dst_hi == !p->dst_lo, "ns_cast2d: dst_hi and dst_lo come as a pair");
  if (p->rows <= 0 || p->cols <= 0) return NS_OK;
  {
    auto al = [](const void* q, int b) { return ((uintptr_t)q % b) == 0; };
    const int dsz = p->dst_dtype == NS_BF16 ? 8 : 16;
    const bool vec = p->cols % 4 == 0 && p->ld_src % 4 == 0 && p->ld_dst % 4 == 0 && (!p->transpose || p->rows % 4 == 0) &&
                     al(p->src, 16) && al(p->dst, dsz) && al(p->dst_hi, 8) && al(p->dst_lo, 8) && (long)p->rows * p->cols >= 4096;
    if (vec) {
      const dim3 grid(ceil_div(p->cols, 64), ceil_div(p->rows, 64));
      if (p->transpose) hipLaunchKernelGGL(cast2d_vec_kernel<true>, grid, dim3(256), 0, (hipStream_t)s, *p);
      else hipLaunchKernelGGL(cast2d_vec_kernel<false>, grid, dim3(256), 0, (hipStream_t)s, *p);
      NS_CHECK_LAUNCH("cast2d");
      return NS_OK;
    }
  }
  dim3 grid(ceil_div(p->cols, 32), ceil_div(p->rows, 32));
  hipLaunchKernelGGL(cast2d_kernel, grid, dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("cast2d");
  return NS_OK;
}

// ------------------------------------------------------------------ hi/lo split
__global__ void split_kernel(ns_split_params p) {
  bf16_t* hi = (bf16_t*)p.hi;
  bf16_t* lo = (bf16_t*)p.lo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long)gridDim.x * blockDim.x) {
    const float x = p.src[i];
    const bf16_t h = (bf16_t)x;
    hi[i] = h;
    lo[i] = (bf16_t)(x - (float)h);
  }
}
extern "C" int ns_split_hi_lo(const ns_split_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->src && p->hi && p->lo, "ns_split_hi_lo: null");
  if (p->n <= 0) return NS_OK;
  int grid = (int)min((long)4096, (long)((p->n + 255) / 256));
  hipLaunchKernelGGL(split_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("split_hi_lo");
  return NS_OK;
}

// ------------------------------------------------------------------ GRU element-wise pieces
template <typename T>
__global__ void gru_pointwise_kernel(ns_gru_pointwise_params p) {
  const int total = p.N * p.H;
  const int H = p.H;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int n = idx / H, u = idx % H;
    const int len = p.lengths ? p.lengths[n] : p.T;
    const bool masked = p.lengths && p.t >= len;
    const bool first = p.h_init && (p.reverse ? p.t == len - 1 : p.t == 0);
    const float hp = first ? p.h_init[(long)n * p.hi_sn + u] : (p.h_prev ? ldf((const T*)p.h_prev + (long)n * p.hp_sn + u) : 0.f);
    const float r = p.ru ? p.ru[(long)n * p.ru_sn + u] : 0.f;
    const float uu = p.ru ? p.ru[(long)n * p.ru_sn + H + u] : 0.f;
    if (p.mode == 0) {
      stf((T*)p.out + (long)n * p.out_sn + u, r * hp);
    } else if (p.mode == 1) {
      const float c = p.c[(long)n * p.c_sn + u];
      const float h = masked ? 0.f : uu * hp + (1.f - uu) * c;
      stf((T*)p.out + (long)n * p.out_sn + u, h);
      if (p.out2) stf((T*)p.out2 + (long)n * p.out2_sn + u, h);
    } else if (p.mode == 2) {
      const float c = p.c[(long)n * p.c_sn + u];
      const float dh = masked ? 0.f : p.dh[(long)n * p.dh_sn + u] + (p.dh_add ? p.dh_add[(long)n * p.dha_sn + u] : 0.f);
      stf((T*)p.out + (long)n * p.out_sn + u, dh * (1.f - uu) * (1.f - c * c));
      stf((T*)p.dzg + (long)n * p.dzg_sn + H + u, dh * (hp - c) * uu * (1.f - uu));
      if (!masked) p.carry[(long)n * p.carry_sn + u] = dh * uu;
    } else {
      const float drh = masked ? 0.f : p.dh[(long)n * p.dh_sn + u];
      stf((T*)p.dzg + (long)n * p.dzg_sn + u, drh * hp * r * (1.f - r));
      p.carry[(long)n * p.carry_sn + u] += drh * r;
    }
  }
}
extern "C" int ns_gru_pointwise(const ns_gru_pointwise_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->mode >= 0 && p->mode <= 3, "ns_gru_pointwise: bad mode");
  const int total = p->N * p->H;
  if (total <= 0) return NS_OK;
  const int grid = min(1024, ceil_div(total, 256));
  if (p->dtype == NS_BF16) hipLaunchKernelGGL(gru_pointwise_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  else hipLaunchKernelGGL(gru_pointwise_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("gru_pointwise");
  return NS_OK;
}

// ------------------------------------------------------------------ highway combine
template <typename T>
__global__ void highway_kernel(ns_highway_params p) {
  const T* h = (const T*)p.h; const T* t = (const T*)p.t; const T* x = (const T*)p.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long)gridDim.x * blockDim.x) {
    const float hv = ldf(h + i), tv = ldf(t + i), xv = ldf(x + i);
    if (!p.backward) {
      stf((T*)p.y + i, hv * tv + xv * (1.f - tv));
    } else {
      const float dy = p.dy[i];
      stf((T*)p.dhpre + i, hv > 0.f ? dy * tv : 0.f);
      stf((T*)p.dtpre + i, dy * (hv - xv) * tv * (1.f - tv));
      p.dx[i] = dy * (1.f - tv);
    }
  }
}
extern "C" int ns_highway(const ns_highway_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->h && p->t && p->x, "ns_highway: null");
  if (p->n <= 0) return NS_OK;
  const int grid = (int)min((long)4096, (long)((p->n + 255) / 256));
  if (p->dtype == NS_BF16) hipLaunchKernelGGL(highway_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  else hipLaunchKernelGGL(highway_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("highway");
  return NS_OK;
}

// ------------------------------------------------------------------ activation backward
template <typename T>
__global__ void act_bwd_kernel(ns_act_bwd_params p) {
  const long total = (long)p.rows * p.C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    bool valid = true;
    if (p.row_period > 0) {
      const int t = (int)((i / p.C) % p.row_period);
      valid = t >= p.row_lo && t < p.row_hi;
    }
    float d = 0.f;
    if (valid) {
      d = p.dy[i];
      const float y = ldf((const T*)p.y + i);
      if (p.act == NS_ACT_RELU) d = y > 0.f ? d : 0.f;
      else if (p.act == NS_ACT_TANH) d *= (1.f - y * y);
      else if (p.act == NS_ACT_SIGMOID) d *= y * (1.f - y);
      else if (p.act == NS_ACT_SOFTSIGN) d *= (1.f - fabsf(y)) * (1.f - fabsf(y));   // y = x/(1+|x|): dy/dx = (1-|y|)^2
    }
    stf((T*)p.dpre + i, d);
  }
}
extern "C" int ns_act_bwd(const ns_act_bwd_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->dy && p->y && p->dpre, "ns_act_bwd: null");
  const long total = (long)p->rows * p->C;
  if (total <= 0) return NS_OK;
  const int grid = (int)min((long)4096, (total + 255) / 256);
  if (p->dtype == NS_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  else hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("act_bwd");
  return NS_OK;
}
