// Persistent LSTM recurrence for WIDE cells at small batch (the two decoder LSTMs of tacotron2.py:67-73: 1024 units,
// 32 rows, 200 steps): ONE launch for the whole sequence instead of one per time step.
//
// The launch-per-step kernels re-stream the whole recurrent matrix (16.8 MB as hi + lo bf16 planes, forward) through
// every CU each step and pay a dependent-launch boundary on top: ~9.9 us per step.  Here a workgroup keeps its slice of
// W_h in REGISTERS for all steps (forward: 8 units x 4 gates x H as hi / lo MFMA B fragments, 64 VGPRs per lane), the
// cell state too, and only the state travels - through the history array the kernel has to write anyway:
//   workgroup (rg, ub) = 16 batch rows x 8 units (forward) / 16 units (backward); its 8 waves split K
//   per step:  wait until every workgroup of the row group has published step t-1 (8 counters per row group, one
//              per 128-byte line, bumped by one lane per workgroup AFTER its write-through stores have drained),
//              load the own K slice of h[t-1] (backward: of the bf16 gate gradients of step t+1) with 16-byte sc1
//              loads - all 8 waves, 8..16 loads in flight per lane, served by the XCD's L2 after the first touch:
//              64 KB per CU in ~1.1 us measured (profiles/tools/allgather_bench.hip) -, MFMA, partial sums meet in
//              LDS, cell update, h[t] (backward: dgates[t] as bf16) out as 16-byte sc1 stores, drain, signal.
// This is the hand-off form of the cdna guide's G16 table, first row: payload stored sc1 and drained, ONE lane per
// storing workgroup adds to a counter (sharded), the consumer polls the counter with sc1 loads and every load of the
// payload is an sc1 load issued after the poll matched.  Every spin is bounded; a timeout raises the status word and
// all waves of the workgroup leave together.  The grid (row groups x unit blocks <= 256 workgroups, one per CU) must
// be resident at once: ns_lstm_wide_supported() refuses shapes that do not fit.
#include "common.h"
#include <stdint.h>

namespace {
constexpr int WT = 512, WW = WT / 64;
constexpr unsigned WSPIN = 3000000u;
constexpr int CNT_STRIDE = 32;       // uints between counter shards: 128 bytes, one line each
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct WideArgs {
  ns_lstm_seq_params p;
  unsigned* cnt;        // [row groups][8 shards][CNT_STRIDE]
  int* status;
  int nub;              // unit blocks per row group
};

__device__ __forceinline__ bool wait_counters(const unsigned* c, unsigned need, int lane, int* status, int* abortf, int code) {
  unsigned spins = 0;
  for (;;) {
    const unsigned v = lane < 8 ? __hip_atomic_load(c + lane * CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
    if (__all(v >= need)) return true;
    ++spins;
    if (spins > WSPIN) { if (lane == 0) { atomicExch(status, code); *abortf = 1; } return false; }
    if ((spins & 1023u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { if (lane == 0) *abortf = 1; return false; }
  }
}

// 8 consecutive state values as an MFMA A fragment pair (hi, lo), read with sc1 (L1-bypassing) 16-byte loads
__device__ __forceinline__ void ld_frag_sc1(const float* base, size_t bytes_total, unsigned off_bytes, bool ok, bf16x8& hi, bf16x8& lo) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes_total, 0x00020000);
  u32x4 a = {0u, 0u, 0u, 0u}, b = a;
  if (ok) { a = __builtin_amdgcn_raw_buffer_load_b128(rs, off_bytes, 0, 16); b = __builtin_amdgcn_raw_buffer_load_b128(rs, off_bytes + 16, 0, 16); }
  const float f[8] = {__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]), __uint_as_float(a[3]),
                      __uint_as_float(b[0]), __uint_as_float(b[1]), __uint_as_float(b[2]), __uint_as_float(b[3])};
#pragma unroll
  for (int i = 0; i < 8; ++i) { const bf16_t h = (bf16_t)f[i]; hi[i] = h; lo[i] = (bf16_t)(f[i] - (float)h); }
}
__device__ __forceinline__ void ld_frag_sc1(const bf16_t* base, size_t bytes_total, unsigned off_bytes, bool ok, bf16x8& hi, bf16x8& lo) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes_total, 0x00020000);
  u32x4 a = {0u, 0u, 0u, 0u};
  if (ok) a = __builtin_amdgcn_raw_buffer_load_b128(rs, off_bytes, 0, 16);
  hi = *(bf16x8*)&a;
  (void)lo;
}

// ===================================================================================== forward
// NCH = 32-wide K chunks per wave (H / 256); PASSES = 3: fp32 state, hi / lo weight planes; 1: bf16 everywhere
template <typename T, int PASSES, int NCH>
__global__ __launch_bounds__(WT) void lstm_wide_fwd_kernel(WideArgs a) {
  __shared__ float red[WW][16][33];
  __shared__ __attribute__((aligned(16))) T hst[16][8];
  __shared__ int abortf;
  const ns_lstm_seq_params& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, NUB = a.nub;
  const int rg = blockIdx.x / NUB, ub = blockIdx.x % NUB;
  const int n0 = rg * 16, u0 = ub * 8;
  const int r16 = lane & 15, g = lane >> 4;
  const int k0 = wave * (H / WW);
  if (tid == 0) abortf = 0;
  // ---- resident weight fragments: tile j, column r16 -> gate 2j + (r16 >> 3), unit u0 + (r16 & 7)
  bf16x8 bh[2][NCH], bl[2][NCH];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long wrow = ((long)(2 * j + (r16 >> 3)) * H + u0 + (r16 & 7)) * H + k0 + g * 8;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if constexpr (PASSES == 3) {
        bh[j][c] = *(const bf16x8*)((const bf16_t*)p.whT_hi + wrow + c * 32);
        bl[j][c] = *(const bf16x8*)((const bf16_t*)p.whT_lo + wrow + c * 32);
      } else {
        bh[j][c] = *(const bf16x8*)((const bf16_t*)p.whT + wrow + c * 32);
        bl[j][c] = bh[j][c];
      }
    }
  }
  // ---- epilogue ownership: threads < 128 = (row er, unit eu)
  const int er = tid >> 3, eu = tid & 7;
  const bool eown = tid < 128;
  const int en = n0 + er;
  const bool eok = eown && en < p.N;
  const int elen = (eok && p.lengths) ? p.lengths[en] : p.T;
  float cst = 0.f;
  float pz[4] = {0.f, 0.f, 0.f, 0.f};
  auto load_xg = [&](int t) {
    if (eok) {
      const float* xr = p.xg + ((long)en * p.P + p.padl + t) * p.ld_xg + u0 + eu;
#pragma unroll
      for (int j = 0; j < 4; ++j) pz[j] = xr[(long)j * H];
    }
  };
  load_xg(0);
  const size_t hbytes = (size_t)p.N * p.P * p.ld_h * sizeof(T);
  unsigned* mycnt = a.cnt + (size_t)(rg * 8 + (ub & 7)) * CNT_STRIDE;
  const unsigned* rgcnt = a.cnt + (size_t)rg * 8 * CNT_STRIDE;
  const unsigned per_shard = (unsigned)(NUB / 8);
  __syncthreads();

  for (int t = 0; t < p.T; ++t) {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (t > 0) {
      wait_counters(rgcnt, per_shard * (unsigned)t, lane, a.status, &abortf, 1);
      const unsigned rowoff = (unsigned)(((long)(n0 + r16) * p.P + p.padl + t - 1) * p.ld_h + k0 + g * 8) * (unsigned)sizeof(T);
      const bool ok = n0 + r16 < p.N;
      bf16x8 ah[NCH], al[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) ld_frag_sc1((const T*)p.h, hbytes, rowoff + c * 32 * (unsigned)sizeof(T), ok, ah[c], al[c]);
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (PASSES == 3) acc[j] = mfma_split<3>(ah[c], al[c], bh[j][c], bl[j][c], acc[j]);
          else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[c], bh[j][c], acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[wave][g * 4 + q][j * 16 + r16] = acc[j][q];
    __syncthreads();
    if (abortf) return;
    float gi = 0.f, gj = 0.f, gf = 0.f, go = 0.f, hv = 0.f;
    if (eown) {
      float z[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = pz[j];
#pragma unroll
        for (int w = 0; w < WW; ++w) s += red[w][er][(j >> 1) * 16 + (j & 1) * 8 + eu];
        z[j] = s;
      }
      gi = sigmoidf_(z[0]); gj = tanhf_(z[1]); gf = sigmoidf_(z[2] + p.forget_bias); go = sigmoidf_(z[3]);
      cst = gf * cst + gi * gj;
      hv = go * tanhf_(cst);
      if (t >= elen) { cst = 0.f; hv = 0.f; gi = gj = gf = go = 0.f; }
      hst[er][eu] = (T)hv;
    }
    __syncthreads();
    // ---- publish h[t]: 16-byte write-through stores by wave 0, drained, then ONE counter add
    if (wave == 0) {
      constexpr int PPR = 8 * (int)sizeof(T) / 16;        // 16-byte pieces per row (2 for fp32, 1 for bf16)
      if (lane < 16 * PPR) {
        const int row = lane / PPR, pc = lane % PPR;
        if (n0 + row < p.N) {
          const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.h, 0, (int)hbytes, 0x00020000);
          const u32x4 v = *(const u32x4*)((const char*)&hst[row][0] + pc * 16);
          const unsigned off = (unsigned)(((long)(n0 + row) * p.P + p.padl + t) * p.ld_h + u0) * (unsigned)sizeof(T) + pc * 16;
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);             // aux 16 = sc1
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0 && t + 1 < p.T) __hip_atomic_fetch_add(mycnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- what only the backward pass reads (plain stores, off the critical path), next step's input gates
    if (eok) {
      const long rowi = (long)en * p.P + p.padl + t;
      p.c[rowi * H + u0 + eu] = cst;
      if (p.gates) {
        T* gp = (T*)p.gates + rowi * 4 * H + u0 + eu;
        stf(gp, gi); stf(gp + H, gj); stf(gp + 2 * H, gf); stf(gp + 3 * H, go);
      }
    }
    if (t + 1 < p.T) load_xg(t + 1);
  }
}

// ===================================================================================== backward
// dh[t] = dh_out[t] + dgates[t+1] . Wh^T.  Workgroup = 16 rows x 16 units (rows of Wh [H, 4H], K = 4H split over the 8
// waves, NCH = 32-wide chunks per wave = H / 64); the exchanged payload is the bf16 copy of the gate gradients
// (dgates_bf16, or dgates itself when the storage type is bf16).
template <typename T, int NCH>
__global__ __launch_bounds__(WT) void lstm_wide_bwd_kernel(WideArgs a) {
  __shared__ float red[WW][16][17];
  __shared__ __attribute__((aligned(16))) bf16_t dst[16][4][16];      // this step's gate gradients (row, gate, unit)
  __shared__ int abortf;
  const ns_lstm_seq_params& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, K = 4 * H, NUB = a.nub;
  const int rg = blockIdx.x / NUB, ub = blockIdx.x % NUB;
  const int n0 = rg * 16, u0 = ub * 16;
  const int r16 = lane & 15, g = lane >> 4;
  const int k0 = wave * (K / WW);
  if (tid == 0) abortf = 0;
  const bf16_t* W = sizeof(T) == 2 ? (const bf16_t*)p.wh : (const bf16_t*)p.wh_bf16;
  bf16_t* xb = sizeof(T) == 2 ? (bf16_t*)p.dgates : (bf16_t*)p.dgates_bf16;     // exchange payload [N*P, 4H] bf16
  bf16x8 bw[NCH];
  {
    const bf16_t* row = W + (long)(u0 + r16) * K + k0 + g * 8;
#pragma unroll
    for (int c = 0; c < NCH; ++c) bw[c] = *(const bf16x8*)(row + c * 32);
  }
  // epilogue ownership: threads < 256 = (row er, unit eu)
  const int er = tid >> 4, eu = tid & 15;
  const bool eown = tid < 256;
  const int en = n0 + er;
  const bool eok = eown && en < p.N;
  const int elen = (eok && p.lengths) ? p.lengths[en] : p.T;
  float dcc = 0.f;
  float pg[4] = {0.f, 0.f, 0.f, 0.f}, pdh = 0.f, pc = 0.f, pcp = 0.f;
  auto load_ops = [&](int t) {
    if (eok) {
      const long rowi = (long)en * p.P + p.padl + t;
      const T* gp = (const T*)p.gates + rowi * 4 * H + u0 + eu;
#pragma unroll
      for (int j = 0; j < 4; ++j) pg[j] = ldf(gp + (long)j * H);
      pdh = p.dh[rowi * p.ld_dh + u0 + eu];
      pc = p.c[rowi * H + u0 + eu];
      pcp = t > 0 ? p.c[(rowi - 1) * H + u0 + eu] : 0.f;
    }
  };
  load_ops(p.T - 1);
  const size_t xbytes = (size_t)p.N * p.P * K * sizeof(bf16_t);
  unsigned* mycnt = a.cnt + (size_t)(rg * 8 + (ub & 7)) * CNT_STRIDE;
  const unsigned* rgcnt = a.cnt + (size_t)rg * 8 * CNT_STRIDE;
  const unsigned per_shard = (unsigned)(NUB / 8);
  __syncthreads();

  for (int t = p.T - 1; t >= 0; --t) {
    const int bs = p.T - 1 - t;                       // backward step index
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (bs > 0) {
      wait_counters(rgcnt, per_shard * (unsigned)bs, lane, a.status, &abortf, 2);
      const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)xbytes, 0x00020000);
      const unsigned rowoff = (unsigned)(((long)(n0 + r16) * p.P + p.padl + t + 1) * K + k0 + g * 8) * 2u;
      const bool ok = n0 + r16 < p.N;
      u32x4 av[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) av[c] = ok ? __builtin_amdgcn_raw_buffer_load_b128(rs, rowoff + c * 64, 0, 16) : (u32x4){0u, 0u, 0u, 0u};
      f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCH; c += 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&av[c], bw[c], acc, 0, 0, 0);
        if (c + 1 < NCH) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&av[c + 1], bw[c + 1], acc2, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += acc2[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][g * 4 + q][r16] = acc[q];
    __syncthreads();
    if (abortf) return;
    float dgv[4] = {0.f, 0.f, 0.f, 0.f};
    if (eown) {
      float dh = pdh;
#pragma unroll
      for (int w = 0; w < WW; ++w) dh += red[w][er][eu];
      const float gi = pg[0], gj = pg[1], gf = pg[2], go = pg[3];
      const float tc = tanhf_(pc);
      const float dc = dh * go * (1.f - tc * tc) + dcc;
      dgv[0] = dc * gj * gi * (1.f - gi);
      dgv[1] = dc * gi * (1.f - gj * gj);
      dgv[2] = dc * pcp * gf * (1.f - gf);
      dgv[3] = dh * tc * go * (1.f - go);
      dcc = dc * gf;
      if (t >= elen || !eok) { dgv[0] = dgv[1] = dgv[2] = dgv[3] = 0.f; dcc = 0.f; }
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[er][j][eu] = (bf16_t)dgv[j];
    }
    __syncthreads();
    // ---- publish dgates[t] (bf16): 16 rows x 4 gates x 32 bytes = 128 pieces of 16 bytes, waves 0 and 1
    if (wave < 2) {
      const int piece = wave * 64 + lane, row = piece >> 3, gate = (piece >> 1) & 3, hf = piece & 1;
      if (n0 + row < p.N) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)xbytes, 0x00020000);
        const u32x4 v = *(const u32x4*)((const char*)&dst[row][gate][0] + hf * 16);
        const unsigned off = (unsigned)(((long)(n0 + row) * p.P + p.padl + t) * K + (long)gate * H + u0) * 2u + hf * 16;
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0 && t > 0) __hip_atomic_fetch_add(mycnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- fp32 copy of the gate gradients for the weight-gradient products (storage type float), next operands
    if (eok && sizeof(T) == 4) {
      float* dg = (float*)p.dgates + ((long)en * p.P + p.padl + t) * K + u0 + eu;
#pragma unroll
      for (int j = 0; j < 4; ++j) dg[(long)j * H] = dgv[j];
    }
    if (t > 0) load_ops(t - 1);
  }
}

bool wide_shape_ok(const ns_lstm_seq_params* p, int backward, int* nub_out) {
  if (!p || p->reverse || p->T < 2) return false;
  const int H = p->H;
  if (H != 256 && H != 512 && H != 1024) return false;
  const int nrg = (p->N + 15) / 16;
  const int nub = backward ? H / 16 : H / 8;
  if (nub % 8 != 0 || nrg * nub > 256) return false;
  if (nub_out) *nub_out = nub;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  const long esz = p->dtype == NS_BF16 ? 2 : 4;
  if (!backward) {
    if (p->dtype == NS_F32 && !(p->f32_passes == 3 && p->whT_hi && p->whT_lo)) return false;
    if (p->dtype == NS_BF16 && !p->whT) return false;
    if (!p->xg || !p->h || !p->c || !al16(p->h) || (p->ld_h * esz) % 16 != 0) return false;
    if ((double)p->N * p->P * p->ld_h * esz >= 2.0e9) return false;
  } else {
    if (p->dtype == NS_F32 && !(p->f32_passes == 1 && p->wh_bf16 && p->dgates_bf16)) return false;
    if (p->dtype == NS_BF16 && !p->wh) return false;
    if (!p->gates || !p->c || !p->dh || !p->dgates) return false;
    if ((double)p->N * p->P * 4 * H * 2 >= 2.0e9) return false;
  }
  return true;
}
}  // namespace

extern "C" int ns_lstm_wide_supported(const ns_lstm_seq_params* p, int backward) { return wide_shape_ok(p, backward, nullptr) ? 1 : 0; }
extern "C" size_t ns_lstm_wide_work_bytes(const ns_lstm_seq_params* p) {
  if (!p) return 0;
  return 256 + sizeof(unsigned) * (size_t)((p->N + 15) / 16) * 8 * CNT_STRIDE;
}

template <typename T, int PASSES>
static int launch_wide_fwd(const WideArgs& a, int grid, hipStream_t s) {
  switch (a.p.H) {
    case 256: hipLaunchKernelGGL((lstm_wide_fwd_kernel<T, PASSES, 1>), dim3(grid), dim3(WT), 0, s, a); break;
    case 512: hipLaunchKernelGGL((lstm_wide_fwd_kernel<T, PASSES, 2>), dim3(grid), dim3(WT), 0, s, a); break;
    default: hipLaunchKernelGGL((lstm_wide_fwd_kernel<T, PASSES, 4>), dim3(grid), dim3(WT), 0, s, a); break;
  }
  NS_CHECK_LAUNCH("lstm_wide_fwd");
  return NS_OK;
}

// Whole-sequence forward recurrence of one wide LSTM cell, one launch (see the header of this file).  Same parameter
// block and outputs as ns_lstm_seq_fwd.  work: ns_lstm_wide_work_bytes(); work[0] (int) is a status word, non-zero
// after the call completes = a wait timed out and the outputs are invalid.
extern "C" int ns_lstm_wide_fwd(const ns_lstm_seq_params* p, void* work, ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  int nub = 0;
  NS_CHECK_ARG(p && work, "ns_lstm_wide_fwd: null");
  NS_CHECK_ARG(wide_shape_ok(p, 0, &nub), "ns_lstm_wide_fwd: unsupported (needs H in {256, 512, 1024}, row groups x H/8 <= 256, "
               "fp32 with pre-split whT_hi / whT_lo and f32_passes 3, or bf16)");
  WideArgs a;
  a.p = *p; a.status = (int*)work; a.cnt = (unsigned*)((char*)work + 256); a.nub = nub;
  int rc = ns_zero_async(work, (ns_lstm_wide_work_bytes(p) + 15) & ~(size_t)15, s);
  if (rc) return rc;
  const int grid = ((p->N + 15) / 16) * nub;
  if (p->dtype == NS_BF16) return launch_wide_fwd<bf16_t, 1>(a, grid, s);
  return launch_wide_fwd<float, 3>(a, grid, s);
}

template <typename T>
static int launch_wide_bwd(const WideArgs& a, int grid, hipStream_t s) {
  switch (a.p.H) {
    case 256: hipLaunchKernelGGL((lstm_wide_bwd_kernel<T, 4>), dim3(grid), dim3(WT), 0, s, a); break;
    case 512: hipLaunchKernelGGL((lstm_wide_bwd_kernel<T, 8>), dim3(grid), dim3(WT), 0, s, a); break;
    default: hipLaunchKernelGGL((lstm_wide_bwd_kernel<T, 16>), dim3(grid), dim3(WT), 0, s, a); break;
  }
  NS_CHECK_LAUNCH("lstm_wide_bwd");
  return NS_OK;
}

extern "C" int ns_lstm_wide_bwd(const ns_lstm_seq_params* p, void* work, ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  int nub = 0;
  NS_CHECK_ARG(p && work, "ns_lstm_wide_bwd: null");
  NS_CHECK_ARG(wide_shape_ok(p, 1, &nub), "ns_lstm_wide_bwd: unsupported (needs H in {256, 512, 1024}, row groups x H/16 <= 256, "
               "bf16, or fp32 with wh_bf16 + dgates_bf16 and f32_passes 1)");
  WideArgs a;
  a.p = *p; a.status = (int*)work; a.cnt = (unsigned*)((char*)work + 256); a.nub = nub;
  int rc = ns_zero_async(work, (ns_lstm_wide_work_bytes(p) + 15) & ~(size_t)15, s);
  if (rc) return rc;
  const int grid = ((p->N + 15) / 16) * nub;
  if (p->dtype == NS_BF16) return launch_wide_bwd<bf16_t>(a, grid, s);
  return launch_wide_bwd<float>(a, grid, s);
}
