// Persistent BiLSTM recurrence: ONE launch for the whole sequence instead of one per time step.
//
// A "chain" = (direction, group of 16 batch rows) is an independent recurrence.  It runs on a
// cluster of CS = H/64 workgroups (one per CU); workgroup c owns hidden units [64c, 64c+64) and
// keeps its slice of W_h^T - 64 units x 4 gates x H - in REGISTERS as MFMA B fragments for the
// whole sequence (8 waves x 8 units, 64 VGPRs per lane), and the cell state in registers too.
// Per step the only inter-workgroup traffic is the new h slice (16 rows x 64 units), exchanged
// through global memory as 8-byte {step tag, 2 x bf16} granules written and polled with relaxed
// agent-scope atomics (sc1; no fences, data is its own flag), double-buffered by step parity.
// Measured exchange cost: ~1.3 us per step for 4 workgroups (vs ~6-8 us per dependent launch).
//
// Correctness of the 2-deep buffering: a workgroup publishes step s+2 into the slot of step s only
// after it has gathered every peer's step s+1, which each peer published only after gathering step s.
// Every spin is bounded; on timeout the kernel sets *status and every workgroup leaves.
#include "common.h"
#include <stdlib.h>

typedef unsigned long long u64;
constexpr int CW = 8;            // waves per workgroup
constexpr int CTHREADS = CW * 64;
constexpr unsigned SPIN_LIMIT = 4000000u;

struct LstmClusterArgs {
  int N, T, H, P, padl, CS;
  // per direction d (0 = forward in time, 1 = reversed)
  const float* xg[2]; long ld_xg;
  const bf16_t* whT[2];           // [4H, H]
  const bf16_t* wh[2];            // [H, 4H] (backward)
  bf16_t* h[2]; long ld_h;        // h[d] already offset to this direction's columns
  float* c[2];
  bf16_t* gates[2];
  const float* dh[2]; long ld_dh; // backward: grad wrt h outputs (offset to direction's columns)
  bf16_t* dgates[2];
  const int* lengths;
  float forget_bias;
  u64* xbuf;                      // [chains][2][16][granules per row]
  int* status;
  int dbg;                        // timing experiments only (NS_CLUSTER_DBG), 0 in production
};

__device__ __forceinline__ int swz_off(int row, int k, int H) {   // bf16 element offset in the LDS h image
  const int chunk = k >> 3;
  return row * H + (((chunk ^ (row & 15)) << 3) | (k & 7));
}

// ------------------------------------------------------------------ forward
__global__ __launch_bounds__(CTHREADS) void lstm_cluster_fwd_kernel(LstmClusterArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* hs = (bf16_t*)smem;                       // [16][H] swizzled
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int CS = a.CS, H = a.H;
  const int nrg = (a.N + 15) / 16;
  const int chain = blockIdx.x / CS, wgc = blockIdx.x % CS;
  const int d = chain / nrg, rg = chain % nrg;
  const int n0 = rg * 16;
  const int r16 = lane & 15, g = lane >> 4;
  const int GPR = H / 2;                            // granules per row
  u64* xb = a.xbuf + (size_t)chain * 2 * 16 * GPR;
  const int uw0 = wgc * 64 + wave * 8;              // this wave's 8 units
  const int ksteps = H / 32;

  // ---- resident weight fragments: tile 0 = [i | j], tile 1 = [f | o] for 8 units
  bf16x8 bw[2][16];
  {
    const bf16_t* W = a.whT[d];
    const int unit = uw0 + (r16 & 7);
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
      const int gate = tl * 2 + (r16 >> 3);
      const bf16_t* row = W + ((long)gate * H + unit) * H;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        bw[tl][ks] = ks < ksteps ? *(const bf16x8*)(row + ks * 32 + g * 8) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  float cst[4] = {0.f, 0.f, 0.f, 0.f};
  const bool cell_lane = r16 < 8;
  const int unit = uw0 + (r16 & 7);
  int len[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + g * 4 + r;
    len[r] = (a.lengths && n < a.N) ? a.lengths[n] : a.T;
  }
  const float* xg = a.xg[d];
  // xg prefetch for step 0
  float xa[4], xb2[4], xa_n[4], xb_n[4];
  auto load_xg = [&](int t, float* pa, float* pb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + g * 4 + r;
      const long rowi = (long)n * a.P + a.padl + t;
      const bool ok = n < a.N;
      const int ga = (r16 >> 3), gb = 2 + (r16 >> 3);
      pa[r] = (ok && !(a.dbg & 1)) ? xg[rowi * a.ld_xg + (long)ga * H + unit] : 0.f;
      pb[r] = (ok && !(a.dbg & 1)) ? xg[rowi * a.ld_xg + (long)gb * H + unit] : 0.f;
    }
  };
  load_xg(d ? a.T - 1 : 0, xa, xb2);

  for (int step = 0; step < a.T; ++step) {
    const int t = d ? a.T - 1 - step : step;
    if (step + 1 < a.T) load_xg(d ? t - 1 : t + 1, xa_n, xb_n);
    f32x4 accA = {xa[0], xa[1], xa[2], xa[3]};
    f32x4 accB = {xb2[0], xb2[1], xb2[2], xb2[3]};
    if (step > 0) {
      // ---- gather h of the previous step from the whole cluster into LDS
      const u64* cur = xb + (size_t)(step & 1) * 16 * GPR;   // written at the end of step-1 with tag = step
      const int total = 16 * GPR;
      for (int i0 = tid; i0 < total; i0 += CTHREADS * 4) {
        u64 v[4];
        unsigned spins = 0;
        bool ok;
        do {
          ok = true;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = i0 + j * CTHREADS;
            v[j] = i < total ? __hip_atomic_load(cur + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : ((u64)(unsigned)step << 32);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) ok = ok && ((unsigned)(v[j] >> 32) == (unsigned)step);
          if (a.dbg & 4) ok = true;
          if (!ok && ++spins > SPIN_LIMIT) { atomicExch(a.status, 1); ok = true; }
        } while (!ok);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = i0 + j * CTHREADS;
          if (i < total) {
            const int row = i / GPR, pr = i % GPR;
            *(unsigned*)(hs + swz_off(row, pr * 2, H)) = (unsigned)v[j];
          }
        }
      }
      __syncthreads();
      if (*(volatile int*)a.status) return;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks < ksteps) {
          const bf16x8 af = *(const bf16x8*)(hs + swz_off(r16, ks * 32 + g * 8, H));
          accA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bw[0][ks], accA, 0, 0, 0);
          accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bw[1][ks], accB, 0, 0, 0);
        }
      }
      __syncthreads();   // hs is rewritten by the next gather
    }
    // ---- cell update: lanes r16 < 8 hold (i, f); their partners r16+8 hold (j, o)
    float hv[4], sgi[4], sgj[4], sgf[4], sgo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float zj = __shfl_down(accA[r], 8, 64);
      const float zo = __shfl_down(accB[r], 8, 64);
      const bool masked = t >= len[r];
      const float gi = sigmoidf_(accA[r]), gj = tanhf_(zj), gf = sigmoidf_(accB[r] + a.forget_bias), go = sigmoidf_(zo);
      float cn = gf * cst[r] + gi * gj;
      float hn = go * tanhf_(cn);
      if (masked) { cn = 0.f; hn = 0.f; }
      cst[r] = cn;
      hv[r] = hn;
      sgi[r] = masked ? 0.f : gi; sgj[r] = masked ? 0.f : gj; sgf[r] = masked ? 0.f : gf; sgo[r] = masked ? 0.f : go;
    }
    // ---- publish h first (tag = step + 1): the peers are waiting on it; even unit lanes pack (h[u], h[u+1])
    if (step + 1 < a.T) {
      u64* nxt = xb + (size_t)((step + 1) & 1) * 16 * GPR;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hp = __shfl_down(hv[r], 1, 64);
        if (cell_lane && !(r16 & 1)) {
          const bf16_t b0 = (bf16_t)hv[r], b1 = (bf16_t)hp;
          const unsigned pay = (unsigned)(*(const unsigned short*)&b0) | ((unsigned)(*(const unsigned short*)&b1) << 16);
          const int row = g * 4 + r;
          __hip_atomic_store(nxt + (size_t)row * GPR + (unit >> 1), ((u64)(unsigned)(step + 1) << 32) | pay,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    // ---- then the saves for the backward pass / the consumers of h
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + g * 4 + r;
      if (cell_lane && n < a.N && !(a.dbg & 2)) {
        const long rowi = (long)n * a.P + a.padl + t;
        a.h[d][rowi * a.ld_h + unit] = (bf16_t)hv[r];
        a.c[d][rowi * H + unit] = cst[r];
        bf16_t* gp = a.gates[d] + rowi * 4 * H;
        gp[unit] = (bf16_t)sgi[r];
        gp[H + unit] = (bf16_t)sgj[r];
        gp[2 * H + unit] = (bf16_t)sgf[r];
        gp[3 * H + unit] = (bf16_t)sgo[r];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { xa[r] = xa_n[r]; xb2[r] = xb_n[r]; }
  }
}

// ------------------------------------------------------------------ backward
// dh[t] = dh_out[t] + dgates[next].Wh^T ; this workgroup owns 64 units (rows of Wh [H,4H]); the
// contraction runs over all 4H gate gradients of the next step, gathered from the cluster.
// Wave w holds the K-slice [w*4H/8, (w+1)*4H/8) of the 4 unit tiles in registers.
__global__ __launch_bounds__(CTHREADS) void lstm_cluster_bwd_kernel(LstmClusterArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int CS = a.CS, H = a.H, K = 4 * a.H;
  bf16_t* dgs = (bf16_t*)smem;                                 // [16][4H] swizzled gathered gate grads
  float* red = (float*)(smem + (size_t)16 * K * 2);            // [CW][16][65]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nrg = (a.N + 15) / 16;
  const int chain = blockIdx.x / CS, wgc = blockIdx.x % CS;
  const int d = chain / nrg, rg = chain % nrg;
  const int n0 = rg * 16;
  const int r16 = lane & 15, g = lane >> 4;
  const int GPR = K / 2;
  u64* xb = a.xbuf + (size_t)chain * 2 * 16 * GPR;
  const int u0 = wgc * 64;
  const int kpw = K / CW;                  // K-slice per wave (multiple of 32)
  const int ksteps = kpw / 32;             // <= 8 for H <= 512
  bf16x8 bw[4][8];
  {
    const bf16_t* W = a.wh[d];
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
      const bf16_t* row = W + (long)(u0 + tl * 16 + r16) * K + wave * kpw;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        bw[tl][ks] = ks < ksteps ? *(const bf16x8*)(row + ks * 32 + g * 8) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  // cell ownership for the epilogue: thread -> (row = tid >> 5 (16 rows), 2 units)
  const int er = tid >> 5, eu = (tid & 31) * 2;
  const int en = n0 + er;
  const int elen = (a.lengths && en < a.N) ? a.lengths[en] : a.T;
  float dcc[2] = {0.f, 0.f};

  for (int step = a.T - 1; step >= 0; --step) {      // walk the forward order backwards
    const int t = d ? a.T - 1 - step : step;
    const int tp = d ? t + 1 : t - 1;                 // forward-pass predecessor
    const bool has_prev = step > 0;
    const bool has_next = step < a.T - 1;
    const int bs = a.T - 1 - step;                    // backward step index, 0-based
    // prefetch epilogue operands
    float pdh[2] = {0.f, 0.f}, pg[2][4], pc[2] = {0.f, 0.f}, pcp[2] = {0.f, 0.f};
    const long rowi = (long)en * a.P + a.padl + t;
    const bool ok = en < a.N;
    const bool masked = t >= elen;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = u0 + eu + q;
#pragma unroll
      for (int j = 0; j < 4; ++j) pg[q][j] = ok ? (float)a.gates[d][rowi * 4 * H + (long)j * H + u] : 0.f;
      if (ok) {
        pdh[q] = a.dh[d][rowi * a.ld_dh + u];
        pc[q] = a.c[d][rowi * H + u];
        if (has_prev) pcp[q] = a.c[d][((long)en * a.P + a.padl + tp) * H + u];
      }
    }
    float dhp[2] = {0.f, 0.f};
    if (has_next) {
      // ---- gather dgates of the step after (tag = bs) from the whole cluster
      const u64* cur = xb + (size_t)(bs & 1) * 16 * GPR;
      const int total = 16 * GPR;
      for (int i0 = tid; i0 < total; i0 += CTHREADS * 8) {
        u64 v[8];
        unsigned spins = 0;
        bool okk;
        do {
          okk = true;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int i = i0 + j * CTHREADS;
            v[j] = i < total ? __hip_atomic_load(cur + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : ((u64)(unsigned)bs << 32);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) okk = okk && ((unsigned)(v[j] >> 32) == (unsigned)bs);
          if (!okk && ++spins > SPIN_LIMIT) { atomicExch(a.status, 2); okk = true; }
        } while (!okk);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = i0 + j * CTHREADS;
          if (i < total) {
            const int row = i / GPR, pr = i % GPR;
            *(unsigned*)(dgs + swz_off(row, pr * 2, K)) = (unsigned)v[j];
          }
        }
      }
      __syncthreads();
      if (*(volatile int*)a.status) return;
      f32x4 acc[4];
#pragma unroll
      for (int tl = 0; tl < 4; ++tl) acc[tl] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks < ksteps) {
          const bf16x8 af = *(const bf16x8*)(dgs + swz_off(r16, wave * kpw + ks * 32 + g * 8, K));
#pragma unroll
          for (int tl = 0; tl < 4; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bw[tl][ks], acc[tl], 0, 0, 0);
        }
      }
#pragma unroll
      for (int tl = 0; tl < 4; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 16 + g * 4 + r) * 65 + tl * 16 + r16] = acc[tl][r];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < CW; ++w) s += red[(w * 16 + er) * 65 + eu + q];
        dhp[q] = s;
      }
    }
    // ---- cell gradient for (row er, units eu, eu+1)
    float dgv[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float gi = pg[q][0], gj = pg[q][1], gf = pg[q][2], go = pg[q][3];
      const float dh = pdh[q] + dhp[q];
      const float tc = tanhf_(pc[q]);
      const float d_o = dh * tc * go * (1.f - go);
      const float dc = dh * go * (1.f - tc * tc) + dcc[q];
      dgv[q][0] = dc * gj * gi * (1.f - gi);
      dgv[q][1] = dc * gi * (1.f - gj * gj);
      dgv[q][2] = dc * pcp[q] * gf * (1.f - gf);
      dgv[q][3] = d_o;
      dcc[q] = dc * gf;
      if (masked || !ok) {
        dgv[q][0] = dgv[q][1] = dgv[q][2] = dgv[q][3] = 0.f;
        dcc[q] = 0.f;
      }
    }
    if (step > 0) {   // publish this step's gate gradients first (tag = bs + 1)
      u64* nxt = xb + (size_t)((bs + 1) & 1) * 16 * GPR;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16_t b0 = (bf16_t)dgv[0][j], b1 = (bf16_t)dgv[1][j];
        const unsigned pay = (unsigned)(*(const unsigned short*)&b0) | ((unsigned)(*(const unsigned short*)&b1) << 16);
        __hip_atomic_store(nxt + (size_t)er * GPR + ((j * H + u0 + eu) >> 1), ((u64)(unsigned)(bs + 1) << 32) | pay,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (ok) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16_t b0 = (bf16_t)dgv[0][j], b1 = (bf16_t)dgv[1][j];
        const unsigned pay = (unsigned)(*(const unsigned short*)&b0) | ((unsigned)(*(const unsigned short*)&b1) << 16);
        *(unsigned*)(a.dgates[d] + rowi * 4 * H + (long)j * H + u0 + eu) = pay;
      }
    }
    __syncthreads();   // red / dgs reuse
  }
}

// ------------------------------------------------------------------ C ABI
static int cluster_supported(const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1) {
  return p0->dtype == NS_BF16 && p1->dtype == NS_BF16 && p0->H % 64 == 0 && p0->H <= 512 && p0->T >= 2;
}

extern "C" size_t ns_lstm_cluster_work_bytes(const ns_lstm_seq_params* p) {
  if (!p) return 0;
  const size_t chains = 2 * (size_t)((p->N + 15) / 16);
  // exchange buffers for the larger (backward) payload + status word
  return chains * 2 * 16 * (size_t)(4 * p->H / 2) * sizeof(u64) + 256;
}

static void fill(LstmClusterArgs& a, const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1, void* work) {
  const ns_lstm_seq_params* pp[2] = {p0, p1};
  a.N = p0->N; a.T = p0->T; a.H = p0->H; a.P = p0->P; a.padl = p0->padl; a.CS = p0->H / 64;
  a.ld_xg = p0->ld_xg; a.ld_h = p0->ld_h; a.ld_dh = p0->ld_dh;
  a.lengths = p0->lengths; a.forget_bias = p0->forget_bias;
  for (int d = 0; d < 2; ++d) {
    a.xg[d] = pp[d]->xg; a.whT[d] = (const bf16_t*)pp[d]->whT; a.wh[d] = (const bf16_t*)pp[d]->wh;
    a.h[d] = (bf16_t*)pp[d]->h; a.c[d] = pp[d]->c; a.gates[d] = (bf16_t*)pp[d]->gates;
    a.dh[d] = pp[d]->dh; a.dgates[d] = (bf16_t*)pp[d]->dgates;
  }
  a.status = (int*)work;
  a.xbuf = (u64*)((char*)work + 256);
  const char* dbg = getenv("NS_CLUSTER_DBG");
  a.dbg = dbg ? atoi(dbg) : 0;
}

// Both directions of a BiLSTM, whole sequence, one launch.  p0 must be the forward-in-time direction
// (reverse = 0) and p1 the reversed one.  `work` (ns_lstm_cluster_work_bytes) holds the exchange
// buffers; its first int is a status word: 0 ok, non-zero = a spin timed out (results invalid).
extern "C" int ns_lstm_cluster_fwd(const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1, void* work,
                                   ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  NS_CHECK_ARG(p0 && p1 && work, "ns_lstm_cluster_fwd: null");
  NS_CHECK_ARG(cluster_supported(p0, p1), "ns_lstm_cluster_fwd: needs bf16, H %% 64 == 0, H <= 512, T >= 2");
  NS_CHECK_ARG(p0->reverse == 0 && p1->reverse == 1 && p0->N == p1->N && p0->T == p1->T && p0->H == p1->H,
               "ns_lstm_cluster_fwd: p0 forward / p1 reversed with equal shapes expected");
  LstmClusterArgs a = {};
  fill(a, p0, p1, work);
  const size_t chains = 2 * (size_t)((a.N + 15) / 16);
  const size_t xbytes = chains * 2 * 16 * (size_t)(a.H / 2) * sizeof(u64);
  if (hipMemsetAsync(work, 0, 256 + xbytes, s) != hipSuccess) { ns_set_error("ns_lstm_cluster_fwd: memset failed"); return NS_ERR_LAUNCH; }
  const size_t lds = (size_t)16 * a.H * 2;
  hipLaunchKernelGGL(lstm_cluster_fwd_kernel, dim3((unsigned)(chains * a.CS)), dim3(CTHREADS), lds, s, a);
  NS_CHECK_LAUNCH("lstm_cluster_fwd");
  return NS_OK;
}

extern "C" int ns_lstm_cluster_bwd(const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1, void* work,
                                   ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  NS_CHECK_ARG(p0 && p1 && work, "ns_lstm_cluster_bwd: null");
  NS_CHECK_ARG(cluster_supported(p0, p1), "ns_lstm_cluster_bwd: needs bf16, H %% 64 == 0, H <= 512, T >= 2");
  NS_CHECK_ARG(p0->reverse == 0 && p1->reverse == 1 && p0->N == p1->N && p0->T == p1->T && p0->H == p1->H,
               "ns_lstm_cluster_bwd: p0 forward / p1 reversed with equal shapes expected");
  LstmClusterArgs a = {};
  fill(a, p0, p1, work);
  const size_t chains = 2 * (size_t)((a.N + 15) / 16);
  const size_t xbytes = chains * 2 * 16 * (size_t)(4 * a.H / 2) * sizeof(u64);
  if (hipMemsetAsync(work, 0, 256 + xbytes, s) != hipSuccess) { ns_set_error("ns_lstm_cluster_bwd: memset failed"); return NS_ERR_LAUNCH; }
  const size_t lds = (size_t)16 * 4 * a.H * 2 + sizeof(float) * CW * 16 * 65;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)lstm_cluster_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(lstm_cluster_bwd_kernel, dim3((unsigned)(chains * a.CS)), dim3(CTHREADS), lds, s, a);
  NS_CHECK_LAUNCH("lstm_cluster_bwd");
  return NS_OK;
}
