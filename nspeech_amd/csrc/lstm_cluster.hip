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
#include <stdint.h>

typedef unsigned long long u64;
constexpr int CW = 8;            // waves per workgroup
constexpr int CTHREADS = CW * 64;

struct LstmClusterArgs {
  int N, T, H, P, padl, CS;
  // per direction d (0 = forward in time, 1 = reversed)
  const float* xg[2]; long ld_xg;
  const bf16_t* whT[2];           // [4H, H]
  const bf16_t* wh[2];            // [H, 4H] (backward)
  bf16_t* h[2]; long ld_h;        // h[d] already offset to this direction's columns
  float* c[2];
  bf16_t* gates[2];
  // fp32-state forward (lstm_cluster3_fwd_kernel): pre-split recurrent weights, fp32 h out, optional bf16 copy of h
  const bf16_t* whT_hi[2]; const bf16_t* whT_lo[2];
  float* hf[2]; bf16_t* hb[2]; long ld_hb;
  const float* dh[2]; long ld_dh; // backward: grad wrt h outputs (offset to direction's columns)
  bf16_t* dgates[2];
  const int* lengths;
  float forget_bias, cell_clip;
  u64* xbuf;                      // [chains][2][16][granules per row]
  int* status;
  int dbg;                        // timing experiments only (NS_CLUSTER_DBG), 0 in production
  unsigned* flags;                // role-split backward: [chains][8] published-step counters (zeroed per launch)
  long long* trace;               // dbg bit 4: [step][8] timestamps of workgroup 0 (100 MHz clock)
};

__device__ __forceinline__ int swz_off(int row, int k, int H) {   // bf16 element offset in the LDS h image
  const int chunk = k >> 3;
  const int m = ((H >> 3) & 15) ? 7 : 15;   // the XOR must stay inside an aligned group of chunks of the row
  return row * H + (((chunk ^ (row & m)) << 3) | (k & 7));
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
        unsigned spins = 0, clk0 = 0;
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
          if (!ok && (++spins & 1023u) == 0 && ns_spin_timed_out(clk0)) { atomicExch(a.status, 1); ok = true; }
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
      float cn = ns_cell_clip(gf * cst[r] + gi * gj, a.cell_clip);
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
        unsigned spins = 0, clk0 = 0;
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
          if (!okk && (++spins & 1023u) == 0 && ns_spin_timed_out(clk0)) { atomicExch(a.status, 2); okk = true; }
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

// ==================================================================== role-split kernels (H <= 256)
// Same clustering and exchange protocol as above, with two changes that take the exchange latency
// (~2 us publish -> gathered, most of a step) off the critical path:
//
//  * ROLES.  No wave mixes global loads and global stores (gfx9 has one vmcnt for both, so a polling
//    load issued behind a store waits for the store's acknowledgement):
//      - XW compute waves (16 units each): MFMA from LDS, cell update, publish + saves.  Stores only.
//      - poller wave(s): spin on the exchange granules and drop them into the LDS operand image
//        (double-buffered).  Loads only.
//      - one prefetcher wave: streams the per-step operands (xg rows / saved gates, dh, c) one slot
//        ahead into an LDS stage.  Loads only.
//    One workgroup barrier per slot hands the LDS images over.
//  * INTERLEAVED CHAINS.  A workgroup serves R row groups of the same direction with the same resident
//    weights, round-robin: while row group A's new h is in flight to the peers, the compute waves work
//    on row group B.  Slot q = step * R + rg.
//
// Exchange layout (per chain and parity): a publishing lane's granules are contiguous, so one base
// register + immediates address them; the poller decodes granule index -> (row, k) when it fills LDS.
//
// LDS hand-over audit (round 3; the class of the attention-backward race fixed in 844cc13 - an image filled by one
// role and first read by another with no barrier in between).  Every image below is double-buffered by slot parity and
// handed over by the ONE wg_barrier() per slot that every role joins (wg_barrier waits for lgkmcnt(0) first, so a
// role's LDS writes AND reads of the slot have completed when it arrives):
//   forward   hs[q&1]   pollers fill it in front of barrier q, compute waves read it behind barrier q; the pollers'
//                       next fill of the same image is for slot q+2, behind barrier q+1, which the compute waves join
//                       after their reads of slot q
//             xgs[q&1]  prefetcher: slot 0 in front of barrier 0, slot q+1 behind barrier q; read behind barrier q+1
//             svs[q&1]  compute waves write it in slot q, the saver reads it behind barrier q+1 and joins barrier q+2
//                       (after its reads returned) before the compute waves write that image again in slot q+2
//             abortf    zeroed by wave 0 in front of its first wg_barrier, read by every role behind that barrier
//   backward  dgs[q&1], ops stage[q&1]: as hs / xgs;  c0: prefetcher in front of barrier 0, read in slot 0
//             outs[q&1] compute waves write, then add to `ready` with release; the publisher acquires `ready` >= 4(q+1)
//                       before it reads; its reads complete (the stores need the data) before it joins the next slot
//                       barrier, two barriers before the compute waves write that image again
//             ready     zeroed by wave 0 in front of barrier 0; the publisher joins barrier 0 before its first look
// The single-role kernels above use one image and __syncthreads() on both sides of every use.
constexpr int XW = 4;
constexpr int FW_POLL = 2;               // poller waves (XW and XW + 3): half of the granules each, so a sweep is half as long
                                         // (expand BiLSTM forward 3.2 -> 3.0 ms)
constexpr int FW_WAVES = XW + 2 + FW_POLL;   // compute, poller, prefetcher, saver, second poller
constexpr int BW_POLL = 2;
constexpr int BW_WAVES = XW + 1 + BW_POLL + 1;   // compute, publisher, pollers, prefetcher
constexpr int XG_LD = 256 + 4;          // floats per row of the xg stage (pad: rows 4 apart hit different banks)

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  const bf16_t b0 = (bf16_t)lo, b1 = (bf16_t)hi;
  return (unsigned)(*(const unsigned short*)&b0) | ((unsigned)(*(const unsigned short*)&b1) << 16);
}

// Round 3 form.  The product is taken TRANSPOSED - the resident weights are the MFMA's A operand (M = the wave's 16
// units), the gathered h its B operand (N = the 16 batch rows) - so a lane's accumulators are 4 consecutive UNITS of
// one batch row: the new h packs into the exchange's {tag, 2 x bf16} unit-pair granules in the lane (two full-wave
// stores, no cross-lane shuffle), the xg operands are one 16-byte LDS read per gate instead of 16 scalar ones, and the
// saver's images take 6 vector writes instead of 24.  The workgroup's own h block goes into the next operand image
// directly (LDS), only the peers' blocks are polled.  The xg operands of slot q + 1 are read at the end of slot q (the
// prefetcher runs one interval ahead), so behind the slot barrier the chain starts with the MFMAs.
// (audit) one barrier per slot, joined by every role: hs[q&1] is filled by the pollers (peers' blocks) and the compute
// waves (own block: R = 1 in slot q-1 behind barrier(q-1), R = 2 behind barrier(q-1) of the slot after the producing
// one) in front of barrier(q), read behind it, refilled behind barrier(q+1); xgs[(q+1)&1] is stored between
// barrier(q-1) and barrier(q), read between barrier(q) and barrier(q+1); svs as before.
template <int HB, int R>
__global__ __launch_bounds__(FW_WAVES * 64) void lstm_cluster2_fwd_kernel(LstmClusterArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int H = HB * 64, KS = H / 32, CS = HB;
  constexpr int GPD = XW * 64 * 2;                              // granules per source workgroup and slot: [wave][lane][2]
  bf16_t* hs = (bf16_t*)smem;                                   // [2][16][H] swizzled
  float* xgs = (float*)(smem + (size_t)2 * 16 * H * 2);         // [2][16][XG_LD]: row, gate * 64 + unit
  // this slot's results for the saver wave: h bf16 [16][64], c f32 [16][64], gates bf16 [16][4][64]  (2 + 4 + 8 KB)
  constexpr int SV_H = 16 * 64 * 2, SV_C = 16 * 64 * 4, SV_G = 16 * 4 * 64 * 2, SV_BYTES = SV_H + SV_C + SV_G;
  char* svs = (char*)(xgs + 2 * 16 * XG_LD);                    // [2][SV_BYTES]
  int* abortf = (int*)(svs + 2 * SV_BYTES);                     // [2]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsets = ((a.N + 15) / 16 + R - 1) / R;
  const int set = blockIdx.x / CS, wgc = blockIdx.x % CS;
  const int d = set / nsets, rg0 = (set % nsets) * R;           // this workgroup serves row groups rg0 .. rg0+R-1
  const int r16 = lane & 15, g = lane >> 4;
  u64* xb0 = a.xbuf + (size_t)(d * nsets * R + rg0) * 2 * CS * GPD;   // + rg * 2*CS*GPD + parity * CS*GPD + source * GPD
  const int u0 = wgc * 64;
  const int T = a.T, Q = a.T * R;
  if (tid < 3) abortf[tid] = 0;

  if (wave < XW) {
    // ================================================================ compute role: batch row r16, units wq .. wq + 3
    const int wq = wave * 16 + g * 4;                  // first of the lane's 4 units inside the workgroup's 64
    bf16x8 bw[4][KS];                                  // A fragments: row (unit) wave * 16 + r16, k chunk g
    {
      const bf16_t* W = a.whT[d];
#pragma unroll
      for (int gate = 0; gate < 4; ++gate) {
        const bf16_t* row = W + ((long)gate * H + u0 + wave * 16 + r16) * H;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bw[gate][ks] = *(const bf16x8*)(row + ks * 32 + g * 8);
      }
    }
    float cst[R][4];
    int len[R];
    uint2 pend[R];                                     // R = 2: the own h block waits for the next slot's barrier
#pragma unroll
    for (int rg = 0; rg < R; ++rg) {
      const int n = (rg0 + rg) * 16 + r16;
      len[rg] = (a.lengths && n < a.N) ? a.lengths[n] : T;
      pend[rg] = make_uint2(0u, 0u);
#pragma unroll
      for (int r = 0; r < 4; ++r) cst[rg][r] = 0.f;
    }
    f32x4 acc[4];
    auto load_xg = [&](int q) {
      const float* xr = xgs + ((size_t)(q & 1) * 16 + r16) * XG_LD + wq;
#pragma unroll
      for (int gate = 0; gate < 4; ++gate) acc[gate] = *(const f32x4*)(xr + gate * 64);
    };
    wg_barrier();                                      // xg of slot 0 and abortf are in place
    load_xg(0);
    for (int step = 0; step < T; ++step) {
      const int t = d ? T - 1 - step : step;
#pragma unroll
      for (int rg = 0; rg < R; ++rg) {
        const int q = step * R + rg, buf = q & 1;
        wg_barrier();
        if (abortf[buf]) return;
        const bool tr = (a.dbg & 16) && blockIdx.x == 0 && tid == 0 && q < 512;
        if (tr) a.trace[q * 8 + 0] = wall_clock64();
        if (R == 2 && q > 0)                           // the own block of slot q - 1, for slot q + 1
          *(uint2*)(hs + (size_t)((q + 1) & 1) * 16 * H + swz_off(r16, u0 + wq, H)) = pend[(rg + 1) % R];
        if (step > 0) {
          const bf16_t* hb = hs + (size_t)buf * 16 * H;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 hf = *(const bf16x8*)(hb + swz_off(r16, ks * 32 + g * 8, H));
#pragma unroll
            for (int gate = 0; gate < 4; ++gate)
              acc[gate] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[gate][ks], hf, acc[gate], 0, 0, 0);
          }
        }
        const bool masked = t >= len[rg];
        float hv[4], sg[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gi = sigmoidf_(acc[0][r]), gj = tanhf_(acc[1][r]);
          const float gf = sigmoidf_(acc[2][r] + a.forget_bias), go = sigmoidf_(acc[3][r]);
          float cn = ns_cell_clip(gf * cst[rg][r] + gi * gj, a.cell_clip);
          float hn = go * tanhf_(cn);
          if (masked) { cn = 0.f; hn = 0.f; }
          cst[rg][r] = cn;
          hv[r] = hn;
          sg[r][0] = masked ? 0.f : gi; sg[r][1] = masked ? 0.f : gj; sg[r][2] = masked ? 0.f : gf; sg[r][3] = masked ? 0.f : go;
        }
        if (tr) a.trace[q * 8 + 1] = wall_clock64();
        // publish first (tag = step + 1): units (wq, wq + 1) and (wq + 2, wq + 3) of row r16
        uint2 hp;
        hp.x = pack_bf16(hv[0], hv[1]);
        hp.y = pack_bf16(hv[2], hv[3]);
        if (step + 1 < T) {
          if (CS > 1) {
            u64* nxt = xb0 + ((size_t)rg * 2 + ((step + 1) & 1)) * CS * GPD + (size_t)wgc * GPD + (wave * 64 + lane) * 2;
            __hip_atomic_store(nxt, ((u64)(unsigned)(step + 1) << 32) | hp.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(nxt + 1, ((u64)(unsigned)(step + 1) << 32) | hp.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (R == 1) *(uint2*)(hs + (size_t)((q + 1) & 1) * 16 * H + swz_off(r16, u0 + wq, H)) = hp;
          else pend[rg] = hp;
        }
        if (tr) a.trace[q * 8 + 2] = wall_clock64();
        // results for the backward pass / the consumers of h go to LDS; the saver wave writes them out
        {
          char* sv = svs + (size_t)buf * SV_BYTES;
          *(uint2*)((bf16_t*)sv + r16 * 64 + wq) = hp;
          *(f32x4*)((float*)(sv + SV_H) + r16 * 64 + wq) = (f32x4){cst[rg][0], cst[rg][1], cst[rg][2], cst[rg][3]};
          bf16_t* gp = (bf16_t*)(sv + SV_H + SV_C) + r16 * 4 * 64 + wq;
#pragma unroll
          for (int gate = 0; gate < 4; ++gate) {
            uint2 pk;
            pk.x = pack_bf16(sg[0][gate], sg[1][gate]);
            pk.y = pack_bf16(sg[2][gate], sg[3][gate]);
            *(uint2*)(gp + gate * 64) = pk;
          }
        }
        if (q + 1 < Q) load_xg(q + 1);
      }
    }
    wg_barrier();
  } else if (wave == XW || wave == XW + 3) {
    // ================================================================ poller role: the peers' h blocks
    constexpr int NPG = (CS - 1) * GPD;                // granules to gather per slot
    constexpr int PPG = NPG / (FW_POLL * 64) > 0 ? NPG / (FW_POLL * 64) : 1;
    static_assert(CS == 1 || NPG == PPG * FW_POLL * 64, "poller coverage");
    const int j0 = (wave == XW ? 0 : 1) * PPG;
    wg_barrier();
    for (int q = 0; q < Q; ++q) {
      const int step = q / R, rg = q % R, buf = q & 1;
      const bool tr = (a.dbg & 16) && blockIdx.x == 0 && lane == 0 && wave == XW && q < 512;
      if (tr) a.trace[q * 8 + 4] = wall_clock64();
      if (step > 0 && CS > 1) {
        const u64* cur = xb0 + ((size_t)rg * 2 + (step & 1)) * CS * GPD;   // published by the peers with tag = step
        u64 v[PPG];
        unsigned spins = 0, clk0 = 0;
        bool ok;
        do {
          ok = true;
#pragma unroll
          for (int j = 0; j < PPG; ++j) {
            const int li = lane + 64 * (j0 + j), sx = li / GPD;
            const int ws = sx < wgc ? sx : sx + 1;
            v[j] = __hip_atomic_load(cur + (size_t)ws * GPD + (li % GPD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int j = 0; j < PPG; ++j) ok = ok && ((unsigned)(v[j] >> 32) == (unsigned)step);
          if (!ok) {
            if ((++spins & 1023u) == 0) {
              if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { abortf[buf] = 1; ok = true; }
              else if (ns_spin_timed_out(clk0)) { atomicExch(a.status, 1); abortf[buf] = 1; ok = true; }
            }
          }
        } while (!ok);
        if (tr) { a.trace[q * 8 + 5] = wall_clock64(); a.trace[q * 8 + 6] = spins; }
        bf16_t* dst = hs + (size_t)buf * 16 * H;
#pragma unroll
        for (int j = 0; j < PPG; ++j) {
          // granule li of source ws: wave (li / 128) % 4, lane' = (li / 2) % 64 -> row lane' % 16, units 4 (lane' / 16) + 2 (li % 2)
          const int li = lane + 64 * (j0 + j), sx = li / GPD, lw = li % GPD;
          const int ws = sx < wgc ? sx : sx + 1;
          const int lp = (lw >> 1) & 63;
          const int k = ws * 64 + (lw >> 7) * 16 + (lp >> 4) * 4 + (lw & 1) * 2;
          *(unsigned*)(dst + swz_off(lp & 15, k, H)) = (unsigned)v[j];
        }
      }
      wg_barrier();
      if (abortf[buf]) return;
    }
    wg_barrier();
  } else if (wave == XW + 2) {
    // ================================================================ saver role (stores only)
    // Runs one slot behind the compute waves (slot q-1 is complete once barrier q has passed), so its store
    // issue overlaps their next slot.  Per slot: h 128 chunks of 16 B, c 256, gates 512 -> 14 per lane.
    auto save = [&](int q) {
      const int step = q / R, rg = q % R;
      const int t = d ? T - 1 - step : step;
      const int n0 = (rg0 + rg) * 16;
      const char* sv = svs + (size_t)(q & 1) * SV_BYTES;
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        const int idx = lane + 64 * j;
        const f32x4 v = *(const f32x4*)(sv + idx * 16);
        if (idx < 128) {
          const int row = idx >> 3, cc = idx & 7;
          if (n0 + row < a.N)
            *(f32x4*)(a.h[d] + ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)a.ld_h + (unsigned)(u0 + cc * 8))) = v;
        } else if (idx < 384) {
          const int jj = idx - 128, row = jj >> 4, cc = jj & 15;
          if (n0 + row < a.N)
            *(f32x4*)(a.c[d] + ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)H + (unsigned)(u0 + cc * 4))) = v;
        } else {
          const int jj = idx - 384, row = jj >> 5, gate = (jj >> 3) & 3, cc = jj & 7;
          if (n0 + row < a.N)
            *(f32x4*)(a.gates[d] + ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)(4 * H) + (unsigned)(gate * H + u0 + cc * 8))) = v;
        }
      }
    };
    wg_barrier();
    for (int q = 0; q < Q; ++q) {
      wg_barrier();
      if (abortf[q & 1]) return;
      if (q > 0) save(q - 1);
    }
    wg_barrier();
    save(Q - 1);
  } else {
    // ================================================================ prefetcher role, one interval ahead:
    // xg of slot q + 1 is in LDS before barrier(q).  Stage row j, gate = lane / 16, 4 floats at (lane % 16) * 4
    f32x4 pf[16];
    const float* xg = a.xg[d];
    const int pgate = lane >> 4, pf4 = (lane & 15) * 4;
    auto pf_load = [&](int q) {
      const int step = q / R, rg = q % R;
      const int t = d ? T - 1 - step : step;
      const int n0 = (rg0 + rg) * 16;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int n = n0 + j;
        pf[j] = n < a.N ? *(const f32x4*)(xg + ((unsigned)(n * a.P + a.padl + t) * (unsigned)a.ld_xg + (unsigned)(pgate * H + u0 + pf4)))
                        : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    };
    auto pf_store = [&](int buf) {
#pragma unroll
      for (int j = 0; j < 16; ++j) *(f32x4*)(xgs + ((size_t)buf * 16 + j) * XG_LD + pgate * 64 + pf4) = pf[j];
    };
    pf_load(0);
    pf_store(0);
    if (Q > 1) pf_load(1);
    wg_barrier();
    for (int q = 0; q < Q; ++q) {
      if (q + 1 < Q) {
        pf_store((q + 1) & 1);
        if (q + 2 < Q) pf_load(q + 2);
      }
      wg_barrier();
      if (abortf[q & 1]) return;
    }
    wg_barrier();
  }
}


// ==================================================================== fp32-state forward (round 3)
// The encoder BiLSTM of the `mixed` / `bf16x3` modes keeps its state in fp32 and forms the recurrent product as three
// split-bf16 MFMA passes (h = hi + lo, W = hi + lo: hi.hi + hi.lo + lo.hi); until round 3 that arrangement ran one
// launch per time step (160 + 160 launches of 6 us).  Same roles and interleaving as lstm_cluster2_fwd_kernel, with what
// the fp32 state changes:
//   * a granule carries ONE unit: {step tag, fp32 h}; the pollers split it into the (hi, lo) LDS images;
//   * the resident weights are twice the registers per unit (hi and lo planes), and a workgroup of 8 waves has 256 VGPRs
//     per lane: a compute wave owns 8 units (two MFMA tiles [i | j], [f | o]: 128 VGPRs of weights), a workgroup 32
//     units, a chain H / 32 workgroups; the cell update pairs lanes r16 and r16 + 8;
//   * the saver writes h as fp32 (+ an optional bf16 copy for the weight gradients) and the gates as bf16: the backward
//     pass of this arrangement is the bf16 kernel (single-pass products).
constexpr int X3W = 4, X3_POLL = 2, X3_UPW = 32;
constexpr int X3_WAVES = X3W + X3_POLL + 2;          // compute, pollers, prefetcher, saver: 8 waves, two per SIMD
constexpr int XG3_LD = 4 * X3_UPW + 4;

template <int HB, int R>        // HB = H / 64
__global__ __launch_bounds__(X3_WAVES * 64) void lstm_cluster3_fwd_kernel(LstmClusterArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int H = HB * 64, KS = H / 32, GPC = 16 * H;          // granules per chain and step
  constexpr int CS = H / X3_UPW;                                   // workgroups per chain
  constexpr int PPG = GPC / (X3_POLL * 64);                        // granules per poller lane (H / 8)
  static_assert(GPC % (X3_POLL * 64) == 0, "poller coverage");
  bf16_t* hsh = (bf16_t*)smem;                                    // [2][16][H] swizzled, high parts
  bf16_t* hsl = hsh + 2 * 16 * H;                                 // [2][16][H] low parts
  float* xgs = (float*)(hsl + 2 * 16 * H);                        // [2][16][XG3_LD]: row, gate * 32 + unit
  // this slot's results for the saver: h f32 [16][32], c f32 [16][32], gates bf16 [16][4][32], h bf16 [16][32]
  constexpr int SV_H = 16 * 32 * 4, SV_C = 16 * 32 * 4, SV_G = 16 * 4 * 32 * 2, SV_HB = 16 * 32 * 2;
  constexpr int SV_BYTES = SV_H + SV_C + SV_G + SV_HB;           // 9216
  char* svs = (char*)(xgs + 2 * 16 * XG3_LD);                     // [2][SV_BYTES]
  int* abortf = (int*)(svs + 2 * SV_BYTES);                       // [2]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsets = ((a.N + 15) / 16 + R - 1) / R;
  const int set = blockIdx.x / CS, wgc = blockIdx.x % CS;
  const int d = set / nsets, rg0 = (set % nsets) * R;
  const int r16 = lane & 15, g = lane >> 4;
  u64* xb0 = a.xbuf + (size_t)(d * nsets * R + rg0) * 2 * GPC;   // + rg * 2*GPC + parity * GPC
  const int u0 = wgc * X3_UPW;
  const int T = a.T, Q = a.T * R;
  if (tid < 2) abortf[tid] = 0;          // (audit) written by wave 0 in front of its first wg_barrier, read behind it

  if (wave < X3W) {
    // ================================================================ compute role
    const int ul = wave * 8 + (r16 & 7);                     // unit inside the workgroup's 32
    const bool cell_lane = r16 < 8;
    const int pub0 = (((wgc * X3W + wave) * 32) + g * 8 + (r16 & 7)) * 4;    // this lane's 4 granules (rows g*4 .. g*4+3)
    bf16x8 bwh[2][KS], bwl[2][KS];
    {
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        const int gate = tl * 2 + (r16 >> 3);
        const long wrow = ((long)gate * H + u0 + ul) * H + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bwh[tl][ks] = *(const bf16x8*)(a.whT_hi[d] + wrow + ks * 32);
          bwl[tl][ks] = *(const bf16x8*)(a.whT_lo[d] + wrow + ks * 32);
        }
      }
    }
    float cst[R][4];
    int len[R][4];
#pragma unroll
    for (int rg = 0; rg < R; ++rg)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (rg0 + rg) * 16 + g * 4 + r;
        cst[rg][r] = 0.f;
        len[rg][r] = (a.lengths && n < a.N) ? a.lengths[n] : T;
      }
    for (int step = 0; step < T; ++step) {
      const int t = d ? T - 1 - step : step;
#pragma unroll
      for (int rg = 0; rg < R; ++rg) {
        const int q = step * R + rg, buf = q & 1;
        wg_barrier();
        if (abortf[buf]) return;
        // tile 0 = [i | j], tile 1 = [f | o]: column r16 -> gate 2*tile + (r16 >> 3), unit ul
        const float* xr = xgs + (size_t)buf * 16 * XG3_LD + (r16 >> 3) * X3_UPW + ul;
        f32x4 accA, accB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          accA[r] = xr[(g * 4 + r) * XG3_LD];
          accB[r] = xr[(g * 4 + r) * XG3_LD + 2 * X3_UPW];
        }
        if (step > 0) {
          const bf16_t* hbh = hsh + (size_t)buf * 16 * H;
          const bf16_t* hbl = hsl + (size_t)buf * 16 * H;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int so = swz_off(r16, ks * 32 + g * 8, H);
            const bf16x8 ah = *(const bf16x8*)(hbh + so), al = *(const bf16x8*)(hbl + so);
            accA = mfma_split<3>(ah, al, bwh[0][ks], bwl[0][ks], accA);
            accB = mfma_split<3>(ah, al, bwh[1][ks], bwl[1][ks], accB);
          }
        }
        float hv[4], sg[4][4], cn4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float zj = __shfl_down(accA[r], 8, 64), zo = __shfl_down(accB[r], 8, 64);
          const bool masked = t >= len[rg][r];
          const float gi = sigmoidf_(accA[r]), gj = tanhf_(zj), gf = sigmoidf_(accB[r] + a.forget_bias), go = sigmoidf_(zo);
          float cn = ns_cell_clip(gf * cst[rg][r] + gi * gj, a.cell_clip);
          float hn = go * tanhf_(cn);
          if (masked) { cn = 0.f; hn = 0.f; }
          cst[rg][r] = cn;
          cn4[r] = cn;
          hv[r] = hn;
          sg[r][0] = masked ? 0.f : gi; sg[r][1] = masked ? 0.f : gj; sg[r][2] = masked ? 0.f : gf; sg[r][3] = masked ? 0.f : go;
        }
        // publish first (tag = step + 1): one granule per (row, unit), fp32
        if (step + 1 < T && cell_lane) {
          u64* nxt = xb0 + ((size_t)rg * 2 + ((step + 1) & 1)) * GPC + pub0;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __hip_atomic_store(nxt + r, ((u64)(unsigned)(step + 1) << 32) | (u64)__float_as_uint(hv[r]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        // results for the backward pass / the consumers of h go to LDS; the saver wave writes them out
        if (cell_lane) {
          char* sv = svs + (size_t)buf * SV_BYTES;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = g * 4 + r;
            ((float*)sv)[row * X3_UPW + ul] = hv[r];
            ((float*)(sv + SV_H))[row * X3_UPW + ul] = cn4[r];
            bf16_t* gp = (bf16_t*)(sv + SV_H + SV_C) + row * 4 * X3_UPW + ul;
#pragma unroll
            for (int gate = 0; gate < 4; ++gate) gp[gate * X3_UPW] = (bf16_t)sg[r][gate];
            ((bf16_t*)(sv + SV_H + SV_C + SV_G))[row * X3_UPW + ul] = (bf16_t)hv[r];
          }
        }
      }
    }
    wg_barrier();
  } else if (wave < X3W + X3_POLL) {
    // ================================================================ poller role: granule lane + 64 * jj of the chain
    // decodes as r = lane & 3, unit in wave = (lane >> 2) & 7, g = ((jj & 1) << 1) | (lane >> 5), publishing wave =
    // (jj >> 1) & 3, publishing workgroup = jj >> 3
    const int j0 = (wave - X3W) * PPG;
    const int pr_ = lane & 3, pu = (lane >> 2) & 7, pgl = lane >> 5;
    for (int q = 0; q < Q; ++q) {
      const int step = q / R, rg = q % R, buf = q & 1;
      if (step > 0) {
        const u64* cur = xb0 + ((size_t)rg * 2 + (step & 1)) * GPC;
        u64 v[PPG];
        unsigned spins = 0, clk0 = 0;
        bool ok;
        do {
          ok = true;
#pragma unroll
          for (int j = 0; j < PPG; ++j) v[j] = __hip_atomic_load(cur + lane + (j0 + j) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int j = 0; j < PPG; ++j) ok = ok && ((unsigned)(v[j] >> 32) == (unsigned)step);
          if (!ok) {
            if ((++spins & 1023u) == 0) {
              if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { abortf[buf] = 1; ok = true; }
              else if (ns_spin_timed_out(clk0)) { atomicExch(a.status, 1); abortf[buf] = 1; ok = true; }
            }
          }
        } while (!ok);
        bf16_t* dh_ = hsh + (size_t)buf * 16 * H;
        bf16_t* dl_ = hsl + (size_t)buf * 16 * H;
#pragma unroll
        for (int j = 0; j < PPG; ++j) {
          const int jj = j0 + j;
          const int row = (((jj & 1) << 1) | pgl) * 4 + pr_;
          const int k = (jj >> 3) * X3_UPW + ((jj >> 1) & 3) * 8 + pu;
          const float x = __uint_as_float((unsigned)v[j]);
          const bf16_t hi = (bf16_t)x;
          const int so = swz_off(row, k, H);
          dh_[so] = hi;
          dl_[so] = (bf16_t)(x - (float)hi);
        }
      }
      wg_barrier();
      if (abortf[buf]) return;
    }
    wg_barrier();
  } else if (wave == X3W + X3_POLL) {
    // ================================================================ saver role (stores only), one slot behind
    // per slot: h f32 128 chunks of 16 B, c 128, gates 256, h bf16 64 -> 9 per lane
    auto save = [&](int q) {
      const int step = q / R, rg = q % R;
      const int t = d ? T - 1 - step : step;
      const int n0 = (rg0 + rg) * 16;
      const char* sv = svs + (size_t)(q & 1) * SV_BYTES;
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const int idx = lane + 64 * j;
        const f32x4 v = *(const f32x4*)(sv + idx * 16);
        if (idx < 128) {
          const int row = idx >> 3, cc = idx & 7;
          if (n0 + row < a.N)
            *(f32x4*)(a.hf[d] + ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)a.ld_h + (unsigned)(u0 + cc * 4))) = v;
        } else if (idx < 256) {
          const int jj = idx - 128, row = jj >> 3, cc = jj & 7;
          if (n0 + row < a.N)
            *(f32x4*)(a.c[d] + ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)H + (unsigned)(u0 + cc * 4))) = v;
        } else if (idx < 512) {
          const int jj = idx - 256, row = jj >> 4, gate = (jj >> 2) & 3, cc = jj & 3;
          if (n0 + row < a.N)
            *(f32x4*)(a.gates[d] + ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)(4 * H) + (unsigned)(gate * H + u0 + cc * 8))) = v;
        } else {
          const int jj = idx - 512, row = jj >> 2, cc = jj & 3;
          if (a.hb[d] && n0 + row < a.N)
            *(f32x4*)(a.hb[d] + ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)a.ld_hb + (unsigned)(u0 + cc * 8))) = v;
        }
      }
    };
    for (int q = 0; q < Q; ++q) {
      wg_barrier();
      if (abortf[q & 1]) return;
      if (q > 0) save(q - 1);
    }
    wg_barrier();
    save(Q - 1);
  } else {
    // ================================================================ prefetcher role: 16 rows x 4 gates x 32 units = 512
    // float4 per slot, 8 per lane: row 2j + (lane >> 5), gate (lane >> 3) & 3, 4 floats at (lane & 7) * 4
    f32x4 pf[8];
    const float* xg = a.xg[d];
    const int prow = lane >> 5, pgate = (lane >> 3) & 3, pf4 = (lane & 7) * 4;
    auto pf_load = [&](int q) {
      const int step = q / R, rg = q % R;
      const int t = d ? T - 1 - step : step;
      const int n0 = (rg0 + rg) * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = n0 + 2 * j + prow;
        pf[j] = n < a.N ? *(const f32x4*)(xg + ((unsigned)(n * a.P + a.padl + t) * (unsigned)a.ld_xg + (unsigned)(pgate * H + u0 + pf4)))
                        : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    };
    auto pf_store = [&](int buf) {
#pragma unroll
      for (int j = 0; j < 8; ++j) *(f32x4*)(xgs + ((size_t)buf * 16 + 2 * j + prow) * XG3_LD + pgate * X3_UPW + pf4) = pf[j];
    };
    pf_load(0);
    pf_store(0);
    if (Q > 1) pf_load(1);
    for (int q = 0; q < Q; ++q) {
      wg_barrier();
      if (abortf[q & 1]) return;
      if (q + 1 < Q) {
        pf_store((q + 1) & 1);
        if (q + 2 < Q) pf_load(q + 2);
      }
    }
    wg_barrier();
  }
}

// Backward.  Compute wave w owns 16 units and the full K = 4H contraction for them (W_h rows as B
// fragments, K/32 k-steps), so no cross-wave reduction is needed.
//
// The backward exchange is 4x the forward one (every workgroup needs all 4H gate gradients of the
// step after), and a granule sweep of that size is bound by the CU's outstanding-miss budget
// (~12 GB/s).  So the payload travels dense instead: the gate gradients are saved to the dgates
// array anyway (the weight-gradient GEMMs read them later), so a PUBLISHER wave writes this
// workgroup's slice there with 16-byte write-through (sc1) stores, drains them, and raises one flag
// per (chain, workgroup); the pollers wait for the cluster's flags and read the rows back with
// 16-byte sc1 loads straight into the LDS operand image.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int HB, int R>
__global__ __launch_bounds__(BW_WAVES * 64) void lstm_cluster2_bwd_kernel(LstmClusterArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int H = HB * 64, K = 4 * H, KS = K / 32;
  constexpr int CPL = K / 64;                                         // 16-byte chunks per poller lane: 16 * (K/8) / 128
  const int CS = HB;
  bf16_t* dgs = (bf16_t*)smem;                                        // [2][16][K] swizzled
  char* ops = smem + (size_t)2 * 16 * K * 2;                          // [2] stages of {gates, dh, cprev}
  constexpr int OPS_STAGE = 16 * 4 * 64 * 2 + 2 * 16 * 64 * 4;        // 16 KB
  bf16_t* outs = (bf16_t*)(ops + 2 * OPS_STAGE);                      // [2][16][4][64] this slot's gate gradients
  float* c0 = (float*)(outs + 2 * 16 * 4 * 64);                       // [R][16][64] cell state at the first processed step
  int* abortf = (int*)(c0 + R * 16 * 64);                             // [2] + [2] = compute waves done with their slot (counter)
  int* ready = abortf + 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsets = ((a.N + 15) / 16 + R - 1) / R;
  const int set = blockIdx.x / CS, wgc = blockIdx.x % CS;
  const int d = set / nsets, rg0 = (set % nsets) * R;
  const int r16 = lane & 15, g = lane >> 4;
  unsigned* flags0 = a.flags + (size_t)(d * nsets * R + rg0) * 8;     // [chain][8]: published backward steps per workgroup
  const int u0 = wgc * 64;
  const int T = a.T, Q = a.T * R;
  if (tid < 3) abortf[tid] = 0;                                        // abortf[0..1], ready
  auto t_of = [&](int step) { return d ? T - 1 - step : step; };

  if (wave < XW) {
    // ================================================================ compute role (LDS only)
    const int wu = wave * 16 + r16;                    // unit inside the workgroup's 64
    bf16x8 bw[KS];
    {
      const bf16_t* row = a.wh[d] + (long)(u0 + wu) * K;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bw[ks] = *(const bf16x8*)(row + ks * 32 + g * 8);
    }
    int asw[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) asw[m] = r16 * K + (((m * 4 + g) ^ r16) << 3);
    float dcc[R][4], pc[R][4];
    int len[R][4];
#pragma unroll
    for (int rg = 0; rg < R; ++rg)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (rg0 + rg) * 16 + g * 4 + r;
        dcc[rg][r] = 0.f; pc[rg][r] = 0.f;
        len[rg][r] = (a.lengths && n < a.N) ? a.lengths[n] : T;
      }
    for (int bs = 0; bs < T; ++bs) {                 // backward step index; forward step = T-1-bs
      const int t = t_of(T - 1 - bs);
#pragma unroll
      for (int rg = 0; rg < R; ++rg) {
        const int q = bs * R + rg, buf = q & 1;
        const int n0 = (rg0 + rg) * 16;
        wg_barrier();
        if (abortf[buf]) return;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (bs > 0) {
          const bf16_t* db = dgs + (size_t)buf * 16 * K;
          f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ks += 2) {
            // swz_off(r16, ks*32 + g*8): the XOR only touches the low 4 chunk bits -> 4 lane bases + immediates
            const bf16x8 a0 = *(const bf16x8*)(db + asw[ks & 3] + (ks >> 2) * 128);
            const bf16x8 a1 = *(const bf16x8*)(db + asw[(ks + 1) & 3] + ((ks + 1) >> 2) * 128);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bw[ks], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bw[ks + 1], acc2, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
        }
        const char* st = ops + (size_t)buf * OPS_STAGE;
        const bf16_t* sgt = (const bf16_t*)st;
        const float* sdh = (const float*)(st + 8192);
        const float* scp = (const float*)(st + 12288);
        bf16_t* so = outs + (size_t)buf * 16 * 4 * 64;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = g * 4 + r;
          const int n = n0 + row;
          const float gi = (float)sgt[(row * 4 + 0) * 64 + wu], gj = (float)sgt[(row * 4 + 1) * 64 + wu];
          const float gf = (float)sgt[(row * 4 + 2) * 64 + wu], go = (float)sgt[(row * 4 + 3) * 64 + wu];
          const float cprev = scp[row * 64 + wu];
          const float ccur = bs == 0 ? c0[(rg * 16 + row) * 64 + wu] : pc[rg][r];
          const float dh = sdh[row * 64 + wu] + acc[r];
          const float tc = tanhf_(ccur);
          const float d_o = dh * tc * go * (1.f - go);
          const float dc = dh * go * (1.f - tc * tc) + dcc[rg][r];
          float dgv[4] = {dc * gj * gi * (1.f - gi), dc * gi * (1.f - gj * gj), dc * cprev * gf * (1.f - gf), d_o};
          dcc[rg][r] = dc * gf;
          if (t >= len[rg][r] || n >= a.N) {
            dgv[0] = dgv[1] = dgv[2] = dgv[3] = 0.f;
            dcc[rg][r] = 0.f;
          }
          pc[rg][r] = cprev;
#pragma unroll
          for (int j = 0; j < 4; ++j) so[(row * 4 + j) * 64 + wu] = (bf16_t)dgv[j];
        }
        // tell the publisher this wave's part of the slot is in LDS (release: the writes above are ordered before it)
        if (lane == 0) __hip_atomic_fetch_add(ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    wg_barrier();
  } else if (wave == XW) {
    // ================================================================ publisher role (stores only)
    // slot q's tile: 64 (row, gate) lines of 128 B = 512 chunks of 16 B, 8 per lane; LDS offset = chunk * 16
    const long dg_bytes = (long)a.N * a.P * K * 2;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dgates[d], 0, (int)dg_bytes, 0x00020000);
    wg_barrier();
    for (int q = 0; q < Q; ++q) {
      const int bs = q / R, rg = q % R;
      const int t = t_of(T - 1 - bs);
      const int n0 = (rg0 + rg) * 16;
      // wait for the XW compute waves of slot q (LDS counter)
      unsigned spins = 0, clk0 = 0;
      while (__hip_atomic_load(ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < XW * (q + 1)) {
        __builtin_amdgcn_s_sleep(1);
        if (((++spins) & 255) == 0) {
          if (abortf[0] | abortf[1]) return;
          if ((spins & 1023u) == 0 && ns_spin_timed_out(clk0)) { atomicExch(a.status, 3); return; }
        }
      }
      const char* so = (const char*)(outs + (size_t)(q & 1) * 16 * 4 * 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = lane + 64 * j, c8 = idx & 7, gate = (idx >> 3) & 3, row = idx >> 5;
        const u32x4 v = *(const u32x4*)(so + idx * 16);
        if (n0 + row < a.N) {
          const unsigned off = ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)K + (unsigned)(gate * H + u0 + c8 * 8)) * 2u;
          __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 16);           // aux 16 = sc1 (write-through)
        }
      }
      // R >= 2: join the slot barrier first (the compute waves are waiting there; the flag is not needed before the
      // slot after next).  R == 1: the pollers need this flag to reach the barrier at all.
      if (R > 1) wg_barrier();
      if (bs + 1 < T) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // drain before the flag
        if (lane == 0) __hip_atomic_store(flags0 + rg * 8 + wgc, (unsigned)(bs + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (R == 1) wg_barrier();
    }
  } else if (wave < XW + 1 + BW_POLL) {
    // ================================================================ poller role (loads only)
    const int pl = (wave - XW - 1) * 64 + lane;
    const long dg_bytes = (long)a.N * a.P * K * 2;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dgates[d], 0, (int)dg_bytes, 0x00020000);
    for (int q = 0; q < Q; ++q) {
      const int bs = q / R, rg = q % R, buf = q & 1;
      if (bs > 0) {
        // every workgroup of the chain must have published backward step bs-1 (flag >= bs)
        const unsigned* fl = flags0 + rg * 8;
        unsigned spins = 0, clk0 = 0;
        for (;;) {
          const unsigned v = lane < CS ? __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
          if (__all(v >= (unsigned)bs)) break;
          if ((++spins & 1023u) == 0) {
            if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { abortf[buf] = 1; break; }
            if (ns_spin_timed_out(clk0)) { atomicExch(a.status, 2); abortf[buf] = 1; break; }
          }
        }
        const int tn = t_of(T - bs);                   // time index of the step processed just before
        const int n0 = (rg0 + rg) * 16;
        bf16_t* dst = dgs + (size_t)buf * 16 * K;
        u32x4 v[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const int idx = pl + 128 * j, row = idx / (K / 8), ch = idx % (K / 8);
          v[j] = (u32x4){0u, 0u, 0u, 0u};
          if (n0 + row < a.N) {
            const unsigned off = ((unsigned)((n0 + row) * a.P + a.padl + tn) * (unsigned)K + (unsigned)(ch * 8)) * 2u;
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16);       // sc1: bypasses this CU's L1
          }
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const int idx = pl + 128 * j, row = idx / (K / 8), ch = idx % (K / 8);
          *(u32x4*)(dst + swz_off(row, ch * 8, K)) = v[j];
        }
      }
      wg_barrier();
      if (abortf[buf]) return;
    }
    wg_barrier();
  } else {
    // ================================================================ prefetcher role (loads only)
    // gates 8 x 16 B per lane, dh 4 x 16 B, cprev 4 x 16 B per slot
    f32x4 pg[8], pd[4], pcp[4];
    auto pf_load = [&](int q) {
      const int bs = q / R, rg = q % R, step = T - 1 - bs;
      const int t = t_of(step), tp = d ? t + 1 : t - 1;
      const bool has_prev = step > 0;
      const int n0 = (rg0 + rg) * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = lane + 64 * j, c8 = idx & 7, gate = (idx >> 3) & 3, row = idx >> 5;
        const int n = n0 + row;
        pg[j] = n < a.N ? *(const f32x4*)(a.gates[d] + ((unsigned)(n * a.P + a.padl + t) * (unsigned)(4 * H) + (unsigned)(gate * H + u0 + c8 * 8)))
                        : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = lane + 64 * j, c16 = idx & 15, row = idx >> 4;
        const int n = n0 + row;
        pd[j] = n < a.N ? *(const f32x4*)(a.dh[d] + ((unsigned)(n * a.P + a.padl + t) * (unsigned)a.ld_dh + (unsigned)(u0 + c16 * 4)))
                        : (f32x4){0.f, 0.f, 0.f, 0.f};
        pcp[j] = (n < a.N && has_prev) ? *(const f32x4*)(a.c[d] + ((unsigned)(n * a.P + a.padl + tp) * (unsigned)H + (unsigned)(u0 + c16 * 4)))
                                       : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    };
    auto pf_store = [&](int buf) {
      char* st = ops + (size_t)buf * OPS_STAGE;
#pragma unroll
      for (int j = 0; j < 8; ++j) *(f32x4*)(st + (size_t)(lane + 64 * j) * 16) = pg[j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *(f32x4*)(st + 8192 + (size_t)(lane + 64 * j) * 16) = pd[j];
        *(f32x4*)(st + 12288 + (size_t)(lane + 64 * j) * 16) = pcp[j];
      }
    };
    {
      const int t0 = t_of(T - 1);
#pragma unroll
      for (int rg = 0; rg < R; ++rg)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = lane + 64 * j, c16 = idx & 15, row = idx >> 4;
          const int n = (rg0 + rg) * 16 + row;
          const f32x4 v = n < a.N ? *(const f32x4*)(a.c[d] + ((unsigned)(n * a.P + a.padl + t0) * (unsigned)H + (unsigned)(u0 + c16 * 4)))
                                  : (f32x4){0.f, 0.f, 0.f, 0.f};
          *(f32x4*)(c0 + (rg * 16 * 64) + idx * 4) = v;
        }
      pf_load(0);
      pf_store(0);
      if (Q > 1) pf_load(1);
    }
    for (int q = 0; q < Q; ++q) {
      wg_barrier();
      if (abortf[q & 1]) return;
      if (q + 1 < Q) {
        pf_store((q + 1) & 1);
        if (q + 2 < Q) pf_load(q + 2);
      }
    }
    wg_barrier();
  }
}


// ==================================================================== backward, partial-sum exchange (round 3)
// lstm_cluster2_bwd_kernel above sends every workgroup ALL 4H gate gradients of the step (16 x 4H bf16 = 32 KB per
// chain at H = 256, 4x the forward payload) as dense rows behind drained stores and a flag: 4.2 us per step against
// the forward kernel's 2.9.  The same product can be cut the other way: dh[t-1] = dgates[t] . Wh^T is a sum over the
// gate columns, and a workgroup OWNS 256 of them (4 gates x its 64 units).  So it forms, from its own gate gradients
// alone and straight after the cell update, its partial sum for EVERY unit of the layer - P[16, H] = dg_own[16, 256] .
// Wh[:, own columns]^T, the same 128 MFMAs per step - and sends each peer only the 16 x 64 block of that peer's units:
// (CS - 1) x 512 granules of {step tag, 2 x bf16} in and out per step (12 KB at H = 256, less than the forward
// kernel's), the data is its own flag - no drain, no flag, no second round trip.  The receiver adds its own block (fp32,
// LDS) and the peers' (bf16) in a fixed order.  Weights per workgroup: Wh[all H units][own 256 columns], the same
// 128 VGPRs per lane.  Roles as in the forward kernel: 4 compute waves (wave w: the output tiles of units
// [w * 16 HB, (w + 1) * 16 HB)), 2 pollers, prefetcher, saver (the gate gradients for the weight-gradient products).
// Two workgroup barriers per slot: the slot hand-over, and one between the cell update (which writes the gate
// gradients of all four compute waves into the LDS operand image) and the product that reads them.
constexpr int BP_WAVES = XW + 2;         // compute, saver, prefetcher
constexpr int DGI_LD = 256 + 8;          // bf16 per row of the operand image (row stride 4 banks mod 64: conflict-free 16-byte reads)

// Who computes what is chosen so that NOTHING of the step's dependent chain crosses waves except the operand image:
//   * compute wave w owns units [16 w, 16 w + 16) of the workgroup's 64 in the cell update, and in the product the HB
//     output tiles {(destination workgroup wd, units 16 w .. 16 w + 16 of ITS 64)}: the tile for wd = this workgroup is
//     the own block of exactly the units the wave updates next step, in exactly the accumulator layout the cell update
//     uses (lane (r16, g): unit r16, rows 4 g .. 4 g + 3) - it stays in registers;
//   * the peers' blocks are polled by the lane that consumes them (a lane needs only the sums of ITS unit and rows:
//     4 rows x (CS - 1) peers; the forward kernels cannot do this, every lane of theirs needs the whole gathered vector
//     as an MFMA operand): no poller waves, no LDS image of the gathered sums, no hand-over;
//   * everything of the cell update that does not depend on the exchanged sums (operand reads, tanh, the gate
//     derivatives) is done in front of the poll, inside the hop.
// ONE workgroup barrier per slot is left - between the cell update (the four compute waves write the gate gradients
// into the LDS operand image) and the product that reads it; the prefetcher's and the saver's hand-overs ride on it
// (audit: stage[(q+1)&1] is stored between barrier(q-1) and barrier(q), last read in front of barrier(q-1), next read
// behind barrier(q); dgi[q&1] is written in front of barrier(q), read by the product and - one slot behind - the saver
// between barrier(q) and barrier(q+1), rewritten behind barrier(q+1)).
// Slot timings (profiles/r03_cluster_bwd_trace.txt): poller-wave form 3.51 us, polling compute lanes 3.09, row-major
// prefetcher hand-over 2.97, this form see the profile.
template <int HB, int R>
__global__ __launch_bounds__(BP_WAVES * 64) void lstm_cluster2p_bwd_kernel(LstmClusterArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int H = HB * 64, K4 = 4 * H, CS = HB;
  constexpr int GPD = 16 * 32;                                        // granules per (destination, source) block
  bf16_t* dgi = (bf16_t*)smem;                                        // [2][16][DGI_LD] this slot's gate gradients (row, gate*64 + unit)
  // The operand stage keeps the row-major layout of the global arrays (a transposing prefetcher store was measured:
  // 96 scattered LDS writes per lane, slower than the compute lanes' scalar reads of this layout)
  char* ops = (char*)(dgi + 2 * 16 * DGI_LD);                         // [2] stages of {gates bf16 [16][4][64], dh f32 [16][64], cprev f32 [16][64]}
  constexpr int OPS_G = 16 * 4 * 64 * 2, OPS_F = 16 * 64 * 4;         // 8192, 4096
  constexpr int OPS_STAGE = OPS_G + 2 * OPS_F;                        // 16384
  float* c0 = (float*)(ops + 2 * OPS_STAGE);                          // [R][16][64] cell state at the first processed step
  int* abortf = (int*)(c0 + R * 16 * 64);                             // [1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsets = ((a.N + 15) / 16 + R - 1) / R;
  const int set = blockIdx.x / CS, wgc = blockIdx.x % CS;
  const int d = set / nsets, rg0 = (set % nsets) * R;
  const int r16 = lane & 15, g = lane >> 4;
  // exchange: [chain][parity][destination][source][16 rows][32 unit pairs]
  u64* xb0 = a.xbuf + (size_t)(d * nsets * R + rg0) * 2 * CS * CS * GPD;
  const int u0 = wgc * 64;
  const int T = a.T, Q = a.T * R;
  if (tid == 0) abortf[0] = 0;           // (audit) by wave 0 in front of its first wg_barrier, read behind it
  auto t_of = [&](int step) { return d ? T - 1 - step : step; };

  if (wave < XW) {
    // ================================================================ compute role
    const int wu = wave * 16 + r16;                    // unit inside the workgroup's 64
    // Wh[unit wu of workgroup j][own 256 gate columns], the columns in the operand image's order k = unit * 4 + gate
    bf16x8 bw[HB][8];
#pragma unroll
    for (int j = 0; j < HB; ++j) {
      const bf16_t* row = a.wh[d] + (long)(j * 64 + wu) * K4 + u0;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) bw[j][ks][e] = row[(e & 3) * H + ks * 8 + g * 2 + (e >> 2)];
    }
    float dcc[R][4], pc[R][4];
    f32x4 ownp[R];                                     // own block of the partial sums of the step before
    int len[R][4];
#pragma unroll
    for (int rg = 0; rg < R; ++rg) {
      ownp[rg] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (rg0 + rg) * 16 + g * 4 + r;
        dcc[rg][r] = 0.f; pc[rg][r] = 0.f;
        len[rg][r] = (a.lengths && n < a.N) ? a.lengths[n] : T;
      }
    }
    // the part of a slot's cell update that does not need the exchanged sums; computed one slot ahead, under the
    // product's MFMAs (the stage of slot q + 1 is complete behind barrier(q))
    float kdo[4], kdc[4], ki[4], kj[4], kf[4], dhx[4], cpv[4], gfv[4];
    auto indep = [&](int q) {
      const int bs = q / R, rg = q % R;
      const char* st = ops + (size_t)(q & 1) * OPS_STAGE;
      const bf16_t* sgt = (const bf16_t*)st;
      const float* sdh = (const float*)(st + OPS_G);
      const float* scp = (const float*)(st + OPS_G + OPS_F);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = g * 4 + r;
        const float gi = (float)sgt[(row * 4 + 0) * 64 + wu], gj = (float)sgt[(row * 4 + 1) * 64 + wu];
        const float gf = (float)sgt[(row * 4 + 2) * 64 + wu], go = (float)sgt[(row * 4 + 3) * 64 + wu];
        const float cprev = scp[row * 64 + wu];
        float ccur = c0[(rg * 16 + row) * 64 + wu];
        if (bs > 0) ccur = R == 1 ? pc[0][r] : (rg ? pc[R - 1][r] : pc[0][r]);
        const float tc = tanhf_(ccur);
        dhx[r] = sdh[row * 64 + wu];
        kdo[r] = tc * go * (1.f - go);             // d_o = dh * kdo
        kdc[r] = go * (1.f - tc * tc);             // dc = dh * kdc + dcc
        ki[r] = gj * gi * (1.f - gi);
        kj[r] = gi * (1.f - gj * gj);
        kf[r] = cprev * gf * (1.f - gf);
        cpv[r] = cprev; gfv[r] = gf;
      }
    };
    wg_barrier();                                      // stage 0, c0 and abortf are in place
    indep(0);
    for (int bs = 0; bs < T; ++bs) {                 // backward step index; forward step = T-1-bs
      const int t = t_of(T - 1 - bs);
#pragma unroll
      for (int rg = 0; rg < R; ++rg) {
        const int q = bs * R + rg, buf = q & 1;
        const int n0 = (rg0 + rg) * 16;
        const bool tr = (a.dbg & 16) && blockIdx.x == 0 && tid == 0 && q < 512;
        if (tr) a.trace[q * 8 + 0] = wall_clock64();
        // ---- dh of the step after, summed in a fixed order: own block, then the peers' in workgroup order.
        // Granule (source ws, row pair 2 g + h, unit wu) of this workgroup's block: {tag bs, rows 2h | 2h + 1 of the lane}
        f32x4 rec = ownp[rg];
        if (bs > 0 && CS > 1) {
          const u64* cur = xb0 + ((size_t)rg * 2 + (bs & 1)) * CS * CS * GPD + (size_t)wgc * CS * GPD + (g * 2) * 64 + wu;
          u64 v[CS > 1 ? CS - 1 : 1][2];
          unsigned spins = 0, clk0 = 0;
          bool ok;
          do {
            ok = true;
#pragma unroll
            for (int sx = 0; sx < CS - 1; ++sx) {
              const int ws = sx < wgc ? sx : sx + 1;
#pragma unroll
              for (int h = 0; h < 2; ++h) v[sx][h] = __hip_atomic_load(cur + (size_t)ws * GPD + h * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int sx = 0; sx < CS - 1; ++sx)
#pragma unroll
              for (int h = 0; h < 2; ++h) ok = ok && ((unsigned)(v[sx][h] >> 32) == (unsigned)bs);
            if (!ok && (++spins & 1023u) == 0) {
              if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { abortf[0] = 1; ok = true; }
              else if (ns_spin_timed_out(clk0)) { atomicExch(a.status, 2); abortf[0] = 1; ok = true; }
            }
          } while (!ok);
          if (tr) { a.trace[q * 8 + 4] = wall_clock64(); a.trace[q * 8 + 6] = spins; }
#pragma unroll
          for (int sx = 0; sx < CS - 1; ++sx)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const unsigned pay = (unsigned)v[sx][h];
              rec[2 * h] += __uint_as_float(pay << 16);
              rec[2 * h + 1] += __uint_as_float(pay & 0xffff0000u);
            }
        }
        bf16_t* di = dgi + (size_t)buf * 16 * DGI_LD;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = g * 4 + r;
          const int n = n0 + row;
          const float dh = dhx[r] + rec[r];
          const float d_o = dh * kdo[r];
          const float dc = dh * kdc[r] + dcc[rg][r];
          float dgv[4] = {dc * ki[r], dc * kj[r], dc * kf[r], d_o};
          dcc[rg][r] = dc * gfv[r];
          if (t >= len[rg][r] || n >= a.N) {
            dgv[0] = dgv[1] = dgv[2] = dgv[3] = 0.f;
            dcc[rg][r] = 0.f;
          }
          pc[rg][r] = cpv[r];
          uint2 pk;
          pk.x = pack_bf16(dgv[0], dgv[1]);
          pk.y = pack_bf16(dgv[2], dgv[3]);
          *(uint2*)(di + row * DGI_LD + wu * 4) = pk;       // k = unit * 4 + gate
        }
        if (tr) a.trace[q * 8 + 1] = wall_clock64();
        wg_barrier();            // the operand image is complete (all four compute waves)
        if (abortf[0]) return;
        if (tr) a.trace[q * 8 + 2] = wall_clock64();
        // ---- partial sums of dh from the own 256 gate columns; wave w: units 16 w .. 16 w + 16 of every workgroup
        if (bs + 1 < T) {
          f32x4 acc[HB];
#pragma unroll
          for (int j = 0; j < HB; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 af = *(const bf16x8*)(di + r16 * DGI_LD + ks * 32 + g * 8);
#pragma unroll
            for (int j = 0; j < HB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bw[j][ks], acc[j], 0, 0, 0);
          }
          indep(q + 1);
          if (tr) a.trace[q * 8 + 3] = wall_clock64();
          // D: column r16 = unit wu of workgroup j, rows g*4 + r.  A peer's tile -> its granules; the own tile stays here
          u64* nxt = xb0 + ((size_t)rg * 2 + ((bs + 1) & 1)) * CS * CS * GPD;
#pragma unroll
          for (int j = 0; j < HB; ++j) {
            if (j == wgc) {
              ownp[rg] = acc[j];
            } else {
              u64* dst = nxt + (size_t)(j * CS + wgc) * GPD + (g * 2) * 64 + wu;
#pragma unroll
              for (int h = 0; h < 2; ++h)
                __hip_atomic_store(dst + h * 64, ((u64)(unsigned)(bs + 1) << 32) | pack_bf16(acc[j][2 * h], acc[j][2 * h + 1]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          if (tr) a.trace[q * 8 + 7] = wall_clock64();
        } else if (q + 1 < Q) {
          indep(q + 1);
        }
      }
    }
  } else if (wave == XW) {
    // ================================================================ saver role (stores only), one slot behind:
    // the slot's gate gradients out of the operand image [16 rows][unit * 4 + gate] into dgates[row][gate * H + unit]:
    // a lane takes two (row, 8 units) blocks, reads their 8 x {4 gates} and writes one 16-byte chunk per gate
    auto save = [&](int q) {
      const int bs = q / R, rg = q % R;
      const int t = t_of(T - 1 - bs);
      const int n0 = (rg0 + rg) * 16;
      const bf16_t* di = dgi + (size_t)(q & 1) * 16 * DGI_LD;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int idx = lane + 64 * jj, c8 = idx & 7, row = idx >> 3;
        uint2 x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = *(const uint2*)(di + row * DGI_LD + (c8 * 8 + e) * 4);
        if (n0 + row < a.N) {
          bf16_t* out = a.dgates[d] + ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)K4 + (unsigned)(u0 + c8 * 8));
#pragma unroll
          for (int gate = 0; gate < 4; ++gate) {
            unsigned w[4];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
              const unsigned lo = gate < 2 ? x[2 * e2].x : x[2 * e2].y, hi = gate < 2 ? x[2 * e2 + 1].x : x[2 * e2 + 1].y;
              w[e2] = (gate & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
            }
            *(uint4*)(out + gate * H) = make_uint4(w[0], w[1], w[2], w[3]);
          }
        }
      }
    };
    wg_barrier();
    for (int q = 0; q < Q; ++q) {
      if (q > 0) save(q - 1);
      wg_barrier();
      if (abortf[0]) return;
    }
    save(Q - 1);
  } else {
    // ================================================================ prefetcher role (loads only)
    f32x4 pg[8], pd[4], pcp[4];
    auto pf_load = [&](int q) {
      const int bs = q / R, rg = q % R, step = T - 1 - bs;
      const int t = t_of(step), tp = d ? t + 1 : t - 1;
      const bool has_prev = step > 0;
      const int n0 = (rg0 + rg) * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = lane + 64 * j, c8 = idx & 7, gate = (idx >> 3) & 3, row = idx >> 5;
        const int n = n0 + row;
        pg[j] = n < a.N ? *(const f32x4*)(a.gates[d] + ((unsigned)(n * a.P + a.padl + t) * (unsigned)(4 * H) + (unsigned)(gate * H + u0 + c8 * 8)))
                        : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = lane + 64 * j, c16 = idx & 15, row = idx >> 4;
        const int n = n0 + row;
        pd[j] = n < a.N ? *(const f32x4*)(a.dh[d] + ((unsigned)(n * a.P + a.padl + t) * (unsigned)a.ld_dh + (unsigned)(u0 + c16 * 4)))
                        : (f32x4){0.f, 0.f, 0.f, 0.f};
        pcp[j] = (n < a.N && has_prev) ? *(const f32x4*)(a.c[d] + ((unsigned)(n * a.P + a.padl + tp) * (unsigned)H + (unsigned)(u0 + c16 * 4)))
                                       : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    };
    auto pf_store = [&](int buf) {
      char* st = ops + (size_t)buf * OPS_STAGE;
#pragma unroll
      for (int j = 0; j < 8; ++j) *(f32x4*)(st + (size_t)(lane + 64 * j) * 16) = pg[j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *(f32x4*)(st + OPS_G + (size_t)(lane + 64 * j) * 16) = pd[j];
        *(f32x4*)(st + OPS_G + OPS_F + (size_t)(lane + 64 * j) * 16) = pcp[j];
      }
    };
    {
      const int t0 = t_of(T - 1);
#pragma unroll
      for (int rg = 0; rg < R; ++rg)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = lane + 64 * j, c16 = idx & 15, row = idx >> 4;
          const int n = (rg0 + rg) * 16 + row;
          const f32x4 v = n < a.N ? *(const f32x4*)(a.c[d] + ((unsigned)(n * a.P + a.padl + t0) * (unsigned)H + (unsigned)(u0 + c16 * 4)))
                                  : (f32x4){0.f, 0.f, 0.f, 0.f};
          *(f32x4*)(c0 + (rg * 16 * 64) + idx * 4) = v;
        }
      pf_load(0);
      pf_store(0);
      if (Q > 1) pf_load(1);
    }
    wg_barrier();
    for (int q = 0; q < Q; ++q) {
      if (q + 1 < Q) {
        pf_store((q + 1) & 1);
        if (q + 2 < Q) pf_load(q + 2);
      }
      wg_barrier();
      if (abortf[0]) return;
    }
  }
}

// ------------------------------------------------------------------ C ABI
static int cluster_supported(const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1) {
  return p0->dtype == NS_BF16 && p1->dtype == NS_BF16 && p0->H % 64 == 0 && p0->H <= 512 && p0->T >= 2;
}
// the fp32-state forward form: fp32 h, pre-split recurrent weights, three passes, H <= 256
static int cluster3_supported(const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1) {
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  const ns_lstm_seq_params* pp[2] = {p0, p1};
  if (p0->H % 64 != 0 || p0->H > 256 || p0->T < 2 || p0->N < 1) return 0;
  if (2 * ((p0->N + 15) / 16 + 1) * 8 * sizeof(unsigned) > 4096) return 0;
  for (int d = 0; d < 2; ++d) {
    const ns_lstm_seq_params* p = pp[d];
    if (p->dtype != NS_F32 || p->f32_passes != 3 || !p->whT_hi || !p->whT_lo || !p->xg || !p->h || !p->c || !p->gates) return 0;
    if (!(al16(p->xg) && p->ld_xg % 4 == 0 && al16(p->h) && p->ld_h % 4 == 0 && al16(p->c) && al16(p->gates) &&
          al16(p->whT_hi) && al16(p->whT_lo))) return 0;
    if (p->h_bf16 && !(al16(p->h_bf16) && p->ld_h_bf16 % 8 == 0)) return 0;
    const long widest = p->ld_xg > 4L * p->H ? p->ld_xg : 4L * p->H;
    if ((long)p->N * p->P * widest >= (1L << 31)) return 0;
  }
  return 1;
}
extern "C" int ns_lstm_cluster_supported(const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1, int backward) {
  if (!p0 || !p1) return 0;
  if (!(p0->reverse == 0 && p1->reverse == 1 && p0->N == p1->N && p0->T == p1->T && p0->H == p1->H)) return 0;
  // a chain (direction, 16-row group) = H / 64 workgroups (H / 32 in the fp32 form) that must be resident together; the
  // chains are independent of one another
  if ((p0->H + 31) / 32 > ns_device_cus()) return 0;
  if (cluster_supported(p0, p1)) return 1;
  return !backward && cluster3_supported(p0, p1);
}

extern "C" size_t ns_lstm_cluster_work_bytes(const ns_lstm_seq_params* p) {
  if (!p) return 0;
  const size_t chains = 2 * (size_t)((p->N + 15) / 16 + 1);   // one spare row group per direction (pairs of interleaved chains)
  // exchange buffers for the larger (backward) payload + status word + debug trace (the fp32 forward form's granules,
  // one per unit, are half of that)
  return chains * 2 * 16 * (size_t)(4 * p->H / 2) * sizeof(u64) + 256 + 4096 + 512 * 8 * sizeof(long long);
}

constexpr size_t FLAG_BYTES = 4096;   // 128 chains x 8 counters
static void fill(LstmClusterArgs& a, const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1, void* work) {
  const ns_lstm_seq_params* pp[2] = {p0, p1};
  a.N = p0->N; a.T = p0->T; a.H = p0->H; a.P = p0->P; a.padl = p0->padl; a.CS = p0->H / 64;
  a.ld_xg = p0->ld_xg; a.ld_h = p0->ld_h; a.ld_dh = p0->ld_dh;
  a.lengths = p0->lengths; a.forget_bias = p0->forget_bias; a.cell_clip = p0->cell_clip;
  for (int d = 0; d < 2; ++d) {
    a.xg[d] = pp[d]->xg; a.whT[d] = (const bf16_t*)pp[d]->whT; a.wh[d] = (const bf16_t*)pp[d]->wh;
    a.h[d] = (bf16_t*)pp[d]->h; a.c[d] = pp[d]->c; a.gates[d] = (bf16_t*)pp[d]->gates;
    a.dh[d] = pp[d]->dh; a.dgates[d] = (bf16_t*)pp[d]->dgates;
    a.whT_hi[d] = (const bf16_t*)pp[d]->whT_hi; a.whT_lo[d] = (const bf16_t*)pp[d]->whT_lo;
    a.hf[d] = (float*)pp[d]->h; a.hb[d] = (bf16_t*)pp[d]->h_bf16;
  }
  a.ld_hb = p0->ld_h_bf16;
  a.status = (int*)work;
  a.flags = (unsigned*)((char*)work + 256);               // FLAG_BYTES, zeroed together with the status word
  a.xbuf = (u64*)((char*)work + 256 + FLAG_BYTES);
  const char* dbg = getenv("NS_CLUSTER_DBG");
  a.dbg = dbg ? atoi(dbg) : 0;
  const size_t chains = 2 * (size_t)((a.N + 15) / 16 + 1);
  a.trace = (long long*)((char*)work + 256 + FLAG_BYTES + chains * 2 * 16 * (size_t)(4 * a.H / 2) * sizeof(u64));
}

// The role-split kernels cover H <= 256 with 16-byte aligned operand rows; NS_CLUSTER_DBG bit 3 forces
// the single-role kernels (A/B timing).
static bool role_split_ok(const LstmClusterArgs& a, bool bwd) {
  if (a.dbg & 8) return false;
  if (a.H > 256 || a.H % 64) return false;
  if (2 * ((a.N + 15) / 16 + 1) * 8 * sizeof(unsigned) > FLAG_BYTES) return false;
  const long widest = a.ld_xg > 4L * a.H ? a.ld_xg : 4L * a.H;
  if ((long)a.N * a.P * (widest > a.ld_dh ? widest : a.ld_dh) >= (1L << 31)) return false;   // 32-bit element offsets
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  for (int d = 0; d < 2; ++d) {
    if (!bwd && !(al16(a.xg[d]) && a.ld_xg % 4 == 0 && al16(a.h[d]) && a.ld_h % 8 == 0 && al16(a.c[d]) && al16(a.gates[d]))) return false;
    if (bwd && !(al16(a.dh[d]) && a.ld_dh % 4 == 0 && al16(a.c[d]) && al16(a.gates[d]) && al16(a.dgates[d]))) return false;
  }
  return true;
}

// Both directions of a BiLSTM, whole sequence, one launch.  p0 must be the forward-in-time direction
// (reverse = 0) and p1 the reversed one.  `work` (ns_lstm_cluster_work_bytes) holds the exchange
// buffers; its first int is a status word: 0 ok, non-zero = a spin timed out (results invalid).
extern "C" int ns_lstm_cluster_fwd(const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1, void* work,
                                   ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  NS_CHECK_ARG(p0 && p1 && work, "ns_lstm_cluster_fwd: null");
  if (p0->dtype != NS_F32) NS_CHECK_ARG(cluster_supported(p0, p1), "ns_lstm_cluster_fwd: needs bf16, H %% 64 == 0, H <= 512, T >= 2");
  NS_CHECK_ARG(p0->reverse == 0 && p1->reverse == 1 && p0->N == p1->N && p0->T == p1->T && p0->H == p1->H,
               "ns_lstm_cluster_fwd: p0 forward / p1 reversed with equal shapes expected");
  if (p0->dtype == NS_F32) {
    NS_CHECK_ARG(cluster3_supported(p0, p1), "ns_lstm_cluster_fwd: the fp32 form needs H %% 64 == 0, H <= 256, T >= 2, "
                 "f32_passes 3 with whT_hi / whT_lo, 16-byte aligned operands");
    LstmClusterArgs a = {};
    fill(a, p0, p1, work);
    // NS_CLUSTER_DBG bit 256: one row group per workgroup set (no interleaving), for A/B timing
    const int nrg = (a.N + 15) / 16, R = (nrg >= 2 && (a.dbg & 1024)) ? 2 : 1;        // see the bf16 forward kernel's launch: one set per row group
    const size_t xbytes = (2 * (size_t)nrg + 2) * 2 * 16 * (size_t)a.H * sizeof(u64);
    { const int zrc = ns_zero_async(work, ((256 + FLAG_BYTES + xbytes) + 15) & ~(size_t)15, s); if (zrc) return zrc; }
    const size_t lds3 = (size_t)2 * 2 * 16 * a.H * 2 + sizeof(float) * 2 * 16 * XG3_LD + 2 * 9216 + 32;
    const dim3 grid((unsigned)(2 * ((nrg + R - 1) / R) * (a.H / X3_UPW))), block(X3_WAVES * 64);
#define NS_LAUNCH_F3(HB_) \
    do { \
      static bool attr3 = false; \
      if (!attr3) { \
        (void)hipFuncSetAttribute((const void*)lstm_cluster3_fwd_kernel<HB_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        (void)hipFuncSetAttribute((const void*)lstm_cluster3_fwd_kernel<HB_, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        attr3 = true; \
      } \
      if (R == 2) hipLaunchKernelGGL((lstm_cluster3_fwd_kernel<HB_, 2>), grid, block, lds3, s, a); \
      else hipLaunchKernelGGL((lstm_cluster3_fwd_kernel<HB_, 1>), grid, block, lds3, s, a); \
    } while (0)
    switch (a.H / 64) {
      case 1: NS_LAUNCH_F3(1); break;
      case 2: NS_LAUNCH_F3(2); break;
      case 3: NS_LAUNCH_F3(3); break;
      default: NS_LAUNCH_F3(4); break;
    }
#undef NS_LAUNCH_F3
    NS_CHECK_LAUNCH("lstm_cluster3_fwd");
    return NS_OK;
  }
  NS_CHECK_ARG(cluster_supported(p0, p1), "ns_lstm_cluster_fwd: needs bf16, H %% 64 == 0, H <= 512, T >= 2");
  LstmClusterArgs a = {};
  fill(a, p0, p1, work);
  const size_t chains = 2 * (size_t)((a.N + 15) / 16);
  const size_t xbytes = (chains + 2) * 2 * 16 * (size_t)(a.H / 2) * sizeof(u64);
  { const int zrc = ns_zero_async(work, ((256 + FLAG_BYTES + xbytes) + 15) & ~(size_t)15, s); if (zrc) return zrc; }
  if (role_split_ok(a, false)) {
    const size_t lds2 = (size_t)2 * 16 * a.H * 2 + sizeof(float) * 2 * 16 * XG_LD + 2 * (2048 + 4096 + 8192) + 32;
    // two row groups interleaved per workgroup (R = 2) pay when a slot's compute chain is clearly shorter than the hop;
    // round 3 (two forward pollers, shorter hop): expand BiLSTM 2.92 ms with R = 2, 2.75 ms with one set per row group
    // Re-measured at the end of round 3 (transposed product: the compute chain of a slot is 0.7 us against a hop of 1.5):
    // expand BiLSTM forward 2.38 ms with one set per row group, 3.56 ms with two row groups interleaved per workgroup; the
    // backward kernel 2.63 against 4.36, the fp32 forward 1.08 against 1.17 (encoder).  Interleaving only doubles the
    // slots a workgroup walks through: every row group gets its own set of workgroups (2 x 4 x row groups <= 256 CUs up
    // to batch 512); NS_CLUSTER_DBG bits 512 / 1024 / 2048 force R = 2 (bf16 forward / fp32 forward / backward).
    const int nrg = (a.N + 15) / 16, R = (nrg >= 2 && (a.dbg & 512)) ? 2 : 1;
    const dim3 grid((unsigned)(2 * ((nrg + R - 1) / R) * a.CS)), block(FW_WAVES * 64);
#define NS_LAUNCH_F(HB_) \
    if (R == 2) hipLaunchKernelGGL((lstm_cluster2_fwd_kernel<HB_, 2>), grid, block, lds2, s, a); \
    else hipLaunchKernelGGL((lstm_cluster2_fwd_kernel<HB_, 1>), grid, block, lds2, s, a)
    switch (a.H / 64) {
      case 1: NS_LAUNCH_F(1); break;
      case 2: NS_LAUNCH_F(2); break;
      case 3: NS_LAUNCH_F(3); break;
      default: NS_LAUNCH_F(4); break;
    }
#undef NS_LAUNCH_F
    NS_CHECK_LAUNCH("lstm_cluster2_fwd");
    return NS_OK;
  }
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
  // dense-row role-split kernel: exchanges through the dgates array + flags; partial-sum kernel: [chain][2][CS][CS][512]
  // granules; single-role kernel: [chain][2][16][2H]
  const size_t xbytes = role_split_ok(a, true)
                            ? ((a.dbg & 64) ? 0 : (chains + 2) * 2 * (size_t)a.CS * a.CS * 512 * sizeof(u64))
                            : (chains + 2) * 2 * 16 * (size_t)(4 * a.H / 2) * sizeof(u64);
  { const int zrc = ns_zero_async(work, ((256 + FLAG_BYTES + xbytes) + 15) & ~(size_t)15, s); if (zrc) return zrc; }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)lstm_cluster_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lstm_cluster2_bwd_kernel<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lstm_cluster2_bwd_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lstm_cluster2_bwd_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lstm_cluster2_bwd_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lstm_cluster2_bwd_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (role_split_ok(a, true) && !(a.dbg & 64)) {
    // partial-sum exchange (lstm_cluster2p_bwd_kernel); NS_CLUSTER_DBG bit 64: the dense-row exchange below; bit 32: no
    // interleaving of row groups
    // R = 2 (two row groups interleaved per workgroup) pays when the slot's compute chain is shorter than the hop;
    // measured on the expand BiLSTM (T = 1000, H = 256, 2 row groups): R = 1 3.8 ms, R = 2 4.4 ms
    const int nrg = (a.N + 15) / 16, R = (nrg >= 2 && (a.dbg & 2048)) ? 2 : 1;
    const int CS = a.CS;
    const size_t ldsp = (size_t)2 * 16 * DGI_LD * 2 + 2 * 16384 + sizeof(float) * (size_t)R * 16 * 64 + 32;
    const dim3 grid((unsigned)(2 * ((nrg + R - 1) / R) * CS)), block(BP_WAVES * 64);
#define NS_LAUNCH_BP(HB_) \
    do { \
      if (R == 2) hipLaunchKernelGGL((lstm_cluster2p_bwd_kernel<HB_, 2>), grid, block, ldsp, s, a); \
      else hipLaunchKernelGGL((lstm_cluster2p_bwd_kernel<HB_, 1>), grid, block, ldsp, s, a); \
    } while (0)
    switch (a.H / 64) {
      case 1: NS_LAUNCH_BP(1); break;
      case 2: NS_LAUNCH_BP(2); break;
      case 3: NS_LAUNCH_BP(3); break;
      default: NS_LAUNCH_BP(4); break;
    }
#undef NS_LAUNCH_BP
    NS_CHECK_LAUNCH("lstm_cluster2p_bwd");
    return NS_OK;
  }
  if (role_split_ok(a, true)) {
    // two row groups: one set per group (R = 1, 16 workgroups) measured 4.33 ms against 4.55 ms for the interleaved
    // form on the expand BiLSTM (T = 1000, H = 256); the forward kernel is the other way round (3.2 vs 4.1 ms)
    const int nrg = (a.N + 15) / 16, R = (nrg >= 3 && !(a.dbg & 32)) ? 2 : 1;
    const size_t lds2 = (size_t)2 * 16 * 4 * a.H * 2 + 2 * 16384 + 2 * 8192 + (size_t)R * 4096 + 32;
    const dim3 grid((unsigned)(2 * ((nrg + R - 1) / R) * a.CS)), block(BW_WAVES * 64);
#define NS_LAUNCH_B(HB_) \
    if (R == 2) hipLaunchKernelGGL((lstm_cluster2_bwd_kernel<HB_, 2>), grid, block, lds2, s, a); \
    else hipLaunchKernelGGL((lstm_cluster2_bwd_kernel<HB_, 1>), grid, block, lds2, s, a)
    switch (a.H / 64) {
      case 1: NS_LAUNCH_B(1); break;
      case 2: NS_LAUNCH_B(2); break;
      case 3: NS_LAUNCH_B(3); break;
      default: NS_LAUNCH_B(4); break;
    }
#undef NS_LAUNCH_B
    NS_CHECK_LAUNCH("lstm_cluster2_bwd");
    return NS_OK;
  }
  const size_t lds = (size_t)16 * 4 * a.H * 2 + sizeof(float) * CW * 16 * 65;
  hipLaunchKernelGGL(lstm_cluster_bwd_kernel, dim3((unsigned)(chains * a.CS)), dim3(CTHREADS), lds, s, a);
  NS_CHECK_LAUNCH("lstm_cluster_bwd");
  return NS_OK;
}
