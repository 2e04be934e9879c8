// Persistent single-direction LSTM recurrence for WIDE cells (the Tacotron-2 decoder LSTMs,
// tacotron2.py:67-73: 1024 units, batch 32): one launch for the whole sequence.
//
// Per time step the per-launch path re-streams the whole W_h (8-16 MB) through every CU's
// load queue; here every workgroup keeps its slice of W_h in REGISTERS for the whole sequence
// (forward: UW units x 4 gates x H as hi/lo bf16 planes; backward: 16 units x 4H) and only the
// state travels: h[t-1] (forward) or dgates[t+1] (backward), which the kernel has to write to
// its history array anyway.  Exchange (cdna guide G16, R1): a publisher wave stores the
// workgroup's slice of the history row with 16-byte write-through (sc1) stores, drains them and
// raises ONE flag per (chain, workgroup); poller waves wait for the chain's flags and read the
// rows back with 16-byte sc1 loads into the LDS operand image.
//
// A "chain" is a group of RG batch rows (16 forward, 8 backward - the backward operand is 4x
// wider and two images of it must fit in LDS).  A workgroup serves R chains round-robin with
// the same resident weights, so one chain's exchange latency is covered by the others' compute.
// Waves have fixed roles so that no wave mixes global loads and stores (one vmcnt on gfx9):
//   0-3 compute (K split 4 ways, partial sums meet in LDS, software barrier on an LDS counter)
//   4   publisher (stores + flag), 5-6 pollers (flag wait + gather), 7 prefetcher (per-slot operands)
// One hardware barrier per slot hands the LDS images over.  Every spin is bounded; a timeout
// sets the status word (work[0]) and every workgroup leaves.
#include "common.h"
#include <stdint.h>
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int WTHREADS = 512;
constexpr unsigned WSPIN = 4000000u;
constexpr size_t WIDE_HDR = 256;            // status word
constexpr size_t WIDE_FLAGS = 16 * 1024;    // [chains][NWG] published-step counters
constexpr size_t WIDE_TRACE = 256 * 16 * 8;  // debug timestamps
#define WTRACE(slot, k) do { if (a.trace && blockIdx.x == 0 && lane == 0 && (slot) < 256) a.trace[(slot) * 16 + (k)] = wall_clock64(); } while (0)

struct LstmWideArgs {
  int N, T, H, P, padl, reverse;
  const float* xg; long ld_xg;
  const bf16_t* w_hi; const bf16_t* w_lo;   // forward: W_h^T [4H, H] planes (lo null for one pass); backward: W_h [H, 4H] in w_hi
  void* h; long ld_h;                       // storage dtype T
  float* c;
  void* gates;                              // T [rows, 4H]
  const float* dh; long ld_dh;
  void* dgates;                             // T [rows, 4H]
  bf16_t* dgates_b;                         // bf16 [rows, 4H] exchange payload (== dgates when T is bf16)
  const int* lengths;
  float forget_bias;
  unsigned* flags;
  int* status;
  long long* trace;   // NS_WIDE_DBG: [slot][16] timestamps of workgroup 0 (100 MHz), else null
};

__device__ __forceinline__ void wbarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// element offset in a swizzled [rows][W] bf16 image (W % 128 == 0): 16-byte chunks XORed with the row
__device__ __forceinline__ int wswz(int row, int k, int W) { return row * W + ((((k >> 3) ^ (row & 15)) << 3) | (k & 7)); }

// software barrier among the compute waves: lane 0 of each arrives (release), everyone waits for `target`
__device__ __forceinline__ bool soft_barrier(int* ctr, int target, int lane) {
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  unsigned spins = 0;
  while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
    if (++spins > WSPIN) return false;
  return true;
}

template <typename T> struct wide_traits;
template <> struct wide_traits<float> { static constexpr int EPC = 4; };    // elements per 16-byte chunk
template <> struct wide_traits<bf16_t> { static constexpr int EPC = 8; };

// ==================================================================================== forward
// T = storage type of h / gates; PASSES = 1 (bf16 operands) or 3 (hi/lo split of fp32 operands);
// UW = units per workgroup (8 with two weight planes, 16 with one); R = interleaved 16-row chains.
template <typename T, int PASSES, int UW, int R>
__global__ __launch_bounds__(WTHREADS) void lstm_wide_fwd_kernel(LstmWideArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPL = PASSES == 3 ? 2 : 1;
  constexpr int NT = UW / 4, NCOL = UW * 4;
  constexpr int EPC = wide_traits<T>::EPC;
  constexpr int CH_H = UW / EPC;                     // 16-byte chunks per row of this workgroup's h slice
  constexpr int CH_C = UW / 4;
  constexpr int OUT_H = 16 * UW * (int)sizeof(T), OUT_C = 16 * UW * 4, OUT_G = 16 * 4 * UW * (int)sizeof(T);
  constexpr int OUT_BYTES = OUT_H + OUT_C + OUT_G;
  const int H = a.H, KSL = H / 4, KS = KSL / 32, NWG = H / UW;
  bf16_t* aimg = (bf16_t*)smem;                                        // [2][NPL][16][H] swizzled
  float* red = (float*)(aimg + (size_t)2 * NPL * 16 * H);              // [4][16][NCOL]
  float* xgs = red + 4 * 16 * NCOL;                                    // [2][16][NCOL]
  char* outs = (char*)(xgs + 2 * 16 * NCOL);                           // [2][OUT_BYTES]
  int* sync = (int*)(outs + 2 * OUT_BYTES);                            // abort[2], ready, red_ready
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int set = blockIdx.x / NWG, wgc = blockIdx.x % NWG;
  const int rg0 = set * R;
  const int u0 = wgc * UW;
  const int r16 = lane & 15, g = lane >> 4;
  const int T_ = a.T, Q = a.T * R;
  unsigned* flags0 = a.flags + (size_t)rg0 * NWG;                      // + rg * NWG + wgc
  if (tid < 4) sync[tid] = 0;
  auto t_of = [&](int step) { return a.reverse ? T_ - 1 - step : step; };

  if (wave < 4) {
    // ================================================================ compute role (LDS only)
    const int k0 = wave * KSL;
    bf16x8 bwh[NT][8], bwl[NPL == 2 ? NT : 1][8];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      const int col = tl * 16 + r16, gate = col / UW, unit = col % UW;
      const long wr = ((long)gate * H + u0 + unit) * H + k0 + g * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        bwh[tl][ks] = ks < KS ? *(const bf16x8*)(a.w_hi + wr + ks * 32) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (NPL == 2) bwl[tl][ks] = ks < KS ? *(const bf16x8*)(a.w_lo + wr + ks * 32) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    // cell ownership after the reduction: thread -> (row, unit) for tid < 16 * UW
    const bool cell = tid < 16 * UW;
    const int prow = tid / UW, punit = tid % UW;
    float cst[R];
    int len[R];
#pragma unroll
    for (int rg = 0; rg < R; ++rg) {
      const int n = (rg0 + rg) * 16 + prow;
      cst[rg] = 0.f;
      len[rg] = (a.lengths && n < a.N) ? a.lengths[n] : T_;
    }
    for (int step = 0; step < T_; ++step) {
      const int t = t_of(step);
#pragma unroll
      for (int rg = 0; rg < R; ++rg) {
        const int q = step * R + rg, buf = q & 1;
        wbarrier();
        if (sync[buf]) return;
        if (wave == 0) WTRACE(q, 0);
        f32x4 acc[NT];
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) acc[tl] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
          const bf16_t* ah_ = aimg + (size_t)(buf * NPL) * 16 * H;
          const bf16_t* al_ = ah_ + 16 * H;
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            if (ks < KS) {
              const int o = wswz(r16, k0 + ks * 32 + g * 8, H);
              const bf16x8 ah = *(const bf16x8*)(ah_ + o);
              bf16x8 al = ah;
              if (NPL == 2) al = *(const bf16x8*)(al_ + o);
#pragma unroll
              for (int tl = 0; tl < NT; ++tl)
                acc[tl] = mfma_split<PASSES>(ah, al, bwh[tl][ks], bwl[NPL == 2 ? tl : 0][ks], acc[tl]);
            }
          }
        }
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(wave * 16 + g * 4 + r) * NCOL + tl * 16 + r16] = acc[tl][r];
        if (!soft_barrier(sync + 3, 4 * (q + 1), lane)) { if (lane == 0) { atomicExch(a.status, 4); sync[0] = sync[1] = 1; } return; }
        if (wave == 0) WTRACE(q, 1);
        if (tid < ((16 * UW + 63) & ~63)) {
          if (cell) {
            float z[4];
#pragma unroll
            for (int gate = 0; gate < 4; ++gate) {
              const int col = gate * UW + punit;
              float v = xgs[(buf * 16 + prow) * NCOL + col];
#pragma unroll
              for (int w = 0; w < 4; ++w) v += red[(w * 16 + prow) * NCOL + col];
              z[gate] = v;
            }
            const bool masked = t >= len[rg];
            const float gi = sigmoidf_(z[0]), gj = tanhf_(z[1]), gf = sigmoidf_(z[2] + a.forget_bias), go = sigmoidf_(z[3]);
            float cn = gf * cst[rg] + gi * gj;
            float hn = go * tanhf_(cn);
            if (masked) { cn = 0.f; hn = 0.f; }
            cst[rg] = cn;
            char* ob = outs + (size_t)buf * OUT_BYTES;
            stf((T*)ob + prow * UW + punit, hn);
            ((float*)(ob + OUT_H))[prow * UW + punit] = cn;
            T* og = (T*)(ob + OUT_H + OUT_C) + prow * 4 * UW + punit;
            stf(og, masked ? 0.f : gi); stf(og + UW, masked ? 0.f : gj); stf(og + 2 * UW, masked ? 0.f : gf); stf(og + 3 * UW, masked ? 0.f : go);
          }
          if (wave == 0) WTRACE(q, 2);
          if (lane == 0) __hip_atomic_fetch_add(sync + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    wbarrier();
  } else if (wave == 4) {
    // ================================================================ publisher role (stores only)
    constexpr int CELLW = (16 * UW + 63) / 64;                         // waves that signal `ready` per slot
    constexpr int NCH = 16 * (CH_H + CH_C + 4 * CH_H);
    const long rows = (long)a.N * a.P;
    const auto rs_h = __builtin_amdgcn_make_buffer_rsrc(a.h, 0, (int)(rows * a.ld_h * sizeof(T)), 0x00020000);
    const auto rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)a.c, 0, (int)(rows * H * 4), 0x00020000);
    const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(a.gates, 0, (int)(rows * 4 * H * sizeof(T)), 0x00020000);
    wbarrier();
    for (int q = 0; q < Q; ++q) {
      const int step = q / R, rg = q % R;
      const int t = t_of(step);
      const int n0 = (rg0 + rg) * 16;
      unsigned spins = 0;
      while (__hip_atomic_load(sync + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < CELLW * (q + 1)) {
        __builtin_amdgcn_s_sleep(1);
        if (((++spins) & 255) == 0 && (sync[0] | sync[1])) return;
        if (spins > WSPIN) { atomicExch(a.status, 3); return; }
      }
      WTRACE(q, 4);
      const char* ob = outs + (size_t)(q & 1) * OUT_BYTES;
#pragma unroll
      for (int it = 0; it < (NCH + 63) / 64; ++it) {
        const int idx = lane + 64 * it;
        if (idx < NCH) {
          const u32x4 v = *(const u32x4*)(ob + idx * 16);
          if (idx < 16 * CH_H) {
            const int row = idx / CH_H, cc = idx % CH_H;
            if (n0 + row < a.N) {
              const unsigned off = ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)a.ld_h + (unsigned)u0) * (unsigned)sizeof(T) + cc * 16;
              __builtin_amdgcn_raw_buffer_store_b128(v, rs_h, off, 0, 16);        // sc1: the peers read it back
            }
          } else if (idx < 16 * (CH_H + CH_C)) {
            const int j = idx - 16 * CH_H, row = j / CH_C, cc = j % CH_C;
            if (n0 + row < a.N) {
              const unsigned off = ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)H + (unsigned)u0) * 4u + cc * 16;
              __builtin_amdgcn_raw_buffer_store_b128(v, rs_c, off, 0, 0);
            }
          } else {
            const int j = idx - 16 * (CH_H + CH_C), row = j / (4 * CH_H), gate = (j / CH_H) % 4, cc = j % CH_H;
            if (n0 + row < a.N) {
              const unsigned off = ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)(4 * H) + (unsigned)(gate * H + u0)) * (unsigned)sizeof(T) + cc * 16;
              __builtin_amdgcn_raw_buffer_store_b128(v, rs_g, off, 0, 0);
            }
          }
        }
      }
      WTRACE(q, 5);
      if (R > 1) wbarrier();
      WTRACE(q, 6);
      if (step + 1 < T_) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // drain before the flag
        if (lane == 0) __hip_atomic_store(flags0 + rg * NWG + wgc, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      WTRACE(q, 7);
      if (R == 1) wbarrier();
    }
  } else if (wave < 7) {
    // ================================================================ poller role (loads only)
    const int pl = (wave - 5) * 64 + lane;
    const long rows = (long)a.N * a.P;
    const auto rs_h = __builtin_amdgcn_make_buffer_rsrc(a.h, 0, (int)(rows * a.ld_h * sizeof(T)), 0x00020000);
    const int cpr = H / EPC;                            // chunks per row
    for (int q = 0; q < Q; ++q) {
      const int step = q / R, rg = q % R, buf = q & 1;
      if (wave == 5) WTRACE(q, 8);
      if (step > 0) {
        const unsigned* fl = flags0 + rg * NWG;
        unsigned spins = 0;
        for (;;) {
          bool ok = true;
          for (int i = lane; i < NWG; i += 64)
            ok = ok && __hip_atomic_load(fl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)step;
          if (__all(ok)) break;
          ++spins;
          if (spins > WSPIN) { atomicExch(a.status, 2); sync[buf] = 1; break; }
          if ((spins & 1023) == 0 && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { sync[buf] = 1; break; }
        }
        if (wave == 5) { WTRACE(q, 9); if (a.trace && blockIdx.x == 0 && lane == 0 && q < 256) a.trace[q * 16 + 12] = spins; }
        const int tp = t_of(step - 1);
        const int n0 = (rg0 + rg) * 16;
        bf16_t* dh_ = aimg + (size_t)(buf * NPL) * 16 * H;
        bf16_t* dl_ = dh_ + 16 * H;
        const int total = 16 * cpr;
        for (int base = 0; base < total; base += 128 * 16) {
          u32x4 v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int idx = base + pl + 128 * j, row = idx / cpr, cc = idx % cpr;
            v[j] = (u32x4){0u, 0u, 0u, 0u};
            if (idx < total && n0 + row < a.N) {
              const unsigned off = ((unsigned)((n0 + row) * a.P + a.padl + tp) * (unsigned)a.ld_h) * (unsigned)sizeof(T) + cc * 16;
              v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_h, off, 0, 16);      // sc1: bypasses this CU's L1
            }
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int idx = base + pl + 128 * j, row = idx / cpr, cc = idx % cpr;
            if (idx < total) {
              if constexpr (sizeof(T) == 4) {
                bf16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float f = __uint_as_float(v[j][e]);
                  const bf16_t hb = (bf16_t)f;
                  hi[e] = hb;
                  lo[e] = (bf16_t)(f - (float)hb);
                }
                const int o = wswz(row, cc * 4, H);
                *(bf16x4*)(dh_ + o) = hi;
                if (NPL == 2) *(bf16x4*)(dl_ + o) = lo;
              } else {
                *(u32x4*)(dh_ + wswz(row, cc * 8, H)) = v[j];
              }
            }
          }
        }
      }
      if (wave == 5) WTRACE(q, 10);
      wbarrier();
      if (sync[buf]) return;
    }
    wbarrier();
  } else {
    // ================================================================ prefetcher role (loads only)
    // per slot: 16 rows x 4 gates x UW floats of xg = 16 * UW chunks; LDS offset = chunk * 16
    constexpr int XPL = UW / 4;
    f32x4 pf[XPL];
    auto pf_load = [&](int q) {
      const int step = q / R, rg = q % R;
      const int t = t_of(step), n0 = (rg0 + rg) * 16;
#pragma unroll
      for (int j = 0; j < XPL; ++j) {
        const int idx = lane + 64 * j, row = idx / (4 * CH_C), gate = (idx / CH_C) % 4, cc = idx % CH_C;
        const int n = n0 + row;
        pf[j] = n < a.N ? *(const f32x4*)(a.xg + ((unsigned)(n * a.P + a.padl + t) * (unsigned)a.ld_xg + (unsigned)(gate * H + u0 + cc * 4)))
                        : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    };
    auto pf_store = [&](int buf) {
#pragma unroll
      for (int j = 0; j < XPL; ++j) *(f32x4*)(xgs + (size_t)buf * 16 * NCOL + (lane + 64 * j) * 4) = pf[j];
    };
    pf_load(0);
    pf_store(0);
    if (Q > 1) pf_load(1);
    for (int q = 0; q < Q; ++q) {
      wbarrier();
      if (sync[q & 1]) return;
      if (q + 1 < Q) {
        pf_store((q + 1) & 1);
        if (q + 2 < Q) pf_load(q + 2);
      }
    }
    wbarrier();
  }
}

// ==================================================================================== backward
// dh[t] = dh_out[t] + dgates[t+1].W_h^T with bf16 operands (one pass); chains of 8 rows.
template <typename T, int R>
__global__ __launch_bounds__(WTHREADS) void lstm_wide_bwd_kernel(LstmWideArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int UW = 16, RG = 8;
  constexpr bool SIDE = sizeof(T) == 4;                 // fp32 storage: bf16 side copy is the payload
  constexpr int EPC = wide_traits<T>::EPC;
  constexpr int CH_T = UW / EPC;                        // chunks per (row, gate) segment in storage type (4 or 2)
  constexpr int OPS_G = RG * 4 * UW * (int)sizeof(T), OPS_F = RG * UW * 4;
  constexpr int OPS_BYTES = OPS_G + 2 * OPS_F;          // gates, dh, cprev
  constexpr int OUT_T = RG * 4 * UW * (int)sizeof(T), OUT_B = SIDE ? RG * 4 * UW * 2 : 0;
  constexpr int OUT_BYTES = OUT_T + OUT_B;
  const int H = a.H, K = 4 * a.H, KSL = H, KS = KSL / 32, NWG = H / UW;
  bf16_t* aimg = (bf16_t*)smem;                                        // [2][RG][K] swizzled
  float* red = (float*)(aimg + (size_t)2 * RG * K);                    // [4][RG][16]
  char* ops = (char*)(red + 4 * RG * 16);                              // [2][OPS_BYTES]
  char* outs = ops + 2 * OPS_BYTES;                                    // [2][OUT_BYTES]
  float* c0 = (float*)(outs + 2 * OUT_BYTES);                          // [R][RG][UW]
  int* sync = (int*)(c0 + R * RG * UW);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int set = blockIdx.x / NWG, wgc = blockIdx.x % NWG;
  const int rg0 = set * R;
  const int u0 = wgc * UW;
  const int r16 = lane & 15, g = lane >> 4;
  const int T_ = a.T, Q = a.T * R;
  unsigned* flags0 = a.flags + (size_t)rg0 * NWG;
  if (tid < 4) sync[tid] = 0;
  auto t_of = [&](int step) { return a.reverse ? T_ - 1 - step : step; };

  if (wave < 4) {
    // ================================================================ compute role (LDS only)
    const int k0 = wave * KSL;
    bf16x8 bw[32];
    {
      const bf16_t* row = a.w_hi + (long)(u0 + r16) * K + k0 + g * 8;
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) bw[ks] = ks < KS ? *(const bf16x8*)(row + ks * 32) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    // A fragment rows 8..15 do not exist: those lanes feed zeros
    const bool arow = r16 < RG;
    int asw[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) asw[m] = r16 * K + k0 + (((m * 4 + g) ^ r16) << 3);   // k0 % 128 == 0 keeps the XOR group
    const bool cell = tid < RG * UW;
    const int prow = tid / UW, punit = tid % UW;
    float dcc[R], pc[R];
    int len[R];
#pragma unroll
    for (int rg = 0; rg < R; ++rg) {
      const int n = (rg0 + rg) * RG + prow;
      dcc[rg] = 0.f; pc[rg] = 0.f;
      len[rg] = (a.lengths && n < a.N) ? a.lengths[n] : T_;
    }
    for (int bs = 0; bs < T_; ++bs) {
      const int t = t_of(T_ - 1 - bs);
#pragma unroll
      for (int rg = 0; rg < R; ++rg) {
        const int q = bs * R + rg, buf = q & 1;
        const int n0 = (rg0 + rg) * RG;
        wbarrier();
        if (sync[buf]) return;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
        if (bs > 0) {
          const bf16_t* db = aimg + (size_t)buf * RG * K;
#pragma unroll
          for (int ks = 0; ks < 32; ks += 2) {
            if (ks < KS) {
              bf16x8 a0 = {0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0;
              if (arow) {
                a0 = *(const bf16x8*)(db + asw[ks & 3] + (ks >> 2) * 128);
                a1 = *(const bf16x8*)(db + asw[(ks + 1) & 3] + ((ks + 1) >> 2) * 128);
              }
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bw[ks], acc, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bw[ks + 1], acc2, 0, 0, 0);
            }
          }
        }
        if (g < 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(wave * RG + g * 4 + r) * 16 + r16] = acc[r] + acc2[r];
        }
        if (!soft_barrier(sync + 3, 4 * (q + 1), lane)) { if (lane == 0) { atomicExch(a.status, 4); sync[0] = sync[1] = 1; } return; }
        if (tid < RG * UW) {
          const char* st = ops + (size_t)buf * OPS_BYTES;
          const T* sgt = (const T*)st;
          const float* sdh = (const float*)(st + OPS_G);
          const float* scp = (const float*)(st + OPS_G + OPS_F);
          const int n = n0 + prow;
          const float gi = ldf(sgt + (prow * 4 + 0) * UW + punit), gj = ldf(sgt + (prow * 4 + 1) * UW + punit);
          const float gf = ldf(sgt + (prow * 4 + 2) * UW + punit), go = ldf(sgt + (prow * 4 + 3) * UW + punit);
          const float cprev = scp[prow * UW + punit];
          const float ccur = bs == 0 ? c0[(rg * RG + prow) * UW + punit] : pc[rg];
          float dh = sdh[prow * UW + punit];
#pragma unroll
          for (int w = 0; w < 4; ++w) dh += red[(w * RG + prow) * 16 + punit];
          const float tc = tanhf_(ccur);
          const float d_o = dh * tc * go * (1.f - go);
          const float dc = dh * go * (1.f - tc * tc) + dcc[rg];
          float dgv[4] = {dc * gj * gi * (1.f - gi), dc * gi * (1.f - gj * gj), dc * cprev * gf * (1.f - gf), d_o};
          dcc[rg] = dc * gf;
          if (t >= len[rg] || n >= a.N) {
            dgv[0] = dgv[1] = dgv[2] = dgv[3] = 0.f;
            dcc[rg] = 0.f;
          }
          pc[rg] = cprev;
          char* ob = outs + (size_t)buf * OUT_BYTES;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            stf((T*)ob + (prow * 4 + j) * UW + punit, dgv[j]);
            if (SIDE) ((bf16_t*)(ob + OUT_T))[(prow * 4 + j) * UW + punit] = (bf16_t)dgv[j];
          }
          if (lane == 0) __hip_atomic_fetch_add(sync + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    wbarrier();
  } else if (wave == 4) {
    // ================================================================ publisher role (stores only)
    constexpr int CELLW = RG * UW / 64;
    constexpr int NCH_T = RG * 4 * CH_T, NCH_B = SIDE ? RG * 4 * 2 : 0, NCH = NCH_T + NCH_B;
    const long rows = (long)a.N * a.P;
    const auto rs_t = __builtin_amdgcn_make_buffer_rsrc(a.dgates, 0, (int)(rows * K * sizeof(T)), 0x00020000);
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.dgates_b, 0, (int)(rows * K * 2), 0x00020000);
    wbarrier();
    for (int q = 0; q < Q; ++q) {
      const int bs = q / R, rg = q % R;
      const int t = t_of(T_ - 1 - bs);
      const int n0 = (rg0 + rg) * RG;
      unsigned spins = 0;
      while (__hip_atomic_load(sync + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < CELLW * (q + 1)) {
        __builtin_amdgcn_s_sleep(1);
        if (((++spins) & 255) == 0 && (sync[0] | sync[1])) return;
        if (spins > WSPIN) { atomicExch(a.status, 3); return; }
      }
      const char* ob = outs + (size_t)(q & 1) * OUT_BYTES;
#pragma unroll
      for (int it = 0; it < (NCH + 63) / 64; ++it) {
        const int idx = lane + 64 * it;
        if (idx < NCH) {
          const u32x4 v = *(const u32x4*)(ob + idx * 16);
          if (idx < NCH_T) {
            const int row = idx / (4 * CH_T), gate = (idx / CH_T) % 4, cc = idx % CH_T;
            if (n0 + row < a.N) {
              const unsigned off = ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)K + (unsigned)(gate * H + u0)) * (unsigned)sizeof(T) + cc * 16;
              __builtin_amdgcn_raw_buffer_store_b128(v, rs_t, off, 0, SIDE ? 0 : 16);
            }
          } else {
            const int j = idx - NCH_T, row = j / 8, gate = (j / 2) % 4, cc = j % 2;
            if (n0 + row < a.N) {
              const unsigned off = ((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)K + (unsigned)(gate * H + u0)) * 2u + cc * 16;
              __builtin_amdgcn_raw_buffer_store_b128(v, rs_b, off, 0, 16);
            }
          }
        }
      }
      if (R > 1) wbarrier();
      if (bs + 1 < T_) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(flags0 + rg * NWG + wgc, (unsigned)(bs + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (R == 1) wbarrier();
    }
  } else if (wave < 7) {
    // ================================================================ poller role (loads only)
    const int pl = (wave - 5) * 64 + lane;
    const long rows = (long)a.N * a.P;
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.dgates_b, 0, (int)(rows * K * 2), 0x00020000);
    const int cpr = K / 8;
    for (int q = 0; q < Q; ++q) {
      const int bs = q / R, rg = q % R, buf = q & 1;
      if (bs > 0) {
        const unsigned* fl = flags0 + rg * NWG;
        unsigned spins = 0;
        for (;;) {
          bool ok = true;
          for (int i = lane; i < NWG; i += 64)
            ok = ok && __hip_atomic_load(fl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)bs;
          if (__all(ok)) break;
          ++spins;
          if (spins > WSPIN) { atomicExch(a.status, 2); sync[buf] = 1; break; }
          if ((spins & 1023) == 0 && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { sync[buf] = 1; break; }
        }
        const int tn = t_of(T_ - bs);                    // time index of the step processed just before
        const int n0 = (rg0 + rg) * RG;
        bf16_t* dst = aimg + (size_t)buf * RG * K;
        const int total = RG * cpr;
        for (int base = 0; base < total; base += 128 * 16) {
          u32x4 v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int idx = base + pl + 128 * j, row = idx / cpr, cc = idx % cpr;
            v[j] = (u32x4){0u, 0u, 0u, 0u};
            if (idx < total && n0 + row < a.N) {
              const unsigned off = ((unsigned)((n0 + row) * a.P + a.padl + tn) * (unsigned)K) * 2u + cc * 16;
              v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, off, 0, 16);
            }
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int idx = base + pl + 128 * j, row = idx / cpr, cc = idx % cpr;
            if (idx < total) *(u32x4*)(dst + wswz(row, cc * 8, K)) = v[j];
          }
        }
      }
      wbarrier();
      if (sync[buf]) return;
    }
    wbarrier();
  } else {
    // ================================================================ prefetcher role (loads only)
    constexpr int NG = RG * 4 * CH_T, NF = RG * 4;       // chunks: gates, dh (and cprev)
    constexpr int NOP = NG + 2 * NF;
    constexpr int OPL = (NOP + 63) / 64;
    u32x4 pf[OPL];
    auto pf_load = [&](int q) {
      const int bs = q / R, rg = q % R, step = T_ - 1 - bs;
      const int t = t_of(step), tp = a.reverse ? t + 1 : t - 1;
      const bool has_prev = step > 0;
      const int n0 = (rg0 + rg) * RG;
#pragma unroll
      for (int it = 0; it < OPL; ++it) {
        const int idx = lane + 64 * it;
        pf[it] = (u32x4){0u, 0u, 0u, 0u};
        if (idx < NG) {
          const int row = idx / (4 * CH_T), gate = (idx / CH_T) % 4, cc = idx % CH_T;
          if (n0 + row < a.N)
            pf[it] = *(const u32x4*)((const char*)a.gates + (((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)K + (unsigned)(gate * H + u0)) * (unsigned)sizeof(T) + cc * 16));
        } else if (idx < NG + NF) {
          const int j = idx - NG, row = j / 4, cc = j % 4;
          if (n0 + row < a.N)
            pf[it] = *(const u32x4*)((const char*)a.dh + (((unsigned)((n0 + row) * a.P + a.padl + t) * (unsigned)a.ld_dh + (unsigned)u0) * 4u + cc * 16));
        } else if (idx < NOP) {
          const int j = idx - NG - NF, row = j / 4, cc = j % 4;
          if (n0 + row < a.N && has_prev)
            pf[it] = *(const u32x4*)((const char*)a.c + (((unsigned)((n0 + row) * a.P + a.padl + tp) * (unsigned)H + (unsigned)u0) * 4u + cc * 16));
        }
      }
    };
    auto pf_store = [&](int buf) {
#pragma unroll
      for (int it = 0; it < OPL; ++it) {
        const int idx = lane + 64 * it;
        if (idx < NOP) *(u32x4*)(ops + (size_t)buf * OPS_BYTES + idx * 16) = pf[it];
      }
    };
    {
      const int t0 = t_of(T_ - 1);
      for (int idx = lane; idx < R * RG * 4; idx += 64) {
        const int rg = idx / (RG * 4), row = (idx / 4) % RG, cc = idx % 4;
        const int n = (rg0 + rg) * RG + row;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (n < a.N) v = *(const u32x4*)((const char*)a.c + (((unsigned)(n * a.P + a.padl + t0) * (unsigned)H + (unsigned)u0) * 4u + cc * 16));
        *(u32x4*)((char*)c0 + idx * 16) = v;
      }
      pf_load(0);
      pf_store(0);
      if (Q > 1) pf_load(1);
    }
    for (int q = 0; q < Q; ++q) {
      wbarrier();
      if (sync[q & 1]) return;
      if (q + 1 < Q) {
        pf_store((q + 1) & 1);
        if (q + 2 < Q) pf_load(q + 2);
      }
    }
    wbarrier();
  }
}

// ==================================================================================== C ABI
static int device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 1;
  }
  return cus;
}

struct WidePlan { int ok, passes, uw, R, nsets, grid; size_t lds; };

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static WidePlan plan_fwd(const ns_lstm_seq_params* p) {
  WidePlan w = {};
  if (!p || p->T < 2 || p->H % 128 || p->H > 1024 || p->H < 128) return w;
  const bool f32 = p->dtype == NS_F32;
  if (f32) {
    if (!(p->whT_hi && (p->f32_passes == 1 || (p->f32_passes == 3 && p->whT_lo)))) return w;
    w.passes = p->f32_passes;
  } else if (p->dtype == NS_BF16) {
    if (!p->whT) return w;
    w.passes = 1;
  } else return w;
  w.uw = w.passes == 3 ? 8 : 16;
  const int esz = f32 ? 4 : 2;
  if (!(al16(p->xg) && al16(p->h) && al16(p->c) && al16(p->gates)) || p->ld_xg % 4 || (p->ld_h * esz) % 16) return w;
  if ((long)p->N * p->P * (4L * p->H > p->ld_xg ? 4L * p->H : p->ld_xg) * 4 >= (1L << 31)) return w;
  const int nrg = (p->N + 15) / 16;
  w.R = nrg >= 2 ? 2 : 1;
  w.nsets = (nrg + w.R - 1) / w.R;
  const int nwg = p->H / w.uw;
  w.grid = w.nsets * nwg;
  if (w.grid > device_cus()) return w;                                   // every workgroup must be resident
  if ((size_t)w.nsets * w.R * nwg * sizeof(unsigned) > WIDE_FLAGS) return w;
  const int npl = w.passes == 3 ? 2 : 1, ncol = w.uw * 4;
  const size_t outb = 16 * w.uw * esz + 16 * w.uw * 4 + 16 * 4 * w.uw * esz;
  w.lds = (size_t)2 * npl * 16 * p->H * 2 + sizeof(float) * (4 * 16 * ncol + 2 * 16 * ncol) + 2 * outb + 32;
  if (w.lds > 160 * 1024) return w;
  w.ok = 1;
  return w;
}

static WidePlan plan_bwd(const ns_lstm_seq_params* p) {
  WidePlan w = {};
  if (!p || p->T < 2 || p->H % 128 || p->H > 1024 || p->H < 128) return w;
  const bool f32 = p->dtype == NS_F32;
  if (f32) {
    if (!(p->wh_bf16 && p->dgates_bf16 && p->f32_passes == 1)) return w;
  } else if (p->dtype == NS_BF16) {
    if (!p->wh) return w;
  } else return w;
  const int esz = f32 ? 4 : 2;
  if (!(al16(p->dh) && al16(p->c) && al16(p->gates) && al16(p->dgates)) || p->ld_dh % 4) return w;
  if ((long)p->N * p->P * 4L * p->H * 4 >= (1L << 31)) return w;
  w.passes = 1; w.uw = 16;
  const int nrg = (p->N + 7) / 8;
  w.R = nrg >= 4 ? 4 : (nrg >= 2 ? 2 : 1);
  w.nsets = (nrg + w.R - 1) / w.R;
  const int nwg = p->H / 16;
  w.grid = w.nsets * nwg;
  if (w.grid > device_cus()) return w;
  if ((size_t)w.nsets * w.R * nwg * sizeof(unsigned) > WIDE_FLAGS) return w;
  const size_t opsb = 8 * 4 * 16 * esz + 2 * 8 * 16 * 4, outb = 8 * 4 * 16 * esz + (f32 ? 8 * 4 * 16 * 2 : 0);
  w.lds = (size_t)2 * 8 * 4 * p->H * 2 + sizeof(float) * 4 * 8 * 16 + 2 * opsb + 2 * outb + sizeof(float) * w.R * 8 * 16 + 32;
  if (w.lds > 160 * 1024) return w;
  w.ok = 1;
  return w;
}

extern "C" int ns_lstm_wide_supported(const ns_lstm_seq_params* p, int backward) {
  return backward ? plan_bwd(p).ok : plan_fwd(p).ok;
}

extern "C" size_t ns_lstm_wide_work_bytes(const ns_lstm_seq_params*) { return WIDE_HDR + WIDE_FLAGS + WIDE_TRACE; }

static void fill_wide(LstmWideArgs& a, const ns_lstm_seq_params* p, void* work) {
  a.N = p->N; a.T = p->T; a.H = p->H; a.P = p->P; a.padl = p->padl; a.reverse = p->reverse;
  a.xg = p->xg; a.ld_xg = p->ld_xg;
  a.h = p->h; a.ld_h = p->ld_h; a.c = p->c; a.gates = p->gates;
  a.dh = p->dh; a.ld_dh = p->ld_dh; a.dgates = p->dgates;
  a.lengths = p->lengths; a.forget_bias = p->forget_bias;
  a.status = (int*)work;
  a.flags = (unsigned*)((char*)work + WIDE_HDR);
  a.trace = getenv("NS_WIDE_DBG") ? (long long*)((char*)work + WIDE_HDR + WIDE_FLAGS) : nullptr;
}

template <typename K>
static void set_lds(K kernel, size_t lds) {
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)lds;
}

// Whole-sequence forward recurrence in one launch.  `work` (ns_lstm_wide_work_bytes) holds the status word
// (work[0]: non-zero after completion = an exchange timed out, outputs invalid) and the flags.
extern "C" int ns_lstm_wide_fwd(const ns_lstm_seq_params* p, void* work, ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  NS_CHECK_ARG(p && work, "ns_lstm_wide_fwd: null");
  const WidePlan w = plan_fwd(p);
  NS_CHECK_ARG(w.ok, "ns_lstm_wide_fwd: unsupported shape / operands (see ns_lstm_wide_supported)");
  LstmWideArgs a = {};
  fill_wide(a, p, work);
  const bool f32 = p->dtype == NS_F32;
  a.w_hi = (const bf16_t*)(f32 ? p->whT_hi : p->whT);
  a.w_lo = (const bf16_t*)(f32 ? p->whT_lo : nullptr);
  if (hipMemsetAsync(work, 0, WIDE_HDR + WIDE_FLAGS, s) != hipSuccess) { ns_set_error("ns_lstm_wide_fwd: memset failed"); return NS_ERR_LAUNCH; }
  const dim3 grid(w.grid), block(WTHREADS);
#define NS_WF(T_, PASSES_, UW_) do { \
    if (w.R == 2) { set_lds(lstm_wide_fwd_kernel<T_, PASSES_, UW_, 2>, w.lds); hipLaunchKernelGGL((lstm_wide_fwd_kernel<T_, PASSES_, UW_, 2>), grid, block, w.lds, s, a); } \
    else { set_lds(lstm_wide_fwd_kernel<T_, PASSES_, UW_, 1>, w.lds); hipLaunchKernelGGL((lstm_wide_fwd_kernel<T_, PASSES_, UW_, 1>), grid, block, w.lds, s, a); } } while (0)
  if (f32 && w.passes == 3) NS_WF(float, 3, 8);
  else if (f32) NS_WF(float, 1, 16);
  else NS_WF(bf16_t, 1, 16);
#undef NS_WF
  NS_CHECK_LAUNCH("lstm_wide_fwd");
  return NS_OK;
}

extern "C" int ns_lstm_wide_bwd(const ns_lstm_seq_params* p, void* work, ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  NS_CHECK_ARG(p && work, "ns_lstm_wide_bwd: null");
  const WidePlan w = plan_bwd(p);
  NS_CHECK_ARG(w.ok, "ns_lstm_wide_bwd: unsupported shape / operands (see ns_lstm_wide_supported)");
  LstmWideArgs a = {};
  fill_wide(a, p, work);
  const bool f32 = p->dtype == NS_F32;
  a.w_hi = (const bf16_t*)(f32 ? p->wh_bf16 : p->wh);
  a.dgates_b = (bf16_t*)(f32 ? p->dgates_bf16 : p->dgates);
  if (hipMemsetAsync(work, 0, WIDE_HDR + WIDE_FLAGS, s) != hipSuccess) { ns_set_error("ns_lstm_wide_bwd: memset failed"); return NS_ERR_LAUNCH; }
  const dim3 grid(w.grid), block(WTHREADS);
#define NS_WB(T_) do { \
    if (w.R == 4) { set_lds(lstm_wide_bwd_kernel<T_, 4>, w.lds); hipLaunchKernelGGL((lstm_wide_bwd_kernel<T_, 4>), grid, block, w.lds, s, a); } \
    else if (w.R == 2) { set_lds(lstm_wide_bwd_kernel<T_, 2>, w.lds); hipLaunchKernelGGL((lstm_wide_bwd_kernel<T_, 2>), grid, block, w.lds, s, a); } \
    else { set_lds(lstm_wide_bwd_kernel<T_, 1>, w.lds); hipLaunchKernelGGL((lstm_wide_bwd_kernel<T_, 1>), grid, block, w.lds, s, a); } } while (0)
  if (f32) NS_WB(float); else NS_WB(bf16_t);
#undef NS_WB
  NS_CHECK_LAUNCH("lstm_wide_bwd");
  return NS_OK;
}
