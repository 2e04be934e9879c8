// Persistent LSTM recurrence for WIDE cells at small batch (the two decoder LSTMs of tacotron2.py:67-73: 1024 units,
// 32 rows, 200 steps): ONE launch for the whole sequence instead of one per time step.
//
// The launch-per-step kernels re-stream the whole recurrent matrix (16.8 MB as hi + lo bf16 planes, forward) through
// every CU each step and pay a dependent-launch boundary on top: ~9.9 us per step.  Here a workgroup keeps its slice of
// W_h in REGISTERS for all steps (forward: 8 units x 4 gates x H as hi / lo MFMA B fragments, 64 VGPRs per lane), the
// cell state too, and only the state travels - through the history array the kernel has to write anyway:
//   workgroup (rg, ub) = 16 batch rows x 8 units (forward) / 8 or 16 rows x 16 units (backward)
//   before the launch: a fill kernel writes an all-ones NaN pattern (the SENTINEL, a value the recurrence never
//              stores) over every (row, step) of the exchanged array;
//   per step:  8 SWEEPER waves split K.  Each polls ONE 16-byte piece per producing workgroup of its K slice (sc1 =
//              L1- and L2-bypassing loads) and fetches the 16 rows of a producer's units as soon as that producer's
//              piece is no longer the sentinel - the data is its own flag: a step costs one store -> load hop with
//              no drain, no counter and no second round trip -, then MFMA and partial sums into LDS;
//              2 (backward: 4) CELL waves add the partials, update the cell, publish h[t] (backward: dgates[t] as
//              bf16) with 16-byte sc1 write-through stores and do every other memory access of the step.
// 16-byte sc1 stores arrive as untorn 8-byte halves on gfx950 (MI355X_MICROARCH.md, hand-off table) and every element
// of every fetched piece is checked, so a piece passes only when all of it is new; a stale piece is fetched again.
// The cell update never produces the sentinel (a NaN of that bit pattern is rewritten to the canonical quiet NaN).
//
// Why the roles are separate waves: a wave's vector-memory operations complete in issue order, so a polling load
// queued behind the write-through publish store, the late stores or an operand fetch from HBM returns only after
// those.  Why a probe instead of polling with the sweep itself: 256 CUs re-reading 64 KB each per pass is 16 MB per
// pass on the memory side (sc1 loads do not hit in L2) - the passes then take 2.2 us each and the CU's own publish
// store queues behind them.  Measured per step at the benchmark shape (profiles/r02_wide_trace.txt), forward /
// backward: arrival counters + drained stores 8.7 / 6.65 us; sentinel polling by full sweeps 7.9 / 10.7; probe,
// then one sweep 5.2 / 7.3; probe and sweep interleaved 4.9 / 7.0; backward with 8-row groups 6.6; backward with a
// PROBER wave (an idle cell wave polls for the whole workgroup with nothing else in its memory queue and hands the
// producers' bits over in LDS) 6.0 - the forward kernel got slower with one (5.25) and keeps the sweepers' own probes.
// What is left is
// the hop itself: store -> visible + one probe round trip (1.9 us) + one data round trip (1.5 us; 2.2 us for the
// backward pass's 64 KB per CU), then ~1.5 us of MFMA, barrier and cell update.
// Round 3: the BACKWARD recurrence runs on lstm_wide_bwd_ps_kernel (further down) - partial sums of dh as self-flagging
// granules, one hop, 4.1 us per step; the sweep-form backward kernel below stays for NS_WIDE_PS=0 and the shapes the new
// one does not take (H not a multiple of 128, more than 256 workgroups).
// Every spin is bounded; a timeout raises the status word and all waves of the workgroup leave together.  The grid
// (row groups x unit blocks <= 256 workgroups, one per CU) must be resident at once: ns_lstm_wide_supported() refuses
// shapes that do not fit.
#include "common.h"
#include <stdint.h>
#include <stdlib.h>

namespace {
constexpr int WT = 512, WW = WT / 64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr size_t WIDE_TRACE_BYTES = 256 * 8 * sizeof(long long);

struct WideArgs {
  ns_lstm_seq_params p;
  int* status;
  long long* trace;     // NS_WIDE_TRACE=1: [step][8] timestamps (100 MHz) of workgroup 0, else null
  int nub;              // unit blocks per row group
};
__device__ __forceinline__ void wstamp(const WideArgs& a, int st, int k) {
  if (a.trace && blockIdx.x == 0 && threadIdx.x == 0 && st < 256) a.trace[st * 8 + k] = wall_clock64();
}

// ---- the sentinel: all ones (a NaN in both storage types)
template <typename T> __device__ __forceinline__ bool has_sentinel(const u32x4& v);
template <> __device__ __forceinline__ bool has_sentinel<float>(const u32x4& v) {
  return (v[0] == 0xffffffffu) | (v[1] == 0xffffffffu) | (v[2] == 0xffffffffu) | (v[3] == 0xffffffffu);
}
template <> __device__ __forceinline__ bool has_sentinel<bf16_t>(const u32x4& v) {
  bool b = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) b = b | ((v[i] & 0xffffu) == 0xffffu) | (v[i] >= 0xffff0000u);
  return b;
}
__device__ __forceinline__ float clean(float x, float*) { return __float_as_uint(x) == 0xffffffffu ? __uint_as_float(0x7fc00000u) : x; }
__device__ __forceinline__ bf16_t clean(float x, bf16_t*) {
  const bf16_t b = (bf16_t)x;
  return __builtin_bit_cast(unsigned short, b) == 0xffffu ? __builtin_bit_cast(bf16_t, (unsigned short)0x7fc0u) : b;
}

template <typename T>
__global__ __launch_bounds__(256) void wide_fill_kernel(T* base, int N, long P, int padl, int T_, long ld, int width) {
  // 16-byte pieces of rows (n, padl + t), t < T_
  const int ppr = width * (int)sizeof(T) / 16;
  const long total = (long)N * T_ * ppr;
  const u32x4 s = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / ppr;
    const int pc = (int)(i - row * ppr);
    const long n = row / T_, t = row - n * T_;
    *(u32x4*)((char*)(base + (n * P + padl + t) * ld) + pc * 16) = s;
  }
}

// Probe and sweep interleaved: the wave polls one piece per producer of its K slice (as probe()) and issues each of its NL
// sweep loads as soon as the PPCH producers that load covers (LPC consecutive loads share them) have published, so that
// when the last producer arrives only its own pieces are still to be fetched.  Every piece is checked once all are in
// (the probe looked at one row only) and a stale one is fetched again.
template <typename T, int NL, int LPC, int PPCH>
__device__ __forceinline__ unsigned sweep_progressive(const void* base, size_t bytes, unsigned poff0, unsigned pstride,
                                                      unsigned off0, unsigned in_grp, unsigned per_grp, u32x4 (&v)[NL], int lane,
                                                      int* status, int* abortf, int code, long long* tslot,
                                                      const unsigned long long* pmask = nullptr, int pbit0 = 0) {
  constexpr int NP = (NL / LPC) * PPCH;                       // producers of this wave's K slice
  constexpr unsigned long long ALLP = NP >= 64 ? ~0ull : ((1ull << NP) - 1ull);
  constexpr unsigned ALLL = (1u << NL) - 1u;
  const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)base >> 32));
  const int nrec = __builtin_amdgcn_readfirstlane((int)bytes);
  const unsigned poff = lane < NP ? poff0 + (unsigned)lane * pstride : 0x80000000u;
  unsigned long long ready = 0ull;
  unsigned done = 0u, spins = 0u, clk0 = 0u;
  for (;;) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)bhi << 32) | blo), 0, nrec, 0x00020000);
    if (ready != ALLP) {
      if (pmask) {            // a prober wave polls for the whole workgroup and keeps the producers' bits in LDS
        ready = (ns_lds_peek(pmask + (pbit0 >> 6)) >> (pbit0 & 63)) & ALLP;      // an LDS read: does not wait for the sweep's loads
      } else {
        const u32x4 pv = __builtin_amdgcn_raw_buffer_load_b128(rs, poff, 0, 16);
        ready |= __builtin_amdgcn_ballot_w64(lane < NP && !has_sentinel<T>(pv));
      }
      if (tslot && ready == ALLP && blockIdx.x == 0 && threadIdx.x == 0) *tslot = wall_clock64();
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      constexpr unsigned long long one = 1ull;
      const unsigned long long m = ((one << PPCH) - one) << ((l / LPC) * PPCH);
      if (!(done & (1u << l)) && (ready & m) == m) {
        const unsigned o = off0 + (unsigned)(l % LPC) * in_grp + (unsigned)(l / LPC) * per_grp;
        v[l] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 16);      // sc1, as every load of handed-off bytes
        done |= 1u << l;
      }
    }
    ++spins;
    if (done == ALLL) {
      unsigned stale = 0u;
#pragma unroll
      for (int l = 0; l < NL; ++l) if (__any(has_sentinel<T>(v[l]))) stale |= 1u << l;
      if (!stale) return spins;
      done &= ~stale;
    }
    if (pmask) {              // the prober owns the time-out and the look at the status word
      if (ns_lds_peek(abortf)) return 0;
      if ((spins & 0xffffu) == 0 && ns_spin_timed_out(clk0)) { if (lane == 0) { atomicExch(status, code); *abortf = 1; } return 0; }
      continue;
    }
    if ((spins & 255u) == 0) {
      if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { if (lane == 0) *abortf = 1; return 0; }
      if ((spins & 1023u) == 0 && ns_spin_timed_out(clk0)) { if (lane == 0) { atomicExch(status, code); *abortf = 1; } return 0; }
    }
  }
}

// The prober wave of a workgroup: polls ONE 16-byte piece per producing workgroup of the row group (lane l: producers
// l, l + 64; byte offset poff0 + producer * pstride) with nothing else in its memory queue, and keeps the bits of the
// producers that have published in pm[0..1] (LDS) for the sweepers.  Returns false on time-out / raised status.
template <typename T>
__device__ __forceinline__ bool probe_all(const void* base, size_t bytes, unsigned poff0, unsigned pstride, int np,
                                          unsigned long long* pm, int lane, int* status, int* abortf, int code) {
  const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)base >> 32));
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)bhi << 32) | blo), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  const unsigned o0 = lane < np ? poff0 + (unsigned)lane * pstride : 0x80000000u;
  const unsigned o1 = lane + 64 < np ? poff0 + (unsigned)(lane + 64) * pstride : 0x80000000u;
  const unsigned long long all0 = np >= 64 ? ~0ull : ((1ull << np) - 1ull);
  const unsigned long long all1 = np > 64 ? (np >= 128 ? ~0ull : ((1ull << (np - 64)) - 1ull)) : 0ull;
  unsigned long long m0 = 0ull, m1 = 0ull;
  unsigned spins = 0, clk0 = 0;
  for (;;) {
    const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, o0, 0, 16);
    const u32x4 v1 = np > 64 ? __builtin_amdgcn_raw_buffer_load_b128(rs, o1, 0, 16) : (u32x4){0u, 0u, 0u, 0u};
    m0 |= __builtin_amdgcn_ballot_w64(lane < np && !has_sentinel<T>(v0));
    m1 |= __builtin_amdgcn_ballot_w64(lane + 64 < np && !has_sentinel<T>(v1));
    if (lane == 0) { ns_lds_poke(pm, m0); ns_lds_poke(pm + 1, m1); }
    if (m0 == all0 && m1 == all1) return true;
    ++spins;
    if ((spins & 255u) == 0) {
      if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { if (lane == 0) *abortf = 1; return false; }
      if ((spins & 1023u) == 0 && ns_spin_timed_out(clk0)) { if (lane == 0) { atomicExch(status, code); *abortf = 1; } return false; }
    }
  }
}

// ===================================================================================== forward
// Waves 0..7 (the sweepers) split K: sweep, MFMA, partial sums into LDS.  Waves 8, 9 (the cell waves; thread = (row,
// unit)) add the partials, update the cell, publish h[t] and do every other memory access of the step.  The roles are
// separate because a wave's vector-memory operations complete in issue order: a polling load queued behind the
// write-through publish store, the late stores or an operand fetch from HBM returns only after those (the one-role
// version measured 3.7 us for a sweep that succeeded at its first pass).
// NCH = 32-wide K chunks per sweeper (H / 256); PASSES = 3: fp32 state, hi / lo weight planes; 1: bf16 everywhere
constexpr bool FWD_PROBER = false;          // measured: 5.25 us per step with a prober wave, 5.04 with the sweepers' own probes
                                            // (round 3, the mask polled as a plain LDS read instead of a flat load: 5.2 against 5.08)
constexpr int WTF = WT + 128 + (FWD_PROBER ? 64 : 0);
template <typename T, int PASSES, int NCH>
__global__ __launch_bounds__(WTF) void lstm_wide_fwd_kernel(WideArgs a) {
  // Partial sums, sweepers -> cell waves.  Row stride = 4 mod 16 floats: conflict-free for the MFMA C layout's writes
  // (+1 is not).  Two images by step parity (LDS hand-over audit, round 3): the barrier of step t orders the sweepers'
  // writes before the cell waves' reads, but a sweeper whose K slice does not hold this workgroup's own units needs
  // nothing from these cell waves to finish step t+1 - with ONE image its next write was kept behind their reads only by
  // the two memory round trips every sweep takes.  With two, image (t & 1) is written again in step t+2, which a sweeper
  // enters through the barrier of step t+1, and the cell waves reach that barrier after their reads of step t.
  __shared__ float red[2][WW][16][36];
  __shared__ __attribute__((aligned(16))) T hst[16][8];
  __shared__ int abortf;
  __shared__ unsigned long long pmask[2][2];       // [step parity][producers 0..63, 64..127] published bits (prober wave)
  const ns_lstm_seq_params& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, NUB = a.nub;
  const int rg = blockIdx.x / NUB, ub = blockIdx.x % NUB;
  const int n0 = rg * 16, u0 = ub * 8;
  if (tid == 0) abortf = 0;                          // (audit) both initialised in front of the barrier below
  if (tid < 4) pmask[tid >> 1][tid & 1] = 0ull;
  const size_t hbytes = (size_t)p.N * p.P * p.ld_h * sizeof(T);
  constexpr int PPC = 8 * (int)sizeof(T) / 16;        // 16-byte pieces per 8-value fragment (2 for fp32, 1 for bf16)
  __syncthreads();

  if (wave < WW) {
    // ------------------------------------------------------------------ sweepers
    const int r16 = lane & 15, g = lane >> 4;
    const int k0 = wave * (H / WW);
    // resident weight fragments: tile j, column r16 -> gate 2j + (r16 >> 3), unit u0 + (r16 & 7)
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
    const bool ok = n0 + r16 < p.N;
    // the probed row differs from workgroup to workgroup and wave to wave: 2,000 waves polling the same few lines would
    // all queue on the same memory channels
    const int prb = (ub * WW + wave) % min(16, p.N - n0);
    for (int t = 0; t < p.T; ++t) {
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      wstamp(a, t, 0);
      if (t > 0) {
        const unsigned rowoff = ok ? (unsigned)(((long)(n0 + r16) * p.P + p.padl + t - 1) * p.ld_h + k0 + g * 8) * (unsigned)sizeof(T) : 0x80000000u;
        u32x4 v[NCH * PPC];
        const unsigned prow = (unsigned)(((long)(n0 + prb) * p.P + p.padl + t - 1) * p.ld_h + k0) * (unsigned)sizeof(T);
        const unsigned got = sweep_progressive<T, NCH * PPC, PPC, 4>(p.h, hbytes, prow, 8u * (unsigned)sizeof(T), rowoff, 16u, 32u * (unsigned)sizeof(T),
                                                                     v, lane, a.status, &abortf, 1, a.trace && t < 256 ? a.trace + t * 8 + 6 : nullptr,
                                                                     FWD_PROBER ? &pmask[t & 1][0] : nullptr, wave * (H / (8 * WW)));
        wstamp(a, t, 1);
        if (a.trace && blockIdx.x == 0 && tid == 0 && t < 256) a.trace[t * 8 + 5] = got;
        if (got) {
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            bf16x8 ah, al;
            if constexpr (sizeof(T) == 4) {
              const u32x4 x = v[2 * c], y = v[2 * c + 1];
              const float f[8] = {__uint_as_float(x[0]), __uint_as_float(x[1]), __uint_as_float(x[2]), __uint_as_float(x[3]),
                                  __uint_as_float(y[0]), __uint_as_float(y[1]), __uint_as_float(y[2]), __uint_as_float(y[3])};
#pragma unroll
              for (int i = 0; i < 8; ++i) { const bf16_t h = (bf16_t)f[i]; ah[i] = h; al[i] = (bf16_t)(f[i] - (float)h); }
            } else {
              ah = *(bf16x8*)&v[c];
              al = ah;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              if constexpr (PASSES == 3) acc[j] = mfma_split<3>(ah, al, bh[j][c], bl[j][c], acc[j]);
              else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j][c], acc[j], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[t & 1][wave][g * 4 + q][j * 16 + r16] = acc[j][q];
      __syncthreads();
      if (abortf) return;
      wstamp(a, t, 2);
    }
    return;
  }
  if (FWD_PROBER && wave == WW + 2) {
    // ------------------------------------------------------------------ prober: one piece per producer of the row group
    const int prb = ub % min(16, p.N - n0);            // the probed row differs from workgroup to workgroup
    for (int t = 0; t < p.T; ++t) {
      if (t > 0) {
        const unsigned prow = (unsigned)(((long)(n0 + prb) * p.P + p.padl + t - 1) * p.ld_h) * (unsigned)sizeof(T);
        probe_all<T>(p.h, hbytes, prow, 8u * (unsigned)sizeof(T), H / 8, &pmask[t & 1][0], lane, a.status, &abortf, 1);
      }
      // (audit) pmask[parity]: filled by this prober, read by the sweepers of the same step.  The other parity is cleared
      // here, after this step's probe and before this step's barrier: its last readers were the sweepers of step t-1,
      // who passed the barrier of step t-1 before this iteration began; its next readers start after this barrier.
      if (lane < 2) pmask[(t + 1) & 1][lane] = 0ull;
      __syncthreads();
      if (abortf) return;
    }
    return;
  }
  // -------------------------------------------------------------------- cell waves: thread = (row er, unit eu)
  const int e = tid - WT, er = e >> 3, eu = e & 7;
  const int en = n0 + er;
  const bool eok = en < p.N;
  const int elen = (eok && p.lengths) ? p.lengths[en] : p.T;
  const auto hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.h, 0, (int)hbytes, 0x00020000);
  const bool tr = a.trace && blockIdx.x == 0 && e == 0;
  float cst = 0.f;
  float hprev = 0.f;          // zoneout: this thread's h of the step before (fp32, before the store's rounding)
  const bool zone = p.zoneout_thr_cell != 0u || p.zoneout_thr_output != 0u;
  float pz[4] = {0.f, 0.f, 0.f, 0.f};
  auto load_xg = [&](int t) {
    if (eok) {
      const float* xr = p.xg + ((long)en * p.P + p.padl + t) * p.ld_xg + u0 + eu;
#pragma unroll
      for (int j = 0; j < 4; ++j) pz[j] = xr[(long)j * H];
    }
  };
  load_xg(0);
  for (int t = 0; t < p.T; ++t) {
    __syncthreads();
    if (abortf) return;
    if (tr && t < 256) a.trace[t * 8 + 3] = wall_clock64();
    float z[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = pz[j];
#pragma unroll
      for (int w = 0; w < WW; ++w) s += red[t & 1][w][er][(j >> 1) * 16 + (j & 1) * 8 + eu];
      z[j] = s;
    }
    float gi = sigmoidf_(z[0]), gj = tanhf_(z[1]), gf = sigmoidf_(z[2] + p.forget_bias), go = sigmoidf_(z[3]);
    float hv;
    if (!zone) {
      cst = ns_cell_clip(gf * cst + gi * gj, p.cell_clip);
      hv = go * tanhf_(cst);
    } else {                  // ns_lstm_seq_params: a kept unit carries c / h of step t-1 on, h' comes from the plain c'
      const float cn = ns_cell_clip(gf * cst + gi * gj, p.cell_clip);
      hv = go * tanhf_(cn);
      if (!ns_zone_keep(p.zoneout_seed_cell, (uint32_t)t, (uint32_t)en, (uint32_t)(u0 + eu), p.zoneout_thr_cell)) cst = cn;
      if (ns_zone_keep(p.zoneout_seed_output, (uint32_t)t, (uint32_t)en, (uint32_t)(u0 + eu), p.zoneout_thr_output)) hv = hprev;
      hprev = hv;
    }
    if (t >= elen) { cst = 0.f; hv = 0.f; gi = gj = gf = go = 0.f; }
    // (audit) hst: written and read by the SAME cell wave (wave w owns rows 8w .. 8w+7 on both sides), so the release
    // fence + wave barrier below is all the ordering it needs; no other role touches it
    hst[er][eu] = clean(hv, (T*)nullptr);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // LDS write -> read inside this wave (its own 8 rows)
    __builtin_amdgcn_wave_barrier();
    // ---- publish h[t]: 16-byte write-through stores, each cell wave its own 8 rows
    if (lane < 8 * PPC) {
      const int row = (wave - WW) * 8 + lane / PPC, pc = lane % PPC;
      if (n0 + row < p.N) {
        const u32x4 x = *(const u32x4*)((const char*)&hst[row][0] + pc * 16);
        const unsigned off = (unsigned)(((long)(n0 + row) * p.P + p.padl + t) * p.ld_h + u0) * (unsigned)sizeof(T) + pc * 16;
        __builtin_amdgcn_raw_buffer_store_b128(x, hrs, off, 0, 16);            // aux 16 = sc1
      }
    }
    if (tr && t < 256) a.trace[t * 8 + 4] = wall_clock64();
    // ---- what only the backward pass reads, next step's input gates
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
// dh[t] = dh_out[t] + dgates[t+1] . Wh^T.  Workgroup = RPG rows x 16 units; RPG = 8 (the MFMA tile's other 8 rows stay
// zero, their lanes load nothing) when that still fits the device: the sweep is bound by what ONE CU can load per step
// (RPG x 4H bf16 = 64 KB at 8 rows), and 4 row groups x 64 unit blocks use all 256 CUs where 2 x 64 used half.
// Workgroup = RPG rows x 16 units (rows of Wh [H, 4H], K = 4H split over the 8
// sweepers, NCH = 32-wide chunks per sweeper = H / 64) + 4 cell waves (thread = (row, unit)); the exchanged payload is
// the bf16 copy of the gate gradients (dgates_bf16, or dgates itself when the storage type is bf16).
constexpr int WTB = WT + 256;
template <typename T, int NCH, int RPG>
__global__ __launch_bounds__(WTB) void lstm_wide_bwd_kernel(WideArgs a) {
  __shared__ float red[2][WW][16][20];    // two images by step parity, as in the forward kernel
  __shared__ __attribute__((aligned(16))) bf16_t dst[16][4][16];      // this step's gate gradients (row, gate, unit)
  __shared__ int abortf;
  __shared__ unsigned long long pmask[2][2];
  constexpr bool PROBER = RPG == 8;                // 8-row groups leave the last two cell waves idle: one of them probes
  const ns_lstm_seq_params& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, K = 4 * H, NUB = a.nub;
  const int rg = blockIdx.x / NUB, ub = blockIdx.x % NUB;
  const int n0 = rg * RPG, u0 = ub * 16;
  if (tid == 0) abortf = 0;
  if (tid < 4) pmask[tid >> 1][tid & 1] = 0ull;
  bf16_t* xb = sizeof(T) == 2 ? (bf16_t*)p.dgates : (bf16_t*)p.dgates_bf16;     // exchange payload [N*P, 4H] bf16
  const size_t xbytes = (size_t)p.N * p.P * K * sizeof(bf16_t);
  __syncthreads();

  if (wave < WW) {
    // ------------------------------------------------------------------ sweepers
    const int r16 = lane & 15, g = lane >> 4;
    const int k0 = wave * (K / WW);
    const bf16_t* W = sizeof(T) == 2 ? (const bf16_t*)p.wh : (const bf16_t*)p.wh_bf16;
    bf16x8 bw[NCH];
    {
      const bf16_t* row = W + (long)(u0 + r16) * K + k0 + g * 8;
#pragma unroll
      for (int c = 0; c < NCH; ++c) bw[c] = *(const bf16x8*)(row + c * 32);
    }
    const bool ok = r16 < RPG && n0 + r16 < p.N;
    const int prb = (ub * WW + wave) % min(RPG, p.N - n0);     // as in the forward kernel
    for (int t = p.T - 1; t >= 0; --t) {
      const int bs = p.T - 1 - t;                       // backward step index
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      wstamp(a, bs, 0);
      if (bs > 0) {
        const unsigned rowoff = ok ? (unsigned)(((long)(n0 + r16) * p.P + p.padl + t + 1) * K + k0 + g * 8) * 2u : 0x80000000u;
        u32x4 av[NCH];
        const unsigned prow = (unsigned)(((long)(n0 + prb) * p.P + p.padl + t + 1) * K + k0) * 2u;
        const unsigned got = sweep_progressive<bf16_t, NCH, 1, 2>(xb, xbytes, prow, 32u, rowoff, 0u, 64u, av, lane, a.status, &abortf, 2,
                                                                  a.trace && bs < 256 ? a.trace + bs * 8 + 6 : nullptr,
                                                                  PROBER ? &pmask[bs & 1][0] : nullptr, (wave & 1) * (H / 32));
        wstamp(a, bs, 1);
        if (a.trace && blockIdx.x == 0 && tid == 0 && bs < 256) a.trace[bs * 8 + 5] = got;
        if (got) {
          f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < NCH; c += 2) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&av[c], bw[c], acc, 0, 0, 0);
            if (c + 1 < NCH) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&av[c + 1], bw[c + 1], acc2, 0, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] += acc2[q];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) red[bs & 1][wave][g * 4 + q][r16] = acc[q];
      __syncthreads();
      if (abortf) return;
      wstamp(a, bs, 2);
    }
    return;
  }
  if (PROBER && wave == WW + 3) {
    // ------------------------------------------------------------------ prober (see the forward kernel): gate 0's piece of
    //                                                                      every producer's 16 units
    const int prb = ub % min(RPG, p.N - n0);
    for (int t = p.T - 1; t >= 0; --t) {
      const int bs = p.T - 1 - t;
      if (bs > 0) {
        const unsigned prow = (unsigned)(((long)(n0 + prb) * p.P + p.padl + t + 1) * K) * 2u;
        probe_all<bf16_t>(xb, xbytes, prow, 32u, H / 16, &pmask[bs & 1][0], lane, a.status, &abortf, 2);
      }
      if (lane < 2) pmask[(bs + 1) & 1][lane] = 0ull;
      __syncthreads();
      if (abortf) return;
    }
    return;
  }
  // -------------------------------------------------------------------- cell waves: thread = (row er, unit eu)
  const int e = tid - WT, er = e >> 4, eu = e & 15;
  const int en = n0 + er;
  const bool eok = er < RPG && en < p.N;
  const int elen = (eok && p.lengths) ? p.lengths[en] : p.T;
  const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)xbytes, 0x00020000);
  const bool tr = a.trace && blockIdx.x == 0 && e == 0;
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
  for (int t = p.T - 1; t >= 0; --t) {
    const int bs = p.T - 1 - t;
    __syncthreads();
    if (abortf) return;
    if (tr && bs < 256) a.trace[bs * 8 + 3] = wall_clock64();
    float dh = pdh;
#pragma unroll
    for (int w = 0; w < WW; ++w) dh += red[bs & 1][w][er][eu];
    const float gi = pg[0], gj = pg[1], gf = pg[2], go = pg[3];
    const float tc = tanhf_(pc);
    const float dc = dh * go * (1.f - tc * tc) + dcc;
    float dgv[4];
    dgv[0] = dc * gj * gi * (1.f - gi);
    dgv[1] = dc * gi * (1.f - gj * gj);
    dgv[2] = dc * pcp * gf * (1.f - gf);
    dgv[3] = dh * tc * go * (1.f - go);
    dcc = dc * gf;
    if (t >= elen || !eok) { dgv[0] = dgv[1] = dgv[2] = dgv[3] = 0.f; dcc = 0.f; }
#pragma unroll
    // (audit) dst: rows 4w .. 4w+3 are written (er = e >> 4) and published (row = 4w + lane / 8) by the same cell wave w
    for (int j = 0; j < 4; ++j) dst[er][j][eu] = clean(dgv[j], (bf16_t*)nullptr);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // LDS write -> read inside this wave (its own 4 rows)
    __builtin_amdgcn_wave_barrier();
    // ---- publish dgates[t] (bf16): this wave's 4 rows x 4 gates x 32 bytes = 32 pieces of 16 bytes
    if (lane < 32) {
      const int row = (wave - WW) * 4 + (lane >> 3), gate = (lane >> 1) & 3, hf = lane & 1;
      if (row < RPG && n0 + row < p.N) {
        const u32x4 x = *(const u32x4*)((const char*)&dst[row][gate][0] + hf * 16);
        const unsigned off = (unsigned)(((long)(n0 + row) * p.P + p.padl + t) * K + (long)gate * H + u0) * 2u + hf * 16;
        __builtin_amdgcn_raw_buffer_store_b128(x, xrs, off, 0, 16);
      }
    }
    if (tr && bs < 256) a.trace[bs * 8 + 4] = wall_clock64();
    // ---- fp32 copy of the gate gradients for the weight-gradient products (storage type float), next operands
    if (eok && sizeof(T) == 4) {
      float* dg = (float*)p.dgates + ((long)en * p.P + p.padl + t) * K + u0 + eu;
#pragma unroll
      for (int j = 0; j < 4; ++j) dg[(long)j * H] = dgv[j];
    }
    if (t > 0) load_ops(t - 1);
  }
}

// ===================================================================================== backward, partial sums (round 3)
// The backward kernel above gathers, per workgroup and step, ALL 4H gate gradients of its 8 rows (64 KB at H = 1024)
// behind a probe: two dependent memory round trips, 5.9 us per step.  dh[t-1] = dgates[t] . Wh^T is a sum over the gate
// columns, and a workgroup OWNS 64 of them (4 gates x its 16 units): as in lstm_cluster2p_bwd_kernel it forms, from its
// own gate gradients alone and straight after the cell update, its partial sum for EVERY unit of the layer - P[8, H] =
// dg_own[8, 64] . Wh[:, own columns]^T, H / 16 MFMA tiles of K = 64 - and sends each peer only the 8 x 16 block of that
// peer's units, as {step tag, 2 x bf16} granules (two rows of one unit): 64 granules per (destination, source), the data
// is its own flag - no probe, no second round trip, and half the bytes (32 KB in and out per workgroup and step).
// The receiver adds the blocks in a fixed order (own block fp32 first, then sources in the polling order below).
//   8 product waves  : resident Wh[units of their H/128 destination tiles][own 64 columns] (64 VGPRs per lane at
//                      H = 1024); per step: poll the granules of 1/8 of the sources (lane = one (unit, row pair) item),
//                      sum them, LDS -> barrier A -> (cell waves) -> barrier B -> MFMA over the operand image ->
//                      publish (the own tile goes to LDS as fp32)
//   2 cell waves     : thread = (row, unit); behind barrier A add the eight partial sums, own block and dh_out, update
//                      the cell, write the bf16 operand image, barrier B; then the stores only later kernels read and
//                      the next step's operand loads.
// Exchange buffer (work): [row group][step parity][destination][source][64 items] of 8 bytes, zeroed per launch.
constexpr int PSW = 8;                       // product waves
constexpr int PST = PSW * 64 + 128;          // + two cell waves
typedef unsigned long long ps_u64;
// orders LDS only: __syncthreads() would also drain the vector-memory queue (the publish stores' acknowledgements)
__device__ __forceinline__ void ps_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// UPB = units per workgroup (16 or 32).  What a step costs is the write-through traffic: every (row, destination unit)
// gets one partial sum from every SOURCE block, rows x H x (H / UPB) values per step chip-wide - 8 MB of granules at
// UPB = 16, and the fabric takes write-through stores at ~3.3 TB/s (2.4 us per step just to issue them, measured); 32-unit
// blocks halve the sources and with them every byte written and polled (the resident weights double: Wh[H units][own
// 128 columns] = 256 KB per workgroup, three quarters in registers, the last k-step in LDS), on half the workgroups.
template <typename T, int NT, int UPB>       // NT = destination tiles per product wave = H / 128
__global__ __launch_bounds__(PST) void lstm_wide_bwd_ps_kernel(WideArgs a, ps_u64* xbuf) {
  constexpr int TB = UPB / 16;               // 16-unit tiles per block
  constexpr int KO = 4 * UPB, KS = KO / 32;  // own gate columns, k-steps of the product
  constexpr int ITEMS = UPB * 4;             // (unit, row pair) items per (destination, source) block
  constexpr int IPL = ITEMS / 64;            // items per polling lane
  constexpr int LDW = KO + 8;                // bf16 per row of the operand image
  constexpr int KSR = UPB == 32 ? KS - 1 : KS;          // k-steps of the weights kept in registers (the rest: LDS)
  extern __shared__ __attribute__((aligned(16))) char ps_smem[];
  bf16_t* dgi = (bf16_t*)ps_smem;                               // [16][LDW], rows 8 .. 15 stay zero
  float* red = (float*)(dgi + 16 * LDW);                        // [PSW][ITEMS][2]
  float* ownp = red + PSW * ITEMS * 2;                          // [ITEMS][2]
  int* abortf = (int*)(ownp + ITEMS * 2);                       // [4]
  bf16x8* wl = (bf16x8*)(abortf + 4);                           // [PSW * NT tiles][64 lanes] the last k-step's fragments (UPB = 32)
  const ns_lstm_seq_params& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, K = 4 * H, NUB = H / UPB;
  const int rg = blockIdx.x / NUB, ub = blockIdx.x % NUB;
  const int n0 = rg * 8, u0 = ub * UPB;
  ps_u64* xb0 = xbuf + (size_t)rg * 2 * NUB * NUB * ITEMS;
  if (tid == 0) abortf[0] = 0;
  for (int i = tid; i < 16 * LDW; i += PST) dgi[i] = (bf16_t)0.f;
  for (int i = tid; i < ITEMS * 2; i += PST) ownp[i] = 0.f;
  for (int i = tid; i < PSW * ITEMS * 2; i += PST) red[i] = 0.f;

  if (wave < PSW) {
    // ------------------------------------------------------------------ product / polling waves
    const int r16 = lane & 15, g = lane >> 4;
    const bf16_t* W = sizeof(T) == 2 ? (const bf16_t*)p.wh : (const bf16_t*)p.wh_bf16;
    // B fragments: tile j -> destination units (wave * NT + j) * 16 ..; lane (n = r16, k chunk g): own column k = 32 ks +
    // 8 g + e -> gate k / UPB, own unit k % UPB (8 consecutive units: one 16-byte load)
    bf16x8 bw[NT][KSR];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k = 32 * ks + 8 * g;
        const bf16x8 w = *(const bf16x8*)(W + (long)((wave * NT + j) * 16 + r16) * K + (long)(k / UPB) * H + u0 + (k % UPB));
        if (ks < KSR) bw[j][ks] = w;
        else wl[(wave * NT + j) * 64 + lane] = w;
      }
    __syncthreads();
    for (int bs = 0; bs < p.T; ++bs) {
      // ---- the peers' blocks of the step before: lane = items lane, lane + 64 ..; sources wave + 8 j (fixed order)
      float s0[IPL], s1[IPL];
#pragma unroll
      for (int i = 0; i < IPL; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
      wstamp(a, bs, 0);
      unsigned passes = 0;
      if (bs > 0) {
        const ps_u64* cur = xb0 + ((size_t)(bs & 1) * NUB + ub) * NUB * ITEMS + lane;
        constexpr int NSRC = NT * 16 / UPB;            // sources per wave = NUB / PSW
        ps_u64 v[NSRC][IPL];
        unsigned spins = 0, clk0 = 0;
        bool ok;
        do {
          ok = true;
#pragma unroll
          for (int j = 0; j < NSRC; ++j) {
            const int src = wave + PSW * j;
#pragma unroll
            for (int i = 0; i < IPL; ++i)
              v[j][i] = src == ub ? ((ps_u64)(unsigned)bs << 32)
                                  : __hip_atomic_load((const NS_GLOBAL ps_u64*)(cur + (size_t)src * ITEMS + 64 * i), __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int j = 0; j < NSRC; ++j)
#pragma unroll
            for (int i = 0; i < IPL; ++i) ok = ok && ((unsigned)(v[j][i] >> 32) == (unsigned)bs);
          if (!ok && (++spins & 1023u) == 0) {   /* spins = polling passes beyond the first */
            if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ns_lds_poke(abortf, 1); ok = true; }
            else if (ns_spin_timed_out(clk0)) { atomicExch(a.status, 2); ns_lds_poke(abortf, 1); ok = true; }
          }
        } while (!ok);
        passes = spins;
#pragma unroll
        for (int j = 0; j < NSRC; ++j)
#pragma unroll
          for (int i = 0; i < IPL; ++i) {
            const unsigned pay = (unsigned)v[j][i];
            s0[i] += __uint_as_float(pay << 16);
            s1[i] += __uint_as_float(pay & 0xffff0000u);
          }
      }
      wstamp(a, bs, 1);
      if (a.trace && blockIdx.x == 0 && tid == 0 && bs < 256) a.trace[bs * 8 + 5] = passes;
      if (a.trace && blockIdx.x == 0 && tid == 7 * 64 && bs < 256) a.trace[bs * 8 + 7] = wall_clock64();      // wave 7's polls done
#pragma unroll
      for (int i = 0; i < IPL; ++i) {
        red[((wave * ITEMS) + lane + 64 * i) * 2] = s0[i];
        red[((wave * ITEMS) + lane + 64 * i) * 2 + 1] = s1[i];
      }
      ps_barrier();                          // A: the partial sums are in LDS
      wstamp(a, bs, 2);
      if (abortf[0]) return;
      ps_barrier();                          // B: the operand image of this step is complete
      wstamp(a, bs, 3);
      if (bs + 1 < p.T) {
        bf16x8 af[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = *(const bf16x8*)(dgi + r16 * LDW + 32 * ks + 8 * g);
        ps_u64* nxt = xb0 + (size_t)((bs + 1) & 1) * NUB * NUB * ITEMS;
        // four tiles at a time, k-step outermost: four independent accumulator chains keep the matrix pipe issuing
        // (one tile after the other is a chain of KS dependent MFMAs each), and the first group's stores go out under
        // the second group's products
        constexpr int TG = NT < 4 ? NT : 4;
#pragma unroll
        for (int j0 = 0; j0 < NT; j0 += TG) {
          f32x4 acc[TG];
#pragma unroll
          for (int jj = 0; jj < TG; ++jj) acc[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int jj = 0; jj < TG; ++jj)
              acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], ks < KSR ? bw[j0 + jj][ks < KSR ? ks : 0] : wl[(wave * NT + j0 + jj) * 64 + lane],
                                                                acc[jj], 0, 0, 0);
#pragma unroll
          for (int jj = 0; jj < TG; ++jj) {
            // D: column r16 = unit of the destination tile, rows 4 g + q (rows 0 .. 7 are this workgroup's):
            // item = unit in the destination block + UPB * (row pair)
            const int tile = wave * NT + j0 + jj, dest = tile / TB, ui = (tile % TB) * 16 + r16;
            if (g < 2) {
              if (dest == ub) {
                ownp[(ui + UPB * (2 * g)) * 2] = acc[jj][0]; ownp[(ui + UPB * (2 * g)) * 2 + 1] = acc[jj][1];
                ownp[(ui + UPB * (2 * g + 1)) * 2] = acc[jj][2]; ownp[(ui + UPB * (2 * g + 1)) * 2 + 1] = acc[jj][3];
              } else {
                NS_GLOBAL ps_u64* dst = (NS_GLOBAL ps_u64*)(nxt + ((size_t)dest * NUB + ub) * ITEMS + ui + 2 * UPB * g);
                const ps_u64 tg = (ps_u64)(unsigned)(bs + 1) << 32;
                const bf16_t b0 = (bf16_t)acc[jj][0], b1 = (bf16_t)acc[jj][1], b2 = (bf16_t)acc[jj][2], b3 = (bf16_t)acc[jj][3];
                const unsigned p01 = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
                const unsigned p23 = (unsigned)__builtin_bit_cast(unsigned short, b2) | ((unsigned)__builtin_bit_cast(unsigned short, b3) << 16);
                __hip_atomic_store(dst, tg | p01, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + UPB, tg | p23, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
          }
        }
      }
      wstamp(a, bs, 4);
    }
    return;
  }
  // -------------------------------------------------------------------- cell waves: thread = (row er, units eu + 16 c)
  __syncthreads();                           // pairs with the product waves' barrier behind their weight loads
  const int e = tid - PSW * 64, er = e >> 4, eu = e & 15;
  const int en = n0 + er;
  const bool eok = en < p.N;
  const int elen = (eok && p.lengths) ? p.lengths[en] : p.T;
  const int half = er & 1;
  float dcc[TB], dhc[TB];
  const bool zone = p.zoneout_thr_cell != 0u || p.zoneout_thr_output != 0u;
  float pg[TB][4], pdh[TB], pc[TB], pcp[TB];
#pragma unroll
  for (int c = 0; c < TB; ++c) { dcc[c] = 0.f; dhc[c] = 0.f; pdh[c] = 0.f; pc[c] = 0.f; pcp[c] = 0.f; pg[c][0] = pg[c][1] = pg[c][2] = pg[c][3] = 0.f; }
  auto load_ops = [&](int t) {
    if (eok) {
      const long rowi = (long)en * p.P + p.padl + t;
#pragma unroll
      for (int c = 0; c < TB; ++c) {
        const int u = u0 + eu + 16 * c;
        const T* gp = (const T*)p.gates + rowi * 4 * H + u;
#pragma unroll
        for (int j = 0; j < 4; ++j) pg[c][j] = ldf(gp + (long)j * H);
        pdh[c] = p.dh[rowi * p.ld_dh + u];
        pc[c] = p.c[rowi * H + u];
        pcp[c] = t > 0 ? p.c[(rowi - 1) * H + u] : 0.f;
      }
    }
  };
  load_ops(p.T - 1);
  for (int t = p.T - 1; t >= 0; --t) {
    if (a.trace && blockIdx.x == 0 && e == 0 && p.T - 1 - t < 256) a.trace[(p.T - 1 - t) * 8 + 6] = wall_clock64();      // a cell wave reaches A
    ps_barrier();                            // A
    if (abortf[0]) return;
    float dgv[TB][4];
#pragma unroll
    for (int c = 0; c < TB; ++c) {
      const int item = eu + 16 * c + UPB * (er >> 1);
      float dh = pdh[c] + ownp[item * 2 + half];
#pragma unroll
      for (int w = 0; w < PSW; ++w) dh += red[(w * ITEMS + item) * 2 + half];
      const float gi = pg[c][0], gj = pg[c][1], gf = pg[c][2], go = pg[c][3];
      float cc = pc[c], dcin = dcc[c], dckeep = 0.f;
      if (zone) {             // the gradient of the forward kernel's masks (include/nspeech_hip.h, ns_lstm_seq_params)
        const uint32_t uu = (uint32_t)(u0 + eu + 16 * c);
        dh += dhc[c];
        const bool mh = ns_zone_keep(p.zoneout_seed_output, (uint32_t)t, (uint32_t)en, uu, p.zoneout_thr_output);
        const bool mc = ns_zone_keep(p.zoneout_seed_cell, (uint32_t)t, (uint32_t)en, uu, p.zoneout_thr_cell);
        dhc[c] = mh ? dh : 0.f;
        if (mh) dh = 0.f;
        if (mc) { dckeep = dcin; dcin = 0.f; }
        cc = gf * pcp[c] + gi * gj;
      }
      const float tc = tanhf_(cc);
      const float dc = dh * go * (1.f - tc * tc) + dcin;
      dgv[c][0] = dc * gj * gi * (1.f - gi);
      dgv[c][1] = dc * gi * (1.f - gj * gj);
      dgv[c][2] = dc * pcp[c] * gf * (1.f - gf);
      dgv[c][3] = dh * tc * go * (1.f - go);
      dcc[c] = dc * gf + dckeep;
      if (t >= elen || !eok) { dgv[c][0] = dgv[c][1] = dgv[c][2] = dgv[c][3] = 0.f; dcc[c] = 0.f; dhc[c] = 0.f; }
#pragma unroll
      for (int j = 0; j < 4; ++j) dgi[er * LDW + j * UPB + eu + 16 * c] = (bf16_t)dgv[c][j];
    }
    ps_barrier();                            // B
    // ---- what only later kernels read: the gate gradients (bf16 copy and, for fp32 storage, fp32), next operands
    if (eok) {
      const long rowi = (long)en * p.P + p.padl + t;
      bf16_t* xb = sizeof(T) == 2 ? (bf16_t*)p.dgates : (bf16_t*)p.dgates_bf16;
#pragma unroll
      for (int c = 0; c < TB; ++c) {
        bf16_t* db = xb + rowi * K + u0 + eu + 16 * c;
#pragma unroll
        for (int j = 0; j < 4; ++j) db[(long)j * H] = (bf16_t)dgv[c][j];
        if (sizeof(T) == 4) {
          float* dg = (float*)p.dgates + rowi * K + u0 + eu + 16 * c;
#pragma unroll
          for (int j = 0; j < 4; ++j) dg[(long)j * H] = dgv[c][j];
        }
      }
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
  if (nub % 8 != 0 || nrg * nub > ns_device_cus()) return false;
  if (nub_out) *nub_out = nub;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  const long esz = p->dtype == NS_BF16 ? 2 : 4;
  if (!backward) {
    if (p->dtype == NS_F32 && !(p->f32_passes == 3 && p->whT_hi && p->whT_lo)) return false;
    if (p->dtype == NS_BF16 && !p->whT) return false;
    if (!p->xg || !p->h || !p->c || !al16(p->h) || (p->ld_h * esz) % 16 != 0) return false;
    if ((double)p->N * p->P * p->ld_h * esz >= 2.0e9) return false;
  } else {
    if (p->zoneout_thr_cell || p->zoneout_thr_output) {       // only the partial-sum backward kernel applies the masks
      const char* ps_env = getenv("NS_WIDE_PS");
      const int ps_mode = ps_env ? atoi(ps_env) : 32;
      const int upb = ps_mode == 16 ? 16 : 32;
      if (!(H % 128 == 0 && ps_mode != 0 && ((p->N + 7) / 8) * (H / upb) <= ns_device_cus())) return false;
    }
    if (p->dtype == NS_F32 && !(p->f32_passes == 1 && p->wh_bf16 && p->dgates_bf16)) return false;
    if (p->dtype == NS_BF16 && !p->wh) return false;
    if (!p->gates || !p->c || !p->dh || !p->dgates) return false;
    if (!al16(p->dtype == NS_BF16 ? p->dgates : p->dgates_bf16)) return false;
    if ((double)p->N * p->P * 4 * H * 2 >= 2.0e9) return false;
  }
  return true;
}
}  // namespace

extern "C" int ns_lstm_wide_supported(const ns_lstm_seq_params* p, int backward) { return wide_shape_ok(p, backward, nullptr) ? 1 : 0; }
extern "C" size_t ns_lstm_wide_work_bytes(const ns_lstm_seq_params* p) {
  if (!p) return 0;
  // status word, the NS_WIDE_TRACE timestamps, the partial-sum backward kernel's exchange buffer
  size_t ex = 0;
  if (p->H % 128 == 0 && p->H >= 256 && p->H <= 1024 && ((p->N + 7) / 8) * (p->H / 32) <= 256)
    ex = (size_t)((p->N + 7) / 8) * 2 * (p->H / 16) * (p->H / 16) * 64 * sizeof(ps_u64);      // the 16-unit form's (the 32-unit form needs half)
  return 256 + WIDE_TRACE_BYTES + ex + 64;
}

template <typename T, int PASSES>
static int launch_wide_fwd(const WideArgs& a, int grid, hipStream_t s) {
  switch (a.p.H) {
    case 256: hipLaunchKernelGGL((lstm_wide_fwd_kernel<T, PASSES, 1>), dim3(grid), dim3(WTF), 0, s, a); break;
    case 512: hipLaunchKernelGGL((lstm_wide_fwd_kernel<T, PASSES, 2>), dim3(grid), dim3(WTF), 0, s, a); break;
    default: hipLaunchKernelGGL((lstm_wide_fwd_kernel<T, PASSES, 4>), dim3(grid), dim3(WTF), 0, s, a); break;
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
  a.p = *p; a.status = (int*)work; a.nub = nub;
  a.trace = getenv("NS_WIDE_TRACE") ? (long long*)((char*)work + 256) : nullptr;
  int rc = ns_zero_async(work, 256, s);
  if (rc) return rc;
  const int grid = ((p->N + 15) / 16) * nub;
  if (p->dtype == NS_BF16) {
    hipLaunchKernelGGL(wide_fill_kernel<bf16_t>, dim3(512), dim3(256), 0, s, (bf16_t*)p->h, p->N, (long)p->P, p->padl, p->T, (long)p->ld_h, p->H);
    NS_CHECK_LAUNCH("lstm_wide_fill");
    return launch_wide_fwd<bf16_t, 1>(a, grid, s);
  }
  hipLaunchKernelGGL(wide_fill_kernel<float>, dim3(512), dim3(256), 0, s, (float*)p->h, p->N, (long)p->P, p->padl, p->T, (long)p->ld_h, p->H);
  NS_CHECK_LAUNCH("lstm_wide_fill");
  return launch_wide_fwd<float, 3>(a, grid, s);
}

template <typename T, int RPG>
static int launch_wide_bwd(const WideArgs& a, int grid, hipStream_t s) {
  switch (a.p.H) {
    case 256: hipLaunchKernelGGL((lstm_wide_bwd_kernel<T, 4, RPG>), dim3(grid), dim3(WTB), 0, s, a); break;
    case 512: hipLaunchKernelGGL((lstm_wide_bwd_kernel<T, 8, RPG>), dim3(grid), dim3(WTB), 0, s, a); break;
    default: hipLaunchKernelGGL((lstm_wide_bwd_kernel<T, 16, RPG>), dim3(grid), dim3(WTB), 0, s, a); break;
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
  a.p = *p; a.status = (int*)work; a.nub = nub;
  a.trace = getenv("NS_WIDE_TRACE") ? (long long*)((char*)work + 256) : nullptr;
  int rc = ns_zero_async(work, 256, s);
  if (rc) return rc;
  const bool r8 = ((p->N + 7) / 8) * nub <= ns_device_cus();
  // the partial-sum exchange (lstm_wide_bwd_ps_kernel): 8-row groups, H a multiple of 128; NS_WIDE_PS=0 keeps the sweep
  // form, 16 the 16-unit blocks
  {
    const char* ps_env = getenv("NS_WIDE_PS");
    const int ps_mode = ps_env ? atoi(ps_env) : 32;
    const int nrg = (p->N + 7) / 8;
    const int upb = ps_mode == 16 ? 16 : 32, nb = p->H / upb, items = upb * 4;
    if (p->H % 128 == 0 && ps_mode != 0 && nrg * nb <= ns_device_cus()) {      // 8-row groups x unit blocks, one workgroup per CU
      ps_u64* xbuf = (ps_u64*)(((uintptr_t)work + 256 + WIDE_TRACE_BYTES + 15) & ~(uintptr_t)15);
      const size_t xbytes = (size_t)nrg * 2 * nb * nb * items * sizeof(ps_u64);
      rc = ns_zero_async(xbuf, xbytes, s);
      if (rc) return rc;
      const int grid = nrg * nb;
      const int nt = p->H / 128;
      const size_t ldsb = (size_t)16 * (4 * upb + 8) * 2 + sizeof(float) * (PSW * items * 2 + items * 2) + 16 +
                          (upb == 32 ? (size_t)PSW * nt * 64 * 16 : 0);
#define NS_LAUNCH_PS2(T_, NT_, UPB_) \
      do { \
        static bool attr_ = false; \
        if (!attr_) { (void)hipFuncSetAttribute((const void*)lstm_wide_bwd_ps_kernel<T_, NT_, UPB_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_ = true; } \
        hipLaunchKernelGGL((lstm_wide_bwd_ps_kernel<T_, NT_, UPB_>), dim3(grid), dim3(PST), ldsb, s, a, xbuf); \
      } while (0)
#define NS_LAUNCH_PS(T_) \
      if (upb == 16) { \
        switch (nt) { case 2: NS_LAUNCH_PS2(T_, 2, 16); break; case 4: NS_LAUNCH_PS2(T_, 4, 16); break; default: NS_LAUNCH_PS2(T_, 8, 16); break; } \
      } else { \
        switch (nt) { case 2: NS_LAUNCH_PS2(T_, 2, 32); break; case 4: NS_LAUNCH_PS2(T_, 4, 32); break; default: NS_LAUNCH_PS2(T_, 8, 32); break; } \
      }
      if (p->dtype == NS_BF16) { NS_LAUNCH_PS(bf16_t) } else { NS_LAUNCH_PS(float) }
#undef NS_LAUNCH_PS
#undef NS_LAUNCH_PS2
      NS_CHECK_LAUNCH("lstm_wide_bwd_ps");
      return NS_OK;
    }
  }
  const int grid = (r8 ? (p->N + 7) / 8 : (p->N + 15) / 16) * nub;
  bf16_t* xb = p->dtype == NS_BF16 ? (bf16_t*)p->dgates : (bf16_t*)p->dgates_bf16;
  hipLaunchKernelGGL(wide_fill_kernel<bf16_t>, dim3(1024), dim3(256), 0, s, xb, p->N, (long)p->P, p->padl, p->T, 4L * p->H, 4 * p->H);
  NS_CHECK_LAUNCH("lstm_wide_fill");
  if (p->dtype == NS_BF16) return r8 ? launch_wide_bwd<bf16_t, 8>(a, grid, s) : launch_wide_bwd<bf16_t, 16>(a, grid, s);
  return r8 ? launch_wide_bwd<float, 8>(a, grid, s) : launch_wide_bwd<float, 16>(a, grid, s);
}
