// Persistent GRU recurrences for the Tacotron-1 path: ONE launch per direction pair instead of four launches per
// time step (gate product, r * h, candidate product, state update - models/tacotron.py: _gru_seq).
//
// tf.contrib.rnn.GRUCell (modules.py:172-181 under bidirectional_dynamic_rnn for both CBHGs; tacotron.py:69-76 for the
// residual decoder cells), input halves hoisted by the caller:
//   [r | u] = sigmoid(xg[t] + h . Wg_h)     c = tanh(xc[t] + (r * h) . Wc_h)     h' = u * h + (1 - u) * c
// TWO dependent products per step, so a cell spread over several CUs would pay two CU-to-CU hops (~1.2 us each) per
// step.  The recurrent matrices of the 128-unit cells are 128 x 384 values = 96 KB as bf16, 192 KB as (hi, lo) planes:
// they fit ONE CU's registers.  So
//   H = 128: a chain = (direction, 16 batch rows) is ONE workgroup and nothing leaves the CU inside the loop: the weights
//            are MFMA A fragments in registers (8 compute waves x one 16-unit block x {r, u, c} tiles), the state and
//            r * h are bf16 (hi / lo) operand images in LDS, two LDS-only barriers per step;
//   H = 256: (hi, lo) planes are 768 KB, more than a CU holds: a chain is a cluster of 4 workgroups with 64 units each
//            (the same 48 K values per workgroup); r * h and h travel as {step tag, fp32} granules written and polled
//            with relaxed agent-scope atomics (the data is its own flag, cdna guide G16), gathered by two poller waves.
//            A 16-unit block's K = 256 is split over two compute waves (the halves meet through LDS), which keeps the
//            resident weights of every compute wave at 96 VGPRs in both shapes.
// The product is taken transposed (weights = A operand, state = B operand): a lane's accumulators are 4 consecutive
// UNITS of one batch row, so gate math, state, stores and granules need no cross-lane traffic.
// Roles as in lstm_cluster.hip (no wave mixes global loads and global stores - one vmcnt queue): compute waves touch LDS
// and issue stores only; prefetcher waves stream the per-step operands one slot ahead into an LDS stage; poller waves
// (H = 256) spin on the granules.  Every spin is bounded in wall-clock time; on a time-out the status word is raised
// and every role of every workgroup of the chain leaves.
//
// LDS hand-over audit (one image per operand, no double buffering needed):
//   forward   himg   written in phase 2 of slot s (own block, G = 1) or by the pollers between Bp2(s) and B0(s+1); read in
//                    phase 1 of slot s+1 behind B0(s+1); phase 1 reads of slot s are complete at B1(s) (wg_barrier waits
//                    for lgkmcnt(0)), in front of any write of slot s.  A peer's h(s) granule can only exist once every
//                    workgroup's r*h(s) is published, i.e. after all phase-1 reads of slot s everywhere.
//             rhimg  written in phase 1 of slot s / by the pollers between B0(s) [Bp1(s)] and B1(s); read in phase 2 behind
//                    B1(s); the next write is behind B0(s+1), which every wave joins after its phase-2 reads.
//             xs[b]  prefetcher writes slot s+1 behind B0(s); its last readers (phase 2 of slot s-1) joined B0(s) after
//                    their reads; read behind B0(s+1).
//             pbuf   (K split) written in front of Bp, partner reads behind it; the next write is behind the next B.
//   backward  dzcimg written in phase 1 of slot s (behind B2(s-1) in program order), read behind B1(s), reads complete at
//                    B2(s); dzgimg written in phase 2 (behind B1(s)), read behind B2(s), reads complete at B1(s+1);
//             st[b]  prefetchers write slot s+1 between B1(s) and B2(s); read in phase 1 of slot s+1 behind B2(s).
#include "common.h"
#include <stdint.h>
#include <stdlib.h>

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

namespace {

struct GruArgs {
  int N, T, P, padl, ndir, nrg;
  int reverse[2];
  const int* lengths;
  const float* xg[2]; const float* xc[2]; int ld_xg, ld_xc;
  const void* wgT[2]; const void* wcT[2];
  const void* wg[2]; const void* wc[2]; int ld_wg, ld_wc;
  void* h[2]; int ld_h;
  float* ru[2]; float* c[2]; void* rh[2];
  const float* h_init[2]; int ld_hi;
  const float* dh[2]; int ld_dh;
  void* dzg[2]; void* dzc[2];
  float* dh_init[2]; int ld_dhi;
  u64* xbuf;                          // [chains][granules per chain]
  int* status;
  int dbg;                            // NS_GRU_DBG (timing experiments only, 0 in production): 1 no prefetch, 2 no result stores
  long long* trace;                   // NS_GRU_TRACE=1 (diagnostic instantiation): [wave][8] segment sums of workgroup 0, 10 ns ticks
};

template <int H_> struct GCfg {
  static constexpr int H = H_;
  static constexpr int G = H == 128 ? 1 : 4;        // workgroups per chain
  static constexpr int U = H / G;                   // units per workgroup
  static constexpr int NB = U / 16;                 // 16-unit blocks per workgroup
  // compute waves: eight in both shapes (96 weight registers per wave: at H = 256 a block's K is split over two waves).
  // Measured and dropped (GW = NB: four waves x 192 weight registers with the whole K, no partner barrier): a lone wave per
  // SIMD spends 1.3 us in the gate / publish / store segment of a slot where two waves per SIMD take 0.55 - 1.15:
  // forward 7.6 against 6.2 us per step, backward (one pass) 6.35 against 6.9.
  static constexpr int GW = 8;
  static constexpr int KSPLIT = GW / NB;            // compute waves per block (each takes 1 / KSPLIT of K); 2 = the first form
  static constexpr int KS = H / 32;                 // k-steps of a product over H
  static constexpr int KSW = KS / KSPLIT;           // ... per wave
  static constexpr int NR = 4 / KSPLIT;             // accumulator registers (units) a wave finishes per lane
  static constexpr int NPOLL = G > 1 ? 2 : 0;
  static constexpr int FW_WAVES = GW + 1 + NPOLL;   // compute, prefetcher, pollers
  static constexpr int BW_WAVES = GW + 2 + NPOLL;   // compute, two prefetchers, pollers
  // The per-step operands come in by LDS-DMA (global_load_lds_dwordx4: no register staging, loads two slots ahead).  An
  // instruction lands 64 x 16 bytes lane-linear in LDS, so the stage is laid out [16-byte chunk][16 batch rows]: lane l of
  // instruction j fetches chunk 4 j + (l >> 4) of row l & 15 - and the compute lanes' float4 reads (row = l & 15) walk 16
  // consecutive 16-byte units: conflict free.
  static constexpr int NBUF = 3;                    // stage buffers: slot s + 2 is fetched while slot s is computed
  static constexpr int SECF = (U / 4) * 64;         // floats of one fp32 section (U / 4 chunks x 16 rows x 4)
  static constexpr int XS_F = 3 * SECF;             // forward slot: r | u | c parts of the own units
  static constexpr int FW_NI = XS_F / 256;          // DMA instructions per forward slot (one prefetcher wave)
  static_assert(H == 128 || H == 256, "H");
  static_assert(NB * KSPLIT == GW && KS % KSPLIT == 0, "shape");
};

// bf16 element offset in a [16][W + 8] operand image.  Rows are padded by 16 bytes instead of XOR-swizzled: a row stride of
// (W + 8) * 2 bytes = 4 banks mod 64 keeps the 16 rows of a ds_read_b128 fragment read on different banks, and every offset
// is (a lane's base) + (a compile-time constant) - the pollers' 32 - 64 image writes per gather carry no address arithmetic.
__device__ __forceinline__ int swz(int row, int k, int W) { return row * (W + 8) + k; }
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 8 consecutive weights as an MFMA fragment: hi = bf16(w), lo = bf16(w - hi) (lo only for three passes)
template <typename T, int NPL>
__device__ __forceinline__ void load_frag(const T* p, bf16x8& hi, bf16x8& lo) {
  if constexpr (sizeof(T) == 2) {
    hi = *(const bf16x8*)p;
  } else {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bf16_t hv = (bf16_t)f[i];
      hi[i] = hv;
      if (NPL == 2) lo[i] = (bf16_t)(f[i] - (float)hv);
    }
  }
}

// acc += W . x with W = (wh, wl) the A operand and x = (xh, xl) the B operand
template <int P>
__device__ __forceinline__ f32x4 mm(const bf16x8& wh, const bf16x8& wl, const bf16x8& xh, const bf16x8& xl, f32x4 acc) {
  if (P == 3) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc, 0, 0, 0);
}

// NR values of one batch row -> the operand image(s), 4-byte aligned runs
template <int NR, int NPL>
__device__ __forceinline__ void put_image(bf16_t* img, int plane_stride, int off, const float (&v)[NR]) {
  bf16_t hi[NR], lo[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) { hi[i] = (bf16_t)v[i]; lo[i] = (bf16_t)(v[i] - (float)hi[i]); }
  if constexpr (NR == 4) {
    *(uint2*)(img + off) = *(const uint2*)hi;
    if (NPL == 2) *(uint2*)(img + plane_stride + off) = *(const uint2*)lo;
  } else {
    *(unsigned*)(img + off) = *(const unsigned*)hi;
    if (NPL == 2) *(unsigned*)(img + plane_stride + off) = *(const unsigned*)lo;
  }
}

// A granule = {step tag, bf16 hi | bf16 lo << 16}: the publishing lane splits its values, the pollers copy halves into the
// operand images without touching the VALU.  Layout of an exchange buffer (16 rows x W values): LANE-LINEAR in the
// publishers' order - [16-unit block][K half kh][lane = (row, unit quad)][2 granules] - so that a compute wave's publish is
// ONE 16-byte write-through store per lane and 1 KB of consecutive bytes per wave (with [row][unit] order a wave wrote
// 64 scattered 8-byte pieces per instruction, two instructions per exchange: the store path, not the fabric, set the
// publish time).  `slot` = (block * 2 + kh) * 64 + lane.
typedef __attribute__((ext_vector_type(4))) unsigned gru_u32x4;
__device__ __forceinline__ void publish2(u64* buf, int nslots, int slot, unsigned tag, const float (&v)[2]) {
  unsigned pay[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const bf16_t hi = (bf16_t)v[i], lo = (bf16_t)(v[i] - (float)hi);
    pay[i] = (unsigned)(*(const unsigned short*)&hi) | ((unsigned)(*(const unsigned short*)&lo) << 16);
  }
  const uintptr_t base = (uintptr_t)buf;
  const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)bhi << 32) | blo), 0, nslots * 16, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128((gru_u32x4){pay[0], tag, pay[1], tag}, rs, (unsigned)slot * 16u, 0, 16);      // aux 16 = sc1
}
template <int NR> __device__ __forceinline__ void publish2(u64*, int, int, unsigned, const float (&)[NR]) {}      // (NR = 4: one workgroup per chain, nothing is published)

template <typename T, int NR> __device__ __forceinline__ void store_vals(T* p, const float (&v)[NR]) {
  if constexpr (sizeof(T) == 4) {
    if constexpr (NR == 4) *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    else *(float2*)p = make_float2(v[0], v[1]);
  } else {
    bf16_t b[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) b[i] = (bf16_t)v[i];
    if constexpr (NR == 4) *(uint2*)p = *(const uint2*)b;
    else *(unsigned*)p = *(const unsigned*)b;
  }
}

// The poller waves' gather of one exchange buffer (16 rows x W values in the publishers' lane-linear order, see publish2)
// into a [16][ILD] operand image.  Two forms, both with every load of a pass in flight (a pass is one memory round trip)
// and every image offset = the lane's base + a compile-time constant:
//   gather   16-byte L2-bypassing buffer loads = one publishing lane's two granules (the 8-byte halves arrive untorn, both
//            tags are checked), one 4-byte LDS write per plane;
//   gather8  8-byte loads, one granule each, 2-byte LDS writes.
// Poller pw takes the second half of the slots.  Returns false when the wave gave up (status raised / wall-clock bound).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int W, int NPL, int ILD = W + 8>       // ILD: row stride of the image (a gather may fill a column range of a wider one)
__device__ __forceinline__ bool gather(const u64* src_, unsigned tag, bf16_t* img, int pw, int lane, int* status) {
  constexpr int NLD = W / 16;                    // 16-byte loads per lane: (W / 16 blocks x 2 halves x 64 lanes) / 2 pollers / 64
  const uintptr_t base = (uintptr_t)(src_ + (size_t)pw * NLD * 128);
  const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)bhi << 32) | blo), 0, NLD * 1024, 0x00020000);
  // slot jj = pw * NLD + j of lane l: row l & 15, units 16 (jj >> 1) + 4 (l >> 4) + 2 (jj & 1) + {0, 1}
  unsigned* dst = (unsigned*)(img + (lane & 15) * ILD + 4 * (lane >> 4) + 16 * ((pw * NLD) >> 1));
  bool gave_up = false;
  u32x4 v[NLD];
  unsigned spins = 0, clk0 = 0;
  bool ok;
  do {
    ok = true;
#pragma unroll
    for (int j = 0; j < NLD; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)lane * 16u, j * 1024, 16);
#pragma unroll
    for (int j = 0; j < NLD; ++j) ok = ok && v[j][1] == tag && v[j][3] == tag;
    if (!ok && (++spins & 1023u) == 0) {
      if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = gave_up = true;
      else if (ns_spin_timed_out(clk0)) { atomicExch(status, 1); ok = gave_up = true; }
    }
  } while (!ok);
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int o = (16 * (j >> 1) + 2 * (j & 1)) / 2;                  // in 4-byte units (NLD is even: pw * NLD is too)
    const unsigned p0 = v[j][0], p1 = v[j][2];
    dst[o] = (p0 & 0xffffu) | (p1 << 16);
    if (NPL == 2) dst[16 * ILD / 2 + o] = (p0 >> 16) | (p1 & 0xffff0000u);
  }
  return !gave_up;
}

template <int W, int NPL, int ILD = W + 8>
__device__ __forceinline__ bool gather8(const u64* src_, unsigned tag, bf16_t* img, int pw, int lane, int* status) {
  constexpr int PPG = W / 8;                     // granules per lane: 16 W / 128
  const NS_GLOBAL char* src = (const NS_GLOBAL char*)((const NS_GLOBAL u64*)src_ + pw * PPG * 64);
  const unsigned lo8 = (unsigned)lane * 8u;
  // granule jj * 64 + l (jj = pw * PPG + j): row (l / 2) & 15, unit 16 (jj >> 2) + 8 (jj & 1) + 4 (l >> 5) + 2 ((jj >> 1) & 1) + (l & 1)
  unsigned short* dst = (unsigned short*)img + ((lane >> 1) & 15) * ILD + 4 * (lane >> 5) + (lane & 1) + 16 * ((pw * PPG) >> 2);
  bool gave_up = false;
  u64 v[PPG];
  unsigned spins = 0, clk0 = 0;
  bool ok;
  unsigned lo = lo8;
  asm volatile("" : "+v"(lo));          // opaque: the addresses are formed here, not hoisted out of the step loop as 64-bit pairs
  do {
    ok = true;
#pragma unroll
    for (int j = 0; j < PPG; ++j)
      v[j] = __hip_atomic_load((const NS_GLOBAL u64*)(src + j * 512 + lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int j = 0; j < PPG; ++j) ok = ok && ((unsigned)(v[j] >> 32) == tag);
    if (!ok && (++spins & 1023u) == 0) {
      if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = gave_up = true;
      else if (ns_spin_timed_out(clk0)) { atomicExch(status, 1); ok = gave_up = true; }
    }
  } while (!ok);
#pragma unroll
  for (int j = 0; j < PPG; ++j) {
    const int o = 16 * (j >> 2) + 8 * (j & 1) + 2 * ((j >> 1) & 1);   // (PPG is a multiple of 4: pw * PPG is too)
    const unsigned pay = (unsigned)v[j];
    dst[o] = (unsigned short)pay;
    if (NPL == 2) dst[16 * ILD + o] = (unsigned short)(pay >> 16);
  }
  return !gave_up;
}

// Which form: measured on one box, A/B in one process (us per step, H = 256, three passes / one pass):
//   forward   8-byte loads 6.21 / 5.69    16-byte pairs 8.55 / 6.90
//   backward  8-byte loads 8.44 / 7.95    16-byte pairs 7.08 / 6.95
// so the forward kernel polls single granules and the backward kernel pairs; NS_GRU_DBG bit 4 swaps them.
template <int W, int NPL, int ILD, bool PAIRS>
__device__ __forceinline__ bool gather_sel(int dbg, const u64* src, unsigned tag, bf16_t* img, int pw, int lane, int* status) {
  return (PAIRS != ((dbg & 4) != 0)) ? gather<W, NPL, ILD>(src, tag, img, pw, lane, status)
                                     : gather8<W, NPL, ILD>(src, tag, img, pw, lane, status);
}

// ===================================================================================================== forward
template <int H, int P, typename T, bool TRACE = false>
__global__ __launch_bounds__(GCfg<H>::FW_WAVES * 64) void gru_fwd_kernel(GruArgs a) {
  using C = GCfg<H>;
  constexpr int G = C::G, U = C::U, NB = C::NB, KSPLIT = C::KSPLIT, KSW = C::KSW, NR = C::NR, XS_F = C::XS_F, SECF = C::SECF;
  constexpr int NPL = P == 3 ? 2 : 1, PS = 16 * (H + 8);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* himg = (bf16_t*)smem;                       // [NPL][16][H] the state, swizzled
  bf16_t* rhimg = himg + NPL * PS;                    // [NPL][16][H] r * h
  float* xs = (float*)(rhimg + NPL * PS);             // [NBUF][XS_F] stage, filled by LDS-DMA
  float* pbuf = xs + C::NBUF * XS_F;                  // [C::GW][64][4] (K split)
  int* abortf = (int*)(pbuf + (KSPLIT == 2 ? C::GW * 64 * 4 : 0));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chain = blockIdx.x / G, g = blockIdx.x % G;
  const int d = chain / a.nrg, rg = chain % a.nrg;
  const bool rev = a.reverse[d] != 0;
  const int T_ = a.T;
  const int col = lane & 15, q4 = lane >> 4;
  u64* xr = a.xbuf + (size_t)chain * 2 * 16 * H;      // r * h granules [16][H]
  u64* xh = xr + 16 * H;                              // h granules
  if (tid == 0) abortf[0] = 0;
  for (int i = tid; i < 16 * H; i += C::FW_WAVES * 64) {      // the initial state as the first operand image
    const int row = i / H, k = i % H, n = rg * 16 + row;
    const float v = (a.h_init[d] && n < a.N) ? a.h_init[d][(long)n * a.ld_hi + k] : 0.f;
    const bf16_t hi = (bf16_t)v;
    himg[swz(row, k, H)] = hi;
    if (NPL == 2) himg[PS + swz(row, k, H)] = (bf16_t)(v - (float)hi);
  }

  if (wave < C::GW) {
    // ================================================================ compute role
    const int b = wave % NB, kh = wave / NB;
    const int ul0 = b * 16, ug0 = g * U + ul0;
    const int mo = KSPLIT == 2 ? 2 * kh : 0;               // first of the wave's own registers (units) per lane
    const int xoff = ((b * 4 + q4) * 16 + col) * 4;        // this lane's float4 inside a stage section
    bf16x8 wr[NPL][KSW], wu[NPL][KSW], wc[NPL][KSW];
    {
      const T* gT = (const T*)a.wgT[d];
      const T* cT = (const T*)a.wcT[d];
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) {
        const int k = (kh * KSW + ks) * 32 + q4 * 8;
        load_frag<T, NPL>(gT + (long)(ug0 + col) * H + k, wr[0][ks], wr[NPL - 1][ks]);
        load_frag<T, NPL>(gT + (long)(H + ug0 + col) * H + k, wu[0][ks], wu[NPL - 1][ks]);
        load_frag<T, NPL>(cT + (long)(ug0 + col) * H + k, wc[0][ks], wc[NPL - 1][ks]);
      }
    }
    const int n = rg * 16 + col;
    const bool nvalid = n < a.N;
    const int len = (a.lengths && nvalid) ? a.lengths[n] : T_;
    const int un = ug0 + 4 * q4 + mo;                       // first of this lane's own units
    float hst[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) hst[i] = (a.h_init[d] && nvalid) ? a.h_init[d][(long)n * a.ld_hi + un + i] : 0.f;
    T* hout = (T*)a.h[d];
    T* rhout = (T*)a.rh[d];
    long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = TRACE ? wall_clock64() : 0;
#define GRU_STAMP(k) do { if (TRACE) { __builtin_amdgcn_sched_barrier(0); const long long tn = wall_clock64(); tsum[k] += tn - tprev; tprev = tn; __builtin_amdgcn_sched_barrier(0); } } while (0)
    for (int s = 0; s < T_; ++s) {
      const int t = rev ? T_ - 1 - s : s, buf = s % C::NBUF;
      const unsigned rowi = (unsigned)(n * a.P + a.padl + t);
      wg_barrier();                                        // B0: state image and stage of this slot are complete
      if (abortf[0]) return;
      GRU_STAMP(0);
      const float* xrow = xs + buf * XS_F + xoff;
      f32x4 ar = {0.f, 0.f, 0.f, 0.f}, au = ar;
      if (kh == 0) { ar = *(const f32x4*)xrow; au = *(const f32x4*)(xrow + SECF); }
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) {
        const int so = swz(col, (kh * KSW + ks) * 32 + q4 * 8, H);
        const bf16x8 xh_ = *(const bf16x8*)(himg + so);
        const bf16x8 xl_ = NPL == 2 ? *(const bf16x8*)(himg + PS + so) : xh_;
        ar = mm<P>(wr[0][ks], wr[NPL - 1][ks], xh_, xl_, ar);
        au = mm<P>(wu[0][ks], wu[NPL - 1][ks], xh_, xl_, au);
      }
      if (TRACE) { asm volatile("" : "+v"(ar), "+v"(au)); }
      GRU_STAMP(1);
      float pr[NR], pu[NR];
      if constexpr (KSPLIT == 2) {
        // the partner wave (the other half of K) gets the sums of ITS registers, this wave takes the partner's for its own
        const float4 give = kh ? make_float4(ar[0], ar[1], au[0], au[1]) : make_float4(ar[2], ar[3], au[2], au[3]);
        *(float4*)(pbuf + (wave * 64 + lane) * 4) = give;
        wg_barrier();                                      // Bp1
        const float4 take = *(const float4*)(pbuf + (((wave + NB) % C::GW) * 64 + lane) * 4);
        pr[0] = (kh ? ar[2] : ar[0]) + take.x; pr[1] = (kh ? ar[3] : ar[1]) + take.y;
        pu[0] = (kh ? au[2] : au[0]) + take.z; pu[1] = (kh ? au[3] : au[1]) + take.w;
      } else {
#pragma unroll
        for (int i = 0; i < NR; ++i) { pr[i] = ar[i]; pu[i] = au[i]; }
      }
      float r[NR], u[NR], rhv[NR];
#pragma unroll
      for (int i = 0; i < NR; ++i) { r[i] = sigmoidf_(pr[i]); u[i] = sigmoidf_(pu[i]); rhv[i] = r[i] * hst[i]; }
      if constexpr (G == 1) put_image<NR, NPL>(rhimg, PS, swz(col, un, H), rhv);
      else publish2(xr, 16 * H / 2, ((g * NB + b) * KSPLIT + kh) * 64 + lane, (unsigned)(s + 1), rhv);
      if (!(a.dbg & 2)) {
        // the saved gates go out in the BACKWARD stage's layout (see ns_gru_seq_params.ru): 16 bytes per lane, 1 KB of
        // consecutive bytes per wave, every row (rows past N hold finite values nobody uses)
        float* sv = a.ru[d] + ((size_t)(rg * T_ + t) * G + g) * (2 * SECF) + xoff + mo;
        store_vals<float, NR>(sv, r);
        store_vals<float, NR>(sv + SECF, u);
        if (nvalid) store_vals<T, NR>(rhout + rowi * (unsigned)H + un, rhv);
      }
      GRU_STAMP(2);
      wg_barrier();                                        // B1: r * h image complete
      if (abortf[0]) return;
      GRU_STAMP(3);
      f32x4 ac = {0.f, 0.f, 0.f, 0.f};
      if (kh == 0) ac = *(const f32x4*)(xrow + 2 * SECF);
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) {
        const int so = swz(col, (kh * KSW + ks) * 32 + q4 * 8, H);
        const bf16x8 xh_ = *(const bf16x8*)(rhimg + so);
        const bf16x8 xl_ = NPL == 2 ? *(const bf16x8*)(rhimg + PS + so) : xh_;
        ac = mm<P>(wc[0][ks], wc[NPL - 1][ks], xh_, xl_, ac);
      }
      if (TRACE) { asm volatile("" : "+v"(ac)); }
      GRU_STAMP(4);
      float pc[NR];
      if constexpr (KSPLIT == 2) {
        const float2 give = kh ? make_float2(ac[0], ac[1]) : make_float2(ac[2], ac[3]);
        *(float2*)(pbuf + (wave * 64 + lane) * 4) = give;
        wg_barrier();                                      // Bp2
        const float2 take = *(const float2*)(pbuf + (((wave + NB) % C::GW) * 64 + lane) * 4);
        pc[0] = (kh ? ac[2] : ac[0]) + take.x; pc[1] = (kh ? ac[3] : ac[1]) + take.y;
      } else {
#pragma unroll
        for (int i = 0; i < NR; ++i) pc[i] = ac[i];
      }
      const bool masked = t >= len;
      float cv[NR], ho[NR];
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        cv[i] = tanhf_(pc[i]);
        const float hn = u[i] * hst[i] + (1.f - u[i]) * cv[i];
        ho[i] = masked ? 0.f : hn;                         // outputs are zero past the length,
        hst[i] = masked ? hst[i] : hn;                     // the state is carried through (tf.nn.dynamic_rnn)
      }
      if (s + 1 < T_) {
        if constexpr (G == 1) put_image<NR, NPL>(himg, PS, swz(col, un, H), hst);
        else publish2(xh, 16 * H / 2, ((g * NB + b) * KSPLIT + kh) * 64 + lane, (unsigned)(s + 1), hst);
      }
      if (!(a.dbg & 2)) {
        store_vals<float, NR>(a.c[d] + ((size_t)(rg * T_ + t) * G + g) * SECF + xoff + mo, cv);
        if (nvalid) store_vals<T, NR>(hout + rowi * (unsigned)a.ld_h + un, ho);
      }
      GRU_STAMP(5);
    }
#undef GRU_STAMP
    if (TRACE && a.trace && blockIdx.x == 0 && lane == 0)
      for (int k = 0; k < 8; ++k) a.trace[wave * 8 + k] = tsum[k];
  } else if (wave == C::GW) {
    // ================================================================ prefetcher role: LDS-DMA, two slots ahead
    // Uniform parts of every address stay in scalar registers; a lane keeps two 32-bit offsets (xg and xc rows).  Rows past
    // N fetch row N - 1 again (their results are never stored): every slot issues exactly FW_NI instructions, which is
    // what the counted vmcnt waits below rely on.
    constexpr int NI = C::FW_NI, IPS = (U / 4) / 4;        // instructions per slot / per section
    const int lrow = lane & 15, lq = lane >> 4;
    const int nrow = min(rg * 16 + lrow, a.N - 1);
    const unsigned vo_g = (unsigned)(nrow * a.P) * (unsigned)a.ld_xg + lq * 4, vo_c = (unsigned)(nrow * a.P) * (unsigned)a.ld_xc + lq * 4;
    auto issue = [&](int s) {
      const int t = rev ? T_ - 1 - s : s;
      const float* bg = a.xg[d] + (size_t)(a.padl + t) * a.ld_xg + g * U;
      const float* bc = a.xc[d] + (size_t)(a.padl + t) * a.ld_xc + g * U;
      float* dst = xs + (s % C::NBUF) * XS_F;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int sec = j / IPS, jc = (j % IPS) * 16;      // section, first float of the instruction's 4 chunks
        const float* src = sec < 2 ? bg + sec * H + jc + vo_g : bc + jc + vo_c;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + j * 256), 16, 0, 0);
      }
    };
    issue(0);
    if (T_ > 1) { issue(1); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NI) : "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int s = 0; s < T_; ++s) {
      wg_barrier();                                        // B0: slot s handed over
      if (abortf[0]) return;
      const bool more = s + 2 < T_ && !(a.dbg & 1);
      if (more) issue(s + 2);                              // its buffer was last read in slot s - 1
      if (KSPLIT == 2) wg_barrier();                       // Bp1
      wg_barrier();                                        // B1
      if (abortf[0]) return;
      if (KSPLIT == 2) wg_barrier();                       // Bp2
      if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NI) : "memory");     // slot s + 1 has landed
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
    // ================================================================ poller role (H = 256)
    if constexpr (G > 1) {
      const int pw = wave - C::GW - 1;
      for (int s = 0; s < T_; ++s) {
        if (s > 0 && !gather_sel<H, NPL, H + 8, false>(a.dbg, xh, (unsigned)s, himg, pw, lane, a.status)) abortf[0] = 1;
        wg_barrier();                                      // B0
        if (abortf[0]) return;
        if (KSPLIT == 2) wg_barrier();                     // Bp1: the own compute waves publish behind it
        if (!gather_sel<H, NPL, H + 8, false>(a.dbg, xr, (unsigned)(s + 1), rhimg, pw, lane, a.status)) abortf[0] = 1;
        wg_barrier();                                      // B1
        if (abortf[0]) return;
        if (KSPLIT == 2) wg_barrier();                     // Bp2
      }
    }
  }
}

// ===================================================================================================== backward
// Walks the steps of the forward pass in reverse.  carry = the gradient wrt the state that leaves step t towards the
// step processed next.  Per step, with dh = masked ? 0 : carry + dh_out[t]:
//   dzc = dh (1 - u)(1 - c^2)     dzu = dh (h_prev - c) u (1 - u)     carry' = masked ? carry : dh u
//   drh = dzc . Wc_h^T            dzr = drh h_prev r (1 - r)          carry' += drh r
//   carry' += [dzr | dzu] . Wg_h^T
// Two dependent products again; the weights are rows of the ORIGINAL kernels (A[row = unit k][j] = W[k][j]).
template <int H, int P, typename T>
__global__ __launch_bounds__(GCfg<H>::BW_WAVES * 64) void gru_bwd_kernel(GruArgs a) {
  using C = GCfg<H>;
  constexpr int G = C::G, U = C::U, NB = C::NB, KSPLIT = C::KSPLIT, KSW = C::KSW, NR = C::NR, SECF = C::SECF;
  // backward slot: r | u | c | dh sections (fp32) + h_prev in the history's own type (bf16: half a section)
  constexpr int HPF = sizeof(T) == 4 ? SECF : SECF / 2, ST_F = 4 * SECF + HPF, NI = ST_F / 256, NIW = NI / 2;
  constexpr int NPL = P == 3 ? 2 : 1, PSC = 16 * (H + 8), PSG = 16 * (2 * H + 8);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* dzcimg = (bf16_t*)smem;                     // [NPL][16][H]
  bf16_t* dzgimg = dzcimg + NPL * PSC;                // [NPL][16][2H]
  float* st = (float*)(dzgimg + NPL * PSG);           // [NBUF][ST_F] stage, filled by LDS-DMA
  float* pbuf = st + C::NBUF * ST_F;                  // [C::GW][64][2] (K split)
  int* abortf = (int*)(pbuf + (KSPLIT == 2 ? C::GW * 64 * 2 : 0));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chain = blockIdx.x / G, g = blockIdx.x % G;
  const int d = chain / a.nrg, rg = chain % a.nrg;
  const bool rev = a.reverse[d] != 0;
  const int T_ = a.T;
  const int col = lane & 15, q4 = lane >> 4;
  u64* xc_ = a.xbuf + (size_t)chain * 3 * 16 * H;     // dzc granules [16][H]
  u64* xg_ = xc_ + 16 * H;                            // dzr granules [16][H], then dzu granules [16][H]
  if (tid == 0) abortf[0] = 0;

  if (wave < C::GW) {
    // ================================================================ compute role
    const int b = wave % NB, kh = wave / NB;
    const int ul0 = b * 16, ug0 = g * U + ul0;
    const int mo = KSPLIT == 2 ? 2 * kh : 0;
    bf16x8 wc[NPL][KSW], wg[NPL][2 * KSW];
    {
      const T* Wc = (const T*)a.wc[d];
      const T* Wg = (const T*)a.wg[d];
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks)
        load_frag<T, NPL>(Wc + (long)(ug0 + col) * a.ld_wc + (kh * KSW + ks) * 32 + q4 * 8, wc[0][ks], wc[NPL - 1][ks]);
#pragma unroll
      for (int ks = 0; ks < 2 * KSW; ++ks)
        load_frag<T, NPL>(Wg + (long)(ug0 + col) * a.ld_wg + (kh * 2 * KSW + ks) * 32 + q4 * 8, wg[0][ks], wg[NPL - 1][ks]);
    }
    const int n = rg * 16 + col;
    const bool nvalid = n < a.N;
    const int len = (a.lengths && nvalid) ? a.lengths[n] : T_;
    const int un = ug0 + 4 * q4 + mo;
    float carry[NR], hini[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      carry[i] = 0.f;
      hini[i] = (a.h_init[d] && nvalid) ? a.h_init[d][(long)n * a.ld_hi + un + i] : 0.f;
    }
    const int xoff = ((b * 4 + q4) * 16 + col) * 4 + mo;   // this lane's values inside an fp32 stage section
    const int hoff = sizeof(T) == 4 ? xoff : ((b * 2 + (q4 >> 1)) * 16 + col) * 8 + (q4 & 1) * 4 + mo;     // ... the h_prev section
    T* dzg = (T*)a.dzg[d];
    T* dzc = (T*)a.dzc[d];
    wg_barrier();                                          // stage of the first slot
    for (int s = 0; s < T_; ++s) {
      const int t = rev ? s : T_ - 1 - s, buf = s % C::NBUF;      // the forward pass' steps, last first
      const int tp = rev ? t + 1 : t - 1;                  // the step whose output is this step's h_prev
      const unsigned rowi = (unsigned)(n * a.P + a.padl + t);
      const float* srow = st + buf * ST_F + xoff;
      const T* hrow = (const T*)(st + buf * ST_F + 4 * SECF) + hoff;
      const bool masked = t >= len;
      // h_prev: the initial state at a sequence's first step (t = 0; reversed: t = len - 1), zero outside the sequence
      const bool first = rev ? t == len - 1 : t == 0, tp_ok = tp >= 0 && tp < T_;
      float r[NR], hp[NR], dzcv[NR], dzuv[NR], cn[NR];
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        r[i] = srow[i];
        const float u = srow[SECF + i], c = srow[2 * SECF + i];
        hp[i] = first ? hini[i] : (tp_ok ? ldf(hrow + i) : 0.f);
        const float dh = masked ? 0.f : carry[i] + srow[3 * SECF + i];
        dzcv[i] = dh * (1.f - u) * (1.f - c * c);
        dzuv[i] = dh * (hp[i] - c) * u * (1.f - u);
        cn[i] = masked ? carry[i] : dh * u;
      }
      if constexpr (G == 1) put_image<NR, NPL>(dzcimg, PSC, swz(col, un, H), dzcv);
      else {
        publish2(xc_, 16 * H / 2, ((g * NB + b) * KSPLIT + kh) * 64 + lane, (unsigned)(s + 1), dzcv);
        publish2(xg_ + 16 * H, 16 * H / 2, ((g * NB + b) * KSPLIT + kh) * 64 + lane, (unsigned)(s + 1), dzuv);       // not needed before the second product: gathered off the chain
      }
      if (nvalid) {
        store_vals<T, NR>(dzc + rowi * (unsigned)H + un, dzcv);
        store_vals<T, NR>(dzg + rowi * (unsigned)(2 * H) + H + un, dzuv);
      }
      wg_barrier();                                        // B1: dzc image complete
      if (abortf[0]) return;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) {
        const int so = swz(col, (kh * KSW + ks) * 32 + q4 * 8, H);
        const bf16x8 xh_ = *(const bf16x8*)(dzcimg + so);
        const bf16x8 xl_ = NPL == 2 ? *(const bf16x8*)(dzcimg + PSC + so) : xh_;
        acc = mm<P>(wc[0][ks], wc[NPL - 1][ks], xh_, xl_, acc);
      }
      float drh[NR];
      if constexpr (KSPLIT == 2) {
        const float2 give = kh ? make_float2(acc[0], acc[1]) : make_float2(acc[2], acc[3]);
        *(float2*)(pbuf + (wave * 64 + lane) * 2) = give;
        wg_barrier();                                      // Bpa
        const float2 take = *(const float2*)(pbuf + (((wave + NB) % C::GW) * 64 + lane) * 2);
        drh[0] = (kh ? acc[2] : acc[0]) + take.x; drh[1] = (kh ? acc[3] : acc[1]) + take.y;
      } else {
#pragma unroll
        for (int i = 0; i < NR; ++i) drh[i] = acc[i];
      }
      float dzrv[NR];
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        dzrv[i] = drh[i] * hp[i] * r[i] * (1.f - r[i]);
        cn[i] += drh[i] * r[i];
      }
      if constexpr (G == 1) {
        put_image<NR, NPL>(dzgimg, PSG, swz(col, un, 2 * H), dzrv);
        put_image<NR, NPL>(dzgimg, PSG, swz(col, H + un, 2 * H), dzuv);
      } else {
        publish2(xg_, 16 * H / 2, ((g * NB + b) * KSPLIT + kh) * 64 + lane, (unsigned)(s + 1), dzrv);
      }
      if (nvalid) store_vals<T, NR>(dzg + rowi * (unsigned)(2 * H) + un, dzrv);
      wg_barrier();                                        // B2: dzg image complete (and the next slot's stage)
      if (abortf[0]) return;
      acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2 * KSW; ++ks) {
        const int so = swz(col, (kh * 2 * KSW + ks) * 32 + q4 * 8, 2 * H);
        const bf16x8 xh_ = *(const bf16x8*)(dzgimg + so);
        const bf16x8 xl_ = NPL == 2 ? *(const bf16x8*)(dzgimg + PSG + so) : xh_;
        acc = mm<P>(wg[0][ks], wg[NPL - 1][ks], xh_, xl_, acc);
      }
      if constexpr (KSPLIT == 2) {
        const float2 give = kh ? make_float2(acc[0], acc[1]) : make_float2(acc[2], acc[3]);
        *(float2*)(pbuf + (wave * 64 + lane) * 2) = give;
        wg_barrier();                                      // Bpb
        const float2 take = *(const float2*)(pbuf + (((wave + NB) % C::GW) * 64 + lane) * 2);
        carry[0] = cn[0] + (kh ? acc[2] : acc[0]) + take.x;
        carry[1] = cn[1] + (kh ? acc[3] : acc[1]) + take.y;
      } else {
#pragma unroll
        for (int i = 0; i < NR; ++i) carry[i] = cn[i] + acc[i];
      }
    }
    if (a.dh_init[d] && nvalid) store_vals<float, NR>(a.dh_init[d] + (long)n * a.ld_dhi + un, carry);
  } else if (wave < C::GW + 2) {
    // ================================================================ prefetcher role (two waves): LDS-DMA, two slots ahead
    // Instruction j of a slot: j < 4 * IPS -> section j / IPS (r, u from ru; c; dh), else the h_prev rows of the history
    // (type T: 8 bf16 or 4 fp32 per 16-byte chunk).  Wave pw issues the instructions j = pw, pw + 2, ...; rows past N and a
    // h_prev step outside the sequence fetch a valid row instead (never used): the count per slot is constant.
    constexpr int IPS = (U / 4) / 4, EPC = 16 / (int)sizeof(T);
    const int pw = wave - C::GW;
    const T* hist = (const T*)a.h[d];
    const int lrow = lane & 15, lq = lane >> 4;
    const int nrow = min(rg * 16 + lrow, a.N - 1);
    const unsigned vo_dh = (unsigned)(nrow * a.P) * (unsigned)a.ld_dh + lq * 4, vo_h = (unsigned)(nrow * a.P) * (unsigned)a.ld_h + lq * EPC;
    auto issue = [&](int s) {
      const int t = rev ? s : T_ - 1 - s;
      const int tp = min(max(rev ? t + 1 : t - 1, 0), T_ - 1);
      const size_t r0 = (size_t)(a.padl + t);
      // the saved gates lie in this stage's own order (the forward kernel wrote them so): a straight lane-linear copy
      const float* bru = a.ru[d] + ((size_t)(rg * T_ + t) * G + g) * (2 * SECF) + lane * 4;
      const float* bc = a.c[d] + ((size_t)(rg * T_ + t) * G + g) * SECF + lane * 4;
      const float* bdh = a.dh[d] + r0 * a.ld_dh + g * U;
      const T* bh = hist + (size_t)(a.padl + tp) * a.ld_h + g * U;
      float* dst = st + (s % C::NBUF) * ST_F;
#pragma unroll
      for (int jj = 0; jj < (NI + 1) / 2; ++jj) {
        const int j = 2 * jj + pw;                         // pw is wave-uniform: both arms below are scalar branches
        if (j >= NI) break;
        const int sec = j / IPS, jc = (j % IPS) * 16;
        const void* src;
        if (sec < 2) src = bru + j * 256;
        else if (sec == 2) src = bc + (j - 2 * IPS) * 256;
        else if (sec == 3) src = bdh + jc + vo_dh;
        else src = bh + (j - 4 * IPS) * 4 * EPC + vo_h;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + j * 256), 16, 0, 0);
      }
    };
    // instructions this wave issues per slot (NI may be odd for bf16 histories)
    const int mine = (NI - pw + 1) / 2;
    auto wait_one_slot_left = [&]() {
      // all but the newest slot's loads: the count differs between the two waves only when NI is odd
      if (mine == (NI + 1) / 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NI + 1) / 2) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NI / 2) : "memory");
    };
    issue(0);
    if (T_ > 1) { issue(1); wait_one_slot_left(); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    for (int s = 0; s < T_; ++s) {
      const bool more = s + 2 < T_;
      if (more) issue(s + 2);                              // its buffer was last read in phase 1 of slot s - 1
      wg_barrier();                                        // B1
      if (abortf[0]) return;
      if (KSPLIT == 2) wg_barrier();                       // Bpa
      if (more) wait_one_slot_left(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // slot s + 1 has landed
      wg_barrier();                                        // B2: hands slot s + 1 over
      if (abortf[0]) return;
      if (KSPLIT == 2) wg_barrier();                       // Bpb
    }
  } else {
    // ================================================================ poller role (H = 256)
    if constexpr (G > 1) {
      const int pw = wave - C::GW - 2;
      wg_barrier();
      for (int s = 0; s < T_; ++s) {
        if (!gather_sel<H, NPL, H + 8, true>(a.dbg, xc_, (unsigned)(s + 1), dzcimg, pw, lane, a.status)) abortf[0] = 1;
        wg_barrier();                                      // B1
        if (abortf[0]) return;
        if (KSPLIT == 2) wg_barrier();                     // Bpa: the own compute waves publish dzr / dzu behind it
        // dzu went out in phase 1: this pass finds it at once, under the compute waves' first product; then dzr
        if (!gather_sel<H, NPL, 2 * H + 8, true>(a.dbg, xg_ + 16 * H, (unsigned)(s + 1), dzgimg + H, pw, lane, a.status)) abortf[0] = 1;
        if (!gather_sel<H, NPL, 2 * H + 8, true>(a.dbg, xg_, (unsigned)(s + 1), dzgimg, pw, lane, a.status)) abortf[0] = 1;
        wg_barrier();                                      // B2
        if (abortf[0]) return;
        if (KSPLIT == 2) wg_barrier();                     // Bpb
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
template <int H> size_t fwd_lds(int P) {
  using C = GCfg<H>;
  const int npl = P == 3 ? 2 : 1;
  return (size_t)2 * npl * 16 * (H + 8) * 2 + sizeof(float) * C::NBUF * C::XS_F + (C::KSPLIT == 2 ? sizeof(float) * C::GW * 64 * 4 : 0) + 16;
}
template <int H> size_t bwd_lds(int P, bool bf16_hist) {
  using C = GCfg<H>;
  const int npl = P == 3 ? 2 : 1;
  const size_t st_f = 4 * C::SECF + (bf16_hist ? C::SECF / 2 : C::SECF);
  return (size_t)npl * 16 * (3 * H + 16) * 2 + sizeof(float) * C::NBUF * st_f + (C::KSPLIT == 2 ? sizeof(float) * C::GW * 64 * 2 : 0) + 16;
}

int passes_of(const ns_gru_seq_params* p) { return p->dtype == NS_BF16 ? 1 : p->f32_passes; }

bool pair_ok(const ns_gru_seq_params* p0, const ns_gru_seq_params* p1) {
  if (!p1) return true;
  return p0->N == p1->N && p0->T == p1->T && p0->H == p1->H && p0->P == p1->P && p0->padl == p1->padl &&
         p0->dtype == p1->dtype && p0->f32_passes == p1->f32_passes && p0->lengths == p1->lengths &&
         p0->ld_xg == p1->ld_xg && p0->ld_xc == p1->ld_xc && p0->ld_h == p1->ld_h && p0->ld_hi == p1->ld_hi &&
         p0->ld_wg == p1->ld_wg && p0->ld_wc == p1->ld_wc && p0->ld_dh == p1->ld_dh && p0->ld_dhi == p1->ld_dhi;
}

bool one_ok(const ns_gru_seq_params* p, int backward) {
  auto al = [](const void* q, int by) { return ((uintptr_t)q % by) == 0; };
  if (!(p->H == 128 || p->H == 256) || p->N < 1 || p->T < 1) return false;
  if (!(p->dtype == NS_BF16 || (p->dtype == NS_F32 && (p->f32_passes == 1 || p->f32_passes == 3)))) return false;
  const int eb = p->dtype == NS_BF16 ? 2 : 4;        // element bytes of the dtype arrays; vector accesses are 4 elements
  const long widest = (long)p->N * p->P * (2L * p->H > p->ld_h ? 2L * p->H : p->ld_h);
  if (widest >= (1L << 31) || (long)p->N * p->P * p->ld_xg >= (1L << 31) || (long)p->N * p->P * p->ld_dh >= (1L << 31)) return false;
  if (!p->h || !p->ru || !p->c || !al(p->h, 4 * eb) || p->ld_h % 4 || !al(p->ru, 16) || !al(p->c, 16)) return false;
  if (p->h_init && (!al(p->h_init, 16) || p->ld_hi % 4)) return false;
  if (!backward) {
    if (!p->xg || !p->xc || !p->wgT || !p->wcT || !p->rh) return false;
    if (!al(p->xg, 16) || p->ld_xg % 4 || !al(p->xc, 16) || p->ld_xc % 4 || !al(p->wgT, 16) || !al(p->wcT, 16) || !al(p->rh, 4 * eb)) return false;
  } else {
    if (!p->wg || !p->wc || !p->dh || !p->dzg || !p->dzc) return false;
    if (!al(p->wg, 16) || !al(p->wc, 16) || (p->ld_wg * eb) % 16 || (p->ld_wc * eb) % 16) return false;
    if (!al(p->dh, 16) || p->ld_dh % 4 || !al(p->dzg, 4 * eb) || !al(p->dzc, 4 * eb)) return false;
    if (p->dh_init && (!al(p->dh_init, 16) || p->ld_dhi % 4)) return false;
  }
  return true;
}

size_t xbytes(const ns_gru_seq_params* p, int ndir) {
  const size_t chains = (size_t)ndir * ((p->N + 15) / 16);
  return p->H == 256 ? chains * 3 * 16 * (size_t)p->H * sizeof(u64) : 0;
}
void fill(GruArgs& a, const ns_gru_seq_params* p0, const ns_gru_seq_params* p1, void* work) {
  const ns_gru_seq_params* pp[2] = {p0, p1 ? p1 : p0};
  a.N = p0->N; a.T = p0->T; a.P = p0->P; a.padl = p0->padl; a.ndir = p1 ? 2 : 1; a.nrg = (p0->N + 15) / 16;
  a.lengths = p0->lengths;
  a.ld_xg = p0->ld_xg; a.ld_xc = p0->ld_xc; a.ld_wg = p0->ld_wg; a.ld_wc = p0->ld_wc; a.ld_h = p0->ld_h;
  a.ld_hi = p0->ld_hi; a.ld_dh = p0->ld_dh; a.ld_dhi = p0->ld_dhi;
  for (int d = 0; d < 2; ++d) {
    const ns_gru_seq_params* p = pp[d];
    a.reverse[d] = p->reverse;
    a.xg[d] = p->xg; a.xc[d] = p->xc; a.wgT[d] = p->wgT; a.wcT[d] = p->wcT; a.wg[d] = p->wg; a.wc[d] = p->wc;
    a.h[d] = p->h; a.ru[d] = p->ru; a.c[d] = p->c; a.rh[d] = p->rh; a.h_init[d] = p->h_init;
    a.dh[d] = p->dh; a.dzg[d] = p->dzg; a.dzc[d] = p->dzc; a.dh_init[d] = p->dh_init;
  }
  a.status = (int*)work;
  a.xbuf = (u64*)((char*)work + 256);
  const char* tr = getenv("NS_GRU_TRACE");
  a.trace = (tr && atoi(tr)) ? (long long*)((char*)work + 256 + xbytes(p0, 2)) : nullptr;
  const char* dbg = getenv("NS_GRU_DBG");
  a.dbg = dbg ? atoi(dbg) : 0;
}

}  // namespace

extern "C" int ns_gru_seq_supported(const ns_gru_seq_params* p0, const ns_gru_seq_params* p1, int backward) {
  if (!p0 || !one_ok(p0, backward) || (p1 && !one_ok(p1, backward)) || !pair_ok(p0, p1)) return 0;
  // every workgroup of a chain must be resident at once (H = 256: 4 per chain; chains are independent of one another)
  return (p0->H == 256 ? 4 : 1) <= ns_device_cus();
}

extern "C" size_t ns_gru_seq_work_bytes(const ns_gru_seq_params* p) {
  if (!p) return 0;
  return 256 + xbytes(p, 2) + 16 * 8 * sizeof(long long);      // status, exchange buffers, NS_GRU_TRACE sums
}

template <int H, int P, typename T>
static void launch_fwd(const GruArgs& a, hipStream_t s) {
  using C = GCfg<H>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gru_fwd_kernel<H, P, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (a.trace) {       // diagnostic instantiation with in-kernel stamps (never the production path)
    if constexpr (P == 3) {
      (void)hipFuncSetAttribute((const void*)gru_fwd_kernel<H, P, T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((gru_fwd_kernel<H, P, T, true>), dim3((unsigned)(a.ndir * a.nrg * C::G)), dim3(C::FW_WAVES * 64), fwd_lds<H>(P), s, a);
      return;
    }
  }
  hipLaunchKernelGGL((gru_fwd_kernel<H, P, T>), dim3((unsigned)(a.ndir * a.nrg * C::G)), dim3(C::FW_WAVES * 64), fwd_lds<H>(P), s, a);
}
template <int H, int P, typename T>
static void launch_bwd(const GruArgs& a, hipStream_t s) {
  using C = GCfg<H>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gru_bwd_kernel<H, P, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((gru_bwd_kernel<H, P, T>), dim3((unsigned)(a.ndir * a.nrg * C::G)), dim3(C::BW_WAVES * 64), bwd_lds<H>(P, sizeof(T) == 2), s, a);
}

static int gru_seq_run(const ns_gru_seq_params* p0, const ns_gru_seq_params* p1, void* work, hipStream_t s, int backward) {
  const char* name = backward ? "ns_gru_seq_bwd" : "ns_gru_seq_fwd";
  NS_CHECK_ARG(p0 && work, "%s: null", name);
  NS_CHECK_ARG(ns_gru_seq_supported(p0, p1, backward), "%s: needs H in {128, 256}, dtype bf16 or fp32 with f32_passes 1 / 3, "
               "16-byte aligned operand rows and two directions of equal shape", name);
  GruArgs a = {};
  fill(a, p0, p1, work);
  { const int zrc = ns_zero_async(work, (256 + xbytes(p0, a.ndir) + 15) & ~(size_t)15, s); if (zrc) return zrc; }
  const int P = passes_of(p0);
#define NS_GRU_LAUNCH(FN) \
  do { \
    if (p0->H == 128) { \
      if (p0->dtype == NS_BF16) FN<128, 1, bf16_t>(a, s); \
      else if (P == 3) FN<128, 3, float>(a, s); \
      else FN<128, 1, float>(a, s); \
    } else { \
      if (p0->dtype == NS_BF16) FN<256, 1, bf16_t>(a, s); \
      else if (P == 3) FN<256, 3, float>(a, s); \
      else FN<256, 1, float>(a, s); \
    } \
  } while (0)
  if (backward) NS_GRU_LAUNCH(launch_bwd); else NS_GRU_LAUNCH(launch_fwd);
#undef NS_GRU_LAUNCH
  NS_CHECK_LAUNCH(name);
  return NS_OK;
}

extern "C" int ns_gru_seq_fwd(const ns_gru_seq_params* p0, const ns_gru_seq_params* p1, void* work, ns_stream_t s) {
  return gru_seq_run(p0, p1, work, (hipStream_t)s, 0);
}
extern "C" int ns_gru_seq_bwd(const ns_gru_seq_params* p0, const ns_gru_seq_params* p1, void* work, ns_stream_t s) {
  return gru_seq_run(p0, p1, work, (hipStream_t)s, 1);
}
