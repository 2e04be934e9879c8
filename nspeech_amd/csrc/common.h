// Shared device/host helpers for libnspeech_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/nspeech_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---------------------------------------------------------------- errors
void ns_set_error(const char* fmt, ...);
int ns_zero_async(void* p, size_t bytes, hipStream_t s);   // core.hip: kernel fill (graph-replay safe, see there)
// core.hip: compute units of the CURRENT device (cached per device; 256 on a whole MI355X, fewer in a CPX / DPX partition).
// The persistent kernels need every workgroup of a launch resident at once, one per CU: their *_supported() checks
// compare the grid with this instead of a literal 256, so a smaller device takes the launch-per-step kernels.
int ns_device_cus();
// gemm.hip: s1[n] = sum over the slots of part[slot][n], s2[n] (nullable) = the same over part[slots + slot][n], fixed order
int ns_stats_finalize(const float* part, int slots, int N, float* s1, float* s2, hipStream_t s);

#define NS_CHECK_ARG(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) {                                                \
      ns_set_error(__VA_ARGS__);                                  \
      return NS_ERR_BAD_ARG;                                      \
    }                                                             \
  } while (0)

#define NS_CHECK_LAUNCH(name)                                     \
  do {                                                            \
    hipError_t e__ = hipGetLastError();                           \
    if (e__ != hipSuccess) {                                      \
      ns_set_error("%s: launch failed: %s", name,                 \
                   hipGetErrorString(e__));                       \
      return NS_ERR_LAUNCH;                                       \
    }                                                             \
  } while (0)

// ---------------------------------------------------------------- typed load/store
template <typename T> struct dt_of;
template <> struct dt_of<float> { static constexpr int v = NS_F32; };
template <> struct dt_of<bf16_t> { static constexpr int v = NS_BF16; };

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = (bf16_t)v; }

// v_exp_f32 + v_rcp_f32 forms (absolute error ~1e-7): sigmoid(x) = 1/(1+2^(-x log2 e)),
// tanh(x) = 1 - 2/(2^(2x log2 e) + 1); both saturate correctly through exp -> 0 / inf.
__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.8853900817779268f * x) + 1.0f);
}

// Hand-off of data between workgroups WITHOUT an agent-scope fence (a fence writes back / invalidates the whole per-XCD
// L2 - measured: the fixed-order split-K with __threadfence() doubled the weight-gradient products' time): every
// handed-off store is write-through (sc1) and drained (`ns_drain_stores`) in front of the arrival counter, every load of
// handed-off bytes bypasses the L2s (sc1).  MI355X_MICROARCH.md, "Correctness boundaries", second valid form.
__device__ __forceinline__ void ns_st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ns_ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ns_st_sc1(float4* p, const float4& v) {
  unsigned long long* q = (unsigned long long*)p;
  __hip_atomic_store(q, ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 ns_ld_sc1(const float4* p) {
  const unsigned long long* q = (const unsigned long long*)p;
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b),
                     __uint_as_float((unsigned)(b >> 32)));
}
__device__ __forceinline__ void ns_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Zoneout masks (include/nspeech_hip.h, ns_lstm_seq_params): true = the unit keeps its old value at this step.
__host__ __device__ __forceinline__ uint32_t ns_fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ bool ns_zone_keep(uint32_t seed, uint32_t t, uint32_t n, uint32_t u, uint32_t thr) {
  uint32_t x = ns_fmix32(seed ^ (t * 0x9E3779B9u));
  x = ns_fmix32(x ^ (n * 0x7FEB352Du));
  x = ns_fmix32(x ^ (u * 0x846CA68Bu));
  return (x >> 8) < thr;
}

// tf.contrib.rnn.LSTMBlockCell's cell_clip (lstm_ops: cs = clip(ci .* i + cs_prev .* f, -cell_clip, cell_clip) when the
// attribute is > 0; its gradient kernel applies no mask for clipped values, so the backward kernels stay as they are and
// read the clipped states the forward kernels saved).  clip <= 0: off - the reference's cells (modules.py:41-42,
// tacotron2.py:69-70 pass no cell_clip; whether TF 1.7's default clips is the [3P] question the switch exists for).
__device__ __forceinline__ float ns_cell_clip(float c, float clip) { return clip > 0.f ? fminf(fmaxf(c, -clip), clip) : c; }

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case NS_ACT_RELU: return v > 0.f ? v : 0.f;
    case NS_ACT_TANH: return tanhf_(v);
    case NS_ACT_SIGMOID: return sigmoidf_(v);
    case NS_ACT_SOFTSIGN: return v / (1.0f + fabsf(v));
    default: return v;
  }
}

// Cross-lane reductions: inside a DPP row (16 lanes) with VALU-rate DPP moves - quad swaps, then the two row mirrors
// leave the row's result in every lane - and only the last two steps (lane ^ 16, lane ^ 32) through ds_bpermute.
#define NS_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float row16_sum(float v) {      // sum over the 16 lanes that share lane >> 4
  v += NS_DPP_F(v, 0xB1);     // quad_perm [1,0,3,2]
  v += NS_DPP_F(v, 0x4E);     // quad_perm [2,3,0,1]
  v += NS_DPP_F(v, 0x141);    // row_half_mirror
  v += NS_DPP_F(v, 0x140);    // row_mirror
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, NS_DPP_F(v, 0xB1));
  v = fmaxf(v, NS_DPP_F(v, 0x4E));
  v = fmaxf(v, NS_DPP_F(v, 0x141));
  v = fmaxf(v, NS_DPP_F(v, 0x140));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// block-wide sum with 256..1024 threads; `red` is >= 32 floats of LDS scratch
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = -INFINITY;
  for (int i = 0; i < nw; ++i) s = fmaxf(s, red[i]);
  return s;
}

// ---------------------------------------------------------------- split-bf16 helpers
// fp32 value x = hi + lo (hi = bf16(x), lo = bf16(x - hi)); products hi*hi + hi*lo + lo*hi
__device__ __forceinline__ void ldsplit8(const float* p, bool ok, bf16x8& hi, bf16x8& lo) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (ok) { a = *(const float4*)p; b = *(const float4*)(p + 4); }
  const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bf16_t h = (bf16_t)f[i];
    hi[i] = h;
    lo[i] = (bf16_t)(f[i] - (float)h);
  }
}
template <int PASSES>
__device__ __forceinline__ f32x4 mfma_split(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl,
                                            f32x4 acc) {
  // PASSES 3: a.b ~ lo.hi + hi.lo + hi.hi;  2: b rounded to bf16, a exact to ~16 bits (lo.hi + hi.hi);  1: hi.hi
  if (PASSES > 1) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
  if (PASSES > 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}


// ns_rows32's packed activation rows (rows32.hip): value v of row n, column k as its (hi, lo) pair at its MFMA fragment
// position - [row tile n / 16][chunk k / 32][lane (k / 8 % 4) * 16 + n % 16][plane][k % 8]
__device__ __forceinline__ void ns_rows32_store(bf16_t* base, int nkc, int n, int k, float v) {
  const long i16 = (((long)(n >> 4) * nkc + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (n & 15)) * 2;
  const bf16_t h = (bf16_t)v;
  base[i16 * 8 + (k & 7)] = h;
  base[(i16 + 1) * 8 + (k & 7)] = (bf16_t)(v - (float)h);
}

// ---------------------------------------------------------------- bounded spins of the persistent kernels
// Every wait of a persistent kernel is bounded in WALL-CLOCK time, not in iterations: a workgroup that cannot be placed
// because a foreign kernel (a collective's channel kernel, a weight-gradient product on the second stream) holds its CU
// keeps its peers polling for as long as that kernel lives, and an iteration count says nothing about how long that is.
// The clock (s_memrealtime, 100 MHz) is read once per 1024 polls; 32 bits of it wrap after 42 s, far beyond the bound.
constexpr unsigned NS_SPIN_TICKS = 200000000u;      // 2 s
// Volatile accesses to LDS words that another wave of the workgroup polls / raises.  Through a GENERIC pointer a volatile
// access compiles to flat_load / flat_store ... sc0 sc1, and its s_waitcnt vmcnt(0) also waits for every vector-memory
// load the wave has in flight (the sweep of lstm_wide, the weight prefetches of the WaveNet chain); with the LDS address
// space spelled out it is a ds_read / ds_write that only touches lgkmcnt.
#define NS_GLOBAL __attribute__((address_space(1)))      // a pointer known to be global memory (pointers out of by-value
                                                        // argument structs are generic: flat_load / flat_store)
template <typename T> __device__ __forceinline__ T ns_lds_peek(const T* p) {
  return *(const volatile __attribute__((address_space(3))) T*)p;
}
template <typename T> __device__ __forceinline__ void ns_lds_poke(T* p, T v) {
  *(volatile __attribute__((address_space(3))) T*)p = v;
}
__device__ __forceinline__ bool ns_spin_timed_out(unsigned& t0) {
  const unsigned now = (unsigned)wall_clock64() | 1u;
  if (t0 == 0u) { t0 = now; return false; }
  return now - t0 > NS_SPIN_TICKS;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
