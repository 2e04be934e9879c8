// Microbenchmark for the wide-LSTM question (DESIGN: decoder LSTM(1024) as a persistent kernel): how long does it take
// EVERY CU to read a vector that all CUs have just written (512 B each, 128 KB in all), inside one launch, per step,
// at FRESH addresses each step (a history array)?  Variants of the producer store / consumer load:
//   0: plain stores + agent release fence | flag counter | agent acquire + plain 16-B loads   (L2 may serve the readers)
//   1: sc1 (write-through) stores, drained | flag counter | sc1 16-B loads
//   2: sc1 stores, drained | flag counter | agent acquire + plain 16-B loads
// One 512-thread workgroup per CU (256), one monotonic arrival counter per step (relaxed sc1 polling).
// Prints per variant: mean microseconds per step of (publish -> all arrived) and of the 64 KB read that follows.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef unsigned int u32;
constexpr int NWG = 256, NT = 512, STEPS = 64;
constexpr int SLICE = 128;            // floats written per workgroup and step (512 B)
constexpr int VEC = NWG * SLICE;      // 32768 floats = 128 KB per step
constexpr int READ = VEC / 2;         // each workgroup reads one half (its "row group"): 64 KB

__global__ __launch_bounds__(NT) void k(float* hist, u32* counters, long long* tr, float* sink, int variant) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  __shared__ float lds[READ];
  float acc = 0.f;
  for (int s = 0; s < STEPS; ++s) {
    float* cur = hist + (size_t)s * VEC;
    // publish my slice
    if (tid < SLICE) {
      const float v = (float)(s * 1000 + wg) + tid * 1e-3f;
      if (variant == 0) cur[wg * SLICE + tid] = v;
      else __hip_atomic_store(cur + wg * SLICE + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    long long t0 = 0, t1 = 0, t2 = 0;
    if (tid == 0) {
      t0 = wall_clock64();
      if (variant == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __hip_atomic_fetch_add(counters + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(counters + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NWG && ++spins < 4000000u) __builtin_amdgcn_s_sleep(1);
      if (variant != 1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      t1 = wall_clock64();
    }
    __syncthreads();
    // read my half: 64 KB = 4096 x 16 B, 8 per thread, all in flight
    const float4* src = (const float4*)(cur + (wg & 1) * READ);
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (variant == 1) {
        const auto r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, READ * 4, 0x00020000);
        auto q = __builtin_amdgcn_raw_buffer_load_b128(r, (tid + j * NT) * 16, 0, 16);
        v[j] = *(float4*)&q;
      } else v[j] = src[tid + j * NT];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { ((float4*)lds)[tid + j * NT] = v[j]; acc += v[j].x + v[j].w; }
    __syncthreads();
    if (tid == 0) {
      t2 = wall_clock64();
      tr[((size_t)wg * STEPS + s) * 2 + 0] = t1 - t0;
      tr[((size_t)wg * STEPS + s) * 2 + 1] = t2 - t1;
      // check one value from the far end
      const float want = (float)(s * 1000 + ((wg & 1) * 128 + 127)) + 127 * 1e-3f;
      if (lds[READ - 1] != want) atomicAdd((u32*)(counters + STEPS), 1u);
    }
    __syncthreads();
  }
  if (acc == 12345.f) sink[wg] = acc;
}

int main() {
  float* hist; u32* counters; long long* tr; float* sink;
  hipMalloc(&hist, sizeof(float) * (size_t)STEPS * VEC);
  hipMalloc(&counters, sizeof(u32) * (STEPS + 16));
  hipMalloc(&tr, sizeof(long long) * NWG * STEPS * 2);
  hipMalloc(&sink, sizeof(float) * NWG);
  std::vector<long long> h(NWG * STEPS * 2);
  for (int variant = 0; variant < 3; ++variant) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(hist, 0, sizeof(float) * (size_t)STEPS * VEC);
      hipMemset(counters, 0, sizeof(u32) * (STEPS + 16));
      hipLaunchKernelGGL(k, dim3(NWG), dim3(NT), 0, 0, hist, counters, tr, sink, variant);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), tr, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    u32 bad = 0;
    hipMemcpy(&bad, counters + STEPS, 4, hipMemcpyDeviceToHost);
    double a = 0, b = 0, bmax = 0; int n = 0;
    for (int wg = 0; wg < NWG; ++wg)
      for (int s = 8; s < STEPS; ++s) { a += h[(wg * STEPS + s) * 2]; b += h[(wg * STEPS + s) * 2 + 1]; bmax = std::max(bmax, (double)h[(wg * STEPS + s) * 2 + 1]); ++n; }
    printf("variant %d: barrier %.2f us, 64 KB read %.2f us (max %.2f), stale checks %u\n", variant, a / n * 1e-2, b / n * 1e-2, bmax * 1e-2, bad);
  }
  return 0;
}
