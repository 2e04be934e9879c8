// The exchange step of a small recurrence cluster, alone: G workgroups; per iteration every workgroup PUBLISHES one
// 1 KB block (64 lanes x 16 bytes {payload, tag, payload, tag}, one write-through store per lane) and GATHERS the
// blocks of its G - 1 peers (16-byte L2-bypassing loads, every tag checked, re-polled until all are new), then
// "computes" for `work` clock ticks.  Question (round 5): the persistent kernels measure 0.93 - 1.9 us from publish to
// "peers' data in" where a single-lane ping-pong (handoff_bench.hip) sees 0.35 - 0.4 us one way - what does the full-wave
// form cost with nothing else in the kernel, and does the placement (same XCD / different XCDs) or the poller count matter?
//   hipcc --offload-arch=gfx950 -O3 -o bin/ring_bench ring_bench.hip && bin/ring_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int POLLERS, int NB, int PW = 1>       // PW: waves that share the publish (NB / PW stores each)
       // POLLERS 1: one wave gathers all peers; 0: one wave per peer.  NB: KB published per member (16-byte pieces per lane)
__global__ __launch_bounds__(512) void ring(unsigned* buf, int G, int stride, int iters, int work, long long* out) {
  // member m of the cluster = block m * stride (the other blocks leave at once)
  if (blockIdx.x % stride != 0 || (int)(blockIdx.x / stride) >= G) return;
  const int me = blockIdx.x / stride, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int flag;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 2 * G * 1024 * NB, 0x00020000);
  long long t0 = wall_clock64(), wait_sum = 0;
  unsigned sink = 0;
  for (int it = 1; it <= iters; ++it) {
    const unsigned par = (it & 1) * G * 1024 * NB;
    if (wave >= 4 && wave < 4 + PW) {           // publish (waves 4 ..: the pollers are waves 1 ..)
      const u32x4 v = {(unsigned)lane, (unsigned)it, (unsigned)me, (unsigned)it};
#pragma unroll
      for (int b = wave - 4; b < NB; b += PW) __builtin_amdgcn_raw_buffer_store_b128(v, rs, par + (me * NB + b) * 1024 + lane * 16, 0, 16);
    }
    long long tw = wall_clock64();
    if (POLLERS == 2) {
      if (wave >= 1 && wave <= NB && wave < 4) {
        bool ok; unsigned spins = 0;
        do {
          asm volatile("" ::: "memory");
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, par + ((1 - me) * NB + wave - 1) * 1024 + lane * 16, 0, 16);
          ok = __all((v[1] == (unsigned)it) & (v[3] == (unsigned)it));
          sink += v[0];
        } while (!ok && ++spins < (1u << 20));
      }
    } else if (POLLERS == 1 ? wave == 1 : (wave >= 1 && wave < G)) {
      bool ok;
      unsigned spins = 0;
      do {
        ok = true;
        asm volatile("" ::: "memory");
        if (POLLERS == 1) {
          for (int p = 0; p < G; ++p) {
            if (p == me) continue;
            u32x4 v[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) v[b] = __builtin_amdgcn_raw_buffer_load_b128(rs, par + (p * NB + b) * 1024 + lane * 16, 0, 16);
#pragma unroll
            for (int b = 0; b < NB; ++b) { ok = ok & (v[b][1] == (unsigned)it) & (v[b][3] == (unsigned)it); sink += v[b][0] + v[b][2]; }
          }
        } else {
          const int p = (me + wave) % G;
          u32x4 v[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) v[b] = __builtin_amdgcn_raw_buffer_load_b128(rs, par + (p * NB + b) * 1024 + lane * 16, 0, 16);
#pragma unroll
          for (int b = 0; b < NB; ++b) { ok = ok & (v[b][1] == (unsigned)it) & (v[b][3] == (unsigned)it); sink += v[b][0] + v[b][2]; }
        }
        ok = __all(ok);
      } while (!ok && ++spins < (1u << 20));
    }
    __syncthreads();
    if (threadIdx.x == 64) wait_sum += wall_clock64() - tw;
    if (work > 0) { const long long w0 = wall_clock64(); while (wall_clock64() - w0 < work) {} }
  }
  if (sink == 0xdeadbeefu) out[7] = sink;
  if (threadIdx.x == 64 && me == 0) { out[0] = wall_clock64() - t0; out[1] = wait_sum; }
}

int main() {
  unsigned* buf; long long* out; long long h[2];
  hipMalloc(&buf, 1 << 20); hipMalloc(&out, 64);
  const int iters = 4000;
#define RUN(POL, NB, PW)                                                                                                   \
  hipMemset(buf, 0, 1 << 20); hipMemset(out, 0, 64);                                                                     \
  hipLaunchKernelGGL((ring<POL, NB, PW>), dim3(G * stride), dim3(512), 0, 0, buf, G, stride, iters, work, out);             \
  hipDeviceSynchronize();                                                                                                 \
  hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);                                                                           \
  printf("G %d %-12s %-18s publish by %d wave(s), %d KB per member, work %.1f us : %.2f us per iteration, publish -> all peers in %.2f us\n", G,  \
         stride == 1 ? "across XCDs" : "one XCD", POL == 2 ? "a wave per KB" : POL ? "one gathering wave" : "a wave per peer", PW, NB, work * 0.01,        \
         h[0] * 0.01 / iters, h[1] * 0.01 / iters);
  for (int G : {2, 4})
    for (int stride : {1, 8})              // 1: members on different XCDs; 8: on one XCD (round-robin dispatch)
      for (int work : {0, 60}) {           // 0.6 us of "compute" between the exchanges
        if (G == 2) { RUN(0, 1, 1) RUN(0, 3, 1) RUN(2, 3, 1) }
        else { RUN(0, 1, 1) RUN(0, 4, 1) RUN(0, 4, 2) RUN(0, 4, 4) RUN(0, 8, 1) RUN(0, 8, 4) }
      }
  return 0;
}
