// One-way hand-off latency between two workgroups (one wave each), as a ping-pong through two 8-byte granules, by
// store flavour (plain | sc1 = agent-scope atomic store) and load flavour (sc1 = agent-scope atomic load | nt =
// nontemporal: bypasses the L1, served by the XCD's L2), for a same-XCD pair (blocks b and b + 8 under round-robin
// dispatch) and a cross-XCD pair.  Question (round 3): is an L2-level hand-off (plain store + nt load), which is only
// valid inside one XCD, enough faster than the placement-independent sc1 form to justify XCD-local recurrence clusters?
//   hipcc --offload-arch=gfx950 -O3 -o bin/handoff_bench handoff_bench.hip && bin/handoff_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
__device__ __forceinline__ u64 ld(const u64* p, int kind) {
  if (kind == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_nontemporal_load(p);
}
__device__ __forceinline__ void st(u64* p, u64 v, int kind) {
  if (kind == 0) *(volatile u64*)p = v;
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void pp(u64* slots, int* xcc, int pa, int pb, int iters, int skind, int lkind, long long* cycles) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) xcc[blockIdx.x] = (int)(id & 0xf);
  if (threadIdx.x != 0) return;
  const int me = blockIdx.x == pa ? 0 : (blockIdx.x == pb ? 1 : -1);
  if (me < 0) return;
  u64* mine = slots + 64 * me;
  u64* other = slots + 64 * (1 - me);
  long long t0 = wall_clock64();
  for (int i = 1; i <= iters; ++i) {
    if (me == 0) st(mine, (u64)i, skind);
    unsigned spins = 0;
    while (ld(other, lkind) != (u64)i) { if (++spins > (1u << 22)) { cycles[2] = -i; return; } }
    if (me == 1) st(mine, (u64)i, skind);
  }
  cycles[me] = wall_clock64() - t0;
}
int main() {
  u64* slots; int* xcc; long long* cyc;
  hipMalloc(&slots, 4096); hipMalloc(&xcc, 64 * 4); hipMalloc(&cyc, 64);
  int hx[64]; long long hc[3];
  const int iters = 2000;
  int pairs[2][2] = {{0, 8}, {0, 1}};
  const char* sn[2] = {"plain", "sc1"};
  const char* ln[2] = {"sc1", "nt"};
  for (int p = 0; p < 2; ++p)
    for (int sk = 0; sk < 2; ++sk)
      for (int lk = 0; lk < 2; ++lk) {
        hipMemset(slots, 0, 4096); hipMemset(cyc, 0, 64);
        hipLaunchKernelGGL(pp, dim3(16), dim3(64), 0, 0, slots, xcc, pairs[p][0], pairs[p][1], iters, sk, lk, cyc);
        hipDeviceSynchronize();
        hipMemcpy(hx, xcc, 64, hipMemcpyDeviceToHost); hipMemcpy(hc, cyc, 24, hipMemcpyDeviceToHost);
        printf("blocks %d,%d (xcc %d,%d) store %-5s load %-3s : %7.1f ns one way%s\n", pairs[p][0], pairs[p][1], hx[pairs[p][0]],
               hx[pairs[p][1]], sn[sk], ln[lk], (double)hc[0] / iters * 10.0 / 2, hc[2] ? "  TIMEOUT (never seen)" : "");
      }
  return 0;
}
