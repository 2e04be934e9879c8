// Companion of pk_opsel_probe.hip: do OTHER instruction classes this library's kernels use beside matrix-core waves (their
// own or another kernel's) ever return a wrong result while MFMA waves share the CU?  Exact operands, results compared
// with the instruction's definition inside the kernel, alone and beside a kernel of v_mfma_f32_16x16x32_bf16 loops.
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/tools/bin/valu_beside_mfma_probe profiles/tools/valu_beside_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NF = 12;
static const char* NAMES[NF] = {
    "v_fma_f32", "v_add_f32 dpp quad_perm:[1,0,3,2]", "v_add_f32 dpp row_mirror", "v_add_f32 dpp row_half_mirror",
    "v_cvt_pk_bf16_f32", "v_exp_f32 (powers of two)", "v_rcp_f32 (powers of two)", "ds_bpermute_b32 (lane ^ 16)",
    "ds_bpermute_b32 (lane ^ 32)", "v_perm_b32", "ds_write_b32 + ds_read_b32 (own lane)", "v_pk_fma_f32 (no select)"};

#define DPP_ADD(v, ctrl) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true)))

__global__ __launch_bounds__(256) void victim(unsigned* bad, int iters) {
  __shared__ float lds[256];
  const int lane = threadIdx.x & 63;
  unsigned cnt[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) cnt[i] = 0;
  float x = (float)lane, a = 3.f, b = 5.f, c = 7.f;
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(x), "+v"(a), "+v"(b), "+v"(c));
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    cnt[0] += r != 22.f;
    r = DPP_ADD(x, 0xB1);   asm volatile("" : "+v"(r)); cnt[1] += r != (float)(lane + (lane ^ 1));
    r = DPP_ADD(x, 0x140);  asm volatile("" : "+v"(r)); cnt[2] += r != (float)(lane + ((lane & ~15) | (15 - (lane & 15))));
    r = DPP_ADD(x, 0x141);  asm volatile("" : "+v"(r)); cnt[3] += r != (float)(lane + ((lane & ~7) | (7 - (lane & 7))));
    {
      unsigned pk;
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
      cnt[4] += pk != ((0x40A0u << 16) | 0x4040u);           // bf16(5) = 0x40A0, bf16(3) = 0x4040
    }
    asm volatile("v_exp_f32 %0, %1\n\ts_nop 1" : "=v"(r) : "v"(a)); cnt[5] += r != 8.f;
    {
      float p2 = 4.f;
      asm volatile("" : "+v"(p2));
      asm volatile("v_rcp_f32 %0, %1\n\ts_nop 1" : "=v"(r) : "v"(p2)); cnt[6] += r != 0.25f;
    }
    {
      const int v = __builtin_amdgcn_ds_bpermute(((lane ^ 16) << 2), __builtin_bit_cast(int, x));
      cnt[7] += __builtin_bit_cast(float, v) != (float)(lane ^ 16);
      const int w = __builtin_amdgcn_ds_bpermute(((lane ^ 32) << 2), __builtin_bit_cast(int, x));
      cnt[8] += __builtin_bit_cast(float, w) != (float)(lane ^ 32);
    }
    {
      unsigned s0 = 0x03020100u, s1 = 0x07060504u, sel = 0x00010405u, o;
      asm volatile("" : "+v"(s0), "+v"(s1));
      asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(o) : "v"(s1), "v"(s0), "v"(sel));
      cnt[9] += o != 0x00010405u;          // bytes {S0 = s1 (bytes 4..7), S1 = s0 (bytes 0..3)}: selector picks 0, 1, 4, 5
    }
    lds[threadIdx.x] = x + (float)it;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    cnt[10] += lds[threadIdx.x] != x + (float)it;
    {
      f2 pa = {1.f, 2.f}, pb = {3.f, 5.f}, pc = {7.f, 11.f}, pr;
      asm volatile("" : "+v"(pa), "+v"(pb), "+v"(pc));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pr) : "v"(pa), "v"(pb), "v"(pc));
      cnt[11] += (pr.x != 10.f) | (pr.y != 21.f);
    }
  }
#pragma unroll
  for (int i = 0; i < NF; ++i) if (cnt[i]) atomicAdd(bad + i, cnt[i]);
}

__global__ __launch_bounds__(256) void aggressor(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, b, c3, 0, 0, 0);
  }
  out[(long)blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
  const int iters = 4000, blocks = 1024;
  unsigned* bad; float* sink;
  (void)hipMalloc(&bad, sizeof(unsigned) * NF);
  (void)hipMalloc(&sink, sizeof(float) * 2048 * 256);
  hipStream_t s0, s1;
  (void)hipStreamCreate(&s0); (void)hipStreamCreate(&s1);
  for (int ag = 0; ag < 2; ++ag) {
    unsigned tot[NF] = {0};
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipMemsetAsync(bad, 0, sizeof(unsigned) * NF, s0);
      (void)hipStreamSynchronize(s0);
      if (ag) hipLaunchKernelGGL(aggressor, dim3(2048), dim3(256), 0, s1, sink, iters * 40);
      hipLaunchKernelGGL(victim, dim3(blocks), dim3(256), 0, s0, bad, iters);
      (void)hipDeviceSynchronize();
      unsigned h[NF];
      (void)hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
      for (int i = 0; i < NF; ++i) tot[i] += h[i];
    }
    printf("victim %s: wrong results of %.3g evaluations per instruction (3 runs)\n", ag ? "beside an MFMA kernel" : "alone", 3.0 * iters * blocks * 256);
    for (int i = 0; i < NF; ++i) printf("  %-46s %u\n", NAMES[i], tot[i]);
  }
  return 0;
}
