// Which operand-select forms of the packed-fp32 VALU instructions misread an operand when MFMA waves of ANOTHER kernel
// share the SIMD?  (profiles/r04_determinism.txt item 4: attn_post_kernel's odd filter taps came out wrong by ~0.15 %
// whenever this library's weight-gradient products ran beside it.)
// victim: every lane evaluates r = OP(a, b, c) on constant exact operands a = (1, 2), b = (3, 5), c = (7, 11) and counts the
// results that differ from what the instruction's definition gives; aggressor: a loop of v_mfma_f32_16x16x32_bf16 on a
// second stream, small enough in registers and LDS to share CUs with the victim.
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/tools/bin/pk_opsel_probe profiles/tools/pk_opsel_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define FORMS3(X)                                                                                  \
  X(0, "v_pk_fma_f32 %0, %1, %2, %3", 1 * 3 + 7, 2 * 5 + 11)                                        \
  X(1, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]", 2 * 3 + 7, 2 * 5 + 11)                         \
  X(2, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]", 1 * 5 + 7, 2 * 5 + 11)                         \
  X(3, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]", 1 * 3 + 11, 2 * 5 + 11)                        \
  X(4, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]", 1 * 3 + 7, 1 * 5 + 11)                      \
  X(5, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]", 1 * 3 + 7, 2 * 3 + 11)                      \
  X(6, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]", 1 * 3 + 7, 2 * 5 + 7)                       \
  X(7, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]", 1 * 5 + 7, 2 * 3 + 11)       \
  X(12, "v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0]", -(1 * 3) + 7, 2 * 5 + 11)                     \
  X(13, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] neg_lo:[0,1,0]", -(1 * 5) + 7, 2 * 5 + 11)
#define FORMS2(X)                                                                                  \
  X(8, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]", 1 * 5, 2 * 5)                                        \
  X(9, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]", 2 * 3, 2 * 5)                                        \
  X(10, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1]", 1 + 5, 2 + 5)                                       \
  X(11, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0]", 2 + 3, 2 + 5)                                       \
  X(14, "v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]", 1, 5)                                               \
  X(15, "v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]", 2, 3)                                               \
  X(16, "v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]", 2, 5)
constexpr int NFORMS = 17;

__global__ __launch_bounds__(256) void victim(unsigned* bad, int iters) {
  f2 a = (f2){1.f, 2.f}, b = (f2){3.f, 5.f}, c = (f2){7.f, 11.f};
  unsigned cnt[NFORMS];
#pragma unroll
  for (int i = 0; i < NFORMS; ++i) cnt[i] = 0;
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
    f2 r;
#define X(ID, TXT, ELO, EHI)                                              \
    asm volatile(TXT : "=v"(r) : "v"(a), "v"(b), "v"(c));                    \
    cnt[ID] += (r.x != (float)(ELO)) | (r.y != (float)(EHI));
    FORMS3(X)
#undef X
#define X(ID, TXT, ELO, EHI)                                              \
    asm volatile(TXT : "=v"(r) : "v"(a), "v"(b));                            \
    cnt[ID] += (r.x != (float)(ELO)) | (r.y != (float)(EHI));
    FORMS2(X)
#undef X
  }
#pragma unroll
  for (int i = 0; i < NFORMS; ++i) if (cnt[i]) atomicAdd(bad + i, cnt[i]);
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
// a VALU-only neighbour of the same footprint: is it the matrix core, or any second kernel?
__global__ __launch_bounds__(256) void valu_neighbour(float* out, int iters) {
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  for (int it = 0; it < iters * 16; ++it) { x = fmaf(x, y, 1e-6f); y = fmaf(y, 0.99999f, 1e-7f); }
  out[(long)blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}

int main() {
  const int iters = 4000, blocks = 1024;
  unsigned* bad; float* sink;
  hipMalloc(&bad, sizeof(unsigned) * NFORMS);
  hipMalloc(&sink, sizeof(float) * 2048 * 256);
  hipStream_t s0, s1;
  hipStreamCreate(&s0); hipStreamCreate(&s1);
  const char* names[NFORMS];
#define X(ID, TXT, ELO, EHI) names[ID] = TXT;
  FORMS3(X) FORMS2(X)
#undef X
  const char* who[3] = {"alone", "beside an MFMA kernel", "beside a VALU-only kernel"};
  for (int ag = 0; ag < 3; ++ag) {
    unsigned tot[NFORMS] = {0};
    for (int rep = 0; rep < 3; ++rep) {
      hipMemsetAsync(bad, 0, sizeof(unsigned) * NFORMS, s0);
      hipStreamSynchronize(s0);
      if (ag == 1) hipLaunchKernelGGL(aggressor, dim3(2048), dim3(256), 0, s1, sink, iters * 30);
      if (ag == 2) hipLaunchKernelGGL(valu_neighbour, dim3(2048), dim3(256), 0, s1, sink, iters * 4);
      hipLaunchKernelGGL(victim, dim3(blocks), dim3(256), 0, s0, bad, iters);
      hipDeviceSynchronize();
      unsigned h[NFORMS];
      hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
      for (int i = 0; i < NFORMS; ++i) tot[i] += h[i];
    }
    printf("victim %s: wrong results of %.3g evaluations per form (3 runs)\n", who[ag], 3.0 * iters * blocks * 256);
    for (int i = 0; i < NFORMS; ++i) printf("  %-70s %u\n", names[i], tot[i]);
  }
  return 0;
}
