// Library-wide state: thread-local error string, version, architecture tag.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void ns_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ns_version(void) { return 100; }
extern "C" const char* ns_device_arch(void) { return "gfx950"; }
extern "C" const char* ns_last_error(void) { return g_err; }

// Zero `bytes` (a multiple of 16, 16-byte aligned) on the stream with a kernel.  The persistent kernels clear their
// polled words with this before every launch instead of hipMemsetAsync: a memset NODE of a captured HIP graph was seen
// to leave pointer-like garbage in its destination from the second replay on (ROCm 7.2: status / tag words of the
// BiLSTM cluster kernel inside the synthesis graph), a kernel node replays faithfully.
__global__ void ns_zero_kernel(uint4* p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_uint4(0u, 0u, 0u, 0u);
}
int ns_zero_async(void* p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return NS_OK;
  NS_CHECK_ARG(p && (bytes & 15) == 0 && (((uintptr_t)p) & 15) == 0, "ns_zero_async: 16-byte granularity");
  const size_t n16 = bytes / 16;
  const int grid = (int)((n16 + 255) / 256 < 1024 ? (n16 + 255) / 256 : 1024);
  hipLaunchKernelGGL(ns_zero_kernel, dim3(grid), dim3(256), 0, s, (uint4*)p, n16);
  NS_CHECK_LAUNCH("ns_zero");
  return NS_OK;
}
