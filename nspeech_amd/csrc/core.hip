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

int ns_device_cus() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    cached[dev] = n;
  }
  return cached[dev];
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
  const int grid = (int)((n16 + 255) / 256 < 4096 ? (n16 + 255) / 256 : 4096);
  hipLaunchKernelGGL(ns_zero_kernel, dim3(grid), dim3(256), 0, s, (uint4*)p, n16);
  NS_CHECK_LAUNCH("ns_zero");
  return NS_OK;
}
extern "C" int ns_zero(void* p, size_t bytes, ns_stream_t stream) { return ns_zero_async(p, bytes, (hipStream_t)stream); }

// Several buffers in ONE launch (a backward pass clears dozens of small gradient accumulators: one 5 us launch each
// otherwise).  blockIdx.y = buffer, blockIdx.x strides over it.
struct ZeroMany { uint4* p[NS_ZERO_MANY_MAX]; size_t n16[NS_ZERO_MANY_MAX]; };
__global__ void ns_zero_many_kernel(ZeroMany z) {
  uint4* p = z.p[blockIdx.y];
  const size_t n16 = z.n16[blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_uint4(0u, 0u, 0u, 0u);
}
extern "C" int ns_zero_many(void* const* ptrs, const size_t* bytes, int n, ns_stream_t stream) {
  NS_CHECK_ARG(n >= 0 && (n == 0 || (ptrs && bytes)), "ns_zero_many: null");
  for (int i0 = 0; i0 < n; i0 += NS_ZERO_MANY_MAX) {
    ZeroMany z = {};
    const int m = n - i0 < NS_ZERO_MANY_MAX ? n - i0 : NS_ZERO_MANY_MAX;
    size_t most = 0;
    for (int i = 0; i < m; ++i) {
      NS_CHECK_ARG((bytes[i0 + i] & 15) == 0 && (((uintptr_t)ptrs[i0 + i]) & 15) == 0 && (ptrs[i0 + i] || !bytes[i0 + i]),
                   "ns_zero_many: 16-byte granularity");
      z.p[i] = (uint4*)ptrs[i0 + i];
      z.n16[i] = bytes[i0 + i] / 16;
      most = z.n16[i] > most ? z.n16[i] : most;
    }
    if (most == 0) continue;
    const int gx = (int)((most + 255) / 256 < 256 ? (most + 255) / 256 : 256);
    hipLaunchKernelGGL(ns_zero_many_kernel, dim3(gx, m), dim3(256), 0, (hipStream_t)stream, z);
    NS_CHECK_LAUNCH("ns_zero_many");
  }
  return NS_OK;
}

// ns_occupy / ns_wait_counter: hold CUs the way a collective's channel kernel does (see the header).  wall_clock64
// ticks at 100 MHz.
__global__ void ns_occupy_kernel(unsigned long long ticks, int* started) {
  extern __shared__ char occupy_lds[];
  if (threadIdx.x == 0) {
    occupy_lds[0] = 1;      // the LDS allocation must be real
    if (started) atomicAdd(started, 1);
  }
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void ns_wait_counter_kernel(const int* counter, int target, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && wall_clock64() - t0 < ticks)
    __builtin_amdgcn_s_sleep(8);
}
// The same with a register footprint: v127 is clobbered, so the kernel is allocated 128 VGPRs per lane (a collective's
// channel kernel is register-heavy; beside 2 x 248 VGPRs of a persistent recurrence's waves there is no room for it).
__global__ void ns_occupy_heavy_kernel(unsigned long long ticks, int* started) {
  extern __shared__ char occupy_lds[];
  if (threadIdx.x == 0) {
    occupy_lds[0] = 1;
    if (started) atomicAdd(started, 1);
  }
  asm volatile("v_mov_b32 v127, 0" ::: "v127");
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
extern "C" int ns_occupy(int blocks, int threads, int lds_bytes, int heavy, double usec, int* started, ns_stream_t stream) {
  NS_CHECK_ARG(blocks >= 1 && blocks <= 4096 && threads >= 64 && threads <= 1024 && (threads & 63) == 0,
               "ns_occupy: 1..4096 blocks of 64..1024 threads (a multiple of 64)");
  NS_CHECK_ARG(lds_bytes >= 0 && lds_bytes <= 160 * 1024 && usec >= 0 && usec <= 1e6, "ns_occupy: lds <= 160 KB, usec <= 1e6");
  if (lds_bytes > 64 * 1024)
    (void)hipFuncSetAttribute(heavy ? (const void*)ns_occupy_heavy_kernel : (const void*)ns_occupy_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const size_t lds = (size_t)(lds_bytes < 16 ? 16 : lds_bytes);
  const unsigned long long ticks = (unsigned long long)(usec * 100.0);
  if (heavy) hipLaunchKernelGGL(ns_occupy_heavy_kernel, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, ticks, started);
  else hipLaunchKernelGGL(ns_occupy_kernel, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, ticks, started);
  NS_CHECK_LAUNCH("ns_occupy");
  return NS_OK;
}
extern "C" int ns_wait_counter(const int* counter, int target, double timeout_usec, ns_stream_t stream) {
  NS_CHECK_ARG(counter && timeout_usec >= 0 && timeout_usec <= 1e6, "ns_wait_counter: null counter or timeout > 1 s");
  hipLaunchKernelGGL(ns_wait_counter_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, target,
                     (unsigned long long)(timeout_usec * 100.0));
  NS_CHECK_LAUNCH("ns_wait_counter");
  return NS_OK;
}

// ------------------------------------------------------------------ do two streams share a hardware queue?
__global__ void ns_probe_wait_kernel(int* w) {      // w[0]: the word, w[1]: seen
  const long long t0 = wall_clock64();
  int seen = 0;
  while (wall_clock64() - t0 < 20000) {             // 200 us at 100 MHz
    if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { seen = 1; break; }
  }
  w[1] = seen;
}
__global__ void ns_probe_set_kernel(int* w) { __hip_atomic_store(w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
extern "C" int ns_streams_concurrent(ns_stream_t a_, ns_stream_t b_, void* work) {
  hipStream_t a = (hipStream_t)a_, b = (hipStream_t)b_;
  NS_CHECK_ARG(work && (((uintptr_t)work) & 15) == 0, "ns_streams_concurrent: 16 bytes of device memory");
  if (a == b) return 0;
  int rc = ns_zero_async(work, 16, a);
  if (rc) return rc;
  if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) {
    ns_set_error("ns_streams_concurrent: synchronize failed");
    return NS_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(ns_probe_wait_kernel, dim3(1), dim3(1), 0, a, (int*)work);
  hipLaunchKernelGGL(ns_probe_set_kernel, dim3(1), dim3(1), 0, b, (int*)work);
  NS_CHECK_LAUNCH("ns_streams_concurrent");
  int host[2] = {0, 0};
  if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess ||
      hipMemcpy(host, work, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) {
    ns_set_error("ns_streams_concurrent: synchronize / copy failed");
    return NS_ERR_LAUNCH;
  }
  return host[1] ? 1 : 0;
}
