// rows32: the step products of batched free-running synthesis (include/nspeech_hip.h, ns_rows32).
//
// A decoder step of tacotron2.py:55-83 under TacoTestHelper (helpers.py:7-38) multiplies at most 32 activation rows with
// matrices that never change during a synthesis call; nothing can be hoisted out of the time loop, so every step
// streams every weight once: 29 + 33 MB for the two decoder LSTMs in the split-bf16 (hi, lo) form the mel path needs.
// The per-step kernels of lstm.hip read k-contiguous [4H, K] weight rows and fp32 activation rows: a wave instruction
// touches 16 rows x 64 bytes, neighbouring lanes never share a cache line, both 16-row workgroups of a unit block read
// the same weights, and every workgroup splits the same activations into (hi, lo) again (18 us per LSTM step at 32 rows).
// Here BOTH operands live in memory as the MFMA fragments the kernel consumes:
//
//   weights      packed[tile t][k chunk kc][lane l][plane p][8 bf16]   tile = 16 output columns, chunk = 32 k, p = hi | lo
//   activations  packed[row tile][k chunk kc][lane l][plane p][8 bf16] row tile = 16 rows (two of them)
//
// lane (r16 = l & 15, g = l >> 4) holds column (row) r16 of the tile at k = 32 kc + 8 g .. + 7 - its B (A) fragment of
// v_mfma_f32_16x16x32_bf16 - so every wave instruction loads consecutive bytes and nothing is converted in the loop: the
// producing epilogue (this kernel's own, or the attention step's context kernel) stores each value as its (hi, lo) pair at
// its fragment position (ns_rows32_store, common.h).  One workgroup per weight tile (256 for H = 1024: one per CU), all
// 32 rows in it (every weight byte is read by exactly one workgroup), eight waves split K, every load of a wave is in
// flight before its first MFMA, partial sums meet in LDS.  fp32 activation rows are accepted too (split in registers).
#include "common.h"
#include <stdlib.h>

namespace {
constexpr int RW = 8;            // waves per workgroup (K split)

struct Rows32Args {
  int N, K, C, nkc, tiles;
  const float* a; long a_sn;
  const uint4* a_pk; int a_nkc, a_kc0;
  const uint4* packed;
  const float* bias;
  const float* add; long add_sn;
  int act;
  float* out; long out_sn;
  float* out2; long out2_sn;
  bf16_t* pk1; int pk1_nkc, pk1_col;
  bf16_t* pk2; int pk2_nkc, pk2_col;
  int H;
  const float* c_prev; long c_sn;
  float* c_out; long co_sn;
  float forget_bias, zc, zh, cell_clip;
  const float* h_prev; long hp_sn;
};

__device__ __forceinline__ void split8(const float4& x, const float4& y, bf16x8& hi, bf16x8& lo) {
  const float f[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bf16_t h = (bf16_t)f[i];
    hi[i] = h;
    lo[i] = (bf16_t)(f[i] - (float)h);
  }
}

// THREE: split-bf16 product (hi.hi + hi.lo + lo.hi), else hi.hi only; TWO: more than 16 rows.  Compile-time, and the
// chunks past a wave's share are loaded from its last valid chunk against a zeroed weight fragment, so that the operand
// stream is straight-line code: with run-time flags the compiler unswitched the loop into a few hundred blocks with a
// vmcnt(0) in front of every product.
// RG: k chunks a wave has in flight (its whole share where that is <= 8)
template <bool CELL, bool APK, bool THREE, bool TWO, int RG>
__global__ __launch_bounds__(RW * 64) void rows32_kernel(Rows32Args a) {
  __shared__ float red[RW][32][20];      // row stride 20 = 4 mod 16 floats: the C layout's four row groups land 4 banks apart
  __shared__ float zs[32][17];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x;
  const int cpw = (a.nkc + RW - 1) / RW;
  const int kc0 = wave * cpw, kc1 = min(a.nkc, kc0 + cpw);
  // epilogue operands of this thread's element, issued in front of the operand stream: their latency hides behind it
  float e_bias = 0.f, e_add = 0.f, e_cp = 0.f, e_hp = 0.f;
  if constexpr (!CELL) {
    const int row = tid >> 4, c = tile * 16 + (tid & 15);
    if (row < a.N && c < a.C) {
      if (a.bias) e_bias = a.bias[c];
      if (a.add) e_add = a.add[(long)row * a.add_sn + c];
    }
  } else {
    const int col = tid & 15, u = tile * 4 + (col & 3);
    if (a.bias && u < a.H) e_bias = a.bias[(col >> 2) * a.H + u];
    if (tid < 128) {
      const int n = tid >> 2, uu = tile * 4 + (tid & 3);
      if (n < a.N && uu < a.H) {
        if (a.c_prev) e_cp = a.c_prev[(long)n * a.c_sn + uu];
        if ((a.zc > 0.f || a.zh > 0.f) && a.h_prev) e_hp = a.h_prev[(long)n * a.hp_sn + uu];
      }
    }
  }
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const uint4* wp = a.packed + ((long)tile * a.nkc * 64 + lane) * 2;
  const float* arow0 = nullptr;
  const float* arow1 = nullptr;
  const uint4* ap0 = nullptr;
  const uint4* ap1 = nullptr;
  bool ok0 = false, ok1 = false;
  if constexpr (APK) {
    ap0 = a.a_pk + ((long)a.a_kc0 * 64 + lane) * 2;
    ap1 = ap0 + (long)a.a_nkc * 128;
  } else {
    arow0 = a.a + (long)r16 * a.a_sn + g * 8;
    arow1 = a.a + (long)(16 + r16) * a.a_sn + g * 8;
    ok0 = r16 < a.N; ok1 = 16 + r16 < a.N;
  }
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  for (int base = kc0; base < kc1; base += RG) {
    uint4 bh[RG], bl[RG];
    uint4 x0[RG][2], x1[RG][2];       // APK: (hi, lo) fragments; else the two float4 halves of 8 fp32 values
#pragma unroll
    for (int q = 0; q < RG; ++q) {
      const int kc = min(base + q, kc1 - 1);          // wave-uniform; past the share: the last chunk again (weights zeroed below)
      bh[q] = wp[(long)kc * 128];
      bl[q] = zero4;
      if constexpr (THREE) bl[q] = wp[(long)kc * 128 + 1];
      x0[q][0] = x0[q][1] = x1[q][0] = x1[q][1] = zero4;
      if constexpr (APK) {
        x0[q][0] = ap0[(long)kc * 128];
        if constexpr (THREE) x0[q][1] = ap0[(long)kc * 128 + 1];
        if constexpr (TWO) {
          x1[q][0] = ap1[(long)kc * 128];
          if constexpr (THREE) x1[q][1] = ap1[(long)kc * 128 + 1];
        }
      } else {
        const bool okk = kc * 32 + g * 8 < a.K;
        if (okk && ok0) { x0[q][0] = *(const uint4*)(arow0 + kc * 32); x0[q][1] = *(const uint4*)(arow0 + kc * 32 + 4); }
        if (TWO && okk && ok1) { x1[q][0] = *(const uint4*)(arow1 + kc * 32); x1[q][1] = *(const uint4*)(arow1 + kc * 32 + 4); }
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // every load of the group is issued before the first product (the scheduler sinks
                                           // them to their uses otherwise: five in flight per wave instead of fifty)
#pragma unroll
    for (int q = 0; q < RG; ++q) {
      const bool valid = base + q < kc1;
      bf16x8 ah, al;
      const bf16x8 wh = __builtin_bit_cast(bf16x8, valid ? bh[q] : zero4), wl = __builtin_bit_cast(bf16x8, valid ? bl[q] : zero4);
      if constexpr (APK) { ah = __builtin_bit_cast(bf16x8, x0[q][0]); al = __builtin_bit_cast(bf16x8, x0[q][1]); }
      else split8(__builtin_bit_cast(float4, x0[q][0]), __builtin_bit_cast(float4, x0[q][1]), ah, al);
      acc[0] = mfma_split<THREE ? 3 : 1>(ah, al, wh, wl, acc[0]);
      if constexpr (TWO) {
        if constexpr (APK) { ah = __builtin_bit_cast(bf16x8, x1[q][0]); al = __builtin_bit_cast(bf16x8, x1[q][1]); }
        else split8(__builtin_bit_cast(float4, x1[q][0]), __builtin_bit_cast(float4, x1[q][1]), ah, al);
        acc[1] = mfma_split<THREE ? 3 : 1>(ah, al, wh, wl, acc[1]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][i * 16 + g * 4 + r][r16] = acc[i][r];
  __syncthreads();
  const int row = tid >> 4, col = tid & 15;
  float z = e_bias;
#pragma unroll
  for (int w = 0; w < RW; ++w) z += red[w][row][col];
  if constexpr (!CELL) {
    const int c = tile * 16 + col;
    if (row < a.N && c < a.C) {
      z = apply_act(z + e_add, a.act);
      if (a.out) a.out[(long)row * a.out_sn + c] = z;
      if (a.out2) a.out2[(long)row * a.out2_sn + c] = z;
      if (a.pk1) ns_rows32_store(a.pk1, a.pk1_nkc, row, a.pk1_col + c, z);
      if (a.pk2) ns_rows32_store(a.pk2, a.pk2_nkc, row, a.pk2_col + c, z);
    }
  } else {
    const int H = a.H;
    zs[row][col] = z;          // column = gate * 4 + unit of the tile
    __syncthreads();
    if (tid < 128) {
      const int n = tid >> 2, ul = tid & 3, u = tile * 4 + ul;
      if (n < a.N && u < H) {
        const float gi = sigmoidf_(zs[n][ul]), gj = tanhf_(zs[n][4 + ul]);
        const float gf = sigmoidf_(zs[n][8 + ul] + a.forget_bias), go = sigmoidf_(zs[n][12 + ul]);
        float c = ns_cell_clip(gf * e_cp + gi * gj, a.cell_clip);
        float h = go * tanhf_(c);
        if (a.zc > 0.f || a.zh > 0.f) {       // zoneout at inference = the expectation of the training masks
          c = a.zc * e_cp + (1.f - a.zc) * c;
          h = a.zh * e_hp + (1.f - a.zh) * h;
        }
        a.c_out[(long)n * a.co_sn + u] = c;
        if (a.out) a.out[(long)n * a.out_sn + u] = h;
        if (a.out2) a.out2[(long)n * a.out2_sn + u] = h;
        if (a.pk1) ns_rows32_store(a.pk1, a.pk1_nkc, n, a.pk1_col + u, h);
        if (a.pk2) ns_rows32_store(a.pk2, a.pk2_nkc, n, a.pk2_col + u, h);
      }
    }
  }
}

// one thread per (tile, chunk, lane): its 8 k of one column, split into the hi and lo planes
__global__ __launch_bounds__(256) void rows32_pack_kernel(const float* w, long ldw, int K, int C, int H, int nkc, long total,
                                                          uint4* packed) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const long tk = i >> 6;
  const int kc = (int)(tk % nkc);
  const int tile = (int)(tk / nkc);
  const int r16 = lane & 15, g = lane >> 4;
  const int c = H > 0 ? (r16 >> 2) * H + tile * 4 + (r16 & 3) : tile * 16 + r16;
  const bool okc = H > 0 ? tile * 4 + (r16 & 3) < H : c < C;
  bf16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kc * 32 + g * 8 + j;
    const float x = (okc && k < K) ? w[(long)k * ldw + c] : 0.f;
    const bf16_t h = (bf16_t)x;
    hi[j] = h;
    lo[j] = (bf16_t)(x - (float)h);
  }
  packed[i * 2] = __builtin_bit_cast(uint4, hi);
  packed[i * 2 + 1] = __builtin_bit_cast(uint4, lo);
}

// fp32 rows -> the packed activation layout, columns [col0, col0 + K) of a packed buffer of nkc chunks
__global__ __launch_bounds__(256) void rows32_pack_rows_kernel(const float* a, long a_sn, int N, int K, bf16_t* pk, int nkc, int col0) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)N * K) return;
  const int n = (int)(i / K), k = (int)(i % K);
  ns_rows32_store(pk, nkc, n, col0 + k, a[(long)n * a_sn + k]);
}

inline int tiles_of(int C, int H) { return H > 0 ? (H + 3) / 4 : (C + 15) / 16; }
}  // namespace

extern "C" size_t ns_rows32_packed_bytes(int K, int C) {
  if (K <= 0 || C <= 0) return 0;
  // the cell form has C / 16 tiles too (C = 4 H, four units a tile)
  return (size_t)((C + 15) / 16) * ((K + 31) / 32) * 64 * 32;
}

extern "C" int ns_rows32_pack(const float* w, int64_t ldw, int K, int C, int cell_units, void* packed, ns_stream_t s_) {
  NS_CHECK_ARG(w && packed && K > 0 && C > 0 && ldw >= C, "ns_rows32_pack: null / empty");
  NS_CHECK_ARG(K % 8 == 0, "ns_rows32_pack: K %% 8 required (K = %d)", K);
  NS_CHECK_ARG(cell_units == 0 || (C == 4 * cell_units && cell_units % 4 == 0), "ns_rows32_pack: cell form needs C = 4 H, H %% 4 == 0");
  const int nkc = (K + 31) / 32;
  const long total = (long)tiles_of(C, cell_units) * nkc * 64;
  hipLaunchKernelGGL(rows32_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s_, w, (long)ldw, K, C,
                     cell_units, nkc, total, (uint4*)packed);
  NS_CHECK_LAUNCH("rows32_pack");
  return NS_OK;
}

extern "C" size_t ns_rows32_rows_bytes(int K) { return K > 0 ? (size_t)2 * ((K + 31) / 32) * 64 * 32 : 0; }

extern "C" int ns_rows32_pack_rows(const float* a, int64_t a_sn, int N, int K, void* rows, int rows_K, int col0, ns_stream_t s_) {
  NS_CHECK_ARG(a && rows && N >= 1 && N <= 32 && K > 0 && col0 >= 0 && col0 + K <= ((rows_K + 31) / 32) * 32,
               "ns_rows32_pack_rows: null / out of range");
  const long total = (long)N * K;
  hipLaunchKernelGGL(rows32_pack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s_, a, (long)a_sn, N,
                     K, (bf16_t*)rows, (rows_K + 31) / 32, col0);
  NS_CHECK_LAUNCH("rows32_pack_rows");
  return NS_OK;
}

extern "C" int ns_rows32(const ns_rows32_params* p, ns_stream_t s_) {
  NS_CHECK_ARG(p && (p->a || p->a_rows) && p->packed, "ns_rows32: null");
  NS_CHECK_ARG(p->N >= 1 && p->N <= 32, "ns_rows32: 1 <= N <= 32 rows (N = %d)", p->N);
  NS_CHECK_ARG(p->K > 0 && p->K % 8 == 0, "ns_rows32: K %% 8 required");
  NS_CHECK_ARG(p->f32_passes == 1 || p->f32_passes == 3, "ns_rows32: f32_passes 1 or 3");
  const int H = p->cell_units;
  NS_CHECK_ARG(H == 0 || (p->C == 4 * H && H % 4 == 0 && p->c_out), "ns_rows32: cell form needs C = 4 H, H %% 4 == 0, c_out");
  NS_CHECK_ARG(p->out || p->rows_out, "ns_rows32: no destination");
  Rows32Args a = {};
  a.N = p->N; a.K = p->K; a.C = p->C; a.nkc = (p->K + 31) / 32;
  a.packed = (const uint4*)p->packed; a.bias = p->bias;
  a.add = p->add; a.add_sn = p->add_sn; a.act = p->act;
  a.out = p->out; a.out_sn = p->out_sn; a.out2 = p->out2; a.out2_sn = p->out2_sn;
  a.pk1 = (bf16_t*)p->rows_out; a.pk1_nkc = (p->rows_out_K + 31) / 32; a.pk1_col = p->rows_out_col;
  a.pk2 = (bf16_t*)p->rows_out2; a.pk2_nkc = (p->rows_out2_K + 31) / 32; a.pk2_col = p->rows_out2_col;
  NS_CHECK_ARG(!a.pk1 || p->rows_out_col + (H ? H : p->C) <= a.pk1_nkc * 32, "ns_rows32: rows_out columns out of range");
  NS_CHECK_ARG(!a.pk2 || p->rows_out2_col + (H ? H : p->C) <= a.pk2_nkc * 32, "ns_rows32: rows_out2 columns out of range");
  a.H = H; a.c_prev = p->c_prev; a.c_sn = p->c_sn; a.c_out = p->c_out; a.co_sn = p->co_sn;
  a.forget_bias = p->forget_bias; a.zc = p->zoneout_cell; a.zh = p->zoneout_output; a.cell_clip = p->cell_clip;
  a.h_prev = p->h_prev; a.hp_sn = p->hp_sn;
  const int tiles = tiles_of(p->C, H);
  a.tiles = tiles;
  hipStream_t s = (hipStream_t)s_;
  const bool three = p->f32_passes >= 3, two = p->N > 16;
  const int cpw = (a.nkc + RW - 1) / RW;
#define NS_R32_G(CELL_, APK_, RG_)                                                                                         \
  do {                                                                                                                     \
    if (three && two) hipLaunchKernelGGL((rows32_kernel<CELL_, APK_, true, true, RG_>), dim3(tiles), dim3(RW * 64), 0, s, a);        \
    else if (three) hipLaunchKernelGGL((rows32_kernel<CELL_, APK_, true, false, RG_>), dim3(tiles), dim3(RW * 64), 0, s, a);         \
    else if (two) hipLaunchKernelGGL((rows32_kernel<CELL_, APK_, false, true, RG_>), dim3(tiles), dim3(RW * 64), 0, s, a);           \
    else hipLaunchKernelGGL((rows32_kernel<CELL_, APK_, false, false, RG_>), dim3(tiles), dim3(RW * 64), 0, s, a);                   \
  } while (0)
#define NS_R32(CELL_, APK_)                                 \
  do {                                                      \
    if (APK_ && cpw <= 2) NS_R32_G(CELL_, APK_, 2);         \
    else if (APK_ && cpw <= 4) NS_R32_G(CELL_, APK_, 4);    \
    else NS_R32_G(CELL_, APK_, 8);                          \
  } while (0)
  if (p->a_rows) {
    NS_CHECK_ARG(p->a_rows_col % 32 == 0 && p->a_rows_col + a.nkc * 32 <= ((p->a_rows_K + 31) / 32) * 32,
                 "ns_rows32: a_rows_col %% 32 and the operand inside the packed rows required");
    a.a_pk = (const uint4*)p->a_rows; a.a_nkc = (p->a_rows_K + 31) / 32; a.a_kc0 = p->a_rows_col / 32;
    if (H > 0) NS_R32(true, true); else NS_R32(false, true);
  } else {
    NS_CHECK_ARG(p->a_sn % 4 == 0 && ((uintptr_t)p->a & 15) == 0, "ns_rows32: 16-byte aligned activation rows required");
    a.a = p->a; a.a_sn = p->a_sn;
    if (H > 0) NS_R32(true, false); else NS_R32(false, false);
  }
#undef NS_R32
#undef NS_R32_G
  NS_CHECK_LAUNCH("rows32");
  return NS_OK;
}
