// GEMM family for the Tacotron hot path (dense, conv1d-as-strided-GEMM, LSTM/GRU gate
// products and all of their data/weight gradients).  Three kernels behind ns_gemm():
//   gemm_mfma_kernel   bf16 operands, 128x128x64 tiles, v_mfma_f32_16x16x32_bf16,
//                      swizzled LDS images, k-slow operands read with ds_read_b64_tr_b16
//   gemm_skinny_kernel bf16, M <= 32 rows (decoder / recurrent steps): fragments straight
//                      from L2 to VGPRs, K split over the 4 waves of a workgroup
//   gemm_generic_kernel any dtype / alignment (fp32 "exact" mode used by the parity tests,
//                      and odd shapes)
#include "common.h"

// ------------------------------------------------------------------ shared epilogue
struct Epi {
  void* C; long ldc; int c_dtype; int accumulate;
  const float* bias; int act; float alpha;
  int row_period, row_lo, row_hi, row_shift;
  int M, N;
  const void* addend; long ld_add; int add_bf16;
  const void* gate; long ld_gate; int gate_dtype;
  const void* sz; long ld_sz; int sz_bf16; const float* smean; const float* sistd;    // BatchNorm-backward statistics
};

__device__ __forceinline__ Epi make_epi(const ns_gemm_params& p) {
  Epi e;
  e.C = p.C; e.ldc = p.ldc; e.c_dtype = p.c_dtype; e.accumulate = p.accumulate;
  e.bias = p.bias; e.act = p.act; e.alpha = p.alpha;
  e.row_period = p.row_period; e.row_lo = p.row_lo; e.row_hi = p.row_hi; e.row_shift = p.row_shift;
  e.M = p.M; e.N = p.N;
  e.addend = p.addend; e.ld_add = p.ld_add; e.add_bf16 = p.addend_dtype == NS_BF16;
  e.gate = p.gate; e.ld_gate = p.ld_gate; e.gate_dtype = p.dtype;
  e.sz = p.stat_z; e.ld_sz = p.ld_stat_z; e.sz_bf16 = p.stat_z_dtype == NS_BF16; e.smean = p.stat_mean; e.sistd = p.stat_istd;
  return e;
}

__device__ __forceinline__ bool row_valid(const Epi& e, int m) {
  if (e.row_period <= 0) return true;
  int t = (m + e.row_shift) % e.row_period;
  return t >= e.row_lo && t < e.row_hi;
}

// value after bias/act/mask (what BN statistics see)
__device__ __forceinline__ float epi_value(const Epi& e, int m, int n, float acc, bool add_bias, bool valid) {
  float v = e.alpha * acc;
  if (add_bias && e.bias) v += e.bias[n];
  if (add_bias && e.addend) {
    const long ao = (long)m * e.ld_add + n;
    v += e.add_bf16 ? (float)((const bf16_t*)e.addend)[ao] : ((const float*)e.addend)[ao];
  }
  v = apply_act(v, e.act);
  if (e.gate) {
    const long go = (long)m * e.ld_gate + n;
    const float gv = e.gate_dtype == NS_BF16 ? (float)((const bf16_t*)e.gate)[go] : ((const float*)e.gate)[go];
    if (!(gv > 0.f)) v = 0.f;
  }
  return valid ? v : 0.f;
}

// returns the value now in C (what the statistics are taken on)
__device__ __forceinline__ float epi_store(const Epi& e, int m, int n, float v) {
  long off = (long)m * e.ldc + n;
  if (e.accumulate == 2) {
    atomicAdd((float*)e.C + off, v);
  } else if (e.accumulate == 1) {
    v += ((float*)e.C)[off];
    ((float*)e.C)[off] = v;
  } else if (e.c_dtype == NS_BF16) {
    const bf16_t b = (bf16_t)v;
    ((bf16_t*)e.C)[off] = b;
    v = (float)b;
  } else {
    ((float*)e.C)[off] = v;
  }
  return v;
}
// second statistic of a stored value: its square (BatchNorm forward: sum of squares), or, with stat_z, its product with
// the normalised saved input of the BatchNorm below (BatchNorm backward: sum dy * xhat)
__device__ __forceinline__ float stat_second(const Epi& e, int m, int n, float vs) {
  if (!e.sz) return vs * vs;
  const long o = (long)m * e.ld_sz + n;
  const float zv = e.sz_bf16 ? (float)((const bf16_t*)e.sz)[o] : ((const float*)e.sz)[o];
  return vs * ((zv - e.smean[n]) * e.sistd[n]);
}

// batched call: item z works on A + z*batch_stride_a, B + z*batch_stride_b, C + z*batch_stride_c (elements)
__device__ __forceinline__ void batch_shift(ns_gemm_params& p, int z) {
  if (z == 0) return;
  const long ea = p.dtype == NS_BF16 ? 2 : 4, ec = p.c_dtype == NS_BF16 ? 2 : 4;
  p.A = (const char*)p.A + (long)z * p.batch_stride_a * ea;
  p.B = (const char*)p.B + (long)z * p.batch_stride_b * ea;
  p.C = (char*)p.C + (long)z * p.batch_stride_c * ec;
}

// ---- deterministic split-K (ns_gemm_params.splitk_work): every k slice of an output tile parks its partial sums in the
// scratch - PER float4 per thread, thread-major, so the stores and the loads are whole 16-byte lines of one wave and the
// layout never has to be interpreted - and raises the tile's counter; the LAST slice to arrive reads all of them back in
// slice order (its own included: the order must not depend on who is last) and goes on into the epilogue.  Visibility:
// write-through partial stores, drained -> counter; L2-bypassing loads (common.h: ns_st_sc1 / ns_ld_sc1 - no fence).  Returns true for the workgroup that runs the epilogue.  A (tile, slice) slot is 256 x PER float4 = the
// tile's elements x 4 bytes.
template <int PER>
__device__ __forceinline__ bool splitk_gather(const ns_gemm_params& p, float4 (&v)[PER], int tile, int ksl, int tid) {
  __shared__ int sk_last;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned SLOT_BYTES = 256u * PER * 16u;
  // 16-byte write-through stores / L2-bypassing loads through a buffer descriptor over the scratch (aux 16 = sc1)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.splitk_work, 0, 0x7ffffff0, 0x00020000);
  const unsigned tile0 = (unsigned)tile * (unsigned)p.split_k * SLOT_BYTES + (unsigned)tid * 16u;
  const unsigned mine = tile0 + (unsigned)ksl * SLOT_BYTES;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const u32x4 x = {__float_as_uint(v[i].x), __float_as_uint(v[i].y), __float_as_uint(v[i].z), __float_as_uint(v[i].w)};
    __builtin_amdgcn_raw_buffer_store_b128(x, rs, mine + (unsigned)i * 4096u, 0, 16);
  }
  ns_drain_stores();
  __syncthreads();
  if (tid == 0) sk_last = atomicAdd(p.splitk_count + tile, 1) == p.split_k - 1;
  __syncthreads();
  if (!sk_last) return false;
  // the tail of the launch: one workgroup reads split_k slots - the next slot's loads are in flight while this one is added
#pragma unroll
  for (int i = 0; i < PER; ++i) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  u32x4 cur[PER], nxt[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) cur[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, tile0 + (unsigned)i * 4096u, 0, 16);
  for (int s = 0; s < p.split_k; ++s) {
    if (s + 1 < p.split_k) {
#pragma unroll
      for (int i = 0; i < PER; ++i)
        nxt[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, tile0 + (unsigned)(s + 1) * SLOT_BYTES + (unsigned)i * 4096u, 0, 16);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      v[i].x += __uint_as_float(cur[i][0]); v[i].y += __uint_as_float(cur[i][1]);
      v[i].z += __uint_as_float(cur[i][2]); v[i].w += __uint_as_float(cur[i][3]);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) cur[i] = nxt[i];
  }
  if (tid == 0) p.splitk_count[tile] = 0;         // left clean for the next call
  return true;
}
template <int NJ>
__device__ __forceinline__ bool splitk_gather_acc(const ns_gemm_params& p, f32x4 (&acc)[4][NJ], int tile, int ksl, int tid) {
  float4 v[4 * NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[i * NJ + j] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  if (!splitk_gather<4 * NJ>(p, v, tile, ksl, tid)) return false;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){v[i * NJ + j].x, v[i * NJ + j].y, v[i * NJ + j].z, v[i * NJ + j].w};
  return true;
}

// ------------------------------------------------------------------ generic kernel
template <typename T>
__device__ __forceinline__ float ld_a(const ns_gemm_params& p, int m, int k) {
  const T* A = (const T*)p.A;
  return p.a_mode == 0 ? ldf(A + (long)m * p.lda + k) : ldf(A + (long)k * p.lda + m);
}
template <typename T>
__device__ __forceinline__ float ld_b(const ns_gemm_params& p, int k, int n) {
  const T* B = (const T*)p.B;
  if (p.b_seg_len > 0) {
    int s = k / p.b_seg_len;
    B += (long)s * p.b_seg_stride;
    k -= s * p.b_seg_len;
  }
  return p.b_mode == 0 ? ldf(B + (long)n * p.ldb + k) : ldf(B + (long)k * p.ldb + n);
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_generic_kernel(ns_gemm_params p) {
  constexpr int BM = 64, BN = 64, BK = 16;
  __shared__ float As[BK][BM + 1];
  __shared__ float Bs[BK][BN + 1];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tx = tid & 15, ty = tid >> 4;
  // split-K over blockIdx.z
  const int nk = (p.K + BK - 1) / BK;
  const int per = (nk + p.split_k - 1) / p.split_k;
  const int ksl = blockIdx.z % p.split_k;
  batch_shift(p, blockIdx.z / p.split_k);
  const int kt0 = ksl * per, kt1 = min(nk, kt0 + per);
  float acc[4][4] = {};
  for (int kt = kt0; kt < kt1; ++kt) {
    const int k0 = kt * BK;
    for (int i = tid; i < BK * BM; i += 256) {
      int kk, mm;
      if (p.a_mode == 0) { kk = i % BK; mm = i / BK; } else { mm = i % BM; kk = i / BM; }
      int m = m0 + mm, k = k0 + kk;
      As[kk][mm] = (m < p.M && k < p.K) ? ld_a<T>(p, m, k) : 0.f;
    }
    for (int i = tid; i < BK * BN; i += 256) {
      int kk, nn;
      if (p.b_mode == 0) { kk = i % BK; nn = i / BK; } else { nn = i % BN; kk = i / BN; }
      int n = n0 + nn, k = k0 + kk;
      Bs[kk][nn] = (n < p.N && k < p.K) ? ld_b<T>(p, k, n) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  bool add_bias = (ksl == 0);
  if (p.splitk_work && p.split_k > 1) {           // deterministic split-K: the last slice of the tile sums and stores
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    if (!splitk_gather<4>(p, v, blockIdx.y * gridDim.x + blockIdx.x, ksl, tid)) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i][0] = v[i].x; acc[i][1] = v[i].y; acc[i][2] = v[i].z; acc[i][3] = v[i].w; }
    add_bias = true;
  }
  Epi e = make_epi(p);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i) {
    int m = m0 + ty * 4 + i;
    if (m >= p.M) continue;
    bool valid = row_valid(e, m);
    for (int j = 0; j < 4; ++j) {
      int n = n0 + tx * 4 + j;
      if (n >= p.N) continue;
      float v = epi_value(e, m, n, acc[i][j], add_bias, valid);
      const float vs = epi_store(e, m, n, v);       // stats are taken on the value as the consumer will read it back
      if (valid) {
        s1[j] += vs;
        s2[j] += stat_second(e, m, n, vs);
      }
    }
  }
  if (p.stat_part) {   // this block's 64 rows = one statistics slot; fixed summation order (no atomics)
#pragma unroll
    for (int j = 0; j < 4; ++j) { As[ty][tx * 4 + j] = s1[j]; Bs[ty][tx * 4 + j] = s2[j]; }
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float a = 0.f, b = 0.f;
      for (int r = 0; r < 16; ++r) { a += As[r][tid]; b += Bs[r][tid]; }
      p.stat_part[(long)blockIdx.y * p.N + n0 + tid] = a;
      p.stat_part[((long)p.stat_slots + blockIdx.y) * p.N + n0 + tid] = b;
    }
  }
}

// second stage of the BatchNorm statistics: adds the 64-row slots up in a fixed order.  Block = 32 columns x 32 slot
// lanes (1024 threads): lane q adds slots q, q + 32, ... in order, then the 32 partial sums pairwise in a fixed tree.
__global__ __launch_bounds__(1024) void gemm_stats_finalize_kernel(const float* part, int slots, int N, float* col_sum,
                                                                   float* col_sumsq) {
  __shared__ float red[32][33][2];
  const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c;
  float a = 0.f, b = 0.f;
  if (n < N) {
    for (int s = q; s < slots; s += 32) {
      a += part[(long)s * N + n];
      b += part[((long)slots + s) * N + n];
    }
  }
  red[q][c][0] = a; red[q][c][1] = b;
  __syncthreads();
#pragma unroll
  for (int h = 16; h >= 1; h >>= 1) {
    if (q < h) { red[q][c][0] += red[q + h][c][0]; red[q][c][1] += red[q + h][c][1]; }
    __syncthreads();
  }
  if (q == 0 && n < N) {
    col_sum[n] = red[0][c][0];
    if (col_sumsq) col_sumsq[n] = red[0][c][1];
  }
}

// ------------------------------------------------------------------ MFMA kernel
// LDS images (bf16):
//   k-contiguous operand ("row" image): [128 rows][64 k], 128 B rows, 16-B chunk index
//       XOR ((row>>1)&7)  -> ds_read_b128 fragment reads are bank-conflict free
//   k-slow operand ("col" image):       [64 k][128 cols], 256 B rows, 8-B unit index
//       XOR (s(k)<<2), s(k) = (k&3)|((k>>1)&4) -> ds_read_b64_tr_b16 conflict free
constexpr int GBM = 128, GBN = 128, GBK = 64;

__device__ __forceinline__ int row_img_off(int row, int chunk) {  // bytes
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}
__device__ __forceinline__ int col_swz(int k) { return (k & 3) | ((k >> 1) & 4); }
__device__ __forceinline__ int col_img_off_chunk(int k, int chunk) {  // 16-B chunk (write side)
  return k * 256 + ((chunk ^ (col_swz(k) << 1)) << 4);
}
__device__ __forceinline__ int col_img_off_unit(int k, int unit) {  // 8-B unit (tr-read side)
  return k * 256 + ((unit ^ (col_swz(k) << 2)) << 3);
}

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x8 frag_row(const char* img, int row, int chunk) {
  return *(const bf16x8*)(img + row_img_off(row, chunk));
}
// 16x16x32 operand from a k-slow image: lane (i = l&15, g = l>>4) needs [k = kbase+8g+j][col0+i]
__device__ __forceinline__ bf16x8 frag_col(const char* img, int kbase, int col0, int lane) {
  const int i16 = lane & 15, g = lane >> 4;
  const int q = i16 >> 2, pp = i16 & 3;
  const int k = kbase + 8 * g + q;
  const int unit = (col0 >> 2) + pp;
  bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + col_img_off_unit(k, unit)));
  bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + col_img_off_unit(k + 4, unit)));
  return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct Stage { uint4 v[4]; };

// global -> registers for one 128x64 operand tile.  r0 = first row/col of the tile in the
// non-contracted dim, extent = size of that dim, k0 = first k, K = contraction size.
template <int MODE, int BK = GBK, int ROWS = 128>     // ROWS: rows of a k-contiguous tile (64: the half-height A tile)
__device__ __forceinline__ void stage_load(Stage& s, const bf16_t* base, long ld, int r0, int extent,
                                           int k0, int K, int tid) {
  if (MODE == 0) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const int r = r0 + row, k = k0 + c * 8;
      if (r < extent && k < K) s.v[i] = *(const uint4*)(base + (long)r * ld + k);
      else s.v[i] = make_uint4(0, 0, 0, 0);
    }
  } else {
    const int c = tid & 15;
#pragma unroll
    for (int i = 0; i < BK / 16; ++i) {
      const int kr = (tid >> 4) + 16 * i;
      const int k = k0 + kr, r = r0 + c * 8;
      if (k < K && r < extent) s.v[i] = *(const uint4*)(base + (long)k * ld + r);
      else s.v[i] = make_uint4(0, 0, 0, 0);
    }
  }
}
template <int MODE, int BK = GBK, int ROWS = 128>
__device__ __forceinline__ void stage_store(const Stage& s, char* img, int tid) {
  if (MODE == 0) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int row = (tid >> 3) + 32 * i;
      *(uint4*)(img + row_img_off(row, c)) = s.v[i];
    }
  } else {
    const int c = tid & 15;
#pragma unroll
    for (int i = 0; i < BK / 16; ++i) {
      const int kr = (tid >> 4) + 16 * i;
      *(uint4*)(img + col_img_off_chunk(kr, c)) = s.v[i];
    }
  }
}

// C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg; each wave owns 64x64
__device__ __forceinline__ void mfma_epilogue(const ns_gemm_params& p, f32x4 (&acc)[4][4], int m0, int n0, int wm,
                                              int wn, int lane, bool add_bias) {
  Epi e = make_epi(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + (lane & 15);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (m < p.M && n < p.N) {
          const bool valid = row_valid(e, m);
          float v = epi_value(e, m, n, acc[i][j][r], add_bias, valid);
          const float vs = epi_store(e, m, n, v);
          if (valid) {
            s1 += vs;
            s2 += stat_second(e, m, n, vs);
          }
        }
      }
    }
    if (p.stat_part) {     // this wave's 64 rows = one statistics slot
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      if ((lane >> 4) == 0 && n < p.N) {
        const long slot = (m0 >> 6) + wm;
        p.stat_part[slot * p.N + n] = s1;
        p.stat_part[(p.stat_slots + slot) * p.N + n] = s2;
      }
    }
  }
}

// Which (output tile, k slice) a workgroup of the 128-tile kernels takes.  The hardware deals consecutive workgroups of the
// dispatch order (x fastest, then y) to the 8 XCDs in turn, and each XCD has its own L2.
//  - no split-K: the tiles one XCD gets are neighbours (column tiles fastest), so its L2 holds their shared A rows;
//  - split-K (round 3): the XCDs take contiguous runs of the SLICE-major list of (k slice, tile) items, so the workgroups
//    of one XCD read ONE k range of both operands and every byte of it is fetched by one L2 only.  Before, an XCD had a
//    fixed tenth of the tiles for every k slice: the weight gradient of a 5-tap convolution (2560 x 512 x 32124, split 6)
//    fetched 385 MB per launch for 66 MB of unique operand bytes - all of dY once per XCD, the input once per tap.
__device__ __forceinline__ void xcd_work_item(const ns_gemm_params& p, int nwg, int& tile, int& ksl) {
  if (p.split_k > 1 && p.batch == 1) {
    const int W = nwg * p.split_k;
    const int L = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = L & 7, q = W >> 3, r = W & 7;
    const int j = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    ksl = j / nwg;
    tile = j - ksl * nwg;
  } else {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    ksl = blockIdx.y;
  }
}

// ---- vector epilogue.  The 256-tile kernel issues its MFMAs with the operands SWAPPED (a = the B fragment, b = the A
// fragment), so a 16 x 16 accumulator holds the TRANSPOSED tile: lane l owns output row m = l & 15 and the four
// consecutive columns n = (l >> 4) * 4 + r.  One 16-byte (fp32) / 8-byte (bf16) store per tile and lane instead of four
// 4-byte ones, one row-mask test per row instead of per element, and bias / addend / gate / the BatchNorm-backward
// operand (stat_z) come in as vectors.  Round 3: the scalar form cost ~10 us of a 60 us tile, and +60 us per launch
// with the BatchNorm-backward statistics in it.
__device__ __forceinline__ float4 ldv4(const void* base, bool bf16, long off) {
  if (bf16) {
    const bf16x4 v = *(const bf16x4*)((const bf16_t*)base + off);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  return *(const float4*)((const float*)base + off);
}
template <int NJ>       // 16-column tiles per quadrant: 2 in the 256-tile kernel, 4 (a wave's 64 x 64) in the 128-tile kernels
__device__ __forceinline__ void x256_quadrant(const ns_gemm_params& p, f32x4 (&acc)[4][NJ], int mq, int nq, int lane) {
  const Epi e = make_epi(p);
  const int li = lane & 15, c4 = (lane >> 4) * 4;
  const bool want_stats = p.stat_part != nullptr;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = nq + j * 16 + c4;               // this lane's four columns n .. n + 3: all inside or all outside N (N % 4 == 0)
    if (n >= p.N) continue;                       // whole 16-lane rows take this branch together (row16_sum below)
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), smean = bias, sistd = bias;
    if (e.bias) bias = *(const float4*)(e.bias + n);
    if (e.sz) { smean = *(const float4*)(e.smean + n); sistd = *(const float4*)(e.sistd + n); }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mq + i * 16 + li;
      if (m >= p.M) continue;
      const bool valid = row_valid(e, m);
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      const float bb[4] = {bias.x, bias.y, bias.z, bias.w};
      float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e.addend) ad = ldv4(e.addend, e.add_bf16, (long)m * e.ld_add + n);
      const float aa[4] = {ad.x, ad.y, ad.z, ad.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act(e.alpha * v[r] + bb[r] + aa[r], e.act);
      if (e.gate) {
        const float4 gt = ldv4(e.gate, e.gate_dtype == NS_BF16, (long)m * e.ld_gate + n);
        const float gg[4] = {gt.x, gt.y, gt.z, gt.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) if (!(gg[r] > 0.f)) v[r] = 0.f;
      }
      if (!valid) { v[0] = v[1] = v[2] = v[3] = 0.f; }
      const long off = (long)m * e.ldc + n;
      if (e.accumulate == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd((float*)e.C + off + r, v[r]);
      } else if (e.accumulate == 1) {
        const float4 old = *(const float4*)((float*)e.C + off);
        v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w;
        *(float4*)((float*)e.C + off) = make_float4(v[0], v[1], v[2], v[3]);
      } else if (e.c_dtype == NS_BF16) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o[r] = (bf16_t)v[r]; v[r] = (float)o[r]; }    // statistics on the stored values
        *(bf16x4*)((bf16_t*)e.C + off) = o;
      } else {
        *(float4*)((float*)e.C + off) = make_float4(v[0], v[1], v[2], v[3]);
      }
      if (want_stats && valid) {
        if (e.sz) {
          const float4 zz = ldv4(e.sz, e.sz_bf16, (long)m * e.ld_sz + n);
          s2[0] += v[0] * ((zz.x - smean.x) * sistd.x); s2[1] += v[1] * ((zz.y - smean.y) * sistd.y);
          s2[2] += v[2] * ((zz.z - smean.z) * sistd.z); s2[3] += v[3] * ((zz.w - smean.w) * sistd.w);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) s2[r] += v[r] * v[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[r] += v[r];
      }
    }
    if (want_stats) {      // this quadrant's 64 rows = one statistics slot: add the 16 row lanes of each column group
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[r] = row16_sum(s1[r]); s2[r] = row16_sum(s2[r]); }
      if (li == 0) {
        const long slot = mq >> 6;
        *(float4*)(p.stat_part + slot * p.N + n) = make_float4(s1[0], s1[1], s1[2], s1[3]);
        *(float4*)(p.stat_part + (p.stat_slots + slot) * p.N + n) = make_float4(s2[0], s2[1], s2[2], s2[3]);
      }
    }
  }
}

// VEC: the products are issued with the operands swapped, so the accumulators hold the transposed tiles and the vector
// epilogue of the 256-tile kernel applies (x256_quadrant; the host asks for it when C and the epilogue operands allow
// 16-byte accesses and nothing is added atomically).  Round 3, switches in the kernel: of the 64 us of the encoder's
// convolution data gradient (5244 x 512 x 2560) 25 were the element-wise epilogue - 64 four-byte stores per lane.
// BK = 32 (both operands k-slow only: the weight-gradient products): 32 KB of LDS and half the stage registers per
// workgroup, so three workgroups share a CU where BK = 64 allows two.
// TM = 64 (k-contiguous A, vector epilogue only): 64 x 128 tiles, the four waves side by side (64 x 32 each), for
// products whose 128-row tiles would leave a third of the CUs without a workgroup (M ~ 5 000 rows: 164 tiles).
template <int AMODE, int BMODE, bool VEC, int BK = GBK, int TM = GBM>
__global__ __launch_bounds__(256, BK == 64 ? 2 : 3) void gemm_mfma_kernel(ns_gemm_params p) {
  static_assert(TM == 128 || (TM == 64 && AMODE == 0 && VEC && BK == 64), "TM = 64: k-contiguous A, vector epilogue");
  constexpr int WN = TM == 128 ? 2 : 4;         // waves along N
  constexpr int NJ = 8 / WN;                    // 16-column tiles per wave
  static_assert(BK == 64 || (BK == 32 && AMODE == 1 && BMODE == 1), "BK = 32: k-slow images only (their rows are k)");
  constexpr int IMG = BK * 256;                 // bytes per operand image (BK = 64: 16 KB for either layout)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [stage][A IMG | B IMG]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + GBN - 1) / GBN;
  const int tiles_m = (p.M + TM - 1) / TM;
  // XCD-aware bijective remap: workgroups that share an XCD (id % 8) walk neighbouring tiles
  const int nwg = tiles_m * tiles_n;
  int wgid, ksl;                                 // output tile, k slice
  xcd_work_item(p, nwg, wgid, ksl);
  const int tm = wgid / tiles_n, tn = wgid % tiles_n;
  const int m0 = tm * TM, n0 = tn * GBN;

  const int nk = (p.K + BK - 1) / BK;
  const int per = (nk + p.split_k - 1) / p.split_k;
  const int kt0 = ksl * per, kt1 = min(nk, kt0 + per);
  batch_shift(p, blockIdx.z);

  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto b_tile_base = [&](int k0, int& kin) -> const bf16_t* {
    if (p.b_seg_len > 0) {
      int s = k0 / p.b_seg_len;
      kin = k0 - s * p.b_seg_len;
      return B + (long)s * p.b_seg_stride;
    }
    kin = k0;
    return B;
  };
  const int KB = p.b_seg_len > 0 ? p.b_seg_len : p.K;

  Stage sa, sb;
  if (kt0 < kt1) {
    int kin;
    const bf16_t* bb = b_tile_base(kt0 * BK, kin);
    stage_load<AMODE, BK, TM>(sa, A, p.lda, m0, p.M, kt0 * BK, p.K, tid);
    stage_load<BMODE, BK>(sb, bb, p.ldb, n0, p.N, kin, KB, tid);
    stage_store<AMODE, BK, TM>(sa, smem, tid);
    stage_store<BMODE, BK>(sb, smem + IMG, tid);
  }
  __syncthreads();

  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    char* imgA = smem + cur * 2 * IMG;
    char* imgB = imgA + IMG;
    const bool more = kt + 1 < kt1;
    if (more) {
      int kin;
      const bf16_t* bb = b_tile_base((kt + 1) * BK, kin);
      stage_load<AMODE, BK, TM>(sa, A, p.lda, m0, p.M, (kt + 1) * BK, p.K, tid);
      stage_load<BMODE, BK>(sb, bb, p.ldb, n0, p.N, kin, KB, tid);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 af[4], bfr[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (AMODE == 0) af[i] = frag_row(imgA, wm * 64 + i * 16 + (lane & 15), ks * 4 + (lane >> 4));
        else af[i] = frag_col(imgA, ks * 32, wm * 64 + i * 16, lane);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (BMODE == 0) bfr[j] = frag_row(imgB, wn * (NJ * 16) + j * 16 + (lane & 15), ks * 4 + (lane >> 4));
        else bfr[j] = frag_col(imgB, ks * 32, wn * (NJ * 16) + j * 16, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = VEC ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      char* nA = smem + (cur ^ 1) * 2 * IMG;
      stage_store<AMODE, BK, TM>(sa, nA, tid);
      stage_store<BMODE, BK>(sb, nA + IMG, tid);
    }
    __syncthreads();
  }

  bool add_bias = ksl == 0;
  if (p.splitk_work && p.split_k > 1) {           // deterministic split-K: the last slice of the tile sums and stores
    if (!splitk_gather_acc<NJ>(p, acc, wgid, ksl, tid)) return;
    add_bias = true;
  }
  if constexpr (VEC) x256_quadrant(p, acc, m0 + wm * 64, n0 + wn * (NJ * 16), lane);
  else mfma_epilogue(p, acc, m0, n0, wm, wn, lane, add_bias);
}

// ------------------------------------------------------------------ 256 x 256 tiles, 8 phases per two K-tiles
// Both operands k-contiguous bf16 (a_mode 0, b_mode 0), K % 64 == 0.  One workgroup per CU: 8 waves as 2 (M) x 4 (N),
// 128 x 64 outputs per wave held as acc[A half][4][B half][2].  LDS = 2 buffers x {A0 A1 B0 B1} half-tiles of
// 128 rows x 64 k (16 KB each, 128-B rows, 16-B chunk index XOR ((row >> 1) & 7) as in the 128^2 kernel), filled by
// global_load_lds_dwordx4 (LDS image lane-linear per wave -> the swizzle sits on the SOURCE address).
// Wave (wr, wc) owns rows wr*64.. of BOTH A halves and columns wc*32.. of BOTH B halves, so a phase touches one
// half-tile per operand for all waves and half-tiles free up one by one.  Fragment reads are spread 8 / 4 / 8 / 4 over the
// phases (a wave group's reads of one phase then take the LDS 256 cycles - the length of the other group's 16 MFMAs):
//   phase 1: read A0       | A0 x B0      phase 2: read B1       | A0 x B1
//   phase 3: read A1       | A1 x B1      phase 4: read B0(t+1)  | A1 x B0   (into the registers B1 just left: the two
//                                                                             B fragment sets swap roles every tile)
// Half-tiles are restaged TWO phases after their last read: P1 <- A1(t+1), P2 <- B0(t+2), P3 <- A0(t+2), P4 <- B1(t+2),
// and every phase carries one counted wait, vmcnt(10): everything but the five newest half-tiles has landed = the
// half-tile the NEXT phase reads (each load has five phases to arrive).
// Each phase is [ds_read + stage + vmcnt] barrier [lgkmcnt(0), 16 MFMA] barrier; the wr = 1 waves run one barrier behind
// the wr = 0 waves, so on every SIMD one wave feeds the matrix core while the other one loads, and a group's fragment
// reads complete in the shadow of the other group's MFMAs rather than in front of the barrier (round 5; rounds 2 - 4
// waited in front of it and read 12 / 4 / 8 / 0: 1090 -> 1177 TFLOP/s at 4096^3 for the wait alone, A/B in one box).
// Hazards.  RAW: the wait that retires a half-tile sits in front of phase p's first barrier, the read behind phase p's
// second: two barriers, one of which the staggered group has also passed behind its own wait.  WAR: a read retires at
// the lgkmcnt(0) behind its phase's first barrier, in front of that phase's MFMAs; the restage two phases later is
// issued behind two more barriers of the issuing group, i.e. at least one that the other group reaches only after its
// MFMAs of the reading phase.
// NSEG = 3: split-bf16 product of pre-split operands, K-tiles walk (A, B), (A, B_lo), (A_lo, B).
// NSEG = 2 (f32_passes = 2 with pre-split operands): (A, B), (A_lo, B) - B (the weights) rounded to bf16, A exact to
// ~16 bits; for products whose error budget allows it (the non-recurrent postnet convolutions).
// Measured on MI355X (uniform random operands, profiles/r05_x256_ab.txt): 4096^3 1203 TFLOP/s, 8192^3 1212, conv data
// gradient 32124 x 512 x 2560 1044 (rounds 2 - 4, same box: 1082 / 1157 / 988; 128^2 kernel: 662 / - / 610);
// three-segment product at 4096^3 434 algorithmic = 1302 issued (round 4: 383; in-kernel split: 239).
constexpr int XHALF = 16384, XBUF = 65536;

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NSEG>
__global__ __launch_bounds__(512, 1) void gemm_x256_kernel(ns_gemm_params p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (p.N + 255) / 256, tiles_m = (p.M + 255) / 256;
  const int nwg = tiles_m * tiles_n;
  int wgid;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  // column tiles fastest: the workgroups of one XCD share their A rows through its L2
  const int tm = wgid / tiles_n, tn = wgid % tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int nk = p.K / 64, NT = nk * NSEG;

  // staging: piece j of a half-tile covers rows (j*8 + wave)*8 + (lane >> 3), slot lane & 7 holds chunk slot ^ swz(row)
  unsigned offA00, offA01, offA10, offA11, offB00, offB01, offB10, offB11;
  {
    const int r0 = wave * 8 + (lane >> 3), r1 = r0 + 64;
    const int c0 = (lane & 7) ^ ((r0 >> 1) & 7), c1 = (lane & 7) ^ ((r1 >> 1) & 7);
    auto oa = [&](int h, int r, int c) { return (unsigned)(((long)min(m0 + h * 128 + r, p.M - 1) * p.lda + c * 8) * 2); };
    auto ob = [&](int h, int r, int c) { return (unsigned)(((long)min(n0 + h * 128 + r, p.N - 1) * p.ldb + c * 8) * 2); };
    offA00 = oa(0, r0, c0); offA01 = oa(0, r1, c1); offA10 = oa(1, r0, c0); offA11 = oa(1, r1, c1);
    offB00 = ob(0, r0, c0); offB01 = ob(0, r1, c1); offB10 = ob(1, r0, c0); offB11 = ob(1, r1, c1);
  }
  const char* const Ahi = (const char*)p.A;
  const char* const Bhi = (const char*)p.B;
  const char* const Alo = (const char*)p.A_lo;
  const char* const Blo = (const char*)p.B_lo;
  char* const lstage = smem + wave * 1024;

#define X_ISSUE(T, HALF, O0, O1, ISA)                                                                   \
  do {                                                                                                   \
    const int t_ = (T);                                                                                  \
    if (t_ < NT) {                                                                                       \
      const int sg_ = NSEG == 1 ? 0 : t_ / nk;                                                           \
      const int kt_ = NSEG == 1 ? t_ : t_ - sg_ * nk;                                                    \
      const char* g_ = (ISA) ? (sg_ == NSEG - 1 && NSEG > 1 ? Alo : Ahi) : (sg_ == 1 && NSEG == 3 ? Blo : Bhi); \
      if (!(ISA) && p.b_seg_len > 0) {      /* segment s of B's K range starts at B + s * b_seg_stride */ \
        const int bs_ = (kt_ * 64) / p.b_seg_len;                                                        \
        g_ += ((long)bs_ * p.b_seg_stride + (kt_ * 64 - bs_ * p.b_seg_len)) * 2;                         \
      } else {                                                                                           \
        g_ += (long)kt_ * 128;                                                                           \
      }                                                                                                  \
      char* l_ = lstage + (t_ & 1) * XBUF + (HALF) * XHALF;                                              \
      __builtin_amdgcn_global_load_lds((gptr_t)(g_ + (O0)), (lptr_t)l_, 16, 0, 0);                       \
      __builtin_amdgcn_global_load_lds((gptr_t)(g_ + (O1)), (lptr_t)(l_ + 8192), 16, 0, 0);             \
    }                                                                                                    \
  } while (0)
#define X_ISSUE_A0(T) X_ISSUE(T, 0, offA00, offA01, 1)
#define X_ISSUE_A1(T) X_ISSUE(T, 1, offA10, offA11, 1)
#define X_ISSUE_B0(T) X_ISSUE(T, 2, offB00, offB01, 0)
#define X_ISSUE_B1(T) X_ISSUE(T, 3, offB10, offB11, 0)

  // fragment addresses: row = w*{64,32} + tile*16 + (lane & 15), chunk = kk*4 + (lane >> 4), swizzle from lane only
  const int li = lane & 15, sw = (li >> 1) & 7, gq = lane >> 4;
  const int fa0 = (wr * 64 + li) * 128 + ((gq ^ sw) << 4);
  const int fa1 = (wr * 64 + li) * 128 + (((4 + gq) ^ sw) << 4);
  const int fb0 = 2 * XHALF + (wc * 32 + li) * 128 + ((gq ^ sw) << 4);
  const int fb1 = 2 * XHALF + (wc * 32 + li) * 128 + (((4 + gq) ^ sw) << 4);

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[h][g][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[4][2], fbe[2][2], fbo[2][2];     // B fragments: even tiles keep B0 in fbe and B1 in fbo, odd tiles the other way round

#define X_READ_A(BUF, H)                                                                   \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                          \
    fa[i][0] = *(const bf16x8*)(smem + (BUF) * XBUF + (H) * XHALF + fa0 + i * 2048);       \
    fa[i][1] = *(const bf16x8*)(smem + (BUF) * XBUF + (H) * XHALF + fa1 + i * 2048);       \
  }
#define X_READ_B(DST, BUF, H)                                                              \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                          \
    DST[j][0] = *(const bf16x8*)(smem + (BUF) * XBUF + (H) * XHALF + fb0 + j * 2048);      \
    DST[j][1] = *(const bf16x8*)(smem + (BUF) * XBUF + (H) * XHALF + fb1 + j * 2048);      \
  }
#define X_MFMA(H, G, BR)                                                                                   \
  do {                                                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                       \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
          acc[H][G][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BR[j][kk], fa[i][kk], acc[H][G][i][j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                         \
  } while (0)
  // the counted wait of a phase: everything but the newest FULL / 2 loads has landed while tile t + 2 exists, PEN / 2 when
  // t + 1 is the last tile, LAST / 2 when t is
#define X_WAIT(FULL, PEN, LAST)                                                  \
  do {                                                                           \
    if (t + 2 < NT) asm volatile("s_waitcnt vmcnt(" #FULL ")" ::: "memory");     \
    else if (t + 1 < NT) asm volatile("s_waitcnt vmcnt(" #PEN ")" ::: "memory"); \
    else asm volatile("s_waitcnt vmcnt(" #LAST ")" ::: "memory");                \
  } while (0)
  // first barrier of a phase; the fragment reads issued in front of it are waited for behind it, in the shadow of the
  // other wave group's MFMA phase
#define X_SYNC_LOADS()                                                 \
  do {                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                 \
    __builtin_amdgcn_s_barrier();                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 \
    __builtin_amdgcn_sched_barrier(0);                                 \
  } while (0)
#define X_SYNC_MFMA()                               \
  do {                                              \
    __builtin_amdgcn_sched_barrier(0);              \
    __builtin_amdgcn_s_barrier();                   \
    __builtin_amdgcn_sched_barrier(0);              \
  } while (0)

  // prologue: the loads a steady-state phase 1 of tile 0 finds issued (tile 0, and tile 1 without its A1), the first two
  // half-tiles landed, B0 of tile 0 in registers
  X_ISSUE_B0(0); X_ISSUE_A0(0); X_ISSUE_B1(0); X_ISSUE_A1(0);
  X_ISSUE_B0(1); X_ISSUE_A0(1); X_ISSUE_B1(1);
  if (NT > 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  X_READ_B(fbe, 0, 0)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) __builtin_amdgcn_s_barrier();          // the stagger
  __builtin_amdgcn_sched_barrier(0);

  // B0R holds B0 of tile t on entry; B1R takes B1 of tile t in phase 2 and B0 of tile t + 1 in phase 4
#define X_TILE(BUF, B0R, B1R)                                           \
  do {                                                                  \
    /* phase 1 */                                                       \
    X_READ_A(BUF, 0)                                                    \
    X_ISSUE_A1(t + 1);                                                  \
    if (t + 1 < NT) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   \
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");               \
    X_SYNC_LOADS();                                                     \
    X_MFMA(0, 0, B0R);                                                  \
    X_SYNC_MFMA();                                                      \
    /* phase 2 */                                                       \
    X_READ_B(B1R, BUF, 1)                                               \
    X_ISSUE_B0(t + 2);                                                  \
    X_WAIT(10, 8, 0);                                                   \
    X_SYNC_LOADS();                                                     \
    X_MFMA(0, 1, B1R);                                                  \
    X_SYNC_MFMA();                                                      \
    /* phase 3 */                                                       \
    X_READ_A(BUF, 1)                                                    \
    X_ISSUE_A0(t + 2);                                                  \
    X_WAIT(10, 6, 0);                                                   \
    X_SYNC_LOADS();                                                     \
    X_MFMA(1, 1, B1R);                                                  \
    X_SYNC_MFMA();                                                      \
    /* phase 4 */                                                       \
    X_READ_B(B1R, 1 - (BUF), 0)                                         \
    X_ISSUE_B1(t + 2);                                                  \
    X_WAIT(10, 4, 0);                                                   \
    X_SYNC_LOADS();                                                     \
    X_MFMA(1, 0, B0R);                                                  \
    X_SYNC_MFMA();                                                      \
  } while (0)

  int t = 0;
  for (; t + 1 < NT; t += 2) {
    X_TILE(0, fbe, fbo);
    ++t;
    X_TILE(1, fbo, fbe);
    --t;
  }
  if (t < NT) X_TILE(0, fbe, fbo);
  if (wr == 0) __builtin_amdgcn_s_barrier();          // pairs with the stagger
#undef X_TILE
#undef X_SYNC_MFMA
#undef X_SYNC_LOADS
#undef X_WAIT
#undef X_MFMA
#undef X_READ_B
#undef X_READ_A
#undef X_ISSUE_B1
#undef X_ISSUE_B0
#undef X_ISSUE_A1
#undef X_ISSUE_A0
#undef X_ISSUE

  // epilogue, one 64 x 32 quadrant at a time
  x256_quadrant(p, acc[0][0], m0 + wr * 64, n0 + wc * 32, lane);
  x256_quadrant(p, acc[0][1], m0 + wr * 64, n0 + 128 + wc * 32, lane);
  x256_quadrant(p, acc[1][0], m0 + 128 + wr * 64, n0 + wc * 32, lane);
  x256_quadrant(p, acc[1][1], m0 + 128 + wr * 64, n0 + 128 + wc * 32, lane);
}

// ------------------------------------------------------------------ skinny kernel (M <= 32)
// Both operands k-contiguous (a_mode 0, b_mode 0).  One workgroup = 32 rows x 64 columns,
// the 4 waves split K; fragments are 16-B global loads (weights stay L2 / MALL resident
// across the time loop), partial sums meet in LDS.
constexpr int SKW = 8;      // waves per workgroup
constexpr int SKG = 4;      // K chunks in flight per wave
template <int NT>
__global__ __launch_bounds__(SKW * 64) void gemm_skinny_kernel(ns_gemm_params p) {
  __shared__ float red[SKW][32][NT * 16 + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * (NT * 16);
  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;
  const int r16 = lane & 15, g = lane >> 4;
  const bf16x8 z8 = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // K chunks of 32 dealt round-robin to waves; each wave keeps SKG chunks of loads in flight
  const int nkc = (p.K + 31) / 32;
  const bf16_t* arow0 = A + (long)r16 * p.lda;
  const bf16_t* arow1 = A + (long)(16 + r16) * p.lda;
  const bool ok0 = r16 < p.M, ok1 = 16 + r16 < p.M;
  for (int kc0 = wave; kc0 < nkc; kc0 += SKW * SKG) {
    bf16x8 af[SKG][2], bfr[SKG][NT];
#pragma unroll
    for (int q = 0; q < SKG; ++q) {
      const int k = (kc0 + q * SKW) * 32 + g * 8;
      const bool okk = k < p.K;
      af[q][0] = (ok0 && okk) ? *(const bf16x8*)(arow0 + k) : z8;
      af[q][1] = (ok1 && okk) ? *(const bf16x8*)(arow1 + k) : z8;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 16 + r16;
        bfr[q][j] = (n < p.N && okk) ? *(const bf16x8*)(B + (long)n * p.ldb + k) : z8;
      }
    }
#pragma unroll
    for (int q = 0; q < SKG; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[q][i], bfr[q][j], acc[i][j], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][i * 16 + g * 4 + r][j * 16 + r16] = acc[i][j][r];
  __syncthreads();
  Epi e = make_epi(p);
  for (int idx = tid; idx < 32 * NT * 16; idx += SKW * 64) {
    const int mm = idx / (NT * 16), nn = idx % (NT * 16);
    const int m = mm, n = n0 + nn;
    if (m >= p.M || n >= p.N) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SKW; ++w) v += red[w][mm][nn];
    const bool valid = row_valid(e, m);
    v = epi_value(e, m, n, v, true, valid);
    epi_store(e, m, n, v);
  }
}

// ------------------------------------------------------------------ fp32 operands on bf16 MFMA
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): a.b ~ hi.hi + hi.lo + lo.hi (PASSES = 3) keeps
// ~16 mantissa bits at 3/16 of the bf16 MFMA rate - 3x faster than the native fp32 MFMA (1/16).
// 128x128x32 tiles; per stage four 8 KB images (A hi/lo, B hi/lo); same tr-read scheme for
// k-slow operands.
constexpr int FBK = 32;
__device__ __forceinline__ int row32_img_off(int row, int chunk16) {   // 64-B rows, 4 chunks of 16 B
  return row * 64 + ((chunk16 ^ ((row >> 2) & 3)) << 4);
}
struct StageF { float4 v[4]; };

template <int MODE, int ROWS = 128>
__device__ __forceinline__ void stagef_load(StageF& s, const float* base, long ld, int r0, int extent, int k0,
                                            int K, int tid) {
  if (MODE == 0) {
    const int c = tid & 7;            // 8 float4 per 32-float row
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const int r = r0 + row, k = k0 + c * 4;
      if (r < extent && k < K) s.v[i] = *(const float4*)(base + (long)r * ld + k);
      else s.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    const int c = tid & 31;           // 32 float4 per 128-float row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kr = (tid >> 5) + 8 * i;
      const int k = k0 + kr, r = r0 + c * 4;
      if (k < K && r < extent) s.v[i] = *(const float4*)(base + (long)k * ld + r);
      else s.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}
__device__ __forceinline__ void split4(const float4& x, bf16x4& hi, bf16x4& lo) {
  const float f[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16_t h = (bf16_t)f[i];
    hi[i] = h;
    lo[i] = (bf16_t)(f[i] - (float)h);
  }
}
template <int MODE, int PASSES, int ROWS = 128>
__device__ __forceinline__ void stagef_store(const StageF& s, char* img_hi, char* img_lo, int tid) {
#pragma unroll
  for (int i = 0; i < (MODE == 0 ? ROWS / 32 : 4); ++i) {
    bf16x4 hi, lo;
    split4(s.v[i], hi, lo);
    int off;
    if (MODE == 0) {
      const int c = tid & 7, row = (tid >> 3) + 32 * i;
      off = row32_img_off(row, c >> 1) + (c & 1) * 8;
    } else {
      const int c = tid & 31, kr = (tid >> 5) + 8 * i;
      off = col_img_off_unit(kr, c);
    }
    *(bf16x4*)(img_hi + off) = hi;
    if (PASSES > 1) *(bf16x4*)(img_lo + off) = lo;
  }
}

template <int AMODE, int BMODE, int PASSES, bool VEC, int TM = GBM>      // VEC, TM: as in gemm_mfma_kernel
__global__ __launch_bounds__(256, 2) void gemm_mfma_f32_kernel(ns_gemm_params p) {
  static_assert(TM == 128 || (TM == 64 && AMODE == 0 && VEC), "TM = 64: k-contiguous A, vector epilogue");
  constexpr int WN = TM == 128 ? 2 : 4;
  constexpr int NJ = 8 / WN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [stage][A hi 8K | A lo 8K | B hi 8K | B lo 8K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + GBN - 1) / GBN;
  const int tiles_m = (p.M + TM - 1) / TM;
  const int nwg = tiles_m * tiles_n;
  int wgid, ksl;                                 // output tile, k slice
  xcd_work_item(p, nwg, wgid, ksl);
  const int tm = wgid / tiles_n, tn = wgid % tiles_n;
  const int m0 = tm * TM, n0 = tn * GBN;
  const int nk = (p.K + FBK - 1) / FBK;
  const int per = (nk + p.split_k - 1) / p.split_k;
  const int kt0 = ksl * per, kt1 = min(nk, kt0 + per);
  batch_shift(p, blockIdx.z);
  const float* A = (const float*)p.A;
  const float* B = (const float*)p.B;
  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto b_tile_base = [&](int k0, int& kin) -> const float* {
    if (p.b_seg_len > 0) {
      int sg = k0 / p.b_seg_len;
      kin = k0 - sg * p.b_seg_len;
      return B + (long)sg * p.b_seg_stride;
    }
    kin = k0;
    return B;
  };
  const int KB = p.b_seg_len > 0 ? p.b_seg_len : p.K;
  StageF sa, sb;
  if (kt0 < kt1) {
    int kin;
    const float* bb = b_tile_base(kt0 * FBK, kin);
    stagef_load<AMODE, TM>(sa, A, p.lda, m0, p.M, kt0 * FBK, p.K, tid);
    stagef_load<BMODE>(sb, bb, p.ldb, n0, p.N, kin, KB, tid);
    stagef_store<AMODE, PASSES, TM>(sa, smem, smem + 8192, tid);
    stagef_store<BMODE, PASSES>(sb, smem + 16384, smem + 24576, tid);
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    char* iAh = smem + cur * 32768;
    char* iAl = iAh + 8192;
    char* iBh = iAh + 16384;
    char* iBl = iAh + 24576;
    const bool more = kt + 1 < kt1;
    if (more) {
      int kin;
      const float* bb = b_tile_base((kt + 1) * FBK, kin);
      stagef_load<AMODE, TM>(sa, A, p.lda, m0, p.M, (kt + 1) * FBK, p.K, tid);
      stagef_load<BMODE>(sb, bb, p.ldb, n0, p.N, kin, KB, tid);
    }
    bf16x8 ah[4], al[4], bh[NJ], bl[NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (AMODE == 0) {
        const int off = row32_img_off(wm * 64 + i * 16 + (lane & 15), lane >> 4);
        ah[i] = *(const bf16x8*)(iAh + off);
        if (PASSES > 1) al[i] = *(const bf16x8*)(iAl + off);
      } else {
        ah[i] = frag_col(iAh, 0, wm * 64 + i * 16, lane);
        if (PASSES > 1) al[i] = frag_col(iAl, 0, wm * 64 + i * 16, lane);
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (BMODE == 0) {
        const int off = row32_img_off(wn * (NJ * 16) + j * 16 + (lane & 15), lane >> 4);
        bh[j] = *(const bf16x8*)(iBh + off);
        if (PASSES > 1) bl[j] = *(const bf16x8*)(iBl + off);
      } else {
        bh[j] = frag_col(iBh, 0, wn * (NJ * 16) + j * 16, lane);
        if (PASSES > 1) bl[j] = frag_col(iBl, 0, wn * (NJ * 16) + j * 16, lane);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (VEC) {         // same terms in the same order, operands swapped
          if (PASSES > 1) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        } else {
          if (PASSES > 1) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
      }
    if (more) {
      char* nb = smem + (cur ^ 1) * 32768;
      stagef_store<AMODE, PASSES, TM>(sa, nb, nb + 8192, tid);
      stagef_store<BMODE, PASSES>(sb, nb + 16384, nb + 24576, tid);
    }
    __syncthreads();
  }
  bool add_bias = ksl == 0;
  if (p.splitk_work && p.split_k > 1) {           // deterministic split-K: the last slice of the tile sums and stores
    if (!splitk_gather_acc<NJ>(p, acc, wgid, ksl, tid)) return;
    add_bias = true;
  }
  if constexpr (VEC) x256_quadrant(p, acc, m0 + wm * 64, n0 + wn * (NJ * 16), lane);
  else mfma_epilogue(p, acc, m0, n0, wm, wn, lane, add_bias);
}

// skinny (M <= 32) variant: fp32 fragments straight from memory, split in registers (common.h).
// MT = 16-row tiles per workgroup (2: all 32 rows; 1: grid.y picks the half), NC = columns per workgroup
// (16, or 8 = half a tile: 4x the workgroups with half the operand bytes each when N is small - these
// launches are latency bound and a handful of workgroups leaves the load queues of most CUs idle).
template <int PASSES, int MT, int NC>
__global__ __launch_bounds__(SKW * 64) void gemm_skinny_f32_kernel(ns_gemm_params p) {
  __shared__ float red[SKW][16 * MT][20];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * NC;
  const int m0 = blockIdx.y * 16 * MT;
  const float* A = (const float*)p.A;
  const float* B = (const float*)p.B;
  const int r16 = lane & 15, g = lane >> 4;
  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nkc = (p.K + 31) / 32;
  const bool ok0 = m0 + r16 < p.M, ok1 = MT > 1 && m0 + 16 + r16 < p.M, okn = r16 < NC && n0 + r16 < p.N;
  for (int kc0 = wave; kc0 < nkc; kc0 += SKW * 2) {
    bf16x8 ah[2][MT], al[2][MT], bh[2], bl[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = (kc0 + q * SKW) * 32 + g * 8;
      const bool okk = k < p.K;
      ldsplit8(A + (long)(m0 + r16) * p.lda + k, ok0 && okk, ah[q][0], al[q][0]);
      if (MT > 1) ldsplit8(A + (long)(m0 + 16 + r16) * p.lda + k, ok1 && okk, ah[q][MT - 1], al[q][MT - 1]);
      ldsplit8(B + (long)(n0 + r16) * p.ldb + k, okn && okk, bh[q], bl[q]);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i] = mfma_split<PASSES>(ah[q][i], al[q][i], bh[q], bl[q], acc[i]);
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][i * 16 + g * 4 + r][r16] = acc[i][r];
  __syncthreads();
  Epi e = make_epi(p);
  for (int idx = tid; idx < 16 * MT * NC; idx += SKW * 64) {
    const int mm = idx / NC, nn = idx % NC;
    const int m = m0 + mm, n = n0 + nn;
    if (m >= p.M || n >= p.N) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SKW; ++w) v += red[w][mm][nn];
    const bool valid = row_valid(e, m);
    v = epi_value(e, m, n, v, true, valid);
    epi_store(e, m, n, v);
  }
}

// ------------------------------------------------------------------ host dispatch
static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// which kernel the last ns_gemm call of this thread launched (bench.py groups its HIP-event timings by it, so that
// its per-kernel averages can be read against the rocprofv3 kernel statistics)
static thread_local const char* g_last_kernel = "";
extern "C" const char* ns_gemm_last_kernel(void) { return g_last_kernel; }

// the 256-tile kernel: large k-contiguous bf16 products whose tiles fill the chip
// the vector epilogue (x256_quadrant): 16-byte (fp32) / 8-byte (bf16) accesses of C and of every per-element epilogue
// operand, four consecutive columns per lane, plain stores
static bool vec_epilogue_ok(const ns_gemm_params& p) {
  auto al = [](const void* q, long ld, bool bf16) { return !q || ((((uintptr_t)q) & (bf16 ? 7 : 15)) == 0 && ld % 4 == 0); };
  if (p.accumulate == 2 || p.split_k != 1 || p.N % 4 != 0) return false;
  if (p.batch > 1 && p.batch_stride_c % 4 != 0) return false;
  return al(p.C, p.ldc, p.c_dtype == NS_BF16) && al(p.addend, p.ld_add, p.addend_dtype == NS_BF16) &&
         al(p.gate, p.ld_gate, p.dtype == NS_BF16) && al(p.stat_z, p.ld_stat_z, p.stat_z_dtype == NS_BF16) &&
         al(p.bias, 4, false) && al(p.stat_mean, 4, false) && al(p.stat_istd, 4, false) && al(p.stat_part, 4, false);
}
// 64 x 128 tiles instead of 128 x 128: when the 128-row tiling leaves CUs idle and halving the tile height adds workgroups
static bool half_tiles_wanted(const ns_gemm_params& p, int tiles128) {
  static const int env = [] { const char* e = getenv("NS_GEMM_HALF"); return e ? atoi(e) : -1; }();
  if (env == 0 || env == 1) return env != 0;
  return tiles128 * p.batch <= (env > 1 ? env : 256) && p.M > 64;      // (NS_GEMM_HALF = n > 1: the threshold; 256: the halves still fit two per CU)
}
static bool x256_ok(const ns_gemm_params& p) {
  if (p.dtype != NS_BF16 || p.a_mode != 0 || p.b_mode != 0 || p.split_k != 1) return false;
  if (p.b_seg_len != 0 && (p.b_seg_len % 64 != 0 || p.b_seg_stride % 8 != 0)) return false;
  if (p.K % 64 != 0 || p.K < 128 || p.M < 1024 || p.N % 128 != 0) return false;
  if ((p.A_lo != nullptr) != (p.B_lo != nullptr)) return false;
  if (p.A_lo && (!aligned16(p.A_lo) || !aligned16(p.B_lo))) return false;
  {   // the vector epilogue: 16-byte (fp32) / 8-byte (bf16) accesses of C and of every per-element epilogue operand
    auto al = [](const void* q, long ld, bool bf16) { return !q || ((((uintptr_t)q) & (bf16 ? 7 : 15)) == 0 && ld % 4 == 0); };
    if (!al(p.C, p.ldc, p.c_dtype == NS_BF16) || !al(p.addend, p.ld_add, p.addend_dtype == NS_BF16) ||
        !al(p.gate, p.ld_gate, true) || !al(p.stat_z, p.ld_stat_z, p.stat_z_dtype == NS_BF16) || !al(p.bias, 4, false) ||
        !al(p.stat_mean, 4, false) || !al(p.stat_istd, 4, false) || !al(p.stat_part, 4, false)) return false;
  }
  if ((double)p.M * (double)p.lda * 2.0 >= 4.0e9 || (double)p.N * (double)p.ldb * 2.0 >= 4.0e9) return false;
  if (getenv("NS_GEMM_NO256")) return false;
  // at least ~3/4 of the CUs busy, or the 128-tile kernel's finer grain wins
  return ceil_div(p.M, 256) * ceil_div(p.N, 256) >= 96;
}

extern "C" size_t ns_gemm_stat_part_floats(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (size_t)2 * 4 * ceil_div(M, 256) * (size_t)N;     // the 256-tile kernel has the most slots per row
}

static int gemm_dispatch(ns_gemm_params& p, hipStream_t stream);

// the fixed-order second stage on its own (ns_bn_bwd's fallback reduction and its bias-gradient partials use it too)
int ns_stats_finalize(const float* part, int slots, int N, float* s1, float* s2, hipStream_t stream) {
  hipLaunchKernelGGL(gemm_stats_finalize_kernel, dim3(ceil_div(N, 32)), dim3(1024), 0, stream, part, slots, N, s1, s2);
  NS_CHECK_LAUNCH("gemm_stats_finalize");
  return NS_OK;
}

extern "C" size_t ns_gemm_splitk_work_bytes(int M, int N, int split_k) {
  if (split_k <= 1) return 0;
  // a (tile, slice) slot holds the tile's elements as fp32; the largest tiles of a split-K kernel are 128 x 128
  return (size_t)split_k * ((size_t)ceil_div(M, 128) * 128) * ((size_t)ceil_div(N, 128) * 128) * sizeof(float);
}
extern "C" size_t ns_gemm_splitk_counters(int M, int N) { return (size_t)ceil_div(M, 64) * ceil_div(N, 64); }

extern "C" int ns_gemm(const ns_gemm_params* pp, ns_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NS_CHECK_ARG(pp != nullptr, "ns_gemm: null params");
  ns_gemm_params p = *pp;
  NS_CHECK_ARG(!p.col_sumsq || p.col_sum, "ns_gemm: col_sumsq needs col_sum");
  NS_CHECK_ARG(!p.col_sum || p.stat_part, "ns_gemm: col_sum needs the stat_part scratch (ns_gemm_stat_part_floats)");
  NS_CHECK_ARG(!p.stat_z || (p.col_sum && p.col_sumsq && p.stat_mean && p.stat_istd && p.accumulate != 2 && p.split_k <= 1),
               "ns_gemm: stat_z needs col_sum, col_sumsq, stat_mean, stat_istd and a non-atomic store");
  if (!p.col_sum) p.stat_part = nullptr;
  p.stat_slots = 0;
  int rc = gemm_dispatch(p, stream);
  if (rc || !p.stat_part || p.M == 0 || p.N == 0) return rc;
  hipLaunchKernelGGL(gemm_stats_finalize_kernel, dim3(ceil_div(p.N, 32)), dim3(1024), 0, stream, p.stat_part, p.stat_slots,
                     p.N, p.col_sum, p.col_sumsq);
  NS_CHECK_LAUNCH("gemm_stats_finalize");
  return NS_OK;
}

static int gemm_dispatch(ns_gemm_params& p, hipStream_t stream) {
  NS_CHECK_ARG(p.M >= 0 && p.N >= 0 && p.K >= 0, "ns_gemm: negative dims");
  if (p.M == 0 || p.N == 0) return NS_OK;
  NS_CHECK_ARG(p.A && p.B && p.C, "ns_gemm: null operand");
  NS_CHECK_ARG(p.dtype == NS_F32 || p.dtype == NS_BF16, "ns_gemm: bad dtype %d", p.dtype);
  NS_CHECK_ARG(p.c_dtype == NS_F32 || p.c_dtype == NS_BF16, "ns_gemm: bad c_dtype %d", p.c_dtype);
  if (p.split_k < 1) p.split_k = 1;
  NS_CHECK_ARG((p.splitk_work != nullptr) == (p.splitk_count != nullptr), "ns_gemm: splitk_work and splitk_count come together");
  NS_CHECK_ARG(!p.splitk_work || p.split_k == 1 || p.batch <= 1, "ns_gemm: deterministic split-K takes no batch");
  NS_CHECK_ARG(!p.splitk_work || ns_gemm_splitk_work_bytes(p.M, p.N, p.split_k) < 0x7ffffff0ull,
               "ns_gemm: deterministic split-K scratch beyond 2 GB (32-bit buffer offsets): lower split_k");
  NS_CHECK_ARG(p.accumulate >= 0 && p.accumulate <= 2, "ns_gemm: bad accumulate");
  NS_CHECK_ARG(p.accumulate == 0 || p.c_dtype == NS_F32, "ns_gemm: accumulate needs fp32 C");
  NS_CHECK_ARG(p.split_k == 1 || (p.accumulate == 2 && p.act == NS_ACT_NONE && !p.col_sum),
               "ns_gemm: split_k>1 needs atomic accumulate, no activation, no stats");
  NS_CHECK_ARG(p.b_seg_len == 0 || (p.b_seg_len > 0 && p.K % p.b_seg_len == 0),
               "ns_gemm: K must be a multiple of b_seg_len");
  if (p.alpha == 0.f) p.alpha = 1.f;
  if (p.batch < 1) p.batch = 1;
  NS_CHECK_ARG(p.batch == 1 || (!p.col_sum && !p.bias && !p.addend && !p.gate && !p.A_lo && p.batch <= 65535),
               "ns_gemm: batched calls take no bias / addend / gate / statistics / pre-split operands");

  bool fast = (p.dtype == NS_BF16) && aligned16(p.A) && aligned16(p.B) && (p.lda % 8 == 0) &&
              (p.ldb % 8 == 0) && (p.b_seg_stride % 8 == 0) && (p.batch_stride_a % 8 == 0) &&
              (p.batch_stride_b % 8 == 0);
  if (fast) {
    // contiguous-dim extents must be whole 16-B chunks
    if (p.a_mode == 0) fast = fast && (p.K % 8 == 0); else fast = fast && (p.M % 8 == 0);
    if (p.b_mode == 0) fast = fast && (p.K % 8 == 0); else fast = fast && (p.N % 8 == 0);
    if (p.b_seg_len > 0) fast = fast && (p.b_seg_len % GBK == 0);
  }
  if (fast && p.M <= 32 && p.a_mode == 0 && p.b_mode == 0 && p.b_seg_len == 0 && p.split_k == 1 &&
      !p.col_sum && p.batch == 1) {
    // enough workgroups to spread the weight stream over the chip
    g_last_kernel = "gemm_skinny_kernel";
    if (p.N >= 32 * 128) hipLaunchKernelGGL(gemm_skinny_kernel<2>, dim3(ceil_div(p.N, 32)), dim3(SKW * 64), 0, stream, p);
    else hipLaunchKernelGGL(gemm_skinny_kernel<1>, dim3(ceil_div(p.N, 16)), dim3(SKW * 64), 0, stream, p);
    NS_CHECK_LAUNCH("gemm_skinny");
    return NS_OK;
  }
  if (fast && p.batch == 1 && x256_ok(p)) {
    const int tiles = ceil_div(p.M, 256) * ceil_div(p.N, 256);
    const size_t lds = 2 * XBUF;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)gemm_x256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)gemm_x256_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)gemm_x256_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    p.stat_slots = 4 * ceil_div(p.M, 256);
    const bool two = p.A_lo && p.f32_passes == 2;
    g_last_kernel = p.A_lo ? (two ? "gemm_x256_kernel<2>" : "gemm_x256_kernel<3>") : "gemm_x256_kernel<1>";
    if (two) hipLaunchKernelGGL(gemm_x256_kernel<2>, dim3(tiles), dim3(512), lds, stream, p);
    else if (p.A_lo) hipLaunchKernelGGL(gemm_x256_kernel<3>, dim3(tiles), dim3(512), lds, stream, p);
    else hipLaunchKernelGGL(gemm_x256_kernel<1>, dim3(tiles), dim3(512), lds, stream, p);
    NS_CHECK_LAUNCH("gemm_x256");
    return NS_OK;
  }
  NS_CHECK_ARG(!p.A_lo && !p.B_lo, "ns_gemm: pre-split operands (A_lo / B_lo) need the 256-tile path "
               "(bf16, a_mode 0, b_mode 0, K %% 64 == 0, M >= 1024, N %% 128 == 0, >= 96 tiles, no split_k)");
  if (fast) {
    const int tiles = ceil_div(p.M, GBM) * ceil_div(p.N, GBN);
    dim3 grid(tiles, p.split_k, p.batch);
    const size_t lds = 65536;
    p.stat_slots = 2 * ceil_div(p.M, GBM);
    const bool vec = vec_epilogue_ok(p);
    // 64-row tiles when the 128-row ones leave CUs without a workgroup (<= 192 tiles) and A is k-contiguous
    const bool half = vec && p.a_mode == 0 && half_tiles_wanted(p, tiles);
    if (half) {
      grid.x = ceil_div(p.M, 64) * ceil_div(p.N, GBN);
      p.stat_slots = ceil_div(p.M, 64);
    }
#define LAUNCH_MFMA(AM, BM_)                                                                      \
  do {                                                                                            \
    static bool attr_set = false;                                                                 \
    if (!attr_set) {                                                                              \
      (void)hipFuncSetAttribute((const void*)gemm_mfma_kernel<AM, BM_, false>,                        \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
      (void)hipFuncSetAttribute((const void*)gemm_mfma_kernel<AM, BM_, true>,                         \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
      attr_set = true;                                                                            \
    }                                                                                             \
    g_last_kernel = vec ? "gemm_mfma_kernel<" #AM ", " #BM_ ", true, 64, 128>" : "gemm_mfma_kernel<" #AM ", " #BM_ ", false, 64, 128>";  \
    if (vec) hipLaunchKernelGGL((gemm_mfma_kernel<AM, BM_, true>), grid, dim3(256), lds, stream, p);  \
    else hipLaunchKernelGGL((gemm_mfma_kernel<AM, BM_, false>), grid, dim3(256), lds, stream, p);     \
  } while (0)
#define LAUNCH_MFMA_HALF(BM_)                                                                     \
  do {                                                                                            \
    static bool attr_set = false;                                                                 \
    if (!attr_set) {                                                                              \
      (void)hipFuncSetAttribute((const void*)gemm_mfma_kernel<0, BM_, true, 64, 64>,                  \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
      attr_set = true;                                                                            \
    }                                                                                             \
    g_last_kernel = "gemm_mfma_kernel<0, " #BM_ ", true, 64, 64>";                                \
    hipLaunchKernelGGL((gemm_mfma_kernel<0, BM_, true, 64, 64>), grid, dim3(256), lds, stream, p);    \
  } while (0)
    static const int bk32_env = [] { const char* e = getenv("NS_GEMM_BK32"); return e ? atoi(e) : -1; }();
    const bool bk32 = bk32_env >= 0 ? bk32_env != 0 : (long)tiles * p.split_k * p.batch >= 600;
    // BK = 32 (four workgroups per CU instead of two) pays when a launch has more workgroups than the 512 the BK = 64
    // form keeps resident: 1280 workgroups 202 -> 169 us, 640: 73 -> 52 us; at <= 512 (what the models' split-K rule
    // asks for) it is slower alone (84 -> 99 us) and no faster beside another stream's kernels
    // (profiles/r03_gemm_128_ablation.txt).  NS_GEMM_BK32 = 0 / 1 forces it off / on.
    if (half) {
      if (p.b_mode == 0) LAUNCH_MFMA_HALF(0); else LAUNCH_MFMA_HALF(1);
    } else if (p.a_mode == 1 && p.b_mode == 1 && bk32 && (p.b_seg_len == 0 || p.b_seg_len % 32 == 0)) {
      g_last_kernel = vec ? "gemm_mfma_kernel<1, 1, true, 32, 128>" : "gemm_mfma_kernel<1, 1, false, 32, 128>";
      if (vec) hipLaunchKernelGGL((gemm_mfma_kernel<1, 1, true, 32>), grid, dim3(256), 32768, stream, p);
      else hipLaunchKernelGGL((gemm_mfma_kernel<1, 1, false, 32>), grid, dim3(256), 32768, stream, p);
    } else if (p.a_mode == 0 && p.b_mode == 0) LAUNCH_MFMA(0, 0);
    else if (p.a_mode == 0 && p.b_mode == 1) LAUNCH_MFMA(0, 1);
    else if (p.a_mode == 1 && p.b_mode == 0) LAUNCH_MFMA(1, 0);
    else LAUNCH_MFMA(1, 1);
#undef LAUNCH_MFMA_HALF
#undef LAUNCH_MFMA
    NS_CHECK_LAUNCH("gemm_mfma");
    return NS_OK;
  }
  if (p.dtype == NS_F32 && p.f32_passes > 0) {
    bool ok = aligned16(p.A) && aligned16(p.B) && (p.lda % 4 == 0) && (p.ldb % 4 == 0) && (p.b_seg_stride % 4 == 0) &&
              (p.batch_stride_a % 4 == 0) && (p.batch_stride_b % 4 == 0);
    if (p.a_mode == 0) ok = ok && (p.K % 4 == 0); else ok = ok && (p.M % 4 == 0);
    if (p.b_mode == 0) ok = ok && (p.K % 4 == 0); else ok = ok && (p.N % 4 == 0);
    if (p.b_seg_len > 0) ok = ok && (p.b_seg_len % FBK == 0);
    const bool three = p.f32_passes >= 3;
    if (ok && p.M <= 32 && p.a_mode == 0 && p.b_mode == 0 && p.b_seg_len == 0 && p.split_k == 1 && !p.col_sum &&
        p.K % 8 == 0 && p.lda % 4 == 0 && p.batch == 1) {
      g_last_kernel = "gemm_skinny_f32_kernel";
      if (ceil_div(p.N, 16) <= 64) {      // few column tiles: 16 rows x 8 columns per workgroup
        const dim3 grid(ceil_div(p.N, 8), ceil_div(p.M, 16));
        if (three) hipLaunchKernelGGL((gemm_skinny_f32_kernel<3, 1, 8>), grid, dim3(SKW * 64), 0, stream, p);
        else hipLaunchKernelGGL((gemm_skinny_f32_kernel<1, 1, 8>), grid, dim3(SKW * 64), 0, stream, p);
      } else {
        if (three) hipLaunchKernelGGL((gemm_skinny_f32_kernel<3, 2, 16>), dim3(ceil_div(p.N, 16)), dim3(SKW * 64), 0, stream, p);
        else hipLaunchKernelGGL((gemm_skinny_f32_kernel<1, 2, 16>), dim3(ceil_div(p.N, 16)), dim3(SKW * 64), 0, stream, p);
      }
      NS_CHECK_LAUNCH("gemm_skinny_f32");
      return NS_OK;
    }
    if (ok) {
      const int tiles = ceil_div(p.M, GBM) * ceil_div(p.N, GBN);
      dim3 grid(tiles, p.split_k, p.batch);
      const size_t lds = 65536;
      p.stat_slots = 2 * ceil_div(p.M, GBM);
      const bool vec = vec_epilogue_ok(p);
      const bool half = vec && p.a_mode == 0 && half_tiles_wanted(p, tiles);
      if (half) {
        grid.x = ceil_div(p.M, 64) * ceil_div(p.N, GBN);
        p.stat_slots = ceil_div(p.M, 64);
      }
#define LAUNCH_F32(AM, BM_, PS)                                                                     \
  do {                                                                                              \
    static bool attr_set = false;                                                                   \
    if (!attr_set) {                                                                                \
      (void)hipFuncSetAttribute((const void*)gemm_mfma_f32_kernel<AM, BM_, PS, false>,              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
      (void)hipFuncSetAttribute((const void*)gemm_mfma_f32_kernel<AM, BM_, PS, true>,               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
      attr_set = true;                                                                              \
    }                                                                                               \
    g_last_kernel = vec ? "gemm_mfma_f32_kernel<" #AM ", " #BM_ ", " #PS ", true, 128>"             \
                        : "gemm_mfma_f32_kernel<" #AM ", " #BM_ ", " #PS ", false, 128>";           \
    if (vec) hipLaunchKernelGGL((gemm_mfma_f32_kernel<AM, BM_, PS, true>), grid, dim3(256), lds, stream, p);   \
    else hipLaunchKernelGGL((gemm_mfma_f32_kernel<AM, BM_, PS, false>), grid, dim3(256), lds, stream, p);      \
  } while (0)
#define LAUNCH_F32_HALF(BM_, PS)                                                                    \
  do {                                                                                              \
    static bool attr_set = false;                                                                   \
    if (!attr_set) {                                                                                \
      (void)hipFuncSetAttribute((const void*)gemm_mfma_f32_kernel<0, BM_, PS, true, 64>,            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
      attr_set = true;                                                                              \
    }                                                                                               \
    g_last_kernel = "gemm_mfma_f32_kernel<0, " #BM_ ", " #PS ", true, 64>";                         \
    hipLaunchKernelGGL((gemm_mfma_f32_kernel<0, BM_, PS, true, 64>), grid, dim3(256), lds, stream, p);         \
  } while (0)
#define LAUNCH_F32_MODES(PS)                                                   \
  do {                                                                         \
    if (half && p.b_mode == 0) LAUNCH_F32_HALF(0, PS);                         \
    else if (half) LAUNCH_F32_HALF(1, PS);                                     \
    else if (p.a_mode == 0 && p.b_mode == 0) LAUNCH_F32(0, 0, PS);             \
    else if (p.a_mode == 0 && p.b_mode == 1) LAUNCH_F32(0, 1, PS);             \
    else if (p.a_mode == 1 && p.b_mode == 0) LAUNCH_F32(1, 0, PS);             \
    else LAUNCH_F32(1, 1, PS);                                                 \
  } while (0)
      if (three) LAUNCH_F32_MODES(3); else LAUNCH_F32_MODES(1);
#undef LAUNCH_F32_HALF
#undef LAUNCH_F32_MODES
#undef LAUNCH_F32
      NS_CHECK_LAUNCH("gemm_mfma_f32");
      return NS_OK;
    }
  }
  dim3 grid(ceil_div(p.N, 64), ceil_div(p.M, 64), p.split_k * p.batch);
  p.stat_slots = ceil_div(p.M, 64);
  g_last_kernel = "gemm_generic_kernel";
  if (p.dtype == NS_F32) hipLaunchKernelGGL(gemm_generic_kernel<float>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(gemm_generic_kernel<bf16_t>, grid, dim3(256), 0, stream, p);
  NS_CHECK_LAUNCH("gemm_generic");
  return NS_OK;
}
