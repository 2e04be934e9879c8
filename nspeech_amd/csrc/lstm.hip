// LSTM recurrences (LSTMBlockCell semantics: gate order i,j,f,o, forget_bias added at compute
// time).  One fused kernel per time step and direction pair:
//   forward : gates = xg[t] + h[t-1].Wh for 16 hidden units (= 64 gate columns) x 32 batch rows
//             per workgroup, K dealt in 32-wide chunks to 8 wavefronts that issue all of their
//             16-byte fragment loads before the first MFMA (the step is L2-latency bound), partial
//             sums meet in LDS, then the cell update;
//   backward: dh = dh_out + dgates[t+1].Wh^T for 16 units, then the cell gradient, same shape.
// The input products x.Wx and every weight gradient are hoisted by the caller into big GEMMs.
#include "common.h"
#include "lstm_step.h"

constexpr int LW = 8;          // waves per workgroup
constexpr int LTHREADS = LW * 64;
constexpr int GROUP = 4;       // K chunks whose loads are in flight together, per wave

__device__ __forceinline__ bf16x8 zero8() { return (bf16x8){0, 0, 0, 0, 0, 0, 0, 0}; }

// ------------------------------------------------------------------ fused forward step
// HALF = false: 16 units x 32 rows per workgroup; HALF = true: 8 units x 16 rows (4x the workgroups, half the
// operand bytes each: wide cells at small batch, where the big tiling fills only a quarter of the CUs).
template <typename T, bool HALF>
__global__ __launch_bounds__(LTHREADS) void lstm_step_kernel(LstmStepPair<T> pp) {
  constexpr int UPW = HALF ? 8 : 16, RPW = HALF ? 16 : 32, MT = RPW / 16, NTL = UPW / 4;
  __shared__ float red[LW][RPW][NTL * 16 + 4];   // row stride = 4 mod 16 floats: the MFMA C layout writes rows g*4+q from four lane groups, +1 would put them 4 banks apart
  const LstmStep<T>& a = pp.s[blockIdx.z];
  const int tid = threadIdx.x;
  const int u0 = blockIdx.x * UPW;
  const int nb = blockIdx.y * RPW;
  const int H = a.H;
  // epilogue operands of this thread's (row, unit): issued now so that their memory latency
  // overlaps the weight / state fragment loads instead of following the LDS reduction
  const int er = tid / UPW, euu = tid % UPW;
  const int en = nb + er, eu = u0 + euu;
  const bool eok = tid < RPW * UPW && en < a.N && eu < H;
  float pz[4] = {0.f, 0.f, 0.f, 0.f};
  float pcp = 0.f, php = 0.f;
  bool pmask = false;
  if (eok) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (a.xg) pz[j] = a.xg[(long)en * a.xg_sn + (long)j * H + eu];
      if (a.bias) pz[j] += a.bias[j * H + eu];
    }
    if (a.c_prev) pcp = a.c_prev[(long)en * a.c_sn + eu];
    pmask = a.lengths && a.t >= a.lengths[en];
    if (a.zmode && a.hp) php = ldf(a.hp + (long)en * a.hp_sn + eu);
  }

  if constexpr (sizeof(T) == 2) {
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.a) {
      const int nkc = (a.K + 31) / 32;
      const bf16_t* arow0 = a.a + (long)(nb + r16) * a.a_sn;
      const bf16_t* arow1 = a.a + (long)(nb + 16 + r16) * a.a_sn;
      const bool ok0 = nb + r16 < a.N, ok1 = MT > 1 && nb + 16 + r16 < a.N;
      // tile j, column r16 -> (gate, unit): 16 units: gate j; 8 units: gates 2j, 2j+1
      const int bu = u0 + (HALF ? (r16 & 7) : r16), bg0 = HALF ? (r16 >> 3) : 0;
      const bool oku = bu < H;
      const long gstride = (long)H * a.K * (HALF ? 2 : 1);
      const bf16_t* brow = a.wT + ((long)bg0 * H + bu) * a.K;
      for (int kc0 = wave; kc0 < nkc; kc0 += LW * GROUP) {
        bf16x8 af[GROUP][MT], bfr[GROUP][NTL];
#pragma unroll
        for (int q = 0; q < GROUP; ++q) {
          const int k = (kc0 + q * LW) * 32 + g * 8;
          const bool okk = k < a.K;
          af[q][0] = (ok0 && okk) ? *(const bf16x8*)(arow0 + k) : zero8();
          if (MT > 1) af[q][MT - 1] = (ok1 && okk) ? *(const bf16x8*)(arow1 + k) : zero8();
#pragma unroll
          for (int j = 0; j < NTL; ++j) bfr[q][j] = (oku && okk) ? *(const bf16x8*)(brow + j * gstride + k) : zero8();
        }
#pragma unroll
        for (int q = 0; q < GROUP; ++q)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[q][i], bfr[q][j], acc[i][j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][i * 16 + g * 4 + r][j * 16 + r16] = acc[i][j][r];
  } else if (a.passes > 0) {
    // fp32 state / weights on the bf16 MFMA via the hi/lo split (common.h)
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const bool three = a.passes >= 3, two = a.passes == 2;
    constexpr int QG = HALF ? 4 : 2;       // K chunks in flight per wave
    f32x4 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.a) {
      const int nkc = (a.K + 31) / 32;
      const float* arow0 = (const float*)a.a + (long)(nb + r16) * a.a_sn;
      const float* arow1 = (const float*)a.a + (long)(nb + 16 + r16) * a.a_sn;
      const bool ok0 = nb + r16 < a.N, ok1 = MT > 1 && nb + 16 + r16 < a.N;
      const int bu = u0 + (HALF ? (r16 & 7) : r16), bg0 = HALF ? (r16 >> 3) : 0;
      const bool oku = bu < H;
      const long gstride = (long)H * a.K * (HALF ? 2 : 1);
      const long bo0 = ((long)bg0 * H + bu) * a.K;
      const float* brow = (const float*)a.wT + bo0;
      for (int kc0 = wave; kc0 < nkc; kc0 += LW * QG) {
        bf16x8 ah[QG][MT], al[QG][MT], bh[QG][NTL], bl[QG][NTL];
#pragma unroll
        for (int q = 0; q < QG; ++q) {
          const int k = (kc0 + q * LW) * 32 + g * 8;
          const bool okk = k < a.K;
          ldsplit8(arow0 + k, ok0 && okk, ah[q][0], al[q][0]);
          if (MT > 1) ldsplit8(arow1 + k, ok1 && okk, ah[q][MT - 1], al[q][MT - 1]);
          if (a.wT_hi) {
            const long bo = bo0 + k;
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
              bh[q][j] = (oku && okk) ? *(const bf16x8*)(a.wT_hi + bo + j * gstride) : zero8();
              bl[q][j] = (oku && okk && three) ? *(const bf16x8*)(a.wT_lo + bo + j * gstride) : zero8();
            }
          } else {
#pragma unroll
            for (int j = 0; j < NTL; ++j) ldsplit8(brow + j * gstride + k, oku && okk, bh[q][j], bl[q][j]);
          }
        }
#pragma unroll
        for (int q = 0; q < QG; ++q)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j)
              acc[i][j] = three ? mfma_split<3>(ah[q][i], al[q][i], bh[q][j], bl[q][j], acc[i][j])
                          : two ? mfma_split<2>(ah[q][i], al[q][i], bh[q][j], bl[q][j], acc[i][j])
                                : mfma_split<1>(ah[q][i], al[q][i], bh[q][j], bl[q][j], acc[i][j]);
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][i * 16 + g * 4 + r][j * 16 + r16] = acc[i][j][r];
  } else {
    // exact fp32 path (parity tests): thread = (row, 4 gate columns)
    constexpr int TPR = UPW;                 // threads per row, 4 gate columns each (column = gate * UPW + unit)
    const int r = tid / TPR, cb = (tid % TPR) * 4;
    const int n = nb + r;
    if (tid < RPW * TPR) {
      float s[4] = {0, 0, 0, 0};
      if (a.a && n < a.N) {
        const float* arow = (const float*)a.a + (long)n * a.a_sn;
        for (int k = 0; k < a.K; ++k) {
          const float av = arow[k];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = cb + q, j = col / UPW, u = u0 + (col % UPW);
            if (u < H) s[q] = fmaf(av, ((const float*)a.wT)[((long)j * H + u) * a.K + k], s[q]);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        red[0][r][cb + q] = s[q];
#pragma unroll
        for (int w = 1; w < LW; ++w) red[w][r][cb + q] = 0.f;
      }
    }
  }
  __syncthreads();

  if (eok) {
    const int r = er, uu = euu, n = en, u = eu;
    float z[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = pz[j];
#pragma unroll
      for (int w = 0; w < LW; ++w) v += red[w][r][j * UPW + uu];
      z[j] = v;
    }
    const bool masked = pmask;
    const float cp = pcp;
    const float gi = sigmoidf_(z[0]), gj = tanhf_(z[1]), gf = sigmoidf_(z[2] + a.forget_bias), go = sigmoidf_(z[3]);
    float c = ns_cell_clip(gf * cp + gi * gj, a.cell_clip);
    float h = go * tanhf_(c);
    if (a.zmode == 1) {          // zoneout, training: the unit keeps its old value where the mask says so
      if (ns_zone_keep(a.zseed_c, (uint32_t)a.t, (uint32_t)n, (uint32_t)u, a.zthr_c)) c = cp;
      if (ns_zone_keep(a.zseed_h, (uint32_t)a.t, (uint32_t)n, (uint32_t)u, a.zthr_h)) h = php;
    } else if (a.zmode == 2) {   // zoneout, inference: the expectation of the above
      c = a.zc * cp + (1.f - a.zc) * c;
      h = a.zh * php + (1.f - a.zh) * h;
    }
    if (masked) { c = 0.f; h = 0.f; }
    a.c_out[(long)n * a.co_sn + u] = c;
    stf(a.h_out + (long)n * a.h_sn + u, h);
    if (a.h_out2) stf(a.h_out2 + (long)n * a.h2_sn + u, h);
    if (a.gates_out) {
      T* gp = a.gates_out + (long)n * a.g_sn;
      stf(gp + u, masked ? 0.f : gi);
      stf(gp + H + u, masked ? 0.f : gj);
      stf(gp + 2 * H + u, masked ? 0.f : gf);
      stf(gp + 3 * H + u, masked ? 0.f : go);
    }
  }
}

template <typename T>
int lstm_step_launch2(const LstmStepPair<T>& a, hipStream_t s) {
  // few big tiles leave most CUs idle: switch to the small tiling when it still gives at most ~2 workgroups per CU
  const int big = ceil_div(a.s[0].H, 16) * ceil_div(a.s[0].N, 32) * a.n;
  if (a.s[0].H >= 256 && big <= 128) {
    dim3 grid(ceil_div(a.s[0].H, 8), ceil_div(a.s[0].N, 16), a.n);
    hipLaunchKernelGGL((lstm_step_kernel<T, true>), grid, dim3(LTHREADS), 0, s, a);
  } else {
    dim3 grid(ceil_div(a.s[0].H, 16), ceil_div(a.s[0].N, 32), a.n);
    hipLaunchKernelGGL((lstm_step_kernel<T, false>), grid, dim3(LTHREADS), 0, s, a);
  }
  NS_CHECK_LAUNCH("lstm_step");
  return NS_OK;
}
template <typename T>
int lstm_step_launch(const LstmStep<T>& a, hipStream_t s) {
  LstmStepPair<T> pp;
  pp.s[0] = a; pp.s[1] = a; pp.n = 1;
  return lstm_step_launch2<T>(pp, s);
}
template int lstm_step_launch<float>(const LstmStep<float>&, hipStream_t);
template int lstm_step_launch<bf16_t>(const LstmStep<bf16_t>&, hipStream_t);
template int lstm_step_launch2<float>(const LstmStepPair<float>&, hipStream_t);
template int lstm_step_launch2<bf16_t>(const LstmStepPair<bf16_t>&, hipStream_t);

// ------------------------------------------------------------------ fused backward step
template <typename T, bool HALF>
__global__ __launch_bounds__(LTHREADS) void lstm_bwd_step_kernel(LstmBwdStepPair<T> pp) {
  constexpr int UPW = HALF ? 8 : 16, RPW = HALF ? 16 : 32, MT = RPW / 16;
  __shared__ float red[LW][RPW][20];
  const LstmBwdStep<T>& a = pp.s[blockIdx.z];
  const int tid = threadIdx.x;
  const int u0 = blockIdx.x * UPW;
  const int nb = blockIdx.y * RPW;
  const int H = a.H;
  // epilogue operands first (see the forward kernel)
  const int er = tid / UPW, euu = tid % UPW;
  const int en = nb + er, eu = u0 + euu;
  const bool eok = tid < RPW * UPW && en < a.N && eu < H;
  float pdh = 0.f, pgi = 0.f, pgj = 0.f, pgf = 0.f, pgo = 0.f, pc = 0.f, pcp = 0.f, pdc = 0.f, pdhc = 0.f;
  bool pmask = false;
  if (eok) {
    pmask = a.lengths && a.t >= a.lengths[en];
    if (a.dh_out) pdh += a.dh_out[(long)en * a.dho_sn + eu];
    if (a.dh_out2) pdh += a.dh_out2[(long)en * a.dho2_sn + eu];
    const T* gp = a.gates + (long)en * a.g_sn;
    pgi = ldf(gp + eu); pgj = ldf(gp + H + eu); pgf = ldf(gp + 2 * H + eu); pgo = ldf(gp + 3 * H + eu);
    pc = a.c[(long)en * a.c_sn + eu];
    if (a.c_prev) pcp = a.c_prev[(long)en * a.c_sn + eu];
    if (!a.first) pdc = a.dc_carry[(long)en * H + eu];
    if (a.zmode && !a.first) pdhc = a.dh_carry[(long)en * H + eu];
  }

  if constexpr (sizeof(T) == 2) {
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.dg_next) {
      const int nkc = (a.K + 31) / 32;
      const bf16_t* arow0 = a.dg_next + (long)(nb + r16) * a.dgn_sn;
      const bf16_t* arow1 = a.dg_next + (long)(nb + 16 + r16) * a.dgn_sn;
      const bool ok0 = nb + r16 < a.N, ok1 = MT > 1 && nb + 16 + r16 < a.N;
      const bool oku = r16 < UPW && u0 + r16 < H;
      const bf16_t* brow = a.w + (long)(u0 + r16) * a.K;
      for (int kc0 = wave; kc0 < nkc; kc0 += LW * GROUP) {
        bf16x8 af[GROUP][MT], bfr[GROUP];
#pragma unroll
        for (int q = 0; q < GROUP; ++q) {
          const int k = (kc0 + q * LW) * 32 + g * 8;
          const bool okk = k < a.K;
          af[q][0] = (ok0 && okk) ? *(const bf16x8*)(arow0 + k) : zero8();
          if (MT > 1) af[q][MT - 1] = (ok1 && okk) ? *(const bf16x8*)(arow1 + k) : zero8();
          bfr[q] = (oku && okk) ? *(const bf16x8*)(brow + k) : zero8();
        }
#pragma unroll
        for (int q = 0; q < GROUP; ++q)
#pragma unroll
          for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[q][i], bfr[q], acc[i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][i * 16 + g * 4 + r][r16] = acc[i][r];
  } else if (a.passes > 0) {
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const bool three = a.passes >= 3;
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.dg_next) {
      const int nkc = (a.K + 31) / 32;
      const float* arow0 = (const float*)a.dg_next + (long)(nb + r16) * a.dgn_sn;
      const float* arow1 = (const float*)a.dg_next + (long)(nb + 16 + r16) * a.dgn_sn;
      const bool ok0 = nb + r16 < a.N, ok1 = MT > 1 && nb + 16 + r16 < a.N;
      const bool oku = r16 < UPW && u0 + r16 < H;
      const float* brow = (const float*)a.w + (long)(u0 + r16) * a.K;
      if (a.w_bf16 && !three) {
        const bf16_t* wrow = a.w_bf16 + (long)(u0 + r16) * a.K;
        const bf16_t* brow0 = a.dg_next_b ? a.dg_next_b + (long)(nb + r16) * a.dgn_sn : nullptr;
        const bf16_t* brow1 = a.dg_next_b ? a.dg_next_b + (long)(nb + 16 + r16) * a.dgn_sn : nullptr;
        for (int kc0 = wave; kc0 < nkc; kc0 += LW * GROUP) {
          bf16x8 af[GROUP][MT], bfr[GROUP];
#pragma unroll
          for (int q = 0; q < GROUP; ++q) {
            const int k = (kc0 + q * LW) * 32 + g * 8;
            const bool okk = k < a.K;
            if (brow0) {
              af[q][0] = (ok0 && okk) ? *(const bf16x8*)(brow0 + k) : zero8();
              if (MT > 1) af[q][MT - 1] = (ok1 && okk) ? *(const bf16x8*)(brow1 + k) : zero8();
            } else {
              bf16x8 dummy;
              ldsplit8(arow0 + k, ok0 && okk, af[q][0], dummy);
              if (MT > 1) ldsplit8(arow1 + k, ok1 && okk, af[q][MT - 1], dummy);
            }
            bfr[q] = (oku && okk) ? *(const bf16x8*)(wrow + k) : zero8();
          }
#pragma unroll
          for (int q = 0; q < GROUP; ++q)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[q][i], bfr[q], acc[i], 0, 0, 0);
        }
      } else
      for (int kc0 = wave; kc0 < nkc; kc0 += LW * 2) {
        bf16x8 ah[2][MT], al[2][MT], bh[2], bl[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int k = (kc0 + q * LW) * 32 + g * 8;
          const bool okk = k < a.K;
          ldsplit8(arow0 + k, ok0 && okk, ah[q][0], al[q][0]);
          if (MT > 1) ldsplit8(arow1 + k, ok1 && okk, ah[q][MT - 1], al[q][MT - 1]);
          ldsplit8(brow + k, oku && okk, bh[q], bl[q]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < MT; ++i)
            acc[i] = three ? mfma_split<3>(ah[q][i], al[q][i], bh[q], bl[q], acc[i])
                           : mfma_split<1>(ah[q][i], al[q][i], bh[q], bl[q], acc[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][i * 16 + g * 4 + r][r16] = acc[i][r];
  } else {
    const int r = er, uu = euu;
    const int n = nb + r, u = u0 + uu;
    if (tid < RPW * UPW) {
      float s = 0.f;
      if (a.dg_next && n < a.N && u < H) {
        const float* arow = (const float*)a.dg_next + (long)n * a.dgn_sn;
        const float* brow = (const float*)a.w + (long)u * a.K;
        for (int k = 0; k < a.K; ++k) s = fmaf(arow[k], brow[k], s);
      }
      red[0][r][uu] = s;
#pragma unroll
      for (int w = 1; w < LW; ++w) red[w][r][uu] = 0.f;
    }
  }
  __syncthreads();

  if (!eok) return;
  const int r = er, uu = euu, n = en, u = eu;
  T* dg = a.dgates + (long)n * a.dg_sn;
  const long ci = (long)n * H + u;
  bf16_t* dgb = a.dgates_b ? a.dgates_b + (long)n * a.dg_sn : nullptr;
  if (pmask) {
    stf(dg + u, 0.f); stf(dg + H + u, 0.f); stf(dg + 2 * H + u, 0.f); stf(dg + 3 * H + u, 0.f);
    if (dgb) { dgb[u] = (bf16_t)0.f; dgb[H + u] = (bf16_t)0.f; dgb[2 * H + u] = (bf16_t)0.f; dgb[3 * H + u] = (bf16_t)0.f; }
    a.dc_carry[ci] = 0.f;
    if (a.zmode) a.dh_carry[ci] = 0.f;
    return;
  }
  float dh = pdh;
#pragma unroll
  for (int w = 0; w < LW; ++w) dh += red[w][r][uu];
  const float gi = pgi, gj = pgj, gf = pgf, go = pgo, cp = pcp;
  float c = pc, dcin = pdc, dckeep = 0.f;
  if (a.zmode) {
    // h[t] = m_h ? h[t-1] : h'[t]: the gradient of a kept unit passes to step t-1 untouched; c[t] likewise, and the
    // saved c is the ZONED state, so tanh needs the plain cell's c' = f c[t-1] + i j again
    dh += pdhc;
    const bool mh = ns_zone_keep(a.zseed_h, (uint32_t)a.t, (uint32_t)n, (uint32_t)u, a.zthr_h);
    const bool mc = ns_zone_keep(a.zseed_c, (uint32_t)a.t, (uint32_t)n, (uint32_t)u, a.zthr_c);
    a.dh_carry[ci] = mh ? dh : 0.f;
    if (mh) dh = 0.f;
    if (mc) { dckeep = dcin; dcin = 0.f; }
    c = gf * cp + gi * gj;
  }
  const float tc = tanhf_(c);
  const float d_o = dh * tc * go * (1.f - go);
  const float dc = dh * go * (1.f - tc * tc) + dcin;
  const float d_i = dc * gj * gi * (1.f - gi);
  const float d_j = dc * gi * (1.f - gj * gj);
  const float d_f = dc * cp * gf * (1.f - gf);
  a.dc_carry[ci] = dc * gf + dckeep;
  stf(dg + u, d_i); stf(dg + H + u, d_j); stf(dg + 2 * H + u, d_f); stf(dg + 3 * H + u, d_o);
  if (dgb) { dgb[u] = (bf16_t)d_i; dgb[H + u] = (bf16_t)d_j; dgb[2 * H + u] = (bf16_t)d_f; dgb[3 * H + u] = (bf16_t)d_o; }
}

template <typename T>
int lstm_bwd_step_launch2(const LstmBwdStepPair<T>& a, hipStream_t s) {
  const int big = ceil_div(a.s[0].H, 16) * ceil_div(a.s[0].N, 32) * a.n;
  if (a.s[0].H >= 256 && big <= 128) {
    dim3 grid(ceil_div(a.s[0].H, 8), ceil_div(a.s[0].N, 16), a.n);
    hipLaunchKernelGGL((lstm_bwd_step_kernel<T, true>), grid, dim3(LTHREADS), 0, s, a);
  } else {
    dim3 grid(ceil_div(a.s[0].H, 16), ceil_div(a.s[0].N, 32), a.n);
    hipLaunchKernelGGL((lstm_bwd_step_kernel<T, false>), grid, dim3(LTHREADS), 0, s, a);
  }
  NS_CHECK_LAUNCH("lstm_bwd_step");
  return NS_OK;
}
template <typename T>
int lstm_bwd_step_launch(const LstmBwdStep<T>& a, hipStream_t s) {
  LstmBwdStepPair<T> pp;
  pp.s[0] = a; pp.s[1] = a; pp.n = 1;
  return lstm_bwd_step_launch2<T>(pp, s);
}
template int lstm_bwd_step_launch<float>(const LstmBwdStep<float>&, hipStream_t);
template int lstm_bwd_step_launch<bf16_t>(const LstmBwdStep<bf16_t>&, hipStream_t);
template int lstm_bwd_step_launch2<float>(const LstmBwdStepPair<float>&, hipStream_t);
template int lstm_bwd_step_launch2<bf16_t>(const LstmBwdStepPair<bf16_t>&, hipStream_t);

// ------------------------------------------------------------------ host time loops
extern "C" size_t ns_lstm_seq_work_bytes(const ns_lstm_seq_params* p) {
  if (!p) return 0;
  const bool zone = p->zoneout_thr_cell || p->zoneout_thr_output;      // + the dh carry through kept units
  return sizeof(float) * (size_t)p->N * p->H * (zone ? 2 : 1) + 256;
}

template <typename T>
static void fill_fwd(LstmStep<T>& a, const ns_lstm_seq_params& p, int step) {
  const long P = p.P, H = p.H;
  const int t = p.reverse ? p.T - 1 - step : step;
  const int tp = p.reverse ? t + 1 : t - 1;
  const long row = p.padl + t, rowp = p.padl + tp;
  const bool has_prev = rowp >= 0 && rowp < P && step > 0;
  a = LstmStep<T>{};
  a.N = p.N; a.H = p.H; a.K = p.H; a.forget_bias = p.forget_bias; a.cell_clip = p.cell_clip;
  a.a = has_prev ? (const T*)p.h + rowp * p.ld_h : nullptr;
  a.a_sn = P * p.ld_h;
  a.wT = (const T*)p.whT;
  a.xg = p.xg + row * p.ld_xg; a.xg_sn = P * p.ld_xg;
  a.c_prev = has_prev ? p.c + rowp * H : nullptr; a.c_sn = P * H;
  a.h_out = (T*)p.h + row * p.ld_h; a.h_sn = P * p.ld_h;
  a.c_out = p.c + row * H; a.co_sn = P * H;
  a.gates_out = p.gates ? (T*)p.gates + row * 4 * H : nullptr; a.g_sn = P * 4 * H;
  a.lengths = p.lengths; a.t = t;
  a.passes = p.f32_passes;
  a.wT_hi = (const bf16_t*)p.whT_hi; a.wT_lo = (const bf16_t*)p.whT_lo;
  if (p.zoneout_thr_cell || p.zoneout_thr_output) {
    a.zmode = 1;
    a.zthr_c = p.zoneout_thr_cell; a.zthr_h = p.zoneout_thr_output;
    a.zseed_c = p.zoneout_seed_cell; a.zseed_h = p.zoneout_seed_output;
    a.hp = has_prev ? (const T*)p.h + rowp * p.ld_h : nullptr; a.hp_sn = P * p.ld_h;
  }
}

template <typename T>
static void fill_bwd(LstmBwdStep<T>& a, const ns_lstm_seq_params& p, int step, float* dc_carry) {
  const long P = p.P, H = p.H;
  const int t = p.reverse ? p.T - 1 - step : step;
  const int tp = p.reverse ? t + 1 : t - 1;   // forward-pass predecessor
  const int tn = p.reverse ? t - 1 : t + 1;   // forward-pass successor (whose dgates feed dh)
  const long row = p.padl + t, rowp = p.padl + tp, rown = p.padl + tn;
  const bool has_prev = step > 0 && rowp >= 0 && rowp < P;
  const bool has_next = step < p.T - 1;
  a = LstmBwdStep<T>{};
  a.N = p.N; a.H = p.H; a.t = t; a.lengths = p.lengths; a.K = 4 * p.H;
  a.first = has_next ? 0 : 1;
  a.dg_next = has_next ? (const T*)p.dgates + rown * 4 * H : nullptr; a.dgn_sn = P * 4 * H;
  a.w = (const T*)p.wh;
  a.dh_out = p.dh + row * p.ld_dh; a.dho_sn = P * p.ld_dh;
  a.gates = (const T*)p.gates + row * 4 * H; a.g_sn = P * 4 * H;
  a.c = p.c + row * H; a.c_sn = P * H;
  a.c_prev = has_prev ? p.c + rowp * H : nullptr;
  a.dc_carry = dc_carry;
  a.dgates = (T*)p.dgates + row * 4 * H; a.dg_sn = P * 4 * H;
  a.passes = p.f32_passes;
  a.w_bf16 = (const bf16_t*)p.wh_bf16;
  if (p.dgates_bf16 && sizeof(T) == 4) {
    a.dgates_b = (bf16_t*)p.dgates_bf16 + row * 4 * H;
    a.dg_next_b = has_next ? (const bf16_t*)p.dgates_bf16 + rown * 4 * H : nullptr;
  }
  if (p.zoneout_thr_cell || p.zoneout_thr_output) {
    a.zmode = 1;
    a.zthr_c = p.zoneout_thr_cell; a.zthr_h = p.zoneout_thr_output;
    a.zseed_c = p.zoneout_seed_cell; a.zseed_h = p.zoneout_seed_output;
    a.dh_carry = dc_carry + (long)p.N * H;
  }
}

static int check_fwd(const ns_lstm_seq_params* p, const char* who) {
  NS_CHECK_ARG(p && p->xg && p->whT && p->h && p->c, "%s: null", who);
  NS_CHECK_ARG(p->H % 16 == 0 && p->ld_h % 8 == 0, "%s: H %% 16 and ld_h %% 8 required", who);
  NS_CHECK_ARG(p->padl + p->T <= p->P, "%s: P too small", who);
  return NS_OK;
}
static int check_bwd(const ns_lstm_seq_params* p, const char* who) {
  NS_CHECK_ARG(p && p->wh && p->gates && p->c && p->dh && p->dgates && p->work, "%s: null", who);
  NS_CHECK_ARG(p->H % 16 == 0, "%s: H %% 16 required", who);
  return NS_OK;
}

template <typename T>
static int seq_fwd_t(const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1, hipStream_t s) {
  for (int step = 0; step < p0->T; ++step) {
    LstmStepPair<T> pp;
    pp.n = p1 ? 2 : 1;
    fill_fwd<T>(pp.s[0], *p0, step);
    if (p1) fill_fwd<T>(pp.s[1], *p1, step); else pp.s[1] = pp.s[0];
    int rc = lstm_step_launch2<T>(pp, s);
    if (rc) return rc;
  }
  return NS_OK;
}
template <typename T>
static int seq_bwd_t(const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1, hipStream_t s) {
  for (int step = p0->T - 1; step >= 0; --step) {
    LstmBwdStepPair<T> pp;
    pp.n = p1 ? 2 : 1;
    fill_bwd<T>(pp.s[0], *p0, step, p0->work);
    if (p1) fill_bwd<T>(pp.s[1], *p1, step, p1->work); else pp.s[1] = pp.s[0];
    int rc = lstm_bwd_step_launch2<T>(pp, s);
    if (rc) return rc;
  }
  return NS_OK;
}

extern "C" int ns_lstm_seq_fwd(const ns_lstm_seq_params* p, ns_stream_t s) {
  int rc = check_fwd(p, "ns_lstm_seq_fwd");
  if (rc) return rc;
  if (p->dtype == NS_BF16) return seq_fwd_t<bf16_t>(p, nullptr, (hipStream_t)s);
  return seq_fwd_t<float>(p, nullptr, (hipStream_t)s);
}
extern "C" int ns_lstm_seq_bwd(const ns_lstm_seq_params* p, ns_stream_t s) {
  int rc = check_bwd(p, "ns_lstm_seq_bwd");
  if (rc) return rc;
  if (p->dtype == NS_BF16) return seq_bwd_t<bf16_t>(p, nullptr, (hipStream_t)s);
  return seq_bwd_t<float>(p, nullptr, (hipStream_t)s);
}
// Two independent recurrences with identical N/T/H (the two directions of a BiLSTM) advanced
// together, one launch per time step.
extern "C" int ns_lstm_seq2_fwd(const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1, ns_stream_t s) {
  int rc = check_fwd(p0, "ns_lstm_seq2_fwd");
  if (rc) return rc;
  rc = check_fwd(p1, "ns_lstm_seq2_fwd");
  if (rc) return rc;
  NS_CHECK_ARG(p0->N == p1->N && p0->T == p1->T && p0->H == p1->H && p0->dtype == p1->dtype,
               "ns_lstm_seq2_fwd: the two recurrences must share N, T, H, dtype");
  if (p0->dtype == NS_BF16) return seq_fwd_t<bf16_t>(p0, p1, (hipStream_t)s);
  return seq_fwd_t<float>(p0, p1, (hipStream_t)s);
}
extern "C" int ns_lstm_seq2_bwd(const ns_lstm_seq_params* p0, const ns_lstm_seq_params* p1, ns_stream_t s) {
  int rc = check_bwd(p0, "ns_lstm_seq2_bwd");
  if (rc) return rc;
  rc = check_bwd(p1, "ns_lstm_seq2_bwd");
  if (rc) return rc;
  NS_CHECK_ARG(p0->N == p1->N && p0->T == p1->T && p0->H == p1->H && p0->dtype == p1->dtype,
               "ns_lstm_seq2_bwd: the two recurrences must share N, T, H, dtype");
  if (p0->dtype == NS_BF16) return seq_bwd_t<bf16_t>(p0, p1, (hipStream_t)s);
  return seq_bwd_t<float>(p0, p1, (hipStream_t)s);
}

extern "C" int ns_lstm_step(const ns_lstm_step_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->a && p->wT && p->h_out && p->c_out, "ns_lstm_step: null");
  NS_CHECK_ARG(p->H % 16 == 0 && p->K % 8 == 0 && p->a_sn % 8 == 0, "ns_lstm_step: H %% 16, K %% 8, a_sn %% 8 required");
  auto run = [&](auto tag) -> int {
    using T = decltype(tag);
    LstmStep<T> a = {};
    a.N = p->N; a.H = p->H; a.K = p->K; a.forget_bias = p->forget_bias; a.cell_clip = p->cell_clip; a.passes = p->f32_passes;
    a.a = (const T*)p->a; a.a_sn = p->a_sn; a.wT = (const T*)p->wT;
    a.xg = p->xg; a.xg_sn = p->xg_sn; a.bias = p->bias;
    a.c_prev = p->c_prev; a.c_sn = p->c_sn;
    a.h_out = (T*)p->h_out; a.h_sn = p->h_sn; a.h_out2 = (T*)p->h_out2; a.h2_sn = p->h2_sn;
    a.c_out = p->c_out; a.co_sn = p->co_sn;
    a.wT_hi = (const bf16_t*)p->wT_hi; a.wT_lo = (const bf16_t*)p->wT_lo;
    if (p->zoneout_cell > 0.f || p->zoneout_output > 0.f) {
      a.zmode = 2; a.zc = p->zoneout_cell; a.zh = p->zoneout_output;
      a.hp = (const T*)p->h_prev; a.hp_sn = p->hp_sn;
    }
    return lstm_step_launch<T>(a, (hipStream_t)s);
  };
  if (p->dtype == NS_BF16) return run(bf16_t{});
  return run(float{});
}
