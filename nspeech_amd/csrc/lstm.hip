// LSTM recurrences (LSTMBlockCell semantics: gate order i,j,f,o, forget_bias added at compute
// time).  One fused kernel per time step: the recurrent product h[t-1].Wh for 16 hidden units
// (= 64 gate columns) x 32 batch rows per workgroup, K split over the 4 waves, then the cell
// update in the same launch.  The input product x.Wx is hoisted by the caller into one big GEMM.
#include "common.h"
#include "lstm_step.h"

// ------------------------------------------------------------------ fused forward step
template <typename T>
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmStep<T> a) {
  __shared__ float red[4][32][65];
  const int tid = threadIdx.x;
  const int u0 = blockIdx.x * 16;
  const int nb = blockIdx.y * 32;
  const int H = a.H;

  if constexpr (sizeof(T) == 2) {
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.a) {
      const int nkc = (a.K + 31) / 32;
#pragma unroll 2
      for (int kc = wave; kc < nkc; kc += 4) {
        const int k = kc * 32 + g * 8;
        bf16x8 af[2], bfr[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int n = nb + i * 16 + r16;
          if (n < a.N && k < a.K) af[i] = *(const bf16x8*)(a.a + (long)n * a.a_sn + k);
          else af[i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int u = u0 + r16;
          if (u < H && k < a.K) bfr[j] = *(const bf16x8*)(a.wT + ((long)j * H + u) * a.K + k);
          else bfr[j] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][i * 16 + g * 4 + r][j * 16 + r16] = acc[i][j][r];
  } else {
    // exact fp32 path (parity tests): thread = (row, 8 gate columns)
    const int r = tid >> 3, cb = (tid & 7) * 8;
    const int n = nb + r;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (a.a && n < a.N) {
      const float* arow = (const float*)a.a + (long)n * a.a_sn;
      for (int k = 0; k < a.K; ++k) {
        const float av = arow[k];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int col = cb + q, j = col >> 4, u = u0 + (col & 15);
          if (u < H) s[q] = fmaf(av, ((const float*)a.wT)[((long)j * H + u) * a.K + k], s[q]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      red[0][r][cb + q] = s[q];
      red[1][r][cb + q] = 0.f; red[2][r][cb + q] = 0.f; red[3][r][cb + q] = 0.f;
    }
  }
  __syncthreads();

  for (int idx = tid; idx < 32 * 16; idx += 256) {
    const int r = idx >> 4, uu = idx & 15;
    const int n = nb + r, u = u0 + uu;
    if (n >= a.N || u >= H) continue;
    float z[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = red[0][r][j * 16 + uu] + red[1][r][j * 16 + uu] + red[2][r][j * 16 + uu] + red[3][r][j * 16 + uu];
      if (a.xg) v += a.xg[(long)n * a.xg_sn + (long)j * H + u];
      if (a.bias) v += a.bias[j * H + u];
      z[j] = v;
    }
    const bool masked = a.lengths && a.t >= a.lengths[n];
    const float cp = a.c_prev ? a.c_prev[(long)n * a.c_sn + u] : 0.f;
    const float gi = sigmoidf_(z[0]), gj = tanhf_(z[1]), gf = sigmoidf_(z[2] + a.forget_bias), go = sigmoidf_(z[3]);
    float c = gf * cp + gi * gj;
    float h = go * tanhf_(c);
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
int lstm_step_launch(const LstmStep<T>& a, hipStream_t s) {
  dim3 grid(ceil_div(a.H, 16), ceil_div(a.N, 32));
  hipLaunchKernelGGL(lstm_step_kernel<T>, grid, dim3(256), 0, s, a);
  NS_CHECK_LAUNCH("lstm_step");
  return NS_OK;
}
template int lstm_step_launch<float>(const LstmStep<float>&, hipStream_t);
template int lstm_step_launch<bf16_t>(const LstmStep<bf16_t>&, hipStream_t);

// ------------------------------------------------------------------ backward cell update
template <typename T>
__global__ void lstm_bwd_cell_kernel(LstmBwdCell<T> a) {
  const int total = a.N * a.H;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int n = idx / a.H, u = idx % a.H;
    const int H = a.H;
    T* dg = a.dgates + (long)n * a.dg_sn;
    const bool masked = a.lengths && a.t >= a.lengths[n];
    if (masked) {
      stf(dg + u, 0.f); stf(dg + H + u, 0.f); stf(dg + 2 * H + u, 0.f); stf(dg + 3 * H + u, 0.f);
      a.dc_carry[idx] = 0.f;
      continue;
    }
    float dh = 0.f;
    if (a.dh_out) dh += a.dh_out[(long)n * a.dho_sn + u];
    if (a.dh_out2) dh += a.dh_out2[(long)n * a.dho2_sn + u];
    if (a.dh_carry) dh += a.dh_carry[(long)n * a.dhc_sn + u];
    const T* gp = a.gates + (long)n * a.g_sn;
    const float gi = ldf(gp + u), gj = ldf(gp + H + u), gf = ldf(gp + 2 * H + u), go = ldf(gp + 3 * H + u);
    const float c = a.c[(long)n * a.c_sn + u];
    const float cp = a.c_prev ? a.c_prev[(long)n * a.c_sn + u] : 0.f;
    const float tc = tanhf_(c);
    const float d_o = dh * tc * go * (1.f - go);
    float dc = dh * go * (1.f - tc * tc);
    if (!a.first) dc += a.dc_carry[idx];
    const float d_i = dc * gj * gi * (1.f - gi);
    const float d_j = dc * gi * (1.f - gj * gj);
    const float d_f = dc * cp * gf * (1.f - gf);
    a.dc_carry[idx] = dc * gf;
    stf(dg + u, d_i); stf(dg + H + u, d_j); stf(dg + 2 * H + u, d_f); stf(dg + 3 * H + u, d_o);
  }
}
template <typename T>
int lstm_bwd_cell_launch(const LstmBwdCell<T>& a, hipStream_t s) {
  const int total = a.N * a.H;
  hipLaunchKernelGGL(lstm_bwd_cell_kernel<T>, dim3(ceil_div(total, 256)), dim3(256), 0, s, a);
  NS_CHECK_LAUNCH("lstm_bwd_cell");
  return NS_OK;
}
template int lstm_bwd_cell_launch<float>(const LstmBwdCell<float>&, hipStream_t);
template int lstm_bwd_cell_launch<bf16_t>(const LstmBwdCell<bf16_t>&, hipStream_t);

// ------------------------------------------------------------------ host time loops
extern "C" size_t ns_lstm_seq_work_bytes(const ns_lstm_seq_params* p) {
  if (!p) return 0;
  return sizeof(float) * 2 * (size_t)p->N * p->H + 256;
}

template <typename T>
static int lstm_seq_fwd_t(const ns_lstm_seq_params& p, hipStream_t s) {
  const long P = p.P, H = p.H;
  for (int step = 0; step < p.T; ++step) {
    const int t = p.reverse ? p.T - 1 - step : step;
    const int tp = p.reverse ? t + 1 : t - 1;
    const long row = p.padl + t, rowp = p.padl + tp;
    const bool has_prev = rowp >= 0 && rowp < P && step > 0;
    LstmStep<T> a = {};
    a.N = p.N; a.H = p.H; a.K = p.H; a.forget_bias = p.forget_bias;
    a.a = has_prev ? (const T*)p.h + rowp * p.ld_h : nullptr;
    a.a_sn = P * p.ld_h;
    a.wT = (const T*)p.whT;
    a.xg = p.xg + row * p.ld_xg; a.xg_sn = P * p.ld_xg;
    a.c_prev = has_prev ? p.c + rowp * H : nullptr; a.c_sn = P * H;
    a.h_out = (T*)p.h + row * p.ld_h; a.h_sn = P * p.ld_h;
    a.c_out = p.c + row * H; a.co_sn = P * H;
    a.gates_out = p.gates ? (T*)p.gates + row * 4 * H : nullptr; a.g_sn = P * 4 * H;
    a.lengths = p.lengths; a.t = t;
    int rc = lstm_step_launch<T>(a, s);
    if (rc) return rc;
  }
  return NS_OK;
}

extern "C" int ns_lstm_seq_fwd(const ns_lstm_seq_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->xg && p->whT && p->h && p->c, "ns_lstm_seq_fwd: null");
  NS_CHECK_ARG(p->H % 16 == 0 && p->ld_h % 8 == 0, "ns_lstm_seq_fwd: H %% 16 and ld_h %% 8 required");
  NS_CHECK_ARG(p->padl + p->T <= p->P, "ns_lstm_seq_fwd: P too small");
  if (p->dtype == NS_BF16) return lstm_seq_fwd_t<bf16_t>(*p, (hipStream_t)s);
  return lstm_seq_fwd_t<float>(*p, (hipStream_t)s);
}

template <typename T>
static int lstm_seq_bwd_t(const ns_lstm_seq_params& p, hipStream_t s) {
  const long P = p.P, H = p.H;
  float* dh_carry = p.work;
  float* dc_carry = p.work + (size_t)p.N * H;
  // walk the forward order backwards
  for (int step = p.T - 1; step >= 0; --step) {
    const int t = p.reverse ? p.T - 1 - step : step;
    const int tp = p.reverse ? t + 1 : t - 1;  // forward-pass predecessor
    const long row = p.padl + t, rowp = p.padl + tp;
    const bool has_prev = step > 0 && rowp >= 0 && rowp < P;
    LstmBwdCell<T> a = {};
    a.N = p.N; a.H = p.H; a.t = t; a.lengths = p.lengths;
    a.first = (step == p.T - 1);
    a.dh_out = p.dh + row * p.ld_dh; a.dho_sn = P * p.ld_dh;
    a.dh_carry = a.first ? nullptr : dh_carry; a.dhc_sn = H;
    a.gates = (const T*)p.gates + row * 4 * H; a.g_sn = P * 4 * H;
    a.c = p.c + row * H; a.c_sn = P * H;
    a.c_prev = has_prev ? p.c + rowp * H : nullptr;
    a.dc_carry = dc_carry;
    a.dgates = (T*)p.dgates + row * 4 * H; a.dg_sn = P * 4 * H;
    int rc = lstm_bwd_cell_launch<T>(a, s);
    if (rc) return rc;
    if (step > 0) {
      // dh_carry[N,H] = dgates[t] . Wh^T   (Wh natural [H,4H] is k-contiguous for this product)
      for (int nb = 0; nb < p.N; nb += 32) {
        ns_gemm_params g = {};
        g.dtype = p.dtype; g.M = min(32, p.N - nb); g.N = p.H; g.K = 4 * p.H;
        g.A = (const T*)p.dgates + (row + (long)nb * P) * 4 * H; g.lda = P * 4 * H; g.a_mode = 0;
        g.B = p.wh; g.ldb = 4 * H; g.b_mode = 0;
        g.C = dh_carry + (size_t)nb * H; g.ldc = H; g.c_dtype = NS_F32;
        g.alpha = 1.f; g.split_k = 1;
        rc = ns_gemm(&g, s);
        if (rc) return rc;
      }
    }
  }
  return NS_OK;
}

extern "C" int ns_lstm_seq_bwd(const ns_lstm_seq_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->wh && p->gates && p->c && p->dh && p->dgates && p->work, "ns_lstm_seq_bwd: null");
  NS_CHECK_ARG(p->H % 16 == 0, "ns_lstm_seq_bwd: H %% 16 required");
  if (p->dtype == NS_BF16) return lstm_seq_bwd_t<bf16_t>(*p, (hipStream_t)s);
  return lstm_seq_bwd_t<float>(*p, (hipStream_t)s);
}
