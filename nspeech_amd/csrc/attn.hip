// Tacotron-2 attention RNN (prenet -> attention LSTM -> location-sensitive attention) for the
// teacher-forced decoder, forward and backward through time.  The matrix products are M=batch
// skinny GEMMs (ns_gemm), the LSTM cell is the fused step kernel of lstm.hip, and the
// energies / masked softmax / context and their gradients are the two kernels below: one
// workgroup per batch row, one wavefront per memory position, A/64 units per lane.
#include "common.h"
#include "lstm_step.h"

constexpr int MAXU = 4;   // units per lane (A <= 256)
constexpr int MAXKW = 8;

template <typename T>
struct AttnStep {
  int Ti, A, E, kw, Tia;
  int L;                         // unused (lengths read per row)
  const int* lengths;
  const float* keys; long keys_sn;     // row n base: keys + n*keys_sn, [Ti, A]
  const T* values; long values_sn;     // [Ti, E]
  const float* q; long q_sn;           // [A]
  const float* aprev; long al_sn;      // [Tia]
  float* aout;
  T* ctx_out; long ctx_sn;             // [E]
  T* ctx_out2; long ctx2_sn;           // optional second destination
  const float* wcl; const float* v;
};

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnStep<T> a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* ap = sm;                       // [Ti + 2*MAXKW]
  float* e = ap + a.Ti + 2 * MAXKW;     // [Ti]
  float* red = e + a.Ti;                // [32]
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = min(a.lengths ? a.lengths[n] : a.Ti, a.Ti);
  const int upl = a.A / 64, half = a.kw / 2 - ((a.kw & 1) ? 0 : 1);  // 'same' left pad = (kw-1)/2
  const float* aprev = a.aprev + (long)n * a.al_sn;
  for (int i = tid; i < a.Ti + 2 * MAXKW; i += 256) {
    const int t = i - MAXKW;
    ap[i] = (t >= 0 && t < a.Ti) ? aprev[t] : 0.f;
  }
  float q[MAXU], v[MAXU], w[MAXKW][MAXU];
#pragma unroll
  for (int j = 0; j < MAXU; ++j) {
    const int u = lane * upl + j;
    const bool ok = j < upl;
    q[j] = ok ? a.q[(long)n * a.q_sn + u] : 0.f;
    v[j] = ok ? a.v[u] : 0.f;
#pragma unroll
    for (int k = 0; k < MAXKW; ++k) w[k][j] = (ok && k < a.kw) ? a.wcl[k * a.A + u] : 0.f;
  }
  __syncthreads();
  const float* keys = a.keys + (long)n * a.keys_sn;
  for (int t = wave; t < a.Ti; t += 4) {
    if (t >= L) {
      if (lane == 0) e[t] = -INFINITY;
      continue;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXU; ++j) {
      if (j < upl) {
        float x = keys[(long)t * a.A + lane * upl + j] + q[j];
#pragma unroll
        for (int k = 0; k < MAXKW; ++k)
          if (k < a.kw) x = fmaf(ap[MAXKW + t + k - half], w[k][j], x);
        s = fmaf(v[j], tanhf_(x), s);
      }
    }
    s = wave_sum(s);
    if (lane == 0) e[t] = s;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = tid; t < L; t += 256) mx = fmaxf(mx, e[t]);
  mx = block_max(mx, red);
  float sum = 0.f;
  for (int t = tid; t < L; t += 256) sum += __expf(e[t] - mx);
  sum = block_sum(sum, red);
  const float inv = 1.f / sum;
  __syncthreads();
  float* aout = a.aout + (long)n * a.al_sn;
  for (int t = tid; t < a.Tia; t += 256) {
    const float al = t < L ? __expf(e[t] - mx) * inv : 0.f;
    if (t < a.Ti) e[t] = al;
    aout[t] = al;
  }
  __syncthreads();
  const T* values = a.values + (long)n * a.values_sn;
  for (int c = tid; c < a.E; c += 256) {
    float s = 0.f;
    for (int t = 0; t < L; ++t) s = fmaf(e[t], ldf(values + (long)t * a.E + c), s);
    stf(a.ctx_out + (long)n * a.ctx_sn + c, s);
    if (a.ctx_out2) stf(a.ctx_out2 + (long)n * a.ctx2_sn + c, s);
  }
}

template <typename T>
struct AttnBwdStep {
  int Ti, A, E, kw, Tia;
  const int* lengths;
  const float* keys; long keys_sn;
  const T* values; long values_sn;
  const float* q; long q_sn;
  const float* acur; const float* aprev; long al_sn;
  const float* dctx_ext; long dce_sn;      // [E] from downstream (dhc columns A..)
  const float* dctx_carry;                 // [N,E] or null
  float* dalign_carry;                     // [N,Tia] in: grad wrt acur from step s+1, out: grad wrt aprev
  int has_carry;
  T* dq_out; long dq_sn;
  float* dkeys; float* dvalues;            // row-n bases use keys_sn / values_sn
  float* dv; float* dwcl;
  const float* wcl; const float* v;
};

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnBwdStep<T> a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* ap = sm;                         // [Ti + 2*MAXKW]  previous alignments, padded
  float* dap = ap + a.Ti + 2 * MAXKW;     // [Ti + 2*MAXKW]  grad wrt previous alignments
  float* ac = dap + a.Ti + 2 * MAXKW;     // [Ti] current alignments
  float* da = ac + a.Ti;                  // [Ti]
  float* dctx = da + a.Ti;                // [E]
  float* red = dctx + a.E;                // [32]
  float* xr = red + 32;                   // [4][(2+MAXKW)*A] cross-wave reduction
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = min(a.lengths ? a.lengths[n] : a.Ti, a.Ti);
  const int upl = a.A / 64, half = a.kw / 2 - ((a.kw & 1) ? 0 : 1);
  const float* aprev = a.aprev + (long)n * a.al_sn;
  const float* acur = a.acur + (long)n * a.al_sn;
  float* dal = a.dalign_carry + (long)n * a.Tia;
  for (int i = tid; i < a.Ti + 2 * MAXKW; i += 256) {
    const int t = i - MAXKW;
    ap[i] = (t >= 0 && t < a.Ti) ? aprev[t] : 0.f;
    dap[i] = 0.f;
  }
  for (int t = tid; t < a.Ti; t += 256) ac[t] = acur[t];
  for (int c = tid; c < a.E; c += 256) {
    float d = a.dctx_ext[(long)n * a.dce_sn + c];
    if (a.dctx_carry) d += a.dctx_carry[(long)n * a.E + c];
    dctx[c] = d;
  }
  __syncthreads();
  // (1) context: da[t] = dctx . values[t] (+ carry);  dvalues[t] += a[t] * dctx
  const T* values = a.values + (long)n * a.values_sn;
  float* dvalues = a.dvalues + (long)n * a.values_sn;
  for (int t = wave; t < L; t += 4) {
    float s = 0.f;
    const float at = ac[t];
    for (int c = lane; c < a.E; c += 64) {
      const float dc = dctx[c];
      s = fmaf(dc, ldf(values + (long)t * a.E + c), s);
      dvalues[(long)t * a.E + c] += at * dc;
    }
    s = wave_sum(s);
    if (lane == 0) da[t] = s + (a.has_carry ? dal[t] : 0.f);
  }
  __syncthreads();
  // (2) softmax backward
  float dot = 0.f;
  for (int t = tid; t < L; t += 256) dot += ac[t] * da[t];
  dot = block_sum(dot, red);
  __syncthreads();
  for (int t = tid; t < L; t += 256) da[t] = ac[t] * (da[t] - dot);   // da now holds de
  __syncthreads();
  // (3) energies backward
  float q[MAXU], v[MAXU], w[MAXKW][MAXU];
  float dv[MAXU], dq[MAXU], dw[MAXKW][MAXU];
#pragma unroll
  for (int j = 0; j < MAXU; ++j) {
    const int u = lane * upl + j;
    const bool ok = j < upl;
    q[j] = ok ? a.q[(long)n * a.q_sn + u] : 0.f;
    v[j] = ok ? a.v[u] : 0.f;
    dv[j] = 0.f; dq[j] = 0.f;
#pragma unroll
    for (int k = 0; k < MAXKW; ++k) {
      w[k][j] = (ok && k < a.kw) ? a.wcl[k * a.A + u] : 0.f;
      dw[k][j] = 0.f;
    }
  }
  const float* keys = a.keys + (long)n * a.keys_sn;
  float* dkeys = a.dkeys + (long)n * a.keys_sn;
  for (int t = wave; t < L; t += 4) {
    const float de = da[t];
    float g[MAXKW];
#pragma unroll
    for (int k = 0; k < MAXKW; ++k) g[k] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXU; ++j) {
      if (j < upl) {
        const long ko = (long)t * a.A + lane * upl + j;
        float x = keys[ko] + q[j];
#pragma unroll
        for (int k = 0; k < MAXKW; ++k)
          if (k < a.kw) x = fmaf(ap[MAXKW + t + k - half], w[k][j], x);
        const float th = tanhf_(x);
        const float dpre = de * v[j] * (1.f - th * th);
        dv[j] = fmaf(de, th, dv[j]);
        dq[j] += dpre;
        dkeys[ko] += dpre;
#pragma unroll
        for (int k = 0; k < MAXKW; ++k)
          if (k < a.kw) {
            dw[k][j] = fmaf(ap[MAXKW + t + k - half], dpre, dw[k][j]);
            g[k] = fmaf(dpre, w[k][j], g[k]);
          }
      }
    }
#pragma unroll
    for (int k = 0; k < MAXKW; ++k)
      if (k < a.kw) {
        const float gs = wave_sum(g[k]);
        if (lane == 0) atomicAdd(&dap[MAXKW + t + k - half], gs);
      }
  }
  // cross-wave reduction of dv, dq, dw
  const int RS = (2 + MAXKW) * a.A;
#pragma unroll
  for (int j = 0; j < MAXU; ++j) {
    if (j < upl) {
      const int u = lane * upl + j;
      xr[wave * RS + u] = dv[j];
      xr[wave * RS + a.A + u] = dq[j];
#pragma unroll
      for (int k = 0; k < MAXKW; ++k) xr[wave * RS + (2 + k) * a.A + u] = dw[k][j];
    }
  }
  __syncthreads();
  for (int i = tid; i < (2 + a.kw) * a.A; i += 256) {
    const float s = xr[i] + xr[RS + i] + xr[2 * RS + i] + xr[3 * RS + i];
    if (i < a.A) atomicAdd(a.dv + i, s);
    else if (i < 2 * a.A) stf(a.dq_out + (long)n * a.dq_sn + (i - a.A), s);
    else atomicAdd(a.dwcl + (i - 2 * a.A), s);
  }
  for (int t = tid; t < a.Tia; t += 256) dal[t] = t < a.Ti ? dap[MAXKW + t] : 0.f;
}

// ------------------------------------------------------------------ host loops
static int gemm_small(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                      void* C, long ldc, int c_dtype, const float* bias, int act, const float* addend,
                      long ld_add, const void* gate, long ld_gate, hipStream_t s) {
  ns_gemm_params g = {};
  g.dtype = dtype; g.M = M; g.N = N; g.K = K;
  g.A = A; g.lda = lda; g.a_mode = 0;
  g.B = B; g.ldb = ldb; g.b_mode = 0;
  g.C = C; g.ldc = ldc; g.c_dtype = c_dtype;
  g.bias = bias; g.act = act; g.alpha = 1.f; g.split_k = 1;
  g.addend = addend; g.ld_add = ld_add; g.gate = gate; g.ld_gate = ld_gate;
  return ns_gemm(&g, s);
}

extern "C" size_t ns_taco2_attn_work_bytes(const ns_taco2_attn_params* p) {
  if (!p) return 0;
  // dctx_carry [N,E] + dalign_carry [N,Tia] + dhq [N,A] + dh_carry [N,A] + dc_carry [N,A]
  return sizeof(float) * ((size_t)p->N * (p->E + p->Tia + 3 * p->A)) + 256;
}

static int check_attn(const ns_taco2_attn_params* p, const char* who) {
  NS_CHECK_ARG(p != nullptr, "%s: null params", who);
  NS_CHECK_ARG(p->A % 64 == 0 && p->A <= 64 * MAXU, "%s: attention units must be a multiple of 64, <= 256", who);
  NS_CHECK_ARG(p->kw >= 1 && p->kw <= MAXKW, "%s: location filter width must be 1..%d", who, MAXKW);
  NS_CHECK_ARG(p->Tia >= p->Ti, "%s: Tia < Ti", who);
  NS_CHECK_ARG(p->N <= 32, "%s: batch per call must be <= 32 (shard the batch)", who);
  return NS_OK;
}

template <typename T>
static int attn_fwd_t(const ns_taco2_attn_params& p, hipStream_t s) {
  const long S1 = p.S + 1, A = p.A, E = p.E, D1 = p.D1, D2 = p.D2;
  const long XA = D2 + A, HC = A + E;
  const int dt = p.dtype;
  const size_t lds = sizeof(float) * (2 * p.Ti + 2 * MAXKW + 32);
  for (int st = 0; st < p.S; ++st) {
    const long slot = st + 1, prev = st;
    T* hc = (T*)p.hc; T* xa = (T*)p.xa; T* p1 = (T*)p.p1;
    int rc;
    // p1 = relu(ctx_prev . W1c + F1)
    rc = gemm_small(dt, p.N, D1, E, hc + prev * HC + A, S1 * HC, p.w1cT, E, p1 + slot * D1, S1 * D1, dt,
                    nullptr, NS_ACT_RELU, p.f1 + slot * D1, S1 * D1, nullptr, 0, s);
    if (rc) return rc;
    // p2 = relu(p1 . W2 + b2) -> xa[:, 0:D2]
    rc = gemm_small(dt, p.N, D2, D1, p1 + slot * D1, S1 * D1, p.w2T, D1, xa + slot * XA, S1 * XA, dt, p.b2,
                    NS_ACT_RELU, nullptr, 0, nullptr, 0, s);
    if (rc) return rc;
    // attention LSTM on [p2 | h_prev]
    LstmStep<T> l = {};
    l.N = p.N; l.H = p.A; l.K = (int)XA; l.forget_bias = 1.0f;
    l.a = xa + slot * XA; l.a_sn = S1 * XA; l.wT = (const T*)p.wattT; l.bias = p.batt;
    l.c_prev = st > 0 ? p.ca + prev * A : nullptr; l.c_sn = S1 * A;
    l.h_out = hc + slot * HC; l.h_sn = S1 * HC;
    if (st + 1 < p.S) { l.h_out2 = xa + (slot + 1) * XA + D2; l.h2_sn = S1 * XA; }
    l.c_out = p.ca + slot * A; l.co_sn = S1 * A;
    l.gates_out = (T*)p.ga + slot * 4 * A; l.g_sn = S1 * 4 * A;
    rc = lstm_step_launch<T>(l, s);
    if (rc) return rc;
    // q = h . Wq
    rc = gemm_small(dt, p.N, (int)A, (int)A, hc + slot * HC, S1 * HC, p.wqT, A, p.q + slot * A, S1 * A, NS_F32,
                    nullptr, NS_ACT_NONE, nullptr, 0, nullptr, 0, s);
    if (rc) return rc;
    AttnStep<T> a = {};
    a.Ti = p.Ti; a.A = p.A; a.E = p.E; a.kw = p.kw; a.Tia = p.Tia; a.lengths = p.lengths;
    a.keys = p.keys + (long)p.padl_i * A; a.keys_sn = (long)p.Pi * A;
    a.values = (const T*)p.values + (long)p.padl_i * E; a.values_sn = (long)p.Pi * E;
    a.q = p.q + slot * A; a.q_sn = S1 * A;
    a.aprev = p.align + prev * p.Tia; a.aout = p.align + slot * p.Tia; a.al_sn = S1 * p.Tia;
    a.ctx_out = hc + slot * HC + A; a.ctx_sn = S1 * HC;
    a.wcl = p.wcl; a.v = p.v;
    hipLaunchKernelGGL(attn_fwd_kernel<T>, dim3(p.N), dim3(256), lds, s, a);
    NS_CHECK_LAUNCH("attn_fwd");
  }
  return NS_OK;
}

extern "C" int ns_taco2_attn_fwd(const ns_taco2_attn_params* p, ns_stream_t s) {
  int rc = check_attn(p, "ns_taco2_attn_fwd");
  if (rc) return rc;
  NS_CHECK_ARG(p->keys && p->values && p->f1 && p->w1cT && p->w2T && p->wattT && p->wqT && p->b2 && p->batt &&
                   p->wcl && p->v && p->p1 && p->xa && p->hc && p->ca && p->ga && p->q && p->align,
               "ns_taco2_attn_fwd: null pointer");
  if (p->dtype == NS_BF16) return attn_fwd_t<bf16_t>(*p, (hipStream_t)s);
  return attn_fwd_t<float>(*p, (hipStream_t)s);
}

template <typename T>
static int attn_bwd_t(const ns_taco2_attn_params& p, hipStream_t s) {
  const long S1 = p.S + 1, A = p.A, E = p.E, D1 = p.D1, D2 = p.D2;
  const long XA = D2 + A, HC = A + E;
  const int dt = p.dtype;
  float* dctx_carry = p.work;
  float* dalign_carry = dctx_carry + (size_t)p.N * E;
  float* dhq = dalign_carry + (size_t)p.N * p.Tia;
  float* dh_carry = dhq + (size_t)p.N * A;
  float* dc_carry = dh_carry + (size_t)p.N * A;
  const size_t lds = sizeof(float) * (2 * (p.Ti + 2 * MAXKW) + 2 * p.Ti + p.E + 32 + 4 * (2 + MAXKW) * p.A);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  NS_CHECK_ARG(lds <= 160 * 1024, "ns_taco2_attn_bwd: T_in too long for LDS (%zu bytes)", lds);
  T* hc = (T*)p.hc; T* xa = (T*)p.xa; T* p1 = (T*)p.p1;
  for (int st = p.S - 1; st >= 0; --st) {
    const long slot = st + 1, prev = st;
    const bool last = (st == p.S - 1);
    int rc;
    AttnBwdStep<T> a = {};
    a.Ti = p.Ti; a.A = p.A; a.E = p.E; a.kw = p.kw; a.Tia = p.Tia; a.lengths = p.lengths;
    a.keys = p.keys + (long)p.padl_i * A; a.keys_sn = (long)p.Pi * A;
    a.values = (const T*)p.values + (long)p.padl_i * E; a.values_sn = (long)p.Pi * E;
    a.q = p.q + slot * A; a.q_sn = S1 * A;
    a.acur = p.align + slot * p.Tia; a.aprev = p.align + prev * p.Tia; a.al_sn = S1 * p.Tia;
    a.dctx_ext = p.dhc + slot * HC + A; a.dce_sn = S1 * HC;
    a.dctx_carry = last ? nullptr : dctx_carry;
    a.dalign_carry = dalign_carry; a.has_carry = last ? 0 : 1;
    a.dq_out = (T*)p.dq + slot * A; a.dq_sn = S1 * A;
    a.dkeys = p.dkeys + (long)p.padl_i * A; a.dvalues = p.dvalues + (long)p.padl_i * E;
    a.dv = p.dv; a.dwcl = p.dwcl; a.wcl = p.wcl; a.v = p.v;
    hipLaunchKernelGGL(attn_bwd_kernel<T>, dim3(p.N), dim3(256), lds, s, a);
    NS_CHECK_LAUNCH("attn_bwd");
    // dhq = dq . Wq^T
    rc = gemm_small(dt, p.N, (int)A, (int)A, (T*)p.dq + slot * A, S1 * A, p.wq, A, dhq, A, NS_F32, nullptr,
                    NS_ACT_NONE, nullptr, 0, nullptr, 0, s);
    if (rc) return rc;
    LstmBwdCell<T> c = {};
    c.N = p.N; c.H = p.A; c.t = st; c.first = last ? 1 : 0; c.lengths = nullptr;
    c.dh_out = p.dhc + slot * HC; c.dho_sn = S1 * HC;
    c.dh_out2 = dhq; c.dho2_sn = A;
    c.dh_carry = last ? nullptr : dh_carry; c.dhc_sn = A;
    c.gates = (const T*)p.ga + slot * 4 * A; c.g_sn = S1 * 4 * A;
    c.c = p.ca + slot * A; c.c_prev = st > 0 ? p.ca + prev * A : nullptr; c.c_sn = S1 * A;
    c.dc_carry = dc_carry;
    c.dgates = (T*)p.dga + slot * 4 * A; c.dg_sn = S1 * 4 * A;
    rc = lstm_bwd_cell_launch<T>(c, s);
    if (rc) return rc;
    const T* dga = (const T*)p.dga + slot * 4 * A;
    // dp2pre = (dga . Watt[0:D2]^T) * (p2 > 0)
    rc = gemm_small(dt, p.N, (int)D2, (int)(4 * A), dga, S1 * 4 * A, p.watt, 4 * A, (T*)p.dp2 + slot * D2, S1 * D2,
                    dt, nullptr, NS_ACT_NONE, nullptr, 0, xa + slot * XA, S1 * XA, s);
    if (rc) return rc;
    // dh_carry = dga . Watt[D2:]^T
    if (st > 0) {
      rc = gemm_small(dt, p.N, (int)A, (int)(4 * A), dga, S1 * 4 * A, (const T*)p.watt + D2 * 4 * A, 4 * A, dh_carry,
                      A, NS_F32, nullptr, NS_ACT_NONE, nullptr, 0, nullptr, 0, s);
      if (rc) return rc;
    }
    // dp1pre = (dp2pre . W2^T) * (p1 > 0)
    rc = gemm_small(dt, p.N, (int)D1, (int)D2, (T*)p.dp2 + slot * D2, S1 * D2, p.w2, D2, (T*)p.df1 + slot * D1,
                    S1 * D1, dt, nullptr, NS_ACT_NONE, nullptr, 0, p1 + slot * D1, S1 * D1, s);
    if (rc) return rc;
    // dctx_carry = dp1pre . W1c^T
    if (st > 0) {
      rc = gemm_small(dt, p.N, (int)E, (int)D1, (T*)p.df1 + slot * D1, S1 * D1, p.w1c, D1, dctx_carry, E, NS_F32,
                      nullptr, NS_ACT_NONE, nullptr, 0, nullptr, 0, s);
      if (rc) return rc;
    }
  }
  (void)hc;
  return NS_OK;
}

extern "C" int ns_taco2_attn_bwd(const ns_taco2_attn_params* p, ns_stream_t s) {
  int rc = check_attn(p, "ns_taco2_attn_bwd");
  if (rc) return rc;
  NS_CHECK_ARG(p->keys && p->values && p->w1c && p->w2 && p->watt && p->wq && p->wcl && p->v && p->p1 && p->xa &&
                   p->ca && p->ga && p->q && p->align && p->dhc && p->df1 && p->dp2 && p->dga && p->dq &&
                   p->dkeys && p->dvalues && p->dv && p->dwcl && p->work,
               "ns_taco2_attn_bwd: null pointer");
  if (p->dtype == NS_BF16) return attn_bwd_t<bf16_t>(*p, (hipStream_t)s);
  return attn_bwd_t<float>(*p, (hipStream_t)s);
}
