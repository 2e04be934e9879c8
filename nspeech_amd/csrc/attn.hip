// Tacotron-2 attention RNN (prenet -> attention LSTM -> location-sensitive attention) for the
// teacher-forced decoder, forward and backward through time.  The matrix products are M=batch
// skinny GEMMs (ns_gemm), the LSTM cell is the fused step kernel of lstm.hip, and the
// energies / masked softmax / context and their gradients are the kernels below.
//
// Layout choice that makes the energy kernels reduction-free: one memory position t per LANE
// (keys are transposed once per call to [A][T_in]) and a serial loop over the attention units,
// so e[t] = sum_u v[u] tanh(keys[t,u] + q[u] + loc[t,u]) and the location-filter gradient
// accumulate in registers; quantities that are sums over t (dq, dv, dWcl) are computed by a
// second sweep with one unit per lane over the original [T_in][A] layout.
#include "common.h"
#include "lstm_step.h"

constexpr int MAXKW = 8;
constexpr int PADK = 8;          // zero margin around the alignment vector in LDS
constexpr int ATHREADS = 512;
constexpr int AW = ATHREADS / 64;
constexpr int UB = 32;           // units (or time steps) whose loads are in flight together

__device__ __forceinline__ void ld8(const bf16_t* p, float* o) {
  const bf16x8 v = *(const bf16x8*)p;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
__device__ __forceinline__ void ld8(const float* p, float* o) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// keys [N, Pi, A] (valid rows padl..) <-> keys_t [N, A, Tia]
__global__ void keys_transpose_kernel(const float* keys, float* keys_t, int Ti, int Tia, int Pi, int padl, int A,
                                      int back) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, t0 = blockIdx.x * 32, u0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* kn = keys + ((long)n * Pi + padl) * A;
  float* ktn = keys_t + (long)n * A * Tia;
  if (!back) {
    for (int j = ty; j < 32; j += 8) {
      const int t = t0 + j, u = u0 + tx;
      tile[j][tx] = (t < Ti && u < A) ? kn[(long)t * A + u] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
      const int u = u0 + j, t = t0 + tx;
      if (u < A && t < Tia) ktn[(long)u * Tia + t] = tile[tx][j];
    }
  } else {  // keys[t][u] += keys_t[u][t]
    for (int j = ty; j < 32; j += 8) {
      const int u = u0 + j, t = t0 + tx;
      tile[j][tx] = (u < A && t < Ti) ? ktn[(long)u * Tia + t] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
      const int t = t0 + j, u = u0 + tx;
      if (t < Ti && u < A) ((float*)kn)[(long)t * A + u] += tile[tx][j];
    }
  }
}

// ------------------------------------------------------------------ forward, per step
// Two launches so that every batch row spreads over many CUs: energies (grid = t-chunks x N),
// then masked softmax + context (grid = column chunks x N).
template <typename T>
struct AttnStep {
  int Ti, A, E, kw, Tia;
  const int* lengths;
  const float* keys_t;                 // [N, A, Tia]
  const T* values; long values_sn;     // row-n base + [Ti, E]
  const float* q; long q_sn;           // [A]
  const float* aprev; long al_sn;      // [Tia]
  float* e_raw;                        // [N, Tia] scratch
  float* aout;
  T* aout_t;                           // operand-dtype copy of the alignments (same strides)
  T* ctx_out; long ctx_sn;             // [E]
  T* ctx_out2; long ctx2_sn;           // optional second destination
  const float* wcl; const float* v;
  const float* add; long add_sn; int relu;   // optional: ctx_out = relu?(weighted sum + add[n*add_sn + c])
  // optional second value matrix in the same launch (free-running synthesis: the projected memory, whose weighted sum
  // is the context term of the NEXT step's prenet layer): column chunks past nch take pv / E2 / pv_out
  const T* pv; long pv_sn; int E2, nch; T* pv_out; long pvo_sn;
  bf16_t* ctx_rows; int cr_nkc, cr_col;      // optional: the context into ns_rows32's packed rows (row n = the utterance)
};

// x[t,u] = keys[t,u] + q[u] + sum_k align_prev[t+k-half] Wcl[k,u];  lane = t, wave = unit chunk.
// q, v and Wcl are wave-uniform (scalar loads); tanh is v_exp + v_rcp.
// per-unit constants of one wave's unit chunk, packed for broadcast ds_read_b128:
// cst[u] = {q, v, w0, w1 | w2, w3, w4, w5 | w6, w7, 0, 0}
constexpr int CPU = 12;
template <typename T>
__global__ __launch_bounds__(ATHREADS) void attn_energy_kernel(AttnStep<T> a) {
  __shared__ float part[AW][64];
  __shared__ __attribute__((aligned(16))) float cst[AW][UB * CPU];
  const int n = blockIdx.y, tc = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Ti = a.Ti, A = a.A, Tia = a.Tia;
  const int L = min(a.lengths ? a.lengths[n] : Ti, Ti);
  const int half = (a.kw - 1) / 2;
  const int t = tc * 64 + lane;
  const bool act = t < L;
  const float* aprev = a.aprev + (long)n * a.al_sn;
  float apv[MAXKW];
#pragma unroll
  for (int k = 0; k < MAXKW; ++k) {
    const int tt = t + k - half;
    apv[k] = (k < a.kw && tt >= 0 && tt < Ti) ? aprev[tt] : 0.f;
  }
  const int upc = (A + AW - 1) / AW;
  const int ub0 = wv * upc, ue = min(A, ub0 + upc);
  const float* kt = a.keys_t + (long)n * A * Tia;
  const float* qn = a.q + (long)n * a.q_sn;
  float s = 0.f;
  for (int ub = ub0; ub < ue; ub += UB) {
    float kv[UB];
#pragma unroll
    for (int j = 0; j < UB; ++j) kv[j] = (act && ub + j < ue) ? kt[(long)(ub + j) * Tia + t] : 0.f;
    // stage this block's constants (wave-private LDS region, no block barrier needed): lane -> (unit j, half); half 0
    // fetches q, v, w0..w2, half 1 w3..w7 - ten independent loads per lane in flight at once (the former loop over
    // (unit, field) pairs issued its loads one dependent iteration after the other: 6 round trips per step)
    {
      static_assert(UB == 32, "one unit per half-wave lane");
      const int j = lane & (UB - 1), hf = lane >> 5, u = ub + j;
      float c[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      if (u < ue) {
        if (hf == 0) {
          c[0] = qn[u]; c[1] = a.v[u];
#pragma unroll
          for (int k = 0; k < 3; ++k) c[2 + k] = k < a.kw ? a.wcl[k * A + u] : 0.f;
        } else {
#pragma unroll
          for (int k = 3; k < MAXKW; ++k) c[k - 3] = k < a.kw ? a.wcl[k * A + u] : 0.f;
        }
      }
      float* d = &cst[wv][j * CPU + (hf ? 5 : 0)];
#pragma unroll
      for (int i = 0; i < 5; ++i) d[i] = c[i];
      if (hf) { d[5] = 0.f; d[6] = 0.f; }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const float4 c0 = *(const float4*)&cst[wv][j * CPU];
      const float4 c1 = *(const float4*)&cst[wv][j * CPU + 4];
      const float4 c2 = *(const float4*)&cst[wv][j * CPU + 8];
      float x = kv[j] + c0.x;
      x = fmaf(apv[0], c0.z, x); x = fmaf(apv[1], c0.w, x);
      x = fmaf(apv[2], c1.x, x); x = fmaf(apv[3], c1.y, x); x = fmaf(apv[4], c1.z, x); x = fmaf(apv[5], c1.w, x);
      x = fmaf(apv[6], c2.x, x); x = fmaf(apv[7], c2.y, x);
      s = fmaf(c0.y, tanhf_(x), s);     // v = 0 for units past the chunk end
    }
    __builtin_amdgcn_wave_barrier();
  }
  part[wv][lane] = s;
  __syncthreads();
  if (wv == 0 && t < Tia) {
    float e = -INFINITY;
    if (act) {
      e = 0.f;
#pragma unroll
      for (int w = 0; w < AW; ++w) e += part[w][lane];
    }
    a.e_raw[(long)n * Tia + t] = e;
  }
}

constexpr int CCH = 128;   // context columns per workgroup
// LDS (floats): al[Tia] | red[32] | cpart[4][CCH]
template <typename T>
__global__ __launch_bounds__(256) void attn_context_kernel(AttnStep<T> a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Ti = a.Ti, Tia = a.Tia;
  float* al = sm;
  float* red = al + Tia;
  float* cpart = red + 32;
  const int n = blockIdx.y;
  const bool second = a.pv && (int)blockIdx.x >= a.nch;
  const int ch = second ? blockIdx.x - a.nch : blockIdx.x;
  const int E = second ? a.E2 : a.E;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = min(a.lengths ? a.lengths[n] : Ti, Ti);
  const float* er = a.e_raw + (long)n * Tia;
  float mx = -INFINITY;
  for (int t = tid; t < L; t += 256) mx = fmaxf(mx, er[t]);
  mx = block_max(mx, red);
  float sum = 0.f;
  for (int t = tid; t < L; t += 256) sum += __expf(er[t] - mx);
  sum = block_sum(sum, red);
  const float inv = 1.f / sum;
  for (int t = tid; t < Tia; t += 256) {
    const float v = t < L ? __expf(er[t] - mx) * inv : 0.f;
    al[t] = v;
    if (blockIdx.x == 0) {
      a.aout[(long)n * a.al_sn + t] = v;
      if (a.aout_t) stf(a.aout_t + (long)n * a.al_sn + t, v);
    }
  }
  __syncthreads();
  const T* values = second ? a.pv + (long)n * a.pv_sn : a.values + (long)n * a.values_sn;
  const int c = ch * CCH + (lane & 15) * 8;
  const int rg = lane >> 4;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < E) {
#pragma unroll 4
    for (int t = wave * 4 + rg; t < L; t += 16) {
      float x[8];
      ld8(values + (long)t * E + c, x);
      const float w = al[t];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(w, x[i], acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] += __shfl_xor(acc[i], 16, 64);
    acc[i] += __shfl_xor(acc[i], 32, 64);
  }
  if (rg == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) cpart[wave * CCH + (lane & 15) * 8 + i] = acc[i];
  }
  __syncthreads();
  for (int i = tid; i < CCH; i += 256) {
    const int cc = ch * CCH + i;
    if (cc < E) {
      float v = cpart[i] + cpart[CCH + i] + cpart[2 * CCH + i] + cpart[3 * CCH + i];
      if (second) {
        stf(a.pv_out + (long)n * a.pvo_sn + cc, v);
        continue;
      }
      if (a.add) v += a.add[(long)n * a.add_sn + cc];
      if (a.relu) v = fmaxf(v, 0.f);
      stf(a.ctx_out + (long)n * a.ctx_sn + cc, v);
      if (a.ctx_out2) stf(a.ctx_out2 + (long)n * a.ctx2_sn + cc, v);
      if (a.ctx_rows) ns_rows32_store(a.ctx_rows, a.cr_nkc, n, a.cr_col + cc, v);
    }
  }
}

// ------------------------------------------------------------------ backward, per step
template <typename T>
struct AttnBwdStep {
  int Ti, A, E, kw, Tia;
  const int* lengths;
  const float* keys; long keys_sn;         // [Ti, A] original layout (row-n base = keys + n*keys_sn)
  const float* keys_t;                     // [N, A, Tia]
  const T* values; long values_sn;
  const float* q; long q_sn;
  const float* acur; const float* aprev; long al_sn;
  const float* dctx_ext; long dce_sn;      // [E] from downstream (dhc columns A..)
  const float* dctx_carry;                 // [N,E] or null
  float* gk;                               // [N,Tia,MAXKW] location-filter gradients of the step after
  int has_carry;                           //   (read by da, then overwritten by the energy kernel)
  float* da;                               // [N,Tia] scratch
  T* dq_out; long dq_sn;
  float* de_out;                           // [Tia] per row (stride al_sn): energy gradients of this step
  T* dctx_out; long dco_sn;                // [E] total context gradient of this step (operand dtype); null = not stored
  const float* wcl; const float* v;
  // projected-memory form (values = memory . W1c, E = prenet width): the vector dotted with the memory rows is the
  // prenet gradient of the step after (dvec, operand dtype, null on the last step) and the part of da that needs no
  // recurrence arrives precomputed (da0)
  int pv_mode; const T* dvec; long dv_sn; const float* da0; long da0_sn;
};

// (1) da[t] = dctx . values[t] + sum_k G_next[t-k+half][k];  grid (t-chunks of 32, N)
template <typename T>
__global__ __launch_bounds__(ATHREADS) void attn_bwd_da_kernel(AttnBwdStep<T> a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // dctx[E]
  const int Ti = a.Ti, E = a.E, Tia = a.Tia;
  const int n = blockIdx.y, t0 = blockIdx.x * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = min(a.lengths ? a.lengths[n] : Ti, Ti);
  const int half = (a.kw - 1) / 2;
  for (int c = tid; c < E; c += ATHREADS) {
    float d;
    if (a.pv_mode) d = a.dvec ? ldf(a.dvec + (long)n * a.dv_sn + c) : 0.f;
    else {
      d = a.dctx_ext[(long)n * a.dce_sn + c];
      if (a.dctx_carry) d += a.dctx_carry[(long)n * E + c];
    }
    sm[c] = d;
    if (blockIdx.x == 0 && a.dctx_out) stf(a.dctx_out + (long)n * a.dco_sn + c, d);
  }
  __syncthreads();
  const T* values = a.values + (long)n * a.values_sn;
  const float* gk = a.gk + (long)n * Tia * MAXKW;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = lane * 8; c < E; c += 512) {
    float x[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = t0 + wave * 4 + j;
      if (t < L) ld8(values + (long)t * E + c, x[j]);
      else {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[j][i] = 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) s[j] = fmaf(sm[c + i], x[j][i], s[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = t0 + wave * 4 + j;
    const float sv = wave_sum(s[j]);
    if (lane == 0 && t < Tia) {
      float carry = 0.f;
      if (a.has_carry && t < L) {
        for (int k = 0; k < a.kw; ++k) {
          const int ts = t - k + half;
          if (ts >= 0 && ts < Ti) carry += gk[(long)ts * MAXKW + k];
        }
      }
      const float base = (a.da0 && t < L) ? a.da0[(long)n * a.da0_sn + t] : 0.f;
      a.da[(long)n * Tia + t] = t < L ? sv + carry + base : 0.f;
    }
  }
}

// (2) softmax backward + gradient wrt the previous alignments (as per-tap terms G[t][k]);
//     grid (t-chunks of 64, N), lane = t, wave = unit chunk
template <typename T>
__global__ __launch_bounds__(ATHREADS) void attn_bwd_energy_kernel(AttnBwdStep<T> a) {
  __shared__ float part[AW][64][MAXKW + 1];
  __shared__ float red[32];
  __shared__ __attribute__((aligned(16))) float cst[AW][UB * CPU];
  const int n = blockIdx.y, tc = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Ti = a.Ti, A = a.A, Tia = a.Tia;
  const int L = min(a.lengths ? a.lengths[n] : Ti, Ti);
  const int half = (a.kw - 1) / 2;
  const float* acur = a.acur + (long)n * a.al_sn;
  const float* aprev = a.aprev + (long)n * a.al_sn;
  const float* da = a.da + (long)n * Tia;
  float dot = 0.f;
  for (int t = tid; t < L; t += ATHREADS) dot += acur[t] * da[t];
  dot = block_sum(dot, red);
  const int t = tc * 64 + lane;
  const bool act = t < L;
  const float de = act ? acur[t] * (da[t] - dot) : 0.f;
  if (wv == 0 && t < Tia) a.de_out[(long)n * a.al_sn + t] = de;
  float apv[MAXKW], g[MAXKW];
#pragma unroll
  for (int k = 0; k < MAXKW; ++k) {
    const int tt = t + k - half;
    apv[k] = (k < a.kw && tt >= 0 && tt < Ti) ? aprev[tt] : 0.f;
    g[k] = 0.f;
  }
  const int upc = (A + AW - 1) / AW;
  const int ub0 = wv * upc, ue = min(A, ub0 + upc);
  const float* kt = a.keys_t + (long)n * A * Tia;
  const float* qn = a.q + (long)n * a.q_sn;
  for (int ub = ub0; ub < ue; ub += UB) {
    float kv[UB];
#pragma unroll
    for (int j = 0; j < UB; ++j) kv[j] = (act && ub + j < ue) ? kt[(long)(ub + j) * Tia + t] : 0.f;
    for (int i = lane; i < UB * CPU; i += 64) {
      const int j = i / CPU, f = i % CPU, u = ub + j;
      float val = 0.f;
      if (u < ue) {
        if (f == 0) val = qn[u];
        else if (f == 1) val = a.v[u];
        else if (f - 2 < a.kw) val = a.wcl[(f - 2) * A + u];
      }
      cst[wv][i] = val;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const float4 c0 = *(const float4*)&cst[wv][j * CPU];
      const float4 c1 = *(const float4*)&cst[wv][j * CPU + 4];
      const float4 c2 = *(const float4*)&cst[wv][j * CPU + 8];
      float x = kv[j] + c0.x;
      x = fmaf(apv[0], c0.z, x); x = fmaf(apv[1], c0.w, x);
      x = fmaf(apv[2], c1.x, x); x = fmaf(apv[3], c1.y, x); x = fmaf(apv[4], c1.z, x); x = fmaf(apv[5], c1.w, x);
      x = fmaf(apv[6], c2.x, x); x = fmaf(apv[7], c2.y, x);
      const float th = tanhf_(x);
      const float dpre = de * c0.y * (1.f - th * th);
      g[0] = fmaf(dpre, c0.z, g[0]); g[1] = fmaf(dpre, c0.w, g[1]);
      g[2] = fmaf(dpre, c1.x, g[2]); g[3] = fmaf(dpre, c1.y, g[3]); g[4] = fmaf(dpre, c1.z, g[4]); g[5] = fmaf(dpre, c1.w, g[5]);
      g[6] = fmaf(dpre, c2.x, g[6]); g[7] = fmaf(dpre, c2.y, g[7]);
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int k = 0; k < MAXKW; ++k) part[wv][lane][k] = g[k];
  __syncthreads();
  float* gk = a.gk + (long)n * Tia * MAXKW;
  for (int i = tid; i < 64 * MAXKW; i += ATHREADS) {
    const int l = i / MAXKW, k = i % MAXKW;
    const int tt = tc * 64 + l;
    if (tt < Tia) {
      float sacc = 0.f;
#pragma unroll
      for (int w = 0; w < AW; ++w) sacc += part[w][l][k];
      gk[(long)tt * MAXKW + k] = sacc;
    }
  }
}

// (3) dq[u] = sum_t dpre[t,u];  grid (unit chunks of 64, N), lane = unit, wave = time slice
template <typename T>
__global__ __launch_bounds__(ATHREADS) void attn_bwd_dq_kernel(AttnBwdStep<T> a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // ap[Ti+2P] | de[Tia] | part[AW][64]
  const int Ti = a.Ti, A = a.A, Tia = a.Tia;
  float* ap = sm;
  float* de = ap + Ti + 2 * PADK;
  float* part = de + Tia;
  const int n = blockIdx.y, uw = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, ts = tid >> 6;
  const int L = min(a.lengths ? a.lengths[n] : Ti, Ti);
  const int half = (a.kw - 1) / 2;
  const float* aprev = a.aprev + (long)n * a.al_sn;
  const float* den = a.de_out + (long)n * a.al_sn;
  for (int i = tid; i < Ti + 2 * PADK; i += ATHREADS) {
    const int t = i - PADK;
    ap[i] = (t >= 0 && t < Ti) ? aprev[t] : 0.f;
  }
  for (int t = tid; t < Tia; t += ATHREADS) de[t] = t < L ? den[t] : 0.f;
  __syncthreads();
  const int u = uw * 64 + lane;
  float dq = 0.f;
  if (u < A) {
    const float* keys = a.keys + (long)n * a.keys_sn;
    const float qu = a.q[(long)n * a.q_sn + u], vu = a.v[u];
    float w[MAXKW];
#pragma unroll
    for (int k = 0; k < MAXKW; ++k) w[k] = (k < a.kw) ? a.wcl[k * A + u] : 0.f;
    for (int tb = ts; tb < L; tb += AW * UB) {
      float kv[UB];
#pragma unroll
      for (int j = 0; j < UB; ++j) {
        const int t = tb + j * AW;
        kv[j] = t < L ? keys[(long)t * A + u] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < UB; ++j) {
        const int t = tb + j * AW;
        if (t < L) {
          float x = kv[j] + qu;
#pragma unroll
          for (int k = 0; k < MAXKW; ++k)
            if (k < a.kw) x = fmaf(ap[PADK + t + k - half], w[k], x);
          const float th = tanhf_(x);
          dq = fmaf(de[t] * vu, 1.f - th * th, dq);
        }
      }
    }
  }
  part[ts * 64 + lane] = dq;
  __syncthreads();
  if (ts == 0 && u < A) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < AW; ++w) s += part[w * 64 + lane];
    stf(a.dq_out + (long)n * a.dq_sn + u, s);
  }
}

// Hoisted part of the attention backward: sums over all decoder steps that no recurrence needs.
//   dkeys[t,u] = sum_s dpre_s[t,u],  dv[u] = sum_{s,t} de_s[t] th_s[t,u],
//   dWcl[k,u]  = sum_{s,t} align_{s-1}[t+k-half] dpre_s[t,u]
// One wave = 64 memory positions x PU units, looping over the S steps with everything in registers (th is recomputed;
// nothing is read-modify-written in memory).  The per-step operands - a 64 + kw - 1 wide window of the previous
// alignments and 64 energy gradients, shared by the four waves of the workgroup - come through LDS in blocks of
// POST_SB steps: the next block's global loads are in flight while this block is computed (a register prefetch one
// step ahead did not cover the load latency with three waves per SIMD: 515 us per call, 0.16 ms of arithmetic); the
// query row (one address per wave) comes through scalar loads.
constexpr int PU = 4;
constexpr int POST_SB = 8;                       // steps per staged block
constexpr int POST_AW = 64 + MAXKW;              // alignment window floats per step (64 + kw - 1 used)
constexpr int POST_ROW = POST_AW + 64;           // + energy gradients
constexpr int POST_PT = (POST_SB * POST_ROW + 255) / 256;   // staged values per thread
struct AttnPost {
  int S, Ti, A, kw, Tia;
  const int* lengths;
  const float* keys_t;     // [N,A,Tia]
  const float* q;          // [N,S+1,A]
  const float* align;      // [N,S+1,Tia]
  const float* de;         // [N,S+1,Tia]
  const float* wcl; const float* v;
  float* dkeys_t;          // [N,A,Tia] out (plain store)
  float* dv; float* dwcl;  // +=
  float* part;             // optional [N * position blocks][1 + MAXKW][A]: parked partial sums (fixed-order finish)
  int dbg;                 // NS_POST_DBG (diagnostics): 1 = a barrier in front of the stage store, 2 = the run-time-kw kernel, 4 = no early exit
};
template <int KW>          // KW = the filter width when it is the usual 7 (straight-line step), 0 = a.kw at run time
__global__ __launch_bounds__(256) void attn_post_kernel(AttnPost a) {
  __shared__ __attribute__((aligned(8))) float wl[4][(1 + MAXKW) * PU];
  __shared__ float stg[2][POST_SB][POST_ROW];
  const int n = blockIdx.z, tc = blockIdx.x;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int u0 = (blockIdx.y * 4 + wave) * PU;       // wave-uniform: the query row below comes through scalar loads
  const int t = tc * 64 + lane;
  const int Ti = a.Ti, A = a.A, Tia = a.Tia, S1 = a.S + 1;
  const int L = min(a.lengths ? a.lengths[n] : Ti, Ti);
  const int kw = KW ? KW : a.kw;
  const int half = (kw - 1) / 2;
  const bool wave_on = u0 < A;                       // idle waves still stage and meet the barriers
  if (tc * 64 >= L && !(a.dbg & 4)) {
    // every position of this block lies past the utterance's length: all its terms are zero (d energy is masked there).
    // Store the zeros and leave - with lengths ~U(T/2, T) a fifth of the blocks of a batch end here (block-uniform: no
    // barrier has been reached yet)
    if (wave_on) {
#pragma unroll
      for (int j = 0; j < PU; ++j) {
        if (u0 + j < A) {
          if (t < Tia) a.dkeys_t[((long)n * A + u0 + j) * Tia + t] = 0.f;
          if (a.part && lane <= kw) a.part[(((long)n * gridDim.x + tc) * (1 + MAXKW) + lane) * A + u0 + j] = 0.f;
        }
      }
    }
    return;
  }
  if (wave_on) {
    for (int i = lane; i < (1 + MAXKW) * PU; i += 64) {
      const int j = i % PU, k = i / PU;   // k = 0: v, k >= 1: wcl[k-1]
      float val = 0.f;
      if (u0 + j < A) {
        if (k == 0) val = a.v[u0 + j];
        else if (k - 1 < kw) val = a.wcl[(k - 1) * A + u0 + j];
      }
      wl[wave][i] = val;
    }
  }
  const bool act = t < L;
  // units in pairs: the multiply-adds of the location filter and of its gradient run as v_pk_fma_f32
  typedef float f2 __attribute__((ext_vector_type(2)));
  constexpr int PP = PU / 2;
  f2 kv[PP], dk[PP], dvp[PP], dwp[MAXKW][PP];
#pragma unroll
  for (int j = 0; j < PP; ++j) {
    kv[j].x = (wave_on && act && u0 + 2 * j < A) ? a.keys_t[((long)n * A + u0 + 2 * j) * Tia + t] : 0.f;
    kv[j].y = (wave_on && act && u0 + 2 * j + 1 < A) ? a.keys_t[((long)n * A + u0 + 2 * j + 1) * Tia + t] : 0.f;
    dk[j] = (f2){0.f, 0.f}; dvp[j] = (f2){0.f, 0.f};
#pragma unroll
    for (int k = 0; k < MAXKW; ++k) dwp[k][j] = (f2){0.f, 0.f};
  }
  const float* alb = a.align + (long)n * S1 * Tia;
  const float* deb = a.de + (long)n * S1 * Tia;
  const float* qb = a.q + (long)n * S1 * A + (wave_on ? u0 : 0);
  const f2* wl2 = (const f2*)wl[wave];            // [(1 + MAXKW)][PP] pairs: row 0 = v, row 1 + k = wcl[k]

  // this thread's share of a staged block: POST_PT values, the same (step-in-block, column) every block
  const float* src[POST_PT];
  int sstep[POST_PT];
  bool sok[POST_PT];
#pragma unroll
  for (int i = 0; i < POST_PT; ++i) {
    const int idx = tid + 256 * i;
    const int si = idx / POST_ROW, r = idx % POST_ROW;
    sstep[i] = si;
    if (r < POST_AW) {                      // align_{slot-1}[tc*64 - half + r]
      const int tt = tc * 64 - half + r;
      sok[i] = idx < POST_SB * POST_ROW && r < 64 + kw - 1 && tt >= 0 && tt < Ti;
      src[i] = alb + (long)(si - 1) * Tia + min(max(tt, 0), Ti - 1);
    } else {                                // de_slot[tc*64 + r - POST_AW]
      const int t2 = tc * 64 + r - POST_AW;
      sok[i] = idx < POST_SB * POST_ROW && t2 < L;
      src[i] = deb + (long)si * Tia + min(t2, Tia - 1);
    }
  }
  float sv[POST_PT];
  auto gload = [&](int s0) {               // slots s0 .. s0 + POST_SB - 1
#pragma unroll
    for (int i = 0; i < POST_PT; ++i) {
      const bool ok = sok[i] && s0 + sstep[i] <= a.S;
      const float* pp = ok ? src[i] + (long)s0 * Tia : alb;        // never dereference an address outside the arrays
      sv[i] = *pp;
      sv[i] = ok ? sv[i] : 0.f;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < POST_PT; ++i) {
      const int idx = tid + 256 * i;
      if (idx < POST_SB * POST_ROW) (&stg[buf][0][0])[idx] = sv[i];
    }
  };
  // slot 1 is the first step; src[] was built for si - 1 / si relative to s0, and slot s0 + si - 1 >= 0 always holds
  gload(1);
  sstore(0);
  __syncthreads();
  f2 qn[PP];
  {
    const f2* qs = (const f2*)(qb + (long)1 * A);
#pragma unroll
    for (int j = 0; j < PP; ++j) qn[j] = qs[j];
  }
  int buf = 0;
  for (int s0 = 1; s0 <= a.S; s0 += POST_SB, buf ^= 1) {
    if (s0 + POST_SB <= a.S) gload(s0 + POST_SB);           // in flight during this block's arithmetic
    if (wave_on) {
#pragma unroll
      for (int si = 0; si < POST_SB; ++si) {
        const int slot = s0 + si;
        if (slot <= a.S) {
          float apv[MAXKW];
#pragma unroll
          for (int k = 0; k < MAXKW; ++k) apv[k] = (KW ? k < KW : k < kw) ? stg[buf][si][lane + k] : 0.f;
          const float de = stg[buf][si][POST_AW + lane];
          f2 qv[PP];
#pragma unroll
          for (int j = 0; j < PP; ++j) qv[j] = qn[j];
          {
            const f2* qs = (const f2*)(qb + (long)min(slot + 1, a.S) * A);
#pragma unroll
            for (int j = 0; j < PP; ++j) qn[j] = qs[j];
          }
          const f2 de2 = (f2){de, de};
#pragma unroll
          for (int j = 0; j < PP; ++j) {
            f2 x = kv[j] + qv[j];
#pragma unroll
            for (int k = 0; k < MAXKW; ++k)
              if (KW ? k < KW : k < kw) x = (f2){apv[k], apv[k]} * wl2[(1 + k) * PP + j] + x;
            f2 th;
            th.x = tanhf_(x.x);
            th.y = tanhf_(x.y);
            const f2 dpre = de2 * wl2[j] * ((f2){1.f, 1.f} - th * th);
            dk[j] += dpre;
            dvp[j] = de2 * th + dvp[j];
            // dwp[k] += align[t + k - half] * dpre, two units per instruction.  The instruction form is pinned: the tap
            // pair (apv[k], apv[k + 1]) is src0 and op_sel broadcasts its low (even k) or high (odd k) half.  Left to the
            // compiler the broadcast operand became src1 with op_sel:[0,1,0] for the odd taps, and with that form the
            // odd taps' sums came out wrong by ~0.15 % whenever this library's weight-gradient workgroups (MFMA waves of
            // another kernel) shared the CU - never alone, never for the even taps (op_sel_hi:[1,0,1]), never for x above
            // (src0 forms): profiles/r04_determinism.txt item 4 has the float64 comparison that pins it down.
#pragma unroll
            for (int k = 0; k < MAXKW; k += 2) {
              if (KW ? k < KW : k < kw) {
                const f2 ap2 = (f2){apv[k], k + 1 < MAXKW ? apv[k + 1] : 0.f};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(dwp[k][j]) : "v"(ap2), "v"(dpre));
                if (k + 1 < MAXKW && (KW ? k + 1 < KW : k + 1 < kw))
                  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(dwp[k + 1][j]) : "v"(ap2), "v"(dpre));
              }
            }
          }
        }
      }
    }
    if (a.dbg & 1) __syncthreads();
    if (s0 + POST_SB <= a.S) sstore(buf ^ 1);
    __syncthreads();
  }
  if (!wave_on) return;
#pragma unroll
  for (int j = 0; j < PU; ++j) {
    if (u0 + j < A) {
      const float dkj = (j & 1) ? dk[j >> 1].y : dk[j >> 1].x;
      const float dvj = (j & 1) ? dvp[j >> 1].y : dvp[j >> 1].x;
      if (t < Tia) a.dkeys_t[((long)n * A + u0 + j) * Tia + t] = dkj;
      // dv, dwcl: sums over every utterance and position block.  With `part` this block's share is parked and
      // attn_post_finish_kernel adds the shares in a fixed order; without it they meet in float atomics.
      float* pp = a.part ? a.part + ((long)n * gridDim.x + tc) * (1 + MAXKW) * A : nullptr;
      const float sv_ = wave_sum(dvj);
      if (lane == 0) { if (pp) pp[u0 + j] = sv_; else atomicAdd(a.dv + u0 + j, sv_); }
#pragma unroll
      for (int k = 0; k < MAXKW; ++k)
        if (k < kw) {
          const float sw = wave_sum((j & 1) ? dwp[k][j >> 1].y : dwp[k][j >> 1].x);
          if (lane == 0) { if (pp) pp[(1 + k) * A + u0 + j] = sw; else atomicAdd(a.dwcl + k * A + u0 + j, sw); }
        }
    }
  }
}

// ------------------------------------------------------------------ host loops
static thread_local int g_f32_passes = 0;   // set per call from ns_taco2_attn_params
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
  g.f32_passes = g_f32_passes;
  return ns_gemm(&g, s);
}

extern "C" size_t ns_taco2_attn_work_bytes(const ns_taco2_attn_params* p) {
  if (!p) return 0;
  // dctx_carry [N,E] + e_raw/da [N,Tia] + G [N,Tia,MAXKW] + dhq [N,A] + dc_carry [N,A] + dkeys_t [N,A,Tia]
  return sizeof(float) * ((size_t)p->N * (p->E + (1 + MAXKW) * p->Tia + 2 * p->A + (size_t)p->A * p->Tia)) + 256;
}

static int check_attn(const ns_taco2_attn_params* p, const char* who) {
  NS_CHECK_ARG(p != nullptr, "%s: null params", who);
  NS_CHECK_ARG(p->A % 8 == 0 && p->A <= 256, "%s: attention units must be a multiple of 8, <= 256", who);
  NS_CHECK_ARG(p->E % 8 == 0, "%s: memory depth must be a multiple of 8", who);
  NS_CHECK_ARG(p->kw >= 1 && p->kw <= MAXKW, "%s: location filter width must be 1..%d", who, MAXKW);
  NS_CHECK_ARG(p->Tia >= p->Ti && p->Tia % 4 == 0, "%s: Tia must be >= Ti and a multiple of 4", who);
  NS_CHECK_ARG(p->keys_t != nullptr, "%s: keys_t scratch missing", who);
  NS_CHECK_ARG(p->Dsp >= 0 && p->Dsp % 8 == 0, "%s: speaker projection width must be a multiple of 8", who);
  return NS_OK;
}

static size_t ctx_lds(const ns_taco2_attn_params& p) { return sizeof(float) * ((size_t)p.Tia + 32 + 4 * CCH); }
static size_t dq_lds(const ns_taco2_attn_params& p) {
  return sizeof(float) * ((size_t)p.Ti + 2 * PADK + p.Tia + AW * 64);
}

// projected-memory form, step 0: p1[slot 1] = relu(f1[slot 1]) (the context before the first step is zero)
template <typename T>
__global__ void attn_p1_init_kernel(const float* f1, T* p1, int N, long S1, int D1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * D1) return;
  const long o = ((long)(i / D1) * S1 + 1) * D1 + i % D1;
  stf(p1 + o, fmaxf(f1[o], 0.f));
}

// Projected-memory form: the contexts feed nothing inside the loop, so they are one product per batch item over all
// steps afterwards, hc[n, 1..S, A:] = align[n, 1..S, :] . memory[n]  (also the tail of the persistent forward kernel)
int ns_attn_contexts_after_loop(const ns_taco2_attn_params& p, hipStream_t s) {
  const long S1 = p.S + 1, A = p.A, E = p.E, HC = A + E;
  const long esz = p.dtype == NS_BF16 ? 2 : 4;
  NS_CHECK_ARG(p.align_t != nullptr, "ns_taco2_attn_fwd: the projected-memory form needs align_t");
  ns_gemm_params g = {};
  g.dtype = p.dtype; g.M = p.S; g.N = (int)E; g.K = p.Ti;
  g.A = (const char*)p.align_t + p.Tia * esz; g.lda = p.Tia; g.a_mode = 0;
  g.B = (const char*)p.values + (long)p.padl_i * E * esz; g.ldb = E; g.b_mode = 1;
  g.C = (char*)p.hc + (HC + A) * esz; g.ldc = HC; g.c_dtype = p.dtype;
  g.batch = p.N; g.batch_stride_a = S1 * p.Tia; g.batch_stride_b = (long)p.Pi * E; g.batch_stride_c = S1 * HC;
  g.alpha = 1.f; g.split_k = 1; g.f32_passes = p.f32_passes;
  return ns_gemm(&g, s);
}

template <typename T>
static int attn_fwd_t(const ns_taco2_attn_params& p, hipStream_t s) {
  const long S1 = p.S + 1, A = p.A, E = p.E, D1 = p.D1, D2 = p.D2;
  const long Dsp = p.Dsp, XA = D2 + Dsp + A, HC = A + E;
  const int dt = p.dtype;
  const size_t lds = ctx_lds(p);
  NS_CHECK_ARG(lds <= 64 * 1024, "ns_taco2_attn_fwd: T_in too long for LDS (%zu bytes)", lds);
  NS_CHECK_ARG(p.work != nullptr, "ns_taco2_attn_fwd: work buffer missing");
  float* e_raw = p.work + (size_t)p.N * p.E;
  hipLaunchKernelGGL(keys_transpose_kernel, dim3(ceil_div(p.Tia, 32), ceil_div(p.A, 32), p.N), dim3(256), 0, s,
                     p.keys, p.keys_t, p.Ti, p.Tia, p.Pi, p.padl_i, p.A, 0);
  NS_CHECK_LAUNCH("keys_transpose");
  const bool pvm = p.pv != nullptr;
  T* p1_spill = (T*)(e_raw + (size_t)p.N * p.Tia);      // [N, D1]: where the last step's (unused) next-p1 goes
  if (pvm)
    hipLaunchKernelGGL(attn_p1_init_kernel<T>, dim3(ceil_div(p.N * (int)D1, 256)), dim3(256), 0, s, p.f1, (T*)p.p1, p.N, S1, (int)D1);
  for (int st = 0; st < p.S; ++st) {
    const long slot = st + 1, prev = st;
    T* hc = (T*)p.hc; T* xa = (T*)p.xa; T* p1 = (T*)p.p1;
    int rc;
    for (int nb = 0; nb < p.N; nb += 32) {
      const int nn = min(32, p.N - nb);
      // p1 = relu(ctx_prev . W1c + F1)   (projected-memory form: already written by the previous step's context kernel)
      if (!pvm) {
        rc = gemm_small(dt, nn, D1, E, hc + (nb * S1 + prev) * HC + A, S1 * HC, p.w1cT, E, p1 + (nb * S1 + slot) * D1,
                        S1 * D1, dt, nullptr, NS_ACT_RELU, p.f1 + (nb * S1 + slot) * D1, S1 * D1, nullptr, 0, s);
        if (rc) return rc;
      }
      // p2 = relu(p1 . W2 + b2) -> xa[:, 0:D2]
      rc = gemm_small(dt, nn, D2, D1, p1 + (nb * S1 + slot) * D1, S1 * D1, p.w2T, D1, xa + (nb * S1 + slot) * XA,
                      S1 * XA, dt, p.b2, NS_ACT_RELU, nullptr, 0, nullptr, 0, s);
      if (rc) return rc;
    }
    // attention LSTM on [p2 | h_prev]
    LstmStep<T> l = {};
    l.N = p.N; l.H = p.A; l.K = (int)XA; l.forget_bias = 1.0f; l.cell_clip = p.cell_clip;
    l.a = xa + slot * XA; l.a_sn = S1 * XA; l.wT = (const T*)p.wattT; l.bias = p.batt;
    l.c_prev = st > 0 ? p.ca + prev * A : nullptr; l.c_sn = S1 * A;
    l.h_out = hc + slot * HC; l.h_sn = S1 * HC;
    if (st + 1 < p.S) { l.h_out2 = xa + (slot + 1) * XA + D2 + Dsp; l.h2_sn = S1 * XA; }
    l.c_out = p.ca + slot * A; l.co_sn = S1 * A;
    l.gates_out = (T*)p.ga + slot * 4 * A; l.g_sn = S1 * 4 * A;
    l.passes = p.f32_passes;
    l.wT_hi = (const bf16_t*)p.wattT_hi; l.wT_lo = (const bf16_t*)p.wattT_lo;
    rc = lstm_step_launch<T>(l, s);
    if (rc) return rc;
    // q = h . Wq
    for (int nb = 0; nb < p.N; nb += 32) {
      const int nn = min(32, p.N - nb);
      rc = gemm_small(dt, nn, (int)A, (int)A, hc + (nb * S1 + slot) * HC, S1 * HC, p.wqT, A,
                      p.q + (nb * S1 + slot) * A, S1 * A, NS_F32, nullptr, NS_ACT_NONE, nullptr, 0, nullptr, 0, s);
      if (rc) return rc;
    }
    AttnStep<T> a = {};
    a.Ti = p.Ti; a.A = p.A; a.E = p.E; a.kw = p.kw; a.Tia = p.Tia; a.lengths = p.lengths;
    a.keys_t = p.keys_t;
    a.values = (const T*)p.values + (long)p.padl_i * E; a.values_sn = (long)p.Pi * E;
    a.q = p.q + slot * A; a.q_sn = S1 * A;
    a.aprev = p.align + prev * p.Tia; a.aout = p.align + slot * p.Tia; a.al_sn = S1 * p.Tia;
    a.aout_t = p.align_t ? (T*)p.align_t + slot * p.Tia : nullptr;
    a.ctx_out = hc + slot * HC + A; a.ctx_sn = S1 * HC;
    a.wcl = p.wcl; a.v = p.v; a.e_raw = e_raw;
    hipLaunchKernelGGL(attn_energy_kernel<T>, dim3(ceil_div(p.Ti, 64), p.N), dim3(ATHREADS), 0, s, a);
    if (pvm) {
      // next step's prenet layer straight from the alignments: p1[slot+1] = relu(align . (memory . W1c) + f1[slot+1])
      AttnStep<T> c = a;
      c.E = (int)D1;
      c.values = (const T*)p.pv + (long)p.padl_i * D1; c.values_sn = (long)p.Pi * D1;
      if (st + 1 < p.S) {
        c.ctx_out = p1 + (slot + 1) * D1; c.ctx_sn = S1 * D1;
        c.add = p.f1 + (slot + 1) * D1; c.add_sn = S1 * D1;
      } else {
        c.ctx_out = p1_spill; c.ctx_sn = D1;
      }
      c.relu = 1;
      hipLaunchKernelGGL(attn_context_kernel<T>, dim3(ceil_div((int)D1, CCH), p.N), dim3(256), lds, s, c);
    } else {
      hipLaunchKernelGGL(attn_context_kernel<T>, dim3(ceil_div(p.E, CCH), p.N), dim3(256), lds, s, a);
    }
    NS_CHECK_LAUNCH("attn_fwd");
  }
  if (pvm) return ns_attn_contexts_after_loop(p, s);
  return NS_OK;
}

extern "C" int ns_taco2_attn_fwd(const ns_taco2_attn_params* p, ns_stream_t s) {
  int rc = check_attn(p, "ns_taco2_attn_fwd");
  if (rc) return rc;
  NS_CHECK_ARG(p->keys && p->values && p->f1 && p->w1cT && p->w2T && p->wattT && p->wqT && p->b2 && p->batt &&
                   p->wcl && p->v && p->p1 && p->xa && p->hc && p->ca && p->ga && p->q && p->align,
               "ns_taco2_attn_fwd: null pointer");
  g_f32_passes = p->f32_passes;
  if (p->dtype == NS_BF16) return attn_fwd_t<bf16_t>(*p, (hipStream_t)s);
  return attn_fwd_t<float>(*p, (hipStream_t)s);
}

// second stage of attn_post_kernel's sums with AttnPost.part: 64 (k, unit) columns per workgroup x 4 groups of shares; a
// thread adds its group's shares in block order (eight loads in flight), the four group sums are added in group order
__global__ __launch_bounds__(256) void attn_post_finish_kernel(const float* part, int nblocks, int A, int kw, float* dv, float* dwcl) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;
  const bool ok = i < (1 + kw) * A;
  const int k = ok ? i / A : 0, u = ok ? i - k * A : 0;
  const int per = (nblocks + 3) / 4, b0 = grp * per, b1 = min(nblocks, b0 + per);
  float s = 0.f;
  for (int b = b0; b < b1; b += 8) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (ok && b + j < b1) ? part[((long)(b + j) * (1 + MAXKW) + k) * A + u] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
  }
  red[grp][c] = s;
  __syncthreads();
  if (grp == 0 && ok) {
    const float t = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
    if (k == 0) dv[u] += t; else dwcl[(k - 1) * A + u] += t;
  }
}
static int launch_attn_post(const AttnPost& q0, int N, hipStream_t s) {
  AttnPost q = q0;
  static const int dbg_env = [] { const char* e = getenv("NS_POST_DBG"); return e ? atoi(e) : 0; }();
  q.dbg = dbg_env;
  dim3 grid(ceil_div(q.Tia, 64), ceil_div(q.A, 4 * PU), N);
  if (q.kw == 7 && !(q.dbg & 2)) hipLaunchKernelGGL(attn_post_kernel<7>, grid, dim3(256), 0, s, q);
  else hipLaunchKernelGGL(attn_post_kernel<0>, grid, dim3(256), 0, s, q);
  if (q.part)
    hipLaunchKernelGGL(attn_post_finish_kernel, dim3(ceil_div((1 + q.kw) * q.A, 64)), dim3(256), 0, s, q.part, (int)(grid.x * N),
                       q.A, q.kw, q.dv, q.dwcl);
  return NS_OK;
}
extern "C" size_t ns_attention_post_part_floats(int N, int Tia, int A) {
  if (N <= 0 || Tia <= 0 || A <= 0) return 0;
  return (size_t)N * ceil_div(Tia, 64) * (1 + MAXKW) * (size_t)(ceil_div(A, 4 * PU) * 4 * PU);
}

// Hoisted part of the backward pass, after the time loop (also the tail of the persistent backward kernel): the sums
// over all steps that no recurrence needs (dkeys, dv, dWcl), the total context gradients and dvalues.
template <typename T>
int ns_attn_bwd_post(const ns_taco2_attn_params& p, hipStream_t s) {
  const long S1 = p.S + 1, A = p.A, E = p.E, D1 = p.D1, HC = A + E;
  const int dt = p.dtype;
  const bool pvm = p.pv != nullptr;
  float* dkeys_t = p.work + (size_t)p.N * (E + (1 + MAXKW) * p.Tia + 2 * A);
  // ---- hoisted sums over all steps
  {
    AttnPost q = {};
    q.S = p.S; q.Ti = p.Ti; q.A = p.A; q.kw = p.kw; q.Tia = p.Tia; q.lengths = p.lengths;
    q.keys_t = p.keys_t; q.q = p.q; q.align = p.align; q.de = p.de; q.wcl = p.wcl; q.v = p.v;
    q.dkeys_t = dkeys_t; q.dv = p.dv; q.dwcl = p.dwcl; q.part = p.post_part;
    launch_attn_post(q, p.N, s);
    NS_CHECK_LAUNCH("attn_post");
  }
  if (pvm) {
    // total context gradients of all steps in one product: dctx[n, slot] = dhc[n, slot, A:] + df1[n, slot+1] . W1c^T
    // (df1 slot 0 rows are never written, so the row behind an item's last slot contributes zero; the caller keeps
    // one zero row behind the end of df1 for the very last one)
    ns_gemm_params g = {};
    g.dtype = dt; g.M = (int)(p.N * S1); g.N = (int)E; g.K = (int)D1;
    g.A = (const T*)p.df1 + D1; g.lda = D1; g.a_mode = 0;
    g.B = p.w1c; g.ldb = D1; g.b_mode = 0;
    g.C = p.dctx_t; g.ldc = E; g.c_dtype = dt;
    g.addend = p.dhc + A; g.ld_add = HC; g.addend_dtype = NS_F32;
    g.alpha = 1.f; g.split_k = 1; g.f32_passes = p.f32_passes;
    int rc = ns_gemm(&g, s);
    if (rc) return rc;
  }
  // dvalues[n] += align[n]^T . dctx[n]   (contraction over the decoder steps)
  {
    ns_gemm_params g = {};
    g.dtype = dt; g.M = p.Ti; g.N = (int)E; g.K = (int)S1;
    g.A = (const T*)p.align_t; g.lda = p.Tia; g.a_mode = 1;
    g.B = (const T*)p.dctx_t; g.ldb = E; g.b_mode = 1;
    g.C = p.dvalues + (long)p.padl_i * E; g.ldc = E; g.c_dtype = NS_F32;
    g.batch = p.N; g.batch_stride_a = S1 * p.Tia; g.batch_stride_b = S1 * E; g.batch_stride_c = (long)p.Pi * E;
    g.accumulate = 1; g.alpha = 1.f; g.split_k = 1; g.f32_passes = p.f32_passes;
    int rc = ns_gemm(&g, s);
    if (rc) return rc;
  }
  // dkeys[t][u] += dkeys_t[u][t]
  hipLaunchKernelGGL(keys_transpose_kernel, dim3(ceil_div(p.Tia, 32), ceil_div(p.A, 32), p.N), dim3(256), 0, s,
                     (const float*)p.dkeys, dkeys_t, p.Ti, p.Tia, p.Pi, p.padl_i, p.A, 1);
  NS_CHECK_LAUNCH("keys_transpose_back");
  return NS_OK;
}
template int ns_attn_bwd_post<float>(const ns_taco2_attn_params&, hipStream_t);
template int ns_attn_bwd_post<bf16_t>(const ns_taco2_attn_params&, hipStream_t);

template <typename T>
static int attn_bwd_t(const ns_taco2_attn_params& p, hipStream_t s) {
  const long S1 = p.S + 1, A = p.A, E = p.E, D1 = p.D1, D2 = p.D2;
  const long Dsp = p.Dsp, XA = D2 + Dsp + A, HC = A + E;
  const int dt = p.dtype;
  float* dctx_carry = p.work;
  float* da = dctx_carry + (size_t)p.N * E;
  float* gk = da + (size_t)p.N * p.Tia;
  float* dhq = gk + (size_t)p.N * p.Tia * MAXKW;
  float* dc_carry = dhq + (size_t)p.N * A;
  float* dkeys_t = dc_carry + (size_t)p.N * A;
  const size_t lds = dq_lds(p);
  NS_CHECK_ARG(lds <= 64 * 1024 && sizeof(float) * p.E <= 64 * 1024,
               "ns_taco2_attn_bwd: T_in / memory depth too large for LDS");
  T* xa = (T*)p.xa; T* p1 = (T*)p.p1;
  const bool pvm = p.pv != nullptr;
  NS_CHECK_ARG(!pvm || p.da0 != nullptr, "ns_taco2_attn_bwd: the projected-memory form needs da0");
  for (int st = p.S - 1; st >= 0; --st) {
    const long slot = st + 1, prev = st;
    const bool last = (st == p.S - 1);
    int rc;
    AttnBwdStep<T> a = {};
    a.Ti = p.Ti; a.A = p.A; a.E = p.E; a.kw = p.kw; a.Tia = p.Tia; a.lengths = p.lengths;
    a.keys = p.keys + (long)p.padl_i * A; a.keys_sn = (long)p.Pi * A;
    a.keys_t = p.keys_t;
    a.values = (const T*)p.values + (long)p.padl_i * E; a.values_sn = (long)p.Pi * E;
    a.q = p.q + slot * A; a.q_sn = S1 * A;
    a.acur = p.align + slot * p.Tia; a.aprev = p.align + prev * p.Tia; a.al_sn = S1 * p.Tia;
    a.dctx_ext = p.dhc + slot * HC + A; a.dce_sn = S1 * HC;
    a.dctx_carry = last ? nullptr : dctx_carry;
    a.gk = gk; a.da = da; a.has_carry = last ? 0 : 1;
    a.dq_out = (T*)p.dq + slot * A; a.dq_sn = S1 * A;
    a.de_out = p.de + slot * p.Tia;
    a.dctx_out = (T*)p.dctx_t + slot * E; a.dco_sn = S1 * E;
    a.wcl = p.wcl; a.v = p.v;
    if (pvm) {
      // da = da0 (hoisted: d(hc context) . memory^T) + dp1[s+1] . (memory . W1c)^T
      AttnBwdStep<T> c = a;
      c.pv_mode = 1;
      c.E = (int)D1;
      c.values = (const T*)p.pv + (long)p.padl_i * D1; c.values_sn = (long)p.Pi * D1;
      c.dvec = last ? nullptr : (const T*)p.df1 + (slot + 1) * D1; c.dv_sn = S1 * D1;
      c.da0 = p.da0 + slot * p.Tia; c.da0_sn = S1 * p.Tia;
      c.dctx_out = nullptr;
      hipLaunchKernelGGL(attn_bwd_da_kernel<T>, dim3(ceil_div(p.Tia, 32), p.N), dim3(ATHREADS), sizeof(float) * D1, s, c);
    } else
    hipLaunchKernelGGL(attn_bwd_da_kernel<T>, dim3(ceil_div(p.Tia, 32), p.N), dim3(ATHREADS), sizeof(float) * p.E, s, a);
    hipLaunchKernelGGL(attn_bwd_energy_kernel<T>, dim3(ceil_div(p.Ti, 64), p.N), dim3(ATHREADS), 0, s, a);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<T>, dim3(ceil_div(p.A, 64), p.N), dim3(ATHREADS), lds, s, a);
    NS_CHECK_LAUNCH("attn_bwd");
    // dhq = dq . Wq^T
    for (int nb = 0; nb < p.N; nb += 32) {
      const int nn = min(32, p.N - nb);
      rc = gemm_small(dt, nn, (int)A, (int)A, (T*)p.dq + (nb * S1 + slot) * A, S1 * A, p.wq, A, dhq + (long)nb * A, A,
                      NS_F32, nullptr, NS_ACT_NONE, nullptr, 0, nullptr, 0, s);
      if (rc) return rc;
    }
    // attention LSTM cell: dh = dhc[:, :A] + dhq + dga[s+1] . Watt[D2:]^T
    LstmBwdStep<T> c = {};
    c.N = p.N; c.H = p.A; c.t = st; c.lengths = nullptr; c.K = (int)(4 * A);
    c.first = last ? 1 : 0;
    c.dg_next = last ? nullptr : (const T*)p.dga + (slot + 1) * 4 * A; c.dgn_sn = S1 * 4 * A;
    c.w = (const T*)p.watt + (D2 + Dsp) * 4 * A;
    c.dh_out = p.dhc + slot * HC; c.dho_sn = S1 * HC;
    c.dh_out2 = dhq; c.dho2_sn = A;
    c.gates = (const T*)p.ga + slot * 4 * A; c.g_sn = S1 * 4 * A;
    c.c = p.ca + slot * A; c.c_prev = st > 0 ? p.ca + prev * A : nullptr; c.c_sn = S1 * A;
    c.dc_carry = dc_carry;
    c.dgates = (T*)p.dga + slot * 4 * A; c.dg_sn = S1 * 4 * A;
    c.passes = p.f32_passes;
    c.w_bf16 = p.watt_bf16 ? (const bf16_t*)p.watt_bf16 + (D2 + Dsp) * 4 * A : nullptr;
    if (p.dga_bf16 && sizeof(T) == 4) {
      c.dgates_b = (bf16_t*)p.dga_bf16 + slot * 4 * A;
      c.dg_next_b = last ? nullptr : (const bf16_t*)p.dga_bf16 + (slot + 1) * 4 * A;
    }
    rc = lstm_bwd_step_launch<T>(c, s);
    if (rc) return rc;
    for (int nb = 0; nb < p.N; nb += 32) {
      const int nn = min(32, p.N - nb);
      const T* dga = (const T*)p.dga + (nb * S1 + slot) * 4 * A;
      // dp2pre = (dga . Watt[0:D2]^T) * (p2 > 0)
      rc = gemm_small(dt, nn, (int)D2, (int)(4 * A), dga, S1 * 4 * A, p.watt, 4 * A, (T*)p.dp2 + (nb * S1 + slot) * D2,
                      S1 * D2, dt, nullptr, NS_ACT_NONE, nullptr, 0, xa + (nb * S1 + slot) * XA, S1 * XA, s);
      if (rc) return rc;
      // dp1pre = (dp2pre . W2^T) * (p1 > 0)
      rc = gemm_small(dt, nn, (int)D1, (int)D2, (T*)p.dp2 + (nb * S1 + slot) * D2, S1 * D2, p.w2, D2,
                      (T*)p.df1 + (nb * S1 + slot) * D1, S1 * D1, dt, nullptr, NS_ACT_NONE, nullptr, 0,
                      p1 + (nb * S1 + slot) * D1, S1 * D1, s);
      if (rc) return rc;
      // dctx_carry = dp1pre . W1c^T   (projected-memory form: folded into the next da kernel and the hoisted product)
      if (st > 0 && !pvm) {
        rc = gemm_small(dt, nn, (int)E, (int)D1, (T*)p.df1 + (nb * S1 + slot) * D1, S1 * D1, p.w1c, D1,
                        dctx_carry + (long)nb * E, E, NS_F32, nullptr, NS_ACT_NONE, nullptr, 0, nullptr, 0, s);
        if (rc) return rc;
      }
    }
  }
  return ns_attn_bwd_post<T>(p, s);
}

extern "C" int ns_taco2_attn_bwd(const ns_taco2_attn_params* p, ns_stream_t s) {
  int rc = check_attn(p, "ns_taco2_attn_bwd");
  if (rc) return rc;
  NS_CHECK_ARG(p->keys && p->values && p->w1c && p->w2 && p->watt && p->wq && p->wcl && p->v && p->p1 && p->xa &&
                   p->ca && p->ga && p->q && p->align && p->dhc && p->df1 && p->dp2 && p->dga && p->dq &&
                   p->dkeys && p->dvalues && p->dv && p->dwcl && p->work && p->align_t && p->de && p->dctx_t,
               "ns_taco2_attn_bwd: null pointer");
  g_f32_passes = p->f32_passes;
  if (p->dtype == NS_BF16) return attn_bwd_t<bf16_t>(*p, (hipStream_t)s);
  return attn_bwd_t<float>(*p, (hipStream_t)s);
}

extern "C" int ns_taco2_keys_transpose(const float* keys, float* keys_t, int N, int Ti, int Tia, int Pi, int padl,
                                       int A, ns_stream_t s) {
  NS_CHECK_ARG(keys && keys_t, "ns_taco2_keys_transpose: null");
  hipLaunchKernelGGL(keys_transpose_kernel, dim3(ceil_div(Tia, 32), ceil_div(A, 32), N), dim3(256), 0, (hipStream_t)s,
                     keys, keys_t, Ti, Tia, Pi, padl, A, 0);
  NS_CHECK_LAUNCH("keys_transpose");
  return NS_OK;
}

extern "C" int ns_attention_step(const ns_attention_step_params* p, ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  NS_CHECK_ARG(p && p->keys_t && p->values && p->q && p->aprev && p->aout && p->ctx_out && p->wcl && p->v && p->e_raw,
               "ns_attention_step: null");
  NS_CHECK_ARG(p->A % 8 == 0 && p->A <= 256 && p->E % 8 == 0 && p->kw >= 1 && p->kw <= MAXKW && p->Tia >= p->Ti,
               "ns_attention_step: unsupported shape");
  const size_t lds = sizeof(float) * ((size_t)p->Tia + 32 + 4 * CCH);
  NS_CHECK_ARG(lds <= 64 * 1024, "ns_attention_step: T_in too long for LDS");
  NS_CHECK_ARG(!p->pv || (p->pv_out && p->E2 > 0 && p->E2 % 8 == 0), "ns_attention_step: pv needs pv_out and E2 %% 8 == 0");
  NS_CHECK_ARG(!p->ctx_rows || (p->N <= 32 && p->ctx_rows_col >= 0 && p->ctx_rows_col + p->E <= ((p->ctx_rows_K + 31) / 32) * 32),
               "ns_attention_step: ctx_rows needs N <= 32 and the context columns inside the packed rows");
  auto run = [&](auto tag) -> int {
    using T = decltype(tag);
    AttnStep<T> a = {};
    a.Ti = p->Ti; a.A = p->A; a.E = p->E; a.kw = p->kw; a.Tia = p->Tia; a.lengths = p->lengths;
    a.keys_t = p->keys_t;
    a.values = (const T*)p->values + (long)p->padl_i * p->E; a.values_sn = (long)p->Pi * p->E;
    a.q = p->q; a.q_sn = p->q_sn;
    a.aprev = p->aprev; a.aout = p->aout; a.al_sn = p->al_sn; a.aout_t = nullptr;
    a.ctx_out = (T*)p->ctx_out; a.ctx_sn = p->ctx_sn; a.ctx_out2 = (T*)p->ctx_out2; a.ctx2_sn = p->ctx2_sn;
    a.wcl = p->wcl; a.v = p->v; a.e_raw = p->e_raw;
    a.nch = ceil_div(p->E, CCH);
    int chunks = a.nch;
    if (p->pv) {
      a.pv = (const T*)p->pv + (long)p->padl_i * p->E2; a.pv_sn = (long)p->Pi * p->E2; a.E2 = p->E2;
      a.pv_out = (T*)p->pv_out; a.pvo_sn = p->pv_out_sn;
      chunks += ceil_div(p->E2, CCH);
    }
    if (p->ctx_rows) {
      a.ctx_rows = (bf16_t*)p->ctx_rows; a.cr_nkc = (p->ctx_rows_K + 31) / 32; a.cr_col = p->ctx_rows_col;
    }
    hipLaunchKernelGGL(attn_energy_kernel<T>, dim3(ceil_div(p->Ti, 64), p->N), dim3(ATHREADS), 0, s, a);
    hipLaunchKernelGGL(attn_context_kernel<T>, dim3(chunks, p->N), dim3(256), lds, s, a);
    NS_CHECK_LAUNCH("attention_step");
    return NS_OK;
  };
  if (p->dtype == NS_BF16) return run(bf16_t{});
  return run(float{});
}

extern "C" int ns_taco2_keys_transpose_add(float* keys, const float* keys_t, int N, int Ti, int Tia, int Pi, int padl,
                                           int A, ns_stream_t s) {
  NS_CHECK_ARG(keys && keys_t, "ns_taco2_keys_transpose_add: null");
  hipLaunchKernelGGL(keys_transpose_kernel, dim3(ceil_div(Tia, 32), ceil_div(A, 32), N), dim3(256), 0, (hipStream_t)s,
                     (const float*)keys, (float*)keys_t, Ti, Tia, Pi, padl, A, 1);
  NS_CHECK_LAUNCH("keys_transpose_add");
  return NS_OK;
}

extern "C" int ns_attention_step_bwd(const ns_attention_step_bwd_params* p, ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  NS_CHECK_ARG(p && p->keys && p->keys_t && p->values && p->q && p->acur && p->aprev && p->dctx_ext && p->gk && p->da &&
                   p->dq_out && p->de_out && p->dctx_out && p->wcl && p->v, "ns_attention_step_bwd: null");
  NS_CHECK_ARG(p->A % 8 == 0 && p->A <= 256 && p->E % 8 == 0 && p->kw >= 1 && p->kw <= MAXKW && p->Tia >= p->Ti,
               "ns_attention_step_bwd: unsupported shape");
  const size_t lds = sizeof(float) * ((size_t)p->Ti + 2 * PADK + p->Tia + AW * 64);
  NS_CHECK_ARG(lds <= 64 * 1024 && sizeof(float) * p->E <= 64 * 1024, "ns_attention_step_bwd: too large for LDS");
  auto run = [&](auto tag) -> int {
    using T = decltype(tag);
    AttnBwdStep<T> a = {};
    a.Ti = p->Ti; a.A = p->A; a.E = p->E; a.kw = p->kw; a.Tia = p->Tia; a.lengths = p->lengths;
    a.keys = p->keys + (long)p->padl_i * p->A; a.keys_sn = (long)p->Pi * p->A;
    a.keys_t = p->keys_t;
    a.values = (const T*)p->values + (long)p->padl_i * p->E; a.values_sn = (long)p->Pi * p->E;
    a.q = p->q; a.q_sn = p->q_sn;
    a.acur = p->acur; a.aprev = p->aprev; a.al_sn = p->al_sn;
    a.dctx_ext = p->dctx_ext; a.dce_sn = p->dce_sn; a.dctx_carry = p->dctx_carry;
    a.gk = p->gk; a.da = p->da; a.has_carry = p->has_carry;
    a.dq_out = (T*)p->dq_out; a.dq_sn = p->dq_sn; a.de_out = p->de_out;
    a.dctx_out = (T*)p->dctx_out; a.dco_sn = p->dco_sn;
    a.wcl = p->wcl; a.v = p->v;
    hipLaunchKernelGGL(attn_bwd_da_kernel<T>, dim3(ceil_div(p->Tia, 32), p->N), dim3(ATHREADS), sizeof(float) * p->E, s, a);
    hipLaunchKernelGGL(attn_bwd_energy_kernel<T>, dim3(ceil_div(p->Ti, 64), p->N), dim3(ATHREADS), 0, s, a);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<T>, dim3(ceil_div(p->A, 64), p->N), dim3(ATHREADS), lds, s, a);
    NS_CHECK_LAUNCH("attention_step_bwd");
    return NS_OK;
  };
  if (p->dtype == NS_BF16) return run(bf16_t{});
  return run(float{});
}

extern "C" int ns_attention_post_bwd(const ns_attention_post_bwd_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->keys_t && p->q && p->align && p->de && p->wcl && p->v && p->dkeys_t && p->dv && p->dwcl,
               "ns_attention_post_bwd: null");
  AttnPost q = {};
  q.S = p->S; q.Ti = p->Ti; q.A = p->A; q.kw = p->kw; q.Tia = p->Tia; q.lengths = p->lengths;
  q.keys_t = p->keys_t; q.q = p->q; q.align = p->align; q.de = p->de; q.wcl = p->wcl; q.v = p->v;
  q.dkeys_t = p->dkeys_t; q.dv = p->dv; q.dwcl = p->dwcl; q.part = p->part;
  launch_attn_post(q, p->N, (hipStream_t)s);
  NS_CHECK_LAUNCH("attention_post_bwd");
  return NS_OK;
}
