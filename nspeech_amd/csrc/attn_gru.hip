// Tacotron-1 attention RNN (decoder prenet layer 2 -> GRU(256) attention cell -> query -> Bahdanau energies -> softmax ->
// next step's prenet layer 1) as ONE persistent launch per direction instead of ~9 (forward) / ~16 (backward) dependent
// launches per decoder step (tacotron.py:64-76, modules.py:76-102, rnn_wrappers.py:25-31 under teacher forcing).
//
// Same arrangement as attn_cluster.hip (the Tacotron-2 counterpart): every operation of the chain is independent per
// utterance and its weights are stationary (0.39 M values), so an utterance gets a CLUSTER of 8 workgroups (one per CU,
// 512 threads) that keeps everything on chip for the whole sequence - the weights in REGISTERS as fp32 (one utterance
// per cluster makes every product matrix-VECTOR: exact fp32 FMAs), the utterance's keys and projected memory
// (pv = values . W1c) in LDS:
//   workgroup g owns  * GRU units [32 g, 32 g + 32): the r, u columns of gates/kernel and the columns of candidate/kernel
//                       of those units, all K = 128 + 256 input rows
//                     * the rows of W_query that belong to those units
//                     * all of W_prenet2 (every workgroup computes the whole p2: cheaper than another exchange)
//                     * memory positions [g ts, (g + 1) ts), ts = ceil(length / 8): keys and pv rows
// A GRU step has TWO dependent products (the candidate needs r * h of EVERY unit), so the forward step takes three
// exchanges inside the cluster where the LSTM cell of Tacotron-2 takes two (8-byte {step tag, fp32} granules written and
// polled with relaxed agent-scope atomics; the data is its own flag, cdna guide G16):
//   X1: r * h(s-1) of the own units                                                -> every workgroup has r * h
//   X2: partial queries h_own . Wq[own rows, :] (256 values) + the new h of the own units   -> q, h
//   X3: partial next-prenet sums  sum_t w[t] pv[t, :] (256) + local softmax max / sum + the local weights
//       (combined flash-attention style)                                           -> p1[s+1], the alignment
// Backward mirrors it with four: the softmax-backward dot-product share (E1, hidden behind the energy pass), partial
// query gradients (E2), partial input gradients of the candidate kernel (E3a: they carry d(r * h)), then of the gate
// kernel (E3b).  One buffer per exchange suffices: a workgroup publishes exchange X of step s+1 only after it has gathered
// the last exchange of step s from every peer, and a peer publishes that only after it has gathered X of step s.
// Every spin is bounded in wall-clock time; on a time-out the status word is raised and every workgroup of the cluster
// leaves.  History is written in the layouts models/tacotron.py reads (the launch-per-step form stays as the fallback).
#include "common.h"
#include <stdlib.h>

typedef unsigned long long u64;
namespace {
constexpr int CG = 8;              // workgroups per utterance
constexpr int CT = 512;            // threads per workgroup
constexpr int TSMAX = 32;          // memory positions per workgroup (T_in <= 256)
constexpr int KPAD = 4;            // row pad of the keys image (16 rows x one bank otherwise)
constexpr int A = 256, D1 = 256, D2 = 128;
constexpr int UPW = A / CG;        // 32 GRU units per workgroup
constexpr int K = D2 + A;          // input rows of both GRU kernels
// forward thread maps
constexpr int P2G = CT / D2, P2K = D1 / P2G;             // prenet 2: 4 k groups x 64
constexpr int GC = 2 * UPW, GG = CT / GC;                // gates: 64 columns (r | u of the own units), 8 k groups
constexpr int GKP = D2 / GG, GKH = A / GG;               //   16 prenet rows + 32 h rows per thread
constexpr int CGG = CT / UPW;                            // candidate: 32 columns, 16 k groups
constexpr int CKP = D2 / CGG, CKH = A / CGG;             //   8 prenet rows + 16 r*h rows per thread
constexpr int QG = CT / A, QK = UPW / QG;                // query partials: 2 groups x 16 own rows
constexpr int CXG = CT / D1;                             // context t-groups
constexpr int X1N = UPW, X2N = A + UPW, X3N = D1 + 2 + TSMAX;
constexpr int XMAX = X3N;
// backward
constexpr int NQ = 4;                                    // column groups of the input-gradient products
constexpr int PPT = NQ * K / CT;                         // (row, group) pairs per thread: 3
constexpr int CPQ_C = UPW / NQ, CPQ_G = GC / NQ;         // columns per group: candidate 8, gates 16
constexpr int W2H = CT / D1, W2K = D2 / W2H;             // dp1 product: 2 halves x 64
static_assert(NQ * K == PPT * CT && P2K == 64 && GKP + GKH == 48 && CKP + CKH == 24, "shape");

struct TArgs {
  ns_taco1_attn_params p;
  u64* x1; u64* x2; u64* x3; u64* x4;       // fwd: [N][CG][X1N | X2N | X3N]; bwd: E1 [N][CG], E2 [N][CG][A], E3a / E3b [N][CG][K]
  int* status;
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void put_granule(u64* g_, unsigned tag, float v) {
  NS_GLOBAL u64* g = (NS_GLOBAL u64*)g_;
  __hip_atomic_store(g, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every thread waits for its own granules (PER per thread, stride CT) and drops the values into LDS.  Returns false
// when this thread gave up (a peer raised the status word, or the wall-clock bound passed).
template <int PER>
__device__ __forceinline__ bool gather_granules(const u64* src_, int total, unsigned tag, float* dst, int tid, int* status, int code) {
  const NS_GLOBAL u64* src = (const NS_GLOBAL u64*)src_;
  u64 v[PER];
  unsigned spins = 0, clk0 = 0;
  bool ok, gave_up = false;
  do {
    ok = true;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid + j * CT;
      v[j] = i < total ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((u64)tag << 32);
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) ok = ok && ((unsigned)(v[j] >> 32) == tag);
    if (!ok && (++spins & 1023u) == 0) {
      if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = gave_up = true;
      else if (ns_spin_timed_out(clk0)) { atomicExch(status, code); ok = gave_up = true; }
    }
  } while (!ok);
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid + j * CT;
    if (i < total) dst[i] = __uint_as_float((unsigned)v[j]);
  }
  return !gave_up;
}

// sum_i w[i] * x[i], weights in registers, x in LDS (wave-uniform addresses); chunks fenced so that at most two chunks of
// x are in flight (attn_cluster.hip: dot_regs)
template <int NK>
__device__ __forceinline__ float dot_regs(const float (&w)[NK], const float* x) {
  static_assert(NK % 4 == 0, "chunking");
  float s0 = 0.f, s1 = 0.f;
  float4 cur = *(const float4*)x;
#pragma unroll
  for (int i = 0; i < NK; i += 4) {
    float4 nxt = cur;
    if (i + 4 < NK) nxt = *(const float4*)(x + i + 4);
    s0 = fmaf(w[i], cur.x, s0); s1 = fmaf(w[i + 1], cur.y, s1);
    s0 = fmaf(w[i + 2], cur.z, s0); s1 = fmaf(w[i + 3], cur.w, s1);
    asm volatile("" : "+v"(s0), "+v"(s1) :: "memory");
    cur = nxt;
  }
  return s0 + s1;
}

// ===================================================================================== forward
template <typename T>
__global__ __launch_bounds__(CT) void taco1_attn_fwd_kernel(TArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const ns_taco1_attn_params& p = a.p;
  float* xs = sm;                          // [K]      p2 | h(s-1)
  float* p1s = xs + K;                     // [D1]     prenet layer 1 of the current step
  float* red = p1s + D1;                   // [CT]     partial sums across the k / t groups of a product
  float* rus = red + CT;                   // [GC]     r | u of the own units
  float* hloc = rus + GC;                  // [UPW]    h of the own units
  float* es = hloc + UPW;                  // [TSMAX]  local unnormalised softmax weights
  float* ered = es + TSMAX;                // [8][TSMAX] per-wave energy sums
  float* sc = ered + 8 * TSMAX;            // [16]     [0] local max, [1] local sum, [2] abort flag
  float* gath = sc + 16;                   // [CG][XMAX]
  float* vs = gath + CG * XMAX;            // [A]      attention_v
  float* keys_s = vs + A;                  // [TSMAX][A + KPAD]
  float* pv_s = keys_s + TSMAX * (A + KPAD);      // [TSMAX][D1]
  float* wq_s = pv_s + TSMAX * D1;         // [UPW][A]  W_query rows of the own units

  const int tid_ = threadIdx.x;
  const int n = blockIdx.x / CG, g = blockIdx.x % CG;
  const long S1 = p.S + 1;
  const int HC = A + p.E;
  // multi-speaker (rnn_wrappers.py:28-30): the GRU's input row is [p2 | speaker projection (Dsp) | h]; the projection is
  // the same in every step, so its products with the Dsp kernel rows join the biases and the loop never sees it
  const int Dsp = p.Dsp, XA = D2 + Dsp + A;
  const int L = min(p.lengths ? p.lengths[n] : p.Ti, p.Ti);
  const int ts = max(1, (L + CG - 1) / CG);
  const int t0 = g * ts, tn = max(0, min(L, t0 + ts) - t0);
  u64* x1 = a.x1 + (size_t)n * CG * X1N;
  u64* x2 = a.x2 + (size_t)n * CG * X2N;
  u64* x3 = a.x3 + (size_t)n * CG * X3N;
  if (tid_ == 0) sc[2] = 0.f;

  // ---------------------------------------------------------------- resident weights (registers)
  const T* W2 = (const T*)p.w2;            // [D1][D2]
  const T* Wg = (const T*)p.wg;            // [K][2A]
  const T* Wc = (const T*)p.wc;            // [K][A]
  const T* Wq = (const T*)p.wq;            // [A][A]
  const int tid = tid_;
  float w2r[P2K];
  {
    const T* b = W2 + (long)((tid / D2) * P2K) * D2 + tid % D2;
#pragma unroll
    for (int i = 0; i < P2K; ++i) w2r[i] = ldf(b + i * D2);
  }
  const int gc = tid % GC;
  const int gcol = gc < UPW ? g * UPW + gc : A + g * UPW + (gc - UPW);
  float wgp[GKP], wgh[GKH];
  {
    const T* bp = Wg + (long)((tid / GC) * GKP) * 2 * A + gcol;
    const T* bh = Wg + (long)(D2 + Dsp + (tid / GC) * GKH) * 2 * A + gcol;
#pragma unroll
    for (int i = 0; i < GKP; ++i) wgp[i] = ldf(bp + i * 2 * A);
#pragma unroll
    for (int i = 0; i < GKH; ++i) wgh[i] = ldf(bh + i * 2 * A);
  }
  float wcp[CKP], wch[CKH];
  {
    const int ccol = g * UPW + tid % UPW;
    const T* bp = Wc + (long)((tid / UPW) * CKP) * A + ccol;
    const T* bh = Wc + (long)(D2 + Dsp + (tid / UPW) * CKH) * A + ccol;
#pragma unroll
    for (int i = 0; i < CKP; ++i) wcp[i] = ldf(bp + i * A);
#pragma unroll
    for (int i = 0; i < CKH; ++i) wch[i] = ldf(bh + i * A);
  }
  for (int i = tid; i < UPW * A; i += CT) wq_s[i] = ldf(Wq + (long)g * UPW * A + i);
  for (int i = tid; i < A; i += CT) vs[i] = p.v[i];
  float gbias = tid < GC ? p.bg[gcol] : 0.f;
  float cbias = tid < UPW ? p.bc[g * UPW + tid] : 0.f;
  if (Dsp) {
    const T* spk = (const T*)p.xa + ((long)n * S1 + 1) * XA + D2;       // written into every slot by the caller
    for (int k = 0; k < Dsp; ++k) {
      const float sk = ldf(spk + k);
      if (tid < GC) gbias = fmaf(sk, ldf(Wg + (long)(D2 + k) * 2 * A + gcol), gbias);
      if (tid < UPW) cbias = fmaf(sk, ldf(Wc + (long)(D2 + k) * A + g * UPW + tid), cbias);
    }
  }
  const float b2c = tid < D2 ? p.b2[tid] : 0.f;
  float hpart = 0.f;                       // h(s-1) . Wg[h rows of this k group]: h(-1) = 0

  // ---------------------------------------------------------------- per-utterance LDS images
  {
    const float* kn = p.keys + ((long)n * p.Pi + p.padl_i + t0) * A;
    for (int i = tid; i < TSMAX * A; i += CT) keys_s[(i / A) * (A + KPAD) + i % A] = (i / A) < tn ? kn[i] : 0.f;
    const T* pvn = (const T*)p.pv + ((long)n * p.Pi + p.padl_i + t0) * D1;
    for (int i = tid; i < TSMAX * D1; i += CT) pv_s[i] = (i / D1) < tn ? ldf(pvn + i) : 0.f;
    for (int i = tid; i < K; i += CT) xs[i] = 0.f;
    if (tid < UPW) hloc[tid] = 0.f;
    if (tid < D1) p1s[tid] = fmaxf(p.f1[((long)n * S1 + 1) * D1 + tid], 0.f);     // the context before the first step is 0
  }
  if (tid < D1 && tid / (D1 / CG) == g) stf((T*)p.p1 + ((long)n * S1 + 1) * D1 + tid, fmaxf(p.f1[((long)n * S1 + 1) * D1 + tid], 0.f));
  __syncthreads();

  for (int st = 0; st < p.S; ++st) {
    const long slot = st + 1;
    const long rowS = (long)n * S1 + slot;
    const unsigned tag = (unsigned)(st + 1);
    int tid = tid_;
    asm volatile("" : "+v"(tid));           // opaque per iteration: addresses are recomputed in the loop, not hoisted as 64-bit pairs
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float f1n = 0.f;
    if (tid < D1 && st + 1 < p.S) f1n = p.f1[(rowS + 1) * D1 + tid];

    // ---- (1) p2 = relu(p1 . W2 + b2): every workgroup computes all of it
    red[tid] = dot_regs<P2K>(w2r, p1s + (tid / D2) * P2K);
    lds_barrier();
    if (tid < D2) {
      float s = b2c;
#pragma unroll
      for (int q = 0; q < P2G; ++q) s += red[q * D2 + tid];
      s = fmaxf(s, 0.f);
      xs[tid] = s;
      if (tid / (D2 / CG) == g) {
        stf((T*)p.xa + rowS * XA + tid, s);
        stf((T*)p.xc + rowS * XA + tid, s);
      }
    }
    lds_barrier();
    // ---- (2) r, u of the own units; r * h(s-1) goes out (X1); the candidate's prenet rows meanwhile
    red[tid] = hpart + dot_regs<GKP>(wgp, xs + (tid / GC) * GKP);
    const float cpart = dot_regs<CKP>(wcp, xs + (tid / UPW) * CKP);
    lds_barrier();
    float sv_r = 0.f, sv_u = 0.f, sv_rh = 0.f;
    if (tid < GC) {
      float z = gbias;
#pragma unroll
      for (int q = 0; q < GG; ++q) z += red[q * GC + tid];
      const float v = sigmoidf_(z);
      rus[tid] = v;
      if (tid < UPW) {
        sv_r = v;
        sv_rh = v * hloc[tid];
        put_granule(x1 + (size_t)g * X1N + tid, tag, sv_rh);
      }
    }
    // ---- (3) gather X1: r * h of every unit (granule index = unit)
    if (!gather_granules<1>(x1, CG * X1N, tag, gath, tid, a.status, 1)) sc[2] = 1.f;
    lds_barrier();
    if (sc[2] != 0.f) return;
    // ---- (4) candidate of the own units, the new h; h goes out with the partial queries (X2)
    red[tid] = cpart + dot_regs<CKH>(wch, gath + (tid / UPW) * CKH);
    lds_barrier();
    float sv_c = 0.f, sv_h = 0.f;
    if (tid < UPW) {
      float z = cbias;
#pragma unroll
      for (int q = 0; q < CGG; ++q) z += red[q * UPW + tid];
      sv_c = tanhf_(z);
      sv_u = rus[UPW + tid];
      sv_h = sv_u * hloc[tid] + (1.f - sv_u) * sv_c;
      hloc[tid] = sv_h;
      put_granule(x2 + (size_t)g * X2N + A + tid, tag, sv_h);
    }
    lds_barrier();
    {
      const int qu = tid % A, qq = tid / A;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < QK; ++i) s = fmaf(wq_s[(qq * QK + i) * A + qu], hloc[qq * QK + i], s);
      red[tid] = s;
    }
    lds_barrier();
    if (tid < A) put_granule(x2 + (size_t)g * X2N + tid, tag, red[tid] + red[A + tid]);
    // ---- (5) gather X2: q = sum of the partials (fixed order), h of every unit
    if (!gather_granules<(CG * X2N + CT - 1) / CT>(x2, CG * X2N, tag, gath, tid, a.status, 2)) sc[2] = 1.f;
    lds_barrier();
    if (sc[2] != 0.f) return;
    if (tid < A) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < CG; ++q) s += gath[q * X2N + tid];
      if (tid / UPW == g) p.q[rowS * A + tid] = s;
      xs[D2 + tid] = gath[(tid / UPW) * X2N + A + (tid % UPW)];          // h(s) for the next step's gates
    }
    // ---- (6) energies of the own positions: e[t] = sum_u v[u] tanh(keys[t][u] + q[u]).  Wave w takes units 32 w ..;
    //      lane = (unit c of a 16-unit tile, 4 positions g4): 16 tanh per lane, a DPP row reduction over the unit lanes
    {
      const int c = lane & 15, g4 = lane >> 4;
      float part[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ut = 0; ut < 2; ++ut) {
        const int u = wave * 32 + ut * 16 + c;
        const float vv = vs[u];
        float qv = 0.f;
#pragma unroll
        for (int q = 0; q < CG; ++q) qv += gath[q * X2N + u];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            part[rt][q] = fmaf(vv, tanhf_(keys_s[(rt * 16 + g4 * 4 + q) * (A + KPAD) + u] + qv), part[rt][q]);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float e = row16_sum(part[rt][q]);
          if (c == 0) ered[wave * TSMAX + rt * 16 + g4 * 4 + q] = e;
        }
    }
    lds_barrier();
    if (tid < UPW) {
      // this step's cell for the backward pass / the hoisted products: stored only now, behind the polls of this wave
      const int u = g * UPW + tid;
      p.ru[rowS * 2 * A + u] = sv_r;
      p.ru[rowS * 2 * A + A + u] = sv_u;
      p.cc[rowS * A + u] = sv_c;
      stf((T*)p.xc + rowS * XA + D2 + Dsp + u, sv_rh);
      stf((T*)p.hc + rowS * HC + u, sv_h);
      if (st + 1 < p.S) stf((T*)p.xa + (rowS + 1) * XA + D2 + Dsp + u, sv_h);
    }
    if (wave == 1) {
      float e = -INFINITY;
      if (lane < tn) {
        e = 0.f;
#pragma unroll
        for (int w = 0; w < A / 32; ++w) e += ered[w * TSMAX + lane];
      }
      const float m = wave_max(e);
      const float w = lane < tn ? __expf(e - m) : 0.f;
      const float l = wave_sum(w);
      if (lane < TSMAX) es[lane] = w;
      if (lane == 0) { sc[0] = m; sc[1] = l; }
    }
    lds_barrier();
    // ---- (7) partial next-prenet sums over the own positions, published with the softmax pieces (X3)
    {
      const int c = tid % D1, th = tid / D1;
      float s = 0.f;
      for (int tl = th; tl < tn; tl += CXG) s = fmaf(es[tl], pv_s[tl * D1 + c], s);
      red[tid] = s;
    }
    lds_barrier();
    if (tid < D1) put_granule(x3 + (size_t)g * X3N + tid, tag, red[tid] + red[D1 + tid]);
    else if (tid < D1 + 2) put_granule(x3 + (size_t)g * X3N + tid, tag, sc[tid - D1]);
    else if (tid < D1 + 2 + TSMAX) put_granule(x3 + (size_t)g * X3N + tid, tag, es[tid - D1 - 2]);
    // in the shadow of X3: the h rows of the NEXT step's gates (xs[D2 ..] = h(s) since (5))
    hpart = dot_regs<GKH>(wgh, xs + D2 + (tid / GC) * GKH);
    if (!gather_granules<(CG * X3N + CT - 1) / CT>(x3, CG * X3N, tag, gath, tid, a.status, 3)) sc[2] = 1.f;
    lds_barrier();
    if (sc[2] != 0.f) return;
    float mall = -INFINITY;
#pragma unroll
    for (int q = 0; q < CG; ++q) mall = fmaxf(mall, gath[q * X3N + D1]);
    float scl[CG], lsum = 0.f;
#pragma unroll
    for (int q = 0; q < CG; ++q) {
      const float lq = gath[q * X3N + D1 + 1];
      scl[q] = lq > 0.f ? __expf(gath[q * X3N + D1] - mall) : 0.f;
      lsum = fmaf(lq, scl[q], lsum);
    }
    const float inv = 1.f / lsum;
    for (int t = tid; t < p.Tia; t += CT) {
      float v = 0.f;
      if (t < L) {
        const int q = t / ts;
        v = gath[q * X3N + D1 + 2 + (t - q * ts)] * scl[q] * inv;
      }
      const bool mine = t < L ? (t / ts == g) : (g == CG - 1);
      if (mine) {
        p.align[rowS * p.Tia + t] = v;
        if (p.align_t) stf((T*)p.align_t + rowS * p.Tia + t, v);
      }
    }
    if (tid < D1) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < CG; ++q) s = fmaf(scl[q], gath[q * X3N + tid], s);
      const float v = fmaxf(fmaf(s, inv, f1n), 0.f);
      p1s[tid] = v;
      if (st + 1 < p.S && tid / (D1 / CG) == g) stf((T*)p.p1 + (rowS + 1) * D1 + tid, v);
    }
    lds_barrier();
  }
}

// ===================================================================================== backward
// Walks s = S-1 .. 0.  State carried between steps: dvec = dp1 of the step after ([D1], every workgroup), hrec = the
// gradient that reaches h(s) of the own units through step s+1.  Per step:
//   P2  dalign[t] of the own positions = da0[t] (hoisted) + pv[t] . dvec; the softmax-backward dot-product share -> E1
//   P3  (runs one step AHEAD: it needs only history) g1[t,u] = v[u] (1 - tanh^2(keys[t,u] + q[u])) in registers
//       de[t] = align[t] (dalign[t] - dot)
//   P4  dq partial of unit u = sum_t de[t] g1[t,u] -> E2 -> dq
//   P6  dh = dhc + dq . Wq^T + hrec;  dzc = dh (1-u)(1-c^2), dzu = dh (h_prev - c) u (1-u), direct = dh u
//   P7a partial input gradients of the candidate kernel over the own columns, all K rows -> E3a:
//       rows < 128: dp2 (first part); rows 128 + own units: d(r h_prev) -> dzr = . h_prev r (1-r), direct += . r
//   P7b the same for the gate kernel over the own [dzr | dzu] columns -> E3b: dp2 (second part), hrec = direct + h rows
//   P9  dp1 = (dp2 . W2^T) masked = the next dvec
template <typename T>
__global__ __launch_bounds__(CT) void taco1_attn_bwd_kernel(TArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const ns_taco1_attn_params& p = a.p;
  float* dvec = sm;                        // [D1]
  float* dp2s = dvec + D1;                 // [D2]
  float* dcs = dp2s + D2;                  // [UPW]  dzc of the own units
  float* dgs = dcs + UPW;                  // [GC]   dzr | dzu of the own units
  float* red = dgs + GC;                   // [NQ * K]
  float* dav = red + NQ * K;               // [TSMAX] dalign
  float* dev = dav + TSMAX;                // [TSMAX] energy gradients
  float* hrec = dev + TSMAX;               // [UPW]
  float* dird = hrec + UPW;                // [UPW]  direct part of the gradient wrt h_prev
  float* dq_s = dird + UPW;                // [A]
  float* sc = dq_s + A;                    // [16]  [0] dot, [2] abort
  float* gath = sc + 16;                   // [CG][K]
  float* vs = gath + CG * K;               // [A]
  float* keys_s = vs + A;                  // [TSMAX][A + KPAD]
  float* pv_s = keys_s + TSMAX * (A + KPAD);      // [TSMAX][D1]
  float* wq_s = pv_s + TSMAX * D1;         // [UPW][A]
  // per-step history images, twice (parity of the step): the next step's are filled in the middle of this one
  constexpr int HIMG = A + D1 + D2 + 2 * TSMAX + GC + 3 * UPW;
  float* him = wq_s + UPW * A;             // [2][HIMG]: qs | p1m | p2m | acur | da0s | rus | cs | hps | dhcs

  const int tid_ = threadIdx.x;
  const int n = blockIdx.x / CG, g = blockIdx.x % CG;
  const long S1 = p.S + 1;
  const int HC = A + p.E;
  const int Dsp = p.Dsp, XA = D2 + Dsp + A;                // the loop's K rows skip the speaker rows of both kernels (see the forward kernel)
  const int L = min(p.lengths ? p.lengths[n] : p.Ti, p.Ti);
  const int ts = max(1, (L + CG - 1) / CG);
  const int t0 = g * ts, tn = max(0, min(L, t0 + ts) - t0);
  u64* e1 = a.x1 + (size_t)n * CG;
  u64* e2 = a.x2 + (size_t)n * CG * A;
  u64* e3a = a.x3 + (size_t)n * CG * K;
  u64* e3b = a.x4 + (size_t)n * CG * K;

  // ---------------------------------------------------------------- resident weights
  const T* W2 = (const T*)p.w2;
  const T* Wg = (const T*)p.wg;
  const T* Wc = (const T*)p.wc;
  const T* Wq = (const T*)p.wq;
  const int tid = tid_;
  float wcr[PPT][CPQ_C], wgr[PPT][CPQ_G];  // W[row k][own columns of group qr] for this thread's (k, qr) pairs
#pragma unroll
  for (int jj = 0; jj < PPT; ++jj) {
    const int pi = tid + CT * jj, kl = pi % K, qr = pi / K;
    const int k = kl < D2 ? kl : kl + Dsp;               // kernel row of loop row kl
#pragma unroll
    for (int i = 0; i < CPQ_C; ++i) wcr[jj][i] = ldf(Wc + (long)k * A + g * UPW + qr * CPQ_C + i);
#pragma unroll
    for (int i = 0; i < CPQ_G; ++i) {
      const int c = qr * CPQ_G + i;                      // dgs order: dzr of the own units, then dzu
      wgr[jj][i] = ldf(Wg + (long)k * 2 * A + (c < UPW ? g * UPW + c : A + g * UPW + (c - UPW)));
    }
  }
  float w2r[W2K];                          // W2[c1][hf * W2K + i]
  {
    const T* b = W2 + (long)(tid % D1) * D2 + (tid / D1) * W2K;
#pragma unroll
    for (int i = 0; i < W2K; ++i) w2r[i] = ldf(b + i);
  }
  for (int i = tid; i < UPW * A; i += CT) wq_s[i] = ldf(Wq + (long)g * UPW * A + i);
  for (int i = tid; i < A; i += CT) vs[i] = p.v[i];
  {
    const float* kn = p.keys + ((long)n * p.Pi + p.padl_i + t0) * A;
    for (int i = tid; i < TSMAX * A; i += CT) keys_s[(i / A) * (A + KPAD) + i % A] = (i / A) < tn ? kn[i] : 0.f;
    const T* pvn = (const T*)p.pv + ((long)n * p.Pi + p.padl_i + t0) * D1;
    for (int i = tid; i < TSMAX * D1; i += CT) pv_s[i] = (i / D1) < tn ? ldf(pvn + i) : 0.f;
    for (int i = tid; i < D1; i += CT) dvec[i] = 0.f;
    for (int i = tid; i < 2 * HIMG; i += CT) him[i] = 0.f;
    if (tid < UPW) { hrec[tid] = 0.f; dird[tid] = 0.f; }
    if (tid < TSMAX) { dav[tid] = 0.f; dev[tid] = 0.f; }
    if (tid < 16) sc[tid] = 0.f;
  }
  // history of a step, one value per thread and role
  float h_q = 0.f, h_p1 = 0.f, h_p2 = 0.f, h_a = 0.f, h_da0 = 0.f, h_ru = 0.f, h_c = 0.f, h_hp = 0.f, h_dhc = 0.f;
  auto load_history = [&](int st_, int tid) {
    const long rowS = (long)n * S1 + st_ + 1;
    if (tid < A) h_q = p.q[rowS * A + tid];
    if (tid < D1) h_p1 = ldf((const T*)p.p1 + rowS * D1 + tid);
    if (tid < D2) h_p2 = ldf((const T*)p.xa + rowS * XA + tid);
    h_a = 0.f; h_da0 = 0.f;
    if (tid >= 256 && tid < 256 + TSMAX) {
      const int tl = tid - 256;
      if (tl < tn) { h_a = p.align[rowS * p.Tia + t0 + tl]; h_da0 = p.da0[rowS * p.Tia + t0 + tl]; }
    } else if (tid >= 320 && tid < 320 + GC) {
      const int c = tid - 320;
      h_ru = p.ru[rowS * 2 * A + (c < UPW ? g * UPW + c : A + g * UPW + (c - UPW))];
    } else if (tid >= 384 && tid < 384 + UPW) {
      const int u = g * UPW + tid - 384;
      h_c = p.cc[rowS * A + u];
      h_hp = ldf((const T*)p.xa + rowS * XA + D2 + Dsp + u);
      h_dhc = p.dhc[rowS * HC + u];
    }
  };
  auto store_history = [&](int par, int tid) {
    float* b = him + par * HIMG;
    if (tid < A) b[tid] = h_q;
    if (tid < D1) b[A + tid] = h_p1;
    if (tid < D2) b[A + D1 + tid] = h_p2;
    if (tid >= 256 && tid < 256 + TSMAX) { b[A + D1 + D2 + tid - 256] = h_a; b[A + D1 + D2 + TSMAX + tid - 256] = h_da0; }
    if (tid >= 320 && tid < 320 + GC) b[A + D1 + D2 + 2 * TSMAX + tid - 320] = h_ru;
    if (tid >= 384 && tid < 384 + UPW) {
      float* o = b + A + D1 + D2 + 2 * TSMAX + GC + (tid - 384);
      o[0] = h_c; o[UPW] = h_hp; o[2 * UPW] = h_dhc;
    }
  };
  __syncthreads();                           // the zero fills above are done before the owners' values go in
  load_history(p.S - 1, tid_);
  store_history((p.S - 1) & 1, tid_);
  __syncthreads();
  // energy pass of a step: needs only that step's history (query), runs one step ahead; g1 stays in registers
  float g1v[16];                             // [unit tile ut][position tile rt][q]: unit 32 wave + 16 ut + c, position 16 rt + 4 g4 + q
  auto energy_pass = [&](const float* qs, int lane, int wave) {
    const int c = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int ut = 0; ut < 2; ++ut) {
      const int u = wave * 32 + ut * 16 + c;
      const float qv = qs[u], vv = vs[u];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int pos = rt * 16 + g4 * 4 + q;
          const float th = tanhf_(keys_s[pos * (A + KPAD) + u] + qv);
          g1v[ut * 8 + rt * 4 + q] = pos < tn ? vv * (1.f - th * th) : 0.f;
        }
    }
  };
  energy_pass(him + ((p.S - 1) & 1) * HIMG, tid_ & 63, tid_ >> 6);

  for (int st = p.S - 1; st >= 0; --st) {
    const long rowS = (long)n * S1 + st + 1;
    const unsigned tag = (unsigned)(p.S - st);
    int tid = tid_;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* hb = him + (st & 1) * HIMG;
    const float* p1m = hb + A; const float* p2m = hb + A + D1; const float* acur = hb + A + D1 + D2;
    const float* da0s = acur + TSMAX; const float* rusb = da0s + TSMAX; const float* csb = rusb + GC;
    const float* hpsb = csb + UPW; const float* dhcsb = hpsb + UPW;

    if (st > 0) load_history(st - 1, tid);   // lands by the middle of this step
    lds_barrier();
    // ---- P2: dalign of the own positions: thread = (position tid / 16, 16 columns each)
    {
      const int tl = tid >> 4, cq = tid & 15;
      float s = 0.f;
      if (tl < tn) {
        const float* pr = pv_s + tl * D1 + cq * 4;
        const float* dv = dvec + cq * 4;
#pragma unroll
        for (int i = 0; i < D1 / 4; i += 16) {
          const float4 x = *(const float4*)(pr + 4 * i), y = *(const float4*)(dv + 4 * i);
          s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
        }
      }
      s = row16_sum(s);
      if (cq == 0 && tl < TSMAX) dav[tl] = tl < tn ? da0s[tl] + s : 0.f;
    }
    lds_barrier();
    if (wave == 7) {
      const float d = wave_sum(lane < tn ? acur[lane] * dav[lane] : 0.f);
      if (lane == 0) put_granule(e1 + g, tag, d);
      if (!gather_granules<1>(e1, CG, tag, gath, lane < CG ? lane : CG, a.status, 5)) sc[2] = 1.f;
      float dd = lane < CG ? gath[lane] : 0.f;
      dd = wave_sum(dd);
      if (lane == 0) sc[0] = dd;
    }
    lds_barrier();
    if (tid < TSMAX) {
      float de = 0.f;
      if (tid < tn) {
        de = acur[tid] * (dav[tid] - sc[0]);
        p.de[rowS * p.Tia + t0 + tid] = de;
      }
      dev[tid] = de;
    }
    lds_barrier();
    // ---- P4: dq partials -> E2
    {
      const int c = lane & 15, g4 = lane >> 4;
#pragma unroll
      for (int ut = 0; ut < 2; ++ut) {
        float s = 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int q = 0; q < 4; ++q) s = fmaf(dev[rt * 16 + g4 * 4 + q], g1v[ut * 8 + rt * 4 + q], s);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (g4 == 0) put_granule(e2 + (size_t)g * A + wave * 32 + ut * 16 + c, tag, s);
      }
    }
    if (!gather_granules<(CG * A + CT - 1) / CT>(e2, CG * A, tag, gath, tid, a.status, 3)) sc[2] = 1.f;
    lds_barrier();
    if (sc[2] != 0.f) return;
    if (st > 0) store_history((st & 1) ^ 1, tid);
    if (tid < A) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < CG; ++q) s += gath[q * A + tid];
      dq_s[tid] = s;
      if (tid / UPW == g) stf((T*)p.dq + rowS * A + tid, s);
    }
    lds_barrier();
    // ---- P6: dh of the own units through W_query, the cell's first half: thread = (unit tid / 16, 16 columns each)
    {
      const int j = tid >> 4, uq = tid & 15;
      float s = 0.f;
      if (j < UPW) {
        const float* wr = wq_s + j * A + uq * 4;
        const float* dq = dq_s + uq * 4;
#pragma unroll
        for (int i = 0; i < A / 16; i += 4) {
          const float4 x = *(const float4*)(wr + 16 * i), y = *(const float4*)(dq + 16 * i);
          s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
        }
      }
      s = row16_sum(s);
      if (uq == 0 && j < UPW) {
        const int u = g * UPW + j;
        const float dh = dhcsb[j] + s + hrec[j];
        const float uu = rusb[UPW + j], c = csb[j], hp = hpsb[j];
        const float dzc = dh * (1.f - uu) * (1.f - c * c);
        const float dzu = dh * (hp - c) * uu * (1.f - uu);
        dcs[j] = dzc;
        dgs[UPW + j] = dzu;
        dird[j] = dh * uu;
        stf((T*)p.dzc + rowS * A + u, dzc);
        stf((T*)p.dzg + rowS * 2 * A + A + u, dzu);
      }
    }
    lds_barrier();
    // ---- P7a: candidate kernel, partial input gradients over the own columns -> E3a
#pragma unroll
    for (int jj = 0; jj < PPT; ++jj) {
      const int pi = tid + CT * jj, qr = pi / K;
      red[pi] = dot_regs<CPQ_C>(wcr[jj], dcs + qr * CPQ_C);
    }
    lds_barrier();
    if (tid < K) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) s += red[q * K + tid];
      put_granule(e3a + (size_t)g * K + tid, tag, s);
    }
    if (!gather_granules<(CG * K + CT - 1) / CT>(e3a, CG * K, tag, gath, tid, a.status, 4)) sc[2] = 1.f;
    lds_barrier();
    if (sc[2] != 0.f) return;
    if (tid < D2) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < CG; ++q) s += gath[q * K + tid];
      dp2s[tid] = s;                         // first part; the mask comes with the second
    } else if (tid < D2 + UPW) {
      const int j = tid - D2, u = g * UPW + j;
      float drh = 0.f;
#pragma unroll
      for (int q = 0; q < CG; ++q) drh += gath[q * K + D2 + u];
      const float r = rusb[j], hp = hpsb[j];
      const float dzr = drh * hp * r * (1.f - r);
      dgs[j] = dzr;
      dird[j] += drh * r;
      stf((T*)p.dzg + rowS * 2 * A + u, dzr);
    }
    lds_barrier();
    // ---- P7b: gate kernel -> E3b
#pragma unroll
    for (int jj = 0; jj < PPT; ++jj) {
      const int pi = tid + CT * jj, qr = pi / K;
      red[pi] = dot_regs<CPQ_G>(wgr[jj], dgs + qr * CPQ_G);
    }
    lds_barrier();
    if (tid < K) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) s += red[q * K + tid];
      put_granule(e3b + (size_t)g * K + tid, tag, s);
    }
    // in the shadow of E3b: the energy pass of step s-1 (its history image was stored behind E2)
    if (st > 0) energy_pass(him + ((st & 1) ^ 1) * HIMG, lane, wave);
    if (!gather_granules<(CG * K + CT - 1) / CT>(e3b, CG * K, tag, gath, tid, a.status, 6)) sc[2] = 1.f;
    lds_barrier();
    if (sc[2] != 0.f) return;
    if (tid < D2) {
      float s = dp2s[tid];
#pragma unroll
      for (int q = 0; q < CG; ++q) s += gath[q * K + tid];
      s = p2m[tid] > 0.f ? s : 0.f;
      dp2s[tid] = s;
      if (tid / (D2 / CG) == g) stf((T*)p.dp2 + rowS * D2 + tid, s);
    } else if (tid < D2 + UPW) {
      const int j = tid - D2, u = g * UPW + j;
      float s = dird[j];
#pragma unroll
      for (int q = 0; q < CG; ++q) s += gath[q * K + D2 + u];
      hrec[j] = s;
    }
    lds_barrier();
    // ---- P9: dp1 = (dp2 . W2^T) masked: the next dvec (every workgroup computes all of it)
    red[tid] = dot_regs<W2K>(w2r, dp2s + (tid / D1) * W2K);
    lds_barrier();
    if (tid < D1) {
      float s = red[tid] + red[D1 + tid];
      s = p1m[tid] > 0.f ? s : 0.f;
      dvec[tid] = s;
      if (tid / (D1 / CG) == g) stf((T*)p.df1 + rowS * D1 + tid, s);
    }
    lds_barrier();
  }
}

constexpr size_t FWD_LDS = sizeof(float) * (K + D1 + CT + GC + UPW + TSMAX + 8 * TSMAX + 16 + CG * XMAX + A + TSMAX * (A + KPAD) +
                                             TSMAX * D1 + UPW * A);
constexpr size_t BWD_LDS = sizeof(float) * (D1 + D2 + UPW + GC + NQ * K + 2 * TSMAX + 2 * UPW + A + 16 + CG * K + A +
                                             TSMAX * (A + KPAD) + TSMAX * D1 + UPW * A +
                                             2 * (A + D1 + D2 + 2 * TSMAX + GC + 3 * UPW));
static_assert(FWD_LDS <= 160 * 1024 && BWD_LDS <= 160 * 1024, "LDS");

size_t fwd_granules(int N) { return (size_t)N * CG * (X1N + X2N + X3N); }
size_t bwd_granules(int N) { return (size_t)N * CG * (1 + A + 2 * K); }
}  // namespace

extern "C" int ns_taco1_attn_cluster_supported(const ns_taco1_attn_params* p) {
  if (!p) return 0;
  if (!(p->A == A && p->D1 == D1 && p->D2 == D2 && p->E > 0 && p->Ti >= 1 && p->Ti <= 256 && p->Tia >= p->Ti && p->S >= 1 && p->N >= 1 && p->Dsp >= 0)) return 0;
  if (!(p->dtype == NS_F32 || p->dtype == NS_BF16)) return 0;
  if (!p->keys || !p->pv || !p->f1 || !p->w2 || !p->wg || !p->wc || !p->wq || !p->b2 || !p->bg || !p->bc || !p->v) return 0;
  if (!p->p1 || !p->xa || !p->xc || !p->hc || !p->ru || !p->cc || !p->q || !p->align) return 0;
  if ((long)p->N * (p->S + 1) * (long)(p->A + p->E) >= (1L << 31)) return 0;
  return p->N * CG <= ns_device_cus();      // every workgroup of the launch must be resident at once, one per CU
}

extern "C" size_t ns_taco1_attn_cluster_work_bytes(const ns_taco1_attn_params* p) {
  if (!p) return 0;
  const size_t f = fwd_granules(p->N), b = bwd_granules(p->N);
  return 256 + (f > b ? f : b) * sizeof(u64);
}

static int taco1_run(const ns_taco1_attn_params* p, void* work, hipStream_t s, int backward) {
  const char* name = backward ? "ns_taco1_attn_cluster_bwd" : "ns_taco1_attn_cluster_fwd";
  NS_CHECK_ARG(p && work, "%s: null", name);
  NS_CHECK_ARG(ns_taco1_attn_cluster_supported(p), "%s: needs A = 256, D1 = 256, D2 = 128, T_in <= 256, N * 8 <= the device's CUs", name);
  if (backward) NS_CHECK_ARG(p->dhc && p->da0 && p->df1 && p->dp2 && p->dzg && p->dzc && p->dq && p->de, "%s: null backward operand", name);
  TArgs a = {};
  a.p = *p;
  a.status = (int*)work;
  u64* x = (u64*)((char*)work + 256);
  const size_t N = (size_t)p->N;
  size_t gran;
  if (!backward) {
    a.x1 = x; a.x2 = a.x1 + N * CG * X1N; a.x3 = a.x2 + N * CG * X2N; a.x4 = nullptr;
    gran = fwd_granules(p->N);
  } else {
    a.x1 = x; a.x2 = a.x1 + N * CG; a.x3 = a.x2 + N * CG * A; a.x4 = a.x3 + N * CG * K;
    gran = bwd_granules(p->N);
  }
  { const int zrc = ns_zero_async(work, (256 + gran * sizeof(u64) + 15) & ~(size_t)15, s); if (zrc) return zrc; }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)taco1_attn_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)taco1_attn_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)taco1_attn_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)taco1_attn_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const dim3 grid((unsigned)(p->N * CG)), block(CT);
  if (!backward) {
    if (p->dtype == NS_BF16) hipLaunchKernelGGL(taco1_attn_fwd_kernel<bf16_t>, grid, block, FWD_LDS, s, a);
    else hipLaunchKernelGGL(taco1_attn_fwd_kernel<float>, grid, block, FWD_LDS, s, a);
  } else {
    if (p->dtype == NS_BF16) hipLaunchKernelGGL(taco1_attn_bwd_kernel<bf16_t>, grid, block, BWD_LDS, s, a);
    else hipLaunchKernelGGL(taco1_attn_bwd_kernel<float>, grid, block, BWD_LDS, s, a);
  }
  NS_CHECK_LAUNCH(name);
  return NS_OK;
}

extern "C" int ns_taco1_attn_cluster_fwd(const ns_taco1_attn_params* p, void* work, ns_stream_t s) {
  return taco1_run(p, work, (hipStream_t)s, 0);
}
extern "C" int ns_taco1_attn_cluster_bwd(const ns_taco1_attn_params* p, void* work, ns_stream_t s) {
  return taco1_run(p, work, (hipStream_t)s, 1);
}
