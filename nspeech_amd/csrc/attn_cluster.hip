// Tacotron-2 attention RNN (prenet-2 -> attention LSTM -> query -> location-sensitive energies -> softmax ->
// next step's prenet layer) as ONE persistent launch per direction instead of ~12 dependent launches per step pair
// (rnn_wrappers.py:25-31, modules.py:83-102, attention.py:30-60 under the teacher-forced decoder of tacotron2.py:63-83).
//
// Every operation of this chain is independent per utterance and its weights are stationary (0.98 M values), so an
// utterance gets a CLUSTER of G = 8 workgroups (one per CU, 512 threads) that keeps everything it needs on chip for
// the whole sequence - the weights in REGISTERS as fp32 (the products are matrix-VECTOR at one utterance per
// cluster, so they run as exact fp32 FMAs from registers; an MFMA would spend 15/16 of its rows on nothing), the
// utterance's keys and projected memory in LDS:
//   workgroup g owns  * LSTM units [g*A/8, (g+1)*A/8): the 4 gate columns of each, all K = D2 + A input rows
//                     * the rows of W_query that belong to those units (all A query columns)
//                     * all of W_prenet2 (every workgroup computes the whole p2: cheaper than a third exchange)
//                     * memory positions [g*ts, (g+1)*ts), ts = ceil(length / 8): keys and memory.W1c rows
// and a step needs only TWO exchanges inside the cluster (8-byte {step tag, fp32} granules, written and polled with
// relaxed agent-scope atomics = sc1; the data is its own flag, cdna guide G16 / R2):
//   X2: partial queries h_slice . Wq[slice, :]  (A values) + the new h slice          -> every workgroup sums q, has h
//   X3: partial next-prenet sums  sum_t w[t] pv[t, :]  (D1 values) + the local softmax max / sum + the local
//       unnormalised weights w[t]                          -> every workgroup forms p1[s+1] and the whole alignment
// (the softmax is combined flash-attention style: local max m_g, w = exp(e - m_g), global rescale exp(m_g - m) / L).
// One buffer per exchange suffices: X2 of step s+1 is published only after X3 of step s has been gathered from every
// peer, and a peer publishes that only after it has gathered X2 of step s - nobody can still be reading it.
// Every spin is bounded; on timeout the kernel raises *status and every workgroup of the cluster leaves.
// All history the backward pass and the hoisted products read (p1, xa, hc, ca, ga, q, align, align_t) is written in
// the layouts of ns_taco2_attn_fwd, which stays as the fallback for shapes this kernel does not cover.
#include "common.h"
#include <stdlib.h>

int ns_attn_contexts_after_loop(const ns_taco2_attn_params& p, hipStream_t s);     // attn.hip
template <typename T> int ns_attn_bwd_post(const ns_taco2_attn_params& p, hipStream_t s);

typedef unsigned long long u64;
namespace {
constexpr int CG = 8;              // workgroups per utterance
constexpr int CT = 512;            // threads per workgroup
constexpr int TSMAX = 32;          // memory positions per workgroup (T_in <= 256)
constexpr int KWMAX = 8;
constexpr int APAD = 8;            // zero margin around the alignment vector in LDS
constexpr int KPAD = 4;            // row pad of the keys image: the energy pass reads keys[r][u] with r across 16 lanes and u
                                   // across 4 - a row stride of A floats puts all 16 rows on one bank (16-way conflict)

template <int A_, int D1_, int D2_>
struct Cfg {
  static constexpr int A = A_, D1 = D1_, D2 = D2_;
  static constexpr int UPW = A / CG;           // LSTM units per workgroup
  static constexpr int GC = 4 * UPW;           // gate columns per workgroup
  static constexpr int K = D2 + A;             // in-loop input rows of the attention LSTM (speaker rows fold into the bias)
  static constexpr int GG = CT / GC, GK = K / GG;        // gates: k groups, k per thread
  static constexpr int P2G = CT / D2, P2K = D1 / P2G;    // prenet-2
  static constexpr int QG = CT / A, QK = UPW / QG;       // query partials / energy t-groups
  static constexpr int CXG = CT / D1;                    // context t-groups
  static constexpr int X2N = A + UPW;                    // granules per workgroup, exchange 2
  static constexpr int X3N = D1 + 2 + TSMAX;             // exchange 3
  static constexpr int XMAX = X2N > X3N ? X2N : X3N;
  static_assert(A % CG == 0 && CT % GC == 0 && K % GG == 0 && CT % D2 == 0 && D1 % P2G == 0, "shape");
  static_assert(CT % A == 0 && UPW % QG == 0 && CT % D1 == 0, "shape");
};

struct ACArgs {
  ns_taco2_attn_params p;
  u64* x2; u64* x3;          // [N][CG][X2N], [N][CG][X3N]
  u64* x1;                   // backward: [N][CG] dot-product shares
  int* status;
  long long* trace;          // NS_ATTN_TRACE=1: [step][16] timestamps (100 MHz) of workgroup 0, else null
  // free-running decode (taco2_decode_kernel): the next step's frame term f1 arrives as granules from the decoder-LSTM
  // workgroups ([N][D1], tag = consuming step + 1) and the alignment goes out to them ([N][Tia], tag = step + 1)
  u64* f1x; u64* alx;
};
constexpr size_t TRACE_BYTES = 256 * 16 * sizeof(long long);
__device__ __forceinline__ void stamp(const ACArgs& a, int st, int k) {
  if (a.trace && blockIdx.x == 0 && threadIdx.x == 0 && st < 256) a.trace[st * 16 + k] = wall_clock64();
}

// Workgroup barrier that orders LDS only.  __syncthreads() also drains the vector-memory queue (s_waitcnt vmcnt(0)):
// inside the step loops that would wait, at every barrier, for whatever is in flight - the history prefetch from HBM,
// the write-through acknowledgement of a publish (1.3 us).  The waves of a workgroup talk through LDS only; what goes
// through global memory is either tag-polled (the exchanges) or read by later kernels.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void put_granule(u64* g_, unsigned tag, float v) {
  NS_GLOBAL u64* g = (NS_GLOBAL u64*)g_;
  __hip_atomic_store(g, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every thread waits for its own granules (PER per thread, stride CT) and drops the values into LDS.
// Returns false when this thread gave up (a peer raised the status word, or the wall-clock bound passed): the caller
// raises the workgroup's LDS abort word then - no thread reads the status word from memory on the step's critical path
// (round 2 had thread 0 load it behind every gather: one more memory round trip in front of the barrier, 2 per step).
template <int PER>
__device__ __forceinline__ bool gather_granules(const u64* src_, int total, unsigned tag, float* dst, int tid,
                                                int* status, int code) {
  const NS_GLOBAL u64* src = (const NS_GLOBAL u64*)src_;      // global_load, not flat_load: the poll leaves lgkmcnt alone
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
    if (!ok) {
      if ((++spins & 1023u) == 0) {        // every 1024 polls: has a peer given up, or is the wall-clock bound passed
        if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = gave_up = true;
        else if (ns_spin_timed_out(clk0)) { atomicExch(status, code); ok = gave_up = true; }
      }
    }
  } while (!ok);
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid + j * CT;
    if (i < total) dst[i] = __uint_as_float((unsigned)v[j]);
  }
  return !gave_up;
}

template <typename T> __device__ __forceinline__ float ldw(const T* p, long i) { return ldf(p + i); }

// sum_i w[i] * x[i] with the weights in registers and x in LDS (wave-uniform addresses: broadcast reads).  The chunks
// are fenced so that the compiler keeps at most two chunks of x in flight instead of hoisting every LDS read of the
// product to its top (the weights already take most of the register file).
template <int NK>
__device__ __forceinline__ float dot_regs(const float (&w)[NK], const float* x) {
  static_assert(NK % 4 == 0 || NK < 4, "chunking");
  float s0 = 0.f, s1 = 0.f;
  if constexpr (NK < 4) {
#pragma unroll
    for (int i = 0; i < NK; ++i) s0 = fmaf(w[i], x[i], s0);
    return s0;
  } else {
    float4 cur = *(const float4*)x;
#pragma unroll
    for (int i = 0; i < NK; i += 4) {
      float4 nxt = cur;
      if (i + 4 < NK) nxt = *(const float4*)(x + i + 4);
      s0 = fmaf(w[i], cur.x, s0); s1 = fmaf(w[i + 1], cur.y, s1);
      s0 = fmaf(w[i + 2], cur.z, s0); s1 = fmaf(w[i + 3], cur.w, s1);
      // the sums pass through the fence: this chunk's FMAs cannot sink below it, later reads cannot rise above it
      asm volatile("" : "+v"(s0), "+v"(s1) :: "memory");
      cur = nxt;
    }
    return s0 + s1;
  }
}

// ===================================================================================== forward
// One granule, polled by its owner thread (the decode kernel's frame feedback).  Returns false on time-out / raised status.
__device__ __forceinline__ bool poll_granule(const u64* g, unsigned tag, float& out, int* status, int code) {
  unsigned spins = 0, clk0 = 0;
  for (;;) {
    const u64 v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(v >> 32) == tag) { out = __uint_as_float((unsigned)v); return true; }
    if ((++spins & 1023u) == 0) {
      if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
      if (ns_spin_timed_out(clk0)) { atomicExch(status, code); return false; }
    }
  }
}

// The forward recurrence of one workgroup (bid = utterance * CG + member).  INFER = free-running decode: the frame term
// of the next step is not hoisted (it depends on this step's output through the two decoder LSTMs) but polled from
// a.f1x, and the alignment is published to a.alx for the workgroups that run the first decoder LSTM.
template <typename T, typename C, bool INFER>
__device__ __forceinline__ void attn_fwd_body(const ACArgs& a, float* sm, const int bid) {
  constexpr int A = C::A, D1 = C::D1, D2 = C::D2, UPW = C::UPW, GC = C::GC, K = C::K;
  constexpr int GG = C::GG, GK = C::GK, P2G = C::P2G, P2K = C::P2K, QG = C::QG, QK = C::QK, CXG = C::CXG;
  constexpr int X2N = C::X2N, X3N = C::X3N;
  const ns_taco2_attn_params& p = a.p;
  float* xs = sm;                          // [K]      LSTM input: p2 | h(s-1)
  float* p1s = xs + K;                     // [D1]     prenet layer 1 of the current step
  float* red = p1s + D1;                   // [CT]     partial sums across the k / t groups of a product
  float* qs = red + CT;                    // [A]      query
  float* al = qs + A;                      // [APAD + 256 + APAD] alignment of the previous step, zero margins
  float* es = al + 256 + 2 * APAD;         // [TSMAX]  local unnormalised softmax weights
  float* ered = es + TSMAX;                // [CT/64][TSMAX] per-wave energy sums
  float* hloc = ered + (CT / 64) * TSMAX;  // [UPW] new h of this workgroup's units
  float* sc = hloc + UPW;                  // [16]     [0] local max, [1] local sum, [2] abort flag
  float* gath = sc + 16;                   // [CG][XMAX]
  float* keys_s = gath + CG * C::XMAX;     // [TSMAX][A + KPAD]
  float* pv_s = keys_s + TSMAX * (A + KPAD);        // [TSMAX][D1]
  float* wq_s = pv_s + TSMAX * D1;         // [UPW][A]  W_query rows of the own units
  float* cst_s = wq_s + UPW * A;           // [KWMAX + 1][A]  folded location filter, attention_v

  const int tid_ = threadIdx.x;
  const int n = bid / CG, g = bid % CG;
  const long S1 = p.S + 1;
  const int Dsp = p.Dsp, XA = D2 + Dsp + A, HC = A + p.E;
  const int L = min(p.lengths ? p.lengths[n] : p.Ti, p.Ti);
  const int ts = (L + CG - 1) / CG;                       // positions per workgroup
  const int t0 = g * ts, tn = max(0, min(L, t0 + ts) - t0);      // own slice [t0, t0 + tn)
  const int half = (p.kw - 1) / 2;
  u64* x2 = a.x2 + (size_t)n * CG * X2N;
  u64* x3 = a.x3 + (size_t)n * CG * X3N;
  if (tid_ == 0) { sc[2] = 0.f; sc[3] = 0.f; }            // [2] set by a thread whose gather gave up, [3] by the frame-feedback poll
                                                          // (both in front of the __syncthreads() below)

  // ---------------------------------------------------------------- resident weights (registers)
  const T* W2 = (const T*)p.w2;            // [D1][D2]
  const T* Watt = (const T*)p.watt;        // [D2 + Dsp + A][4A]
  const T* Wq = (const T*)p.wq;            // [A][A]
  const int tid = tid_;
  const int p2c = tid % D2, p2q0 = tid / D2;
  float w2r[P2K];
  {
    const T* b = W2 + (long)(p2q0 * P2K) * D2 + p2c;          // one base pointer, constant strides
#pragma unroll
    for (int i = 0; i < P2K; ++i) w2r[i] = ldf(b + i * D2);
  }
  const int gc = tid % GC, gq0 = tid / GC;
  const int gcol = (gc / UPW) * A + g * UPW + (gc % UPW);          // column of the [.., 4A] kernel
  // every k group takes a share of the prenet rows AND a share of the h rows: the h half of next step's gates is formed in
  // the shadow of exchange 3 (h(s) is final since exchange 2), so only D2 / GG FMAs per thread sit on the step's chain
  constexpr int GKP = D2 / GG, GKH = A / GG;
  static_assert(D2 % GG == 0 && A % GG == 0 && GKP + GKH == GK, "gate row split");
  float wgp[GKP], wgh[GKH];
  {
    // rows behind the prenet part skip the speaker rows (their product is folded into the bias)
    const T* bp = Watt + (long)(gq0 * GKP) * 4 * A + gcol;
    const T* bh = Watt + (long)(D2 + Dsp + gq0 * GKH) * 4 * A + gcol;
#pragma unroll
    for (int i = 0; i < GKP; ++i) wgp[i] = ldf(bp + i * 4 * A);
#pragma unroll
    for (int i = 0; i < GKH; ++i) wgh[i] = ldf(bh + i * 4 * A);
  }
  float hpart = 0.f;                         // h(s-1) . W[h rows of this group]: h(-1) = 0
  // W_query rows of this workgroup's units and the per-unit constants of the energy pass live in LDS (the register
  // file is taken by the two big weight slices): wq_s[UPW][A], cst_s[KWMAX + 1][A] = folded location filter | v
  for (int i = tid; i < UPW * A; i += CT) wq_s[i] = ldf(Wq + (long)g * UPW * A + i);
  for (int i = tid; i < (KWMAX + 1) * A; i += CT) {
    const int k = i / A, u = i % A;
    cst_s[k * (A + KPAD) + u] = k < p.kw ? p.wcl[k * A + u] : (k == KWMAX ? p.v[u] : 0.f);
  }
  // cell owner threads: tid < UPW
  float cstate = 0.f;
  float gbias[4] = {0.f, 0.f, 0.f, 0.f};
  if (tid < UPW) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = j * A + g * UPW + tid;
      float b = p.batt[col];
      // the speaker projection is the same in every step: its product joins the bias
      const T* spk = (const T*)p.xa + ((long)n * S1 + 1) * XA + D2;
      for (int k = 0; k < Dsp; ++k) b = fmaf(ldf(spk + k), ldw(Watt, (long)(D2 + k) * 4 * A + col), b);
      gbias[j] = b;
    }
  }
  const float b2c = tid < D2 ? p.b2[tid] : 0.f;

  // ---------------------------------------------------------------- per-utterance LDS images
  {
    const float* kn = p.keys + ((long)n * p.Pi + p.padl_i + t0) * A;
    for (int i = tid; i < TSMAX * A; i += CT) keys_s[(i / A) * (A + KPAD) + i % A] = (i / A) < tn ? kn[i] : 0.f;
    const T* pvn = (const T*)p.pv + ((long)n * p.Pi + p.padl_i + t0) * D1;
    for (int i = tid; i < TSMAX * D1; i += CT) pv_s[i] = (i / D1) < tn ? ldf(pvn + i) : 0.f;
    for (int i = tid; i < 256 + 2 * APAD; i += CT) al[i] = 0.f;
    for (int i = tid; i < K; i += CT) xs[i] = 0.f;
    if (tid < D1) p1s[tid] = fmaxf(p.f1[((long)n * S1 + 1) * D1 + tid], 0.f);     // context before the first step is 0
  }
  // slot 1 of p1 (history): the share of this workgroup
  if (tid < D1 && tid / (D1 / CG) == g) stf((T*)p.p1 + ((long)n * S1 + 1) * D1 + tid, fmaxf(p.f1[((long)n * S1 + 1) * D1 + tid], 0.f));
  __syncthreads();

  for (int st = 0; st < p.S; ++st) {
    const long slot = st + 1;
    const unsigned tag = (unsigned)(st + 1);
    // The thread index is made opaque once per step: every address below is then recomputed inside the iteration
    // (a few VALU instructions) instead of being hoisted out of the loop as ~40 loop-invariant 64-bit pointers,
    // which the register file has no room for beside the resident weights.
    int tid = tid_;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p2q = tid / D2, gq = tid / GC, qu = tid % A, qq = tid / A;
    // next step's hoisted frame term, needed at the very end of this step
    float f1n = 0.f;
    if (!INFER && tid < D1 && st + 1 < p.S) f1n = p.f1[((long)n * S1 + slot + 1) * D1 + tid];

    stamp(a, st, 0);
    // ---- (1) p2 = relu(p1 . W2 + b2): every workgroup computes all of it
    red[tid] = dot_regs<P2K>(w2r, p1s + p2q * P2K);
    lds_barrier();
    if (tid < D2) {
      float s = b2c;
#pragma unroll
      for (int q = 0; q < P2G; ++q) s += red[q * D2 + tid];
      s = fmaxf(s, 0.f);
      xs[tid] = s;
      if (tid / (D2 / CG) == g) stf((T*)p.xa + ((long)n * S1 + slot) * XA + tid, s);
    }
    lds_barrier();
    stamp(a, st, 1);
    // ---- (2) gates of this workgroup's units, cell update
    float sv[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    red[tid] = hpart + dot_regs<GKP>(wgp, xs + gq * GKP);
    lds_barrier();
    if (tid < UPW) {
      float z[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = gbias[j];
#pragma unroll
        for (int q = 0; q < GG; ++q) s += red[q * GC + j * UPW + tid];
        z[j] = s;
      }
      const float gi = sigmoidf_(z[0]), gj = tanhf_(z[1]), gf = sigmoidf_(z[2] + 1.0f), go = sigmoidf_(z[3]);
      cstate = ns_cell_clip(gf * cstate + gi * gj, p.cell_clip);
      const float h = go * tanhf_(cstate);
      hloc[tid] = h;
      put_granule(x2 + (size_t)g * X2N + A + tid, tag, h);                 // the peers wait for this
      sv[0] = gi; sv[1] = gj; sv[2] = gf; sv[3] = go; sv[4] = h;           // saved after the gather (see below)
    }
    lds_barrier();
    stamp(a, st, 2);
    // ---- (3) partial query of this workgroup's h rows, published
    {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < QK; ++i) s = fmaf(wq_s[(qq * QK + i) * A + qu], hloc[qq * QK + i], s);
      red[tid] = s;
    }
    lds_barrier();
    if (tid < A) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < QG; ++q) s += red[q * A + tid];
      put_granule(x2 + (size_t)g * X2N + tid, tag, s);
    }
    stamp(a, st, 3);
    // ---- (5a) in the shadow of exchange 2: the part of the energies that does not need the query.  The location term
    //      loc[t][u] = sum_k align[t + k - half] w[k][u] is a matrix product with K = 8 taps: two k-steps of
    //      v_mfma_f32_16x16x4_f32 (exact fp32) per 16 positions x 16 units, accumulated onto C = keys[t][u].  A = the
    //      alignment window (row = position), B = the folded filter, D: lane = unit, registers = 4 positions.
    f32x4 eacc[2][2];                        // [unit tile][position tile]
    if (wave < A / 32) {
      const int c = lane & 15, g4 = lane >> 4;
#pragma unroll
      for (int ut = 0; ut < 2; ++ut) {
        const int u = wave * 32 + ut * 16 + c;
        const float w0 = cst_s[g4 * (A + KPAD) + u], w1 = cst_s[(4 + g4) * (A + KPAD) + u];       // B[k = g4 (+ 4)][col = u]
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          f32x4 acc;
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = keys_s[(rt * 16 + g4 * 4 + q) * (A + KPAD) + u];
          const float* aw = al + APAD + t0 + rt * 16 + c - half + g4;                               // A[row = c][k = g4 (+ 4)]
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[0], w0, acc, 0, 0, 0);
          eacc[ut][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[4], w1, acc, 0, 0, 0);
        }
      }
    }
    // ---- (4) gather X2: q = sum of the partials (fixed order), h of every unit
    if (!gather_granules<(CG * X2N + CT - 1) / CT>(x2, CG * X2N, tag, gath, tid, a.status, 1)) sc[2] = 1.f;        // the abort word: zero since the kernel's start
    lds_barrier();
    if (sc[2] != 0.f) return;                 // uniform: every thread reads the same LDS word
    stamp(a, st, 4);
    if (tid < A) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < CG; ++q) s += gath[q * X2N + tid];
      if (tid / UPW == g) p.q[((long)n * S1 + slot) * A + tid] = s;       // (the energy lanes below form the same sum themselves)
      xs[D2 + tid] = gath[(tid / UPW) * X2N + A + (tid % UPW)];          // h(s) for the next step's gates
    }
    // no barrier here: the energy lanes read the gathered partial queries (complete since the barrier above) and add them
    // in the same order; xs[D2 ..] is next read behind the barriers that follow the energy pass
    stamp(a, st, 5);
    // ---- (5b) energies of the own positions: x = (keys + location term, formed in the shadow of exchange 2) + q; tanh,
    //      the product with attention_v and the sum over the wave's 32 units (a DPP row reduction over the 16 unit
    //      lanes) in the D layout.  Round 2 formed x with 128 FMAs per lane out of 112 scalar LDS reads behind the
    //      exchange: 2.0 us per step for this block; with the location term on the matrix core 1.3; see the trace.
    if (wave < A / 32) {
      const int c = lane & 15, g4 = lane >> 4;
      float part[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ut = 0; ut < 2; ++ut) {
        const int u = wave * 32 + ut * 16 + c;
        const float vv = cst_s[KWMAX * (A + KPAD) + u];
        float qv = 0.f;
#pragma unroll
        for (int q = 0; q < CG; ++q) qv += gath[q * X2N + u];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int q = 0; q < 4; ++q) part[rt][q] = fmaf(vv, tanhf_(eacc[ut][rt][q] + qv), part[rt][q]);
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
    stamp(a, st, 6);
    if (tid < UPW) {
      // what the backward pass / the hoisted products read of this step's cell: stored only now (behind the energy pass, while wave 1 takes the softmax), so that these
      // stores did not sit in front of this wave's polling loads (one vmcnt queue for loads and stores)
      const int u = g * UPW + tid;
      p.ca[((long)n * S1 + slot) * A + u] = cstate;
      T* gp = (T*)p.ga + ((long)n * S1 + slot) * 4 * A;
      stf(gp + u, sv[0]); stf(gp + A + u, sv[1]); stf(gp + 2 * A + u, sv[2]); stf(gp + 3 * A + u, sv[3]);
      stf((T*)p.hc + ((long)n * S1 + slot) * HC + u, sv[4]);
      if (st + 1 < p.S) stf((T*)p.xa + ((long)n * S1 + slot + 1) * XA + D2 + Dsp + u, sv[4]);
    }
    if (wave == 1) {
      // local softmax: lane = local position
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
    stamp(a, st, 7);
    // ---- (6) partial next-prenet sums over the own positions, published with the softmax pieces
    {
      const int c = tid % D1, th = tid / D1;
      float s = 0.f;
      for (int tl = th; tl < tn; tl += CXG) s = fmaf(es[tl], pv_s[tl * D1 + c], s);
      red[tid] = s;
    }
    lds_barrier();
    if (tid < D1) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < CXG; ++q) s += red[q * D1 + tid];
      put_granule(x3 + (size_t)g * X3N + tid, tag, s);
    } else if (tid < D1 + 2) {
      put_granule(x3 + (size_t)g * X3N + tid, tag, sc[tid - D1]);
    } else if (tid < D1 + 2 + TSMAX) {
      put_granule(x3 + (size_t)g * X3N + tid, tag, es[tid - D1 - 2]);
    }
    stamp(a, st, 8);
    // ---- in the shadow of exchange 3: the h half of the NEXT step's gates (xs[D2 ..] = h(s) since (4))
    hpart = dot_regs<GKH>(wgh, xs + D2 + gq * GKH);
    // ---- (7) gather X3, combine
    if (!gather_granules<(CG * X3N + CT - 1) / CT>(x3, CG * X3N, tag, gath, tid, a.status, 2)) sc[2] = 1.f;        // the abort word: zero since the kernel's start
    lds_barrier();
    if (sc[2] != 0.f) return;
    stamp(a, st, 9);
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
    lds_barrier();                                   // al / p1s are rewritten below: everyone is done with step s
    // the whole alignment (every workgroup needs its neighbours' positions for the location filter)
    for (int t = tid; t < p.Tia; t += CT) {
      float v = 0.f;
      if (t < L) {
        const int q = t / ts;
        v = gath[q * X3N + D1 + 2 + (t - q * ts)] * scl[q] * inv;
      }
      if (t < 256) al[APAD + t] = v;
      const bool mine = t < L ? (t / ts == g) : (g == CG - 1);
      if (mine) {
        p.align[((long)n * S1 + slot) * p.Tia + t] = v;
        if (p.align_t) stf((T*)p.align_t + ((long)n * S1 + slot) * p.Tia + t, v);
        if (INFER) put_granule(a.alx + (size_t)n * p.Tia + t, tag, v);      // the first decoder LSTM waits for it
      }
    }
    if (tid < D1) {
      // free-running: the next frame term comes back through both decoder LSTMs (its tag = the consuming step + 1);
      // the alignment above went out first - they cannot answer before they have it
      if (INFER && st + 1 < p.S && !poll_granule(a.f1x + (size_t)n * D1 + tid, tag + 1, f1n, a.status, 3)) sc[3] = 1.f;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < CG; ++q) s = fmaf(scl[q], gath[q * X3N + tid], s);
      const float v = fmaxf(fmaf(s, inv, f1n), 0.f);
      p1s[tid] = v;
      if (st + 1 < p.S && tid / (D1 / CG) == g) stf((T*)p.p1 + ((long)n * S1 + slot + 1) * D1 + tid, v);
    }
    lds_barrier();
    if (INFER && sc[3] != 0.f) return;
  }
}

template <typename T, typename C>
__global__ __launch_bounds__(CT) void attn_cluster_fwd_kernel(ACArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm_attn_fwd[];
  attn_fwd_body<T, C, false>(a, sm_attn_fwd, blockIdx.x);
}

// ===================================================================================== backward
// Backward through time, same clusters and ownership as the forward kernel.  Per step (walking s = S .. 1):
//   dalign of the own positions = da0 (hoisted) + pv . dp1(s+1) + carry;  its share of the softmax-backward dot product
//           sum_t align[t] dalign[t] goes out at once (E1, ONE granule per workgroup) and is collected only after the
//           energy pass, which does not need it - the exchange latency is hidden.  (Assembling the dot product from
//           forward-pass sums instead would avoid E1, but de = a (dalign - dot) is a difference of nearly equal numbers
//           in a peaked softmax: only a dot product summed from the SAME dalign values keeps the rounding errors
//           common-mode; measured 3e-3 against 1e-4 on the encoder gradients.)
//   energy pass on the own positions: g1[t,u] = v[u] (1 - tanh^2), laid out as the A operand of the exact fp32 MFMA
//           (16x16x4): Z[t,k] = sum_u g1[t,u] Wcl[k,u] comes out of the matrix core, g1 stays in registers;
//   E2:     dq partials sum_t de[t] g1[t,u] (16-lane row reductions)   -> dq, dh through W_query, cell gradient
//   E3:     partial input gradients dga_own . Watt[:, own]^T (K values) + the workgroup's contributions to the next
//           step's location-filter carry (ts + 6 values)               -> dp2, the recurrent dh, carry
//   then dp1 = (dp2 . W2^T) masked, redundantly in every workgroup (W2 in registers), which is the next step's dvec.
template <int A_, int D1_, int D2_>
struct BCfg : Cfg<A_, D1_, D2_> {
  using B = Cfg<A_, D1_, D2_>;
  static constexpr int NQ = 1536 / B::K;                 // column groups of the input-gradient product (4 | 8)
  static constexpr int CPQ = B::GC / NQ;                 // gate columns per group (32 | 4)
  static constexpr int PPT = 3;                          // (row, group) pairs per thread: NQ * K = 3 * CT
  static constexpr int W2H = CT / B::D1, W2K = B::D2 / W2H;       // dp1 product: halves, k per thread
  static constexpr int CCN = TSMAX + 8;                  // carry contributions (ts + 6 used) + dcar
  static constexpr int E3N = B::K + CCN;                 // granules per workgroup, exchange 3
  static constexpr int EMAX = E3N > B::A ? E3N : B::A;
  static constexpr int UL = B::A / 16;                   // query columns per lane in the dhq product
  static_assert(NQ * B::K == PPT * CT && B::GC % NQ == 0 && B::D2 % W2H == 0 && B::A % 32 == 0 && B::D1 == 256, "shape");
};

template <typename T, typename C>
__global__ __launch_bounds__(CT) void attn_cluster_bwd_kernel(ACArgs a) {
  constexpr int A = C::A, D1 = C::D1, D2 = C::D2, UPW = C::UPW, GC = C::GC, K = C::K;
  constexpr int NQ = C::NQ, CPQ = C::CPQ, PPT = C::PPT, W2H = C::W2H, W2K = C::W2K, CCN = C::CCN, E3N = C::E3N, UL = C::UL;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const ns_taco2_attn_params& p = a.p;
  float* dvec = sm;                        // [D1]  dp1 of the step after (the vector dotted with the pv rows)
  float* dp2s = dvec + D1;                 // [D2]
  float* dgs = dp2s + D2;                  // [GC]  gate gradients of the own units, (gate, unit)
  float* red = dgs + GC;                   // [NQ * K] partial sums
  // the per-step history images exist twice (parity of the step): the next step's are filled in the middle of this
  // one, when its loads have long landed - at the top of a step they would wait behind the write-through stores
  // that end the step before (one vmcnt queue: 1.1 us per step)
  float* qs0 = red + NQ * K;               // [A]
  float* al0 = qs0 + A;                    // [APAD + 256 + APAD] alignment of step s-1
  float* acur0 = al0 + 256 + 2 * APAD;     // [TSMAX] alignment of step s, own positions
  float* da0s0 = acur0 + TSMAX;            // [TSMAX]
  float* dav = da0s0 + TSMAX;              // [TSMAX] dalign
  float* dev = dav + TSMAX;                // [TSMAX] energy gradients
  float* carry = dev + TSMAX;              // [TSMAX] location-filter carry for the own positions
  float* Gs = carry + TSMAX;               // [TSMAX][8]
  float* zred = Gs + TSMAX * 8;            // [8 waves][TSMAX][8]
  float* hrec = zred + 8 * TSMAX * 8;      // [UPW] recurrent part of dh for the own units
  float* dq_s = hrec + UPW;                // [A]
  float* p1m0 = dq_s + A;                  // [D1] p1 of this step (ReLU mask)
  float* p2m0 = p1m0 + D1;                 // [D2]
  float* gts0 = p2m0 + D2;                 // [GC] saved gates of the own units
  float* sc = gts0 + GC;                   // [16]  [0] dot, [1] dcar, [2] abort, [4..] block_sum scratch
  float* gath = sc + 48;                   // [CG][EMAX]
  float* keys_s = gath + CG * C::EMAX;     // [TSMAX][A + KPAD]
  float* pv_s = keys_s + TSMAX * (A + KPAD);        // [TSMAX][D1]
  float* wq_s = pv_s + TSMAX * D1;         // [UPW][A]
  float* cst_s = wq_s + UPW * A;           // [KWMAX + 1][A + KPAD]
  float* qs1 = cst_s + (KWMAX + 1) * (A + KPAD);
  float* al1 = qs1 + A;
  float* acur1 = al1 + 256 + 2 * APAD;
  float* da0s1 = acur1 + TSMAX;
  float* p1m1 = da0s1 + TSMAX;
  float* p2m1 = p1m1 + D1;
  float* gts1 = p2m1 + D2;                 // ... + GC

  const int tid_ = threadIdx.x;
  const int n = blockIdx.x / CG, g = blockIdx.x % CG;
  const long S1 = p.S + 1;
  const int Dsp = p.Dsp, HC = A + p.E;
  const int XA = D2 + Dsp + A;
  const int L = min(p.lengths ? p.lengths[n] : p.Ti, p.Ti);
  const int ts = max(1, (L + CG - 1) / CG);
  const int t0 = g * ts, tn = max(0, min(L, t0 + ts) - t0);
  const int half = (p.kw - 1) / 2;
  u64* e1 = a.x1 + (size_t)n * CG;
  u64* e2 = a.x2 + (size_t)n * CG * A;
  u64* e3 = a.x3 + (size_t)n * CG * E3N;

  // ---------------------------------------------------------------- resident weights
  const T* W2 = (const T*)p.w2;            // [D1][D2]
  const T* Watt = (const T*)p.watt;        // [D2 + Dsp + A][4A]
  const T* Wq = (const T*)p.wq;            // [A][A]
  const int tid = tid_;
  float wxr[PPT][CPQ];                     // Watt[row k][own columns of group qr] for this thread's (k, qr) pairs
#pragma unroll
  for (int jj = 0; jj < PPT; ++jj) {
    const int pi = tid + CT * jj, k = pi % K, qr = pi / K;
    const T* b = Watt + (long)(k < D2 ? k : k + Dsp) * 4 * A;
#pragma unroll
    for (int i = 0; i < CPQ; ++i) {
      const int c = qr * CPQ + i;
      wxr[jj][i] = ldf(b + (c / UPW) * A + g * UPW + (c % UPW));
    }
  }
  float w2r[W2K];                          // W2[c1][hf * W2K + i]
  {
    const T* b = W2 + (long)(tid % D1) * D2 + (tid / D1) * W2K;
#pragma unroll
    for (int i = 0; i < W2K; ++i) w2r[i] = ldf(b + i);
  }
  for (int i = tid; i < UPW * A; i += CT) wq_s[i] = ldf(Wq + (long)g * UPW * A + i);
  for (int i = tid; i < (KWMAX + 1) * A; i += CT) {
    const int k = i / A, u = i % A;
    cst_s[k * (A + KPAD) + u] = k < p.kw ? p.wcl[k * A + u] : (k == KWMAX ? p.v[u] : 0.f);
  }
  {
    const float* kn = p.keys + ((long)n * p.Pi + p.padl_i + t0) * A;
    for (int i = tid; i < TSMAX * A; i += CT) keys_s[(i / A) * (A + KPAD) + i % A] = (i / A) < tn ? kn[i] : 0.f;
    const T* pvn = (const T*)p.pv + ((long)n * p.Pi + p.padl_i + t0) * D1;
    for (int i = tid; i < TSMAX * D1; i += CT) pv_s[i] = (i / D1) < tn ? ldf(pvn + i) : 0.f;
    for (int i = tid; i < 256 + 2 * APAD; i += CT) { al0[i] = 0.f; al1[i] = 0.f; }
    for (int i = tid; i < D1; i += CT) dvec[i] = 0.f;
    for (int i = tid; i < TSMAX; i += CT) { carry[i] = 0.f; dev[i] = 0.f; dav[i] = 0.f; acur0[i] = 0.f; da0s0[i] = 0.f; acur1[i] = 0.f; da0s1[i] = 0.f; }
    for (int i = tid; i < TSMAX * 8; i += CT) Gs[i] = 0.f;
    if (tid < UPW) hrec[tid] = 0.f;
    if (tid < 48) sc[tid] = 0.f;
  }
  float dcc = 0.f;                         // cell-state gradient carried to the step before (owner lanes)
  // history of a step, one value per thread and role (read straight from the forward pass' buffers)
  float h_q = 0.f, h_p1 = 0.f, h_p2 = 0.f, h_alp = 0.f, h_a = 0.f, h_da0 = 0.f, h_gt = 0.f;
  float h_dhc = 0.f, h_c = 0.f, h_cp = 0.f;
  auto load_history = [&](int st_, int tid) {
    const long rowS = (long)n * S1 + st_ + 1;
    if (tid < A) h_q = p.q[rowS * A + tid];
    if (tid < D1) h_p1 = ldf((const T*)p.p1 + rowS * D1 + tid);
    if (tid < D2) h_p2 = ldf((const T*)p.xa + rowS * XA + tid);
    h_a = 0.f; h_da0 = 0.f;
    if (tid < 256) {
      h_alp = tid < p.Tia ? p.align[(rowS - 1) * p.Tia + tid] : 0.f;
    } else if (tid < 256 + TSMAX) {
      const int tl = tid - 256;
      if (tl < tn) { h_a = p.align[rowS * p.Tia + t0 + tl]; h_da0 = p.da0[rowS * p.Tia + t0 + tl]; }
    } else if (tid >= 320 && tid < 320 + GC) {
      const int c = tid - 320;
      h_gt = ldf((const T*)p.ga + rowS * 4 * A + (c / UPW) * A + g * UPW + (c % UPW));
    }
    if ((tid & 15) == 0 && (tid >> 4) < UPW) {          // the cell owners (P6)
      const int u = g * UPW + (tid >> 4);
      h_dhc = p.dhc[rowS * HC + u];
      h_c = p.ca[rowS * A + u];
      h_cp = st_ > 0 ? p.ca[(rowS - 1) * A + u] : 0.f;
    }
  };
  // history registers -> the LDS images of parity `par`
  auto store_history = [&](int par, int tid) {
    float* const qs = par ? qs1 : qs0; float* const al = par ? al1 : al0; float* const acur = par ? acur1 : acur0;
    float* const da0s = par ? da0s1 : da0s0; float* const p1m = par ? p1m1 : p1m0; float* const p2m = par ? p2m1 : p2m0;
    float* const gts = par ? gts1 : gts0;
    if (tid < A) qs[tid] = h_q;
    if (tid < D1) p1m[tid] = h_p1;
    if (tid < D2) p2m[tid] = h_p2;
    if (tid < 256) al[APAD + tid] = h_alp;
    if (tid >= 256 && tid < 256 + TSMAX) { acur[tid - 256] = h_a; da0s[tid - 256] = h_da0; }
    if (tid >= 320 && tid < 320 + GC) gts[tid - 320] = h_gt;
  };
  // The images below are zeroed above by whichever threads the fill loops hand them to and written here by the
  // threads that own the history values - other waves.  Without this barrier a wave that reaches store_history before
  // the zero fill of another wave has run loses its values: acur / da0s (or the previous alignment) of the first
  // backward step read as zero in one workgroup, once in 20 - 50 launches (found as a two-valued gradient of one
  // utterance under profiles/tools/determinism_probe.py; the forward kernel fills every image from one place).
  __syncthreads();
  load_history(p.S - 1, tid_);
  store_history((p.S - 1) & 1, tid_);
  float o_dhc = h_dhc, o_c = h_c, o_cp = h_cp;              // the cell owners' operands of the current step
  __syncthreads();
  // The energy pass of a step needs only that step's HISTORY (query, previous alignment) - nothing of the backward
  // chain - so it runs one step ahead, in the shadow of exchange 3 of the step before (its history images are in LDS
  // since the middle of that step); g1 stays in registers until P4 of its step, Z^T waits in zred.
  float g1v[16];                             // [unit tile][position tile][q]: position 16 rt + c, unit ub + 16 ut + 4 g4 + q
  auto energy_pass = [&](const float* qs, const float* al, int lane, int wave) {
    const int r = lane & 15, kq = lane >> 4;
    if (wave < A / 32) {
    const int ub = wave * 32;
    f32x4 accz[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // Z^T: rows = taps 4 kq + q', col = position r
#pragma unroll
    for (int ut = 0; ut < 2; ++ut) {
      const int u4 = ub + ut * 16 + 4 * kq;
      const f32x4 q4 = *(const f32x4*)(qs + u4), v4 = *(const f32x4*)(cst_s + KWMAX * (A + KPAD) + u4);
      const float wa0 = cst_s[kq * (A + KPAD) + ub + ut * 16 + r], wa1 = cst_s[(4 + kq) * (A + KPAD) + ub + ut * 16 + r];
      f32x4 wz = {0.f, 0.f, 0.f, 0.f};                                  // Wcl[tap r][units u4 .. u4 + 3]; rows past kw are zero
      if (r < KWMAX) wz = *(const f32x4*)(cst_s + r * (A + KPAD) + u4);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const f32x4 k4 = *(const f32x4*)(keys_s + (rt * 16 + r) * (A + KPAD) + u4);
        f32x4 acc = {k4[0] + q4[0], k4[1] + q4[1], k4[2] + q4[2], k4[3] + q4[3]};
        const float* aw = al + APAD + t0 + rt * 16 + r - half + kq;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0, aw[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa1, aw[4], acc, 0, 0, 0);
        const bool live = rt * 16 + r < tn;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float th = tanhf_(acc[q]);
          const float gq = live ? v4[q] * (1.f - th * th) : 0.f;
          g1v[ut * 8 + rt * 4 + q] = gq;
          accz[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[q], gq, accz[rt], 0, 0, 0);
        }
      }
    }
    if (kq < 2) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) zred[(wave * TSMAX + rt * 16 + r) * 8 + 4 * kq + q] = accz[rt][q];
    }
  }
  };
  {
    const int par0 = (p.S - 1) & 1;
    energy_pass(par0 ? qs1 : qs0, par0 ? al1 : al0, tid_ & 63, tid_ >> 6);      // the first step's own pass
  }

  for (int st = p.S - 1; st >= 0; --st) {
    const long slot = st + 1;
    const unsigned tag = (unsigned)(p.S - st);
    int tid = tid_;
    asm volatile("" : "+v"(tid));           // see the forward kernel: no loop-invariant address hoisting
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int par = st & 1;
    float* const qs = par ? qs1 : qs0; float* const al = par ? al1 : al0; float* const acur = par ? acur1 : acur0;
    float* const da0s = par ? da0s1 : da0s0; float* const p1m = par ? p1m1 : p1m0; float* const p2m = par ? p2m1 : p2m0;
    float* const gts = par ? gts1 : gts0;
    const long rowS = (long)n * S1 + slot;

    stamp(a, p.S - 1 - st, 0);
    // ---- P0/P1: the history images of this step were filled in the middle of the step before; the loads of step
    //      s-1 go out now and stay in flight until the middle of this step
    if (st > 0) load_history(st - 1, tid);
    lds_barrier();
    stamp(a, p.S - 1 - st, 1);
    // ---- P2: dalign of the own positions: thread = (position tid / 16, 16 columns each); then this workgroup's share
    //      of the softmax-backward dot product sum_t align[t] dalign[t] goes out (exchange 1, one granule)
    {
      const int tl = tid >> 4, cq = tid & 15;
      float s = 0.f;
      if (tl < tn) {
        // columns cq * 4 + 64 i: the 16 lanes of a position read consecutive 16-byte chunks (16 columns in a row per
        // lane would put every fourth lane on the same banks)
        const float* pr = pv_s + tl * D1 + cq * 4;
        const float* dv = dvec + cq * 4;
#pragma unroll
        for (int i = 0; i < D1 / 4; i += 16) {
          const float4 x = *(const float4*)(pr + 4 * i), y = *(const float4*)(dv + 4 * i);
          s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
        }
      }
      s = row16_sum(s);
      if (cq == 0 && tl < TSMAX) dav[tl] = tl < tn ? da0s[tl] + s + carry[tl] : 0.f;
    }
    lds_barrier();
    if (wave == 7) {                         // a wave that takes no part in the energy pass for A = 64 either
      const float d = wave_sum(lane < tn ? acur[lane] * dav[lane] : 0.f);
      if (lane == 0) put_granule(e1 + g, tag, d);
    }
    stamp(a, p.S - 1 - st, 2);
    // ---- P3 (runs one step AHEAD, see energy_pass above): energy pass.  x^T[u][t] = (keys[t][u] + q[u]) + sum_k Wcl[k][u] align[t + k - half] is taken on the matrix
    //      core TRANSPOSED (A = the folded filter: row = unit, k = tap; B = the alignment window: col = position; C = keys
    //      + q as one 16-byte LDS read each), so a lane holds 4 UNITS of one position - exactly the B operand of the next
    //      product Z^T[k][t] = sum_u Wcl[k][u] g1[t][u] when its k-steps are taken as the unit sets {4 g4 + q}: g1 =
    //      v (1 - tanh^2) goes from the accumulators of the first product straight into the second (round 2 formed x
    //      with 128 FMAs per lane from 112 scalar LDS reads: 2.5 us per step for this block).  Needs no dot product yet.
    const int r = lane & 15, kq = lane >> 4;
    // ---- exchange 1 has been under way all along: the dot product, then the energy gradients
    if (wave == 7) {
      if (!gather_granules<1>(e1, CG, tag, gath, lane < CG ? lane : CG, a.status, 5)) sc[2] = 1.f;      // lanes >= CG read nothing
      float d = lane < CG ? gath[lane] : 0.f;
      d = wave_sum(d);
      if (lane == 0) sc[0] = d;
    }
    lds_barrier();
    stamp(a, p.S - 1 - st, 3);
    if (tid < TSMAX) {
      float de = 0.f;
      if (tid < tn) {
        de = acur[tid] * (dav[tid] - sc[0]);
        p.de[rowS * p.Tia + t0 + tid] = de;
      }
      dev[tid] = de;
    }
    lds_barrier();
    // ---- P4: dq partial of unit u = sum over the positions of de[t] g1[t, u] (a DPP row holds 16 positions, the two
    //      tiles add in-thread), published as exchange 2;  G[t][k] = de[t] * sum over the unit blocks of Z
    if (wave < A / 32) {
      const float de0 = dev[r], de1 = dev[r + 16];
#pragma unroll
      for (int j = 0; j < 8; ++j) g1v[j] = row16_sum(fmaf(de0, g1v[(j >> 2) * 8 + (j & 3)], de1 * g1v[(j >> 2) * 8 + 4 + (j & 3)]));
      if (r == 0) {        // j = 4 ut + q: unit ub + 16 ut + 4 kq + q
#pragma unroll
        for (int j = 0; j < 8; ++j) put_granule(e2 + (size_t)g * A + wave * 32 + (j >> 2) * 16 + 4 * kq + (j & 3), tag, g1v[j]);
      }
    }
    if (tid < TSMAX * 8) {
      const int tl = tid >> 3, k = tid & 7;
      float z = 0.f;
#pragma unroll
      for (int w = 0; w < A / 32; ++w) z += zred[(w * TSMAX + tl) * 8 + k];
      Gs[tid] = (tl < tn && k < p.kw) ? dev[tl] * z : 0.f;
    }
    lds_barrier();
    // ---- P5: this workgroup's contributions to the carry of step s-1 (positions t0-half .. t0+ts+kw-half-2)
    float ccv = 0.f;                         // threads < CCN keep their value for the E3 publish
    if (tid < TSMAX + 6) {
      // carry[t'] += G[t' - k + half][k]; local: t' = t0 - half + tid  ->  G row (tid - k)
#pragma unroll
      for (int k = 0; k < KWMAX; ++k) {
        const int row = tid - k;
        if (k < p.kw && row >= 0 && row < TSMAX) ccv += Gs[row * 8 + k];
      }
    }
    stamp(a, p.S - 1 - st, 4);
    // ---- gather E2: dq = sum of the partials
    if (!gather_granules<(CG * A + CT - 1) / CT>(e2, CG * A, tag, gath, tid, a.status, 3)) sc[2] = 1.f;        // the abort word: zero since the kernel's start
    lds_barrier();
    if (sc[2] != 0.f) return;
    stamp(a, p.S - 1 - st, 5);
    if (st > 0) store_history(par ^ 1, tid);       // step s-1's history: loaded at the top of this step, landed long ago
    if (tid < A) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < CG; ++q) s += gath[q * A + tid];
      dq_s[tid] = s;
      if (tid / UPW == g) stf((T*)p.dq + rowS * A + tid, s);
    }
    lds_barrier();
    // ---- P6: dh of the own units through W_query, then the cell gradient: thread = (unit tid / 16, UL columns each)
    {
      const int j = tid >> 4, uq = tid & 15;
      float s = 0.f;
      if (j < UPW) {
        const float* wr = wq_s + j * A + uq * 4;                // columns uq * 4 + 64 i, as in P2
        const float* dq = dq_s + uq * 4;
#pragma unroll
        for (int i = 0; i < UL; i += 4) {
          const float4 x = *(const float4*)(wr + 16 * i), y = *(const float4*)(dq + 16 * i);
          s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
        }
      }
      s = row16_sum(s);
      if (uq == 0 && j < UPW) {
        const int u = g * UPW + j;
        const float dh = o_dhc + s + hrec[j];
        const float gi = gts[j], gj = gts[UPW + j], gf = gts[2 * UPW + j], go = gts[3 * UPW + j];
        const float c = o_c, cp = o_cp;
        const float tc = tanhf_(c);
        const float d_o = dh * tc * go * (1.f - go);
        const float dc = dh * go * (1.f - tc * tc) + dcc;
        const float d_i = dc * gj * gi * (1.f - gi);
        const float d_j = dc * gi * (1.f - gj * gj);
        const float d_f = dc * cp * gf * (1.f - gf);
        dcc = dc * gf;
        dgs[j] = d_i; dgs[UPW + j] = d_j; dgs[2 * UPW + j] = d_f; dgs[3 * UPW + j] = d_o;
        T* dg = (T*)p.dga + rowS * 4 * A;
        stf(dg + u, d_i); stf(dg + A + u, d_j); stf(dg + 2 * A + u, d_f); stf(dg + 3 * A + u, d_o);
        if (p.dga_bf16 && sizeof(T) == 4) {
          bf16_t* db = (bf16_t*)p.dga_bf16 + rowS * 4 * A;
          db[u] = (bf16_t)d_i; db[A + u] = (bf16_t)d_j; db[2 * A + u] = (bf16_t)d_f; db[3 * A + u] = (bf16_t)d_o;
        }
      }
    }
    if (st > 0) { o_dhc = h_dhc; o_c = h_c; o_cp = h_cp; }
    lds_barrier();
    stamp(a, p.S - 1 - st, 6);
    // ---- P7: partial input gradients dga_own . Watt[k, own]^T for every input row k, published with the carry pieces
#pragma unroll
    for (int jj = 0; jj < PPT; ++jj) {
      const int pi = tid + CT * jj, qr = pi / K;
      red[pi] = dot_regs<CPQ>(wxr[jj], dgs + qr * CPQ);          // red[qr * K + k], pi = qr * K + k
    }
    lds_barrier();
    if (tid < K) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) s += red[q * K + tid];
      put_granule(e3 + (size_t)g * E3N + tid, tag, s);
    }
    if (tid < CCN) put_granule(e3 + (size_t)g * E3N + K + tid, tag, ccv);
    // in the shadow of exchange 3: the energy pass of step s-1 (history images of parity par ^ 1, filled behind E2)
    if (st > 0) energy_pass(par ? qs0 : qs1, par ? al0 : al1, lane, wave);
    stamp(a, p.S - 1 - st, 7);
    if (!gather_granules<(CG * E3N + CT - 1) / CT>(e3, CG * E3N, tag, gath, tid, a.status, 4)) sc[2] = 1.f;        // the abort word: zero since the kernel's start
    lds_barrier();
    if (sc[2] != 0.f) return;
    stamp(a, p.S - 1 - st, 8);
    // ---- P8: dp2 (masked), recurrent dh, carry and dcar for the step before
    if (tid < D2) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < CG; ++q) s += gath[q * E3N + tid];
      s = p2m[tid] > 0.f ? s : 0.f;
      dp2s[tid] = s;
      if (tid / (D2 / CG) == g) stf((T*)p.dp2 + rowS * D2 + tid, s);
    } else if (tid < D2 + UPW) {
      const int j = tid - D2;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < CG; ++q) s += gath[q * E3N + D2 + g * UPW + j];
      hrec[j] = s;
    } else if (tid >= 256 && tid < 256 + TSMAX) {
      const int tl = tid - 256;
      float s = 0.f;
      if (tl < tn) {
        const int tp = t0 + tl;
#pragma unroll
        for (int q = 0; q < CG; ++q) {
          const int j = tp - q * ts + half;            // index into workgroup q's contributions (its t0 - half ...)
          if (j >= 0 && j < TSMAX + 6) s += gath[q * E3N + K + j];
        }
      }
      carry[tl] = s;
    }
    lds_barrier();
    stamp(a, p.S - 1 - st, 9);
    // ---- P9: dp1 = (dp2 . W2^T) masked: next dvec (every workgroup computes all of it)
    red[tid] = dot_regs<W2K>(w2r, dp2s + (tid / D1) * W2K);
    lds_barrier();
    if (tid < D1) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < W2H; ++q) s += red[q * D1 + tid];
      s = p1m[tid] > 0.f ? s : 0.f;
      dvec[tid] = s;
      if (tid / (D1 / CG) == g) stf((T*)p.df1 + rowS * D1 + tid, s);
    }
    lds_barrier();
  }
}


// ===================================================================================== free-running decode (round 3)
// Synthesis (tacotron2.py:78-83 with TacoTestHelper, helpers.py:7-38): the frame fed to step s+1 is the last frame
// predicted at step s, so nothing can be hoisted out of the time loop and the launch-per-step form needs 9 dependent
// launches per decoder step (80 us per step at batch 1).  Here the whole loop is ONE launch with three roles:
//   workgroups [0, 8N)            the attention RNN clusters of attn_fwd_body<INFER> above
//   the next NW workgroups        decoder LSTM 1, UPW = 12 units each (NW = ceil(D / 12))
//   the last NW workgroups        decoder LSTM 2, 12 units each, plus the frame feedback
// Every weight stays in registers as fp32 (matrix-VECTOR products at one or two utterances: exact fp32 FMAs, as in the
// attention clusters); vectors travel as 8-byte {step tag, fp32} granules (the data is its own flag).  Per step:
//   attention:  h_att(s) -> x2 (its own exchange), align(s) -> alx, then waits for f1(s+1)
//   LSTM 1:     gathers h_att(s) + align(s); the context enters through the PROJECTED memory PM = memory . W_ctx of its
//               own 48 gate columns (LDS, built once per call), so no 512-wide context is formed in the loop;
//               adds the recurrent half W_h . h1(s-1) it computed in the shadow of the previous step; cell; h1(s) -> pub1
//   LSTM 2:     gathers h1(s), adds its recurrent half, cell, h2(s) -> pub2 + history; then every LSTM-2 workgroup
//               gathers h2(s) for its next recurrent half and for its columns of the FOLDED feedback
//               f1(s+1) = h2(s) . (W_proj[:, last frame] . W_prenet1[frame rows]) + b  -> f1x   (the output projection
//               itself is off the critical path: ONE product over the h2 history after the loop)
// pub1 / pub2 are double-buffered by step parity (a producer may be a step ahead of a peer's second gather); x2, alx and
// f1x need one buffer: their next writer sits behind a dependency chain through every reader.
// Units per workgroup: 16 for LSTM 1 (K = A + D resident inputs: 160 weights per thread at the shipped widths), 8 for
// LSTM 2 (K = 2D: 128 per thread) - with 12 + 12 the second cell's 192 weights per thread spilled.
constexpr int DEC_UPW1 = 16, DEC_UPW2 = 8, DEC_KS = 32;

struct DecArgs {
  ACArgs att;
  int N, S, E, D, NW1, NW2;
  const float* w1; const float* b1;        // decoder/lstm_1 kernel [(A + E + D), 4D] fp32, bias [4D]
  const float* w2; const float* b2;        // decoder/lstm_2 kernel [2D, 4D], bias
  const float* wpf; const float* bpf;      // folded feedback [D, D1], [D1]
  u64* pub1; u64* pub2;                    // [2][N][D]
  float* h2hist;                           // [N][S + 1][D], slot s + 1
};

// PER granules per thread, all loads of a pass in flight together; at(i, src, dst) maps item i to its granule and its
// LDS word.  Returns false on time-out / raised status.
template <int PER, typename F>
__device__ __forceinline__ bool dec_gather(int total, unsigned tag, int tid, int* status, int code, F at) {
  const u64* src[PER];
  float* dst[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid + j * CT;
    src[j] = nullptr; dst[j] = nullptr;
    if (i < total) at(i, src[j], dst[j]);
  }
  u64 v[PER];
  unsigned spins = 0, clk0 = 0;
  bool ok, good = true;
  do {
    ok = true;
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j] = src[j] ? __hip_atomic_load(src[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((u64)tag << 32);
#pragma unroll
    for (int j = 0; j < PER; ++j) ok = ok && ((unsigned)(v[j] >> 32) == tag);
    if (!ok && (++spins & 1023u) == 0) {
      if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { good = false; ok = true; }
      else if (ns_spin_timed_out(clk0)) { atomicExch(status, code); good = false; ok = true; }
    }
  } while (!ok);
#pragma unroll
  for (int j = 0; j < PER; ++j) if (dst[j]) *dst[j] = __uint_as_float((unsigned)v[j]);
  return good;
}

// One decoder LSTM's workgroup w.  WHICH = 1: inputs h_att (A, registers) + context (through PM) + own h1; 2: h1 + own h2.
// A workgroup owns UPW units = COLS = 4 UPW gate columns (column = gate * UPW + unit); thread (cg = tid & 15, ks = tid >> 4)
// holds the CPT = COLS / 16 columns CPT cg .. of the k slice ks of each input.
template <typename T, typename C, int E, int D, int NR, int WHICH>
__device__ __forceinline__ void dec_lstm_role(const DecArgs& d, float* sm, const int w) {
  constexpr int A = C::A, D1 = C::D1, X2N = C::X2N, UPWA = C::UPW;
  constexpr int UPW = WHICH == 1 ? DEC_UPW1 : DEC_UPW2, COLS = 4 * UPW, CPT = COLS / 16, RS = COLS + 1;
  constexpr int KX = WHICH == 1 ? A : D, KPX = KX / DEC_KS, KPH = D / DEC_KS;
  static_assert(KX % DEC_KS == 0 && D % DEC_KS == 0 && COLS % 16 == 0, "slices");
  const ns_taco2_attn_params& p = d.att.p;
  const int tid = threadIdx.x, cg = tid & 15, ks = tid >> 4;
  const int N = d.N, Tia = p.Tia, u0 = w * UPW;
  const int NW = WHICH == 1 ? d.NW1 : d.NW2;
  const int FPW = (D1 + NW - 1) / NW, f0 = w * FPW, fn = max(0, min(D1, f0 + FPW) - f0);      // own feedback columns (LSTM 2)
  // ---- LDS
  float* xv = sm;                               // [NR][KX]   gathered input of the x part
  float* hv = xv + NR * KX;                     // [NR][D]    gathered own h of the previous step
  float* red = hv + NR * D;                     // [32][COLS + 1] partial sums over the k slices
  float* zr = red + DEC_KS * RS;                // [NR][COLS] recurrent half, computed a phase ahead
  float* zs = zr + NR * COLS;                   // [COLS]     gate pre-activations of the row in work
  float* flag = zs + COLS;                      // [4]        [0] abort
  float* xtra = flag + 4;                       // WHICH 1: al_s [NR][Tia], PM [NR][Tia][COLS]; 2: wpf_s [FPW][D], red2 [32][64]
  float* al_s = xtra;
  float* PM = al_s + NR * Tia;
  float* wpf_s = xtra;
  float* red2 = wpf_s + (size_t)FPW * D;
  // ---- resident weights
  const float* W = WHICH == 1 ? d.w1 : d.w2;
  const float* bias = WHICH == 1 ? d.b1 : d.b2;
  const int rowh0 = WHICH == 1 ? A + E : D;
  float wx[CPT][KPX], wh[CPT][KPH];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int col = CPT * cg + j, gate = col / UPW, ul = col % UPW;
    const bool okc = u0 + ul < D;
    const long wc = (long)gate * D + u0 + ul;
#pragma unroll
    for (int i = 0; i < KPX; ++i) wx[j][i] = okc ? W[(long)(ks * KPX + i) * 4 * D + wc] : 0.f;
#pragma unroll
    for (int i = 0; i < KPH; ++i) wh[j][i] = okc ? W[(long)(rowh0 + ks * KPH + i) * 4 * D + wc] : 0.f;
  }
  if (tid < 4) flag[tid] = 0.f;
  for (int i = tid; i < NR * COLS; i += CT) zr[i] = 0.f;
  if (WHICH == 1) {
    // projected memory of the own gate columns: PM[n][t][col] = sum_e memory[n][t][e] W[A + e][col]
    const T* val = (const T*)p.values;
    for (int o = tid; o < N * Tia * COLS; o += CT) {
      const int col = o % COLS, t = (o / COLS) % Tia, n = o / (COLS * Tia);
      const int gate = col / UPW, ul = col % UPW;
      float s = 0.f;
      if (t < p.Ti && u0 + ul < D) {
        const T* vr = val + ((long)n * p.Pi + p.padl_i + t) * E;
        const float* wc = W + (long)A * 4 * D + (long)gate * D + u0 + ul;
        for (int e = 0; e < E; ++e) s = fmaf(ldf(vr + e), wc[(long)e * 4 * D], s);
      }
      PM[o] = s;
    }
  } else {
    for (int o = tid; o < fn * D; o += CT) wpf_s[o] = d.wpf[(long)(o % D) * D1 + f0 + o / D];      // [j][k]
  }
  float cst[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n) cst[n] = 0.f;
  __syncthreads();

  u64* mypub = WHICH == 1 ? d.pub1 : d.pub2;
  for (int st = 0; st < d.S; ++st) {
    const unsigned tag = (unsigned)(st + 1);
    const size_t par = (size_t)(st & 1) * N * D;
    // ---------------- phase a: the inputs of this step
    bool good;
    if (WHICH == 1) {
      good = dec_gather<(NR * A + CT - 1) / CT>(N * A, tag, tid, d.att.status, 4, [&](int i, const u64*& src, float*& dst) {
        const int n = i / A, u = i % A;
        src = d.att.x2 + ((size_t)n * CG + u / UPWA) * X2N + A + u % UPWA;
        dst = xv + n * KX + u;
      });
      good = dec_gather<(NR * 256 + CT - 1) / CT>(N * Tia, tag, tid, d.att.status, 4, [&](int i, const u64*& src, float*& dst) {
        src = d.att.alx + i;
        dst = al_s + i;
      }) && good;
    } else {
      good = dec_gather<(NR * D + CT - 1) / CT>(N * D, tag, tid, d.att.status, 5, [&](int i, const u64*& src, float*& dst) {
        src = d.pub1 + par + i;
        dst = xv + i;
      });
    }
    if (!good) flag[0] = 1.f;
    lds_barrier();
    if (flag[0] != 0.f) return;
    for (int n = 0; n < N; ++n) {
      float acc[CPT];
#pragma unroll
      for (int j = 0; j < CPT; ++j) acc[j] = 0.f;
#pragma unroll
      for (int i = 0; i < KPX; ++i) {
        const float x = xv[n * KX + ks * KPX + i];
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc[j] = fmaf(wx[j][i], x, acc[j]);
      }
      if (WHICH == 1) {
        for (int t = ks; t < Tia; t += DEC_KS) {
          const float av = al_s[n * Tia + t];
          const float* pm = PM + ((size_t)n * Tia + t) * COLS + CPT * cg;
#pragma unroll
          for (int j = 0; j < CPT; ++j) acc[j] = fmaf(av, pm[j], acc[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < CPT; ++j) red[ks * RS + CPT * cg + j] = acc[j];
      lds_barrier();
      if (tid < COLS) {
        const int gate = tid / UPW, ul = tid % UPW;
        float s = (u0 + ul < D ? bias[gate * D + u0 + ul] : 0.f) + zr[n * COLS + tid];
#pragma unroll
        for (int q = 0; q < DEC_KS; ++q) s += red[q * RS + tid];
        zs[tid] = s;
      }
      lds_barrier();
      if (tid < UPW && u0 + tid < D) {
        const float gi = sigmoidf_(zs[tid]), gj = tanhf_(zs[UPW + tid]);
        const float gf = sigmoidf_(zs[2 * UPW + tid] + 1.0f), go = sigmoidf_(zs[3 * UPW + tid]);
        float c = cst[0];
#pragma unroll
        for (int q = 1; q < NR; ++q) c = n == q ? cst[q] : c;
        c = ns_cell_clip(gf * c + gi * gj, p.cell_clip);      // the decoder cells take the attention block's cell_clip
        const float h = go * tanhf_(c);
#pragma unroll
        for (int q = 0; q < NR; ++q) cst[q] = n == q ? c : cst[q];
        put_granule(mypub + par + (size_t)n * D + u0 + tid, tag, h);
        if (WHICH == 2) d.h2hist[((long)n * (d.S + 1) + st + 1) * D + u0 + tid] = h;
      }
      // red / zs are rewritten by the next row only behind the barrier that follows its partial sums
      lds_barrier();
    }
    // ---------------- phase b: everybody's h of this step -> the recurrent half of the next step (and the feedback)
    if (st + 1 < d.S) {
      good = dec_gather<(NR * D + CT - 1) / CT>(N * D, tag, tid, d.att.status, 6, [&](int i, const u64*& src, float*& dst) {
        src = mypub + par + i;
        dst = hv + i;
      });
      if (!good) flag[0] = 1.f;
      lds_barrier();
      if (flag[0] != 0.f) return;
      for (int n = 0; n < N; ++n) {
        float acc[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc[j] = 0.f;
#pragma unroll
        for (int i = 0; i < KPH; ++i) {
          const float x = hv[n * D + ks * KPH + i];
#pragma unroll
          for (int j = 0; j < CPT; ++j) acc[j] = fmaf(wh[j][i], x, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) red[ks * RS + CPT * cg + j] = acc[j];
        if (WHICH == 2) {
          // the own columns of the folded feedback: (column j, k slice) pairs over the threads
          for (int o = tid; o < fn * DEC_KS; o += CT) {
            const int j = o / DEC_KS, q = o % DEC_KS;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < KPH; ++i) s = fmaf(wpf_s[j * D + q * KPH + i], hv[n * D + q * KPH + i], s);
            red2[q * 64 + j] = s;
          }
        }
        lds_barrier();
        if (tid < COLS) {
          float s = 0.f;
#pragma unroll
          for (int q = 0; q < DEC_KS; ++q) s += red[q * RS + tid];
          zr[n * COLS + tid] = s;
        } else if (WHICH == 2 && tid >= 64 && tid < 64 + fn) {
          const int j = tid - 64;
          float s = d.bpf[f0 + j];
#pragma unroll
          for (int q = 0; q < DEC_KS; ++q) s += red2[q * 64 + j];
          put_granule(d.att.f1x + (size_t)n * D1 + f0 + j, tag + 1, s);
        }
        lds_barrier();
      }
    }
  }
}

template <typename T, typename C, int E, int D, int NR>
__global__ __launch_bounds__(CT) void taco2_decode_kernel(DecArgs d) {
  extern __shared__ __attribute__((aligned(16))) float sm_dec[];
  const int na = d.N * CG;
  const int b = blockIdx.x;
  if (b < na) attn_fwd_body<T, C, true>(d.att, sm_dec, b);
  else if (b < na + d.NW1) dec_lstm_role<T, C, E, D, NR, 1>(d, sm_dec, b - na);
  else dec_lstm_role<T, C, E, D, NR, 2>(d, sm_dec, b - na - d.NW1);
}

template <typename C, int E, int D, int NR>
size_t dec_lds_bytes(int Tia, int NW2) {
  const int FPW = (C::D1 + NW2 - 1) / NW2;
  constexpr int C1 = 4 * DEC_UPW1, C2 = 4 * DEC_UPW2;
  const size_t l1 = (size_t)NR * (C::A + D) + DEC_KS * (C1 + 1) + NR * C1 + C1 + 4 + (size_t)NR * Tia * (1 + C1);
  const size_t l2 = (size_t)NR * (D + D) + DEC_KS * (C2 + 1) + NR * C2 + C2 + 4 + (size_t)FPW * D + DEC_KS * 64;
  return sizeof(float) * (l1 > l2 ? l1 : l2);
}

template <typename C>
size_t fwd_lds_bytes() {
  return sizeof(float) * (C::K + C::D1 + CT + C::A + 256 + 2 * APAD + TSMAX + (CT / 64) * TSMAX + C::UPW + 16 +
                          CG * C::XMAX + TSMAX * (C::A + KPAD) + TSMAX * C::D1 + C::UPW * C::A + (KWMAX + 1) * (C::A + KPAD));
}

bool cluster_shape_ok(const ns_taco2_attn_params* p) {
  if (!p || !p->pv || p->D1 != 256 || p->D2 != 128) return false;
  if (p->A != 256 && p->A != 64) return false;
  if (p->Ti > 256 || p->Tia > 256 || p->kw > KWMAX || p->S < 1 || p->N < 1) return false;
  if (p->Dsp < 0) return false;
  if (CG > ns_device_cus()) return false;      // ONE utterance's cluster of CG workgroups must be resident at once; the clusters
                                               // are independent chains (more of them than CUs / CG just take turns)
  return true;
}
}  // namespace

extern "C" int ns_taco2_attn_cluster_supported(const ns_taco2_attn_params* p) { return cluster_shape_ok(p) ? 1 : 0; }

extern "C" size_t ns_taco2_attn_cluster_work_bytes(const ns_taco2_attn_params* p) {
  if (!p) return 0;
  // status block + the two exchange buffers (sized for the widest instantiation)
  return 256 + sizeof(u64) * (size_t)p->N * CG * (size_t)(2 * (256 + 32 + 2 + TSMAX) + 1024) + TRACE_BYTES;
}

template <typename T, typename C>
static int launch_fwd(const ns_taco2_attn_params* p, void* work, hipStream_t s) {
  ACArgs a;
  a.p = *p;
  a.status = (int*)work;
  a.x2 = (u64*)((char*)work + 256);
  a.x3 = a.x2 + (size_t)p->N * CG * C::X2N;
  a.x1 = nullptr;
  const size_t xbytes = sizeof(u64) * (size_t)p->N * CG * (C::X2N + C::X3N);
  a.trace = getenv("NS_ATTN_TRACE") ? (long long*)((char*)work + ns_taco2_attn_cluster_work_bytes(p) - TRACE_BYTES) : nullptr;
  { const int zrc = ns_zero_async(work, ((256 + xbytes) + 15) & ~(size_t)15, s); if (zrc) return zrc; }
  const size_t lds = fwd_lds_bytes<C>();
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_cluster_fwd_kernel<T, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  // chunks of at most 32 utterances: one workgroup per CU, every cluster of a launch resident at once
  for (int n0 = 0; n0 < p->N; n0 += 32) {
    ACArgs b = a;
    const int nn = min(32, p->N - n0);
    b.p.N = nn;
    const long S1 = p->S + 1;
    const long esz = sizeof(T);
    // shift every per-utterance base by n0 utterances
    b.p.lengths = p->lengths ? p->lengths + n0 : nullptr;
    b.p.keys = p->keys + (long)n0 * p->Pi * p->A;
    b.p.pv = (const char*)p->pv + (long)n0 * p->Pi * p->D1 * esz;
    b.p.f1 = p->f1 + (long)n0 * S1 * p->D1;
    b.p.p1 = (char*)p->p1 + (long)n0 * S1 * p->D1 * esz;
    b.p.xa = (char*)p->xa + (long)n0 * S1 * (p->D2 + p->Dsp + p->A) * esz;
    b.p.hc = (char*)p->hc + (long)n0 * S1 * (p->A + p->E) * esz;
    b.p.ca = p->ca + (long)n0 * S1 * p->A;
    b.p.ga = (char*)p->ga + (long)n0 * S1 * 4 * p->A * esz;
    b.p.q = p->q + (long)n0 * S1 * p->A;
    b.p.align = p->align + (long)n0 * S1 * p->Tia;
    b.p.align_t = p->align_t ? (char*)p->align_t + (long)n0 * S1 * p->Tia * esz : nullptr;
    b.x2 = a.x2 + (size_t)n0 * CG * C::X2N;
    b.x3 = a.x3 + (size_t)n0 * CG * C::X3N;
    hipLaunchKernelGGL((attn_cluster_fwd_kernel<T, C>), dim3(nn * CG), dim3(CT), lds, s, b);
  }
  NS_CHECK_LAUNCH("attn_cluster_fwd");
  // the transposed keys the per-step forward leaves behind for either backward path (hoisted sums over the steps)
  if (p->keys_t) {
    int rc = ns_taco2_keys_transpose(p->keys, p->keys_t, p->N, p->Ti, p->Tia, p->Pi, p->padl_i, p->A, s);
    if (rc) return rc;
  }
  return ns_attn_contexts_after_loop(*p, s);
}

// Forward of the whole attention RNN in one launch (per 32 utterances).  Same parameter block and the same outputs
// as ns_taco2_attn_fwd in its projected-memory form (the contexts hc[:, :, A:] come from the same batched product
// align . memory after the loop).  Reads the natural-layout weights w2 / watt / wq (dtype) instead of the
// k-contiguous shadows, and needs neither keys_t nor the per-step work buffer.  `work`: ns_taco2_attn_cluster_work_bytes; work[0] (int) is the status word.
extern "C" int ns_taco2_attn_cluster_fwd(const ns_taco2_attn_params* p, void* work, ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  NS_CHECK_ARG(p && work, "ns_taco2_attn_cluster_fwd: null");
  NS_CHECK_ARG(cluster_shape_ok(p), "ns_taco2_attn_cluster_fwd: unsupported shape (needs pv, D1 256, D2 128, A 64|256, T_in <= 256)");
  NS_CHECK_ARG(p->keys && p->f1 && p->w2 && p->watt && p->wq && p->b2 && p->batt && p->wcl && p->v && p->p1 && p->xa &&
                   p->hc && p->ca && p->ga && p->q && p->align && p->align_t && p->values, "ns_taco2_attn_cluster_fwd: null pointer");
  if (p->dtype == NS_BF16) {
    if (p->A == 256) return launch_fwd<bf16_t, Cfg<256, 256, 128>>(p, work, s);
    return launch_fwd<bf16_t, Cfg<64, 256, 128>>(p, work, s);
  }
  if (p->A == 256) return launch_fwd<float, Cfg<256, 256, 128>>(p, work, s);
  return launch_fwd<float, Cfg<64, 256, 128>>(p, work, s);
}

template <typename C>
static size_t bwd_lds_bytes() {
  return sizeof(float) * (C::D1 + C::D2 + C::GC + C::NQ * C::K + C::A + 256 + 2 * APAD + 5 * TSMAX + TSMAX * 8 + 8 * TSMAX * 8 +
                          C::UPW + C::A + C::D1 + C::D2 + C::GC + 48 + CG * C::EMAX + TSMAX * (C::A + KPAD) + TSMAX * C::D1 +
                          C::UPW * C::A + (KWMAX + 1) * (C::A + KPAD) +
                          C::A + 256 + 2 * APAD + 2 * TSMAX + C::D1 + C::D2 + C::GC);      // second set of history images
}

template <typename T, typename C>
static int launch_bwd(const ns_taco2_attn_params* p, void* work, hipStream_t s) {
  ACArgs a;
  a.p = *p;
  a.status = (int*)work;
  a.x2 = (u64*)((char*)work + 256);
  a.x3 = a.x2 + (size_t)p->N * CG * C::A;
  a.x1 = a.x3 + (size_t)p->N * CG * C::E3N;
  const size_t xbytes = sizeof(u64) * (size_t)p->N * CG * (C::A + C::E3N + 1);
  a.trace = getenv("NS_ATTN_TRACE") ? (long long*)((char*)work + ns_taco2_attn_cluster_work_bytes(p) - TRACE_BYTES) : nullptr;
  { const int zrc = ns_zero_async(work, ((256 + xbytes) + 15) & ~(size_t)15, s); if (zrc) return zrc; }
  const size_t lds = bwd_lds_bytes<C>();
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_cluster_bwd_kernel<T, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const long S1 = p->S + 1, esz = sizeof(T);
  for (int n0 = 0; n0 < p->N; n0 += 32) {
    ACArgs b = a;
    b.p.N = min(32, p->N - n0);
    b.p.lengths = p->lengths ? p->lengths + n0 : nullptr;
    b.p.keys = p->keys + (long)n0 * p->Pi * p->A;
    b.p.pv = (const char*)p->pv + (long)n0 * p->Pi * p->D1 * esz;
    b.p.p1 = (char*)p->p1 + (long)n0 * S1 * p->D1 * esz;
    b.p.xa = (char*)p->xa + (long)n0 * S1 * (p->D2 + p->Dsp + p->A) * esz;
    b.p.ca = p->ca + (long)n0 * S1 * p->A;
    b.p.ga = (char*)p->ga + (long)n0 * S1 * 4 * p->A * esz;
    b.p.q = p->q + (long)n0 * S1 * p->A;
    b.p.align = p->align + (long)n0 * S1 * p->Tia;
    b.p.da0 = p->da0 + (long)n0 * S1 * p->Tia;
    b.p.dhc = p->dhc + (long)n0 * S1 * (p->A + p->E);
    b.p.de = p->de + (long)n0 * S1 * p->Tia;
    b.p.df1 = (char*)p->df1 + (long)n0 * S1 * p->D1 * esz;
    b.p.dp2 = (char*)p->dp2 + (long)n0 * S1 * p->D2 * esz;
    b.p.dga = (char*)p->dga + (long)n0 * S1 * 4 * p->A * esz;
    b.p.dga_bf16 = p->dga_bf16 ? (char*)p->dga_bf16 + (long)n0 * S1 * 4 * p->A * 2 : nullptr;
    b.p.dq = (char*)p->dq + (long)n0 * S1 * p->A * esz;
    b.x2 = a.x2 + (size_t)n0 * CG * C::A;
    b.x3 = a.x3 + (size_t)n0 * CG * C::E3N;
    b.x1 = a.x1 + (size_t)n0 * CG;
    hipLaunchKernelGGL((attn_cluster_bwd_kernel<T, C>), dim3(b.p.N * CG), dim3(CT), lds, s, b);
  }
  NS_CHECK_LAUNCH("attn_cluster_bwd");
  return ns_attn_bwd_post<T>(*p, s);       // keys_t: written by either forward
}

extern "C" int ns_taco2_attn_cluster_bwd(const ns_taco2_attn_params* p, void* work, ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  NS_CHECK_ARG(p && work, "ns_taco2_attn_cluster_bwd: null");
  NS_CHECK_ARG(cluster_shape_ok(p), "ns_taco2_attn_cluster_bwd: unsupported shape (needs pv, D1 256, D2 128, A 64|256, T_in <= 256)");
  NS_CHECK_ARG(p->keys && p->values && p->w1c && p->w2 && p->watt && p->wq && p->wcl && p->v && p->p1 && p->xa && p->ca &&
                   p->ga && p->q && p->align && p->dhc && p->df1 && p->dp2 && p->dga && p->dq && p->dkeys && p->dvalues &&
                   p->dv && p->dwcl && p->work && p->align_t && p->de && p->dctx_t && p->keys_t && p->da0,
               "ns_taco2_attn_cluster_bwd: null pointer");
  if (p->dtype == NS_BF16) {
    if (p->A == 256) return launch_bwd<bf16_t, BCfg<256, 256, 128>>(p, work, s);
    return launch_bwd<bf16_t, BCfg<64, 256, 128>>(p, work, s);
  }
  if (p->A == 256) return launch_bwd<float, BCfg<256, 256, 128>>(p, work, s);
  return launch_bwd<float, BCfg<64, 256, 128>>(p, work, s);
}

// ------------------------------------------------------------------ free-running decode, C ABI
static bool decode_shape_ok(const ns_taco2_decode_params* q) {
  if (!q) return false;
  const ns_taco2_attn_params* p = &q->att;
  if (!cluster_shape_ok(p)) return false;
  if (!((p->A == 256 && p->E == 512 && q->D == 1024) || (p->A == 64 && p->E == 64 && q->D == 64))) return false;
  if (p->N < 1 || p->N > 2 || p->Tia > 256 || p->Ti > p->Tia) return false;
  const int NW1 = (q->D + DEC_UPW1 - 1) / DEC_UPW1, NW2 = (q->D + DEC_UPW2 - 1) / DEC_UPW2;
  if (p->N * CG + NW1 + NW2 > ns_device_cus()) return false;       // every workgroup resident at once, one per CU
  if ((256 + NW2 - 1) / NW2 > 64) return false;        // feedback columns per LSTM-2 workgroup (red2 rows)
  return true;
}
extern "C" int ns_taco2_decode_supported(const ns_taco2_decode_params* q) { return decode_shape_ok(q) ? 1 : 0; }
extern "C" size_t ns_taco2_decode_work_bytes(const ns_taco2_decode_params* q) {
  if (!q) return 0;
  const ns_taco2_attn_params* p = &q->att;
  return ns_taco2_attn_cluster_work_bytes(p) + sizeof(u64) * ((size_t)p->N * (p->D1 + 256) + 4 * (size_t)p->N * q->D) + 256;
}

template <typename T, typename C, int E, int D>
static int launch_decode(const ns_taco2_decode_params* q, void* work, hipStream_t s) {
  const ns_taco2_attn_params* p = &q->att;
  DecArgs d;
  d.att.p = *p;
  d.att.status = (int*)work;
  d.att.x2 = (u64*)((char*)work + 256);
  d.att.x3 = d.att.x2 + (size_t)p->N * CG * C::X2N;
  d.att.x1 = nullptr;
  d.att.f1x = d.att.x3 + (size_t)p->N * CG * C::X3N;
  d.att.alx = d.att.f1x + (size_t)p->N * C::D1;
  d.pub1 = d.att.alx + (size_t)p->N * 256;
  d.pub2 = d.pub1 + 2 * (size_t)p->N * D;
  const size_t xbytes = (size_t)((char*)(d.pub2 + 2 * (size_t)p->N * D) - (char*)work);
  d.att.trace = nullptr;
  d.N = p->N; d.S = p->S; d.E = E; d.D = D;
  d.NW1 = (D + DEC_UPW1 - 1) / DEC_UPW1; d.NW2 = (D + DEC_UPW2 - 1) / DEC_UPW2;
  d.w1 = q->w_l1; d.b1 = q->b_l1; d.w2 = q->w_l2; d.b2 = q->b_l2; d.wpf = q->wpf; d.bpf = q->bpf;
  d.h2hist = q->h2;
  { const int zrc = ns_zero_async(work, (xbytes + 15) & ~(size_t)15, s); if (zrc) return zrc; }
  const int grid = p->N * CG + d.NW1 + d.NW2;
#define NS_LAUNCH_DEC(NR_)                                                                                           \
  do {                                                                                                               \
    size_t lds = dec_lds_bytes<C, E, D, NR_>(p->Tia, d.NW2);                                                         \
    const size_t la = fwd_lds_bytes<C>();                                                                            \
    if (la > lds) lds = la;                                                                                          \
    NS_CHECK_ARG(lds <= 160 * 1024, "ns_taco2_decode: %zu bytes of LDS needed", lds);                                \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      (void)hipFuncSetAttribute((const void*)taco2_decode_kernel<T, C, E, D, NR_>,                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                             \
      attr = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((taco2_decode_kernel<T, C, E, D, NR_>), dim3(grid), dim3(CT), lds, s, d);                     \
  } while (0)
  if (p->N == 1) NS_LAUNCH_DEC(1); else NS_LAUNCH_DEC(2);
#undef NS_LAUNCH_DEC
  NS_CHECK_LAUNCH("taco2_decode");
  return NS_OK;
}

// The free-running decoder loop (attention RNN + both decoder LSTMs + the frame feedback) as ONE persistent launch,
// see "free-running decode" above.  work: ns_taco2_decode_work_bytes(); work[0] (int) is the status word.
extern "C" int ns_taco2_decode(const ns_taco2_decode_params* q, void* work, ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  NS_CHECK_ARG(q && work, "ns_taco2_decode: null");
  NS_CHECK_ARG(decode_shape_ok(q), "ns_taco2_decode: unsupported shape (needs the attention-cluster shapes, N <= 2, and "
               "(A, E, D) = (256, 512, 1024) or (64, 64, 64))");
  const ns_taco2_attn_params* p = &q->att;
  NS_CHECK_ARG(p->keys && p->f1 && p->w2 && p->watt && p->wq && p->b2 && p->batt && p->wcl && p->v && p->p1 && p->xa &&
                   p->hc && p->ca && p->ga && p->q && p->align && p->values && q->w_l1 && q->b_l1 && q->w_l2 && q->b_l2 &&
                   q->wpf && q->bpf && q->h2, "ns_taco2_decode: null pointer");
  if (p->dtype == NS_BF16) {
    if (p->A == 256) return launch_decode<bf16_t, Cfg<256, 256, 128>, 512, 1024>(q, work, s);
    return launch_decode<bf16_t, Cfg<64, 256, 128>, 64, 64>(q, work, s);
  }
  if (p->A == 256) return launch_decode<float, Cfg<256, 256, 128>, 512, 1024>(q, work, s);
  return launch_decode<float, Cfg<64, 256, 128>, 64, 64>(q, work, s);
}
