// Audio DSP of neural_speech/utils/audio.py on the GPU: STFT -> linear + mel spectrograms, and the
// TF-style Griffin-Lim vocoder.  Transforms are Stockham FFTs that live entirely in LDS (one 256-thread
// workgroup per frame, two ping-pong buffers + the twiddle table): radix-2 over n_fft points for the
// feature kernel, radix-4 over n_fft/2 points (real-input split / merge) for Griffin-Lim, where an
// iteration is ONE kernel: overlap-add gather of the previous iteration's windowed frames -> window ->
// FFT -> phase normalise x magnitude -> inverse FFT -> window.
#include "common.h"

constexpr int FT = 256;  // threads per frame

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// In-LDS forward FFT (e^{-i...}) of N points held in `a`; returns the buffer that holds the result.
// tw[m] = exp(-2*pi*i*m/N) for m < N/2 (in LDS).
__device__ __forceinline__ float2* fft_lds(float2* a, float2* b, const float2* tw, int N, int tid) {
  const int half = N >> 1;
  for (int Ns = 1; Ns < N; Ns <<= 1) {
    const int tstride = half / Ns;
    for (int j = tid; j < half; j += FT) {
      const int k = j & (Ns - 1);
      const float2 v0 = a[j];
      const float2 v1 = cmul(a[j + half], tw[k * tstride]);
      const int j0 = ((j - k) << 1) + k;
      b[j0] = make_float2(v0.x + v1.x, v0.y + v1.y);
      b[j0 + Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
    }
    __syncthreads();
    float2* t = a; a = b; b = t;
  }
  return a;
}

// Radix-4 Stockham FFT of M points (M a power of two; a last radix-2 stage when log2 M is odd), same conventions as
// fft_lds.  twN[m] = exp(-2*pi*i*m/N) for m < N/2 with N = 2M, so exp(-2*pi*i*m/M) = twN[2m] (negated past M/2).
// One radix-4 stage does the work of two radix-2 stages behind ONE barrier and half the LDS traffic.
__device__ __forceinline__ float2 tw_m(const float2* twN, int idx, int M) {
  if (idx < (M >> 1)) return twN[2 * idx];
  const float2 t = twN[2 * idx - M];
  return make_float2(-t.x, -t.y);
}
__device__ __forceinline__ float2* fft_r4_lds(float2* a, float2* b, const float2* twN, int M, int tid) {
  const int q4 = M >> 2;
  int Ns = 1;
  for (; Ns * 4 <= M; Ns <<= 2) {
    const int ts = q4 / Ns;                   // twiddle stride in units of 2*pi/M
    for (int j = tid; j < q4; j += FT) {
      const int k = j & (Ns - 1);
      const float2 v0 = a[j];
      float2 v1 = a[j + q4], v2 = a[j + 2 * q4], v3 = a[j + 3 * q4];
      if (Ns > 1) {
        v1 = cmul(v1, tw_m(twN, k * ts, M));
        v2 = cmul(v2, tw_m(twN, 2 * k * ts, M));
        v3 = cmul(v3, tw_m(twN, 3 * k * ts, M));
      }
      const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y), a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
      const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y);
      const float2 a3 = make_float2(v1.y - v3.y, -(v1.x - v3.x));            // (v1 - v3) * (-i)
      const int j0 = ((j - k) << 2) + k;
      b[j0] = make_float2(a0.x + a2.x, a0.y + a2.y);
      b[j0 + Ns] = make_float2(a1.x + a3.x, a1.y + a3.y);
      b[j0 + 2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
      b[j0 + 3 * Ns] = make_float2(a1.x - a3.x, a1.y - a3.y);
    }
    __syncthreads();
    float2* t = a; a = b; b = t;
  }
  if (Ns < M) {                               // Ns * 2 == M: one radix-2 stage
    const int half = M >> 1;
    for (int j = tid; j < half; j += FT) {
      const int k = j & (Ns - 1);
      const float2 v0 = a[j];
      const float2 v1 = cmul(a[j + half], tw_m(twN, k * (half / Ns), M));
      const int j0 = ((j - k) << 1) + k;
      b[j0] = make_float2(v0.x + v1.x, v0.y + v1.y);
      b[j0 + Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
    }
    __syncthreads();
    float2* t = a; a = b; b = t;
  }
  return a;
}

// ------------------------------------------------------------------ spectrogram + mel
__global__ __launch_bounds__(FT) void spectrogram_kernel(ns_spectrogram_params p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = p.n_fft, F = N / 2 + 1;
  float2* bufa = (float2*)sm;
  float2* bufb = bufa + N;
  float2* tw = bufb + N;
  float* mag = (float*)(tw + N / 2);   // [F]
  const int t = blockIdx.x, tid = threadIdx.x;
  for (int m = tid; m < N / 2; m += FT) tw[m] = ((const float2*)p.twiddle)[m];
  const int lpad = (N - p.win) / 2;
  for (int j = tid; j < N; j += FT) {
    float v = 0.f;
    const int wj = j - lpad;
    if (wj >= 0 && wj < p.win) {
      int idx = t * p.hop + j - N / 2;
      if (idx < 0) idx = -idx;
      if (idx >= p.L) idx = 2 * (p.L - 1) - idx;
      idx = max(0, min(p.L - 1, idx));
      float x = p.wav[idx];
      if (p.preemph != 0.f && idx > 0) x -= p.preemph * p.wav[idx - 1];
      v = x * p.window[wj];
    }
    bufa[j] = make_float2(v, 0.f);
  }
  __syncthreads();
  const float2* X = fft_lds(bufa, bufb, tw, N, tid);
  const float inv_min = 1.f / -p.min_level_db;
  for (int k = tid; k < F; k += FT) {
    const float m = sqrtf(X[k].x * X[k].x + X[k].y * X[k].y);
    mag[k] = m;
    if (p.stft_out) ((float2*)p.stft_out)[(long)t * F + k] = X[k];
    if (p.lin_out) {
      const float db = 20.f * log10f(fmaxf(1e-5f, m)) - p.ref_level_db;
      p.lin_out[(long)t * F + k] = fminf(1.f, fmaxf(0.f, (db - p.min_level_db) * inv_min));
    }
  }
  __syncthreads();
  if (p.mel_out) {
    const int lane = tid & 63, wave = tid >> 6;
    for (int m = wave; m < p.n_mels; m += FT / 64) {
      const float* brow = p.mel_basis + (long)m * F;
      float s = 0.f;
      for (int k = lane; k < F; k += 64) s = fmaf(brow[k], mag[k], s);
      s = wave_sum(s);
      if (lane == 0) {
        const float db = 20.f * log10f(fmaxf(1e-5f, s));
        p.mel_out[(long)t * p.n_mels + m] = fminf(1.f, fmaxf(0.f, (db - p.min_level_db) * inv_min));
      }
    }
  }
}

extern "C" int ns_spectrogram(const ns_spectrogram_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->wav && p->window && p->twiddle, "ns_spectrogram: null");
  NS_CHECK_ARG(p->n_fft >= 64 && (p->n_fft & (p->n_fft - 1)) == 0 && p->n_fft <= 4096,
               "ns_spectrogram: n_fft must be a power of two in [64, 4096]");
  NS_CHECK_ARG(p->win <= p->n_fft && p->L > p->n_fft / 2, "ns_spectrogram: window > n_fft or signal too short");
  NS_CHECK_ARG(!p->mel_out || p->mel_basis, "ns_spectrogram: mel basis missing");
  if (p->T <= 0) return NS_OK;
  const size_t lds = sizeof(float2) * (2 * p->n_fft + p->n_fft / 2) + sizeof(float) * (p->n_fft / 2 + 1);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)spectrogram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(spectrogram_kernel, dim3(p->T), dim3(FT), lds, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("spectrogram");
  return NS_OK;
}

// ------------------------------------------------------------------ Griffin-Lim
struct GlArgs {
  ns_griffin_lim_params p;
  float* mag;          // [N,T,F]
  const float* fprev;  // [N,T,win] windowed inverse-FFT frames of the previous iteration
  float* fnext;
  int init;
};

// overlap-add sum over the (<= win/hop) frames that cover a sample, given the last covering frame f1 and the sample's
// offset inside it
__device__ __forceinline__ float ola_frames(const float* fr, int T, int hop, int win, int f1, int off) {
  if (f1 > T - 1) { off += (f1 - (T - 1)) * hop; f1 = T - 1; }
  float s = 0.f;
  for (int f = f1; f >= 0 && off < win; --f, off += hop) s += fr[(long)f * win + off];
  return s;
}
// y[pos]: the same sum from the absolute sample position
__device__ __forceinline__ float ola_gather(const float* fr, int T, int hop, int win, int pos) {
  int f1 = pos / hop;
  if (f1 > T - 1) f1 = T - 1;
  float s = 0.f;
  for (int f = f1; f >= 0; --f) {
    const int off = pos - f * hop;
    if (off >= win) break;
    s += fr[(long)f * win + off];
  }
  return s;
}

// One Griffin-Lim iteration for one frame.  Both transforms are REAL: the 2M-point real FFT runs as an M-point complex
// FFT of z[m] = x[2m] + i x[2m+1] followed by the split X[k] = ((Z[k] + conj Z[M-k]) - i W^k (Z[k] - conj Z[M-k])) / 2,
// and the inverse as the mirrored merge Zt[k] = (X[k] + conj X[M-k]) + i conj(W^k) (X[k] - conj X[M-k]) followed by an
// M-point FFT of conj(Zt): y[2m] = Re / N, y[2m+1] = -Im / N.  Split, phase normalisation and merge are one pass over
// the bin pairs (k, M-k).  Half-size radix-4 transforms: 5 + 5 barriers per iteration instead of 11 + 11.
__device__ __forceinline__ float2 gl_unit(float2 e, float m) {      // m * e / max(1e-8, |e|)
  const float sc = m / fmaxf(1e-8f, sqrtf(e.x * e.x + e.y * e.y));
  return make_float2(e.x * sc, e.y * sc);
}
__global__ __launch_bounds__(FT) void gl_frame_kernel(GlArgs g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const ns_griffin_lim_params& p = g.p;
  const int N = p.n_fft, M = N >> 1, F = M + 1;
  float2* bufa = (float2*)sm;
  float2* bufb = bufa + M;
  float2* tw = bufb + M;                      // [M] = exp(-2 pi i m / N)
  const int t = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  for (int m = tid; m < M; m += FT) tw[m] = ((const float2*)p.twiddle)[m];
  float* mag = g.mag + ((long)n * p.T + t) * F;
  const float2* Z = nullptr;
  float2* dst = bufa;
  if (g.init) {
    // S = (10^((clip(x)*(-min) + min + ref)/20))^power, zero phase
    const float* sp = p.spec + ((long)n * p.T + t) * F;
    for (int k = tid; k < F; k += FT) {
      const float x = fminf(1.f, fmaxf(0.f, sp[k]));
      const float db = x * -p.min_level_db + p.min_level_db + p.ref_level_db;
      mag[k] = p.raw_magnitude ? sp[k] : __powf(__powf(10.f, db * 0.05f), p.power);
    }
    __syncthreads();                          // mag[] of this frame is re-read below by other threads
  } else {
    const float* fr = g.fprev + (long)n * p.T * p.win;
    const float inv_hop = 1.f / p.hop;
    for (int m = tid; m < M; m += FT) {
      float v0 = 0.f, v1 = 0.f;
      // sample j of frame t lies in frames f = t + j / hop, t + j / hop - 1, ... while the offset stays below win
      // (the quotient of two small integers through a float reciprocal: exact, no integer division in the loop)
      if (2 * m < p.win) {
        const int j0 = 2 * m, j1 = j0 + 1;
        const int q0 = (int)((j0 + 0.5f) * inv_hop), q1 = (int)((j1 + 0.5f) * inv_hop);
        v0 = ola_frames(fr, p.T, p.hop, p.win, t + q0, j0 - q0 * p.hop) * p.window[j0];
        if (j1 < p.win) v1 = ola_frames(fr, p.T, p.hop, p.win, t + q1, j1 - q1 * p.hop) * p.window[j1];
      }
      bufa[m] = make_float2(v0, v1);
    }
    __syncthreads();
    Z = fft_r4_lds(bufa, bufb, tw, M, tid);
    dst = (Z == bufa) ? bufb : bufa;
  }
  // bins k and M-k together: split -> unit phase x magnitude -> merge, conj(Zt) goes to dst
  for (int k = tid; k <= M / 2; k += FT) {
    const int kk = (k == 0) ? 0 : M - k;      // partner bin inside Z (Z[M] = Z[0])
    float2 xa, xb;                            // Xn[k], Xn[M-k]
    const float2 w = (k == 0) ? make_float2(1.f, 0.f) : tw[k];
    if (g.init) {
      xa = make_float2(mag[k], 0.f);
      xb = make_float2(mag[M - k], 0.f);
    } else {
      const float2 A = Z[k], B = Z[kk];
      // X[k] = ((A + conj B) - i w (A - conj B)) / 2 ;  X[M-k] = ((B + conj A) + i conj(w) (B - conj A)) / 2
      const float2 s1 = make_float2(A.x + B.x, A.y - B.y), d1 = make_float2(A.x - B.x, A.y + B.y);
      const float2 wd = cmul(w, d1);
      const float2 ea = make_float2(0.5f * (s1.x + wd.y), 0.5f * (s1.y - wd.x));
      const float2 s2 = make_float2(s1.x, -s1.y), d2 = make_float2(-d1.x, d1.y);
      const float2 wc = cmul(make_float2(w.x, -w.y), d2);
      const float2 eb = make_float2(0.5f * (s2.x - wc.y), 0.5f * (s2.y + wc.x));
      xa = gl_unit(ea, mag[k]);
      xb = gl_unit(eb, mag[M - k]);
    }
    // Zt[k] = (xa + conj xb) + i conj(w) (xa - conj xb) ;  Zt[M-k] = (xb + conj xa) - i w (xb - conj xa)
    const float2 s1 = make_float2(xa.x + xb.x, xa.y - xb.y), d1 = make_float2(xa.x - xb.x, xa.y + xb.y);
    const float2 c1 = cmul(make_float2(w.x, -w.y), d1);
    const float2 zk = make_float2(s1.x - c1.y, s1.y + c1.x);
    dst[k] = make_float2(zk.x, -zk.y);
    if (k != 0 && k != M - k) {
      const float2 s2 = make_float2(s1.x, -s1.y), d2 = make_float2(-d1.x, d1.y);
      const float2 c2 = cmul(w, d2);
      const float2 zm = make_float2(s2.x + c2.y, s2.y - c2.x);
      dst[M - k] = make_float2(zm.x, -zm.y);
    }
  }
  __syncthreads();
  float2* other = (dst == bufa) ? bufb : bufa;
  const float2* Y = fft_r4_lds(dst, other, tw, M, tid);
  const float invN = 1.f / N;
  float* fo = g.fnext + ((long)n * p.T + t) * p.win;
  for (int m = tid; 2 * m < p.win; m += FT) {
    const float2 y = Y[m];
    fo[2 * m] = y.x * invN * p.window[2 * m];
    if (2 * m + 1 < p.win) fo[2 * m + 1] = -y.y * invN * p.window[2 * m + 1];
  }
}

// ------------------------------------------------------------------ Griffin-Lim, n_fft = 2048: one WAVE per frame
// The 1024-point complex transform lives in the registers of one wavefront (16 points per lane) as 16 x 16 x 4:
//   pass 1: lane n' holds z[64 n1 + n'], a 16-point DFT over n1 in registers, twiddle W_1024^(n' k1)
//   one LDS exchange (transpose; the wave's own 8.5 KB, no workgroup barrier: LDS is in order inside a wave)
//   pass 2: lane (k1, n3) holds the 16 values n2 of n' = 4 n2 + n3, 16-point DFT over n2, twiddle W_64^(n3 k2)
//   a second LDS exchange brings the four n3 of a (k1, k2) into one lane, pass 3 = a 4-point DFT over n3 in registers
// which leaves Z[k1 + 16 k2 + 256 k3] in lane (k1, k2 >> 2), register (k2 & 3, k3).  An iteration is: overlap-add gather of the
// previous iteration's frames (4 neighbours, float2 loads) -> window -> FFT -> Z to LDS in natural order -> every lane
// takes the bins 64 n1 + n' (what pass 1 of the next transform wants) and their partners M - k: real-input split, unit
// phase x magnitude, merge -> FFT of conj -> window -> frame out.  5 LDS exchanges and no barrier per iteration where
// the 256-thread kernel above needs 10 barriers; 4 frames per workgroup share the W_1024 table.
constexpr int GL_WPW = 4;                    // frames (waves) per workgroup: 8 KB table + 10 KB per wave, three workgroups = 12 waves per CU
                                             // (7 per workgroup = 14 waves per CU measured slower: 0.83 against 0.71 ms per clip, 8.5 against 7.7 ms)
constexpr int GW_PAD = 1280;                 // float2 per wave: 16 rows of 68 (transposes) / 1024 + 16 per 64 (natural)

// complex values as packed pairs (re, im): an add is one v_pk_add_f32, a product two packed operations
typedef float v2f __attribute__((ext_vector_type(2)));
// -i v = (v.y, -v.x) as ONE instruction of its own with the swap on src0 (left as a vector expression, the compiler folds
// the swap into whichever operand of the consuming instruction it likes - src1 included, see cmulw below)
__device__ __forceinline__ v2f mul_neg_i(v2f v) {
  v2f r;
  const v2f z = {0.f, 0.f};
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(v), "v"(z));
  return r;
}
// Complex arithmetic on packed (re, im) pairs with the swaps, conjugations and sign flips carried by the packed
// instructions' own operand selects (op_sel / op_sel_hi / neg_lo / neg_hi): written from the vector types the compiler
// spends a v_xor (sign), a v_mov (swap) and a wait state per product - 5 issue slots where 2 - 3 do (round 3: 2 426 ->
// ~2 000 instructions per frame and iteration; the results are the same IEEE operations, bit for bit).
// Operand forms: NO instruction here selects the high half of src1 for the low lane (op_sel bit 1).  On MI355X a packed
// fp32 instruction with that bit set misreads the operand about once per million executions while MFMA waves of another
// kernel share the CU (profiles/tools/pk_opsel_probe.hip, profiles/r04_determinism.txt item 4); src0 / src2 selects and
// every op_sel_hi form are not affected, so the swapped operand is always src0 (or src2).
//   x * (w.x + i w.y):  t = (w.y x.x, -w.y x.y);  r = (x.x w.x, x.y w.x) + (t.y, t.x)
__device__ __forceinline__ v2f cmulw(v2f x, v2f w) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %3, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_hi:[0,1]\n\ts_nop 0\n\t"
      "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[0,0,1] op_sel_hi:[1,0,0]"
      : "=&v"(t), "=&v"(r) : "v"(x), "v"(w));
  return r;
}
//   x * conj(w) = x * (w.x - i w.y):  t = (-w.y x.x, w.y x.y)
__device__ __forceinline__ v2f cmulw_conj(v2f x, v2f w) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %3, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1]\n\ts_nop 0\n\t"
      "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[0,0,1] op_sel_hi:[1,0,0]"
      : "=&v"(t), "=&v"(r) : "v"(x), "v"(w));
  return r;
}
__device__ __forceinline__ v2f cmulv(v2f x, float a, float b) { return cmulw(x, (v2f){a, b}); }        // x * (a + i b)
// two independent products interleaved: the packed product with operand selects needs one wait state before its result is
// read, and the other product's instruction fills it
__device__ __forceinline__ void cmulw2(v2f& x0, v2f w0, v2f& x1, v2f w1) {
  v2f t0, t1, r0, r1;
  asm("v_pk_mul_f32 %0, %5, %4 op_sel:[1,0] op_sel_hi:[1,1] neg_hi:[0,1]\n\t"
      "v_pk_mul_f32 %1, %7, %6 op_sel:[1,0] op_sel_hi:[1,1] neg_hi:[0,1]\n\t"
      "v_pk_fma_f32 %2, %4, %5, %0 op_sel:[0,0,1] op_sel_hi:[1,0,0]\n\t"
      "v_pk_fma_f32 %3, %6, %7, %1 op_sel:[0,0,1] op_sel_hi:[1,0,0]"
      : "=&v"(t0), "=&v"(t1), "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(w0), "v"(x1), "v"(w1));
  x0 = r0; x1 = r1;
}
//   a + (u.y, -u.x) = a - i u   and   a - (u.y, -u.x) = a + i u   (u is src0: its halves are the swapped ones)
__device__ __forceinline__ v2f add_mi(v2f a, v2f u) {
  v2f r;
  asm("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(u));
  return r;
}
__device__ __forceinline__ v2f add_pi(v2f a, v2f u) {
  v2f r;
  asm("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(r) : "v"(a), "v"(u));
  return r;
}
//   conj(a + i u) = (a.x - u.y, -a.y - u.x)
__device__ __forceinline__ v2f add_pi_conj(v2f a, v2f u) {
  v2f r;
  asm("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,1]" : "=v"(r) : "v"(a), "v"(u));
  return r;
}
//   a + conj(b)   and   a - conj(b)
__device__ __forceinline__ v2f add_conj(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ v2f sub_conj(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void fft4(v2f& a, v2f& b, v2f& c, v2f& d) {   // in place: X0..X3 of x0..x3
  const v2f t0 = a + c, t1 = a - c, t2 = b + d, u = b - d;
  a = t0 + t2; c = t0 - t2; b = add_mi(t1, u); d = add_pi(t1, u);
}
// 16-point forward DFT in registers, natural order in and out (4 x 4, the digit reversal is register renaming)
__device__ __forceinline__ void fft16(v2f (&x)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, HH = 0.70710678118654752f;
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) fft4(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);
  // element 4 k1 + n2 times W_16^(n2 k1)
  cmulw2(x[5], (v2f){C1, -S1}, x[6], (v2f){HH, -HH});
  cmulw2(x[7], (v2f){S1, -C1}, x[9], (v2f){HH, -HH});
  x[10] = mul_neg_i(x[10]);
  cmulw2(x[11], (v2f){-HH, -HH}, x[13], (v2f){S1, -C1});
  cmulw2(x[14], (v2f){-HH, -HH}, x[15], (v2f){-C1, S1});
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) fft4(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
  // position 4 k1 + k2 holds X[k1 + 4 k2]
  v2f t[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) t[(p >> 2) + 4 * (p & 3)] = x[p];
#pragma unroll
  for (int p = 0; p < 16; ++p) x[p] = t[p];
}
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // s_waitcnt lgkmcnt(0): LDS is in order inside a wave
  __builtin_amdgcn_wave_barrier();
}
// in: lane n' = lane holds z[64 n1 + n'] in x[n1];  out: lane (k1 = lane >> 2, k2hi = lane & 3) holds
// Z[k1 + 16 (4 k2hi + k2lo) + 256 k3] in x[4 k2lo + k3].  tw[j] = exp(-2 pi i j / 1024) (LDS), buf = this wave's GW_PAD complex values.
__device__ __forceinline__ void fft1024_wave(v2f (&x)[16], v2f* buf, const v2f* tw, int lane) {
  fft16(x);
  x[1] = cmulw(x[1], tw[lane]);
#pragma unroll
  for (int k1 = 2; k1 < 16; k1 += 2) cmulw2(x[k1], tw[lane * k1], x[k1 + 1], tw[lane * (k1 + 1)]);
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) buf[k1 * 68 + lane] = x[k1];
  wave_lds_fence();
  const int k1l = lane >> 2, n3 = lane & 3;
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) x[n2] = buf[k1l * 68 + 4 * n2 + n3];
  wave_lds_fence();                                             // the buffer is free again
  fft16(x);
  x[1] = cmulw(x[1], tw[16 * n3]);
#pragma unroll
  for (int k2 = 2; k2 < 16; k2 += 2) cmulw2(x[k2], tw[16 * n3 * k2], x[k2 + 1], tw[16 * n3 * (k2 + 1)]);
  // second exchange: the four n3 of a (k1, k2) come together in one lane - lane (k1 = lane >> 2, k2hi = lane & 3) takes
  // k2 = 4 k2hi + k2lo - and the last radix-4 runs in registers (a DPP quad version of this pass cost 290 VALU
  // instructions per transform where this costs 40 and two dozen LDS instructions)
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) buf[k1l * 68 + k2 * 4 + n3] = x[k2];
  wave_lds_fence();
#pragma unroll
  for (int q = 0; q < 16; ++q) x[q] = buf[k1l * 68 + n3 * 16 + q];          // q = k2lo * 4 + n3'
  wave_lds_fence();
#pragma unroll
  for (int k2lo = 0; k2lo < 4; ++k2lo) fft4(x[4 * k2lo], x[4 * k2lo + 1], x[4 * k2lo + 2], x[4 * k2lo + 3]);
}
// natural order with 16 values of padding per 64: the writers of a transform's output (lane = (k1, k2hi), register
// (k2lo, k3): k = k1 + 64 k2hi + 16 k2lo + 256 k3) and the readers of the bins 64 n1 + lane both stay (nearly) conflict free
__device__ __forceinline__ int gw_nat(int k) { return k + 16 * (k >> 6); }

constexpr int GL_NP = 9;                    // pairs (k, M - k) per lane: k = 64 n1 + lane, n1 < 8; n1 = 8 is lane 0's (512, 512)
// The pairs (k, M - k), k = 64 n1 + lane: split -> unit phase x magnitude -> merge, ONCE per pair for both bins; the two
// results replace Z[k] and Z[M - k] in the wave's natural-order image (no other lane touches those two cells).
// With t1 = Xn[k] + conj Xn[M-k], c1 = conj(w) (Xn[k] - conj Xn[M-k]):  conj Zt[k] = conj(t1 + i c1), conj Zt[M-k] = t1 - i c1.
template <bool INIT>
__device__ __forceinline__ void gl_pairs(v2f* buf, int lane, const float (&mkv)[GL_NP], const float (&mpv)[GL_NP], const v2f (&wv)[GL_NP]) {
#pragma unroll
  for (int n1 = 0; n1 < GL_NP; ++n1) {
    const float mk = mkv[n1], mp = mpv[n1];
    const v2f w = wv[n1];
    // natural-order cells of the two bins: gw_nat(k) = 80 n1 + lane; the partner (M - k) & 1023 (bin 0 for k = 0)
    const int ik = 80 * n1 + lane;
    const int ip = lane == 0 ? (n1 == 0 ? 0 : 80 * (16 - n1)) : (64 - lane) + 80 * (15 - n1);
    v2f xa, xb;
    if constexpr (INIT) {
      xa = (v2f){mk, 0.f}; xb = (v2f){mp, 0.f};
    } else {
      const v2f A = buf[ik], B = buf[ip];
      // X[k] = ((A + conj B) - i w (A - conj B)) / 2 ;  X[M-k] = ((B + conj A) + i conj(w) (B - conj A)) / 2
      // (the factor 1/2 drops out of the unit phase).  With s1 = A + conj B, d1 = A - conj B, wd = w d1:
      //   2 X[k] = s1 - i wd = (s1.x + wd.y, s1.y - wd.x);   2 X[M-k] = (s1.x - wd.y, -s1.y - wd.x)
      const v2f s1 = add_conj(A, B), d1 = sub_conj(A, B);
      const v2f wd = cmulw(d1, w);
      const v2f ea = add_mi(s1, wd);
      const v2f eb = add_pi_conj(s1, wd);
      // m e / max(1e-8, |e|) with e = ea / 2
      const v2f na = ea * ea, nb = eb * eb;
      const float sa = mk * __builtin_amdgcn_rsqf(fmaxf(4e-16f, na.x + na.y));
      const float sb = mp * __builtin_amdgcn_rsqf(fmaxf(4e-16f, nb.x + nb.y));
      xa = ea * (v2f){sa, sa};
      xb = eb * (v2f){sb, sb};
    }
    const v2f t1 = add_conj(xa, xb), u1 = sub_conj(xa, xb);
    const v2f c1 = cmulw_conj(u1, w);                           // conj(w) u1
    if (n1 < 8 || lane == 0) {
      if (!(n1 == 0 && lane == 0)) buf[ip] = add_mi(t1, c1);      // (t1.x + c1.y, t1.y - c1.x): bin M - k (bin M itself does not exist)
      buf[ik] = add_pi_conj(t1, c1);                              // (t1.x - c1.y, -(t1.y + c1.x))
    }
  }
}

// Overlap-add gather of one frame's samples from the previous iteration's windowed frames (at most 4 cover a sample:
// the host checks win <= 4 hop): sample j of frame t lies in frames t + q0, t + q0 - 1, ... at offsets o0, o0 + hop, ...
// = element index e0 + dq (hop - win) of this clip's frames.  Range-checked buffer loads (an out-of-range lane reads 0);
// the caller issues all of them before the first is used.  AUX 16 = sc1: frames written by other CUs inside this launch.
typedef unsigned int gl_u32x2 __attribute__((ext_vector_type(2)));
template <int AUX>
__device__ __forceinline__ void gl_issue_gather(const float* frames_of_clip, int T, int t, int hop, int win, int lane, gl_u32x2 (&xs)[8][4]) {
  const float inv_hop = 1.f / hop;
  const auto frs = __builtin_amdgcn_make_buffer_rsrc((void*)frames_of_clip, 0, T * win * 4, 0x00020000);
#pragma unroll
  for (int n1 = 0; n1 < 8; ++n1) {
    const int j = 2 * (64 * n1 + lane);
    int q0 = (int)((j + 0.5f) * inv_hop);       // the quotient of two small integers through a float reciprocal: exact
    q0 = min(q0, T - 1 - t);
    const int o0 = j - q0 * hop;
    const int e0 = (t + q0) * win + o0;
#pragma unroll
    for (int dq = 0; dq < 4; ++dq) {
      const bool ok = j < win && t + q0 - dq >= 0 && o0 + dq * hop < win;
      xs[n1][dq] = __builtin_amdgcn_raw_buffer_load_b64(frs, ok ? (unsigned)(e0 + dq * (hop - win)) * 4u : 0x80000000u, 0, AUX);
    }
  }
}
__device__ __forceinline__ void gl_window_gathered(const gl_u32x2 (&xs)[8][4], const v2f* winl, int win, int lane, v2f (&z)[16]) {
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    v2f v = {0.f, 0.f};
    if (n1 < 8) {
      const v2f wd = 2 * (64 * n1 + lane) < win ? winl[64 * n1] : (v2f){0.f, 0.f};
      const v2f a0 = {__uint_as_float(xs[n1 & 7][0][0]), __uint_as_float(xs[n1 & 7][0][1])};
      const v2f a1 = {__uint_as_float(xs[n1 & 7][1][0]), __uint_as_float(xs[n1 & 7][1][1])};
      const v2f a2 = {__uint_as_float(xs[n1 & 7][2][0]), __uint_as_float(xs[n1 & 7][2][1])};
      const v2f a3 = {__uint_as_float(xs[n1 & 7][3][0]), __uint_as_float(xs[n1 & 7][3][1])};
      v = ((a0 + a1) + (a2 + a3)) * wd;
    }
    z[n1] = v;
  }
}

template <bool INIT>
__global__ __launch_bounds__(GL_WPW * 64) void gl_wave_kernel(GlArgs g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const ns_griffin_lim_params& p = g.p;
  constexpr int M = 1024, N = 2048, F = M + 1;
  v2f* tw = (v2f*)sm;                                          // [1024] W_1024^j
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  v2f* buf = tw + M + wave * GW_PAD;
  // the W_1024 table is built AFTER this wave's operand loads are in flight (one latency instead of two); a wave
  // beyond the last frame walks the last frame's loads up to the barrier and leaves
  auto build_table = [&]() {
    for (int j = tid; j < M; j += GL_WPW * 64) {
      const v2f v = ((const v2f*)p.twiddle)[j < 512 ? 2 * j : 2 * j - M];
      tw[j] = j < 512 ? v : -v;
    }
    __syncthreads();
  };
  const int traw = blockIdx.x * GL_WPW + wave, n = blockIdx.y;
  const int t = min(traw, p.T - 1);
  float* mag = g.mag + ((long)n * p.T + t) * F;
  v2f z[16];
  constexpr int NP = GL_NP;
  float mkv[NP], mpv[NP];
  v2f wv[NP];
  // per-lane base pointers: every access below is base[constant]
  const v2f* tw2l = (const v2f*)p.twiddle + lane;              // exp(-2 pi i k / 2048), k = 64 n1 + lane
  const float* magk = mag + lane;                              // mag[64 n1 + lane]
  const float* magp = mag + (M - lane);                        // mag[M - k] = magp[-64 n1]
  const v2f* winl = (const v2f*)p.window + lane;               // window[2 (64 n1 + lane)], [.. + 1]
#pragma unroll
  for (int n1 = 0; n1 < NP; ++n1) wv[n1] = tw2l[64 * n1];
  if constexpr (!INIT) {
    // ---- overlap-add gather (the previous iteration's windowed frames) -> window -> z[m] = (x[2m], x[2m+1]).
    //      Sample j of frame t lies in frames t + q0, t + q0 - 1, ... (at most 4: the host checks win <= 4 hop) at
    //      offsets o0, o0 + hop, ...: element index e0 + dq (hop - win) of this clip's frames.  Range-checked buffer
    //      loads (an out-of-range lane reads 0) and every load in flight before the first is used.
    const int win = p.win;
    gl_u32x2 xs[8][4];
    gl_issue_gather<0>(g.fprev + (long)n * p.T * win, p.T, t, p.hop, win, lane, xs);
#pragma unroll
    for (int n1 = 0; n1 < NP; ++n1) { mkv[n1] = magk[64 * n1]; mpv[n1] = magp[-64 * n1]; }
    build_table();
    if (traw >= p.T) return;
    gl_window_gathered(xs, winl, win, lane, z);
    fft1024_wave(z, buf, tw, lane);
    v2f* nat = buf + ((lane >> 2) + 80 * (lane & 3));          // gw_nat(k1 + 64 k2hi + 16 k2lo + 256 k3)
#pragma unroll
    for (int q = 0; q < 16; ++q) nat[16 * (q >> 2) + 320 * (q & 3)] = z[q];
    wave_lds_fence();
  } else {
    build_table();
    if (traw >= p.T) return;
    // S = (10^((clip(x)*(-min) + min + ref)/20))^power, zero phase
    const float* sp = p.spec + ((long)n * p.T + t) * F;
#pragma unroll
    for (int n1 = 0; n1 < NP; ++n1) {
      const int k = 64 * n1 + lane;
      const float x0 = fminf(1.f, fmaxf(0.f, sp[k])), x1 = fminf(1.f, fmaxf(0.f, sp[M - k]));
      mkv[n1] = p.raw_magnitude ? sp[k] : __powf(__powf(10.f, (x0 * -p.min_level_db + p.min_level_db + p.ref_level_db) * 0.05f), p.power);
      mpv[n1] = p.raw_magnitude ? sp[M - k] : __powf(__powf(10.f, (x1 * -p.min_level_db + p.min_level_db + p.ref_level_db) * 0.05f), p.power);
      if (n1 < 8 || lane == 0) { mag[k] = mkv[n1]; mag[M - k] = mpv[n1]; }
    }
  }
  // ---- the pairs (k, M - k): split -> unit phase x magnitude -> merge; then every lane reads its pass-1 inputs
  //      conj(Zt[64 n1 + lane])
  gl_pairs<INIT>(buf, lane, mkv, mpv, wv);
  wave_lds_fence();
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) z[n1] = buf[80 * n1 + lane];
  wave_lds_fence();                                             // every lane has read Z before the buffer is reused
  fft1024_wave(z, buf, tw, lane);
  // ---- y[2m] = Re Y[m] / N, y[2m+1] = -Im Y[m] / N, windowed; m = k1 + 16 k2 + 256 k3
  const float invN = 1.f / N;
  const int m0 = (lane >> 2) + 64 * (lane & 3);                // m = k1 + 64 k2hi + 16 k2lo + 256 k3
  v2f* fo = (v2f*)(g.fnext + ((long)n * p.T + t) * p.win) + m0;
  const v2f* wo = (const v2f*)p.window + m0;
  const v2f sc = {invN, -invN};
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int dm = 16 * (q >> 2) + 256 * (q & 3);
    if (2 * (m0 + dm) < p.win) fo[dm] = z[q] * sc * wo[dm];
  }
}

// ------------------------------------------------------------------ one launch for a whole clip: measured and dropped (round 3)
// A 10 s clip is 797 frames = 200 workgroups of four waves, 61 dependent launches of ~12 us.  A persistent form - every
// wave keeps its frame for all iterations, the overlap-add dependency (frames t-3 .. t+3) carried across the chip
// inside the launch - was built twice and timed with in-kernel stamps (profiles/r03_griffin_lim_trace.txt):
//   per-frame counters behind drained write-through stores: 12.6 us per iteration (0.75 ms per call, as the 61 launches):
//       transforms 1.5 + 2.0, pair pass 1.65, gather 1.1, frame stores + drain + counter 4.4, counter wait 1.9;
//   frames as {iteration tag, sample} granules, the gather's loads repeated until every tag is new (no drain, no
//       counter): 15.3 us (0.95 ms): the 8 write-through store instructions alone take 4.3 us to issue, and a gather
//       pass is 26 MB of sc1 loads chip-wide (2.8 us, bound by the fabric; the samples are read four times over).
// The iteration is NOT launch bound: a lone wave per SIMD spends 5.1 us in its two transforms and the pair pass with
// nothing to hide the LDS round trips, and what crosses CUs inside a launch must bypass the L2s, which costs as much
// as the launch boundary it replaces.  The launch-per-iteration form stays; the next step for a single clip is more
// lanes per frame (two waves, 8 points per lane), not fewer launches.

__global__ void gl_ola_kernel(GlArgs g, int Lout) {
  const ns_griffin_lim_params& p = g.p;
  const int n = blockIdx.y;
  const float* fr = g.fprev + (long)n * p.T * p.win;
  for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < Lout; pos += gridDim.x * blockDim.x)
    p.wav[(long)n * Lout + pos] = ola_gather(fr, p.T, p.hop, p.win, pos);
}

// n_fft 2048 with even hop / window (the shipped hparams): the wave-per-frame kernels
static bool gl_wave_ok(const ns_griffin_lim_params* p) {
  return p->n_fft == 2048 && (p->hop & 1) == 0 && (p->win & 1) == 0 && p->win <= 4 * p->hop && p->win <= 1024 &&
         (((uintptr_t)p->window) & 7) == 0;
}
extern "C" size_t ns_griffin_lim_work_bytes(const ns_griffin_lim_params* p) {
  if (!p) return 0;
  const size_t F = p->n_fft / 2 + 1;
  return sizeof(float) * ((((size_t)p->N * p->T * F + 1) & ~(size_t)1) + 2 * (size_t)p->N * p->T * p->win) + 256;
}

extern "C" int ns_griffin_lim(const ns_griffin_lim_params* p, ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  NS_CHECK_ARG(p && p->spec && p->window && p->twiddle && p->wav && p->work, "ns_griffin_lim: null");
  NS_CHECK_ARG(p->n_fft >= 64 && (p->n_fft & (p->n_fft - 1)) == 0 && p->n_fft <= 4096,
               "ns_griffin_lim: n_fft must be a power of two in [64, 4096]");
  NS_CHECK_ARG(p->win <= p->n_fft && p->hop > 0 && p->hop <= p->win, "ns_griffin_lim: bad window / hop");
  if (p->N <= 0 || p->T <= 0) return NS_OK;
  const size_t F = p->n_fft / 2 + 1;
  GlArgs g;
  g.p = *p;
  g.mag = p->work;
  float* fa = p->work + (((size_t)p->N * p->T * F + 1) & ~(size_t)1);     // 8-byte aligned frames (float2 accesses)
  float* fb = fa + (size_t)p->N * p->T * p->win;
  const size_t lds = sizeof(float2) * (p->n_fft + p->n_fft / 2);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gl_frame_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gl_wave_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gl_wave_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  // n_fft 2048 with even hop / window (the shipped hparams): the wave-per-frame kernel; anything else: one workgroup
  // per frame with the transforms in LDS
  const bool wavek = gl_wave_ok(p) && (((uintptr_t)p->work) & 7) == 0;
  const size_t lds_w = sizeof(float2) * (1024 + GL_WPW * GW_PAD);
  dim3 grid(p->T, p->N), grid_w(ceil_div(p->T, GL_WPW), p->N);
  const int Lout = (p->T - 1) * p->hop + p->win;
  auto launch = [&]() {
    if (wavek && g.init) hipLaunchKernelGGL(gl_wave_kernel<true>, grid_w, dim3(GL_WPW * 64), lds_w, s, g);
    else if (wavek) hipLaunchKernelGGL(gl_wave_kernel<false>, grid_w, dim3(GL_WPW * 64), lds_w, s, g);
    else hipLaunchKernelGGL(gl_frame_kernel, grid, dim3(FT), lds, s, g);
  };
  g.init = 1; g.fprev = nullptr; g.fnext = fa;
  launch();
  float* cur = fa; float* nxt = fb;
  for (int it = 0; it < p->iters; ++it) {
    g.init = 0; g.fprev = cur; g.fnext = nxt;
    launch();
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
  g.fprev = cur;
  hipLaunchKernelGGL(gl_ola_kernel, dim3(ceil_div(Lout, 256), p->N), dim3(256), 0, s, g, Lout);
  NS_CHECK_LAUNCH("griffin_lim");
  return NS_OK;
}

// ------------------------------------------------------------------ the transforms as calls of their own
__global__ __launch_bounds__(FT) void stft_tf_kernel(ns_stft_tf_params p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = p.n_fft, F = N / 2 + 1;
  float2* bufa = (float2*)sm;
  float2* bufb = bufa + N;
  float2* tw = bufb + N;
  const int t = blockIdx.x, tid = threadIdx.x;
  for (int m = tid; m < N / 2; m += FT) tw[m] = ((const float2*)p.twiddle)[m];
  for (int j = tid; j < N; j += FT) bufa[j] = make_float2(j < p.win ? p.wav[(long)t * p.hop + j] * p.window[j] : 0.f, 0.f);
  __syncthreads();
  const float2* X = fft_lds(bufa, bufb, tw, N, tid);
  for (int k = tid; k < F; k += FT) ((float2*)p.out)[(long)t * F + k] = X[k];
}
extern "C" int ns_stft_tf(const ns_stft_tf_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->wav && p->window && p->twiddle && p->out, "ns_stft_tf: null");
  NS_CHECK_ARG(p->n_fft >= 64 && (p->n_fft & (p->n_fft - 1)) == 0 && p->n_fft <= 4096, "ns_stft_tf: n_fft must be a power of two in [64, 4096]");
  NS_CHECK_ARG(p->win <= p->n_fft && p->hop > 0 && p->T >= 0 && (p->T == 0 || (long)(p->T - 1) * p->hop + p->win <= p->L),
               "ns_stft_tf: the last frame ends beyond the signal");
  if (p->T <= 0) return NS_OK;
  const size_t lds = sizeof(float2) * (2 * p->n_fft + p->n_fft / 2);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)stft_tf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(stft_tf_kernel, dim3(p->T), dim3(FT), lds, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("stft_tf");
  return NS_OK;
}

// frame t: Hermitian extension of the one-sided spectrum, inverse transform as conj(FFT(conj X)) / N, the window
__global__ __launch_bounds__(FT) void istft_frame_kernel(ns_istft_params p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = p.n_fft, F = N / 2 + 1;
  float2* bufa = (float2*)sm;
  float2* bufb = bufa + N;
  float2* tw = bufb + N;
  const int t = blockIdx.x, tid = threadIdx.x;
  for (int m = tid; m < N / 2; m += FT) tw[m] = ((const float2*)p.twiddle)[m];
  const float2* X = (const float2*)p.spec + (long)t * F;
  for (int k = tid; k < N; k += FT) {
    float2 v;
    if (k < F) v = X[k]; else { v = X[N - k]; v.y = -v.y; }
    if (k == 0 || k == N / 2) v.y = 0.f;                     // irfft ignores the imaginary part of DC and Nyquist
    bufa[k] = make_float2(v.x, -v.y);
  }
  __syncthreads();
  const float2* Y = fft_lds(bufa, bufb, tw, N, tid);
  const int off = p.center ? (N - p.win) / 2 : 0;
  const float invN = 1.f / N;
  for (int j = tid; j < p.win; j += FT) p.work[(long)t * p.win + j] = Y[off + j].x * invN * p.window[j];
}
__global__ void istft_ola_kernel(ns_istft_params p, int Lout) {
  // segment f starts at f hop (+ the centring offset, which the trim removes again: both are (n_fft - win) / 2 + win / 2 ...)
  const int shift = p.center ? p.n_fft / 2 - (p.n_fft - p.win) / 2 : 0;     // output sample 0 in segment-grid coordinates
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Lout; i += gridDim.x * blockDim.x) {
    const int pos = i + shift;
    int f1 = pos / p.hop;
    if (f1 > p.T - 1) f1 = p.T - 1;
    float s = 0.f, ws = 0.f;
    for (int f = f1; f >= 0; --f) {
      const int o = pos - f * p.hop;
      if (o >= p.win) break;
      s += p.work[(long)f * p.win + o];
      ws += p.window[o] * p.window[o];
    }
    p.wav[i] = (p.center && ws > 1.17549435e-38f) ? s / ws : s;
  }
}
extern "C" int ns_istft(const ns_istft_params* p, ns_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  NS_CHECK_ARG(p && p->spec && p->window && p->twiddle && p->wav && p->work, "ns_istft: null");
  NS_CHECK_ARG(p->n_fft >= 64 && (p->n_fft & (p->n_fft - 1)) == 0 && p->n_fft <= 4096, "ns_istft: n_fft must be a power of two in [64, 4096]");
  NS_CHECK_ARG(p->win <= p->n_fft && p->hop > 0 && p->hop <= p->win, "ns_istft: bad window / hop");
  if (p->T <= 0) return NS_OK;
  const size_t lds = sizeof(float2) * (2 * p->n_fft + p->n_fft / 2);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)istft_frame_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(istft_frame_kernel, dim3(p->T), dim3(FT), lds, s, *p);
  const int Lout = p->center ? (p->T - 1) * p->hop : (p->T - 1) * p->hop + p->win;
  if (Lout > 0) hipLaunchKernelGGL(istft_ola_kernel, dim3(min(1024, ceil_div(Lout, 256))), dim3(256), 0, s, *p, Lout);
  NS_CHECK_LAUNCH("istft");
  return NS_OK;
}

__global__ void audio_pointwise_kernel(ns_audio_pointwise_params p) {
  const float mn = p.min_level_db;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long)gridDim.x * blockDim.x) {
    const float x = p.x[i];
    float y;
    switch (p.mode) {
      case 0: y = 20.f * log10f(fmaxf(1e-5f, x)); break;
      case 1: y = powf(10.f, x * 0.05f); break;
      case 2: y = fminf(1.f, fmaxf(0.f, (x - mn) / -mn)); break;
      default: y = fminf(1.f, fmaxf(0.f, x)) * -mn + mn; break;
    }
    p.y[i] = y;
  }
}
extern "C" int ns_audio_pointwise(const ns_audio_pointwise_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->x && p->y && p->mode >= 0 && p->mode <= 3, "ns_audio_pointwise: null / bad mode");
  if (p->n <= 0) return NS_OK;
  hipLaunchKernelGGL(audio_pointwise_kernel, dim3((int)min((long)4096, (long)((p->n + 255) / 256))), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("audio_pointwise");
  return NS_OK;
}

// ------------------------------------------------------------------ pre-emphasis FIR / IIR
// inverse: y[n] = x[n] + c*y[n-1].  One workgroup walks the signal in 2048-sample tiles; inside a
// tile each thread owns 8 consecutive samples and the (decay, partial) pairs are combined with a
// Hillis-Steele scan of the affine maps, carry handed from tile to tile.
__global__ __launch_bounds__(256) void preemph_kernel(ns_preemphasis_params p) {
  if (!p.inverse) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long)gridDim.x * blockDim.x)
      p.y[i] = p.x[i] - (i > 0 ? p.coef * p.x[i - 1] : 0.f);
    return;
  }
  __shared__ float sa[256], sb[256];
  __shared__ float carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0.f;
  float c8 = 1.f;
  for (int i = 0; i < 8; ++i) c8 *= p.coef;
  for (long base = 0; base < p.n; base += 2048) {
    float loc[8];
    float acc = 0.f;
    const long i0 = base + (long)tid * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xv = (i0 + j < p.n) ? p.x[i0 + j] : 0.f;
      acc = xv + p.coef * acc;
      loc[j] = acc;
    }
    // affine map of this chunk: out = A*in + B with A = c^8, B = acc
    float A = c8, B = acc;
    sa[tid] = A; sb[tid] = B;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      float pa = 1.f, pb = 0.f;
      if (tid >= off) { pa = sa[tid - off]; pb = sb[tid - off]; }
      __syncthreads();
      // compose: apply earlier (pa,pb) first, then (A,B)
      B = A * pb + B;
      A = A * pa;
      sa[tid] = A; sb[tid] = B;
      __syncthreads();
    }
    // state entering this thread's chunk = inclusive scan of the previous thread applied to carry
    const float cin = carry;
    float inS = cin;
    if (tid > 0) inS = sa[tid - 1] * cin + sb[tid - 1];
    float pw = p.coef;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (i0 + j < p.n) p.y[i0 + j] = loc[j] + pw * inS;
      pw *= p.coef;
    }
    __syncthreads();
    if (tid == 255) carry = sa[255] * cin + sb[255];
    __syncthreads();
  }
}

extern "C" int ns_preemphasis(const ns_preemphasis_params* p, ns_stream_t s) {
  NS_CHECK_ARG(p && p->x && p->y, "ns_preemphasis: null");
  if (p->n <= 0) return NS_OK;
  const int grid = p->inverse ? 1 : (int)min((long)1024, (long)((p->n + 255) / 256));
  hipLaunchKernelGGL(preemph_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, *p);
  NS_CHECK_LAUNCH("preemphasis");
  return NS_OK;
}
