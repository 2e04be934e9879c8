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
      mag[k] = __powf(__powf(10.f, db * 0.05f), p.power);
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

__global__ void gl_ola_kernel(GlArgs g, int Lout) {
  const ns_griffin_lim_params& p = g.p;
  const int n = blockIdx.y;
  const float* fr = g.fprev + (long)n * p.T * p.win;
  for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < Lout; pos += gridDim.x * blockDim.x)
    p.wav[(long)n * Lout + pos] = ola_gather(fr, p.T, p.hop, p.win, pos);
}

extern "C" size_t ns_griffin_lim_work_bytes(const ns_griffin_lim_params* p) {
  if (!p) return 0;
  const size_t F = p->n_fft / 2 + 1;
  return sizeof(float) * ((size_t)p->N * p->T * F + 2 * (size_t)p->N * p->T * p->win) + 256;
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
  float* fa = p->work + (size_t)p->N * p->T * F;
  float* fb = fa + (size_t)p->N * p->T * p->win;
  const size_t lds = sizeof(float2) * (p->n_fft + p->n_fft / 2);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gl_frame_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  dim3 grid(p->T, p->N);
  g.init = 1; g.fprev = nullptr; g.fnext = fa;
  hipLaunchKernelGGL(gl_frame_kernel, grid, dim3(FT), lds, s, g);
  float* cur = fa; float* nxt = fb;
  for (int it = 0; it < p->iters; ++it) {
    g.init = 0; g.fprev = cur; g.fnext = nxt;
    hipLaunchKernelGGL(gl_frame_kernel, grid, dim3(FT), lds, s, g);
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
  const int Lout = (p->T - 1) * p->hop + p->win;
  g.fprev = cur;
  hipLaunchKernelGGL(gl_ola_kernel, dim3(ceil_div(Lout, 256), p->N), dim3(256), 0, s, g, Lout);
  NS_CHECK_LAUNCH("griffin_lim");
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
