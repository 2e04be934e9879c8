// Argument blocks of the per-step LSTM kernels, shared by lstm.hip (time loops over hoisted
// inputs) and attn.hip (attention RNN inside the Tacotron-2 decoder loop).
#pragma once
#include "common.h"

template <typename T>
struct LstmStep {
  const T* a; long a_sn; int K;      // recurrent operand rows: a + n*a_sn, K elements (null = zeros)
  const T* wT;                       // [4H, K] k-contiguous
  const float* xg; long xg_sn;       // optional per-row addend [4H]
  const float* bias;                 // optional [4H]
  const float* c_prev; long c_sn;    // null = zeros
  T* h_out; long h_sn;
  T* h_out2; long h2_sn;             // optional second destination
  float* c_out; long co_sn;
  T* gates_out; long g_sn;           // optional, post-activation i,j,f,o
  const int* lengths; int t;
  int N, H;
  float forget_bias;
  float cell_clip;                   // > 0: the new cell state is clipped to [-cell_clip, cell_clip] (ns_cell_clip)
  int passes;                        // fp32 operands: 0 exact FMA, 1/3 split-bf16 MFMA
  const bf16_t* wT_hi; const bf16_t* wT_lo;   // optional pre-split copies of an fp32 wT
  // zoneout (ns_lstm_seq_params): zmode 0 = plain cell, 1 = masks (training), 2 = expectation with rates zc / zh
  int zmode; uint32_t zthr_c, zthr_h, zseed_c, zseed_h; float zc, zh;
  const T* hp; long hp_sn;           // h of the previous step, this cell's H columns (null = zeros); read with zmode != 0
};
// up to two independent cells (the two directions of a BiLSTM) in one launch: blockIdx.z
template <typename T>
struct LstmStepPair { LstmStep<T> s[2]; int n; };
template <typename T> int lstm_step_launch(const LstmStep<T>& a, hipStream_t s);
template <typename T> int lstm_step_launch2(const LstmStepPair<T>& a, hipStream_t s);

// Backward step: dh = dh_out (+dh_out2) + dgates_next . Wh^T, then the cell gradient.
template <typename T>
struct LstmBwdStep {
  int N, H, t;
  const int* lengths;
  const T* dg_next; long dgn_sn; int K;   // [N, K] gate gradients of the step after (null = none)
  const T* w;                              // [H, K] rows = this cell's units, k-contiguous
  const float* dh_out; long dho_sn;        // optional
  const float* dh_out2; long dho2_sn;      // optional
  const T* gates; long g_sn;
  const float* c; const float* c_prev; long c_sn;
  float* dc_carry; int first;              // [N,H] in/out (ignored on input when first)
  T* dgates; long dg_sn;                   // out [N, 4H]
  int passes;
  const bf16_t* w_bf16;                    // optional bf16 copy of an fp32 w (passes == 1)
  const bf16_t* dg_next_b; bf16_t* dgates_b;   // optional bf16 copies of dg_next / dgates (same strides)
  // zoneout masks of the forward pass (zmode 1): dh_carry [N,H] in/out carries dh through units that kept h
  int zmode; uint32_t zthr_c, zthr_h, zseed_c, zseed_h;
  float* dh_carry;
};
template <typename T>
struct LstmBwdStepPair { LstmBwdStep<T> s[2]; int n; };
template <typename T> int lstm_bwd_step_launch(const LstmBwdStep<T>& a, hipStream_t s);
template <typename T> int lstm_bwd_step_launch2(const LstmBwdStepPair<T>& a, hipStream_t s);
