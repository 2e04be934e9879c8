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
};
template <typename T> int lstm_step_launch(const LstmStep<T>& a, hipStream_t s);

template <typename T>
struct LstmBwdCell {
  int N, H, t, first;
  const int* lengths;
  const float* dh_out; long dho_sn;    // optional
  const float* dh_out2; long dho2_sn;  // optional
  const float* dh_carry; long dhc_sn;  // optional
  const T* gates; long g_sn;
  const float* c; const float* c_prev; long c_sn;
  float* dc_carry;                     // [N,H] in/out
  T* dgates; long dg_sn;
};
template <typename T> int lstm_bwd_cell_launch(const LstmBwdCell<T>& a, hipStream_t s);
