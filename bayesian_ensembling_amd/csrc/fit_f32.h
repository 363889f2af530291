// fit_f32.h - the bridge between the C ABI (capi.hip, fp64 build of the primitives) and the fp32 build of the same
// driver and kernels (fit_f32.hip, compiled with -DBAYENS_F32 into namespace bayens_f32).  Plain types only.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

struct BayensF32Args {
  int B, N, R, kernel, n_iters;
  double gamma, lr, jitter;
  const float* X;        // device [B][N][R] (ragged: member b's [N][Rv[b]] block xoff[b] elements in)
  const float* y_mean;   // device [B][N]
  const float* y_var;    // device [B][N]
  float* mu;             // device [B][N]
  float* cov;            // device [B][N][N]; null with `var`: the reduced export (fused small-member kernel, N <= 94)
  float* var = nullptr;  // device [B][N] = diag(cov), nullable
  float* chol;           // device, nullable
  double* logdet;        // device [B], nullable
  double* hyp;           // device [B][2], nullable
  double* loss_trace;    // device [B][ceil(n_iters/25)], nullable
  float* basis;          // device [B][2][N], nullable
  int* info_host = nullptr;        // host [B]
  // a ragged batch (device arrays [B], null: uniform R): see FitParams in driver.h
  const int* Rv = nullptr;
  const long* xoff = nullptr;
};

// `side`: look-ahead stream to borrow (the context's; may be null: the instance then creates its own on first use)
void* bayens_f32_create(hipStream_t stream, hipStream_t side);
void bayens_f32_destroy(void* state);
// returns 0 or a HIP error code / negative argument error; message in *err
int bayens_f32_fit(void* state, const BayensF32Args& a, std::string* err);
// per-launch accounting of the fp32 instance (same families as bayens_profile_get)
void bayens_f32_profile_enable(void* state, int on);
void bayens_f32_profile_add(void* state, int family, double* ms, double* flop, long* launches, double* bytes);
