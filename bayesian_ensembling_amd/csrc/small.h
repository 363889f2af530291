// small.h - host interface of the fused small-member fit (small.hip): one workgroup per member, N <= SMALL_MAX_N.
// Like the other sources of the fit it exists in two precisions (prims.h: `real`, namespace BAYENS_NS).
#pragma once
#include <hip/hip_runtime.h>

#include "prims.h"

namespace BAYENS_NS {
namespace smalln {

// 6 tiles of 16 per side, two padded rows reserved for the bordered Cholesky of the covariance (1^T, mu^T)
constexpr int SMALL_MAX_N = 94;

struct SmallArgs {
  int B, N, R, kernel, n_iters;
  double gamma, lr, jitter;
  const real* X;         // [B][N][R]
  const real* y;         // [B][N]
  const real* nu;        // [B][N]
  real* mu;              // [B][N]
  real* cov;             // [B][N][N] or null (reduced export)
  real* var;             // [B][N] = diag(cov) or null
  real* chol;            // [B][N][N] or null
  double* logdet;        // [B] or null
  double* hyp;           // [B][2] or null
  real* basis;           // [B][2][N] or null
  int* info;             // [B]
  double* loss = nullptr;   // [B][loss_stride]: -ELBO after iterations 0, 25, 50, ... (models.py:211-215) or null
  int loss_stride = 0;
  unsigned long long* stamps = nullptr;   // diagnostic: 32 cycle-counter stamps of workgroup 0 (null in production)
  int tune = 0;                           // measurement switches (BAYENS_SMALL_TUNE)
  // per-workgroup park of the unit-variance kernel values (k1, k2) of an iteration's kernel matrix, so that its hyper-gradient
  // does not evaluate sqrt / exp a second time: small_park_doubles() doubles, device memory (L2-resident: 86 KB per workgroup)
  double* kpark = nullptr;
  // members whose realisation counts differ (ensembles/data.py:385-395 fits them one after the other): Rv[b] realisations of
  // member b, whose [N][Rv[b]] block starts xoff[b] elements into X.  Null: every member has R realisations, X is [B][N][R].
  const int* Rv = nullptr;
  const long* xoff = nullptr;
};

// size of SmallArgs::kpark for a batch of B members of N time points on a device with n_cus compute units
size_t small_park_doubles(int B, int N, int n_cus);

bool small_fit_supported(int N, int R);
// queues the fit of all B members on `stream` (grid = resident workgroups, members dealt round-robin)
void small_fit(hipStream_t stream, const SmallArgs& args, int n_cus);
#ifndef BAYENS_F32
// diagnostic micro-benchmark of the 16x16 diagonal factorisation and of the dependent chains it consists of
void diag_bench(hipStream_t stream, int variant, int waves, int reps, double* out_dev);
#endif

}  // namespace smalln
}  // namespace BAYENS_NS
