// medium.h - host interface of the fused mid-size fit (medium.hip): one workgroup per member, 94 < N <= MEDIUM_MAX_N,
// matrices in an L2-resident HBM scratch.  fp64 only (the reference precision).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace bayens {
namespace mediumn {

// 16 tiles of 16 per side, two padded rows reserved for the bordered Cholesky of the covariance (1^T, mu^T)
constexpr int MEDIUM_MAX_N = 254;

struct MediumArgs {
  int B, N, R, kernel, n_iters;
  double gamma, lr, jitter;
  const double* X;       // [B][N][R]
  const double* y;       // [B][N]
  const double* nu;      // [B][N]
  double* mu;            // [B][N]
  double* cov;           // [B][N][N] or null
  double* var;           // [B][N] = diag(cov) or null
  double* chol;          // [B][N][N] or null
  double* logdet;        // [B] or null
  double* hyp;           // [B][2] or null
  double* basis;         // [B][2][N] or null
  int* info;             // [B]
  double* scratch;       // medium_grid(B, N, n_cus) * medium_scratch_doubles(N) doubles
  double* loss = nullptr;   // [B][loss_stride]: -ELBO after iterations 0, 25, 50, ... (models.py:211-215) or null
  int loss_stride = 0;
  unsigned long long* stamps = nullptr;   // diagnostic: cycle-counter stamps of workgroup 0 (null in production)
  int NT = 0;               // filled in by medium_fit
  long scratch_stride = 0;  // filled in by medium_fit
  int tune = 0;             // measurement switches (BAYENS_MEDIUM_TUNE), filled in by medium_fit
  // members whose realisation counts differ (ensembles/data.py:385-395 fits them one after the other): Rv[b] realisations of
  // member b, whose [N][Rv[b]] block starts xoff[b] elements into X.  Null: every member has R realisations, X is [B][N][R].
  const int* Rv = nullptr;
  const long* xoff = nullptr;
  long x_elems = 0;         // elements of X in all (0: B N R) - only the diagnostics read it
};

bool medium_fit_supported(int N, int R);
size_t medium_scratch_doubles(int N);     // per resident workgroup
int medium_grid(int B, int N, int n_cus); // workgroups medium_fit launches for a batch of B
// queues the fit of all B members on `stream` (members dealt round-robin over the resident workgroups)
void medium_fit(hipStream_t stream, const MediumArgs& args, int n_cus);

}  // namespace mediumn
}  // namespace bayens
