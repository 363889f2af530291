// driver.h - host-side sequencing of the batched GP fit on top of the device primitives (prims.h).
//
// Reference behaviour being reproduced: GPDTW1D.fit after the DBA stage (ensembles/models.py:179-220),
// i.e. T x [ closed-form natural-gradient step (gamma) ; Adam step on (variance, lengthscale) ] followed by
// the whitened full-covariance prediction + diag(y_var) and the distrax Cholesky/logdet of the result
// (ensembles/data.py:38-39).  Formulas: SURVEY.md Appendix A.
//
// Every N^3 operation is expressed as recursive blocked Cholesky / triangular solves whose only heavy
// primitive is the batched fp64 GEMM; the recursion bottoms out at 64x64 diagonal blocks that are
// factored and inverted by one workgroup (potrf_leaf) and applied as a GEMM with the inverted block.
#pragma once
#include "prims.h"

namespace BAYENS_NS {

inline int pad_to_leaf(int n) { return ((n + LEAF - 1) / LEAF) * LEAF; }

// Device workspace for a batch of nb members of padded size Np (all pointers device pointers).
struct FitWorkspace {
  int nb = 0, Np = 0;
  Mat W0, Lb, Lam, Cb, T1, T2;     // [nb][Np][Np] each; T1 carries LEAF extra rows ([nb][Np+64][Np]); W0: spare
                                   // (the fit never stores K0; the polar iteration of nuclear_norms uses it)
  real* invL = nullptr;          // [nb][Np/64][64][64]
  real* invC = nullptr;
  real* theta1 = nullptr;        // [nb][Np]
  real* mvec = nullptr;
  real* avec = nullptr;
  real* y = nullptr;             // padded y_mean
  real* nu = nullptr;            // padded y_var (1 in the padded part)
  real* muvec = nullptr;
  real* uvec = nullptr;          // L^T D (L m - y)
  double* ustate = nullptr;        // [nb][8]
  double* dscratch = nullptr;      // [nb][4] double scalars (log-determinant / trace pieces of the loss)
  int* info = nullptr;             // [nb]
  // Lb / Cb (and invL / invC) are INTERLEAVED member by member: Cb.p = Lb.p + Lb.stride / 2, invC = invL + inv_stride() / 2,
  // so that the pair (K, diag(nu) + gamma K) of the first iteration factors as ONE batch of 2 nb members at half the stride
  // (driver.cpp); info_pair: [2 nb] pivot reports of that batch, folded into info.
  int* info_pair = nullptr;
  long inv_step = 0;               // stride between members of invL (and of invC); 0: the compact Np * LEAF
  long inv_stride() const { return inv_step ? inv_step : (long)Np * LEAF; }
  bool paired() const {
    return info_pair && Lb.stride % 2 == 0 && Cb.p == Lb.p + Lb.stride / 2 && Cb.stride == Lb.stride && Cb.ld == Lb.ld &&
           inv_stride() % 2 == 0 && invC == invL + inv_stride() / 2;
  }
};

// Blocked Cholesky (lower, in place) of the n x n matrices A; inv receives the inverted diagonal blocks
// ([leaf][64][64] per member, leaf index relative to A's first row).  A's strictly-upper 64-blocks are
// never touched (callers keep them zero).
// extra_rows (0 or a multiple of 64) rows stored directly below A in the same buffer are carried to E L^-T.
// lower_out (nullable): compact [nb][n_out][n_out] copy of the factor's leading n_out rows / columns, zeros above the diagonal
// (Prims::extract mode 2) - written block column by block column as the columns become final, beside the rest of the factorisation.
void potrf(Prims& P, int nb, int n, Mat A, real* inv, long inv_stride, int* info, int extra_rows = 0,
           real* lower_out = nullptr, int n_out = 0);

// B[m x n] <- B * L^-T  (solve X L^T = B), L n x n lower with inverted diagonal blocks in inv.
void trsm_rlt(Prims& P, int nb, int m, int n, Mat L, const real* inv, long inv_stride, Mat B);
// B[m x n] <- B * L^-1  (solve X L = B).  rhs_lower: B is n x n lower triangular (rows above the
// diagonal block of each column block are zero and are skipped; the result is lower triangular).
void trsm_rln(Prims& P, int nb, int m, int n, Mat L, const real* inv, long inv_stride, Mat B,
              bool rhs_lower);

struct FitParams {
  int N, R, kernel, n_iters;
  double gamma, lr, jitter;
  // optional: -ELBO after the iterations it = 0, 25, 50, ... (models.py:211-215), device [nb][loss_stride]
  double* loss_trace = nullptr;
  int loss_stride = 0;
  // a ragged batch (device arrays [nb], null: uniform): member b has Rv[b] <= R realisations and its [N][Rv[b]] block starts
  // xoff[b] elements into X (ensembles/data.py:385-395 fits members of different realisation counts one after the other)
  const int* Rv = nullptr;
  const long* xoff = nullptr;
};
constexpr int LOSS_EVERY = 25;   // models.py:211

// Runs the whole fit on resident inputs.  X [nb][N][R], y_in/nu_in [nb][N] (compact, device).
// Outputs (device, compact): mu [nb][N], cov [nb][N][N] (nullable: the reduced export), chol (nullable), logdet (nullable, [nb]),
// hyp (nullable, [nb][2]), basis (nullable, [nb][2][N]: L_c^-1 1 and L_c^-1 mu, the two vectors every
// constant-vector log-likelihood of the posterior needs), var (nullable, [nb][N]: diag(cov)).  info: ws.info (device).
void gp_fit(Prims& P, FitWorkspace& ws, const FitParams& fp, const real* X, const real* y_in,
            const real* nu_in, real* mu, real* cov, real* chol, double* logdet, double* hyp,
            real* basis, real* var = nullptr);

// chol/logdet/basis of caller-supplied covariances already placed (padded, full) in ws.T2; mu_pad [nb][Np].
void mvn_factor(Prims& P, FitWorkspace& ws, int N, const real* mu_pad, real* chol, double* logdet,
                real* basis);

// Nuclear norms ||L_j^T L_i||_* for nb pairs sharing L_i (Li: stride-0 Mat) - the trace term of the Gaussian
// W2 distance of ensembles/wasserstein.py:41-45: tr (S_i^1/2 S_j S_i^1/2)^1/2 = sum of singular values of L_j^T L_i.
// Polar factor by the Cholesky form of the QDWH iteration (Nakatsukasa, Bai, Gygi 2010): per step one X^T X,
// one Cholesky and two triangular solves, all on the existing batched primitives; <= max_iters steps, stops when
// every pair moved by less than tol.  Lj: the nb factors L_j (padded, lower).  Uses ws.W0 (M), ws.Lb (X), ws.Cb (Z),
// ws.T1 (T).  nuc: device [nb], scratch_rel: device [nb].  Returns the number of iterations taken, NEGATED when the
// iteration did not converge (a pair still moved by >= tol after max_iters, or a step produced NaN).
// ell0: a lower bound of sigma_min(X_0) (a too large value only costs speed: the stopping test does not use it).
// cap / cap_scale: optional device array [nb] with cap_scale * cap[b] >= ||L_j^T L_i||_2; X_0 is scaled by the smaller of
// it and the Frobenius norm.  tol: the iteration converges cubically and a step moves X by its distance from the polar
// factor, so a relative step d leaves an error of about (sqrt(N) d)^3 / 4 - the default stops at 1e-17 for N = 2000
// without spending a whole step on confirming it.
int nuclear_norms(Prims& P, FitWorkspace& ws, int nb, Mat Li, Mat Lj, double* nuc, double* scratch_rel,
                  double ell0 = 1e-7, double tol = 1e-7, int max_iters = 16, const double* cap = nullptr,
                  double cap_scale = 0.0);
// The iteration alone, on nb products the caller has already placed in the slots of ws.W0 (so that one batch can mix
// pairs of several anchors): nuc[b] = ||W0_b||_*.
// ell_done > 0: also stop once the bound ell carried along from ell0 is within ell_done of 1 (see the definition).
int polar_nuclear(Prims& P, FitWorkspace& ws, int nb, double* nuc, double* scratch_rel, double ell0 = 1e-7,
                  double tol = 1e-7, int max_iters = 16, const double* cap = nullptr, double cap_scale = 0.0,
                  double ell_done = 0.0);

// Algorithmic flop of one member fit (2 flop per FMA, triangular structure exploited).
double fit_flops(int N, int R, int n_iters);

}  // namespace BAYENS_NS
