// driver.cpp - see driver.h.  Pure host code: no HIP calls here, only Prims launches.
#include "driver.h"
#include <cstdlib>

#include <cmath>
#include <vector>

namespace BAYENS_NS {

constexpr int PANEL_MAX = 256;   // widest diagonal block the fused panel solve takes

// A side section (Prims::side_begin / side_end) that cannot leave the back end on the side stream when a launch
// inside it throws.
namespace {
constexpr int SIDE_PASSES = 1;      // the side channel of the fit's O(N^2) passes (prims.h side_end / side_join)
struct SideSection {
  Prims& P;
  bool open = true;
  int ch;
  explicit SideSection(Prims& p, int channel = 0) : P(p), ch(channel) {
    static const bool no_passes = std::getenv("BAYENS_NO_SIDE_PASSES") != nullptr;      // (measurement switch)
    if (ch == SIDE_PASSES && no_passes) open = false;
    else P.side_begin();
  }
  void end() { if (open) { open = false; P.side_end(ch); } }
  ~SideSection() { if (open) P.side_cancel(); }
};
}  // namespace

static inline int split_point(int n) {
  int leaves = n / LEAF;
  return (leaves / 2) * LEAF;   // >= LEAF whenever leaves >= 2
}

static inline const real* inv_at(const real* inv, int leaf) { return inv + (long)leaf * LEAF * LEAF; }
static inline real* inv_at(real* inv, int leaf) { return inv + (long)leaf * LEAF * LEAF; }

// X L^T = B.  L = [L11 0; L21 L22]  =>  X1 = B1 L11^-T ; B2 -= X1 L21^T ; X2 = B2 L22^-T
void trsm_rlt(Prims& P, int nb, int m, int n, Mat L, const real* inv, long inv_stride, Mat B) {
  if (m <= 0 || n <= 0) return;
  if (n <= PANEL_MAX) {   // the bottom levels of the recursion run fused: one pass over the m x n panel
    P.trsm_panel(nb, m, n, L, inv, inv_stride, B, 0);
    return;
  }
  int n1 = split_point(n), n2 = n - n1;
  trsm_rlt(P, nb, m, n1, L, inv, inv_stride, B);
  P.gemm(nb, m, n2, n1, -1.0, B, L.sub(n1, 0), 1.0, B.sub(0, n1), GEMM_TB, 0.0);
  trsm_rlt(P, nb, m, n2, L.sub(n1, n1), inv_at(inv, n1 / LEAF), inv_stride, B.sub(0, n1));
}

// X L = B.  X2 = B2 L22^-1 ; B1 -= X2 L21 ; X1 = B1 L11^-1
static void trsm_rln_rec(Prims& P, int nb, int m, int n, Mat L, const real* inv, long inv_stride,
                         Mat B, bool rhs_lower, int c0) {
  // B here is the column block [c0, c0+n) of the caller's right-hand side, all m rows.
  // With rhs_lower the rows < c0 of this column block are zero: start at row r0 = c0.
  if (m <= 0 || n <= 0) return;
  int r0 = rhs_lower ? c0 : 0;
  if (r0 >= m) return;
  if (n <= PANEL_MAX) {
    // (with rhs_lower the rows [c0, c0+n) x columns above the diagonal hold zeros and stay zero)
    P.trsm_panel(nb, m - r0, n, L, inv, inv_stride, B.sub(r0, 0), 1);
    return;
  }
  int n1 = split_point(n), n2 = n - n1;
  trsm_rln_rec(P, nb, m, n2, L.sub(n1, n1), inv_at(inv, n1 / LEAF), inv_stride, B.sub(0, n1), rhs_lower,
               c0 + n1);
  int r2 = rhs_lower ? c0 + n1 : 0;   // X2 is zero above row c0+n1 - and lower trapezoidal below it: row i ends at column i
  if (r2 < m)
    P.gemm(nb, m - r2, n1, n2, -1.0, B.sub(r2, n1), L.sub(n1, 0), 1.0, B.sub(r2, 0), rhs_lower ? GEMM_KTRIM_A_LOWER : 0, 0.0);
  trsm_rln_rec(P, nb, m, n1, L, inv, inv_stride, B, rhs_lower, c0);
}

void trsm_rln(Prims& P, int nb, int m, int n, Mat L, const real* inv, long inv_stride, Mat B,
              bool rhs_lower) {
  trsm_rln_rec(P, nb, m, n, L, inv, inv_stride, B, rhs_lower, 0);
}

// Blocked Cholesky with `extra` rows appended below the matrix (rows n .. n+extra-1 of the same buffer).  The
// panel solves and trailing updates that factor A also carry the appended rows E to E L^-T (the bordered-matrix
// identity), which is how L_c^-1 1 and L_c^-1 mu are obtained without a separate triangular solve.
static void potrf_rec(Prims& P, int nb, int n, Mat A, real* inv, long inv_stride, int* info, int base,
                      int extra) {
  if (n == LEAF) {
    P.potrf_leaf(nb, A, inv, inv_stride, base, info);
    if (extra > 0) {
      Mat I{inv, LEAF, inv_stride};
      P.gemm(nb, extra, LEAF, LEAF, 1.0, A.sub(LEAF, 0), I, 0.0, A.sub(LEAF, 0), GEMM_TB, 0.0);
    }
    return;
  }
  if (n == 2 * LEAF && extra == 0) {
    P.potrf_leaf2(nb, A, inv, inv_stride, base, info);
    return;
  }
  if (n == 4 * LEAF && extra == 0 && P.potrf_diag256(nb, A, inv, inv_stride, base, info)) return;   // one fused launch
  int n1 = split_point(n), n2 = n - n1;
  potrf_rec(P, nb, n1, A, inv, inv_stride, info, base, 0);
  // [L21; E1] = [A21; E1] L11^-T
  trsm_rlt(P, nb, n2 + extra, n1, A, inv, inv_stride, A.sub(n1, 0));
  // [A22; E2] -= [L21; E1] L21^T   (lower 64-blocks only; the appended rows lie below every column block)
  P.gemm(nb, n2 + extra, n2, n1, -1.0, A.sub(n1, 0), A.sub(n1, 0), 1.0, A.sub(n1, n1), GEMM_TB | GEMM_TRI_OUT, 0.0);
  potrf_rec(P, nb, n2, A.sub(n1, n1), inv_at(inv, n1 / LEAF), inv_stride, info, base + n1, extra);
}

// Top level: right-looking over POTRF_NB-wide block columns - factor the diagonal block (recursively, it is
// small), solve the whole panel below it in one fused launch, then ONE rank-POTRF_NB update of the entire trailing
// lower part.  The trailing updates are the largest GEMMs the factorisation admits (m x m x NB with m up to n-NB);
// the fully recursive form spends the same flops in many mid-sized products that cannot fill 256 CUs with a
// 32-member batch.
static int potrf_block_width() {
  static int nbw = [] {
    const char* e = std::getenv("BAYENS_POTRF_NB");
    int v = e ? std::atoi(e) : 256;
    return (v >= LEAF && v % LEAF == 0) ? v : 256;
  }();
  return nbw;
}

// Left-looking form of the same factorisation: block column j first receives ALL the updates of the columns before it in ONE
// product, [A_jj; A_below] -= [L_j,0:j; L_below,0:j] L_j,0:j^T - an (n - k0) x NB output contracted over k0 = j NB - then its diagonal
// block is factored and its panel solved.  Same flop as the right-looking form, but the products are long in k (the right-looking
// rank-NB updates read and write the whole trailing triangle per block column and run at 14-50 TFLOP/s, profiles/r04_gemm_shapes.txt:
// half of the dominant kernel's time).  There is no trailing update to run beside the latency-bound diagonal block and panel solve:
// that overlap comes from the other member group's products (fit_in_groups, two groups by default).
static int potrf_left_looking() {
  static int v = [] { const char* e = std::getenv("BAYENS_POTRF_LEFT"); return e ? std::atoi(e) : 0; }();
  return v;
}

static void potrf_left(Prims& P, int nb, int n, Mat A, real* inv, long inv_stride, int* info, int extra_rows) {
  const int NB = potrf_block_width();
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int w = n - k0 < NB ? n - k0 : NB;
    const int below = n - k0 - w;
    real* invk = inv_at(inv, k0 / LEAF);
    const int m = n - k0 + extra_rows;                 // the block column from its diagonal block down, appended rows included
    if (k0 > 0)
      P.gemm(nb, m, w, k0, -1.0, A.sub(k0, 0), A.sub(k0, 0), 1.0, A.sub(k0, k0), GEMM_TB | GEMM_TRI_OUT, 0.0);
    potrf_rec(P, nb, w, A.sub(k0, k0), invk, inv_stride, info, k0, below == 0 ? extra_rows : 0);
    if (below == 0) break;
    trsm_rlt(P, nb, below + extra_rows, w, A.sub(k0, k0), invk, inv_stride, A.sub(k0 + w, k0));
  }
}

void potrf(Prims& P, int nb, int n, Mat A, real* inv, long inv_stride, int* info, int extra_rows, real* lower_out, int n_out) {
  const int NB = potrf_block_width();
  if (n <= NB || potrf_left_looking()) {
    if (n <= NB) potrf_rec(P, nb, n, A, inv, inv_stride, info, 0, extra_rows);
    else potrf_left(P, nb, n, A, inv, inv_stride, info, extra_rows);
    if (lower_out) P.extract(nb, n_out, A, lower_out, 2);
    return;
  }
  // the finished block column [c0, c1) of the factor -> the compact output, on the side queue (a pass bound by HBM, beside
  // products bound by the matrix pipes and a chain bound by latency)
  auto export_cols = [&](int c0, int c1) {
    if (!lower_out || c0 >= n_out) return;
    SideSection side(P, SIDE_PASSES);
    P.extract_cols(nb, n_out, A, lower_out, c0, c1 < n_out ? c1 : n_out);
    side.end();
  };
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int w = n - k0 < NB ? n - k0 : NB;
    const int below = n - k0 - w;
    real* invk = inv_at(inv, k0 / LEAF);
    potrf_rec(P, nb, w, A.sub(k0, k0), invk, inv_stride, info, k0, below == 0 ? extra_rows : 0);
    if (below == 0) {
      export_cols(k0, k0 + w);
      break;
    }
    const int m = below + extra_rows;
    trsm_rlt(P, nb, m, w, A.sub(k0, k0), invk, inv_stride, A.sub(k0 + w, k0));
    // Look-ahead: the next block column is updated first, on the main stream; the rest of the trailing part is
    // updated beside the next diagonal factorisation and panel solve (a few latency-bound workgroups per member),
    // which touch only that next block column.
    const int w2 = below < NB ? below : NB;
    const int c1 = k0 + w;            // first row/column of the trailing part
    P.side_join();                    // the previous step's trailing update wrote the blocks updated next
    P.gemm(nb, m, w2, w, -1.0, A.sub(c1, k0), A.sub(c1, k0), 1.0, A.sub(c1, c1), GEMM_TB | GEMM_TRI_OUT, 0.0);
    if (below > w2) {
      const int c2 = c1 + w2;
      SideSection side(P);            // (restores the main stream if the launch throws)
      P.gemm(nb, m - w2, below - w2, w, -1.0, A.sub(c2, k0), A.sub(c2, k0), 1.0, A.sub(c2, c2),
             GEMM_TB | GEMM_TRI_OUT, 0.0);
      side.end();
    }
    export_cols(k0, c1);              // (behind the trailing update on the side queue: that one is waited for first)
  }
  P.side_join();
  if (lower_out) P.side_join(SIDE_PASSES);
}

// chol(cov) + logdet + the log-likelihood basis (L_c^-1 1, L_c^-1 mu), covariance (lower part valid) in `src`
static void factor_with_basis(Prims& P, FitWorkspace& ws, int N, Mat src, const real* mu_pad, real* chol,
                              double* logdet, real* basis) {
  const int nb = ws.nb, Np = ws.Np;
  // bordering rows (1^T, mu^T): in the padding when the member has at least two padded rows (no extra rows to
  // drag through the factorisation), else in a 64-row block appended below the matrix
  const bool in_pad = Np - N >= 2;
  const int extra = (basis && !in_pad) ? LEAF : 0;
  // The factorisation reads and writes the lower triangle only and `src` is not needed afterwards: factor it where it
  // lies unless the appended rows are needed (only T1 has room for them) - one 2 N^2 pass less.
  const bool in_place = extra == 0 && src.p != ws.T1.p;
  Mat F = in_place ? src : ws.T1;
  if (!in_place) P.tri_copy(nb, Np, src, ws.T1);
  Mat rows = F.sub(in_pad ? N : Np, 0);
  if (basis) P.basis_rows_set(nb, N, Np, rows, mu_pad, in_pad ? 1 : 0);
  static const bool no_col_export = std::getenv("BAYENS_NO_COL_EXPORT") != nullptr;      // (measurement switch)
  potrf(P, nb, Np, F, ws.invL, ws.inv_stride(), ws.info, extra, no_col_export ? nullptr : chol, N);
  if (no_col_export && chol) P.extract(nb, N, F, chol, 2);
  if (logdet) P.logdet(nb, N, F, logdet);
  if (basis) P.basis_rows_get(nb, N, Np, rows, basis);
}

// -ELBO (the training loss the reference samples every 25 iterations, models.py:211-215) at the hypers whose factor
// L is in ws.Lb, with the (m, C) of the last natural-gradient step: GPflow VGP.elbo with the heteroskedastic
// Gaussian expectations of models.py:142-149,
//   sum_t [ log(2 pi)/2 + log(nu_t)/2 + ((y_t - (L m)_t)^2 + (L S L^T)_tt) / (2 nu_t) ] + KL,
//   KL = ( |m|^2 - N - log det S + tr S ) / 2,   S = C^-T C^-1.
// (L S L^T)_tt is the squared norm of row t of L C^-T (one full solve), tr S = ||C^-1||_F^2 (the triangular
// inverse).  Scratch: T1, T2, avec, uvec, dscratch.  The reference evaluates the loss after the Adam step, i.e. at the
// NEXT iteration's hypers: the caller invokes this right after that iteration's (or the prediction's) Cholesky of
// K, before Lam/C are advanced, so no extra kernel matrix or factorisation is needed.
static void neg_elbo(Prims& P, FitWorkspace& ws, const FitParams& fp, int slot, bool c_stale = false) {
  const int nb = ws.nb, Np = ws.Np, N = fp.N;
  const long is = ws.inv_stride();
  if (c_stale) {   // the first iteration took its hyper-gradient without C = chol(Lam) (see gp_fit): factor it for the sample
    P.tri_copy(nb, Np, ws.Lam, ws.Cb);
    potrf(P, nb, Np, ws.Cb, ws.invC, is, ws.info);
  }
  P.resid(nb, Np, ws.Lb, ws.mvec, ws.y, ws.avec);                      // a = L m - y
  P.copy(nb, Np, ws.Lb, ws.T1);
  trsm_rlt(P, nb, Np, Np, ws.Cb, ws.invC, is, ws.T1);                  // L C^-T
  P.row_sqsum(nb, Np, ws.T1, ws.uvec);
  P.identity(nb, Np, ws.T2);
  trsm_rln(P, nb, Np, Np, ws.Cb, ws.invC, is, ws.T2, true);            // C^-1 (lower)
  double* logdetC = ws.dscratch;
  double* trS = ws.dscratch + nb;
  P.logdet(nb, N, ws.Cb, logdetC);
  P.dot_all(nb, Np, ws.T2, ws.T2, trS);
  P.elbo_combine(nb, N, Np, ws.avec, ws.uvec, ws.nu, ws.mvec, ws.Cb, logdetC, trS, fp.loss_trace, fp.loss_stride, slot);
}

void gp_fit(Prims& P, FitWorkspace& ws, const FitParams& fp, const real* X, const real* y_in,
            const real* nu_in, real* mu, real* cov, real* chol, double* logdet, double* hyp,
            real* basis, real* var) {
  const int nb = ws.nb, Np = ws.Np, N = fp.N;
  const long is = ws.inv_stride();
  // (Lam = I is only stored when something reads it: no iteration, or a first iteration in the general form)
  const bool lam_written_first = fp.n_iters >= 2 || (fp.n_iters == 1 && sizeof(real) == sizeof(double));
  P.init_state(nb, N, Np, ws.Lam, ws.theta1, ws.mvec, ws.ustate, y_in, nu_in, ws.y, ws.nu, ws.info,
               !(fp.gamma > 0.0 && lam_written_first));

  if (fp.n_iters == 0) {   // q(v) stays at its initial N(0, I): C = chol(I)
    P.tri_copy(nb, Np, ws.Lam, ws.Cb);
    potrf(P, nb, Np, ws.Cb, ws.invC, is, ws.info);
  }

  const bool in_pad = Np > N;
  const int mrows = in_pad ? Np : Np + LEAF;
  Mat trow = ws.T1.sub(in_pad ? Np - 1 : Np, 0);
  bool c_stale = false;
  for (int it = 0; it < fp.n_iters; ++it) {
    // First iteration: the VGP starts at q_mu = 0, q_sqrt = I, i.e. Lam_old = I EXACTLY and theta1_old = 0, so
    //   Lam = I + gamma G,   G S = (Lam - I) Lam^-1 / gamma = (I - S) / gamma,
    //   Lam m = theta1 = gamma L^T (y/nu)  =>  u = L^T D (L m - y) = G m - theta1 / gamma = -m / gamma
    // (used for a single iteration in fp64; fp32 keeps the general form there: I - S cancels).
    const bool it0 = it == 0 && fp.gamma > 0.0;
    // ... and M = u m^T + G S is SYMMETRIC there, so the Cholesky backpropagation collapses: Phi(M) + Phi(M)^T = M gives
    //   sym(Kbar) = L^-T M L^-1 / 2,   and with Woodbury  S = I - gamma L^T P L,  P = (diag(nu) + gamma (K0 + J I))^-1 :
    //   L^-T (I - S) L^-1 / gamma = P,   m = gamma L^T P y,   sym(Kbar) = (P - gamma (P y)(P y)^T) / 2.
    // The hyper-gradient then needs P alone (a factorisation, its triangular inverse, one lower x lower product: N^3) - no
    // C = chol(Lam), S, Phi, W or Kbar (8/3 N^3).  C itself is only missing when nothing refreshes it before it is used:
    // the prediction after a single iteration (the general form is kept there) and the loss sample of iteration 0
    // (refactored for it).
    const bool woodbury = it0 && fp.n_iters >= 2;
    const bool first = it0 && !woodbury && sizeof(real) == sizeof(double);
    const bool lam_direct = woodbury || first;
    // O(N^2) passes that nothing on the way waits for run on the side queue, beside the products and the factorisation's chain
    // (they are bound by HBM, those by the matrix pipes and by latency): woodbury - the identity the triangular inverse starts from;
    // every iteration - theta1's update; a general iteration - the mirror image of G the solves take as their right-hand side.
    if (woodbury) {
      SideSection side(P, SIDE_PASSES);
      P.identity(nb, Np, ws.T2);
      side.end();
    }
    // K0 and L = chol(K0 + jitter I) at the CURRENT hypers                       (models.py:209)
    P.mark("kmat");
    // (the blocks above the block diagonal of Lb are cleared by the first call of the fit and stay clear; woodbury: the same
    //  pass leaves diag(nu) + gamma (K0 + J I) in Cb)
    if (woodbury) P.kmat(nb, N, Np, fp.R, X, ws.ustate, fp.kernel, fp.jitter, ws.Lb, it == 0, ws.Cb, fp.gamma, ws.nu, fp.Rv, fp.xoff);
    else P.kmat(nb, N, Np, fp.R, X, ws.ustate, fp.kernel, fp.jitter, ws.Lb, it == 0, Mat{nullptr, 0, 0}, 0.0, nullptr, fp.Rv, fp.xoff);
    P.mark("potrf K");
    // woodbury: K0 + J I and diag(nu) + gamma (K0 + J I) factor as ONE batch of 2 nb members (Lb and Cb are interleaved member by
    // member, FitWorkspace::paired): the factorisation is bound by its chain of small kernels, not by its flop, and the
    // chain is as long for 2 nb members as for nb
    static const bool no_pair = std::getenv("BAYENS_NO_PAIR") != nullptr;   // (measurement switch)
    const bool pair = woodbury && ws.paired() && !no_pair;
    if (pair) {
      potrf(P, 2 * nb, Np, Mat{ws.Lb.p, ws.Lb.ld, ws.Lb.stride / 2}, ws.invL, is / 2, ws.info_pair);
      P.info_fold(nb, ws.info_pair, ws.info);
    } else {
      potrf(P, nb, Np, ws.Lb, ws.invL, is, ws.info);
    }
    if (fp.loss_trace && it > 0 && (it - 1) % LOSS_EVERY == 0) {
      neg_elbo(P, ws, fp, (it - 1) / LOSS_EVERY, c_stale);
      c_stale = false;
    }
    P.mark("Lam update");
    // natural-gradient step:  G = L^T D L ;  Lam <- (1-g) Lam + g (I + G),  theta1 <- (1-g) theta1 + g L^T D y
    // (first iteration: the product's epilogue writes Lam = I + gamma G itself, no separate G)
    P.scale_rows_rsqrt(nb, Np, ws.Lb, ws.nu, ws.T1);                 // W = D^1/2 L
    if (lam_direct) {
      P.gemm(nb, Np, Np, Np, fp.gamma, ws.T1, ws.T1, 0.0, ws.Lam, GEMM_TA | GEMM_TRI_OUT | GEMM_KTRIM_TN_LOWER, 1.0);
      if (!woodbury) P.tri_copy(nb, Np, ws.Lam, ws.Cb);
    } else {
      P.gemm(nb, Np, Np, Np, 1.0, ws.T1, ws.T1, 0.0, ws.T2, GEMM_TA | GEMM_TRI_OUT | GEMM_KTRIM_TN_LOWER, 0.0);
      P.lam_update(nb, Np, ws.T2, fp.gamma, ws.Lam, ws.Cb, ws.T1, 1);   // Lam, Cb = tril(Lam)
    }
    {
      SideSection side(P, SIDE_PASSES);
      if (!lam_direct) P.lam_update(nb, Np, ws.T2, fp.gamma, ws.Lam, ws.Cb, ws.T1, 2);   // T1 = G (full)
      P.lt_dvec(nb, Np, ws.Lb, ws.y, ws.nu, fp.gamma, 1.0 - fp.gamma, ws.theta1);
      side.end();
    }
    P.mark("potrf Lam");
    if (!pair) potrf(P, nb, Np, ws.Cb, ws.invC, is, ws.info);         // C = chol(Lam)   (woodbury: chol(diag(nu) + gamma K))
    if (woodbury) {
      P.mark("P = (nu + gamma K)^-1");
      P.side_join(SIDE_PASSES);
      trsm_rln(P, nb, Np, Np, ws.Cb, ws.invC, is, ws.T2, true);       // triangular inverse (lower)
      P.gemm(nb, Np, Np, Np, 1.0, ws.T2, ws.T2, 0.0, ws.T1, GEMM_TA | GEMM_TRI_OUT | GEMM_KTRIM_TN_LOWER, 0.0);   // P (lower blocks)
      P.symv_lower(nb, Np, ws.T1, ws.y, ws.avec);                     // p = P y
      if (fp.loss_trace) P.lt_dvec(nb, Np, ws.Lb, ws.avec, nullptr, fp.gamma, 0.0, ws.mvec);   // m = gamma L^T p, for the sample
      c_stale = true;                                                 // Cb does not hold chol(Lam)
      P.mark("grad+adam");
      P.grad_adam(nb, N, Np, fp.R, X, fp.kernel, ws.T1, ws.ustate, fp.lr, ws.avec, fp.gamma);
      continue;
    }
    P.mark("G S solves");
    P.side_join(SIDE_PASSES);
    // G S is all the Adam step (models.py:210) needs of S: for i >= j
    //   (L^T Lbar)_ij = u_i m_j + (G S)_ij,  Lbar = tril(D ((L m - y) m^T + L S)),  u = L^T D (L m - y),
    // because row k of Lbar only meets L_ki with k >= i >= j, where the tril() does not cut.
    if (first) {
      P.identity(nb, Np, ws.T2);
      trsm_rln(P, nb, Np, Np, ws.Cb, ws.invC, is, ws.T2, true);       // C^-1 (lower)
      P.gemm(nb, Np, Np, Np, 1.0, ws.T2, ws.T2, 0.0, ws.T1, GEMM_TA | GEMM_TRI_OUT | GEMM_KTRIM_TN_LOWER, 0.0);   // S (lower blocks)
      P.symv_lower(nb, Np, ws.T1, ws.theta1, ws.mvec);                // m = S theta1
      P.mark("Phi");
      // Phi(L^T Lbar), clean lower triangle: -(m m^T + S) / gamma + I / gamma
      P.form_phi(nb, Np, ws.T1, ws.mvec, ws.mvec, ws.T2, -1.0 / fp.gamma, 1.0 / fp.gamma, -1.0 / fp.gamma);
    } else {
      // [G; theta1^T] C^-T C^-1 = [G S; m^T]: m = S theta1 rides through the same two solves as one extra row (the last
      // padded row when there is padding).
      P.aug_set(nb, Np, trow, ws.theta1, in_pad ? 0 : LEAF - 1);
      trsm_rlt(P, nb, mrows, Np, ws.Cb, ws.invC, is, ws.T1);          // [G; theta1^T] C^-T
      trsm_rln(P, nb, mrows, Np, ws.Cb, ws.invC, is, ws.T1, false);   // ... C^-1
      P.aug_get(nb, Np, trow, ws.mvec);                               // m
      P.mark("Phi");
      P.resid(nb, Np, ws.Lb, ws.mvec, ws.y, ws.avec);                 // a = L m - y
      P.lt_dvec(nb, Np, ws.Lb, ws.avec, ws.nu, 1.0, 0.0, ws.uvec);    // u = L^T D a
      P.form_phi(nb, Np, ws.T1, ws.uvec, ws.mvec, ws.T2, 1.0, 0.0, 1.0);   // Phi(L^T Lbar), clean lower triangle
    }
    P.mark("Phi L^-1");
    trsm_rln(P, nb, Np, Np, ws.Lb, ws.invL, is, ws.T2, true);         // W = Phi L^-1 (lower)
    P.mark("W^T L^-1");
    P.transpose_lower(nb, Np, ws.T2, ws.T1);                          // W^T
    trsm_rln(P, nb, Np, Np, ws.Lb, ws.invL, is, ws.T1, false);        // Kbar^T = W^T L^-1
    P.mark("grad+adam");
    P.grad_adam(nb, N, Np, fp.R, X, fp.kernel, ws.T1, ws.ustate, fp.lr);
  }
  P.mark("predict: kmat+potrf");

  // prediction at the FINAL hypers with the (m, C) of the last natgrad step       (models.py:217-220)
  // (T2 is free from here on unless the loss is sampled below: the identity the triangular inverse starts from is written
  //  beside the kernel matrix and its factorisation)
  const bool sample_last = fp.loss_trace && fp.n_iters > 0 && (fp.n_iters - 1) % LOSS_EVERY == 0;
  if (!sample_last) {
    SideSection side(P, SIDE_PASSES);
    P.identity(nb, Np, ws.T2);
    P.diag_init(nb, Np, ws.nu, fp.jitter, ws.Lam);                    // Lam reused: diag(nu + J), the first term of the covariance
    side.end();
  }
  P.kmat(nb, N, Np, fp.R, X, ws.ustate, fp.kernel, fp.jitter, ws.Lb, fp.n_iters == 0, Mat{nullptr, 0, 0}, 0.0, nullptr, fp.Rv, fp.xoff);
  potrf(P, nb, Np, ws.Lb, ws.invL, is, ws.info);
  if (sample_last) neg_elbo(P, ws, fp, (fp.n_iters - 1) / LOSS_EVERY);
  // With K0 = L L^T - J I (J the jitter):  A^T = K0 L^-T = L - J L^-T  and  K0 - A^T A = J I - J^2 L^-T L^-1,
  // so the full solve K0 L^-T and the rank-N update A^T A shrink to the triangular inverse (a lower-rhs solve on I,
  // N^3/3) and the lower x lower product L^-T L^-1 (N^3/3):  cov = diag(nu + J) - J^2 L^-T L^-1 + (A^T C^-T)(A^T C^-T)^T.
  P.mark("predict: L^-1, A^T");
  if (sample_last) P.identity(nb, Np, ws.T2);
  P.side_join(SIDE_PASSES);
  trsm_rln(P, nb, Np, Np, ws.Lb, ws.invL, is, ws.T2, true);           // L^-1 (lower)
  P.at_from_linv(nb, Np, ws.Lb, ws.T2, fp.jitter, ws.T1);             // A^T = L - J L^-T
  P.gemv_n(nb, Np, ws.T1, ws.mvec, ws.muvec);                         // mu = A^T m
  P.extract_vec(nb, N, Np, ws.muvec, mu);
  P.mark("predict: A^T C^-T");
  trsm_rlt(P, nb, Np, Np, ws.Cb, ws.invC, is, ws.T1);                 // A^T C^-T (in place)
  P.mark("predict: cov products");
  if (sample_last) P.diag_init(nb, Np, ws.nu, fp.jitter, ws.Lam);     // Lam reused: diag(nu + J)
  P.gemm(nb, Np, Np, Np, -fp.jitter * fp.jitter, ws.T2, ws.T2, 1.0, ws.Lam,
         GEMM_TA | GEMM_TRI_OUT | GEMM_KTRIM_TN_LOWER, 0.0);
  P.gemm(nb, Np, Np, Np, 1.0, ws.T1, ws.T1, 1.0, ws.Lam, GEMM_TB | GEMM_TRI_OUT, 0.0);
  if (cov) P.extract(nb, N, ws.Lam, cov, 1);
  if (var) P.extract_diagonal(nb, N, ws.Lam, var);     // (the reduced export of a per-gridpoint field: no N x N table leaves the fit)
  P.mark("chol(cov)");
  if (chol || logdet || basis) {
    // Distribution.__post_init__ : distrax caches chol(cov) and its log-determinant  (data.py:38-39)
    factor_with_basis(P, ws, N, ws.Lam, ws.muvec, chol, logdet, basis);
  }
  if (hyp) P.hyp_out(nb, ws.ustate, hyp);
  P.mark("end");
}

void mvn_factor(Prims& P, FitWorkspace& ws, int N, const real* mu_pad, real* chol, double* logdet,
                real* basis) {
  factor_with_basis(P, ws, N, ws.T2, mu_pad, chol, logdet, basis);
}

int nuclear_norms(Prims& P, FitWorkspace& ws, int nb, Mat Li, Mat Lj, double* nuc, double* scratch_rel, double ell0,
                  double tol, int max_iters, const double* cap, double cap_scale) {
  // M = L_j^T L_i  (both lower triangular: k >= max(i0, j0))
  P.gemm(nb, ws.Np, ws.Np, ws.Np, 1.0, Lj, Li, 0.0, ws.W0, GEMM_TA | GEMM_KTRIM_TN_LOWER, 0.0);
  return polar_nuclear(P, ws, nb, nuc, scratch_rel, ell0, tol, max_iters, cap, cap_scale, 0.0);
}

int polar_nuclear(Prims& P, FitWorkspace& ws, int nb, double* nuc, double* scratch_rel, double ell0, double tol,
                  int max_iters, const double* cap, double cap_scale, double ell_done) {
  const int Np = ws.Np;
  const long is = ws.inv_stride();
  Mat M = ws.W0, X = ws.Lb, Z = ws.Cb, T = ws.T1;
  P.frob_normalise(nb, Np, M, X, cap, cap_scale);
  double ell = ell0;
  std::vector<double> rel(nb);
  int it = 0;
  bool converged = false;
  for (; it < max_iters; ++it) {
    // dynamically weighted Halley coefficients from the current lower bound of the singular values
    const double l2 = ell * ell;
    const double dd = std::cbrt(4.0 * (1.0 - l2) / (l2 * l2));
    const double a = std::sqrt(1.0 + dd) + 0.5 * std::sqrt(8.0 - 4.0 * dd + 8.0 * (2.0 - l2) / (l2 * std::sqrt(1.0 + dd)));
    const double b = (a - 1.0) * (a - 1.0) / 4.0, c = a + b - 1.0;
    // Z = I + c X^T X = W W^T ;  X <- (b/c) X + (a - b/c) X Z^-1
    P.gemm(nb, Np, Np, Np, c, X, X, 0.0, Z, GEMM_TA | GEMM_TRI_OUT | GEMM_ZERO_SKIPPED, 1.0);
    potrf(P, nb, Np, Z, ws.invC, is, ws.info);
    P.copy(nb, Np, X, T);
    trsm_rlt(P, nb, Np, Np, Z, ws.invC, is, T);
    trsm_rln(P, nb, Np, Np, Z, ws.invC, is, T, false);
    P.axpby_rel(nb, Np, b / c, X, a - b / c, T, scratch_rel);
    // With weights this close to Halley's (3, 1, 3) the step is cubically convergent for every singular value near 1,
    // wherever the true spectrum lies, and the size of the step IS the error it removes: `tol` applies.  While the
    // weights are still the aggressive ones of a small ell a singular value near 1 may be thrown back, so only an
    // (almost) vanishing step counts as convergence there.
    const bool halley = 1.0 - ell < 0.05;
    ell = ell * (a + b * l2) / (1.0 + c * l2);
    if (ell > 1.0) ell = 1.0;
    P.read_back(scratch_rel, rel.data(), nb);
    double worst = 0.0;
    for (double r : rel) worst = r > worst || r != r ? r : worst;
    if (worst < (halley ? tol : (tol < 1e-13 ? tol : 1e-13))) { ++it; converged = true; break; }
    // ell is a lower bound of every singular value of the new X when ell0 was one of X_0: once it is within ell_done
    // of 1 the polar factor is too - provided the step just taken was a small one, i.e. nothing of weight lay below
    // the caller's ell0 and is still on its way.
    if (ell_done > 0.0 && 1.0 - ell <= ell_done && worst < 1e-3) { ++it; converged = true; break; }
  }
  P.dot_all(nb, Np, X, M, nuc);    // ||M||_* = tr(U^T M)
  return converged ? it : -it;     // a NaN step or a pair still moving after max_iters: the caller must not use nuc
}

double fit_flops(int N, int R, int n_iters) {
  // per iteration: chol K 1/3, G = L^T D L 1/3, chol Lam 1/3, G C^-T 1, (.) C^-1 1,
  //                Phi L^-1 1/3, W^T L^-1 1  => 13/3 N^3 ; kernel matrix 2 N^2 R
  // first iteration (Lam_old = I: G S = (I - S) / gamma): C^-1 1/3 + C^-T C^-1 1/3 replace the two solves => 3 N^3
  // prediction: chol K 1/3, L^-1 1/3, A^T C^-T 1, L^-T L^-1 1/3, rank-N update 1 => 3 N^3, chol(cov) 1/3
  // ... and when another iteration follows, the symmetric form: chol K, G, chol(nu + gamma K), its inverse, P => 5/3 N^3
  // Scope of the count (round 4): the fit as THIS precision runs it WITHOUT loss samples.  fp32 keeps a lone first iteration in
  // its general form (the (I - S) / gamma shortcut cancels in fp32: `first` in gp_fit needs sizeof(real) == sizeof(double)), so
  // it saves nothing at n_iters == 1; with loss samples the sample of iteration 0 refactors Lam (general path: + 2/3 N^3 + a
  // solve per sample) and the fused kernels (small.hip, medium.hip) keep the first iteration in its general / (I - S) / gamma
  // form - those runs execute MORE than this figure, and no bench line that quotes a flop rate requests samples.
  double n3 = (double)N * N * N, n2r = 2.0 * N * (double)N * R;
  const bool f64 = sizeof(real) == sizeof(double);
  const double first_saves = n_iters >= 2 ? 8.0 / 3.0 * n3 : (n_iters == 1 && f64 ? 4.0 / 3.0 * n3 : 0.0);
  return n_iters * (13.0 / 3.0 * n3 + n2r) - first_saves + (3.0 * n3 + n2r) + n3 / 3.0;
}

}  // namespace BAYENS_NS
