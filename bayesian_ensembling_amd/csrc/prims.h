// prims.h - the batched device primitives the host-side driver sequences.
//
// Everything is `real` (fp64 - the reference precision - by default; this header, the driver and the kernels are
// compiled a second time with -DBAYENS_F32 into namespace bayens_f32 for the fp32 mode), row-major, batched over
// `nb` independent ensemble members.  Scalars that steer the optimiser (hyper-parameters, Adam moments), reductions
// (log-determinants, traces, the loss) and the kernel-matrix distances stay double in both modes.  A `Mat` is a
// strided view: element (b, i, j) lives at p[b*stride + i*ld + j].  All matrix dimensions handed to
// the primitives are multiples of LEAF (64): the driver pads every member to Np = ceil(N/64)*64 with
// an identity block, which keeps the padded part decoupled from the real part through every
// factorisation, solve and product (block-diagonal stays block-diagonal).
//
// The HIP implementation is in kernels.hip / gemm.hip (the product).  oracle/hostcheck/ holds a naive
// CPU implementation of the same interface that is used ONLY by the CPU test-suite to validate the
// driver's sequencing without a GPU; the product never links it.
#pragma once
#include <cstddef>
#include <cstdint>

#ifdef BAYENS_F32
#define BAYENS_NS bayens_f32
namespace bayens_f32 { typedef float real; }
#else
#define BAYENS_NS bayens
namespace bayens { typedef double real; }
#endif

namespace BAYENS_NS {

constexpr int LEAF = 64;   // diagonal-block size of the blocked factorisations / solves

struct Mat {
  real* p;
  int ld;
  long stride;
  Mat sub(int i, int j) const { return Mat{p + (long)i * ld + j, ld, stride}; }
};

// gemm flags
enum : int {
  GEMM_TA = 1,             // A is stored k x m (use A^T)
  GEMM_TB = 2,             // B is stored n x k (use B^T)
  GEMM_TRI_OUT = 4,        // C square: skip 64x64 blocks strictly above the diagonal (not written)
  GEMM_KTRIM_TN_LOWER = 8, // with TA, !TB: A and B are lower triangular => k starts at max(i0, j0)
  GEMM_KTRIM_A_LOWER = 16, // with !TA: A (m x k) is lower trapezoidal, A_ik = 0 for k > i => k ends at the tile's last row
  GEMM_ZERO_SKIPPED = 32,  // with TRI_OUT: the skipped 64x64 blocks are written with zeros instead of left alone
};

struct Prims {
  virtual ~Prims() {}
  // phase marker for the optional stage profile (no-op unless the implementation records it)
  virtual void mark(const char* /*phase*/) {}

  // C[m x n] = alpha * op(A)[m x k] * op(B)[k x n] + beta * C  (+ diag_add where i == j).
  // C may alias A when n <= 128 and C/A address the same rows (used by the in-place leaf solves).
  virtual void gemm(int nb, int m, int n, int k, double alpha, Mat A, Mat B, double beta, Mat C,
                    int flags, double diag_add) = 0;

  // Whole triangular solve of a narrow diagonal block in one launch: B[m x n] <- B L^-T (mode 0) or B L^-1
  // (mode 1), n in {64, 128, 192, 256}, L the n x n lower block whose inverted 64x64 diagonal blocks are in inv.
  // Rows are independent, so every workgroup carries its 64 rows through all column blocks (one pass over B).
  virtual void trsm_panel(int nb, int m, int n, Mat L, const real* inv, long inv_stride, Mat B, int mode) = 0;

  // Cholesky of one 64x64 diagonal block per member, in place (lower; strict upper set to 0), plus its
  // inverse into inv (64x64 row-major per member, member stride inv_stride).  A non-positive pivot j
  // (0-based inside the block) records info[b] = pivot_base + j + 1 if info[b] was 0.
  // Concurrency hints.  Work issued between side_begin() and side_end() starts after everything issued before
  // side_begin() and may overlap whatever the caller issues after side_end(); side_join() makes everything issued
  // after it wait for that side work.  Program order (these defaults) is always a valid schedule.
  // `ch`: two independent join points over the same side queue - 0 the look-ahead of the blocked factorisation, 1 the O(N^2)
  // passes the fit moves out of the way of its products (SIDE_PASSES); joining one channel does not wait for the other's work.
  // (A queue of its own for channel 1 was measured: no faster, profiles/r05_side_passes.txt.)
  virtual void side_begin() {}
  virtual void side_end(int ch = 0) { (void)ch; }
  virtual void side_join(int ch = 0) { (void)ch; }
  // leaves a side section without publishing it (error path: a launch inside the section threw); never throws
  virtual void side_cancel() noexcept {}

  virtual void potrf_leaf(int nb, Mat Ablk, real* inv, long inv_stride, int pivot_base, int* info) = 0;
  // two adjacent leaves (a 128x128 diagonal block): leaf, 64x64 panel solve, 64^3 update, leaf.  A back end may
  // fuse the four steps; this composition is the definition.
  virtual void potrf_leaf2(int nb, Mat Ablk, real* inv, long inv_stride, int pivot_base, int* info) {
    potrf_leaf(nb, Ablk, inv, inv_stride, pivot_base, info);
    trsm_panel(nb, LEAF, LEAF, Ablk, inv, inv_stride, Ablk.sub(LEAF, 0), 0);
    gemm(nb, LEAF, LEAF, LEAF, -1.0, Ablk.sub(LEAF, 0), Ablk.sub(LEAF, 0), 1.0, Ablk.sub(LEAF, LEAF),
         GEMM_TB | GEMM_TRI_OUT, 0.0);
    potrf_leaf(nb, Ablk.sub(LEAF, LEAF), inv + LEAF * LEAF, inv_stride, pivot_base + LEAF, info);
  }

  // A whole 256 x 256 diagonal block (four leaves): returns false when the back end has no fused form, and the driver then
  // runs the recursive composition (potrf_rec in driver.cpp, which is the definition of the result up to rounding).
  virtual bool potrf_diag256(int /*nb*/, Mat /*Ablk*/, real* /*inv*/, long /*inv_stride*/, int /*pivot_base*/, int* /*info*/) {
    return false;
  }

  virtual void copy(int nb, int n, Mat src, Mat dst) = 0;        // full n x n copy
  virtual void tri_copy(int nb, int n, Mat src, Mat dst) = 0;    // dst = tril(src), zeros above
  virtual void transpose_lower(int nb, int n, Mat W, Mat Wt) = 0;  // Wt = tril(W)^T (zeros below diag)

  // state initialisation: Lam = I (n x n); theta1 = m = 0; u = softplus^-1(1); adam = 0; y/nu padded
  // (lam_identity = false: Lam is left alone - the first natural-gradient step writes it without reading it)
  virtual void init_state(int nb, int N, int Np, Mat Lam, real* theta1, real* mvec, double* ustate,
                          const real* y_in, const real* nu_in, real* y_pad, real* nu_pad,
                          int* info, bool lam_identity = true) = 0;

  // Lb = tril(K0) + jitter*I (zeros above the diagonal; identity in the padded part) from the current hypers
  // softplus(ustate[b][0..1]).  X is [nb][N][R] compact.  K0 itself is never stored: the prediction uses
  // K0 = L L^T - jitter*I (driver.cpp) and the hyper-gradient re-evaluates the kernel (grad_adam).
  // clear_upper = false: the 64x64 blocks above the block diagonal are not written (they hold zeros from an earlier call).
  // With Mb: also Mb = tril(m_scale (K0 + jitter*I) + diag(m_diag)) on the blocks on and below the block diagonal.
  // Rv / xoff (device, [nb]; null: uniform): member b has Rv[b] <= R realisations, its [N][Rv[b]] block starts xoff[b] elements into X.
  virtual void kmat(int nb, int N, int Np, int R, const real* X, const double* ustate, int kernel,
                    double jitter, Mat Lb, bool clear_upper = true, Mat Mb = Mat{nullptr, 0, 0}, double m_scale = 0.0,
                    const real* m_diag = nullptr, const int* Rv = nullptr, const long* xoff = nullptr) = 0;

  virtual void scale_rows_rsqrt(int nb, int n, Mat src, const real* nu, Mat dst) = 0;  // dst_ij = src_ij/sqrt(nu_i)
  // out = beta * out + alpha * L^T (v / nu)   (L lower triangular)
  // (nu == nullptr: L^T v)
  virtual void lt_dvec(int nb, int n, Mat L, const real* v, const real* nu, double alpha, double beta,
                       real* out) = 0;
  // natural-gradient precision update from G = L^T D L (lower 64-blocks of G valid):
  //   Lam <- (1-gamma) Lam + gamma (I + G) on the lower triangle;  Cb <- tril(Lam) (zeros above the diagonal inside
  //   the diagonal 64-blocks; the blocks above the block diagonal are neither written nor ever read);
  //   Gfull <- G mirrored to a full symmetric matrix
  // parts: 1 = Lam and Cb, 2 = Gfull, 3 = both (the two halves touch disjoint outputs and may run on different queues)
  virtual void lam_update(int nb, int n, Mat G, double gamma, Mat Lam, Mat Cb, Mat Gfull, int parts = 3) = 0;
  // Phi(L^T Lbar) without forming Lbar: for i >= j, (L^T Lbar)_ij = u_i m_j + (G S)_ij with u = L^T D (L m - y)
  //   Phi_ij = u_i m_j + GS_ij (i > j), half of it on the diagonal, 0 above,   GS_ij = scale * M_ij + diag_add * delta_ij.
  // (scale, diag_add) = (1, 0): M is G S itself.  In the FIRST iteration Lam_old = I, so Lam = I + gamma G and
  // G S = (Lam - I) Lam^-1 / gamma = (I - S) / gamma: M = S (lower) with (scale, diag_add) = (-1/gamma, 1/gamma) - the two
  // full solves that form G S (2 N^3) shrink to the explicit S = C^-T C^-1 (2/3 N^3).
  // u enters as u_scale * u (the first iteration has u = -m / gamma: Lam m = theta1 = gamma L^T (y/nu) there).
  virtual void form_phi(int nb, int n, Mat M, const real* u, const real* mvec, Mat Phi, double scale, double diag_add,
                        double u_scale) = 0;
  // out = S v for a symmetric S of which only the elements on and below the diagonal are read
  virtual void symv_lower(int nb, int n, Mat S, const real* v, real* out) = 0;
  // a row of an n-column right-hand side that carries a vector through the same triangular solves
  // (x^T C^-T C^-1 = (C^-T C^-1 x)^T): rows[0][:] = vec and the `zero_rows` rows after it are cleared.
  // The row is either the first of a 64-row block appended below the matrix (zero_rows = 63) or, when the
  // member has padding (Np > N), the last padded row (zero_rows = 0): padded rows are decoupled from the real
  // block, so whatever they carry never reaches the real part of any later product.
  virtual void aug_set(int nb, int n, Mat rows, const real* vec, int zero_rows) = 0;
  virtual void aug_get(int nb, int n, Mat rows, real* vec) = 0;           // vec = rows[0][:]
  // the two bordering rows of the covariance factorisation: row 0 = 1 (columns < N, else 0), row 1 = mu_pad.
  // in_padding = 0: rows of an appended 64-row block (rows 2..63 zeroed).  in_padding = 1: `rows` are rows N and
  // N+1 of the padded matrix itself; they also get a huge diagonal entry (1e30) so that their pivots stay
  // positive - it only touches the padded part, which is excluded from chol / logdet / basis outputs.
  virtual void basis_rows_set(int nb, int N, int Np, Mat rows, const real* mu_pad, int in_padding) = 0;
  virtual void basis_rows_get(int nb, int N, int Np, Mat rows, real* basis) = 0;
  virtual void resid(int nb, int n, Mat L, const real* mvec, const real* y, real* a) = 0;  // a = L m - y
  // g = <Kbt, dK/dtheta> over the real N x N part (Kbt = Kbar^T; dK symmetric), then one Adam step on
  // the unconstrained hypers.  ustate per member: [u_var, u_ls, m_var, m_ls, v_var, v_ls, t, pad]
  // With pvec: Kbt holds the lower triangle of a symmetric P and the weight of dK is the symmetric
  // (P - sym_gamma p p^T) / 2 (off-diagonal pairs counted once with twice the weight; nothing above the diagonal is read).
  virtual void grad_adam(int nb, int N, int Np, int R, const real* X, int kernel, Mat Kbt,
                         double* ustate, double lr, const real* pvec = nullptr, double sym_gamma = 0.0) = 0;
  // out_i = sum_k A_ik^2 (all n columns)
  virtual void row_sqsum(int nb, int n, Mat A, real* out) = 0;
  // -ELBO of the whitened VGP with the heteroskedastic Gaussian likelihood from its pieces (models.py:142-149,
  // GPflow VGP.elbo):  sum_{t<N} [ log(2 pi)/2 + log(nu_t)/2 + (a_t^2 + f_t) / (2 nu_t) ]
  //                    + ( sum_{t<N} m_t^2 - N + 2 logdetC + (trS - sum_{t>=N} C_tt^-2) ) / 2
  // a = L m - y, f_t = (L S L^T)_tt, logdetC[b] = sum_{i<N} log C_ii, trS[b] = ||C^-1||_F^2 over the padded
  // matrix (the padded block of C is diagonal - Lam's padding is a multiple of I, not I - and is subtracted).
  // Written to trace[b * stride + slot].
  virtual void elbo_combine(int nb, int N, int Np, const real* a, const real* f, const real* nu,
                            const real* mvec, Mat C, const double* logdetC, const double* trS, double* trace,
                            int stride, int slot) = 0;
  virtual void gemv_n(int nb, int n, Mat A, const real* x, real* y) = 0;        // y = A x
  virtual void identity(int nb, int n, Mat A) = 0;                              // A = I
  // At = L - jitter * Linv^T  (= K0 L^-T, because K0 = L L^T - jitter I); L, Linv lower triangular
  virtual void at_from_linv(int nb, int n, Mat L, Mat Linv, double jitter, Mat At) = 0;
  virtual void diag_init(int nb, int n, const real* nu, double add, Mat dst) = 0;   // dst = diag(nu + add)
  // compact copy-out of the real N x N part: mode 0 as is, 1 symmetric from the lower triangle,
  // 2 lower triangle with zeros above
  virtual void extract(int nb, int N, Mat src, real* dst, int mode) = 0;
  virtual void extract_vec(int nb, int N, int Np, const real* src, real* dst) = 0;
  // columns [c0, c1) of extract(mode 2): dst[b][i][j] = j <= i ? src_b[i][j] : 0 for i < N, c0 <= j < c1 <= N - a factor's block
  // column leaves for the compact output as soon as it is final, beside the rest of the factorisation
  virtual void extract_cols(int nb, int N, Mat src, real* dst, int c0, int c1) = 0;
  // dst[b][i] = src_b[i][i], i < N (the reduced export: marginal variances without the N x N table)
  virtual void extract_diagonal(int nb, int N, Mat src, real* dst) = 0;
  virtual void logdet(int nb, int N, Mat Lc, double* out) = 0;   // sum_{i<N} log Lc_ii
  // info[b] = its own non-zero value, else the first non-zero of pair[2b], pair[2b+1]; pair is cleared
  virtual void info_fold(int nb, int* pair, int* info) = 0;
  virtual void hyp_out(int nb, const double* ustate, double* hyp) = 0;  // softplus(u) -> (variance, lengthscale)

  // ---- pieces of the full-covariance Wasserstein distance (polar iteration on L_j^T L_i) ----
  // X = M / ||M||_F per member (Frobenius norms are not returned)
  virtual void frob_normalise(int nb, int n, Mat M, Mat X, const double* cap = nullptr, double cap_scale = 0.0) = 0;
  // X <- a X + b T ; rel[b] = ||X_new - X_old||_F / ||X_new||_F   (rel: device array [nb])
  virtual void axpby_rel(int nb, int n, double a, Mat X, double b, Mat T, double* rel) = 0;
  // out[b] = sum_ij A_ij B_ij over the n x n matrices (out: device array [nb])
  virtual void dot_all(int nb, int n, Mat A, Mat B, double* out) = 0;
  // synchronise and copy n device doubles to the host (the only host read-back of the polar iteration)
  virtual void read_back(const double* dev, double* host, int n) = 0;
};

}  // namespace BAYENS_NS
