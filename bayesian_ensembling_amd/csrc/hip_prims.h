// hip_prims.h - the gfx950 implementation of the primitive interface (kernels in gemm.hip / kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

#include "prims.h"

// Workgroup barrier that orders LDS traffic only.  `__syncthreads()` also waits for every global load AND store the
// wave has in flight (s_waitcnt vmcnt(0): on gfx9 stores return through the same counter), i.e. it puts an HBM round
// trip on the critical path after every result store.  Legal wherever no thread reads global memory that another
// thread of the workgroup wrote earlier in the kernel (the compiler still waits for a load before its value is used).
#define BAYENS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

namespace BAYENS_NS {

struct HipError {
  int code;
  std::string what;
};

#define BAYENS_HIP(expr)                                                                      \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess)                                                                    \
      throw ::BAYENS_NS::HipError{(int)e__, std::string(#expr) + ": " + hipGetErrorString(e__)}; \
  } while (0)

// kernel families of the per-launch accounting (bayens_profile_get): one per kernel symbol of the GEMM group
enum { FAM_ALL = 0, FAM_GEMM_NT = 1, FAM_GEMM_NN = 2, FAM_GEMM_TN = 3, FAM_PANEL64 = 4, FAM_TRSM_PANEL = 5, FAM_TILE64 = 6, FAM_COUNT = 7 };

struct KernelProfile {
  bool enabled = false;
  struct Ev { hipEvent_t a, b; int fam; int m, n, k, flags; double flop; };
  std::vector<Ev> events;   // per launch
  std::vector<hipEvent_t> pool;
  double flop[FAM_COUNT] = {};
  double bytes[FAM_COUNT] = {};   // algorithmic operand bytes (GEMM families; panel solves: B in+out, L)
  long launches[FAM_COUNT] = {};
  std::vector<std::pair<std::string, hipEvent_t>> marks;   // stage profile of the driver
};

struct HipPrims : Prims {
  hipStream_t stream = nullptr;
  KernelProfile prof;
  // scratch owned by the primitives (scaled inputs for the kernel matrix, gradient partial sums)
  double* xs = nullptr;       // [nb][Np][R4]
  double* ssq = nullptr;      // [nb][Np]
  double* gpart = nullptr;    // [nb][tiles][2]
  size_t xs_cap = 0, ssq_cap = 0, gpart_cap = 0;
  unsigned long long* leaf_stamps = nullptr;   // diagnostic only (bayens_bench_leaf)
  // look-ahead stream of the blocked Cholesky (side_begin / side_end / side_join); created on first use
  hipStream_t side_stream = nullptr, main_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  bool in_side = false, side_pending[2] = {false, false}, owns_side = true;

  // gemm: largest launch (workgroups of the 128 x 128 tiling) that takes the 64 x 64-tile kernel; -1: the measured defaults
  // (gemm.hip small_output), 0: never.  bayens_set_gemm_tile64_max - the primitive tests hold both kernels to the same cases.
  int tile64_max = -1;
  explicit HipPrims(hipStream_t s) : stream(s) {}
  // per-launch accounting on/off: recorded events go back to the pool, counters are cleared
  void profile_reset(bool on) {
    prof.enabled = on;
    for (auto& e : prof.events) { prof.pool.push_back(e.a); prof.pool.push_back(e.b); }
    prof.events.clear();
    for (int f = 0; f < FAM_COUNT; ++f) { prof.flop[f] = 0; prof.bytes[f] = 0; prof.launches[f] = 0; }
  }
  // summed device ms of the recorded launches of one family (FAM_ALL: all); the device must be idle
  double profile_ms(int family) {
    double tot = 0;
    for (auto& e : prof.events) {
      if (family != FAM_ALL && e.fam != family) continue;
      float f = 0;
      BAYENS_HIP(hipEventElapsedTime(&f, e.a, e.b));
      tot += f;
    }
    return tot;
  }
  ~HipPrims() override;
  void ensure_scratch(int nb, int Np, int R4);
  void mark(const char* phase) override;
  void ensure_side_stream();
  // use another instance's look-ahead stream (the fp32 instance of a context borrows the fp64 one's: the two never run
  // together, and the stream's hardware queue was chosen at context creation - profiles/r05_hw_queues.txt)
  void borrow_side_stream(hipStream_t s);
  void side_begin() override;
  void side_end(int ch = 0) override;
  void side_join(int ch = 0) override;
  void side_cancel() noexcept override { if (in_side) { stream = main_stream; in_side = false; } }

  void gemm(int nb, int m, int n, int k, double alpha, Mat A, Mat B, double beta, Mat C, int flags,
            double diag_add) override;
  void trsm_panel(int nb, int m, int n, Mat L, const real* inv, long inv_stride, Mat B, int mode) override;
  void potrf_leaf(int nb, Mat Ablk, real* inv, long inv_stride, int pivot_base, int* info) override;
  void potrf_leaf2(int nb, Mat Ablk, real* inv, long inv_stride, int pivot_base, int* info) override;
  bool potrf_diag256(int nb, Mat Ablk, real* inv, long inv_stride, int pivot_base, int* info) override;
  void copy(int nb, int n, Mat src, Mat dst) override;
  void tri_copy(int nb, int n, Mat src, Mat dst) override;
  void transpose_lower(int nb, int n, Mat W, Mat Wt) override;
  void init_state(int nb, int N, int Np, Mat Lam, real* theta1, real* mvec, double* ustate,
                  const real* y_in, const real* nu_in, real* y_pad, real* nu_pad, int* info, bool lam_identity = true) override;
  void kmat(int nb, int N, int Np, int R, const real* X, const double* ustate, int kernel, double jitter,
            Mat Lb, bool clear_upper = true, Mat Mb = Mat{nullptr, 0, 0}, double m_scale = 0.0,
            const real* m_diag = nullptr, const int* Rv = nullptr, const long* xoff = nullptr) override;
  void scale_rows_rsqrt(int nb, int n, Mat src, const real* nu, Mat dst) override;
  void lt_dvec(int nb, int n, Mat L, const real* v, const real* nu, double alpha, double beta,
               real* out) override;
  void lam_update(int nb, int n, Mat G, double gamma, Mat Lam, Mat Cb, Mat Gfull, int parts = 3) override;
  void form_phi(int nb, int n, Mat M, const real* u, const real* mvec, Mat Phi, double scale, double diag_add,
                double u_scale) override;
  void symv_lower(int nb, int n, Mat S, const real* v, real* out) override;
  void aug_set(int nb, int n, Mat rows, const real* vec, int zero_rows) override;
  void aug_get(int nb, int n, Mat rows, real* vec) override;
  void basis_rows_set(int nb, int N, int Np, Mat rows, const real* mu_pad, int in_padding) override;
  void basis_rows_get(int nb, int N, int Np, Mat rows, real* basis) override;
  void resid(int nb, int n, Mat L, const real* mvec, const real* y, real* a) override;
  void grad_adam(int nb, int N, int Np, int R, const real* X, int kernel, Mat Kbt, double* ustate, double lr,
                 const real* pvec = nullptr, double sym_gamma = 0.0) override;
  void gemv_n(int nb, int n, Mat A, const real* x, real* y) override;
  void row_sqsum(int nb, int n, Mat A, real* out) override;
  void elbo_combine(int nb, int N, int Np, const real* a, const real* f, const real* nu, const real* mvec,
                    Mat C, const double* logdetC, const double* trS, double* trace, int stride, int slot) override;
  void identity(int nb, int n, Mat A) override;
  void at_from_linv(int nb, int n, Mat L, Mat Linv, double jitter, Mat At) override;
  void diag_init(int nb, int n, const real* nu, double add, Mat dst) override;
  void extract(int nb, int N, Mat src, real* dst, int mode) override;
  void extract_vec(int nb, int N, int Np, const real* src, real* dst) override;
  void extract_diagonal(int nb, int N, Mat src, real* dst) override;
  void extract_cols(int nb, int N, Mat src, real* dst, int c0, int c1) override;
  void logdet(int nb, int N, Mat Lc, double* out) override;
  void info_fold(int nb, int* pair, int* info) override;
  void hyp_out(int nb, const double* ustate, double* hyp) override;
  void frob_normalise(int nb, int n, Mat M, Mat X, const double* cap = nullptr, double cap_scale = 0.0) override;
  void axpby_rel(int nb, int n, double a, Mat X, double b, Mat T, double* rel) override;
  void dot_all(int nb, int n, Mat A, Mat B, double* out) override;
  void read_back(const double* dev, double* host, int n) override;
  // ModelSimilarityWeight(mode='single') assembly (compact [M][N] inputs, [M][M] tables)
  void pair_terms(int M, int N, const double* mu, const double* sd, double* gap, double* diagterm);
  void w2_single_finish(int M, const double* tr, const double* nuc, const double* gap, const double* diagterm,
                        double* D, double* w);

  // not part of the driver interface: pad a compact [nb][N][N] matrix into a padded Mat (identity pad)
  void pad_in(int nb, int N, int Np, const double* src, Mat dst);
  // a caller-supplied compact lower factor as the operand of the blocked solves: padded lower image, inverted 64x64
  // diagonal blocks, right-hand-side rows (mode 0: x - mu, mode 1: the pair (1, mu)), log-density of solved rows
  void pad_lower(int nb, int N, int Np, const double* src, Mat dst);
  void tri_inv64(int nb, int Np, Mat Lp, real* inv, long inv_stride, int* info);
  // out[b][0] = max column sum, out[b][1] = max row sum of |A_b| over the lower triangle (out zeroed by the caller)
  void tri_absnorms(int nb, int n, Mat A, double* out);
  void solve_rows_set(int nb, int N, int Np, int nrows, int mode, const double* x, long x_stride, const double* mu, Mat rows);
  void rows_logprob(int nb, int N, int nrows, Mat rows, const double* logdet, double* out);
  void fill_random(double* dst, size_t count, unsigned seed);           // uniform [-1, 1), diagnostics only
  void pad_vec(int nb, int N, int Np, const double* src, double* dst);   // [nb][N] -> [nb][Np], zero pad

  // weights / weighted-posterior kernels (compact layouts, see include/bayens.h)
  // same rows from the precomputed basis [B][2][N] (L_c^-1 1, L_c^-1 mu); dots: scratch [B][3]
  // n_groups > 1: posterior b is evaluated at the observation set b % n_groups (obs [n_groups][R_obs][N])
  void loglik_rows_basis(int B, int N, int R_obs, const double* basis, const double* logdet, const double* obs,
                         double* ll_mean, double* dots, int n_groups = 1);
  void normal_loglik_rows(int B, int N, int R_obs, const double* mu, const double* scale,
                          const double* obs, double* ll_mean);
  void weights_normalise(int M, int P, const double* ll_mean, double c_std, double* w, double* w_raw);
  void w2_temporal(int M, int P, const double* mean, const double* var, double* w);
  void w2_spatial(int M, int T, int G, const double* mean, const double* V, double* w);
  void transpose_rows(int nb, int A, int B, const double* src, double* dst);
  void field_prologue(int M, int R, int N, int G, const double* raw, double* X, double* y_mean, double* y_var);
  void barycentre(int M, int P, const double* w, const double* mean, const double* var, double tol,
                  double init_var, double* mu_out, double* var_out, int* n_bad);
  void weighted_model_mean(int M, int R_max, const int* R, int P, const double* data, const double* w,
                           double* mu_out, double* var_out);
  void extract_diag(int B, int N, const double* cov, double* var);
  // DTW barycentre averaging of every member's realisations (dba.hip); scratch: sq [B][R][R], path [B][R][N][N] bytes,
  // jlo / jhi [B][R][N]
  void dba_batch(int B, int R, int N, const double* series, int n_iterations, double* center, int* medoid, double* sq,
                 signed char* path, int* jlo, int* jhi);
  // kind 0 = CRPS, 1 = KSD; score (nullable) receives the un-inverted scores
  void score_weights(int kind, int M, int P, int R_obs, const double* mean, const double* var, const double* obs,
                     double* w, double* score);
  void inverse_square_weights(int M, int R_max, const int* R, int P, const double* data, int R_obs, const double* obs,
                              double* w);
  void multi_model_mean(int M, int R_max, const int* R, int P, const double* data, double* mu_out, double* var_out);
};

}  // namespace BAYENS_NS
