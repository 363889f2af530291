/*
 * bayens.h - C ABI of libbayens (MI355X / gfx950), the drop-in boundary for the
 * GP-fit -> ensemble-weighting -> weighted-posterior hot path of mattramos/bayesian_ensembling.
 *
 * The reference has no FFI: its boundary is a duck-typed Python plug-in protocol
 * (SURVEY.md 8(b)).  Each entry point below names the reference interface whose arithmetic it
 * replaces; the Python classes in bayesian_ensembling_amd/ keep the reference signatures and call
 * these through ctypes.  Plain pointers and sizes only; no exceptions cross the boundary.
 *
 * Conventions
 *   - all matrices row-major, fp64 (the reference precision: ensembles/__init__.py:10)
 *   - every function returns 0 on success, >0 = hipError_t / ncclResult_t passthrough,
 *     <0 = domain error (BAYENS_E_*); bayens_last_error(ctx) holds the message
 *   - `loc` says where the caller's buffers live: BAYENS_HOST (the library stages them through
 *     HBM) or BAYENS_DEVICE (pointers from bayens_malloc, already resident in HBM)
 *   - one ctx per (process, device); calls on one ctx are not re-entrant
 */
#ifndef BAYENS_H
#define BAYENS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bayens_ctx bayens_ctx;

enum { BAYENS_HOST = 0, BAYENS_DEVICE = 1 };
/* kernel ids: 0 is the only kernel the reference instantiates (models.py:186) */
enum { BAYENS_MATERN32 = 0, BAYENS_MATERN52 = 1, BAYENS_RBF = 2 };
enum {
  BAYENS_E_ARG = -1,      /* bad argument */
  BAYENS_E_NOTPD = -2,    /* a Cholesky pivot was not positive; info[] holds member-wise pivots */
  BAYENS_E_NOGPU = -3,    /* no HIP device / kernel image not loadable */
  BAYENS_E_COMM = -4,     /* RCCL not initialised or not loadable */
  BAYENS_E_ALLOC = -5,
  BAYENS_E_NOCONV = -6,   /* an iteration (the polar iteration of bayens_w2_single) did not converge: outputs invalid */
  BAYENS_E_INTERNAL = -7  /* an internal invariant failed (not raised by a stalled substitution chain: bayens_loglik_rows recovers) */
};

/* ---- context, memory and timing plumbing -------------------------------------------------- */
int bayens_device_count(int* count);                 /* visible HIP devices (does not initialise a context) */
int bayens_ctx_create(int device, bayens_ctx** out);
int bayens_ctx_destroy(bayens_ctx* ctx);
const char* bayens_last_error(bayens_ctx* ctx);
const char* bayens_version(void);
int bayens_device_info(bayens_ctx* ctx, char* name, int name_len, int* compute_units, size_t* hbm_bytes);
int bayens_malloc(bayens_ctx* ctx, size_t bytes, void** dptr);
int bayens_free(bayens_ctx* ctx, void* dptr);
int bayens_memcpy_h2d(bayens_ctx* ctx, void* dst, const void* src, size_t bytes);
int bayens_memcpy_d2h(bayens_ctx* ctx, void* dst, const void* src, size_t bytes);
/* device-to-device copy queued on the ctx stream (no host wait) */
int bayens_memcpy_d2d(bayens_ctx* ctx, void* dst, const void* src, size_t bytes);
int bayens_sync(bayens_ctx* ctx);
/* HIP-event timer on the ctx stream (the stream every kernel of this library is launched on) */
int bayens_timer_start(bayens_ctx* ctx);
int bayens_timer_stop(bayens_ctx* ctx, double* elapsed_ms);
/* per-kernel accounting (HIP events around every launch; off by default).  family 0 = the whole GEMM group,
 * 1/2/3 = gemm_f64_mfma<false,true> (NT) / <false,false> (NN) / <true,false> (TN), 4 = gemm_panel64,
 * 5 = trsm_panel_kernel, 6 = gemm_tile64 (the 64 x 64-tile form small outputs take, any layout).  Returns summed device ms,
 * algorithmic flop, launches. */
int bayens_profile_enable(bayens_ctx* ctx, int on);
int bayens_profile_get(bayens_ctx* ctx, int family, double* ms, double* flop, long* launches);
/* algorithmic operand bytes of the same launches (operands read once, computed part of C written, and read when
 * accumulating): the figure HBM traffic from the PMC counters is compared with */
int bayens_profile_bytes(bayens_ctx* ctx, int family, double* bytes);

/* text table of the profiled GEMM-group launches aggregated by (family, m, n, k, flags): calls, ms, TFLOP/s.
 * flags: the GEMM_* bits, +256 when the launch accumulates into C; for family 5 the solve mode (0 RLT, 1 RLN). */
int bayens_profile_shapes(bayens_ctx* ctx, char* buf, int buf_len);
/* text table "stage  ms" of the fit stages recorded (HIP events) while profiling was enabled, one stream */
int bayens_profile_stages(bayens_ctx* ctx, char* buf, int buf_len);

/* Number of member groups (HIP streams + host threads) a fit batch of the general route is split over; 1 = one stream (plus its
 * look-ahead stream); 0 (the default; env BAYENS_STREAMS at context creation overrides) = the library's choice, which is one
 * group.  At most 16.  Measured on MI355X at 32 x 2000 (profiles/r05_hw_queues.txt): 57.7 / 58.6 / 69 ms per step with 1 / 2 / 4
 * groups once every stream and its look-ahead stream sit on different hardware queues (the context creates them in a fixed
 * order for that).  The result does not depend on the grouping (tests/test_gpu_fit_parity.py::
 * test_member_groups_do_not_change_the_result). */
int bayens_set_groups(bayens_ctx* ctx, int n_groups);

/* ---- the hot path ------------------------------------------------------------------------- */

/* GPDTW1D.fit after the DTW-barycentre stage, for B independent members at once
 * (reference: ensembles/models.py:179-220 called from ModelCollection.fit, ensembles/data.py:385-395;
 *  third-party arithmetic restated in SURVEY.md Appendix A), followed by
 * Distribution.__post_init__ -> distrax MultivariateNormalFullCovariance (ensembles/data.py:38-39):
 *   X       [B][N][R]   realisation_set.T per member (models.py:182)
 *   y_mean  [B][N]      DBA target (models.py:176-178; boundary input)
 *   y_var   [B][N]      np.var(realisation_set, axis=0) (models.py:179)
 *   n_iters             n_optim_nits; gamma 0.5 (models.py:191); lr 0.01 (models.py:192); jitter 1e-6
 * outputs (any may be NULL except mu/cov):
 *   mu [B][N], cov [B][N][N] (predict_f full_cov + diag(y_var), models.py:217-220),
 *   cov_chol [B][N][N] lower Cholesky of cov, logdet [B] = sum(log(diag(chol))),
 *   hyp [B][2] final (variance, lengthscale), info [B] 0 or 1-based failing pivot,
 *   ll_basis [B][2][N] = (L_c^-1 1, L_c^-1 mu): the two vectors a later log_prob at a constant vector needs
 *   (weights.py:98-100); they fall out of the bordered Cholesky of cov at no extra pass.
 *   loss_trace [B][ceil(n_iters/25)]: the training loss (-ELBO) the reference samples after the iterations
 *   0, 25, 50, ... (models.py:211-215: evaluated after that iteration's Adam step, i.e. at the updated
 *   hyper-parameters with the variational parameters of that iteration's natural-gradient step).  Costs one full
 *   triangular solve and one triangular inverse per sample; pass NULL to skip it. */
int bayens_gp_fit_batch(bayens_ctx* ctx, int loc, int B, int N, int R,
                        const double* X, const double* y_mean, const double* y_var,
                        int kernel, int n_iters, double gamma, double lr, double jitter,
                        double* mu, double* cov, double* cov_chol, double* logdet,
                        double* hyp, double* loss_trace, int* info, double* ll_basis);

/* The same for members whose realisation counts DIFFER - what ModelCollection.fit actually meets (ensembles/data.py:385-395 loops
 * over members one after the other; the reference's pickled collections hold R in {25, 9, 16}, {25, 9, 10}, {25, 25, 5, 6, 3, 5},
 * experiments/data R = 3 ... 50) and what experiments/full_experiment_script.py:87-113 does fourteen times in a row: every member
 * of equal N goes into ONE launch, whatever its R and whichever collection it belongs to ("a per-member R[b] array").
 *   R      [B]  realisations of member b (HOST array whatever `loc` says)
 *   x_off  [B]  offset, in doubles, of member b's [N][R[b]] block (realisation_set.T, models.py:182) inside X (HOST array);
 *               NULL: the blocks lie one behind the other in member order
 *   X           the members' blocks; everything else as in bayens_gp_fit_batch.
 * In the fused kernels (N <= 254) R is only the inner extent of the kernel-matrix tile loads, in the general path the
 * contraction length of the distance product: each member's result is bit for bit that of a uniform batch of its own R. */
int bayens_gp_fit_batch_ragged(bayens_ctx* ctx, int loc, int B, int N, const int* R, const size_t* x_off,
                               const double* X, const double* y_mean, const double* y_var,
                               int kernel, int n_iters, double gamma, double lr, double jitter,
                               double* mu, double* cov, double* cov_chol, double* logdet,
                               double* hyp, double* loss_trace, int* info, double* ll_basis);

/* [nb][A][B] -> [nb][B][A] (out of place).  The per-gridpoint field keeps its tables fit-major ([member][grid point][time]);
 * the reference's consumers flatten (time, lat, lon) in the data's order (ensembles/ensemble_scheme.py:51-52,
 * weights.py:93-100 `.ravel()`): this is the device-side change of layout between the two. */
int bayens_transpose_rows(bayens_ctx* ctx, int loc, int nb, int A, int B, const double* src, double* dst);

/* The prologue of GPDTW1D.fit (ensembles/models.py:175-182: y_var = np.var(realisation_set, axis=0), X = realisation_set^T;
 * y_mean = the mean over realisations, the documented stand-in for the tslearn DTW barycentre of models.py:176-178) at
 * every grid point of M fields raw [M][R][N][G] (realisation, time, flattened space - the array as the caller holds it):
 * X [M*G][N][R], y_mean / y_var [M*G][N], fit index b = m * G + g - the inputs of bayens_gp_fit_batch_reduced for a
 * per-gridpoint fit (BASELINE config 5). */
int bayens_field_prologue(bayens_ctx* ctx, int loc, int M, int R, int N, int G, const double* raw, double* X,
                          double* y_mean, double* y_var);

/* Small members (N <= 94, e.g. the reference's annual series N = 86, experiments/full_experiment_script.py:57,70, and
 * BASELINE.json config 5's per-gridpoint fits) are fitted by bayens_gp_fit_batch in ONE fused kernel, one workgroup per
 * member with every matrix resident in LDS (csrc/small.hip; BAYENS_NO_SMALL=1 forces the general path).  This entry is
 * the same fit with the REDUCED export the weighting path needs and nothing else: mu [B][N], var [B][N] = diag(cov)
 * (_dist.variance(), ensemble_scheme.py:64-65), logdet [B], hyp [B][2], ll_basis [B][2][N] (weights.py:98-100) - 2.8 KB
 * per member instead of the 59.9 KB of a full covariance.  Members with 94 < N <= 254 (the
 * reference's historical annual horizon N = 165, full_experiment_script.py:57) take the fused mid-size kernel (csrc/medium.hip)
 * with the same export; N > 254 (the reference's monthly series, N = 1032 / 1980: experiments/data) the general path - the
 * covariance is formed and factored in the fit's workspace (the log-determinant and the log-likelihood basis need it) and only
 * its diagonal leaves. */
int bayens_gp_fit_batch_reduced(bayens_ctx* ctx, int loc, int B, int N, int R,
                                const double* X, const double* y_mean, const double* y_var,
                                int kernel, int n_iters, double gamma, double lr, double jitter,
                                double* mu, double* var, double* logdet, double* hyp, int* info, double* ll_basis);
/* ... in fp32 (N <= 94: the fp32 instance of the fused small-member kernel; above that the fp32 general path - the mid-size kernel
 * is fp64; BASELINE config 5
 * x config 3's precision: per-gridpoint fields with GPDTW1D(per_gridpoint=True, dtype="float32"), 1e-3 relative).  float arrays
 * in and out; logdet / hyp stay fp64.  bayens_narrow_f64 / bayens_widen_f32 convert the tables of the fp64 prologue and the
 * fp64 weight / scheme entry points on the device. */
int bayens_gp_fit_batch_reduced_f32(bayens_ctx* ctx, int loc, int B, int N, int R,
                                    const float* X, const float* y_mean, const float* y_var,
                                    int kernel, int n_iters, double gamma, double lr, double jitter,
                                    float* mu, float* var, double* logdet, double* hyp, int* info, float* ll_basis);
int bayens_narrow_f64(bayens_ctx* ctx, int loc, size_t n, const double* src, float* dst);

/* The same fit in fp32 (BASELINE.json config 3; SURVEY 8(b) "dtype enum"): float arrays in and out, fp32 storage
 * and fp32 MFMA products throughout.  Kept in fp64: the hyper-parameters and Adam state (hyp), the squared
 * distances inside the kernel matrix, the 64x64 diagonal-block factorisations, and every reduction (logdet,
 * loss_trace).  Parity bar: 1e-3 relative against the fp64 oracle (BASELINE.json). */
int bayens_gp_fit_batch_f32(bayens_ctx* ctx, int loc, int B, int N, int R,
                            const float* X, const float* y_mean, const float* y_var,
                            int kernel, int n_iters, double gamma, double lr, double jitter,
                            float* mu, float* cov, float* cov_chol, double* logdet,
                            double* hyp, double* loss_trace, int* info, float* ll_basis);
/* ... and for members whose realisation counts differ (bayens_gp_fit_batch_ragged in fp32; x_off in floats). */
int bayens_gp_fit_batch_ragged_f32(bayens_ctx* ctx, int loc, int B, int N, const int* R, const size_t* x_off,
                                   const float* X, const float* y_mean, const float* y_var,
                                   int kernel, int n_iters, double gamma, double lr, double jitter,
                                   float* mu, float* cov, float* cov_chol, double* logdet, double* hyp,
                                   double* loss_trace, int* info, float* ll_basis);

/* Distribution.__post_init__ for a caller-supplied full covariance (ensembles/data.py:38-39):
 * chol [B][N][N] lower, logdet [B]; with mu [B][N] given, ll_basis [B][2][N] as above (both may be NULL). */
int bayens_mvn_factor_batch(bayens_ctx* ctx, int loc, int B, int N, const double* cov, const double* mu,
                            double* chol, double* logdet, double* ll_basis, int* info);

/* _dist.variance() of a full-covariance posterior = diag(cov)  (ensemble_scheme.py:64-65, weights.py:311) */
int bayens_extract_diag(bayens_ctx* ctx, int loc, int B, int N, const double* cov, double* var);
/* fp32 posteriors (bayens_gp_fit_batch_f32) into the fp64 tables the weight / scheme entry points take */
int bayens_extract_diag_f32(bayens_ctx* ctx, int loc, int B, int N, const float* cov, double* var);
int bayens_widen_f32(bayens_ctx* ctx, int loc, size_t n, const float* src, double* dst);

/* LogLikelihoodWeight._compute inner loops, MVN branch (ensembles/weights.py:88-104):
 * ll_mean[b][t] = mean_r log N(obs[r][t] * 1_N ; mu_b, cov_b)   obs [R_obs][N]
 * cov_chol [B][N][N]: lower factors, only the lower triangle is read (one chained launch over the compact factor).
 * Forward-progress note: the chained launch hands partial results from workgroup to workgroup (a consumer polls device memory
 * for the block its predecessor publishes).  Its block ids are ordered so that a workgroup only waits for workgroups with
 * SMALLER ids, and it assumes what the hardware dispatcher does in practice but no programming model promises - that
 * workgroups are dispatched in id order, so every producer is resident or finished when its consumer starts.  The wait is
 * bounded (2^21 polls); when it runs out - a GPU shared with other work can reorder dispatch - the entry does NOT fail: it
 * recomputes the rows through the blocked triangular solve (no cross-workgroup waits), counts the event
 * (bayens_chain_stalls, bayens_diag.h) and says so once on stderr.  Results are identical to rounding either way. */
int bayens_loglik_rows(bayens_ctx* ctx, int loc, int B, int N, int R_obs,
                       const double* mu, const double* cov_chol, const double* logdet,
                       const double* obs, double* ll_mean);

/* distrax MultivariateNormalFullCovariance.log_prob for ordinary arguments (the reference evaluates it with an (N,)
 * sample at ensembles/utils.py:139-155 on the diagonal type; this is the full-covariance form): log_prob[b][r] =
 * log N(x_r ; mu_b, L_b L_b^T), L = cov_chol.  x [B][n_x][N], or [n_x][N] shared by every member when shared_x != 0. */
int bayens_mvn_log_prob(bayens_ctx* ctx, int loc, int B, int N, int n_x, int shared_x, const double* mu,
                        const double* cov_chol, const double* logdet, const double* x, double* log_prob);

/* the same rows from the basis a fit / factorisation returned (no triangular solve, no pass over chol) */
int bayens_loglik_rows_basis(bayens_ctx* ctx, int loc, int B, int N, int R_obs, const double* ll_basis,
                             const double* logdet, const double* obs, double* ll_mean);

/* the same for a FIELD of posteriors (BASELINE.json config 5: one GP per grid point and member):
 * B = M * n_groups posteriors laid out [M][n_groups], posterior b is evaluated at the observations of its own grid
 * point b % n_groups, obs [n_groups][R_obs][N]; ll_mean [B][N] is then the [M][n_groups * N] table that
 * bayens_weights_normalise / bayens_barycentre take with P = n_groups * N. */
int bayens_loglik_rows_basis_grouped(bayens_ctx* ctx, int loc, int B, int N, int R_obs, int n_groups,
                                     const double* ll_basis, const double* logdet, const double* obs, double* ll_mean);

/* same, dx.Normal branch (ensembles/weights.py:95-96): elementwise log N(obs; mu, scale) */
int bayens_normal_loglik_rows(bayens_ctx* ctx, int loc, int B, int N, int R_obs,
                              const double* mu, const double* scale, const double* obs, double* ll_mean);

/* LogLikelihoodWeight + Barycentre in one call - the reference's weighted posterior (weights.py:87-131 then
 * ensemble_scheme.py:43-81): rows from the basis of bayens_gp_fit_batch, marginal variances from cov [M][N][N],
 * w [M][N] (nullable), barycentre mu_out / var_out [N].  One host synchronisation instead of one per step. */
int bayens_weighted_posterior(bayens_ctx* ctx, int loc, int M, int N, int R_obs, const double* ll_basis,
                              const double* logdet, const double* obs, const double* mu, const double* cov,
                              double c_std, double tol, double init_var, double* w, double* mu_out,
                              double* var_out, int* n_not_converged);

/* standardise + normalise over the model axis (ensembles/weights.py:107,118-123):
 * w_raw = exp(c_std * ll_mean), w = w_raw / sum_m w_raw.  scheme 0 = exp.  w_raw may be NULL. */
int bayens_weights_normalise(bayens_ctx* ctx, int loc, int M, int P, const double* ll_mean,
                             double c_std, int scheme, double* w, double* w_raw);

/* ModelSimilarityWeight mode='temporal' (ensembles/weights.py:302-325,331 with
 * ensembles/wasserstein.py:21-47 on 1x1 dx.Normal(mean, scale=variance)). mean,var,w [M][P]. */
int bayens_w2_temporal(bayens_ctx* ctx, int loc, int M, int P, const double* mean,
                       const double* var, double* w);

/* ModelSimilarityWeight mode='spatial', dx.Normal posteriors (ensembles/weights.py:267-300,331 with
 * ensembles/wasserstein.py:21-47, full_cov=False): per grid point p, over the T time steps,
 * D[i,j,p] = ||mean_i[:,p] - mean_j[:,p]||_2 + sum_t (scale_i[t,p] - scale_j[t,p])^2, scale = Distribution.variance (the
 * reference passes it as dx.Normal's scale); w[i,p] = nanmean_j D / sum_i.  mean, scale [M][T][G] (G = lat * lon grid
 * points, contiguous); w [M][G]. */
int bayens_w2_spatial(bayens_ctx* ctx, int loc, int M, int T, int G, const double* mean, const double* scale, double* w);

/* ModelSimilarityWeight mode='single' (ensembles/weights.py:240-265,331 + ensembles/wasserstein.py:10-47):
 * D[i][j] = ||mu_i - mu_j||_2 + tr(S_i + S_j - 2 (S_i^1/2 S_j S_i^1/2)^1/2), w_i = mean_j D[i][j] / sum.
 * Full-covariance posteriors: chol [M][N][N] are the lower factors (S = L L^T); the trace of the matrix square
 * root is the nuclear norm of L_j^T L_i, obtained from a polar iteration (no SVD).  Diagonal posteriors
 * (dx.Normal): pass chol = NULL and sd [M][N] = the distributions' stddev().  D [M][M] may be NULL. */
int bayens_w2_single(bayens_ctx* ctx, int loc, int M, int N, const double* mu, const double* chol,
                     const double* sd, double* w, double* D);
/* The same table with the M(M-1)/2 polar iterations dealt over the ranks of the communicator (bayens_comm_init):
 * every rank passes ALL M members (all-gather the factors first, e.g. bayens_allgather_rows with P = N*N), computes
 * the pairs of its anchor members, and one ncclAllReduce completes the table on every rank. */
int bayens_w2_single_sharded(bayens_ctx* ctx, int loc, int M, int N, const double* mu, const double* chol,
                             double* w, double* D);

/* Barycentre._compute (ensembles/ensemble_scheme.py:52-78) + gaussian_barycentre
 * (ensembles/wasserstein.py:61-100).  w, mean, var [M][P]; mu_out, var_out [P] (var_out = sigma*^2). */
int bayens_barycentre(bayens_ctx* ctx, int loc, int M, int P, const double* w, const double* mean,
                      const double* var, double tol, double init_var,
                      double* mu_out, double* var_out, int* n_not_converged);

/* WeightedModelMean._compute (ensembles/ensemble_scheme.py:136-160).
 * data [M][R_max][P] (rows r >= R[m] ignored), R [M], w [M][P]. */
int bayens_weighted_model_mean(bayens_ctx* ctx, int loc, int M, int R_max, const int* R, int P,
                               const double* data, const double* w, double* mu_out, double* var_out);

/* ---- callers of the path ("next" rows): the other weight plug-ins and MultiModelMean ---- */
/* CRPSWeight (kind 0, ensembles/weights.py:444-515) / KSDWeight (kind 1, weights.py:336-441) against
 * dx.Normal(mean, scale = variance) per point; mean, var, w [M][P], obs [R_obs][P]; score [M][P] or NULL. */
int bayens_score_weights(bayens_ctx* ctx, int loc, int kind, int M, int P, int R_obs, const double* mean,
                         const double* var, const double* obs, double* w, double* score);
/* InverseSquareWeight (ensembles/weights.py:157-169); data [M][R_max][P], R [M] */
int bayens_inverse_square_weights(bayens_ctx* ctx, int loc, int M, int R_max, const int* R, int P,
                                  const double* data, int R_obs, const double* obs, double* w);
/* MultiModelMean (ensembles/ensemble_scheme.py:108-128) */
int bayens_multi_model_mean(bayens_ctx* ctx, int loc, int M, int R_max, const int* R, int P, const double* data,
                            double* mu_out, double* var_out);

/* The DTW-barycentre front-end of GPDTW1D with the semantics of the reference's own DBA, ensembles/dtwa.py
 * (performDBA :6-20: medoid start, n_iterations DBA updates), for B members at once: series [B][R][N] (R <= 50
 * realisations of equal length), center [B][N], medoid [B] (index of the starting series, may be NULL). */
int bayens_dba_batch(bayens_ctx* ctx, int loc, int B, int R, int N, const double* series, int n_iterations,
                     double* center, int* medoid);

/* ---- multi-GPU: RCCL all-gather of per-member rows ------------------------ */
/* load-only probe: 0 when librccl can be opened and carries the symbols this library uses (no communicator, no
 * ncclUniqueId, no bootstrap listener), BAYENS_E_COMM with the loader's message otherwise */
int bayens_comm_available(bayens_ctx* ctx);
int bayens_comm_unique_id(bayens_ctx* ctx, void* id128);            /* 128-byte ncclUniqueId, rank 0 */
int bayens_comm_init(bayens_ctx* ctx, int rank, int nranks, const void* id128);
int bayens_comm_destroy(bayens_ctx* ctx);
/* rank / size of the communicator (nranks 0 = none) and the path of the RCCL library that was loaded */
int bayens_comm_info(bayens_ctx* ctx, int* rank, int* nranks, char* lib_path, int lib_path_len);
/* local [M_loc][P] device rows -> all [nranks*M_loc][P] device rows (ncclAllGather on the ctx stream).  Every rank
 * must pass the same M_loc and all_dev must hold nranks * M_loc rows. */
int bayens_allgather_rows(bayens_ctx* ctx, const double* local_dev, double* all_dev, int M_loc, int P);
/* Block partition of M members over nranks ranks (the first M % nranks ranks own one member more): members
 * [*lo, *hi) belong to `rank`.  Pure host arithmetic, no context needed. */
int bayens_shard_bounds(int M, int rank, int nranks, int* lo, int* hi);
/* The exchange step of the member-sharded weighted posterior (the reference couples the members only in
 * the normalisation ensembles/weights.py:118-123 and the barycentre ensemble_scheme.py:52-78): this rank's M_loc =
 * hi - lo rows of mean log-likelihood, posterior mean and posterior variance ([M_loc][P] each, device) -> the full
 * [M][P] tables on every rank, in ONE ncclAllGather of packed [ceil(M/nranks)][3][P] rows.  M need not divide evenly. */
int bayens_allgather_member_rows(bayens_ctx* ctx, int M, int P, int M_loc, const double* ll_loc, const double* mean_loc,
                                 const double* var_loc, double* ll_all, double* mean_all, double* var_all);
/* all-reduce(max) of one host double over the communicator (through HBM, ncclAllReduce); returns after the ctx stream
 * has drained on every rank, so it is also the barrier of the multi-GPU timing protocol */
int bayens_comm_max(bayens_ctx* ctx, double* value);

#ifdef __cplusplus
}
#endif
#endif /* BAYENS_H */
