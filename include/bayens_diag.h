/*
 * bayens_diag.h - diagnostic / benchmark entry points of libbayens.  NOT part of the drop-in boundary
 * (include/bayens.h): nothing here replaces a reference interface; these exist for the GPU unit tests of the device
 * primitives (tests/test_gpu_primitives.py), the profiling tools under tools/ and the measurements quoted in DESIGN.md.
 * They may change or disappear between versions.
 */
#ifndef BAYENS_DIAG_H
#define BAYENS_DIAG_H

#include "bayens.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- diagnostic entry points used by the GPU unit tests (device primitives, batched) ------- */
/* C = alpha * op(A) op(B) + beta * C on host buffers; flags as in csrc/prims.h */
int bayens_test_gemm(bayens_ctx* ctx, int nb, int m, int n, int k, double alpha,
                     const double* A, int lda, const double* B, int ldb, double beta,
                     double* C, int ldc, int flags, double diag_add);
/* times `reps` launches of one GEMM shape on device-resident pseudo-random operands (HIP events on the ctx
 * stream); *ms_per_launch is the average.  Used by tools/gemm_bench.py and the PMC profiles. */
int bayens_bench_gemm(bayens_ctx* ctx, int nb, int m, int n, int k, int flags, int reps, double* ms_per_launch);
/* times the 64x64 diagonal-block kernel (copy + leaf per rep) and returns s_memtime stamps of its phases */
int bayens_bench_leaf(bayens_ctx* ctx, int nb, int reps, double* us_per_launch, unsigned long long* stamps16);
/* Diagnostic: nb device-generated SPD matrices of order n (a multiple of 64); us_per_factor = time of one batched
 * blocked Cholesky (the copy that restores the input between repetitions is timed separately and subtracted). */
int bayens_bench_potrf(bayens_ctx* ctx, int nb, int n, int reps, double* us_per_factor);
/* Diagnostic: times the blocked triangular solve  X op(L) = B  (mode 0: op(L) = L^T, 1: op(L) = L; rhs_lower: B lower
 * triangular, mode 1 only) on an m x n right-hand side against the factor of a device-generated SPD matrix of order n;
 * n <= 256 is one fused panel launch (trsm_panel_kernel), above it the recursion of csrc/driver.cpp. */
int bayens_bench_trsm(bayens_ctx* ctx, int nb, int m, int n, int mode, int rhs_lower, int reps, double* us_per_solve);
/* Diagnostic: what the fp64 matrix pipes sustain from registers alone (16 independent v_mfma_f64_16x16x4_f64
 * accumulators per wave, waves_per_simd = 1 or 2 waves on every SIMD, no memory traffic) - the measured ceiling the
 * GEMM's K loop is compared with. */
int bayens_bench_mfma_peak(bayens_ctx* ctx, int waves_per_simd, double* tflops);
/* Diagnostic: the same MFMA stream fed from LDS exactly as the GEMM's K loop feeds it (8 operand reads per 16 MFMAs,
 * two workgroups per CU).  mode bits add the rest of a K step one by one: 1 = a workgroup barrier per 64 MFMAs,
 * 2 = the eight transposing LDS stores of the next tile, 4 = eight 16-byte global loads feeding those stores. */
int bayens_bench_mfma_lds(bayens_ctx* ctx, int mode, double* tflops);
/* Diagnostic: shader cycles per repetition of (variant & 255) 0 = the MFMA-based 16x16 diagonal factorisation of the fused
 * small-member kernel, 1 = 64 dependent v_fma_f64, 2 = 16 dependent v_rsq_f64 (+fma), 3 = 16 dependent refined rsqrt,
 * 4 = 16 dependent v_mfma_f64_16x16x4, 5 = 16 x (v_readlane of an MFMA result feeding the next MFMA), 6 = one tile
 * through LDS; one workgroup of `waves` waves, wave 0 measured; variant | 256: the other waves keep the matrix pipes busy.
 * 7 / 8: EVERY wave runs tile products of 6 steps (7: operands from LDS through the kernel's product loop, 8: 24 MFMAs
 * from registers - the pipe-bound reference); cycles per product. */
int bayens_bench_diag16(bayens_ctx* ctx, int variant, int waves, int reps, double* cycles);
/* Diagnostic: phase stamps (shader cycle counter) of the fused small-member fit kernel (csrc/small.hip): enable = 1
 * makes every later small fit record the phase boundaries of workgroup 0's last member / iteration; stamps32 (nullable)
 * receives the 32 recorded values; enable = 0 switches the recording off again. */
int bayens_small_stamps(bayens_ctx* ctx, int enable, unsigned long long* stamps32);
/* ... and the per-wave stamps of the same recording: stamps128[wave * 16 + k], k-th WSTAMP of wave `wave` (0..7) of workgroup 0
 * (which phases carry them is a property of the build; tools/small_phases.py prints what it finds). */
int bayens_small_wave_stamps(bayens_ctx* ctx, unsigned long long* stamps128);
/* Diagnostic: cycle-counter stamps [8 waves][16] of workgroup 0 of the last launch of the fused 256 x 256 diagonal-block
 * Cholesky (csrc/potrf256.hip): 0 start, 1 inputs in, 3 first diagonal tile factored, 4 / 5 after block columns 0 / 7,
 * 6 after the last, 7 outputs issued, 9 block inverses done (tools/potrf256_phases.py). */
int bayens_potrf256_stamps(bayens_ctx* ctx, unsigned long long* stamps128);
/* in-place blocked Cholesky (lower) of [nb][n][n], n % 64 == 0 */
int bayens_test_potrf(bayens_ctx* ctx, int nb, int n, double* A, int* info);
/* B <- B * L^-T (mode 0) or B * L^-1 (mode 1), L [nb][n][n] lower (already factored), B [nb][m][n] */
int bayens_test_trsm(bayens_ctx* ctx, int nb, int m, int n, const double* L, double* Bm, int mode);

/* Members with 94 < N <= 254 time points (the reference's N = 165 annual series) have two routes through
 * bayens_gp_fit_batch: the general path (batched launches shared by all members) and the fused mid-size kernel
 * (csrc/medium.hip: one workgroup per member, matrices in an L2-resident scratch).  By default the route is chosen by size
 * (N <= 190: always the fused kernel; above: from 128 members on).  min_batch > 0 overrides: the fused kernel is used from
 * min_batch members on (1 = always, a huge value = never - what the parity tests use to hold the two routes against each
 * other); 0 restores the default. */
int bayens_set_medium_min_batch(bayens_ctx* ctx, int min_batch);
/* route = 1 sends EVERY fp64 fit of this context through the general path (csrc/driver.cpp) whatever its size, 0 restores the
 * choice by size (fused kernels for N <= 254).  The parity tests use it to hold all routes to the same fixtures in one
 * process (the environment switches BAYENS_NO_SMALL / BAYENS_NO_MEDIUM are read once per process). */
int bayens_set_fit_route(bayens_ctx* ctx, int route);
/* Which fp64 product kernel a launch takes: a launch of at most max_workgroups workgroups of the 128 x 128 tiling goes to the
 * 64 x 64-tile kernel (csrc/gemm.hip gemm_tile64).  -1 restores the measured defaults, 0 = never, a huge value = always.  The
 * primitive tests hold both kernels to the same cases with it. */
int bayens_set_gemm_tile64_max(bayens_ctx* ctx, int max_workgroups);
/* Number of bayens_loglik_rows calls on this context whose one-launch chained substitution gave up (a consumer workgroup ran
 * out of polls - see the note at bayens_loglik_rows in bayens.h) and were recomputed through the blocked solve.  The results of
 * such a call are correct; its run time is that of the fallback.  bench.py reports the count beside its timing of the entry. */
int bayens_chain_stalls(bayens_ctx* ctx, long* count);

/* Diagnostic: the pack -> [ncclAllGather image] -> unpack index arithmetic of bayens_allgather_member_rows without a
 * communicator: the host tables ll / mean / var [M][P] are split over `nranks` fictitious ranks exactly as
 * bayens_shard_bounds deals them, every rank's block goes through pack_rows_kernel into its slot of the gathered image
 * ([nranks][ceil(M/nranks)][3][P], unused rows poisoned with NaN), and unpack_rows_kernel rebuilds the three [M][P]
 * tables, which are returned.  (ensembles/weights.py:118-123 is the one coupling point this exchange serves.) */
int bayens_test_member_rows_exchange(bayens_ctx* ctx, int M, int P, int nranks, const double* ll, const double* mean,
                                     const double* var, double* ll_out, double* mean_out, double* var_out);

#ifdef __cplusplus
}
#endif
#endif /* BAYENS_DIAG_H */
