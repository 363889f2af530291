// medium.hip - the whole GP fit of a MID-SIZE member (94 < N <= 254 time points) in ONE kernel, one workgroup per
// member: the reference's real workload (historical annual series N = 165 fitted with n_optim_nits = 2000-2500,
// experiments/full_experiment_script.py:57,87-113; pre_fitting_cmip6models.py:76-77), which through the general path is
// ~30 dependent launches per optimiser iteration over matrices padded to 192 - launch / latency bound at ~0.5 ms per
// iteration whatever the batch.
//
// Same arithmetic and the same phases as the fused small-member kernel (small.hip, whose tile helpers this file
// includes: 16x16 tiles, tile-packed lower triangles, explicit triangular inverses riding along with the blocked
// factorisations, MFMA-based 16x16 diagonal factorisation) - reference ensembles/models.py:185-220 + data.py:38-39,
// SURVEY.md Appendix A.  What differs is where the matrices live: at NT = 11 tiles per side a tile-packed triangle is
// 132 KB, so the matrices cannot share the CU's 160 KB of LDS.  They live in a per-workgroup scratch in HBM that stays
// L2-resident (eight triangles, ~1 MB per member at N = 165), every tile operand is fetched from there, LDS keeps the
// vectors and the staging of the prediction's covariance rows, and the workgroup barrier also drains the vector-memory
// counter so that one wave's tile stores are visible to the others (all waves of a workgroup share the CU's L1).
// With no LDS limit there is no need to hold Lam / G / S in registers: they are scratch triangles too, which also
// makes the tile count a RUN-TIME value (one instantiation for every N).
#include <cmath>
#include <cstdlib>

#define BAYENS_SMALL_HELPERS_ONLY
#include "small.hip"
#include "medium.h"

#ifdef BAYENS_F32
#error "medium.hip is built for the fp64 fit only"
#endif

namespace bayens {
namespace mediumn {

using namespace smalln;

// workgroup barrier that also publishes global stores: vector-memory loads AND stores return through vmcnt on gfx9
#define GSYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define MSTAMP(k) do { if (g.stamps && blockIdx.x == 0 && tid == 0) g.stamps[k] = __builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ void unpack_step(int& i, int& j, int step) {      // advance (i, j) by `step` packed positions
  j += step;
  while (j > i) { j -= i + 1; ++i; }
}

// (L x)[16 i + lr], (L^T x)[16 j + lr], (S x)[16 i + lr] on tile-packed triangles; valid in lanes lk == 0
__device__ __forceinline__ double mvr_lower(const double* Lb, const double* x, int i, const Lane& o) {
  double s = 0.0;
#pragma unroll 1
  for (int k = 0; k <= i; ++k) {
    const double* t = Lb + lt(i, k) * TS;
#pragma unroll
    for (int q = 0; q < 4; ++q) s = fma(t[o.rc[q]], x[16 * k + o.kq[q]], s);
  }
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}
__device__ __forceinline__ double mvr_lower_t(const double* Lb, const double* x, int j, int NT, const Lane& o) {
  double s = 0.0;
#pragma unroll 1
  for (int k = j; k < NT; ++k) {
    const double* t = Lb + lt(k, j) * TS;
#pragma unroll
    for (int q = 0; q < 4; ++q) s = fma(t[o.cr[q]], x[16 * k + o.kq[q]], s);
  }
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}
__device__ __forceinline__ double mvr_sym(const double* Sb, const double* x, int i, int NT, const Lane& o) {
  double s = 0.0;
#pragma unroll 1
  for (int k = 0; k < NT; ++k) {
    const double* t = Sb + (k <= i ? lt(i, k) : lt(k, i)) * TS;
#pragma unroll
    for (int q = 0; q < 4; ++q) s = fma(t[k <= i ? o.rc[q] : o.cr[q]], x[16 * k + o.kq[q]], s);
  }
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}

// Blocked Cholesky of the tile-packed triangle A in place (run-time tile count), D[k] = inverse of diagonal tile k;
// WITH_INV: the inverse of the factor is formed alongside in Xb (block forward substitution).  Same two phases per
// block column as smalln::potrf_lds - [panel, every wave] [wave 0: next diagonal tile | waves 1..7: trailing update,
// inverse updates, extra(k)] - with the barriers publishing global stores.
template <bool WITH_INV, class Extra>
__device__ __forceinline__ int potrf_tiles(int NT, double* A, double* D, double* Xb, const Lane& o, int wave, int lane, int bad,
                                           Extra extra) {
  if (wave == 0) diag16(ld_tile_sym64(A, o), A, D, lane, 0, bad, o);
  GSYNC();
#pragma unroll 1
  for (int k = 0; k < NT; ++k) {
    const int below = NT - 1 - k;
    const double* dk = D + k * TS;
#pragma unroll 1
    for (int q = wave; q < below + (WITH_INV ? k + 1 : 0); q += NW) {
      if (q < below) {                                    // panel: L_ik = A_ik inv_kk^T
        double* t = A + lt(k + 1 + q, k) * TS;
        v4d acc = {0, 0, 0, 0};
        acc = mma<false, true>(acc, t, dk, o);
        st_tile(t, o, acc);
      } else {
        const int i = q - below;
        double* t = Xb + lt(k, i) * TS;
        if (i < k) {                                      // X_ki = inv_kk T_ki
          v4d acc = {0, 0, 0, 0};
          acc = mma<false, false>(acc, dk, t, o);
          st_tile(t, o, acc);
        } else {
          st_tile(t, o, ld_tile(dk, o));                  // X_kk = inv_kk
        }
      }
    }
    GSYNC();
    if (below == 0) break;
    if (wave == 0) {                                      // look-ahead: next diagonal tile, then its factorisation
      double* t = A + lt(k + 1, k + 1) * TS;
      v4d acc = ld_tile_sym(t, o);
      acc = mma<false, true>(acc, A + lt(k + 1, k) * TS, A + lt(k + 1, k) * TS, o, -1);
      diag16(acc, t, D + (k + 1) * TS, lane, 16 * (k + 1), bad, o);
    } else {
      int p = 0;
#pragma unroll 1
      for (int i = k + 2; i < NT; ++i)
#pragma unroll 1
        for (int j = k + 1; j <= i; ++j)
          if (p++ % (NW - 1) == wave - 1) {               // A_ij -= L_ik L_jk^T
            double* t = A + lt(i, j) * TS;
            v4d acc = ld_tile(t, o);
            acc = mma<false, true>(acc, A + lt(i, k) * TS, A + lt(j, k) * TS, o, -1);
            st_tile(t, o, acc);
          }
      if (WITH_INV) {
#pragma unroll 1
        for (int j = k + 1; j < NT; ++j)
#pragma unroll 1
          for (int i = 0; i <= k; ++i)
            if (p++ % (NW - 1) == wave - 1) {             // T_ji -= L_jk X_ki   (first touched at k == i)
              double* t = Xb + lt(j, i) * TS;
              v4d acc = {0, 0, 0, 0};
              if (k > i) acc = ld_tile(t, o);
              acc = mma<false, false>(acc, A + lt(j, k) * TS, Xb + lt(k, i) * TS, o, -1);
              st_tile(t, o, acc);
            }
      }
      extra(k);
    }
    GSYNC();
  }
  if (wave != 0) extra(NT - 1);
  return bad;
}

__global__ __launch_bounds__(NTH, 2) void medium_fit_kernel(MediumArgs g) {
  const int NT = g.NT, NTT = NT * (NT + 1) / 2, NP = 16 * NT;
  double* lds = lds_base();
  double* ssq = lds;                    // |x_i|^2
  double* misc = ssq + NP;              // [0..7] optimiser state, [8] variance, [9] lengthscale, [10] 1/ls^2, [11] log det C, [16..47] reductions
  double* yv = misc + 64;               // y (0 in the padding)
  double* dv = yv + NP;                 // 1 / nu (1 in the padding)
  double* rs = dv + NP;                 // 1 / sqrt(nu)
  double* yd = rs + NP;                 // y / nu
  double* th = yd + NP;                 // theta1
  double* mv = th + NP;                 // m
  double* av = mv + NP;                 // L m - y (loss samples); the posterior mean (prediction)
  double* wv = av + NP;                 // L^T (y / nu) of this iteration (loss samples: diag(L S L^T))
  double* uv = wv + NP;                 // u = G m - L^T (y / nu)
  double* E0 = uv + NP;                 // staging of one tile row of A^T S (prediction), ping ...
  double* E1 = E0 + (size_t)NT * TS;    // ... pong
  // the workgroup's matrices: tile-packed lower triangles in an L2-resident HBM scratch
  double* sc = g.scratch + (size_t)blockIdx.x * g.scratch_stride;
  double* Lb = sc;                              // K0 + J I -> L ; later C^-1, W
  double* Cb = Lb + (size_t)NTT * TS;           // Lam -> C ; later Phi
  double* Wb = Cb + (size_t)NTT * TS;           // L^-1
  double* Gb = Wb + (size_t)NTT * TS;           // G = L^T D L ; in the prediction the bordered covariance and its factor
  double* LamB = Gb + (size_t)NTT * TS;         // Lam (the natural parameter that persists over the iterations)
  double* Sb = LamB + (size_t)NTT * TS;         // S = C^-T C^-1
  double* DLi = Sb + (size_t)NTT * TS;          // inverted diagonal tiles of L (of chol(cov) at the end)
  double* DCi = DLi + (size_t)NT * TS;          // inverted diagonal tiles of C
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Lane o;
  lane_init(o, lane);
  const int N = g.N, R = g.R;
  const double J = g.jitter;

#pragma unroll 1
  for (int b = blockIdx.x; b < g.B; b += gridDim.x) {
    const double* X = g.X + (long)b * N * R;
    int bad = 0;
    // ---- member set-up ----
    for (int t = tid; t < NP; t += NTH) {
      const double nu = t < N ? g.nu[(long)b * N + t] : 1.0;
      const double y = t < N ? g.y[(long)b * N + t] : 0.0;
      yv[t] = y; dv[t] = 1.0 / nu; rs[t] = 1.0 / sqrt(nu); yd[t] = y / nu;
      th[t] = 0; mv[t] = 0;
      double sq = 0.0;
      if (t < N)
        for (int d = 0; d < R; ++d) { const double v = X[(long)t * R + d]; sq = fma(v, v, sq); }
      ssq[t] = sq;
    }
    if (tid < 8) misc[tid] = tid < 2 ? 0.5413248546129181 /* softplus^-1(1) */ : 0.0;
    {
      int i = 0, j = 0;
      unpack_step(i, j, wave);
#pragma unroll 1
      for (int p = wave; p < NTT; p += NW) {
        v4d id;
#pragma unroll
        for (int r = 0; r < 4; ++r) id[r] = (i == j && o.kq[r] == o.lr) ? 1.0 : 0.0;
        st_tile(LamB + (size_t)p * TS, o, id);           // Lam = I: q_sqrt = I
        st_tile(Sb + (size_t)p * TS, o, id);             // S = I until the first natural-gradient step (n_iters = 0)
        unpack_step(i, j, NW);
      }
    }
    GSYNC();

    auto refresh_hypers = [&]() {
      if (tid == 0) {
        const double ls = softplus_d(misc[1]);
        misc[8] = softplus_d(misc[0]); misc[9] = ls; misc[10] = 1.0 / (ls * ls);
      }
      GSYNC();
    };
    // Lb <- tril(K0) + J I (identity in the padding)
    auto kmat = [&]() {
      const double var = misc[8], inv_ls2 = misc[10];
      int i = 0, j = 0;
      unpack_step(i, j, wave);
#pragma unroll 1
      for (int p = wave; p < NTT; p += NW) {
        const v4d r2 = r2_tile(X, N, R, inv_ls2, ssq, i, j, o);
        v4d kt;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * i + o.kq[r], col = 16 * j + o.lr;
          double k1, k2;
          kern_unit<false>(g.kernel, r2[r], k1, k2);
          double K = var * k1;
          if (row >= N || col >= N) K = row == col ? 1.0 : 0.0;
          else K = col <= row ? K + (row == col ? J : 0.0) : 0.0;
          kt[r] = K;
        }
        st_tile(Lb + (size_t)p * TS, o, kt);
        unpack_step(i, j, NW);
      }
      GSYNC();
    };
    // -ELBO sample (models.py:211-215) at the hypers whose factor L is in Lb, with the (m, S) of the last natural-gradient
    // step (S in Sb, log det C in misc[11]) - smalln::neg_elbo with run-time tile counts
    auto neg_elbo = [&](int slot) {
#pragma unroll 1
      for (int i = wave; i < NT; i += NW) {
        const double s = mvr_lower(Lb, mv, i, o);
        if (o.lk == 0) av[16 * i + o.lr] = s - yv[16 * i + o.lr];
        double fr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int j = 0; j <= i; ++j) {
          v4d acc = {0, 0, 0, 0};
          acc = acc_run(acc, 0, i + 1, [&](int k, int q) { return Lb[lt(i, k) * TS + o.rc[q]]; },
                        [&](int k, int q) { return k >= j ? Sb[lt(k, j) * TS + o.cr[q]] : Sb[lt(j, k) * TS + o.rc[q]]; });
          const v4d lt_ = ld_tile(Lb + lt(i, j) * TS, o);
#pragma unroll
          for (int r = 0; r < 4; ++r) fr[r] = fma(acc[r], lt_[r], fr[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double v = fr[r];
          v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
          if (o.lr == 0) wv[16 * i + o.kq[r]] = v;
        }
      }
      GSYNC();
      double part = 0.0;
      for (int t = tid; t < N; t += NTH) {
        const double nu = 1.0 / dv[t], a = av[t], m = mv[t];
        const double stt = Sb[lt(t >> 4, t >> 4) * TS + swz(t & 15, t & 15)];
        part += 0.5 * 1.8378770664093453 + 0.5 * log(nu) + 0.5 * (a * a + wv[t]) * dv[t] + 0.5 * (m * m + stt);
      }
      part = wave_sum(part);
      if (lane == 0) misc[16 + wave] = part;
      GSYNC();
      if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < NW; ++w) tot += misc[16 + w];
        g.loss[(long)b * g.loss_stride + slot] = tot - 0.5 * N + misc[11];
      }
      GSYNC();
    };

#pragma unroll 1
    for (int it = 0; it < g.n_iters; ++it) {
      MSTAMP(0);
      refresh_hypers();
      kmat();                                                              // K0 + J I at the current hypers  (models.py:209)
      MSTAMP(1);
      // L = chol(K0 + J I), L^-1 rides along into Wb; G = L^T D L = sum_k (D^1/2 L)_k*^T (D^1/2 L)_k* is accumulated block row by
      // block row behind the factorisation, every tile of Gb by the wave that owns it (waves 1..7)
      auto g_rows = [&](int k) {
        int i = 0, j = 0;
        unpack_step(i, j, wave - 1);
#pragma unroll 1
        for (int p = wave - 1; p < NTT; p += NW - 1) {
          if (i > k) break;                                                // packed order: every later tile has i > k too
          const double* ta = Lb + lt(k, i) * TS;
          const double* tb = Lb + lt(k, j) * TS;
          double* gt = Gb + (size_t)p * TS;
          v4d acc = {0, 0, 0, 0};
          if (k > i) acc = ld_tile(gt, o);                                 // first touched by block row k == i
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double scl = rs[16 * k + o.kq[q]];
            acc = MFMA(ta[o.cr[q]] * scl, tb[o.cr[q]] * scl, acc);
          }
          st_tile(gt, o, acc);
          unpack_step(i, j, NW - 1);
        }
      };
      bad = potrf_tiles<true>(NT, Lb, DLi, Wb, o, wave, lane, bad, g_rows);
      GSYNC();
      if (g.loss && it > 0 && (it - 1) % 25 == 0) neg_elbo((it - 1) / 25);  // after the Adam step = at these hypers
      MSTAMP(2);
      // ---- natural-gradient step: Lam <- (1-g) Lam + g (I + G) -> LamB and Cb; theta1 <- (1-g) theta1 + g L^T (y / nu)
      {
        int i = 0, j = 0;
        unpack_step(i, j, wave);
#pragma unroll 1
        for (int p = wave; p < NTT; p += NW) {
          v4d lam = ld_tile(LamB + (size_t)p * TS, o);
          const v4d gt = ld_tile(Gb + (size_t)p * TS, o);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            lam[r] = (1.0 - g.gamma) * lam[r] + g.gamma * (gt[r] + ((i == j && o.kq[r] == o.lr) ? 1.0 : 0.0));
          st_tile(LamB + (size_t)p * TS, o, lam);
          st_tile(Cb + (size_t)p * TS, o, lam);
          unpack_step(i, j, NW);
        }
      }
#pragma unroll 1
      for (int i = wave; i < NT; i += NW) {
        const double s = mvr_lower_t(Lb, yd, i, NT, o);
        if (o.lk == 0) {
          th[16 * i + o.lr] = (1.0 - g.gamma) * th[16 * i + o.lr] + g.gamma * s;
          wv[16 * i + o.lr] = s;               // L^T (y / nu): u = G m - L^T (y / nu) needs it, and L dies here
        }
      }
      GSYNC();
      MSTAMP(3);
      // C = chol(Lam) in Cb, C^-1 in Lb (L is dead); S = C^-T C^-1 = sum_k (C^-1)_k*^T (C^-1)_k* accumulated into Sb behind it
      auto s_rows = [&](int k) {
        int i = 0, j = 0;
        unpack_step(i, j, wave - 1);
#pragma unroll 1
        for (int p = wave - 1; p < NTT; p += NW - 1) {
          if (i > k) break;
          double* stl = Sb + (size_t)p * TS;
          v4d acc = {0, 0, 0, 0};
          if (k > i) acc = ld_tile(stl, o);
          acc = mma<true, false>(acc, Lb + lt(k, i) * TS, Lb + lt(k, j) * TS, o);
          st_tile(stl, o, acc);
          unpack_step(i, j, NW - 1);
        }
      };
      bad = potrf_tiles<true>(NT, Cb, DCi, Lb, o, wave, lane, bad, s_rows);
      GSYNC();
      MSTAMP(4);
      if (g.loss && it % 25 == 0) {                                         // log det C of this step, for its loss sample
        double s = 0.0;
        for (int t = tid; t < N; t += NTH) s += log(Cb[lt(t >> 4, t >> 4) * TS + swz(t & 15, t & 15)]);
        s = wave_sum(s);
        if (lane == 0) misc[32 + wave] = s;
        GSYNC();
        if (tid == 0) {
          double tot = 0.0;
          for (int w = 0; w < NW; ++w) tot += misc[32 + w];
          misc[11] = tot;
        }
        GSYNC();
      }
      // ---- m = S theta1 ; u = L^T D (L m - y) = G m - L^T (y / nu)
#pragma unroll 1
      for (int i = wave; i < NT; i += NW) {
        const double s = mvr_sym(Sb, th, i, NT, o);
        if (o.lk == 0) mv[16 * i + o.lr] = s;
      }
      GSYNC();
#pragma unroll 1
      for (int i = wave; i < NT; i += NW) {
        const double s = mvr_sym(Gb, mv, i, NT, o);
        if (o.lk == 0) uv[16 * i + o.lr] = s - wv[16 * i + o.lr];
      }
      GSYNC();
      MSTAMP(5);
      // ---- Phi(L^T Lbar) = tril(u m^T + G S), half the diagonal -> Cb (C is dead)
      {
        int i = 0, j = 0;
        unpack_step(i, j, wave);
#pragma unroll 1
        for (int p = wave; p < NTT; p += NW) {
          v4d acc = {0, 0, 0, 0};
          acc = acc_run(acc, 0, NT,
                        [&](int k, int q) { return k <= i ? Gb[lt(i, k) * TS + o.rc[q]] : Gb[lt(k, i) * TS + o.cr[q]]; },
                        [&](int k, int q) { return k >= j ? Sb[lt(k, j) * TS + o.cr[q]] : Sb[lt(j, k) * TS + o.rc[q]]; });
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rl = o.kq[r];
            double v = fma(uv[16 * i + rl], mv[16 * j + o.lr], acc[r]);
            if (i == j) v = rl < o.lr ? 0.0 : (rl == o.lr ? 0.5 * v : v);
            acc[r] = v;
          }
          st_tile(Cb + (size_t)p * TS, o, acc);
          unpack_step(i, j, NW);
        }
      }
      GSYNC();
      MSTAMP(6);
      // ---- W = Phi L^-1 (lower; Phi in Cb, L^-1 in Wb) -> Lb (C^-1 is dead)
      {
        int i = 0, j = 0;
        unpack_step(i, j, wave);
#pragma unroll 1
        for (int p = wave; p < NTT; p += NW) {
          v4d acc = {0, 0, 0, 0};
          acc = acc_run(acc, j, i + 1, [&](int k, int q) { return Cb[lt(i, k) * TS + o.rc[q]]; },
                        [&](int k, int q) { return Wb[lt(k, j) * TS + o.cr[q]]; });
          st_tile(Lb + (size_t)p * TS, o, acc);
          unpack_step(i, j, NW);
        }
      }
      GSYNC();
      MSTAMP(7);
      // ---- Kbar = L^-T W ; g = <Kbar, dK/dtheta> over the real part ; Adam   (models.py:210)
      {
        const double var = misc[8], ls = misc[9], inv_ls2 = misc[10];
        double gv = 0.0, gl = 0.0;
        int i = 0, j = 0;
        unpack_step(i, j, wave);
#pragma unroll 1
        for (int p = wave; p < NTT; p += NW) {
          v4d acc = {0, 0, 0, 0};
          acc = acc_run(acc, i, NT, [&](int k, int q) { return Wb[lt(k, i) * TS + o.cr[q]]; },
                        [&](int k, int q) { return Lb[lt(k, j) * TS + o.cr[q]]; });
          if (i != j)
            acc = acc_run(acc, i, NT, [&](int k, int q) { return Lb[lt(k, i) * TS + o.cr[q]]; },
                          [&](int k, int q) { return Wb[lt(k, j) * TS + o.cr[q]]; });
          const v4d r2 = r2_tile(X, N, R, inv_ls2, ssq, i, j, o);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * i + o.kq[r], col = 16 * j + o.lr;
            double k1, k2;
            kern_unit<true>(g.kernel, r2[r], k1, k2);
            if (row < N && col < N) { gv = fma(acc[r], k1, gv); gl = fma(acc[r], k2, gl); }
          }
          unpack_step(i, j, NW);
        }
        gv = wave_sum(gv);
        gl = wave_sum(gl);
        if (lane == 0) { misc[16 + wave] = gv; misc[24 + wave] = gl; }
        GSYNC();
        if (tid == 0) {
          double sv = 0.0, sl = 0.0;
          for (int w = 0; w < NW; ++w) { sv += misc[16 + w]; sl += misc[24 + w]; }
          double* u = misc;
          double gg[2];
          gg[0] = sv * (1.0 - exp(-var));
          gg[1] = sl * (kern_cl(g.kernel) * var / ls) * (1.0 - exp(-ls));
          const double t = u[6] + 1.0;
          u[6] = t;
          const double lr_t = g.lr * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t));
          for (int q = 0; q < 2; ++q) {               // TF-2.8 Keras Adam (models.py:192)
            u[2 + q] = 0.9 * u[2 + q] + 0.1 * gg[q];
            u[4 + q] = 0.999 * u[4 + q] + 0.001 * gg[q] * gg[q];
            u[q] -= lr_t * u[2 + q] / (sqrt(u[4 + q]) + 1e-7);
          }
        }
        GSYNC();
      }
      MSTAMP(8);
    }

    // ---- prediction at the final hypers with the (m, S) of the last step    (models.py:217-220)
    MSTAMP(9);
    refresh_hypers();
    kmat();
    bad = potrf_tiles<true>(NT, Lb, DLi, Wb, o, wave, lane, bad, NoExtra());         // Lb = L, Wb = L^-1, Sb = S
    GSYNC();
    if (g.loss && g.n_iters > 0 && (g.n_iters - 1) % 25 == 0) neg_elbo((g.n_iters - 1) / 25);
    MSTAMP(10);
    // mu = A^T m with A^T = K0 L^-T = L - J L^-T
#pragma unroll 1
    for (int i = wave; i < NT; i += NW) {
      const double s1 = mvr_lower(Lb, mv, i, o);
      const double s2 = mvr_lower_t(Wb, mv, i, NT, o);
      if (o.lk == 0) av[16 * i + o.lr] = s1 - J * s2;                       // av = mu (padded part 0)
    }
    // cov = diag(nu + J) - J^2 L^-T L^-1 + (A^T S) A, one tile row at a time; E_i* = A^T_i* S is staged in LDS (ping-pong), so
    // that tile row i of cov and the staging of row i + 1 share a phase.  The finished tiles go, bordered by the rows
    // (1^T, mu^T) in the padding, straight into Gb (G is dead) for the factorisation that follows.
    auto at_a = [&](int i, int k, int q) -> double {
      const bool low = k <= i;
      const double* t1 = low ? Lb + lt(i, k) * TS : Wb + lt(k, i) * TS;
      const double v1 = t1[low ? o.rc[q] : o.cr[q]];
      const double v2 = Wb[lt(i, i) * TS + o.cr[q]];
      return fma(k == i ? -J : 0.0, v2, (low ? 1.0 : -J) * v1);
    };
    auto stage_row = [&](int i, double* Eb) {
#pragma unroll 1
      for (int l = NW - 1 - wave; l < NT; l += NW) {
        v4d acc = {0, 0, 0, 0};
        acc = acc_run(acc, 0, NT, [&](int k, int q) { return at_a(i, k, q); },
                      [&](int k, int q) { return k >= l ? Sb[lt(k, l) * TS + o.cr[q]] : Sb[lt(l, k) * TS + o.rc[q]]; });
        st_tile(Eb + (size_t)l * TS, o, acc);
      }
    };
    stage_row(0, E0);
    GSYNC();
#pragma unroll 1
    for (int i = 0; i < NT; ++i) {
      double* Eb = (i & 1) ? E1 : E0;
#pragma unroll 1
      for (int j = wave; j <= i; j += NW) {
        v4d acc = {0, 0, 0, 0};
        acc = acc_run(acc, i, NT, [&](int k, int q) { return Wb[lt(k, i) * TS + o.cr[q]]; },
                      [&](int k, int q) { return Wb[lt(k, j) * TS + o.cr[q]]; });
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = acc[r] * (-J * J);
        acc = acc_run(acc, 0, NT, [&](int l, int q) { return Eb[(size_t)l * TS + o.rc[q]]; },
                      [&](int l, int q) { return at_a(j, l, q); });
        v4d t = acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * i + o.kq[r], col = 16 * j + o.lr;
          if (row == col) acc[r] = acc[r] + (1.0 / dv[row] + J);
          t[r] = acc[r];
          if (row < N && col < N) {
            if (g.cov && (i != j || col <= row)) {
              g.cov[((long)b * N + row) * N + col] = acc[r];
              if (i != j) g.cov[((long)b * N + col) * N + row] = acc[r];
            }
            if (g.var && row == col) g.var[(long)b * N + row] = acc[r];
          } else {
            // bordered matrix for chol(cov): rows N (1^T) and N+1 (mu^T) in the padding, huge pivots there, identity elsewhere
            double v = row == col ? ((row == N || row == N + 1) ? 1e30 : 1.0) : 0.0;
            if (col < N && row == N) v = 1.0;
            if (col < N && row == N + 1) v = av[col];
            t[r] = v;
          }
        }
        st_tile(Gb + (size_t)lt(i, j) * TS, o, t);
      }
      if (i + 1 < NT) stage_row(i + 1, (i & 1) ? E0 : E1);
      GSYNC();
    }
    if (g.mu)
      for (int t = tid; t < N; t += NTH) g.mu[(long)b * N + t] = av[t];
    if (g.cov) {      // exact symmetry inside the diagonal tiles: upper element = lower element
      for (int e = tid; e < NT * 256; e += NTH) {
        const int i = e >> 8, rr = (e >> 4) & 15, cc = e & 15;
        const int row = 16 * i + rr, col = 16 * i + cc;
        if (cc > rr && row < N && col < N) g.cov[((long)b * N + row) * N + col] = Gb[lt(i, i) * TS + swz(cc, rr)];
      }
      GSYNC();      // (the factorisation below overwrites the diagonal tiles)
    }
    MSTAMP(11);
    if (g.chol || g.logdet || g.basis) {
      bad = potrf_tiles<false>(NT, Gb, DLi, (double*)nullptr, o, wave, lane, bad, NoExtra());   // Distribution.__post_init__ (data.py:38-39)
      GSYNC();
      if (g.chol) {
        for (int e = tid; e < NT * NT * 256; e += NTH) {
          const int tt = e >> 8, i = tt / NT, j = tt - i * NT, rr = (e >> 4) & 15, cc = e & 15;
          const int row = 16 * i + rr, col = 16 * j + cc;
          if (row < N && col < N) g.chol[((long)b * N + row) * N + col] = col <= row ? Gb[lt(i, j) * TS + swz(rr, cc)] : 0.0;
        }
      }
      if (g.basis) {
        for (int e = tid; e < 2 * N; e += NTH) {
          const int which = e / N, c = e - which * N, row = N + which;
          g.basis[((long)b * 2 + which) * N + c] = Gb[lt(row >> 4, c >> 4) * TS + swz(row & 15, c & 15)];
        }
      }
      if (g.logdet) {
        double s = 0.0;
        for (int t = tid; t < N; t += NTH) s += log(Gb[lt(t >> 4, t >> 4) * TS + swz(t & 15, t & 15)]);
        s = wave_sum(s);
        if (lane == 0) misc[16 + wave] = s;
        GSYNC();
        if (tid == 0) {
          double tot = 0.0;
          for (int w = 0; w < NW; ++w) tot += misc[16 + w];
          g.logdet[b] = tot;
        }
      }
    }
    if (tid == 0 && g.hyp) { g.hyp[(long)b * 2] = misc[8]; g.hyp[(long)b * 2 + 1] = misc[9]; }
    if (wave == 0 && lane == 0) g.info[b] = bad;
    MSTAMP(12);
    GSYNC();                                                                // LDS and the scratch are re-initialised for the next member
  }
}

int medium_tiles(int N) { return (N + 2 + 15) / 16; }

bool medium_fit_supported(int N, int R) { return N > smalln::SMALL_MAX_N && N <= MEDIUM_MAX_N && R >= 1; }

size_t medium_scratch_doubles(int N) {
  const int NT = medium_tiles(N), NTT = NT * (NT + 1) / 2;
  return ((size_t)6 * NTT + 2 * NT) * TS;
}

static size_t medium_lds_bytes(int N) {
  const int NT = medium_tiles(N), NP = 16 * NT;
  return ((size_t)NP + 64 + 9 * (size_t)NP + 2 * (size_t)NT * TS) * sizeof(double);
}

int medium_grid(int B, int N, int n_cus) {
  (void)N;
  // one workgroup per CU: the kernel needs ~250 registers per lane, i.e. two waves per SIMD = the eight waves of ONE workgroup
  return B < n_cus ? B : n_cus;
}

void medium_fit(hipStream_t stream, const MediumArgs& g_in, int n_cus) {
  if (!medium_fit_supported(g_in.N, g_in.R)) throw HipError{-1, "medium_fit: N out of range"};
  if (g_in.B <= 0) return;
  MediumArgs g = g_in;
  g.NT = medium_tiles(g.N);
  g.scratch_stride = (long)medium_scratch_doubles(g.N);
  const size_t lds = medium_lds_bytes(g.N);
  static bool configured[64] = {};
  int dev = 0;
  BAYENS_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !configured[dev]) {
    BAYENS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(medium_fit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   128 * 1024));
    if (dev >= 0 && dev < 64) configured[dev] = true;
  }
  hipLaunchKernelGGL(medium_fit_kernel, dim3(medium_grid(g.B, g.N, n_cus)), dim3(NTH), lds, stream, g);
  BAYENS_HIP(hipGetLastError());
}

}  // namespace mediumn
}  // namespace bayens
