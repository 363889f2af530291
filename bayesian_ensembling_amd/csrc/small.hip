// small.hip - the whole GP fit of a SMALL member (N <= 94 time points) in ONE kernel, one workgroup per member,
// every matrix resident in the CU's LDS / registers (BASELINE.json config 5: per-gridpoint fits, N = 86; the
// reference's annual series, experiments/full_experiment_script.py:57,70).
//
// Same arithmetic as the general path (driver.cpp; reference ensembles/models.py:185-220 +
// ensembles/data.py:38-39): T x [K -> L = chol(K + J I); natural-gradient step on (Lam, theta1); C = chol(Lam);
// hyper-gradient through Phi(L^T Lbar) = tril(u m^T + G S); Adam], prediction, chol(cov) + log-likelihood basis.
// What differs is where the data lives and how the triangular systems are solved:
//   * a member is 16x16 tiles (NT <= 6 per side, identity padding, two padded rows reserved for the bordered
//     Cholesky), lower triangles stored tile-packed: NT(NT+1)/2 tiles of 2 KB; THREE such buffers (Lb, Cb, Wb) plus
//     the inverted diagonal tiles and the vectors fill the 160 KB of LDS; Lam and G live in registers (accumulator
//     layout, tiles dealt over the 8 waves);
//   * triangular solves become products with EXPLICIT triangular inverses (the inverse rides along with the blocked
//     factorisation as a forward substitution, or is formed in place column by column) - every N^3 term is then a
//     tile product on the fp64 matrix cores with no serial dependency beyond the 16x16 diagonal factorisations;
//   * tiles are XOR-swizzled (column ^ 2*(row>>1)): A-operand reads, B-operand reads and accumulator stores of
//     v_mfma_f64_16x16x4_f64 are all bank-conflict free without padding.
// HBM traffic per fit is the inputs and the exported posterior only (6.9 KB in, 59.9 KB out with the
// full covariance, 2.8 KB with the reduced export).  tests/dev/small_proto.py is the NumPy model of these phases.
#include <cmath>
#include <cstdlib>

#include "hip_prims.h"
#include "small.h"

#ifndef BAYENS_F32
namespace bayens {
namespace smalln {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int TS = 256;            // doubles per tile
constexpr int NW = 8;              // waves per workgroup
constexpr int NTH = NW * 64;
constexpr double SQRT3 = 1.7320508075688772;
constexpr double SQRT5 = 2.23606797749979;

struct Lane {
  int rc[4];   // rc[q] = swizzled offset of element (lr, 4q + lk)
  int cr[4];   // cr[q] = swizzled offset of element (4q + lk, lr)   (= accumulator register q of this lane)
  int lr, lk;
};

__device__ __forceinline__ int swz(int r, int c) { return r * 16 + (c ^ ((r >> 1) << 1)); }
__device__ __forceinline__ int lt(int i, int j) { return ((i * (i + 1)) >> 1) + j; }     // j <= i
__device__ __forceinline__ void unpack_lower(int p, int& i, int& j) {
  i = 0;
  while (((i + 1) * (i + 2) >> 1) <= p) ++i;
  j = p - ((i * (i + 1)) >> 1);
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
#define SYNC() __syncthreads()

__device__ __forceinline__ v4d ld_tile(const double* t, const Lane& o) {
  v4d a;
#pragma unroll
  for (int r = 0; r < 4; ++r) a[r] = t[o.cr[r]];
  return a;
}
__device__ __forceinline__ void st_tile(double* t, const Lane& o, v4d a) {
#pragma unroll
  for (int r = 0; r < 4; ++r) t[o.cr[r]] = a[r];
}
// acc += sa * op(A) op(B), tiles in LDS; TA / TB: use the transposed tile
template <bool TA, bool TB>
__device__ __forceinline__ v4d mma(v4d acc, const double* A, const double* B, const Lane& o, double sa = 1.0) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double a = sa * A[TA ? o.cr[q] : o.rc[q]];
    const double b = B[TB ? o.rc[q] : o.cr[q]];
    acc = MFMA(a, b, acc);
  }
  return acc;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double softplus_d(double u) { return u > 30.0 ? u : log1p(exp(u)); }

__device__ __forceinline__ void kern_eval(int kernel, double var, double ls, double r2, double& K, double& dK) {
  if (kernel == 2) {
    const double r2c = r2 > 0.0 ? r2 : 0.0;
    K = var * exp(-0.5 * r2c);
    dK = K * r2c / ls;
    return;
  }
  const bool clamp = r2 <= 1e-36;
  const double r = sqrt(r2 > 1e-36 ? r2 : 1e-36);
  if (kernel == 0) {
    const double e = exp(-SQRT3 * r);
    K = var * (1.0 + SQRT3 * r) * e;
    dK = 3.0 * var * r * r * e / ls;
  } else {
    const double e = exp(-SQRT5 * r);
    K = var * (1.0 + SQRT5 * r + (5.0 / 3.0) * r * r) * e;
    dK = (5.0 / 3.0) * var * r * r * (1.0 + SQRT5 * r) * e / ls;
  }
  if (clamp) dK = 0.0;
}

// ---- 16x16 diagonal tile: Cholesky + inverse of the factor in registers (one wave; the tile is replicated in the
// four 16-lane rows, lane l owns row l & 15; pivots and columns travel by DPP row broadcast; 1/L_ii from a Newton-
// refined v_rsq_f64).  T: the tile (lower part read; factor written back with zeros above the diagonal);
// Dk: inverse of the factor (zeros above the diagonal).  Same routine as leaf_factor's P1 in kernels.hip.
template <int K>
__device__ __forceinline__ double bcast16_k(double v) {
  return __builtin_amdgcn_mov_dpp(v, 0x150 + K, 0xf, 0xf, false);
}
__device__ __forceinline__ double readlane_d(double v, int k) {
  switch (k) {
    case 0: return bcast16_k<0>(v);   case 1: return bcast16_k<1>(v);   case 2: return bcast16_k<2>(v);
    case 3: return bcast16_k<3>(v);   case 4: return bcast16_k<4>(v);   case 5: return bcast16_k<5>(v);
    case 6: return bcast16_k<6>(v);   case 7: return bcast16_k<7>(v);   case 8: return bcast16_k<8>(v);
    case 9: return bcast16_k<9>(v);   case 10: return bcast16_k<10>(v); case 11: return bcast16_k<11>(v);
    case 12: return bcast16_k<12>(v); case 13: return bcast16_k<13>(v); case 14: return bcast16_k<14>(v);
    default: return bcast16_k<15>(v);
  }
}
__device__ __forceinline__ double rsqrt_refined(double d) {
  double r = __builtin_amdgcn_rsq(d);
  r = r * (1.5 - 0.5 * d * r * r);
  const double e = fma(-d * r, r, 1.0);
  return fma(0.5 * r, e, r);
}
__device__ __forceinline__ void diag16(double* T, double* Dk, int lane, int pivot_base, int& bad) {
  double row[16], rinv[16], m[16];
  const int li = lane & 15;
#pragma unroll
  for (int j = 0; j < 16; ++j) row[j] = T[swz(li, j)];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double d = readlane_d(row[j], j);
    const bool ok = d > 0.0;
    bad = (!ok && bad == 0) ? pivot_base + j + 1 : bad;
    d = ok ? d : 1.0;
    const double r = rsqrt_refined(d);
    rinv[j] = r;
    row[j] = li == j ? d * r : (li > j ? row[j] * r : 0.0);
    const double lij = row[j];
#pragma unroll
    for (int k = j + 1; k < 16; ++k) {
      const double lkj = readlane_d(row[j], k);
      row[k] = fma(-lij, lkj, row[k]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(row[j]));
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    double sacc = li == i ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < i; ++k) sacc = fma(-readlane_d(row[k], i), m[k], sacc);
    m[i] = sacc * rinv[i];
    asm volatile("" : "+v"(m[i]));
    __builtin_amdgcn_sched_barrier(0);
  }
  if (lane < 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      T[swz(li, j)] = row[j];
      Dk[swz(j, li)] = m[j];      // m[j] = inverse[j][li]
    }
  }
}

// ---- blocked Cholesky of the lower-packed buffer A in place; D[k] = inverse of diagonal tile k.  WITH_INV: the
// inverse of the factor is formed alongside in Xb by block forward substitution (X_ji accumulates -sum_k L_jk X_ki and
// is finished by inv_jj when row j is reached): no extra serial chain, the products ride in the same three phases.
template <int NT, bool WITH_INV>
__device__ __forceinline__ void potrf_lds(double* A, double* D, double* Xb, const Lane& o, int wave, int lane, int& bad) {
  for (int k = 0; k < NT; ++k) {
    if (wave == 0) diag16(A + lt(k, k) * TS, D + k * TS, lane, 16 * k, bad);
    SYNC();
    const int below = NT - 1 - k;
    for (int q = wave; q < below + (WITH_INV ? k + 1 : 0); q += NW) {
      if (q < below) {                                    // panel: L_ik = A_ik inv_kk^T
        double* t = A + lt(k + 1 + q, k) * TS;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        acc = mma<false, true>(acc, t, D + k * TS, o);
        st_tile(t, o, acc);
      } else {
        const int i = q - below;
        double* t = Xb + lt(k, i) * TS;
        if (i < k) {                                      // X_ki = inv_kk T_ki
          v4d acc = {0.0, 0.0, 0.0, 0.0};
          acc = mma<false, false>(acc, D + k * TS, t, o);
          st_tile(t, o, acc);
        } else {
          st_tile(t, o, ld_tile(D + k * TS, o));          // X_kk = inv_kk
        }
      }
    }
    SYNC();
    int p = 0;
    for (int i = k + 1; i < NT; ++i)
      for (int j = k + 1; j <= i; ++j, ++p)
        if (p % NW == wave) {                             // A_ij -= L_ik L_jk^T
          double* t = A + lt(i, j) * TS;
          v4d acc = ld_tile(t, o);
          acc = mma<false, true>(acc, A + lt(i, k) * TS, A + lt(j, k) * TS, o, -1.0);
          st_tile(t, o, acc);
        }
    if (WITH_INV) {
      for (int j = k + 1; j < NT; ++j)
        for (int i = 0; i <= k; ++i, ++p)
          if (p % NW == wave) {                           // T_ji -= L_jk X_ki   (first touched at k == i)
            double* t = Xb + lt(j, i) * TS;
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            if (k > i) acc = ld_tile(t, o);
            acc = mma<false, false>(acc, A + lt(j, k) * TS, Xb + lt(k, i) * TS, o, -1.0);
            st_tile(t, o, acc);
          }
    }
    SYNC();
  }
}

// ---- Lb holds a lower factor L with the inverses D of its diagonal tiles; on return it holds L^-1.  Columns from the
// last to the first: X_ij = -(sum_{j<k<=i} X_ik L_kj) inv_jj reads column j of L and finished tiles only.  The sum is
// accumulated TRANSPOSED (L_kj^T X_ik^T), so that the accumulator registers are the A operand of the product with inv_jj.
template <int NT>
__device__ __forceinline__ void invert_in_place(double* Lb, const double* D, const Lane& o, int wave) {
  for (int j = NT - 1; j >= 0; --j) {
    const int i = j + 1 + wave;          // NT - 1 - j <= 5 tiles below the diagonal: one per wave
    v4d x = {0.0, 0.0, 0.0, 0.0};
    const bool has = i < NT;
    if (has) {
      v4d tt = {0.0, 0.0, 0.0, 0.0};
      for (int k = j + 1; k <= i; ++k) tt = mma<true, true>(tt, Lb + lt(k, j) * TS, Lb + lt(i, k) * TS, o);
      const double* dj = D + j * TS;
#pragma unroll
      for (int q = 0; q < 4; ++q) x = MFMA(-tt[q], dj[o.cr[q]], x);
    }
    if (wave == NW - 1) st_tile(Lb + lt(j, j) * TS, o, ld_tile(D + j * TS, o));   // (the diagonal tile is not read above)
    SYNC();
    if (has) st_tile(Lb + lt(i, j) * TS, o, x);
    SYNC();
  }
}

// ---- matrix-vector products on tile-packed buffers: wave w < NT owns the 16 outputs of tile row / column w, lane
// (lr, lk) walks the columns == lk mod 4, the four partial sums meet through two shuffles.  Valid in lanes lk == 0.
template <int NT>
__device__ __forceinline__ double mv_lower(const double* Lb, const double* x, int i, const Lane& o) {     // (L x)[16 i + lr]
  double s = 0.0;
  for (int k = 0; k <= i; ++k) {
    const double* t = Lb + lt(i, k) * TS;
#pragma unroll
    for (int q = 0; q < 4; ++q) s = fma(t[o.rc[q]], x[16 * k + 4 * q + o.lk], s);
  }
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}
template <int NT>
__device__ __forceinline__ double mv_lower_t(const double* Lb, const double* x, int j, const Lane& o) {   // (L^T x)[16 j + lr]
  double s = 0.0;
  for (int k = j; k < NT; ++k) {
    const double* t = Lb + lt(k, j) * TS;
#pragma unroll
    for (int q = 0; q < 4; ++q) s = fma(t[o.cr[q]], x[16 * k + 4 * q + o.lk], s);
  }
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}
template <int NT>
__device__ __forceinline__ double mv_sym(const double* Sb, const double* x, int i, const Lane& o) {       // (S x)[16 i + lr]
  double s = 0.0;
  for (int k = 0; k < NT; ++k) {
    const double* t = Sb + (k <= i ? lt(i, k) : lt(k, i)) * TS;
#pragma unroll
    for (int q = 0; q < 4; ++q) s = fma(t[k <= i ? o.rc[q] : o.cr[q]], x[16 * k + 4 * q + o.lk], s);
  }
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}

// squared distances of tile (i, j): X~ X~^T on the matrix cores (operands straight from the member's X, scaled on the
// fly), r2 = (-2 dot + |x_i|^2) + |x_j|^2 as GPflow / the oracle form it
__device__ __forceinline__ v4d r2_tile(const double* X, int N, int R, double ls, const double* ssq, int i, int j, const Lane& o) {
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  const int ra = 16 * i + o.lr, rb = 16 * j + o.lr;
  for (int k0 = 0; k0 < R; k0 += 4) {
    const int c = k0 + o.lk;
    const double a = (ra < N && c < R) ? X[(long)ra * R + c] / ls : 0.0;
    const double b = (rb < N && c < R) ? X[(long)rb * R + c] / ls : 0.0;
    acc = MFMA(a, b, acc);
  }
  const double sj = ssq[16 * j + o.lr];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = (-2.0 * acc[r] + ssq[16 * i + o.lk + 4 * r]) + sj;
  return acc;
}

template <int NT>
struct Layout {
  static constexpr int NTT = NT * (NT + 1) / 2;
  static constexpr int NP = 16 * NT;
  static constexpr int TPW = (NTT + NW - 1) / NW;       // register tiles per wave
  static constexpr int LB = 0, CB = NTT * TS, WB = 2 * NTT * TS, DL = 3 * NTT * TS, DC = DL + NT * TS;
  static constexpr int VEC = DC + NT * TS;               // vectors of NP doubles
  static constexpr int NVEC = 10;
  static constexpr int MISC = VEC + NVEC * NP;
  static constexpr int TOTAL = MISC + 64;                // doubles
};

template <int NT>
__global__ __launch_bounds__(NTH) void small_fit_kernel(SmallArgs g) {
  typedef Layout<NT> Y;
  constexpr int NTT = Y::NTT, NP = Y::NP, TPW = Y::TPW;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* Lb = lds + Y::LB;
  double* Cb = lds + Y::CB;
  double* Wb = lds + Y::WB;
  double* DLi = lds + Y::DL;
  double* DCi = lds + Y::DC;            // inverted diagonal tiles of C; in the prediction: staging of one tile row of A^T S
  double* yv = lds + Y::VEC;            // y (0 in the padding)
  double* dv = yv + NP;                 // 1 / nu (1 in the padding)
  double* rs = dv + NP;                 // 1 / sqrt(nu)
  double* yd = rs + NP;                 // y / nu
  double* th = yd + NP;                 // theta1
  double* mv = th + NP;                 // m
  double* av = mv + NP;                 // a = L m - y (later: L m of the prediction)
  double* wv = av + NP;                 // a / nu
  double* uv = wv + NP;                 // u = L^T D a (later: Linv^T m)
  double* ssq = uv + NP;                // |x_i / ls|^2
  double* misc = lds + Y::MISC;         // [0..7] optimiser state, [8] variance, [9] lengthscale, [16..31] reduction scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Lane o;
  o.lr = lane & 15; o.lk = lane >> 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) { o.rc[q] = swz(o.lr, 4 * q + o.lk); o.cr[q] = swz(4 * q + o.lk, o.lr); }
  const int N = g.N, R = g.R;
  const double J = g.jitter;

  for (int b = blockIdx.x; b < g.B; b += gridDim.x) {
    const double* X = g.X + (long)b * N * R;
    int bad = 0;
    // ---- member set-up ----
    for (int t = tid; t < NP; t += NTH) {
      const double nu = t < N ? g.nu[(long)b * N + t] : 1.0;
      const double y = t < N ? g.y[(long)b * N + t] : 0.0;
      yv[t] = y; dv[t] = 1.0 / nu; rs[t] = 1.0 / sqrt(nu); yd[t] = y / nu;
      th[t] = 0.0; mv[t] = 0.0;
    }
    if (tid < 8) misc[tid] = tid < 2 ? 0.5413248546129181 /* softplus^-1(1) = log(e - 1) */ : 0.0;
    v4d lam[TPW], gt[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      const int p = wave + s * NW;
      int i = 0, j = 0;
      unpack_lower(p < NTT ? p : 0, i, j);
#pragma unroll
      for (int r = 0; r < 4; ++r) lam[s][r] = (i == j && o.lk + 4 * r == o.lr) ? 1.0 : 0.0;
      gt[s] = v4d{0.0, 0.0, 0.0, 0.0};
      if (p < NTT) st_tile(Cb + p * TS, o, lam[s]);        // S = I until the first natural-gradient step (n_iters = 0)
    }
    SYNC();

    // hyper-parameters of the coming kernel matrix -> misc[8], misc[9]; |x~|^2
    auto refresh_hypers = [&]() {
      if (tid == 0) { misc[8] = softplus_d(misc[0]); misc[9] = softplus_d(misc[1]); }
      SYNC();
      const double ls = misc[9];
      for (int t = tid; t < NP; t += NTH) {
        double s = 0.0;
        if (t < N)
          for (int d = 0; d < R; ++d) { const double v = X[(long)t * R + d] / ls; s += v * v; }
        ssq[t] = s;
      }
      SYNC();
    };
    // Lb <- tril(K0) + J I (identity in the padding)
    auto kmat = [&]() {
      const double var = misc[8], ls = misc[9];
      for (int p = wave; p < NTT; p += NW) {
        int i, j;
        unpack_lower(p, i, j);
        v4d r2 = r2_tile(X, N, R, ls, ssq, i, j, o);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * i + o.lk + 4 * r, col = 16 * j + o.lr;
          double K, dK;
          kern_eval(g.kernel, var, ls, r2[r], K, dK);
          if (row >= N || col >= N) K = row == col ? 1.0 : 0.0;
          else K = col <= row ? K + (row == col ? J : 0.0) : 0.0;
          r2[r] = K;
        }
        st_tile(Lb + p * TS, o, r2);
      }
      SYNC();
    };

    for (int it = 0; it < g.n_iters; ++it) {
      refresh_hypers();
      kmat();                                                              // K0 + J I at the current hypers  (models.py:209)
      potrf_lds<NT, false>(Lb, DLi, nullptr, o, wave, lane, bad);          // L
      // ---- natural-gradient step: G = L^T D L, Lam <- (1-g) Lam + g (I + G), theta1 <- (1-g) theta1 + g L^T (y / nu)
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int p = wave + s * NW;
        if (p < NTT) {
          int i, j;
          unpack_lower(p, i, j);
          v4d acc = {0.0, 0.0, 0.0, 0.0};
          for (int k = i; k < NT; ++k) {
            const double* ta = Lb + lt(k, i) * TS;
            const double* tb = Lb + lt(k, j) * TS;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const double sc = rs[16 * k + 4 * q + o.lk];
              acc = MFMA(ta[o.cr[q]] * sc, tb[o.cr[q]] * sc, acc);
            }
          }
          gt[s] = acc;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            lam[s][r] = (1.0 - g.gamma) * lam[s][r] + g.gamma * (acc[r] + ((i == j && o.lk + 4 * r == o.lr) ? 1.0 : 0.0));
          st_tile(Cb + p * TS, o, lam[s]);
        }
      }
      if (wave < NT) {
        const double s = mv_lower_t<NT>(Lb, yd, wave, o);
        if (o.lk == 0) th[16 * wave + o.lr] = (1.0 - g.gamma) * th[16 * wave + o.lr] + g.gamma * s;
      }
      SYNC();
      potrf_lds<NT, true>(Cb, DCi, Wb, o, wave, lane, bad);                // C = chol(Lam) in Cb, C^-1 in Wb
      // ---- S = C^-T C^-1 (registers), then over C in Cb
      v4d st[TPW];
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int p = wave + s * NW;
        st[s] = v4d{0.0, 0.0, 0.0, 0.0};
        if (p < NTT) {
          int i, j;
          unpack_lower(p, i, j);
          for (int k = i; k < NT; ++k) st[s] = mma<true, false>(st[s], Wb + lt(k, i) * TS, Wb + lt(k, j) * TS, o);
        }
      }
      SYNC();
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int p = wave + s * NW;
        if (p < NTT) { st_tile(Cb + p * TS, o, st[s]); st_tile(Wb + p * TS, o, gt[s]); }     // S -> Cb, G -> Wb
      }
      SYNC();
      // ---- m = S theta1 ; a = L m - y ; u = L^T D a
      if (wave < NT) {
        const double s = mv_sym<NT>(Cb, th, wave, o);
        if (o.lk == 0) mv[16 * wave + o.lr] = s;
      }
      SYNC();
      if (wave < NT) {
        const double s = mv_lower<NT>(Lb, mv, wave, o);
        if (o.lk == 0) {
          const int t = 16 * wave + o.lr;
          const double a = s - yv[t];
          av[t] = a; wv[t] = a * dv[t];
        }
      }
      SYNC();
      if (wave < NT) {
        const double s = mv_lower_t<NT>(Lb, wv, wave, o);
        if (o.lk == 0) uv[16 * wave + o.lr] = s;
      }
      SYNC();
      // ---- Phi(L^T Lbar) = tril(u m^T + G S), half the diagonal   (G in Wb, S in Cb, both symmetric from lower tiles)
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int p = wave + s * NW;
        if (p < NTT) {
          int i, j;
          unpack_lower(p, i, j);
          v4d acc = {0.0, 0.0, 0.0, 0.0};
          for (int k = 0; k < NT; ++k) {
            const double* ta = Wb + (k <= i ? lt(i, k) : lt(k, i)) * TS;
            const double* tb = Cb + (k >= j ? lt(k, j) : lt(j, k)) * TS;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = MFMA(ta[k <= i ? o.rc[q] : o.cr[q]], tb[k >= j ? o.cr[q] : o.rc[q]], acc);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rl = o.lk + 4 * r;
            double v = fma(uv[16 * i + rl], mv[16 * j + o.lr], acc[r]);
            if (i == j) v = rl < o.lr ? 0.0 : (rl == o.lr ? 0.5 * v : v);
            acc[r] = v;
          }
          st[s] = acc;
        }
      }
      SYNC();
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int p = wave + s * NW;
        if (p < NTT) st_tile(Wb + p * TS, o, st[s]);                        // Phi -> Wb
      }
      invert_in_place<NT>(Lb, DLi, o, wave);                               // Lb = L^-1   (its first barrier publishes Phi)
      // ---- W = Phi L^-1 (lower) -> Wb
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int p = wave + s * NW;
        st[s] = v4d{0.0, 0.0, 0.0, 0.0};
        if (p < NTT) {
          int i, j;
          unpack_lower(p, i, j);
          for (int k = j; k <= i; ++k) st[s] = mma<false, false>(st[s], Wb + lt(i, k) * TS, Lb + lt(k, j) * TS, o);
        }
      }
      SYNC();
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int p = wave + s * NW;
        if (p < NTT) st_tile(Wb + p * TS, o, st[s]);
      }
      SYNC();
      // ---- Kbar = L^-T W ; g = <Kbar, dK/dtheta> over the real part ; Adam   (models.py:210)
      {
        const double var = misc[8], ls = misc[9];
        double gv = 0.0, gl = 0.0;
        for (int q = wave; q < NT * NT; q += NW) {
          const int i = q / NT, j = q - i * NT;
          v4d acc = {0.0, 0.0, 0.0, 0.0};
          for (int k = (i > j ? i : j); k < NT; ++k) acc = mma<true, false>(acc, Lb + lt(k, i) * TS, Wb + lt(k, j) * TS, o);
          const v4d r2 = r2_tile(X, N, R, ls, ssq, i, j, o);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * i + o.lk + 4 * r, col = 16 * j + o.lr;
            double K, dK;
            kern_eval(g.kernel, var, ls, r2[r], K, dK);
            if (row < N && col < N) { gv = fma(acc[r], K, gv); gl = fma(acc[r], dK, gl); }
          }
        }
        gv = wave_sum(gv);
        gl = wave_sum(gl);
        if (lane == 0) { misc[16 + wave] = gv; misc[24 + wave] = gl; }
        SYNC();
        if (tid == 0) {
          double sv = 0.0, sl = 0.0;
          for (int w = 0; w < NW; ++w) { sv += misc[16 + w]; sl += misc[24 + w]; }
          double* u = misc;
          double gg[2];
          gg[0] = (sv / var) * (1.0 - exp(-var));     // softplus'(u) = 1 - exp(-softplus(u))
          gg[1] = sl * (1.0 - exp(-ls));
          const double t = u[6] + 1.0;
          u[6] = t;
          const double lr_t = g.lr * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t));
          for (int q = 0; q < 2; ++q) {               // TF-2.8 Keras Adam (models.py:192)
            u[2 + q] = 0.9 * u[2 + q] + 0.1 * gg[q];
            u[4 + q] = 0.999 * u[4 + q] + 0.001 * gg[q] * gg[q];
            u[q] -= lr_t * u[2 + q] / (sqrt(u[4 + q]) + 1e-7);
          }
        }
        SYNC();
      }
    }

    // ---- prediction at the final hypers with the (m, S) of the last step    (models.py:217-220)
    refresh_hypers();
    kmat();
    potrf_lds<NT, false>(Lb, DLi, nullptr, o, wave, lane, bad);
    for (int p = wave; p < NTT; p += NW) st_tile(Wb + p * TS, o, ld_tile(Lb + p * TS, o));
    SYNC();
    invert_in_place<NT>(Wb, DLi, o, wave);                                  // Wb = L^-1, Lb = L, Cb = S
    // mu = A^T m with A^T = K0 L^-T = L - J L^-T
    if (wave < NT) {
      const double s1 = mv_lower<NT>(Lb, mv, wave, o);
      const double s2 = mv_lower_t<NT>(Wb, mv, wave, o);
      if (o.lk == 0) av[16 * wave + o.lr] = s1 - J * s2;                    // av = mu (padded part 0)
    }
    SYNC();
    if (g.mu)
      for (int t = tid; t < N; t += NTH) g.mu[(long)b * N + t] = av[t];
    // cov = diag(nu + J) - J^2 L^-T L^-1 + (A^T S) A, one tile row at a time: E_i* = A^T_i* S staged in LDS (DCi)
    v4d covt[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      if (wave < NT) {
        const int l = wave;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < NT; ++k) {
          const double* tb = Cb + (k >= l ? lt(k, l) : lt(l, k)) * TS;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            double a;
            if (k < i) a = Lb[lt(i, k) * TS + o.rc[q]];
            else if (k > i) a = -J * Wb[lt(k, i) * TS + o.cr[q]];
            else a = Lb[lt(i, i) * TS + o.rc[q]] - J * Wb[lt(i, i) * TS + o.cr[q]];
            acc = MFMA(a, tb[k >= l ? o.cr[q] : o.rc[q]], acc);
          }
        }
        st_tile(DCi + l * TS, o, acc);
      }
      SYNC();
      covt[i] = v4d{0.0, 0.0, 0.0, 0.0};
      if (wave <= i) {
        const int j = wave;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        for (int k = i; k < NT; ++k) acc = mma<true, false>(acc, Wb + lt(k, i) * TS, Wb + lt(k, j) * TS, o);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] *= -J * J;
        for (int l = 0; l < NT; ++l) {
          const double* ta = DCi + l * TS;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            double bb;      // B[kq][col] = A^T(j, l)[col][kq]
            if (l < j) bb = Lb[lt(j, l) * TS + o.rc[q]];
            else if (l > j) bb = -J * Wb[lt(l, j) * TS + o.cr[q]];
            else bb = Lb[lt(j, j) * TS + o.rc[q]] - J * Wb[lt(j, j) * TS + o.cr[q]];
            acc = MFMA(ta[o.rc[q]], bb, acc);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * i + o.lk + 4 * r, col = 16 * j + o.lr;
          if (row == col) acc[r] += 1.0 / dv[row] + J;
          if (row < N && col < N) {
            // (a diagonal tile is a full 16x16 accumulation: only its lower half is exported here, the upper half is
            //  mirrored from it below, so that cov is exactly symmetric like the general path's extraction)
            if (g.cov && (i != j || col <= row)) {
              g.cov[((long)b * N + row) * N + col] = acc[r];
              if (i != j) g.cov[((long)b * N + col) * N + row] = acc[r];
            }
            if (g.var && row == col) g.var[(long)b * N + row] = acc[r];
          }
        }
        covt[i] = acc;
      }
      SYNC();
    }
    // bordered matrix for chol(cov): rows N (1^T) and N+1 (mu^T) in the padding, huge pivots there, identity elsewhere
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      if (wave <= i) {
        const int j = wave;
        v4d t = covt[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * i + o.lk + 4 * r, col = 16 * j + o.lr;
          if (row >= N || col >= N) {
            double v = row == col ? ((row == N || row == N + 1) ? 1e30 : 1.0) : 0.0;
            if (col < N && row == N) v = 1.0;
            if (col < N && row == N + 1) v = av[col];
            t[r] = v;
          }
        }
        st_tile(Lb + lt(i, j) * TS, o, t);
      }
    }
    SYNC();
    if (g.cov) {      // exact symmetry inside the diagonal tiles: upper element = lower element
      for (int e = tid; e < NT * 256; e += NTH) {
        const int i = e >> 8, rr = (e >> 4) & 15, cc = e & 15;
        const int row = 16 * i + rr, col = 16 * i + cc;
        if (cc > rr && row < N && col < N) g.cov[((long)b * N + row) * N + col] = Lb[lt(i, i) * TS + swz(cc, rr)];
      }
    }
    if (g.chol || g.logdet || g.basis) {
      potrf_lds<NT, false>(Lb, DLi, nullptr, o, wave, lane, bad);           // Distribution.__post_init__ (data.py:38-39)
      if (g.chol) {
        for (int e = tid; e < NT * NT * 256; e += NTH) {
          const int tt = e >> 8, i = tt / NT, j = tt - i * NT, rr = (e >> 4) & 15, cc = e & 15;
          const int row = 16 * i + rr, col = 16 * j + cc;
          if (row < N && col < N) g.chol[((long)b * N + row) * N + col] = col <= row ? Lb[lt(i, j) * TS + swz(rr, cc)] : 0.0;
        }
      }
      if (g.basis) {
        for (int e = tid; e < 2 * N; e += NTH) {
          const int which = e / N, c = e - which * N, row = N + which;
          g.basis[((long)b * 2 + which) * N + c] = Lb[lt(row >> 4, c >> 4) * TS + swz(row & 15, c & 15)];
        }
      }
      if (g.logdet) {
        double s = 0.0;
        for (int t = tid; t < N; t += NTH) s += log(Lb[lt(t >> 4, t >> 4) * TS + swz(t & 15, t & 15)]);
        s = wave_sum(s);
        if (lane == 0) misc[16 + wave] = s;
        SYNC();
        if (tid == 0) {
          double tot = 0.0;
          for (int w = 0; w < NW; ++w) tot += misc[16 + w];
          g.logdet[b] = tot;
        }
      }
    }
    if (tid == 0) {
      if (g.hyp) { g.hyp[(long)b * 2] = misc[8]; g.hyp[(long)b * 2 + 1] = misc[9]; }
    }
    if (wave == 0 && lane == 0) g.info[b] = bad;
    SYNC();                                                                 // LDS is re-initialised for the next member
  }
}

template <int NT>
static void launch(hipStream_t stream, const SmallArgs& g, int n_cus) {
  const size_t lds = (size_t)Layout<NT>::TOTAL * sizeof(double);
  static bool configured = false;
  if (!configured) {
    BAYENS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(small_fit_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    configured = true;
  }
  const int per_cu = (int)(163840 / lds) > 0 ? (int)(163840 / lds) : 1;
  int grid = n_cus * per_cu;
  if (grid > g.B) grid = g.B;
  hipLaunchKernelGGL((small_fit_kernel<NT>), dim3(grid), dim3(NTH), lds, stream, g);
  BAYENS_HIP(hipGetLastError());
}

bool small_fit_supported(int N, int R) { return N >= 1 && N <= SMALL_MAX_N && R >= 1; }

void small_fit(hipStream_t stream, const SmallArgs& g, int n_cus) {
  if (!small_fit_supported(g.N, g.R)) throw HipError{-1, "small_fit: N out of range"};
  if (g.B <= 0) return;
  if (g.N + 2 <= 32) launch<2>(stream, g, n_cus);
  else if (g.N + 2 <= 64) launch<4>(stream, g, n_cus);
  else launch<6>(stream, g, n_cus);
}

}  // namespace smalln
}  // namespace bayens
#endif
