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
//   * triangular solves become products with EXPLICIT triangular inverses: L^-1 and C^-1 ride along with their blocked
//     factorisations as block forward substitutions whose products run in the shadow of the serial diagonal tiles, as
//     do G = L^T D L and S = C^-T C^-1 (accumulated block row by block row).  Since u = L^T D (L m - y) = G m - L^T (y/nu),
//     L is dead once theta1 is updated and C^-1 takes its buffer.  Every N^3 term is a tile product on the fp64 matrix
//     cores with no serial dependency beyond the 16x16 diagonal factorisations;
//   * tiles are XOR-swizzled (column ^ 2*(row>>1)): A-operand reads, B-operand reads and accumulator stores of
//     v_mfma_f64_16x16x4_f64 are all bank-conflict free without padding.
// HBM traffic per fit is the inputs and the exported posterior only (6.9 KB in, 59.9 KB out with the
// full covariance, 2.8 KB with the reduced export).  tests/dev/small_proto.py is the NumPy model of these phases.
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "hip_prims.h"
#include "small.h"

// global loads / stores of the fused kernels: plain here; the bounds-checked build of the mid-size kernel (medium.hip,
// BAYENS_MEDIUM_CHECK) routes them through a region check
#ifndef GLD
#define GLD(lv) (lv)
#define GST(lv, v) ((lv) = (v))
#endif

// Compiled twice, like gemm.hip / kernels.hip: `real` = double (namespace bayens, the reference precision) and, with
// -DBAYENS_F32, `real` = float (namespace bayens_f32: fp32 tiles and v_mfma_f32_16x16x4_f32 products; the 16x16 diagonal
// factorisations, the squared distances of the kernel matrix, the vectors and every reduction stay fp64).
namespace BAYENS_NS {
namespace smalln {

typedef double v4d __attribute__((ext_vector_type(4)));   // fp64 accumulator tile (diagonal factorisations, distances)
typedef real v4r __attribute__((ext_vector_type(4)));     // accumulator tile in the fit's precision

constexpr int TS = 256;            // elements per tile
// 8 waves, one workgroup per CU.  (Measured alternatives: 4 waves per workgroup leave 7 register tiles per matrix and
// wave - fp64: 6.5e5 instead of 8.7e5 fits/s at 512 registers; with the 256 registers that two such workgroups per CU
// would be held to - which the 80 KB fp32 footprint would allow - the kernel spills ~1900 registers and crawls.)
constexpr int NW = 8;
typedef double vecr;               // element type of the LDS vectors (theta1, m, u, ...): fp64 in both precisions
constexpr int NTH = NW * 64;
constexpr double SQRT3 = 1.7320508075688772;
constexpr double SQRT5 = 2.23606797749979;

__device__ __forceinline__ double* lds_base() {
  extern __shared__ __attribute__((aligned(16))) double lds_block[];
  return lds_block;
}
__device__ __forceinline__ real* tiles_base() { return reinterpret_cast<real*>(lds_base()); }

// The 16x16x4 MFMA of either precision takes lane (lr, lk)'s A element [lr][k] and B element [k][lr] with k = lk in
// each of the four steps of a 16-deep product; its accumulator register r holds row lk + 4r (fp64) or 4lk + r (fp32) of
// column lr.  Step q of a tile product contracts over the index set kq[q] = {4q + lk} (fp64) / {4lk + q} (fp32) - the
// order of a sum is free - so that in BOTH precisions accumulator register q is element (kq[q], lr): an accumulator
// tile is a B operand as it stands, and one table of offsets serves operand reads and accumulator stores.
struct Lane {
  int rc[4];   // rc[q] = swizzled offset of element (lr, kq(q))
  int cr[4];   // cr[q] = swizzled offset of element (kq(q), lr)   (= accumulator register q of this lane)
  int lane;
  // lr, lk and kq(q) are two or three integer instructions away from the lane id and are NOT kept in registers (round 4): as
  // struct members they were what the register allocator spilled to scratch - and reloaded from memory, a ~500-cycle round
  // trip, in front of every tile epilogue that needed a row / column index.
  // (Rounds 3-4 kept a second, stored-member form of this struct for medium.hip / potrf256.hip because the mid-size kernel
  //  "faulted with the computed form".  The cause was neither form: DESIGN section 5c, tools/check_exec_prologue.py.)
  __device__ __forceinline__ int lr() const { return lane & 15; }
  __device__ __forceinline__ int lk() const { return lane >> 4; }
  __device__ __forceinline__ int kq(int q) const;
};

#ifdef BAYENS_F32
#define MFMAR(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
__device__ __forceinline__ int kq_of(int lk, int q) { return 4 * lk + q; }
// 4-byte elements, 32 banks: column ^ p(row >> 1), p over {0..3, 8..11} (never differs by 4 alone: the two k-groups of a
// half-wave stay on distinct banks in the row-indexed reads; column-indexed reads are 2-way conflicted)
__device__ __forceinline__ int swz(int r, int c) { const int h = r >> 1; return r * 16 + (c ^ ((h & 3) | ((h >> 2) << 3))); }
// row of X that lane lr supplies as A row lr of the fp64 distance product, so that its result register r is row kq[r]
__device__ __forceinline__ int dist_row(int lr) { return 4 * (lr & 3) + (lr >> 2); }
#else
#define MFMAR(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
__device__ __forceinline__ int kq_of(int lk, int q) { return 4 * q + lk; }
__device__ __forceinline__ int swz(int r, int c) { return r * 16 + (c ^ ((r >> 1) << 1)); }
__device__ __forceinline__ int dist_row(int lr) { return lr; }
#endif
__device__ __forceinline__ int Lane::kq(int q) const { return kq_of(lane >> 4, q); }
__device__ __forceinline__ void lane_init(Lane& o, int lane) {
  o.lane = lane;
#pragma unroll
  for (int q = 0; q < 4; ++q) { o.rc[q] = swz(o.lr(), o.kq(q)); o.cr[q] = swz(o.kq(q), o.lr()); }
}
__device__ __forceinline__ int lt(int i, int j) { return ((i * (i + 1)) >> 1) + j; }     // j <= i
__device__ __forceinline__ void unpack_lower(int p, int& i, int& j) {
  i = 0;
  while (((i + 1) * (i + 2) >> 1) <= p) ++i;
  j = p - ((i * (i + 1)) >> 1);
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
// Workgroup barrier for LDS traffic only: every hand-off between waves goes through LDS, global memory holds read-only
// inputs and write-only outputs, so the barrier must not wait for outstanding global stores (__syncthreads() does).
#define SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// diagnostic: wave 0 / lane 0 of workgroup 0 stamps the phase boundaries of its member (g.stamps null in production)
#define STAMP(k) do { if (g.stamps && blockIdx.x == 0 && tid == 0) g.stamps[k] = __builtin_readcyclecounter(); } while (0)
// ... and every wave of workgroup 0 its own (g.stamps[32 + 16 wave + k]; bayens_small_wave_stamps)
#define WSTAMP(k) do { if (g.stamps && blockIdx.x == 0 && lane == 0) g.stamps[32 + 16 * wave + (k)] = __builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ v4r ld_tile(const real* t, const Lane& o) {
  v4r a;
#pragma unroll
  for (int r = 0; r < 4; ++r) a[r] = t[o.cr[r]];
  return a;
}
__device__ __forceinline__ void st_tile(real* t, const Lane& o, v4r a) {
#pragma unroll
  for (int r = 0; r < 4; ++r) t[o.cr[r]] = a[r];
}
// acc += sa * op(A) op(B), tiles in LDS; TA / TB: use the transposed tile
// (two accumulator chains, q = 0, 1 and q = 2, 3: a dependent v_mfma_f64_16x16x4 waits ~20 cycles longer than the 64
//  its predecessor occupies the pipe; with one wave per SIMD nobody else fills that gap)
template <bool TA, bool TB>
__device__ __forceinline__ v4r mma(v4r acc, const real* A, const real* B, const Lane& o, real sa = 1) {
  real a[4], b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { a[q] = sa * A[TA ? o.cr[q] : o.rc[q]]; b[q] = B[TB ? o.rc[q] : o.cr[q]]; }
  v4r acc2 = {0, 0, 0, 0};
  acc = MFMAR(a[0], b[0], acc);
  acc2 = MFMAR(a[2], b[2], acc2);
  acc = MFMAR(a[1], b[1], acc);
  acc2 = MFMAR(a[3], b[3], acc2);
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
  return acc;
}

// (measurement only, tools/worker_bench.py) The same product with ONE accumulator chain and no fp64
// vector arithmetic at all.  fp64 vector operations issue on the matrix pipe and wait out the 64-cycle MFMA the other wave of the
// SIMD has in it, so the sign multiply and the merge of two chains (8 v_*_f64 per product) cost more than the ~14-cycle bubble
// of a dependent MFMA, which the partner wave fills anyway.  NEG: acc -= op(A) op(B), the sign flipped in the integer pipe.
__device__ __forceinline__ double flip_sign(double x) { return __hiloint2double(__double2hiint(x) ^ (int)0x80000000, __double2loint(x)); }
__device__ __forceinline__ float flip_sign(float x) { return __int_as_float(__float_as_int(x) ^ (int)0x80000000); }
template <bool TA, bool TB, bool NEG>
__device__ __forceinline__ v4r mma1(v4r acc, const real* A, const real* B, const Lane& o) {
  real a[4], b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { a[q] = A[TA ? o.cr[q] : o.rc[q]]; b[q] = B[TB ? o.rc[q] : o.cr[q]]; }
  // (all eight operand reads in flight before the first MFMA: left alone, the compiler re-uses ONE register pair per operand and
  //  issues the reads of step q + 1 after the MFMA of step q - four exposed LDS round trips per product)
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
#pragma unroll
  for (int q = 0; q < 4; ++q) acc = MFMAR(NEG ? flip_sign(a[q]) : a[q], b[q], acc);
  return acc;
}

// acc += sum_{k0 <= k < k1} A_k B_k.  la(k, q) / lb(k, q) return this lane's A / B operand of MFMA q of step k (LDS
// reads).  Two steps per trip: the sixteen operand reads of a pair of steps are requested together and their eight
// MFMAs run as two accumulator chains (a dependent v_mfma_f64_16x16x4 waits 78 cycles, its predecessor holds the pipe
// for 64), so a trip has one exposed LDS latency for 512 cycles of matrix-pipe work.
template <class LoadA, class LoadB>
__device__ __forceinline__ v4r acc_run(v4r acc, int k0, int k1, LoadA la, LoadB lb) {
  v4r acc2 = {0, 0, 0, 0};
  int k = k0;
  if (k + 1 < k1) {
    // Software pipeline (round 4): the operands of the NEXT pair of steps are requested before the eight MFMAs of the current pair
    // are issued.  Round 5: TWO operand sets that swap roles, two pairs of steps per trip - with one set and a copy ("a = next a")
    // the trip opened with seven to ten v_mov_b64 in a row in front of its first MFMA (hipcc -S), ~100 cycles per MFMA where the
    // same MFMAs from registers take 70 (tools/diag16_bench.py).  Same products, same order, same two accumulator chains.
    real pa[8], pb[8], qa[8], qb[8];
    auto load = [&](int kk, real (&a)[8], real (&b)[8]) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { a[q] = la(kk, q); b[q] = lb(kk, q); a[4 + q] = la(kk + 1, q); b[4 + q] = lb(kk + 1, q); }
    };
    auto mma8 = [&](const real (&a)[8], const real (&b)[8]) {
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        acc = MFMAR(a[q], b[q], acc);
        acc2 = MFMAR(a[q + 1], b[q + 1], acc2);
      }
    };
    load(k, pa, pb);
    k += 2;                                   // (invariant: the pair (k - 2, k - 1) is in pa / pb and not yet multiplied)
#pragma unroll 1
    for (; k + 3 < k1; k += 4) {
      load(k, qa, qb);
      mma8(pa, pb);
      load(k + 2, pa, pb);
      mma8(qa, qb);
    }
    if (k + 1 < k1) {
      load(k, qa, qb);
      mma8(pa, pb);
      mma8(qa, qb);
      k += 2;
    } else {
      mma8(pa, pb);
    }
  }
  if (k < k1) {
    real a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q] = la(k, q); b[q] = lb(k, q); }
    acc = MFMAR(a[0], b[0], acc);
    acc2 = MFMAR(a[1], b[1], acc2);
    acc = MFMAR(a[2], b[2], acc);
    acc2 = MFMAR(a[3], b[3], acc2);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
  return acc;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double softplus_d(double u) { return u > 30.0 ? u : log1p(exp(u)); }

// exp(x) for x <= 0 (the kernels only exponentiate -c r): n = rint(x log2 e), Cody-Waite reduction, degree-13 Taylor
// polynomial on |t| <= ln2 / 2 (truncation 4e-18), 2^n by v_ldexp_f64.  ~20 instructions against ~35 of the library exp.
__device__ __forceinline__ double exp_neg(double x) {
  x = x < -745.0 ? -745.0 : x;
  const double n = rint(x * 1.4426950408889634);
  double t = fma(n, -0.6931471803691238, x);
  t = fma(n, -1.9082149292705877e-10, t);
  double p = 1.6059043836821613e-10;                 // 1/13!
  p = fma(p, t, 2.08767569878681e-09);               // 1/12!
  p = fma(p, t, 2.505210838544172e-08);              // 1/11!
  p = fma(p, t, 2.755731922398589e-07);              // 1/10!
  p = fma(p, t, 2.7557319223985893e-06);             // 1/9!
  p = fma(p, t, 2.48015873015873e-05);               // 1/8!
  p = fma(p, t, 0.0001984126984126984);              // 1/7!
  p = fma(p, t, 0.001388888888888889);               // 1/6!
  p = fma(p, t, 0.008333333333333333);               // 1/5!
  p = fma(p, t, 0.041666666666666664);               // 1/4!
  p = fma(p, t, 0.16666666666666666);                // 1/3!
  p = fma(p, t, 0.5);
  p = fma(p, t, 1.0);
  p = fma(p, t, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
}
// sqrt(d), d >= 1e-36: v_rsq_f64 seed (2^-26) and two Goldschmidt corrections
__device__ __forceinline__ double sqrt_pos(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  double r = d * y;
  const double h = 0.5 * y;
  r = fma(fma(-r, r, d), h, r);
  r = fma(fma(-r, r, d), h, r);
  return r;
}
// The stationary kernels in unit-variance form (GPflow 2.1.5; Matern-3/2 is the one the reference instantiates,
// models.py:186):  K = variance * k1,  dK/dlengthscale = (variance / lengthscale) * cl * k2  with
//   Matern-3/2  k1 = (1 + sqrt3 r) e^{-sqrt3 r}             k2 = r^2 e^{-sqrt3 r}               cl = 3
//   Matern-5/2  k1 = (1 + sqrt5 r + 5/3 r^2) e^{-sqrt5 r}   k2 = r^2 (1 + sqrt5 r) e^{-sqrt5 r}  cl = 5/3
//   RBF         k1 = e^{-r2/2}                              k2 = r2 e^{-r2/2}                    cl = 1
// r = sqrt(max(r2, 1e-36)); entries on the clamp branch contribute no lengthscale gradient.
template <bool GRAD>
__device__ __forceinline__ void kern_unit(int kernel, double r2, double& k1, double& k2) {
  k2 = 0.0;
  if (kernel == 2) {
    const double r2c = r2 > 0.0 ? r2 : 0.0;
    k1 = exp_neg(-0.5 * r2c);
    if (GRAD) k2 = r2c * k1;
    return;
  }
  const bool clamp = r2 <= 1e-36;
  const double r = sqrt_pos(clamp ? 1e-36 : r2);
  if (kernel == 0) {
    const double e = exp_neg(-SQRT3 * r);
    k1 = fma(SQRT3, r, 1.0) * e;
    if (GRAD) k2 = clamp ? 0.0 : r * r * e;
  } else {
    const double e = exp_neg(-SQRT5 * r);
    const double q = fma(SQRT5, r, 1.0);
    k1 = fma((5.0 / 3.0) * r, r, q) * e;
    if (GRAD) k2 = clamp ? 0.0 : r * r * q * e;
  }
}
__device__ __forceinline__ double kern_cl(int kernel) { return kernel == 0 ? 3.0 : (kernel == 1 ? 5.0 / 3.0 : 1.0); }

// ---- 16x16 diagonal tile: Cholesky + inverse of the factor, one wave, the tile in ONE accumulator (4 doubles per
// lane).  Four sub-steps of four columns; sub-step s:
//   * register s of the (symmetric, fully updated) accumulator IS the 16x4 column panel V = A[:, 4s..4s+3] in MFMA
//     operand layout (lane (lr, lk) holds A[4s+lk][lr] = A[lr][4s+lk]);
//   * its 4x4 diagonal block (10 values) is broadcast with v_readlane and factored + inverted redundantly by every
//     lane: four Newton-refined v_rsq_f64 in a row are the only serial chain (16 per tile instead of 16 column steps);
//   * P = V M4^T (the factor's columns) is ONE v_mfma_f64_16x16x4 with A = M4 (zero-padded to 16x4), B = V: its result
//     register 0 is P in operand layout again, so the rank-4 update A -= P P^T is the next MFMA, with no lane movement;
//   * the inverse rides along: F starts as I, Q = M4 F[4s..4s+3, :] (= rows 4s.. of L^-1), F -= P Q - two more MFMAs
//     off the critical path.
// The factor leaves in operand layout (offsets rc[s]), the inverse in accumulator layout (cr[s]).  ~70 registers.
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(v), src_lane);
  const unsigned hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double((int)hi, (int)lo);
}
// 1/sqrt(d): v_rsq_f64 seed (relative error e0 ~ 2^-26) and ONE third-order correction y = y0 (1 + e/2 + 3 e^2/8),
// e = 1 - d y0^2: the error drops to ~e0^3, full double precision, in four dependent operations (two Newton steps take six)
__device__ __forceinline__ double rsqrt_refined(double d) {
  const double y0 = __builtin_amdgcn_rsq(d);
  const double e = fma(-(d * y0), y0, 1.0);
  return fma(y0 * e, fma(0.375, e, 0.5), y0);
}
// symmetric image of a diagonal tile (only its lower triangle is trusted): in the accumulator layout of the fit's
// precision, and in the fp64 accumulator layout (rows lk + 4r) the factorisation below works in
__device__ __forceinline__ v4r ld_tile_sym(const real* T, const Lane& o) {
  v4r acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = o.kq(r), col = o.lr();
    acc[r] = T[row >= col ? swz(row, col) : swz(col, row)];
  }
  return acc;
}
__device__ __forceinline__ v4d ld_tile_sym64(const real* T, const Lane& o) {
  v4d acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = o.lk() + 4 * r, col = o.lr();
    acc[r] = (double)T[row >= col ? swz(row, col) : swz(col, row)];
  }
  return acc;
}
// acc: the symmetric tile (fp64 accumulator layout, registers); the factor goes to T, its inverse to Dk (fp64 in both
// precisions of the fit: "fp64 accumulate in chol")
__device__ __forceinline__ void diag16(v4d acc, real* T, real* Dk, int lane, int pivot_base, int& bad, const Lane& o) {
  __builtin_amdgcn_s_setprio(3);        // the serial chain of the workgroup: win issue arbitration on its SIMD
  v4d fac;
#pragma unroll
  for (int r = 0; r < 4; ++r) fac[r] = o.lk() + 4 * r == o.lr() ? 1.0 : 0.0;
  double lcol[4], xrow[4];
  // sel[e] = 1 in the one lane (lr < 4, lk <= lr) that holds entry e of the lower 4x4 block, in the order
  // (0,0) (1,1) (1,0) (2,2) (2,1) (2,0) (3,3) (3,2) (3,1) (3,0)
  double sel[10];
  {
    const int code = o.lr() < 4 ? 4 * o.lr() + o.lk() : -1;
    const int at[10] = {0, 5, 4, 10, 9, 8, 15, 14, 13, 12};
#pragma unroll
    for (int e = 0; e < 10; ++e) sel[e] = code == at[e] ? 1.0 : 0.0;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const double v = acc[s];
    // d_ab = A[4s+a][4s+b] sits in lane (lk = a, lr = 4s + b)
    const double d00 = readlane_f64(v, 4 * s), d10 = readlane_f64(v, 16 + 4 * s), d20 = readlane_f64(v, 32 + 4 * s),
                 d30 = readlane_f64(v, 48 + 4 * s), d11 = readlane_f64(v, 16 + 4 * s + 1), d21 = readlane_f64(v, 32 + 4 * s + 1),
                 d31 = readlane_f64(v, 48 + 4 * s + 1), d22 = readlane_f64(v, 32 + 4 * s + 2), d32 = readlane_f64(v, 48 + 4 * s + 2),
                 d33 = readlane_f64(v, 48 + 4 * s + 3);
    const int pb = pivot_base + 4 * s;
    // (a non-positive pivot is recorded, not replaced: the NaNs it produces stay inside this member, every later pivot
    //  test fails too and the first one is what info[] reports - the test is then off the dependent chain)
    const double t0 = d00;
    const double r0 = rsqrt_refined(t0);
    const double l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
    const double t1 = fma(-l10, l10, d11);
    const double r1 = rsqrt_refined(t1);
    const double l21 = fma(-l20, l10, d21) * r1, l31 = fma(-l30, l10, d31) * r1;
    const double t2 = fma(-l21, l21, fma(-l20, l20, d22));
    const double r2 = rsqrt_refined(t2);
    const double l32 = fma(-l31, l21, fma(-l30, l20, d32)) * r2;
    const double t3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33)));
    const double r3 = rsqrt_refined(t3);
    if (bad == 0) bad = !(t0 > 0.0) ? pb + 1 : (!(t1 > 0.0) ? pb + 2 : (!(t2 > 0.0) ? pb + 3 : (!(t3 > 0.0) ? pb + 4 : 0)));
    // M4 = L4^-1 (lower): m_aa = r_a
    const double m10 = -l10 * r0 * r1;
    const double m21 = -l21 * r1 * r2;
    const double m20 = -fma(l21, m10, l20 * r0) * r2;
    const double m32 = -l32 * r2 * r3;
    const double m31 = -fma(l32, m21, l31 * r1) * r3;
    const double m30 = -fma(l32, m20, fma(l31, m10, l30 * r0)) * r3;
    // this lane's entry of the zero-padded 16x4 A operand: M4[lr][lk] for lr < 4, picked WITHOUT branches: the ten
    // candidates are multiplied by 0/1 lane constants (a select tree compiles to ten divergent branches, ~3x the cost
    // of the arithmetic of the whole sub-step)
    const double ia = fma(sel[9], m30, fma(sel[8], m31, fma(sel[7], m32, fma(sel[6], r3,
                      fma(sel[5], m20, fma(sel[4], m21, fma(sel[3], r2, fma(sel[2], m10, fma(sel[1], r1, sel[0] * r0)))))))));
    const v4d zero = {0.0, 0.0, 0.0, 0.0};
    const v4d pt = MFMA(ia, v, zero);                          // register 0 at lane (lr, lk): P[lr][lk] = L[lr][4s+lk]
    const double pm = o.lr() >= 4 * s + o.lk() ? pt[0] : 0.0;      // rows above the diagonal (and finished rows) are zero
    lcol[s] = pm;
    const v4d qt = MFMA(ia, fac[s], zero);                     // register 0: Q[lk][lr] = Linv[4s+lk][lr]
    xrow[s] = qt[0];
    if (s < 3) {
      acc = MFMA(-pm, pm, acc);                                // A -= P P^T
      fac = MFMA(-pm, qt[0], fac);                             // F -= P Q
    }
  }
  // every lane holds 4 entries of the factor (operand layout) and 4 of the inverse (accumulator layout)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    T[swz(o.lr(), 4 * s + o.lk())] = (real)lcol[s];
    Dk[swz(4 * s + o.lk(), o.lr())] = (real)xrow[s];
  }
  __builtin_amdgcn_s_setprio(0);
}

// ---- blocked Cholesky of the lower-packed buffer A in place; D[k] = inverse of diagonal tile k.  WITH_INV: the
// inverse of the factor is formed alongside in Xb by block forward substitution (X_ji accumulates -sum_k L_jk X_ki and
// is finished by inv_jj when row j is reached): no extra serial chain, its products ride in the same phases.
// Two phases per block column with a one-step look-ahead: [panel of column k (every wave)] [wave 0: update of the next
// diagonal tile and its factorisation | waves 1..7: the rest of the trailing update], so the serial 16x16 factorisation
// hides the whole trailing update.
// (not inlined: the factorisation keeps ~110 registers busy; as a function of its own it is compiled without the
//  caller's register tiles in the way, which are saved and restored once per call instead of spilling all along)
// `extra(k)`: work that only needs block row k of the finished factor (and of its inverse, WITH_INV).  Waves 1..7 run it
// behind their share of the trailing update of step k, i.e. in the shadow of wave 0's serial factorisation of the next
// diagonal tile; row NT-1 after the last panel.  (The callers accumulate L^T D L and C^-T C^-1 this way.)
struct NoExtra { __device__ __forceinline__ void operator()(int) const {} };
template <int NT, bool WITH_INV, class Extra = NoExtra>
__device__ __forceinline__ int potrf_lds(int a_off, int d_off, int x_off, const Lane& o, int wave, int lane, int bad,
                                         Extra extra = Extra(), bool w4idle = true) {
  // (LDS operands arrive as offsets into the workgroup's dynamic LDS block: pointers handed to a real function call
  //  would be generic pointers and every access a flat one)
  real* A = tiles_base() + a_off;
  real* D = tiles_base() + d_off;
  real* Xb = tiles_base() + x_off;
  if (wave == 0) diag16(ld_tile_sym64(A, o), A, D, lane, 0, bad, o);
  SYNC();
#pragma unroll 1
  for (int k = 0; k < NT; ++k) {
    const int below = NT - 1 - k;
    const real* dk = D + k * TS;
#pragma unroll 1
    for (int q = wave; q < below + (WITH_INV ? k + 1 : 0); q += NW) {
      if (q < below) {                                    // panel: L_ik = A_ik inv_kk^T
        real* t = A + lt(k + 1 + q, k) * TS;
        v4r acc = {0, 0, 0, 0};
        acc = mma<false, true>(acc, t, dk, o);
        st_tile(t, o, acc);
      } else {
        const int i = q - below;
        real* t = Xb + lt(k, i) * TS;
        if (i < k) {                                      // X_ki = inv_kk T_ki
          v4r acc = {0, 0, 0, 0};
          acc = mma<false, false>(acc, dk, t, o);
          st_tile(t, o, acc);
        } else {
          st_tile(t, o, ld_tile(dk, o));                  // X_kk = inv_kk
        }
      }
    }
    SYNC();
    if (below == 0) break;
    if (wave == 0) {                                      // look-ahead: next diagonal tile first, then its factorisation
      real* t = A + lt(k + 1, k + 1) * TS;
      v4r acc = ld_tile_sym(t, o);
      acc = mma<false, true>(acc, A + lt(k + 1, k) * TS, A + lt(k + 1, k) * TS, o, -1);
#ifdef BAYENS_F32
      st_tile(t, o, acc);                                 // (fp32 -> fp64 accumulator layout: through the wave's own LDS tile)
      diag16(ld_tile_sym64(t, o), t, D + (k + 1) * TS, lane, 16 * (k + 1), bad, o);
#else
      diag16(acc, t, D + (k + 1) * TS, lane, 16 * (k + 1), bad, o);   // the updated tile goes straight into the factorisation
#endif
    } else {
      // The trailing tiles (i, j), i = k+2 .. NT-1, j = k+1 .. i ((k+1, k+1) is wave 0's), and behind them the inverse's tiles
      // (j, i), j = k+1 .. NT-1, i = 0 .. k, form ONE sequence dealt round-robin over waves 1..NW-1.  A wave steps through
      // its own positions directly, NW-1 at a time (round 3: scanning the whole sequence with a modulo test per candidate
      // cost more scalar time than the tile updates it found - measured in medium.hip, where this loop was first fixed).
      const int n_trail = (below - 1) * (below + 2) / 2;                      // sum_{i = k+2}^{NT-1} (i - k)
      // Wave 4 shares its SIMD with wave 0 (waves are dealt round-robin over the four SIMDs), and a v_fma_f64 of the serial
      // 16x16 chain waits out every fp64 MFMA its SIMD partner has in the pipe (tools/diag16_bench.py: the chain takes
      // 8.1 k cycles beside MFMA-issuing partners, 4.0 k alone): wave 4 takes no tile updates in this phase, the other six do.
      const int NWK = w4idle ? NW - 2 : NW - 1;
      int pos = (w4idle && wave > 4) ? wave - 2 : wave - 1;
      if (w4idle && wave == 4) pos = 1 << 20;
      {
        int i = k + 2, j = k + 1 + pos;                                       // row i holds i - k tiles
        while (i < NT && j > i) { j -= i - k; ++i; }
#pragma unroll 1
        while (pos < n_trail) {                                               // A_ij -= L_ik L_jk^T
          real* t = A + lt(i, j) * TS;
          v4r acc = ld_tile(t, o);
          acc = mma<false, true>(acc, A + lt(i, k) * TS, A + lt(j, k) * TS, o, -1);
          st_tile(t, o, acc);
          pos += NWK;
          j += NWK;
          while (i < NT && j > i) { j -= i - k; ++i; }
        }
      }
      if (WITH_INV && !(w4idle && wave == 4)) {
        const int n_inv_ops = below * (k + 1);
        int q = pos - n_trail;                                                // position inside the inverse's part of the sequence
        int j = k + 1 + q / (k + 1), i = q - (j - (k + 1)) * (k + 1);
#pragma unroll 1
        while (q < n_inv_ops) {                                               // T_ji -= L_jk X_ki   (first touched at k == i)
          real* t = Xb + lt(j, i) * TS;
          v4r acc = {0, 0, 0, 0};
          if (k > i) acc = ld_tile(t, o);
          acc = mma<false, false>(acc, A + lt(j, k) * TS, Xb + lt(k, i) * TS, o, -1);
          st_tile(t, o, acc);
          q += NWK;
          i += NWK;
          while (i > k) { i -= k + 1; ++j; }
        }
      }
      extra(k);
    }
    SYNC();
  }
  if (wave != 0) extra(NT - 1);
  return bad;
}

// ---- matrix-vector products on tile-packed buffers: wave w < NT owns the 16 outputs of tile row / column w, lane
// (lr, lk) walks the columns == lk mod 4, the four partial sums meet through two shuffles.  Valid in lanes lk == 0.
template <int NT>
__device__ __forceinline__ double mv_lower(const real* Lb, const vecr* x, int i, const Lane& o) {     // (L x)[16 i + lr]
  double s = 0.0;
#pragma unroll 2
  for (int k = 0; k <= i; ++k) {
    const real* t = Lb + lt(i, k) * TS;
#pragma unroll
    for (int q = 0; q < 4; ++q) s = fma((double)t[o.rc[q]], (double)x[16 * k + o.kq(q)], s);
  }
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}
template <int NT>
__device__ __forceinline__ double mv_lower_t(const real* Lb, const vecr* x, int j, const Lane& o) {   // (L^T x)[16 j + lr]
  double s = 0.0;
#pragma unroll 2
  for (int k = j; k < NT; ++k) {
    const real* t = Lb + lt(k, j) * TS;
#pragma unroll
    for (int q = 0; q < 4; ++q) s = fma((double)t[o.cr[q]], (double)x[16 * k + o.kq(q)], s);
  }
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}
template <int NT>
__device__ __forceinline__ double mv_sym(const real* Sb, const vecr* x, int i, const Lane& o) {       // (S x)[16 i + lr]
  double s = 0.0;
#pragma unroll 1
  for (int k = 0; k < NT; ++k) {
    const real* t = Sb + (k <= i ? lt(i, k) : lt(k, i)) * TS;
#pragma unroll
    for (int q = 0; q < 4; ++q) s = fma((double)t[k <= i ? o.rc[q] : o.cr[q]], (double)x[16 * k + o.kq(q)], s);
  }
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}

// squared distances of tile (i, j): X X^T on the matrix cores with operands straight from the member's X,
// r2 = ((-2 dot + |x_i|^2) + |x_j|^2) / ls^2 - GPflow / the oracle scale X first, this scales the distance (the
// rounding differs in the last bits only; ssq = |x_i|^2 is then a per-member constant)
// (always an fp64 product; result register r is the pair (row kq[r], column lr) of the tile in either precision)
__device__ __forceinline__ v4d r2_tile(const real* X, int N, int R, double inv_ls2, const double* ssq, int i, int j, const Lane& o) {
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  const int ra = 16 * i + dist_row(o.lr()), rb = 16 * j + o.lr();
  const real* xa = X + (long)(ra < N ? ra : 0) * R;
  const real* xb = X + (long)(rb < N ? rb : 0) * R;
  if (R <= 8) {                 // the common case: both operand pairs requested before the first MFMA
    const int c0 = o.lk(), c1 = 4 + o.lk();
    // (unconditional loads from clamped addresses, zeroed afterwards: a load under a condition is a branch whose value is
    //  awaited at the join - four serial round trips instead of one)
    const double la0 = (double)GLD(xa[c0 < R ? c0 : 0]), lb0 = (double)GLD(xb[c0 < R ? c0 : 0]);
    const double la1 = (double)GLD(xa[c1 < R ? c1 : 0]), lb1 = (double)GLD(xb[c1 < R ? c1 : 0]);
    const double a0 = (ra < N && c0 < R) ? la0 : 0.0, b0 = (rb < N && c0 < R) ? lb0 : 0.0;
    const double a1 = (ra < N && c1 < R) ? la1 : 0.0, b1 = (rb < N && c1 < R) ? lb1 : 0.0;
    acc = MFMA(a0, b0, acc);
    if (R > 4) acc = MFMA(a1, b1, acc);
  } else {
#pragma unroll 2
    for (int k0 = 0; k0 < R; k0 += 4) {
      const int c = k0 + o.lk();
      const double la = (double)GLD(xa[c < R ? c : 0]), lb = (double)GLD(xb[c < R ? c : 0]);
      const double a = (ra < N && c < R) ? la : 0.0;
      const double b = (rb < N && c < R) ? lb : 0.0;
      acc = MFMA(a, b, acc);
    }
  }
  const double sj = ssq[16 * j + o.lr()];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = ((-2.0 * acc[r] + ssq[16 * i + o.kq(r)]) + sj) * inv_ls2;
  return acc;
}

// (medium.hip includes this file for the tile helpers above and stops here)
#ifndef BAYENS_SMALL_HELPERS_ONLY
template <int NT>
struct Layout {
  static constexpr int NTT = NT * (NT + 1) / 2;
  static constexpr int NP = 16 * NT;
  static constexpr int TPW = (NTT + NW - 2) / (NW - 1);  // register tiles per wave: dealt over waves 1..NW-1 (wave 0 runs the serial chains)
  // tile regions (offsets in elements of `real` from tiles_base())
  static constexpr int LB = 0, CB = NTT * TS, WB = 2 * NTT * TS, DL = 3 * NTT * TS, DC = DL + NT * TS;
  static constexpr int TILE_ELEMS = DC + NT * TS;
  // behind them (offsets in doubles): |x_i|^2 and 64 scalars in fp64, then NVEC vectors of NP elements of `vecr`
  static constexpr int SSQ = (int)((size_t)TILE_ELEMS * sizeof(real) / sizeof(double));
  static constexpr int MISC = SSQ + NP;
  static constexpr int VEC = MISC + 64;
  static constexpr int NVEC = 9;
  static constexpr size_t BYTES = (size_t)VEC * sizeof(double) + (size_t)NVEC * NP * sizeof(vecr);
};

template <int NT>
__global__ __launch_bounds__(NTH, 2) void small_fit_kernel(SmallArgs g_args) {   // two waves per SIMD: 256 registers each
  // The arguments are READ WHERE THEY ARE USED, from the kernel-argument block (scalar loads from constant memory), not held in
  // ~50 scalar registers from entry to exit: what does not fit is then reloaded from there instead of spilled into vector lanes
  // (NT = 6: 467 -> 388 scalar spills, 2686 -> 1851 v_readlane in the listing, 172 -> 140 B of private segment; the run time of
  // config 5 does not move - 40.6 ms either way: the spill traffic was not what it waits for).
  (void)g_args;
  typedef const __attribute__((address_space(4))) SmallArgs* KArgs;
  const __attribute__((address_space(4))) SmallArgs& g = *(KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  typedef Layout<NT> Y;
  constexpr int NTT = Y::NTT, NP = Y::NP, TPW = Y::TPW;
  double* lds = lds_base();
  real* Lb = tiles_base() + Y::LB;
  real* Cb = tiles_base() + Y::CB;
  real* Wb = tiles_base() + Y::WB;
  real* DLi = tiles_base() + Y::DL;     // inverted diagonal tiles of L; in the prediction also staging (odd tile rows)
  real* DCi = tiles_base() + Y::DC;     // inverted diagonal tiles of C; in the prediction: staging of one tile row of A^T S
  double* ssq = lds + Y::SSQ;           // |x_i|^2 (fp64: the squared distances cancel)
  double* misc = lds + Y::MISC;         // [0..7] optimiser state, [8] variance, [9] lengthscale, [10] 1/ls^2, [16..47] reductions
  vecr* yv = reinterpret_cast<vecr*>(lds + Y::VEC);   // y (0 in the padding)
  vecr* dv = yv + NP;                   // 1 / nu (1 in the padding)
  vecr* rs = dv + NP;                   // 1 / sqrt(nu)
  vecr* yd = rs + NP;                   // y / nu
  vecr* th = yd + NP;                   // theta1
  vecr* mv = th + NP;                   // m
  vecr* av = mv + NP;                   // loss samples: L m - y; prediction: the posterior mean
  vecr* wv = av + NP;                   // L^T (y / nu) of this iteration (loss samples: diag(L S L^T))
  vecr* uv = wv + NP;                   // u = L^T D (L m - y) = G m - L^T (y / nu)
  // (the wave index is made visibly wave-uniform: everything derived from it - tile indices, LDS tile addresses, the branches of
  //  the tile-to-wave tables - stays in scalar registers)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Lane o;
  lane_init(o, lane);
  const int N = g.N;
  const double J = g.jitter;
  const bool w4idle = !(g.tune & 1);

#pragma unroll 1
  for (int b = blockIdx.x; b < g.B; b += gridDim.x) {
    const int R = g.Rv ? g.Rv[b] : g.R;                                   // (a ragged batch: this member's own realisation count)
    const real* X = g.X + (g.xoff ? g.xoff[b] : (long)b * N * g.R);
    int bad = 0;
    // ---- member set-up ----
    for (int t = tid; t < NP; t += NTH) {
      const double nu = t < N ? (double)g.nu[(long)b * N + t] : 1.0;
      const double y = t < N ? (double)g.y[(long)b * N + t] : 0.0;
      yv[t] = (vecr)y; dv[t] = (vecr)(1.0 / nu); rs[t] = (vecr)(1.0 / sqrt(nu)); yd[t] = (vecr)(y / nu);
      th[t] = 0; mv[t] = 0;
      double sq = 0.0;                  // fma chain in k order = the diagonal of the MFMA product: r2_ii is exactly 0
      if (t < N)
        for (int d = 0; d < R; ++d) { const double v = (double)X[(long)t * R + d]; sq = fma(v, v, sq); }
      ssq[t] = sq;
    }
    if (tid < 8) misc[tid] = tid < 2 ? 0.5413248546129181 /* softplus^-1(1) = log(e - 1) */ : 0.0;
    v4r lam[TPW], gt[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      const int p = wave == 0 ? NTT : (wave - 1) + s * (NW - 1);
      int i = 0, j = 0;
      unpack_lower(p < NTT ? p : 0, i, j);
#pragma unroll
      for (int r = 0; r < 4; ++r) lam[s][r] = (i == j && o.kq(r) == o.lr()) ? 1 : 0;
      gt[s] = v4r{0, 0, 0, 0};
      if (p < NTT) st_tile(Cb + p * TS, o, lam[s]);        // S = I until the first natural-gradient step (n_iters = 0)
    }
    SYNC();

    // hyper-parameters of the coming kernel matrix -> misc[8..10]
    auto refresh_hypers = [&]() {
      WSTAMP(8);
      if (tid == 0) {
        const double ls = softplus_d(misc[1]);
        misc[8] = softplus_d(misc[0]); misc[9] = ls; misc[10] = 1.0 / (ls * ls);
      }
      SYNC();
    };
    // Lb <- tril(K0) + J I (identity in the padding)
    // (with_m: also Cb <- tril(gamma (K0 + J I) + diag(nu)), the matrix whose inverse the first iteration's gradient needs)
    // `park`: the unit-variance kernel values (k1, k2) of every tile go to the workgroup's park in device memory (accumulator
    // order, one coalesced 512-byte row per register) for this iteration's hyper-gradient, which then evaluates no sqrt / exp
    // (round 4: the five kernel-function passes of a T = 2 fit were 17 % of it, and they are bound by the fp64 vector rate -
    //  tools/valu64_bench.py: 125 cycles per exp, 74 per sqrt and wave, 6.5 per v_fma_f64 and SIMD - not by their operand loads:
    //  requesting the X operands of all tiles ahead of the hypers, or evaluating the hypers in every lane, moved nothing / cost more)
    double* kpark = g.kpark + (size_t)blockIdx.x * (2 * NTT * TS);
    auto kmat = [&](bool with_m = false, bool park = false) {
      const double var = misc[8], inv_ls2 = misc[10];
      WSTAMP(9);
#pragma unroll 1
      for (int p = wave; p < NTT; p += NW) {
        int i, j;
        unpack_lower(p, i, j);
        const v4d r2 = r2_tile(X, N, R, inv_ls2, ssq, i, j, o);
        if (p == wave) WSTAMP(10);
        v4r kt;
        double k1m[4], k2m[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * i + o.kq(r), col = 16 * j + o.lr();
          double k1, k2;
          if (park) kern_unit<true>(g.kernel, r2[r], k1, k2);
          else kern_unit<false>(g.kernel, r2[r], k1, k2);
          double K = var * k1;
          if (row >= N || col >= N) K = row == col ? 1.0 : 0.0;
          else K = col <= row ? K + (row == col ? J : 0.0) : 0.0;
          kt[r] = (real)K;
          k1m[r] = (row < N && col < N) ? k1 : 0.0;            // (the padding contributes no gradient)
          k2m[r] = (row < N && col < N) ? k2 : 0.0;
        }
        if (park) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            kpark[(size_t)(2 * p) * TS + r * 64 + lane] = k1m[r];
            kpark[(size_t)(2 * p + 1) * TS + r * 64 + lane] = k2m[r];
          }
        }
        st_tile(Lb + p * TS, o, kt);
        if (p == wave) WSTAMP(11);
        if (with_m) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * i + o.kq(r), col = 16 * j + o.lr();
            kt[r] = (real)(g.gamma * (double)kt[r] + (row == col ? 1.0 / (double)dv[row] : 0.0));
          }
          st_tile(Cb + p * TS, o, kt);
        }
      }
      WSTAMP(12);
      SYNC();
      WSTAMP(13);
    };

    // -ELBO (the training loss the reference samples after iterations 0, 25, 50, ..., models.py:211-215; GPflow VGP.elbo with
    // the heteroskedastic Gaussian expectations of models.py:142-149) at the hypers whose factor L is in Lb, with the
    // (m, S) of the last natural-gradient step (S in Cb, log det C in misc[11]):
    //   sum_t [ log(2 pi)/2 + log(nu_t)/2 + ((y_t - (L m)_t)^2 + (L S L^T)_tt) / (2 nu_t) ] + (|m|^2 - N + 2 log det C + tr S) / 2
    // (L S L^T)_tt = sum_j (L S)_tj L_tj: wave i forms the tiles (i, j <= i) of L S and folds them against L.
    auto neg_elbo = [&](int slot) {
#pragma unroll 1
      for (int i = wave; i < NT; i += NW) {
        const double s = mv_lower<NT>(Lb, mv, i, o);
        if (o.lk() == 0) av[16 * i + o.lr()] = (vecr)(s - (double)yv[16 * i + o.lr()]);
        double fr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int j = 0; j <= i; ++j) {
          v4r acc = {0, 0, 0, 0};
          acc = acc_run(acc, 0, i + 1, [&](int k, int q) { return Lb[lt(i, k) * TS + o.rc[q]]; },
                        [&](int k, int q) { return Cb[k >= j ? lt(k, j) * TS + o.cr[q] : lt(j, k) * TS + o.rc[q]]; });
          const v4r lt_ = ld_tile(Lb + lt(i, j) * TS, o);
#pragma unroll
          for (int r = 0; r < 4; ++r) fr[r] = fma((double)acc[r], (double)lt_[r], fr[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double v = fr[r];
          v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
          if (o.lr() == 0) wv[16 * i + o.kq(r)] = (vecr)v;                   // f_t
        }
      }
      SYNC();
      double part = 0.0;
      for (int t = tid; t < N; t += NTH) {
        const double nu = 1.0 / (double)dv[t], a = av[t], m = mv[t];
        const double stt = (double)Cb[lt(t >> 4, t >> 4) * TS + swz(t & 15, t & 15)];
        part += 0.5 * 1.8378770664093453 + 0.5 * log(nu) + 0.5 * (a * a + (double)wv[t]) * (double)dv[t] + 0.5 * (m * m + stt);
      }
      part = wave_sum(part);
      if (lane == 0) misc[16 + wave] = part;
      SYNC();
      if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < NW; ++w) tot += misc[16 + w];
        g.loss[(long)b * g.loss_stride + slot] = tot - 0.5 * N + misc[11];
      }
      SYNC();
    };

#pragma unroll 1
    for (int it = 0; it < g.n_iters; ++it) {
      STAMP(0);
      // First iteration: Lam_old = I exactly and theta1_old = 0, so Lam = I + gamma G, G S = (I - S) / gamma, u = -m / gamma
      // and M = u m^T + G S is symmetric: Phi(M) + Phi(M)^T = M collapses the Cholesky backpropagation to
      //   sym(Kbar) = L^-T M L^-1 / 2 = (P - gamma (P y)(P y)^T) / 2,   P = (diag(nu) + gamma (K0 + J I))^-1
      // (Woodbury: S = I - gamma L^T P L, m = gamma L^T P y) - no C, S, Phi, W or Kbar in that iteration (csrc/driver.cpp has
      // the same form).  Kept to the general form when C itself is needed: a single iteration (the prediction uses it), the loss
      // sample of iteration 0 (there fp64 still takes G S = (I - S) / gamma from the S tiles; fp32 does not: I - S cancels).
      const bool wb = it == 0 && g.gamma > 0.0 && g.n_iters >= 2 && !g.loss;
      const bool first = it == 0 && g.gamma > 0.0 && sizeof(real) == sizeof(double);
      refresh_hypers();
      kmat(wb, true);                                                      // K0 + J I at the current hypers  (models.py:209)
      STAMP(1);
      // L = chol(K0 + J I); G = L^T D L = sum_k (D^1/2 L)_k*^T (D^1/2 L)_k* is accumulated block row by block row behind
      // the factorisation (registers gt[], tiles dealt over waves 1..7)
#pragma unroll
      for (int s = 0; s < TPW; ++s) gt[s] = v4r{0, 0, 0, 0};
      auto g_rows = [&](int k) {
#pragma unroll
        for (int s = 0; s < TPW; ++s) {
          const int p = (wave - 1) + s * (NW - 1);
          int i = 0, j = 0;
          unpack_lower(p < NTT ? p : 0, i, j);
          if (p < NTT && i <= k) {
            const real* ta = Lb + lt(k, i) * TS;
            const real* tb = Lb + lt(k, j) * TS;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const real sc = (real)rs[16 * k + o.kq(q)];
              gt[s] = MFMAR(ta[o.cr[q]] * sc, tb[o.cr[q]] * sc, gt[s]);
            }
          }
        }
      };
      if (wb) bad = potrf_lds<NT, false>(Y::LB, Y::DL, 0, o, wave, lane, bad, g_rows, w4idle);  // (nothing in that iteration uses L^-1)
      else bad = potrf_lds<NT, true>(Y::LB, Y::DL, Y::WB, o, wave, lane, bad, g_rows, w4idle);  // L in Lb, L^-1 rides along into Wb
      if (g.loss && it > 0 && (it - 1) % 25 == 0) neg_elbo((it - 1) / 25);  // after the Adam step = at these hypers
      STAMP(2);
      // ---- natural-gradient step: Lam <- (1-g) Lam + g (I + G), theta1 <- (1-g) theta1 + g L^T (y / nu)
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int p = wave == 0 ? NTT : (wave - 1) + s * (NW - 1);
        if (p < NTT) {
          int i, j;
          unpack_lower(p, i, j);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            lam[s][r] = (real)((1.0 - g.gamma) * lam[s][r] + g.gamma * (gt[s][r] + ((i == j && o.kq(r) == o.lr()) ? 1.0 : 0.0)));
          if (!wb) st_tile(Cb + p * TS, o, lam[s]);
        }
      }
#pragma unroll 1
      for (int i = wave; i < NT; i += NW) {
        const double s = mv_lower_t<NT>(Lb, yd, i, o);
        if (o.lk() == 0) {
          th[16 * i + o.lr()] = (vecr)((1.0 - g.gamma) * (double)th[16 * i + o.lr()] + g.gamma * s);
          wv[16 * i + o.lr()] = (vecr)s;         // L^T (y / nu): u = L^T D (L m - y) = G m - L^T (y / nu) needs it, and L dies here
        }
      }
      SYNC();
      STAMP(3);
      // C = chol(Lam) in Cb, C^-1 in Lb; S = C^-T C^-1 = sum_k (C^-1)_k*^T (C^-1)_k* accumulated behind it (registers st[])
      // (wb: the same for diag(nu) + gamma K - its factor, its inverse factor, and P in st[])
      v4r st[TPW];
#pragma unroll
      for (int s = 0; s < TPW; ++s) st[s] = v4r{0, 0, 0, 0};
      auto s_rows = [&](int k) {
#pragma unroll
        for (int s = 0; s < TPW; ++s) {
          const int p = (wave - 1) + s * (NW - 1);
          int i = 0, j = 0;
          unpack_lower(p < NTT ? p : 0, i, j);
          if (p < NTT && i <= k) st[s] = mma<true, false>(st[s], Lb + lt(k, i) * TS, Lb + lt(k, j) * TS, o);
        }
      };
      bad = potrf_lds<NT, true>(Y::CB, Y::DC, Y::LB, o, wave, lane, bad, s_rows, w4idle);      // (L is dead: C^-1 takes its buffer)
      STAMP(4);
      if (!wb) {
        // ---- m = C^-T (C^-1 theta1) through the inverse FACTOR (two triangular tile mat-vecs while C^-1 still lies in Lb), not
        // S theta1: with a variance of ~1e-11 at one time step (two nearly coincident realisations) |theta1| ~ |G| ~ 1e11, the
        // rounding of the explicit S alone (1e-16) puts 1e-5 into m, and u = G m - L^T (y / nu) below multiplies that by
        // |G| again - the fit came out O(1) wrong (profiles/r03_fuzz_seed32.log case 21; tests/golden/illcond_*.npz hold
        // the 50-digit answer).  Through the factor the error of m stays in the directions S damps: cond(Lam) eps overall.
#pragma unroll 1
        for (int i = wave; i < NT; i += NW) {
          const double s = mv_lower<NT>(Lb, th, i, o);
          if (o.lk() == 0) uv[16 * i + o.lr()] = (vecr)s;                      // z = C^-1 theta1 (uv is free until u is formed)
        }
        SYNC();
#pragma unroll 1
        for (int i = wave; i < NT; i += NW) {
          const double s = mv_lower_t<NT>(Lb, uv, i, o);
          if (o.lk() == 0) mv[16 * i + o.lr()] = (vecr)s;
        }
        // (mv is read after the barriers of the tile moves below)
      }
      if (g.loss && it % 25 == 0) {                                         // log det C of this step, for its loss sample
        double s = 0.0;
        for (int t = tid; t < N; t += NTH) s += log((double)Cb[lt(t >> 4, t >> 4) * TS + swz(t & 15, t & 15)]);
        s = wave_sum(s);
        if (lane == 0) misc[32 + wave] = s;
        SYNC();
        if (tid == 0) {
          double tot = 0.0;
          for (int w = 0; w < NW; ++w) tot += misc[32 + w];
          misc[11] = tot;
        }
      }
      // ---- S over C in Cb, G over C^-1 in Lb
      SYNC();
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int p = wave == 0 ? NTT : (wave - 1) + s * (NW - 1);
        if (p < NTT) { st_tile(Cb + p * TS, o, st[s]); st_tile(Lb + p * TS, o, gt[s]); }     // S -> Cb, G -> Lb
      }
      SYNC();
      STAMP(5);
      if (wb) {                                                             // p = P y -> uv
#pragma unroll 1
        for (int i = wave; i < NT; i += NW) {
          const double s = mv_sym<NT>(Cb, yv, i, o);
          if (o.lk() == 0) uv[16 * i + o.lr()] = (vecr)s;
        }
        SYNC();
      } else {
      // ---- u = L^T D (L m - y) = G m - L^T (y / nu)   (m: above, from the inverse factor)
#pragma unroll 1
      for (int i = wave; i < NT; i += NW) {
        const double s = mv_sym<NT>(Lb, mv, i, o);
        if (o.lk() == 0) uv[16 * i + o.lr()] = (vecr)(s - (double)wv[16 * i + o.lr()]);
      }
      SYNC();
      STAMP(6);
      // ---- Phi(L^T Lbar) = tril(u m^T + G S), half the diagonal   (G in Lb, S in Cb, both symmetric from lower tiles)
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int p = wave == 0 ? NTT : (wave - 1) + s * (NW - 1);
        if (p < NTT) {
          int i, j;
          unpack_lower(p, i, j);
          v4r acc = {0, 0, 0, 0};
          if (first) {
            // first iteration: Lam_old = I exactly, so Lam = I + gamma G and G S = (Lam - I) Lam^-1 / gamma = (I - S) / gamma;
            // this wave still holds its S tile in st[s] (same element layout) - no product
#pragma unroll
            for (int r = 0; r < 4; ++r)
              acc[r] = (real)((((i == j && o.kq(r) == o.lr()) ? 1.0 : 0.0) - (double)st[s][r]) / g.gamma);
          } else {
            acc = acc_run(acc, 0, NT,
                          // (ONE load per operand with a selected address: a choice between two loads is a branch, and operand loads
                          //  in separate basic blocks cannot be issued together - every one of them then exposes its LDS latency)
                          [&](int k, int q) { return Lb[k <= i ? lt(i, k) * TS + o.rc[q] : lt(k, i) * TS + o.cr[q]]; },
                          [&](int k, int q) { return Cb[k >= j ? lt(k, j) * TS + o.cr[q] : lt(j, k) * TS + o.rc[q]]; });
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rl = o.kq(r);
            double v = fma((double)uv[16 * i + rl], (double)mv[16 * j + o.lr()], (double)acc[r]);
            if (i == j) v = rl < o.lr() ? 0.0 : (rl == o.lr() ? 0.5 * v : v);
            acc[r] = (real)v;
          }
          st[s] = acc;
        }
      }
      SYNC();
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int p = wave == 0 ? NTT : (wave - 1) + s * (NW - 1);
        if (p < NTT) st_tile(Lb + p * TS, o, st[s]);                        // Phi -> Lb (over G)
      }
      SYNC();
      STAMP(7);
      STAMP(8);
      // ---- W = Phi L^-1 (lower; L^-1 in Wb since the factorisation) -> Lb
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int p = wave == 0 ? NTT : (wave - 1) + s * (NW - 1);
        st[s] = v4r{0, 0, 0, 0};
        if (p < NTT) {
          int i, j;
          unpack_lower(p, i, j);
          st[s] = acc_run(st[s], j, i + 1, [&](int k, int q) { return Lb[lt(i, k) * TS + o.rc[q]]; },
                          [&](int k, int q) { return Wb[lt(k, j) * TS + o.cr[q]]; });
        }
      }
      SYNC();
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int p = wave == 0 ? NTT : (wave - 1) + s * (NW - 1);
        if (p < NTT) st_tile(Lb + p * TS, o, st[s]);
      }
      SYNC();
      }   // !wb
      STAMP(9);
      // ---- Kbar = L^-T W (L^-1 in Wb, W in Lb) ; g = <Kbar, dK/dtheta> over the real part ; Adam   (models.py:210).  dK is symmetric, so
      // the tiles (i, j) and (j, i) of Kbar meet the same kernel values: Kbar_ij and Kbar_ji^T = sum_k W_ki^T Linv_kj
      // are accumulated together and the kernel function is evaluated once per unordered pair.
      {
        const double var = misc[8], ls = misc[9];
        double gv = 0.0, gl = 0.0;
#pragma unroll 1
        for (int p = wave; p < NTT; p += NW) {
          int i, j;
          unpack_lower(p, i, j);
          // this wave parked the tile's kernel values itself (same tile-to-wave map as kmat): requested ahead of the products
          double k1t[4], k2t[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            k1t[r] = kpark[(size_t)(2 * p) * TS + r * 64 + lane];
            k2t[r] = kpark[(size_t)(2 * p + 1) * TS + r * 64 + lane];
          }
          v4r acc = {0, 0, 0, 0};
          if (wb) {                    // the symmetric weights: (P - gamma p p^T) / 2 for the tile and its mirror image together
            acc = ld_tile(Cb + p * TS, o);
            const double f = i == j ? 0.5 : 1.0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              acc[r] = (real)(f * ((double)acc[r] - g.gamma * (double)uv[16 * i + o.kq(r)] * (double)uv[16 * j + o.lr()]));
          } else {
            acc = acc_run(acc, i, NT, [&](int k, int q) { return Wb[lt(k, i) * TS + o.cr[q]]; },
                          [&](int k, int q) { return Lb[lt(k, j) * TS + o.cr[q]]; });
            if (i != j)
              acc = acc_run(acc, i, NT, [&](int k, int q) { return Lb[lt(k, i) * TS + o.cr[q]]; },
                            [&](int k, int q) { return Wb[lt(k, j) * TS + o.cr[q]]; });
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) { gv = fma((double)acc[r], k1t[r], gv); gl = fma((double)acc[r], k2t[r], gl); }
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
          gg[0] = sv * (1.0 - exp(-var));                               // d/d variance = <Kbar, K0> / variance = <Kbar, k1>
          gg[1] = sl * (kern_cl(g.kernel) * var / ls) * (1.0 - exp(-ls));  // softplus'(u) = 1 - exp(-softplus(u))
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
      STAMP(10);
    }

    // ---- prediction at the final hypers with the (m, S) of the last step    (models.py:217-220)
    STAMP(11);
    refresh_hypers();
    kmat();
    bad = potrf_lds<NT, true>(Y::LB, Y::DL, Y::WB, o, wave, lane, bad, NoExtra(), w4idle);       // Lb = L, Wb = L^-1 (riding along), Cb = S
    SYNC();
    if (g.loss && g.n_iters > 0 && (g.n_iters - 1) % 25 == 0) neg_elbo((g.n_iters - 1) / 25);
    STAMP(12);
    STAMP(13);
    // A^T = K0 L^-T = L - J L^-T enters every product below as an MFMA operand.  Forming its elements on the fly (a read of L, a
    // read of L^-1, an fma - rounds 2-3) put fp64 VECTOR instructions into the operand path, and those share the matrix pipe: each
    // waits out the MFMAs of the SIMD's other wave (tools/diag16_bench.py: 7.5 cycles per dependent fma alone, 71 beside MFMAs) -
    // the products ran at a fifth of the pipe rate.  Round 4: L^-1 is scaled by -J in place once (Wb' = -J L^-1: nothing after
    // this needs the unscaled inverse: A^T = L + Wb'^T, J^2 L^-T L^-1 = Wb'^T Wb'), the six diagonal tiles of A^T are formed
    // explicitly in DLi, and every operand is a plain LDS read again.
#pragma unroll 1
    for (int p = wave; p < NTT; p += NW) {
      v4r t = ld_tile(Wb + p * TS, o);
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] = (real)(-J * (double)t[r]);
      st_tile(Wb + p * TS, o, t);
    }
    SYNC();
    for (int i = wave; i < NT; i += NW) {              // D_i = L_ii + (Wb'_ii)^T (L_ii is zero above its diagonal, L^-1_ii too)
      v4r t = ld_tile(Lb + lt(i, i) * TS, o);
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] = t[r] + Wb[lt(i, i) * TS + o.rc[r]];       // element (kq[r], lr) of the transpose
      st_tile(DLi + i * TS, o, t);
    }
    // mu = A^T m = L m + Wb'^T m
#pragma unroll 1
    for (int i = wave; i < NT; i += NW) {
      const double s1 = mv_lower<NT>(Lb, mv, i, o);
      const double s2 = mv_lower_t<NT>(Wb, mv, i, o);
      if (o.lk() == 0) av[16 * i + o.lr()] = (vecr)(s1 + s2);                   // av = mu (padded part 0)
    }
    SYNC();                                                                 // (D tiles; av is read after the barriers below)
    STAMP(14);
    // cov = diag(nu + J) - J^2 L^-T L^-1 + A^T S A as ONE product phase (round 4; rounds 2-3 staged E = A^T S row by row through
    // LDS: six phases with at most i + 1 of the eight waves on a row's tiles - 83 k of the prediction's 192 k cycles).
    // A^T S A = At F with F = S At^T (At = A^T = L - J L^-T).  A unit of work is half a block column of F: wave (j, h) forms
    // F(l, j) for l in its half of the contraction range in REGISTERS (an accumulator tile is a B operand as it stands), then its
    // partial sums  cov(i, j) += sum_{l in half} At(i, l) F(l, j),  i = j .. NT-1,  with B from registers; the L^-T L^-1 terms are
    // dealt over the two halves by parity (with Wb' = -J L^-1 they are plain Gram products).  The h = 1 partials meet the h = 0 ones through Cb (S is dead by then).  NT = 6:
    // twelve units on eight waves - columns 0, 1 alone, (2, 5) and (3, 4) paired: 46 ... 57 tile steps per wave.
    // A-operand element [lr][kq[q]] of A^T(i, k): L_ik (k < i), D_i (k = i), Wb'_ki^T (k > i) - which tile is a scalar choice
    auto at_a = [&](int i, int k, int q) -> real {
      const int base = k < i ? Y::LB + lt(i, k) * TS : (k == i ? Y::DL + i * TS : Y::WB + lt(k, i) * TS);
      return tiles_base()[base + (k <= i ? o.rc[q] : o.cr[q])];
    };
    constexpr int HL = NT / 2;                       // NT is 2, 4 or 6
    constexpr int T1 = NT == 6 ? 2 : 1;              // tiles of a wave's second unit (NT = 6 only: columns 4 and 5)
    const int ch = wave & 1, cj0 = wave >> 1, cj1 = NT + 1 - cj0;
    const bool on0 = cj0 < NT, on1 = NT == 6 && wave >= 4;
    v4r pc0[NT], pc1[T1];
    auto cov_unit = [&](auto& pc, auto maxt, int j, bool on) {
      constexpr int MAXT = decltype(maxt)::value;
      const int l0 = ch * HL;
      v4r f[HL];
#pragma unroll
      for (int ll = 0; ll < HL; ++ll) {
        const int l = l0 + ll;
        f[ll] = v4r{0, 0, 0, 0};
        if (on)
          f[ll] = acc_run(f[ll], 0, NT,
                          [&](int k, int q) { return Cb[k <= l ? lt(l, k) * TS + o.rc[q] : lt(k, l) * TS + o.cr[q]]; },
                          [&](int k, int q) { return at_a(j, k, q); });
      }
      if (MAXT == NT) WSTAMP(1);
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const int i = j + t;
        pc[t] = v4r{0, 0, 0, 0};
        if (on && i < NT) {
          v4r acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll
          for (int ll = 0; ll < HL; ++ll) {
            real a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = at_a(i, l0 + ll, q);
            acc = MFMAR(a[0], f[ll][0], acc);
            acc2 = MFMAR(a[1], f[ll][1], acc2);
            acc = MFMAR(a[2], f[ll][2], acc);
            acc2 = MFMAR(a[3], f[ll][3], acc2);
          }
          if (((i + j) & 1) == ch) {                 // - J^2 (L^-T L^-1)_ij = - sum_{k >= i} Wb'_ki^T Wb'_kj
            v4r qa = {0, 0, 0, 0};
            qa = acc_run(qa, i, NT, [&](int k, int q) { return Wb[lt(k, i) * TS + o.cr[q]]; },
                         [&](int k, int q) { return Wb[lt(k, j) * TS + o.cr[q]]; });
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = acc[r] - qa[r];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) pc[t][r] = acc[r] + acc2[r];
        }
      }
    };
    WSTAMP(0);
    cov_unit(pc0, std::integral_constant<int, NT>(), cj0, on0);
    WSTAMP(2);
    cov_unit(pc1, std::integral_constant<int, T1>(), on1 ? cj1 : 0, on1);
    WSTAMP(3);
    SYNC();                                          // every product has read S (Cb), L (Lb) and L^-1 (Wb)
    WSTAMP(4);
    // tile (i, j) is completed by the half with h = (i + j) & 1 (the one that added its L^-T L^-1 term); the other half hands
    // its partial sum over through Cb
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (on0 && cj0 + t < NT && ((t & 1) != ch)) st_tile(Cb + lt(cj0 + t, cj0) * TS, o, pc0[t]);
#pragma unroll
    for (int t = 0; t < T1; ++t)
      if (on1 && cj1 + t < NT && ((t & 1) != ch)) st_tile(Cb + lt(cj1 + t, cj1) * TS, o, pc1[t]);
    WSTAMP(6);
    SYNC();
    WSTAMP(7);
    // completion: export, and the bordered matrix of chol(cov) into Lb: rows N
    // (1^T) and N+1 (mu^T) in the padding, huge pivots there, identity elsewhere
    auto cov_finish = [&](v4r acc, int i, int j) {
      // (branch-free up to the exports: every operand - the other half's partial, mu for the bordered row, nu for the diagonal -
      //  is requested up front; LDS reads under per-element conditions cost a round trip each, 2.3 k cycles per tile)
      const v4r other = ld_tile(Cb + lt(i, j) * TS, o);
      const int col = 16 * j + o.lr();
      const double mu_c = (double)av[col];
      double nur[4] = {0.0, 0.0, 0.0, 0.0};
      if (i == j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * i + o.kq(r);
          nur[r] = (double)g.nu[(long)b * N + (row < N ? row : 0)];          // cov += diag(y_var)   (models.py:220)
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + o.kq(r);
        double v = (double)acc[r] + (double)other[r];
        if (row == col) v += nur[r] + J;
        const bool inside = row < N && col < N;
        double bv = row == col ? ((row == N || row == N + 1) ? 1e30 : 1.0) : 0.0;
        bv = (col < N && row == N) ? 1.0 : bv;
        bv = (col < N && row == N + 1) ? mu_c : bv;
        acc[r] = (real)(inside ? v : bv);
        if (inside) {
          // (a diagonal tile is a full 16x16 accumulation: only its lower half is exported here, the upper half is
          //  mirrored from it below, so that cov is exactly symmetric like the general path's extraction)
          if (g.cov && (i != j || col <= row)) {
            g.cov[((long)b * N + row) * N + col] = acc[r];
            if (i != j) g.cov[((long)b * N + col) * N + row] = acc[r];
          }
          if (g.var && row == col) g.var[(long)b * N + row] = acc[r];
        }
      }
      st_tile(Lb + lt(i, j) * TS, o, acc);
    };
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (on0 && cj0 + t < NT && ((t & 1) == ch)) cov_finish(pc0[t], cj0 + t, cj0);
#pragma unroll
    for (int t = 0; t < T1; ++t)
      if (on1 && cj1 + t < NT && ((t & 1) == ch)) cov_finish(pc1[t], cj1 + t, cj1);
    WSTAMP(5);
    if (g.mu)
      for (int t = tid; t < N; t += NTH) g.mu[(long)b * N + t] = (real)av[t];
    SYNC();
    STAMP(15);
    if (g.cov) {      // exact symmetry inside the diagonal tiles: upper element = lower element
      for (int e = tid; e < NT * 256; e += NTH) {
        const int i = e >> 8, rr = (e >> 4) & 15, cc = e & 15;
        const int row = 16 * i + rr, col = 16 * i + cc;
        if (cc > rr && row < N && col < N) g.cov[((long)b * N + row) * N + col] = Lb[lt(i, i) * TS + swz(cc, rr)];
      }
      SYNC();       // (the factorisation below overwrites the diagonal tiles)
    }
    if (g.chol || g.logdet || g.basis) {
      bad = potrf_lds<NT, false>(Y::LB, Y::DL, 0, o, wave, lane, bad, NoExtra(), w4idle);   // Distribution.__post_init__ (data.py:38-39)
      STAMP(16);
      if (g.chol) {
        for (int e = tid; e < NT * NT * 256; e += NTH) {
          const int tt = e >> 8, i = tt / NT, j = tt - i * NT, rr = (e >> 4) & 15, cc = e & 15;
          const int row = 16 * i + rr, col = 16 * j + cc;
          if (row < N && col < N) g.chol[((long)b * N + row) * N + col] = col <= row ? Lb[lt(i, j) * TS + swz(rr, cc)] : (real)0;
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
        for (int t = tid; t < N; t += NTH) s += log((double)Lb[lt(t >> 4, t >> 4) * TS + swz(t & 15, t & 15)]);
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
    STAMP(17);
    SYNC();                                                                 // LDS is re-initialised for the next member
  }
}

static int launch_grid(int B, int NT, int n_cus);
template <int NT>
static void launch(hipStream_t stream, const SmallArgs& g, int n_cus) {
  const size_t lds = Layout<NT>::BYTES;
  // the attribute belongs to the (function, device) pair: one flag per device, so that a second context on another GPU
  // of the node configures its own copy of the kernel
  static bool configured[64] = {};
  int dev = 0;
  BAYENS_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !configured[dev]) {
    BAYENS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(small_fit_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    if (dev >= 0 && dev < 64) configured[dev] = true;
  }
  const int grid = launch_grid(g.B, NT, n_cus);
  hipLaunchKernelGGL((small_fit_kernel<NT>), dim3(grid), dim3(NTH), lds, stream, g);
  BAYENS_HIP(hipGetLastError());
}

#ifndef BAYENS_F32
// ---- diagnostic: the 16x16 diagonal factorisation and the raw dependent-chain latencies it is made of, in isolation
// (one workgroup of `waves` waves; wave 0 runs the measured sequence `reps` times, the others idle at a barrier or, with
// variant bit 8, keep the matrix pipes busy).  out[0] = shader cycles per repetition.
__global__ void diag_bench_kernel(int variant, int reps, double* out, double seed) {
  __shared__ double T0[TS], T[TS], D[TS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Lane o;
  lane_init(o, lane);
  if (wave == 0)
    for (int e = lane; e < 256; e += 64) {
      const int r = e >> 4, c = e & 15;
      T0[swz(r, c)] = (r == c ? 20.0 : 0.0) + 1.0 / (1.0 + r + c) + seed;       // SPD
    }
  __syncthreads();
  const int kind = variant & 255;
  if (kind == 9 || kind == 10 || kind == 11) {   // every wave: throughput of independent fp64 vector work (9: 8 chains of v_fma_f64, 10: exp_neg, 11: sqrt_pos)
    double x[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = 1.0 + seed + 1e-9 * lane + 0.01 * c;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
      if (kind == 9) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int c = 0; c < 8; ++c) x[c] = fma(x[c], 0.999, 1e-3);
      } else if (kind == 10) {
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = exp_neg(-x[c]) + 0.5;
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = sqrt_pos(x[c]) + 0.5;
      }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    double sum = 0.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) sum += x[c];
    if (tid == 0) { out[0] = (double)(t1 - t0) / reps; out[1] = sum; }
    if (sum == 12345.678) out[2] = sum;
    return;
  }
  if (kind >= 12 && kind <= 16) {         // the factorisation's worker pattern: tile updates T_c -= T_a T_b^T on LDS tiles
    __shared__ double Tw[26 * TS];        // (variant & 512: wave 0 runs the serial 16x16 factorisation beside them and wave 4 idles)
    for (int e = tid; e < 26 * TS; e += blockDim.x) Tw[e] = 1.0 + 1e-6 * e;
    __syncthreads();
    const bool chain = (variant & 512) != 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (chain && wave == 0) {
      int bad = 0;
      for (int rep = 0; rep < reps / 4 + 1; ++rep) {
        for (int e = lane; e < 256; e += 64) T[e] = T0[e];
        diag16(ld_tile_sym64(T, o), T, D, lane, 0, bad, o);
      }
      if (bad == 12345) out[3] = 1.0;
    } else if (!(chain && wave == 4)) {
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      if (kind == 12) {
        for (int rep = 0; rep < reps; ++rep) {
          double* t = Tw + (wv + 8 * (rep & 1)) * TS;
          v4r acc = ld_tile(t, o);
          acc = mma<false, true>(acc, Tw + (16 + (rep + wv) % 5) * TS, Tw + (21 + (rep + 2 * wv) % 5) * TS, o, -1);
          st_tile(t, o, acc);
        }
      } else if (kind == 13) {
        for (int rep = 0; rep < reps; ++rep) {
          double* t = Tw + (wv + 8 * (rep & 1)) * TS;
          v4r acc = ld_tile(t, o);
          acc = mma1<false, true, true>(acc, Tw + (16 + (rep + wv) % 5) * TS, Tw + (21 + (rep + 2 * wv) % 5) * TS, o);
          st_tile(t, o, acc);
        }
      } else if (kind == 14) {            // two updates per trip: 24 reads in flight, the two accumulator chains alternate
        for (int rep = 0; rep < reps; rep += 2) {
          double* t0p = Tw + wv * TS;
          double* t1p = Tw + (wv + 8) * TS;
          const double* a0 = Tw + (16 + (rep + wv) % 5) * TS; const double* b0 = Tw + (21 + (rep + 2 * wv) % 5) * TS;
          const double* a1 = Tw + (16 + (rep + 1 + wv) % 5) * TS; const double* b1 = Tw + (21 + (rep + 1 + 2 * wv) % 5) * TS;
          v4r c0 = ld_tile(t0p, o), c1 = ld_tile(t1p, o);
          real x0[4], y0[4], x1[4], y1[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { x0[q] = a0[o.rc[q]]; y0[q] = b0[o.rc[q]]; x1[q] = a1[o.rc[q]]; y1[q] = b1[o.rc[q]]; }
          asm volatile("" : "+v"(x0[0]), "+v"(x0[1]), "+v"(x0[2]), "+v"(x0[3]), "+v"(y0[0]), "+v"(y0[1]), "+v"(y0[2]), "+v"(y0[3]));
          asm volatile("" : "+v"(x1[0]), "+v"(x1[1]), "+v"(x1[2]), "+v"(x1[3]), "+v"(y1[0]), "+v"(y1[1]), "+v"(y1[2]), "+v"(y1[3]));
#pragma unroll
          for (int q = 0; q < 4; ++q) { c0 = MFMAR(flip_sign(x0[q]), y0[q], c0); c1 = MFMAR(flip_sign(x1[q]), y1[q], c1); }
          st_tile(t0p, o, c0);
          st_tile(t1p, o, c1);
        }
      } else if (kind == 15) {            // one update per trip, the operands of the next one requested before this one's MFMAs
        double* t = Tw + wv * TS;
        v4r acc = ld_tile(t, o);
        real x[4], y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { x[q] = Tw[(16 + wv % 5) * TS + o.rc[q]]; y[q] = Tw[(21 + (2 * wv) % 5) * TS + o.rc[q]]; }
        for (int rep = 0; rep < reps; ++rep) {
          double* tn = Tw + (wv + 8 * ((rep + 1) & 1)) * TS;
          const double* an = Tw + (16 + (rep + 1 + wv) % 5) * TS; const double* bn = Tw + (21 + (rep + 1 + 2 * wv) % 5) * TS;
          v4r accn = ld_tile(tn, o);
          real xn[4], yn[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { xn[q] = an[o.rc[q]]; yn[q] = bn[o.rc[q]]; }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = MFMAR(flip_sign(x[q]), y[q], acc);
          __builtin_amdgcn_sched_barrier(0);
          st_tile(t, o, acc);
          t = tn; acc = accn;
#pragma unroll
          for (int q = 0; q < 4; ++q) { x[q] = xn[q]; y[q] = yn[q]; }
        }
        if (acc[0] == 12345.678) out[2] = acc[1];
      } else {                            // 16: two chains per update as shipped, signs flipped in the integer pipe, reads pinned
        for (int rep = 0; rep < reps; ++rep) {
          double* t = Tw + (wv + 8 * (rep & 1)) * TS;
          const double* a0 = Tw + (16 + (rep + wv) % 5) * TS; const double* b0 = Tw + (21 + (rep + 2 * wv) % 5) * TS;
          v4r acc = ld_tile(t, o);
          real x[4], y[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { x[q] = a0[o.rc[q]]; y[q] = b0[o.rc[q]]; }
          asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
          v4r acc2 = {0, 0, 0, 0};
          acc = MFMAR(flip_sign(x[0]), y[0], acc);
          acc2 = MFMAR(flip_sign(x[2]), y[2], acc2);
          acc = MFMAR(flip_sign(x[1]), y[1], acc);
          acc2 = MFMAR(flip_sign(x[3]), y[3], acc2);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
          st_tile(t, o, acc);
        }
      }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (tid == 64) { out[0] = (double)(t1 - t0) / reps; out[1] = Tw[5]; }
    return;
  }
  if (kind == 7 || kind == 8) {           // every wave: `reps` tile products of 6 steps each from LDS tiles (the GS pattern)
    __shared__ double Tl[8 * TS];
    for (int e = tid; e < 8 * TS; e += blockDim.x) Tl[e] = 1.0 + 1e-6 * e;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (int rep = 0; rep < reps; ++rep) {
      if (kind == 7)
        acc = acc_run(acc, 0, 6, [&](int k, int q) { return Tl[((k + wave) & 7) * TS + o.rc[q]]; },
                      [&](int k, int q) { return Tl[((k + rep) & 7) * TS + o.cr[q]]; });
      else {                                // registers only: the pipe-bound reference
        const double a = 1.0 + 1e-9 * lane, b = 0.5;
#pragma unroll
        for (int q = 0; q < 24; ++q) acc = MFMA(a, b, acc);
      }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) { out[0] = (double)(t1 - t0) / reps; out[1] = acc[0]; }
    if (acc[0] == 12345.678) out[2] = acc[1];
    return;
  }
  if (wave == 0) {
    int bad = 0;
    double x = 1.0 + seed + lane * 1e-9, y = 0.999;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
      if (kind == 0) {                     // the factorisation as the fit uses it (tile restored every time)
        for (int e = lane; e < 256; e += 64) T[e] = T0[e];
        diag16(ld_tile_sym64(T, o), T, D, lane, 0, bad, o);
      } else if (kind == 1) {              // 64 dependent v_fma_f64
#pragma unroll
        for (int i = 0; i < 64; ++i) x = fma(x, y, 1e-3);
      } else if (kind == 2) {              // 16 dependent v_rsq_f64 (+ one fma each to keep them dependent and bounded)
#pragma unroll
        for (int i = 0; i < 16; ++i) x = fma(__builtin_amdgcn_rsq(x), 0.5, 1.0);
      } else if (kind == 3) {              // 16 dependent refined rsqrt
#pragma unroll
        for (int i = 0; i < 16; ++i) x = rsqrt_refined(x) + 1.0;
      } else if (kind == 4) {              // 16 dependent MFMAs (one accumulator)
        v4d acc = {x, x, x, x};
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = MFMA(y, 1e-3, acc);
        x = acc[0];
      } else if (kind == 5) {              // 16 x (readlane of an MFMA result -> operand of the next MFMA)
        v4d acc = {x, x, x, x};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const double u = readlane_f64(acc[0], 5);
          acc = MFMA(u * 1e-3, y, acc);
        }
        x = acc[0];
      } else if (kind == 6) {              // tile load + store through LDS (what brackets the factorisation)
        v4d a = ld_tile(T0, o);
        a[0] += x;
        st_tile(T, o, a);
        x = T[o.rc[1]];
      }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[0] = (double)(t1 - t0) / reps; out[1] = x + bad + T[3] + D[5]; }
  } else if (variant & 256) {            // partner waves hammer the matrix pipes meanwhile
    v4d acc = {1.0, 1.0, 1.0, 1.0};
    for (int i = 0; i < reps * 64; ++i) acc = MFMA(1.0 + 1e-9 * lane, 0.5, acc);
    if (acc[0] == 12345.0) out[2] = acc[0];
  }
}

void diag_bench(hipStream_t stream, int variant, int waves, int reps, double* out_dev) {
  hipLaunchKernelGGL(diag_bench_kernel, dim3(1), dim3(64 * waves), 0, stream, variant, reps, out_dev, 0.0);
  BAYENS_HIP(hipGetLastError());
}
#endif

bool small_fit_supported(int N, int R) { return N >= 1 && N <= SMALL_MAX_N && R >= 1; }

static int tiles_per_side(int N) { return N + 2 <= 32 ? 2 : (N + 2 <= 64 ? 4 : 6); }
static size_t lds_bytes(int NT) { return NT == 2 ? Layout<2>::BYTES : (NT == 4 ? Layout<4>::BYTES : Layout<6>::BYTES); }
// workgroups of a launch: as many as are resident at once (LDS-bound: 5 / 2 / 1 per compute unit at 2 / 4 / 6 tiles per side)
static int launch_grid(int B, int NT, int n_cus) {
  const int per_cu = (int)(163840 / lds_bytes(NT)) > 0 ? (int)(163840 / lds_bytes(NT)) : 1;
  const int grid = n_cus * per_cu;
  return grid < B ? grid : B;
}

size_t small_park_doubles(int B, int N, int n_cus) {
  const int NT = tiles_per_side(N);
  const int grid = launch_grid(B > 0 ? B : 1, NT, n_cus);
  return (size_t)grid * (2 * (NT * (NT + 1) / 2) * TS);
}

void small_fit(hipStream_t stream, const SmallArgs& g_in, int n_cus) {
  if (!small_fit_supported(g_in.N, g_in.R)) throw HipError{-1, "small_fit: N out of range"};
  if (g_in.B <= 0) return;
  if (g_in.n_iters > 0 && !g_in.kpark) throw HipError{-1, "small_fit: no kernel-value park (SmallArgs::kpark)"};
  static const int tune = [] { const char* e = std::getenv("BAYENS_SMALL_TUNE"); return e ? std::atoi(e) : 0; }();
  SmallArgs g = g_in;
  g.tune = tune;
  const int NT = tiles_per_side(g.N);
  if (NT == 2) launch<2>(stream, g, n_cus);
  else if (NT == 4) launch<4>(stream, g, n_cus);
  else launch<6>(stream, g, n_cus);
}

#endif  // BAYENS_SMALL_HELPERS_ONLY

}  // namespace smalln
}  // namespace BAYENS_NS
