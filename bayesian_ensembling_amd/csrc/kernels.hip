// kernels.hip - every device kernel of the hot path except the batched GEMM (gemm.hip).
//
// Reference arithmetic (file:line into /root/reference/ensembles) is cited at each kernel; formulas are
// written out in SURVEY.md Appendix A.  These kernels are O(N^2) per member and HBM/latency-bound: they
// are laid out for coalesced 512-byte wave accesses along the contiguous (column / time) axis.
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include <type_traits>

#include "hip_prims.h"

namespace BAYENS_NS {

typedef double v4d __attribute__((ext_vector_type(4)));

namespace {

constexpr double SQRT3 = 1.7320508075688772;
constexpr double SQRT5 = 2.23606797749979;
constexpr double LOG2PI = 1.8378770664093453;

__device__ __forceinline__ double softplus_d(double u) { return u > 30.0 ? u : log1p(exp(u)); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;   // valid in lane 0
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;   // valid in lane 0
}

// exp(x) for x <= 0 and sqrt(d) for d >= 1e-36 as the fused kernels evaluate them (csrc/small.hip: Cody-Waite reduction +
// degree-13 Taylor polynomial + v_ldexp_f64; v_rsq_f64 seed + two Goldschmidt corrections): the kernel-matrix passes are bound by
// the fp64 vector rate, and the library's exp / sqrt cost half as much again for range and special-case handling these arguments
// never need (round 4)
// A double constant held in a scalar register pair.  Left to itself the compiler turns every coefficient of a Horner chain into
// two v_mov_b32 in front of a v_fmac_f64 (the accumulator form wants the addend in the destination register): 26 extra vector
// instructions per exponential, in kernels that are bound by the vector issue rate (VALUBusy 50 % = the fp64 rate, round 5).  An
// opaque s_mov pair (scalar unit, beside the vector work; identical asm statements are merged) makes the addend the one scalar
// operand a v_fma_f64 may take.
#define BAYENS_SCONST(name, hi, lo)                                \
  int name##_h, name##_l;                                          \
  asm("s_mov_b32 %0, " #lo : "=s"(name##_l));                      \
  asm("s_mov_b32 %0, " #hi : "=s"(name##_h));                      \
  const double name = __hiloint2double(name##_h, name##_l)
// p t + c with c in scalar registers, as the three-operand v_fma_f64 (the compiler prefers the accumulator form and copies the
// scalar pair into vector registers for it)
__device__ __forceinline__ double fma_sc(double p, double t, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(t), "s"(c));
  return r;
}
__device__ __forceinline__ double exp_neg(double x) {
  x = x < -745.0 ? -745.0 : x;
  const double n = rint(x * 1.4426950408889634);
  double t = fma(n, -0.6931471803691238, x);
  t = fma(n, -1.9082149292705877e-10, t);
  const double c0v = 1.6059043836821613e-10;
  BAYENS_SCONST(c1, 0x3e21eed8, 0xeff8d898);   // 2.08767569878681e-09
  BAYENS_SCONST(c2, 0x3e5ae645, 0x67f544e4);   // 2.505210838544172e-08
  BAYENS_SCONST(c3, 0x3e927e4f, 0xb7789f5c);   // 2.755731922398589e-07
  BAYENS_SCONST(c4, 0x3ec71de3, 0xa556c734);   // 2.7557319223985893e-06
  BAYENS_SCONST(c5, 0x3efa01a0, 0x1a01a01a);   // 2.48015873015873e-05
  BAYENS_SCONST(c6, 0x3f2a01a0, 0x1a01a01a);   // 0.0001984126984126984
  BAYENS_SCONST(c7, 0x3f56c16c, 0x16c16c17);   // 0.001388888888888889
  BAYENS_SCONST(c8, 0x3f811111, 0x11111111);   // 0.008333333333333333
  BAYENS_SCONST(c9, 0x3fa55555, 0x55555555);   // 0.041666666666666664
  BAYENS_SCONST(c10, 0x3fc55555, 0x55555555);   // 0.16666666666666666
  double p = fma_sc(c0v, t, c1);
  p = fma_sc(p, t, c2);
  p = fma_sc(p, t, c3);
  p = fma_sc(p, t, c4);
  p = fma_sc(p, t, c5);
  p = fma_sc(p, t, c6);
  p = fma_sc(p, t, c7);
  p = fma_sc(p, t, c8);
  p = fma_sc(p, t, c9);
  p = fma_sc(p, t, c10);
  p = fma(p, t, 0.5);
  p = fma(p, t, 1.0);
  p = fma(p, t, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
}
__device__ __forceinline__ double sqrt_pos(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  double r = d * y;
  const double h = 0.5 * y;
  r = fma(fma(-r, r, d), h, r);
  r = fma(fma(-r, r, d), h, r);
  return r;
}
// K(r2) and dK/dlengthscale for the three stationary kernels (GPflow 2.1.5 forms; Matern-3/2 is the only
// one the reference instantiates, models.py:186).
// (KERNEL is a template parameter of the passes that evaluate it: with a run-time switch inside the element loop every element
//  was a chain of its own behind a branch - sqrt, exp, product, store, one after the other - and the independent elements of a
//  thread never overlapped)
template <int KERNEL>
__device__ __forceinline__ void kern_eval(double var, double ls, double r2, double& K, double& dK) {
  constexpr int kernel = KERNEL;
  if (kernel == 2) {
    double r2c = r2 > 0.0 ? r2 : 0.0;
    K = var * exp_neg(-0.5 * r2c);
    dK = K * r2c / ls;
    return;
  }
  const bool clamp = r2 <= 1e-36;
  const double r = sqrt_pos(r2 > 1e-36 ? r2 : 1e-36);
  if (kernel == 0) {
    double e = exp_neg(-SQRT3 * r);
    K = var * (1.0 + SQRT3 * r) * e;
    dK = 3.0 * var * r * r * e / ls;
  } else {
    double e = exp_neg(-SQRT5 * r);
    K = var * (1.0 + SQRT5 * r + (5.0 / 3.0) * r * r) * e;
    dK = (5.0 / 3.0) * var * r * r * (1.0 + SQRT5 * r) * e / ls;
  }
  if (clamp) dK = 0.0;
}

// ------------------------------------------------------------------------------------------------
// elementwise / layout kernels (2-D grid of 64x64 tiles, batch in z)
// ------------------------------------------------------------------------------------------------
__global__ void copy_kernel(int n, Mat src, Mat dst, int mode) {
  // mode 0 full copy, 1 tril copy (zeros above)
  const int b = blockIdx.z;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6);
  const real* s = src.p + b * src.stride;
  real* d = dst.p + b * dst.stride;
  // (loads first, stores after: src and dst may alias as far as the compiler knows, and a load behind a store
  //  waits for it - one in-order memory counter)
  real v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + 4 * q;
    v[q] = s[(long)i * src.ld + j];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) d[(long)(i0 + 4 * q) * dst.ld + j] = (mode == 1 && j > i0 + 4 * q) ? (real)0 : v[q];
}

__global__ void transpose_lower_kernel(int n, Mat W, Mat Wt) {
  __shared__ double tile[64][65];
  const int b = blockIdx.z;
  const int bi = blockIdx.y, bj = blockIdx.x;   // output tile (bi, bj) = input tile (bj, bi)^T
  const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
  const real* w = W.p + b * W.stride;
  real* wt = Wt.p + b * Wt.stride;
  if (bj < bi) {   // strictly below the diagonal of the output: zeros
#pragma unroll
    for (int q = 0; q < 16; ++q) wt[(long)(bi * 64 + r0 + 4 * q) * Wt.ld + bj * 64 + c] = 0.0;
    return;
  }
  // (all sixteen loads of a thread requested before the first is placed: one memory round trip per tile)
  real v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int gi = bj * 64 + r0 + 4 * q, gj = bi * 64 + c;   // input element (gi, gj), lower part only
    v[q] = w[(long)gi * W.ld + gj];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = r0 + 4 * q;
    tile[r][c] = (bi * 64 + c <= bj * 64 + r) ? (double)v[q] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = r0 + 4 * q;
    wt[(long)(bi * 64 + r) * Wt.ld + bj * 64 + c] = tile[c][r];
  }
}

__global__ void fill_identity_kernel(int n, Mat A) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6);
  real* a = A.p + b * A.stride;
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + 4 * q;
    a[(long)i * A.ld + j] = i == j ? 1.0 : 0.0;
  }
}

__global__ void init_vec_kernel(int nb, int N, int Np, real* theta1, real* mvec, double* ustate,
                                const real* y_in, const real* nu_in, real* y_pad, real* nu_pad,
                                int* info) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Np) {
    theta1[(long)b * Np + i] = 0.0;
    mvec[(long)b * Np + i] = 0.0;
    y_pad[(long)b * Np + i] = i < N ? y_in[(long)b * N + i] : 0.0;
    nu_pad[(long)b * Np + i] = i < N ? nu_in[(long)b * N + i] : 1.0;
  }
  if (i < 8) ustate[b * 8 + i] = i < 2 ? 0.5413248546129181 /* log(e-1): softplus -> 1.0 */ : 0.0;
  if (i == 0) info[b] = 0;
}

__global__ void scale_rows_kernel(int n, Mat src, const real* nu, Mat dst) {
  // src is lower triangular and dst only feeds the k-trimmed lower x lower product (G = W^T W), which reads the 64-blocks
  // on or below the diagonal and, inside its 128 x 128 diagonal tiles, the one above it: the rest is neither read nor
  // written here
  if ((blockIdx.x >> 1) > (blockIdx.y >> 1)) return;
  const int b = blockIdx.z;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6);
  const real* s = src.p + b * src.stride;
  real* d = dst.p + b * dst.stride;
  real v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = s[(long)(i0 + 4 * q) * src.ld + j];
  // the sixteen row factors of this wave's rows: lane q < 16 forms 1 / sqrt(nu) of row i0 + 4 q once, the others read it from
  // there (every lane forming all sixteen was 16 square roots and divisions per thread)
  const int lane = threadIdx.x & 63;
  const double fac = 1.0 / sqrt((double)nu[(long)b * n + i0 + 4 * (lane & 15)]);
#pragma unroll
  for (int q = 0; q < 16; ++q) d[(long)(i0 + 4 * q) * dst.ld + j] = v[q] * __shfl(fac, q, 64);
}

// Phi_ij = u_i m_j + GS_ij below the diagonal, half of it on the diagonal, 0 above (prims.h form_phi)
__global__ void form_phi_kernel(int n, Mat GS, const real* u, const real* mvec, Mat Phi, double scale, double diag_add,
                                double u_scale) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6);
  const real* g = GS.p + b * GS.stride;
  real* p = Phi.p + b * Phi.stride;
  if (blockIdx.x > blockIdx.y) {                 // a tile above the diagonal: zeros, nothing read
#pragma unroll
    for (int q = 0; q < 16; ++q) p[(long)(i0 + 4 * q) * Phi.ld + j] = 0.0;
    return;
  }
  // (every load of the thread requested before the first value is formed - a load under `if (j <= i)` is a branch whose value is
  //  awaited at its join, sixteen round trips one behind the other; the elements above the diagonal of a diagonal tile are valid memory)
  const double mj = mvec[(long)b * n + j];
  real gv[16], uv[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    gv[q] = g[(long)(i0 + 4 * q) * GS.ld + j];
    uv[q] = u[(long)b * n + i0 + 4 * q];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + 4 * q;
    double v = u_scale * (double)uv[q] * mj + (scale * (double)gv[q] + (j == i ? diag_add : 0.0));
    if (j == i) v *= 0.5;
    p[(long)i * Phi.ld + j] = j <= i ? v : 0.0;
  }
}

// natural-gradient precision update (prims.h lam_update): one workgroup per lower 64x64 block (bi >= bj) updates
// Lam, seeds Cb (lower; the mirrored block of Cb is cleared) and mirrors G into the full symmetric Gfull.
__global__ __launch_bounds__(256) void lam_update_kernel(int n, Mat G, double gamma, Mat Lam, Mat Cb, Mat Gfull, int parts) {
  __shared__ real tile[64][65];
  const int b = blockIdx.z, bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
  const real* g = G.p + b * G.stride;
  real* lam = Lam.p + b * Lam.stride;
  real* cb = Cb.p + b * Cb.stride;
  real* gf = Gfull.p + b * Gfull.stride;
  const int j = bj * 64 + c;
  // all loads of the tile first, then all stores: the matrices may alias as far as the compiler knows, so a load
  // written after a store stays behind it - and then waits for it (one in-order memory counter)
  real gv[16], lv[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = bi * 64 + r0 + 4 * q;
    gv[q] = g[(long)i * G.ld + j];
    lv[q] = (parts & 1) ? lam[(long)i * Lam.ld + j] : (real)0;       // (the whole tile: valid memory, masked below by j <= i)
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = r0 + 4 * q;
    const int i = bi * 64 + r;
    tile[r][c] = gv[q];
    if (parts & 2) gf[(long)i * Gfull.ld + j] = gv[q];    // the diagonal block of G is complete (both triangles)
    if (!(parts & 1)) continue;
    if (j <= i) {
      const double v = (1.0 - gamma) * lv[q] + gamma * (gv[q] + (i == j ? 1.0 : 0.0));
      lam[(long)i * Lam.ld + j] = v;
      cb[(long)i * Cb.ld + j] = v;
    } else {
      cb[(long)i * Cb.ld + j] = 0.0;                      // strict upper part inside a diagonal block
    }
  }
  if (bi == bj || !(parts & 2)) return;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {                           // mirrored block (bj, bi): G^T into Gfull
    const int r = r0 + 4 * q;
    gf[(long)(bj * 64 + r) * Gfull.ld + bi * 64 + c] = tile[c][r];
  }
  // (Cb's blocks above the block diagonal are left alone: the factorisation and the solves with C never read them)
}

// At = L - jitter * Linv^T (prims.h at_from_linv): output tile (bi, bj) needs L's tile (bi, bj) (lower part) and
// the transpose of Linv's tile (bj, bi) (lower part), which goes through LDS
__global__ __launch_bounds__(256) void at_from_linv_kernel(int n, Mat L, Mat Linv, double jitter, Mat At) {
  __shared__ double tile[64][65];
  const int b = blockIdx.z, bi = blockIdx.y, bj = blockIdx.x;
  const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
  const real* l = L.p + b * L.stride;
  const real* li = Linv.p + b * Linv.stride;
  real* at = At.p + b * At.stride;
  // (whole tiles are requested without conditions - both matrices are padded, every element is valid memory - and masked
  //  afterwards: a load under a condition is a branch, and the sixteen of a thread then wait for one another)
  real tv[16], lv[16];
  if (bj >= bi) {                    // upper (or diagonal) block: Linv tile (bj, bi) exists
#pragma unroll
    for (int q = 0; q < 16; ++q) tv[q] = li[(long)(bj * 64 + r0 + 4 * q) * Linv.ld + bi * 64 + c];
  }
  if (bj <= bi) {                    // lower (or diagonal) block of L
#pragma unroll
    for (int q = 0; q < 16; ++q) lv[q] = l[(long)(bi * 64 + r0 + 4 * q) * L.ld + bj * 64 + c];
  }
  if (bj >= bi) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = r0 + 4 * q;
      tile[r][c] = (bi * 64 + c <= bj * 64 + r) ? (double)tv[q] : 0.0;
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = r0 + 4 * q;
    const int i = bi * 64 + r, j = bj * 64 + c;
    const double lval = (bj <= bi && j <= i) ? (double)lv[q] : 0.0;
    const double lit = bj >= bi ? tile[c][r] : 0.0;        // (Linv^T)_ij = Linv_ji
    at[(long)i * At.ld + j] = lval - jitter * lit;
  }
}

__global__ void diag_init_kernel(int n, const real* nu, double add, Mat dst) {
  if (blockIdx.x > blockIdx.y) return;     // the accumulation that follows, and everything after it, is lower-only
  const int b = blockIdx.z;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6);
  real* d = dst.p + b * dst.stride;
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + 4 * q;
    d[(long)i * dst.ld + j] = i == j ? nu[(long)b * n + i] + add : 0.0;
  }
}

__global__ void extract_kernel(int N, Mat src, real* dst, int mode) {
  __shared__ real tile[64][65];
  const int b = blockIdx.z;
  const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  const int i0 = blockIdx.y * 64 + r0;
  const real* s = src.p + b * src.stride;
  real* d = dst + (long)b * N * N;
  real v[16];
  // mode 1 (the symmetric image of a lower triangle), output tile on or above the diagonal: its values are the TRANSPOSE of the
  // source tile (bx, by), which is read along its rows and turned in LDS (read in place - lane j from row j - every lane would
  // touch a cache line of its own).  src is a padded matrix: whole tiles are valid memory.
  const bool turn = mode == 1 && blockIdx.x >= blockIdx.y;
  if (turn) {
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = s[(long)(blockIdx.x * 64 + r0 + 4 * q) * src.ld + blockIdx.y * 64 + c];
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[r0 + 4 * q][c] = v[q];
    __syncthreads();
    if (j >= N) return;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = r0 + 4 * q, i = i0 + 4 * q;
      // diagonal tile: the element on or below the diagonal is tile[r][c] itself, the one above it the mirror tile[c][r]
      if (i < N) d[(long)i * N + j] = (blockIdx.x == blockIdx.y && j <= i) ? tile[r][c] : tile[c][r];
    }
    return;
  }
  if (j >= N) return;
  // (unconditional loads: the rows up to the padded order are valid memory; the mask is on the store)
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = s[(long)(i0 + 4 * q) * src.ld + j];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + 4 * q;
    if (i < N) d[(long)i * N + j] = (mode == 2 && j > i) ? (real)0 : v[q];      // (mode 1 below the diagonal tiles: j <= i everywhere)
  }
}

__global__ void pad_in_kernel(int N, int Np, const double* src, Mat dst) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6);
  const double* s = src + (long)b * N * N;
  real* d = dst.p + b * dst.stride;
  // (loads from clamped addresses, all sixteen requested before the first store; the identity of the padding is a select)
  const int jc = j < N ? j : N - 1;
  double v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + 4 * q;
    v[q] = s[(long)(i < N ? i : N - 1) * N + jc];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + 4 * q;
    d[(long)i * dst.ld + j] = (i < N && j < N) ? v[q] : (i == j ? 1.0 : 0.0);
  }
}

__global__ void fill_random_kernel(double* dst, size_t count, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    dst[i] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  }
}

__global__ void pad_vec_kernel(int N, int Np, const double* src, double* dst) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Np) dst[(long)b * Np + i] = i < N ? src[(long)b * N + i] : 0.0;
}

__global__ void extract_vec_kernel(int N, int Np, const real* src, real* dst) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) dst[(long)b * N + i] = src[(long)b * Np + i];
}

__global__ void extract_cols_kernel(int N, Mat src, real* dst, int c0, int c1) {
  const int b = blockIdx.z;
  const int j = c0 + blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6);
  const real* s = src.p + b * src.stride;
  real* d = dst + (long)b * N * N;
  if (j >= c1) return;
  real v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + 4 * q;
    v[q] = (i < N && j <= i) ? s[(long)i * src.ld + j] : (real)0;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + 4 * q;
    if (i < N) d[(long)i * N + j] = v[q];
  }
}

__global__ void extract_diagonal_kernel(int N, Mat src, real* dst) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) dst[(long)b * N + i] = src.p[b * src.stride + (long)i * src.ld + i];
}

__global__ void logdet_kernel(int N, Mat Lc, double* out) {
  __shared__ double part[4];
  const int b = blockIdx.x;
  const real* l = Lc.p + b * Lc.stride;
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) s += log(l[(long)i * Lc.ld + i]);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[b] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ void hyp_out_kernel(int nb, const double* ustate, double* hyp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) {
    hyp[b * 2] = softplus_d(ustate[b * 8]);
    hyp[b * 2 + 1] = softplus_d(ustate[b * 8 + 1]);
  }
}

__global__ void info_fold_kernel(int nb, int* pair, int* info) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const int p0 = pair[2 * b], p1 = pair[2 * b + 1];
  if (info[b] == 0) info[b] = p0 ? p0 : p1;
  pair[2 * b] = 0;
  pair[2 * b + 1] = 0;
}

__global__ void extract_diag_kernel(int N, const double* cov, double* var) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) var[(long)b * N + i] = cov[((long)b * N + i) * N + i];
}

// ------------------------------------------------------------------------------------------------
// kernel matrix (GPflow Matern32.K as used at models.py:186-189,198,217): r2 = -2 X~X~^T + s_i + s_j with
// the X~X~^T contraction on the fp64 matrix cores; the epilogue (sqrt/exp) runs on the 4 results per lane.
// ------------------------------------------------------------------------------------------------
__global__ void xprep_kernel(int N, int Np, int R_all, int R4, const real* X_all, const double* ustate,
                             double* xs, double* ssq, const int* Rv, const long* xoff) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Np) return;
  // a ragged batch (Rv): this member's own realisation count and block of X; the columns up to R4 are zeros either way, which
  // add nothing to a squared distance - the kernel matrix is bit for bit that of a uniform batch of Rv[b] realisations
  const int R = Rv ? Rv[b] : R_all;
  const real* X = X_all + (xoff ? xoff[b] : (long)b * N * R_all) - (long)b * N * R;
  const double ls = softplus_d(ustate[b * 8 + 1]);
  if (i == 0) {   // (variance, lengthscale) of this member behind the squared norms: the tile kernels read them instead of
    double* vl = ssq + (long)gridDim.y * Np + 2 * b;   // evaluating softplus in every thread of every workgroup
    vl[0] = softplus_d(ustate[b * 8]);
    vl[1] = ls;
  }
  double s = 0.0;
  for (int d = 0; d < R4; ++d) {
    double v = (i < N && d < R) ? X[((long)b * N + i) * R + d] / ls : 0.0;
    xs[((long)b * Np + i) * R4 + d] = v;
    s += v * v;
  }
  ssq[(long)b * Np + i] = s;
}

// one workgroup = one 64x64 output tile; wave w owns rows 16w..16w+15, four 16x16 MFMA tiles across.
// GRAD=false: write K0 and Lb.  GRAD=true: accumulate <Kbt, K0> and <Kbt, dK/dl> over the real part.
// (launch bounds: the second figure is waves per SIMD - with it the K pass compiles to 88 registers instead of 152 and the gradient
//  pass to 138 instead of 200, without spills: the scheduler interleaves the elements' chains less widely.  K pass 262 -> 243 us.)
template <bool GRAD, int KERNEL>
__global__ __launch_bounds__(256, GRAD ? 2 : 4) void kmat_kernel(int N, int Np, int R4, const double* xs, const double* ssq,
                                                   const double* ustate, double jitter, Mat Lb,
                                                   Mat Kbt, double* gpart, const real* pvec, double sym_gamma,
                                                   int clear_upper, Mat Mb, double m_scale, const real* m_diag) {
  const int b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int i0 = blockIdx.y * 64 + wave * 16, j0 = blockIdx.x * 64;
  const double* vl = ssq + (long)gridDim.z * Np + 2 * b;    // written by xprep_kernel
  const double var = vl[0], ls = vl[1];
  const double* xb = xs + (long)b * Np * R4;
  const double* sb = ssq + (long)b * Np;
  double gv = 0.0, gl = 0.0;
  if (!GRAD && blockIdx.x > blockIdx.y) {   // tiles above the diagonal of the lower image: zeros, no kernel evaluation
    if (!clear_upper) return;
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Lb.p[b * Lb.stride + (long)(i0 + lk + 4 * r) * Lb.ld + j0 + jt * 16 + lr] = 0.0;
    return;
  }
  // GRAD with pvec: the weights are the symmetric (P - sym_gamma p p^T) / 2, P's lower triangle in Kbt: tiles above the
  // diagonal contribute through their mirror images (pairs counted once, twice the weight)
  const bool sym = GRAD && pvec != nullptr;
  // GRAD without pvec (a general iteration: Kbt = Kbar^T, not symmetric): K and dK/dl ARE symmetric, so sum_ij Kbt_ij K_ij runs
  // over the lower tiles only with the weights Kbt_ij + Kbt_ji - the mirror tile (bj, bi) comes in through LDS, read transposed -
  // and the kernel is evaluated for half of the elements (diagonal tiles: every element, plain weights).
  const bool mirror = GRAD && !sym && blockIdx.x < blockIdx.y;
  const int njt = (GRAD && blockIdx.x > blockIdx.y) ? 0 : 4;
  __shared__ real tmir[GRAD ? 64 : 1][GRAD ? 65 : 1];
  // (the mirror tile is REQUESTED here and placed in LDS further down, behind the requests for everything else the workgroup
  //  reads: one memory round trip in front of the first element, not two)
  real mv[GRAD ? 16 : 1];
  if (mirror) {
    const real* src = Kbt.p + b * Kbt.stride + (long)j0 * Kbt.ld + blockIdx.y * 64 + (threadIdx.x & 63);
#pragma unroll
    for (int q = 0; q < 16; ++q) mv[q] = src[(long)((threadIdx.x >> 6) + 4 * q) * Kbt.ld];
  }
  // the weights of this thread's 16 elements are requested up front (one memory round trip; inside the guarded accumulation below
  // every load would be issued, and waited for, on its own); the padded rows / columns are valid memory and are masked by a select
  double kbv[4][4], pi[4] = {0, 0, 0, 0}, pj[4] = {0, 0, 0, 0};
  if (GRAD && njt) {
    const real* krow = Kbt.p + b * Kbt.stride + (long)(i0 + lk) * Kbt.ld + j0 + lr;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) kbv[jt][r] = krow[(long)(4 * r) * Kbt.ld + jt * 16];
    if (sym) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pi[q] = pvec[(long)b * Np + i0 + lk + 4 * q];
        pj[q] = pvec[(long)b * Np + j0 + q * 16 + lr];
      }
    }
  }
  // Tiles that meet neither the diagonal nor the padding (all but ~1/16 of them at N = 2000) need no masking: no select for the
  // padded rows / columns, the triangle or the jitter, and no per-element address arithmetic (row pointers + immediate offsets).
  const bool edge_tile = blockIdx.x == blockIdx.y || (int)blockIdx.y * 64 + 64 > N || (int)blockIdx.x * 64 + 64 > N;
  real* lrow[4];
  real* mrow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    lrow[r] = GRAD ? nullptr : Lb.p + b * Lb.stride + (long)(i0 + lk + 4 * r) * Lb.ld + j0 + lr;
    mrow[r] = (GRAD || !Mb.p) ? nullptr : Mb.p + b * Mb.stride + (long)(i0 + lk + 4 * r) * Mb.ld + j0 + lr;
  }
  // The scaled inputs' products X~ X~^T of all four strips first, k-outer: the five operand loads of a k step (and, unrolled, of
  // two) are in flight together.  (Strip by strip - two dependent loads, a wait and an MFMA per k step - a workgroup spent 8 serial
  // memory round trips here before it evaluated anything: 13-20 us of its life, the whole pass latency-bound.)
  v4d accs[4];
  double si[4], sjv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    accs[q] = v4d{0.0, 0.0, 0.0, 0.0};
    si[q] = sb[i0 + lk + 4 * q];
    sjv[q] = sb[j0 + q * 16 + lr];
  }
  if (njt) {
    const double* pa = xb + (long)(i0 + lr) * R4 + lk;
    const double* pb = xb + (long)(j0 + lr) * R4 + lk;
    const long strip = 16L * R4;
#pragma unroll 2
    for (int kk = 0; kk < R4; kk += 4) {
      const double a = pa[kk];
      double bb[4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) bb[jt] = pb[jt * strip + kk];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) accs[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb[jt], accs[jt], 0, 0, 0);
    }
  }
  if (mirror) {
#pragma unroll
    for (int q = 0; q < 16; ++q) tmir[(threadIdx.x >> 6) + 4 * q][threadIdx.x & 63] = mv[q];
    __syncthreads();
  }
  // (EDGE, and for the gradient pass SYM / MIRROR, are compile-time flags of the body: a uniform run-time test per element is a
  //  branch per element, and the elements of a thread then run one behind the other)
  auto tile_body = [&](auto EDGE_T, auto SYM_T, auto MIRROR_T) {
    constexpr bool EDGE = decltype(EDGE_T)::value, SYM = decltype(SYM_T)::value, MIRROR = decltype(MIRROR_T)::value;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      if (jt >= njt) break;
      const v4d acc = accs[jt];
      const int j = j0 + jt * 16 + lr;
      const double sj = sjv[jt];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + lk + 4 * r;
        const double r2 = (-2.0 * acc[r] + si[r]) + sj;
        double K, dK;
        kern_eval<KERNEL>(var, ls, r2, K, dK);
        if (GRAD) {
          double kb = kbv[jt][r];
          if (MIRROR) kb += (double)tmir[jt * 16 + lr][wave * 16 + lk + 4 * r];
          if (SYM) {
            kb = fma(-sym_gamma * pi[r], pj[jt], kb);
            if (EDGE) kb = i == j ? 0.5 * kb : kb;
          }
          if (EDGE) {
            const bool use = i < N && j < N && !(SYM && j > i);
            gv = fma(use ? kb : 0.0, use ? K : 0.0, gv);
            gl = fma(use ? kb : 0.0, use ? dK : 0.0, gl);
          } else {
            gv = fma(kb, K, gv);
            gl = fma(kb, dK, gl);
          }
        } else if (EDGE) {
          if (i >= N || j >= N) K = i == j ? 1.0 : 0.0;
          const double kj = j <= i ? K + (i == j ? jitter : 0.0) : 0.0;
          lrow[r][jt * 16] = kj;
          if (Mb.p) mrow[r][jt * 16] = m_scale * kj + (i == j ? (double)m_diag[(long)b * Np + i] : 0.0);
        } else {
          lrow[r][jt * 16] = K;
          if (Mb.p) mrow[r][jt * 16] = m_scale * K;
        }
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;
  if (!GRAD) {
    if (edge_tile) tile_body(T(), F(), F());
    else tile_body(F(), F(), F());
  } else if (sym) {
    if (edge_tile) tile_body(T(), T(), F());
    else tile_body(F(), T(), F());
  } else if (mirror) {
    if (edge_tile) tile_body(T(), F(), T());
    else tile_body(F(), F(), T());
  } else {
    tile_body(T(), F(), F());         // (diagonal tiles of a general iteration; the tiles above the diagonal have njt = 0)
  }
  if (GRAD) {
    __shared__ double red[2][4];
    gv = wave_sum(gv);
    gl = wave_sum(gl);
    if (lane == 0) { red[0][wave] = gv; red[1][wave] = gl; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const long t = (long)blockIdx.y * gridDim.x + blockIdx.x;
      const long nt = (long)gridDim.x * gridDim.y;
      gpart[((long)b * nt + t) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
      gpart[((long)b * nt + t) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
  }
}

// deterministic reduction of the tile partials + TF-2.8 Keras Adam on the two unconstrained hypers
// (models.py:192,210)
__global__ void adam_kernel(int ntiles, const double* gpart, double* ustate, double lr) {
  __shared__ double red[2][256];
  const int b = blockIdx.x;
  double gv = 0.0, gl = 0.0;
  for (int t = threadIdx.x; t < ntiles; t += 256) {
    gv += gpart[((long)b * ntiles + t) * 2];
    gl += gpart[((long)b * ntiles + t) * 2 + 1];
  }
  red[0][threadIdx.x] = gv;
  red[1][threadIdx.x] = gl;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* u = ustate + b * 8;
    const double var = softplus_d(u[0]), ls = softplus_d(u[1]);
    double g[2];
    g[0] = (red[0][0] / var) * (1.0 - exp(-var));   // softplus'(u) = 1 - exp(-softplus(u))
    g[1] = red[1][0] * (1.0 - exp(-ls));
    const double t = u[6] + 1.0;
    u[6] = t;
    const double lr_t = lr * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t));
    for (int q = 0; q < 2; ++q) {
      u[2 + q] = 0.9 * u[2 + q] + 0.1 * g[q];
      u[4 + q] = 0.999 * u[4 + q] + 0.001 * g[q] * g[q];
      u[q] -= lr_t * u[2 + q] / (sqrt(u[4 + q]) + 1e-7);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 64x64 diagonal block: Cholesky + inverse of the factor, one workgroup (4 waves) per member.
// The block lives in LDS and is processed as 4x4 sub-blocks of 16x16:
//   P1  the 16x16 diagonal sub-block is factored AND inverted in registers by 16 lanes of wave 0
//       (lane i owns row i; pivots/columns travel by DPP row broadcast; 1/L_ii comes from a Newton-refined v_rsq_f64,
//       so there is no fp64 divide or sqrt sequence on the serial chain)
//   P2  panel below:   P <- P inv(L_dd)^T          fp64 MFMA 16x16x4, one 16-row block per wave
//   P3  trailing part: S_rc -= P_r P_c^T           fp64 MFMA, lower block pairs dealt over the waves
//   P4  off-diagonal blocks of the inverse, block row by block row: M_IJ = -M_II sum_K L_IK M_KJ (MFMA)
// 3 barriers per block column instead of 5 per scalar column.
// ------------------------------------------------------------------------------------------------
constexpr int LS = 66;   // LDS row stride (doubles) of the 64x64 images
constexpr int LT = 18;   // row stride of the per-wave 16x16 scratch tiles

// broadcast of lane k (0..15) of every 16-lane row to the whole row (DPP row_newbcast, gfx90a+).  The 16x16
// diagonal sub-block is replicated in all four rows of the wave, so this is the "value of lane k" for every lane,
// it stays in VGPRs (v_readlane would put 2 x 136 values through the SGPR file and spill it into VGPR lanes).
template <int K>
__device__ __forceinline__ double bcast16_k(double v) {
  // one v_mov_b64_dpp: the 64-bit DPP move exists for exactly this control; every lane has a source lane, so the
  // "old" operand is never taken
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
  double r = __builtin_amdgcn_rsq(d);            // v_rsq_f64: ~2^-26 relative
  r = r * (1.5 - 0.5 * d * r * r);               // Newton: ~2^-51 after one step
  // one correction in fma form: r <- r + r * (1 - d r^2) / 2
  const double e = fma(-d * r, r, 1.0);
  return fma(0.5 * r, e, r);
}

// 16x16 MFMA tile helpers on LDS images (row stride deduced).  acc rows (lane>>4)+4r, col lane&15.
template <int LA>
__device__ __forceinline__ v4d tile_load(const double (*T)[LA], int r0, int c0, int lane) {
  v4d acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = T[r0 + (lane >> 4) + 4 * r][c0 + (lane & 15)];
  return acc;
}
template <int LA>
__device__ __forceinline__ void tile_store(double (*T)[LA], int r0, int c0, int lane, v4d acc) {
#pragma unroll
  for (int r = 0; r < 4; ++r) T[r0 + (lane >> 4) + 4 * r][c0 + (lane & 15)] = acc[r];
}
// acc += sgn * A(16x16 at (ar,ac)) * B, where B[k][j] = Bm[br + j][bc + k] (i.e. A * Bm^T) if BT else Bm[br + k][bc + j]
template <bool BT, int LA, int LB>
__device__ __forceinline__ v4d tile_mma(v4d acc, const double (*Am)[LA], int ar, int ac, const double (*Bm)[LB], int br,
                                        int bc, int lane, double sgn) {
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const double a = sgn * Am[ar + lr][ac + kk * 4 + lk];
    const double b = BT ? Bm[br + lr][bc + kk * 4 + lk] : Bm[br + kk * 4 + lk][bc + lr];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// `stamps` (diagnostic build of the call only, nullptr in the fit): wave 0 lane 0 of member 0 writes
// s_memtime at the phase boundaries into stamps[0..]
#define LEAF_STAMP(k) do { if (stamps && b == 0 && tid == 0) stamps[k] = __builtin_readcyclecounter(); } while (0)

// factor + invert the 64x64 image in S (lower part loaded, strict upper part of the off-diagonal 16-blocks zero),
// Mi zeroed; all 256 threads call, both images are complete and visible on return
__device__ __forceinline__ void leaf_factor(double (*S)[LS], double (*Mi)[LS], double (*Tt)[16][LT], int b, int tid,
                                            int pivot_base, int* info, unsigned long long* stamps) {
  const int lane = tid & 63, wave = tid >> 6;
  for (int bj = 0; bj < 4; ++bj) {
    const int d0 = 16 * bj;
    // ---- P1: factor + invert the 16x16 diagonal sub-block (wave 0, lanes 0..15 own rows) ----
    if (wave == 0) {
      double row[16], rinv[16], m[16];
      const int li = lane & 15;
      int bad = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) row[j] = S[d0 + li][d0 + j];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        double d = readlane_d(row[j], j);
        // branch-free pivot check (one basic block for the whole column loop): remember the first bad pivot
        const bool ok = d > 0.0;
        bad = (!ok && bad == 0) ? d0 + j + 1 : bad;
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
        __builtin_amdgcn_sched_barrier(0);   // keep the broadcasts of later columns from being hoisted (VGPR pressure)
      }
      // (opaque copies: without them the broadcasts below are merged with the factor loop's and 120 doubles stay
      //  live across both loops - more than the VGPR file)
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(row[j]));
      // inverse by columns: lane c owns column c; m[i] = (delta_ic - sum_{k<i} L_ik m_k) * rinv_i
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        double sacc = li == i ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) sacc = fma(-readlane_d(row[k], i), m[k], sacc);
        m[i] = sacc * rinv[i];
        asm volatile("" : "+v"(m[i]));       // pin the row here (else the chain sinks into the store block below
        __builtin_amdgcn_sched_barrier(0);   //  and all 120 broadcasts stay live)
      }
      if (bad != 0 && lane == 0 && info[b] == 0) info[b] = pivot_base + bad;
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          S[d0 + li][d0 + j] = row[j];
          Mi[d0 + j][d0 + li] = m[j];     // m[j] = Minv[j][li]
        }
      }
    }
    BAYENS_LDS_BARRIER();
    LEAF_STAMP(1 + 3 * bj);
    // ---- P2: panel rows below: P <- P * inv(L_dd)^T ----
    const int nrb = 3 - bj;                         // 16-row blocks below the diagonal sub-block
    if (wave < nrb) {
      const int r0 = d0 + 16 * (wave + 1);
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      acc = tile_mma<true>(acc, S, r0, d0, Mi, d0, d0, lane, 1.0);
      tile_store(S, r0, d0, lane, acc);
    }
    BAYENS_LDS_BARRIER();
    LEAF_STAMP(2 + 3 * bj);
    // ---- P3: trailing update of the lower block pairs (rb >= cb > bj) ----
    {
      int p = 0;
      for (int rb = bj + 1; rb < 4; ++rb)
        for (int cb = bj + 1; cb <= rb; ++cb, ++p)
          if ((p & 3) == wave) {
            v4d acc = tile_load(S, 16 * rb, 16 * cb, lane);
            acc = tile_mma<true>(acc, S, 16 * rb, d0, S, 16 * cb, d0, lane, -1.0);
            tile_store(S, 16 * rb, 16 * cb, lane, acc);
          }
    }
    BAYENS_LDS_BARRIER();
    LEAF_STAMP(3 + 3 * bj);
  }

  // ---- P4: off-diagonal 16x16 blocks of the inverse, block row I needs rows < I ----
  for (int I = 1; I < 4; ++I) {
    if (wave < I) {
      const int J = wave;
      v4d t = {0.0, 0.0, 0.0, 0.0};
      for (int K = J; K < I; ++K) t = tile_mma<false>(t, S, 16 * I, 16 * K, Mi, 16 * K, 16 * J, lane, 1.0);
      tile_store(Tt[wave], 0, 0, lane, t);
      // same wave wrote and reads Tt[wave]: make the LDS writes visible to its own later reads
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      acc = tile_mma<false>(acc, Mi, 16 * I, 16 * I, Tt[wave], 0, 0, lane, -1.0);
      tile_store(Mi, 16 * I, 16 * J, lane, acc);
    }
    BAYENS_LDS_BARRIER();
  }
  LEAF_STAMP(13);
}

__global__ __launch_bounds__(256) void potrf_leaf_kernel(Mat Ablk, real* inv, long inv_stride, int pivot_base,
                                                         int* info, unsigned long long* stamps) {
  __shared__ double S[64][LS];     // the block -> its lower factor
  __shared__ double Mi[64][LS];    // inverse of the factor
  __shared__ double Tt[4][16][LT]; // per-wave 16x16 scratch for P4
  const int b = blockIdx.x, tid = threadIdx.x;
  __builtin_amdgcn_s_setprio(3);   // a serial chain: win issue arbitration against co-resident throughput waves
  real* a = Ablk.p + b * Ablk.stride;
  {
    const int c = tid & 63, r0 = tid >> 6;
    for (int q = 0; q < 16; ++q) {
      const int r = r0 + 4 * q;
      S[r][c] = c <= r ? a[(long)r * Ablk.ld + c] : 0.0;
      Mi[r][c] = 0.0;
    }
  }
  BAYENS_LDS_BARRIER();
  LEAF_STAMP(0);
  leaf_factor(S, Mi, Tt, b, tid, pivot_base, info, stamps);
  real* iv = inv + b * inv_stride;
  {
    const int c = tid & 63, r0 = tid >> 6;
    for (int q = 0; q < 16; ++q) {
      const int r = r0 + 4 * q;
      a[(long)r * Ablk.ld + c] = c <= r ? S[r][c] : 0.0;     // lower factor, zeros above the diagonal
      iv[r * 64 + c] = c <= r ? Mi[r][c] : 0.0;
    }
  }
  LEAF_STAMP(14);
}

// Two leaves in one launch: the 128x128 diagonal block [A11 .; A21 A22] is factored by one workgroup -
// leaf(A11), L21 = A21 inv(L11)^T, A22 -= L21 L21^T, leaf(A22) - with L21 held in LDS between the steps.
// Replaces four launches (leaf, 64x64 panel solve, 64^3 update, leaf) at the bottom of the blocked Cholesky.
__global__ __launch_bounds__(256, 2) void potrf_leaf2_kernel(Mat Ablk, real* inv, long inv_stride, int pivot_base,
                                                          int* info) {
  // two 64x64 images rotate through the four steps (77 KB in all, so two workgroups - or one beside two GEMM
  // workgroups - fit a CU):   leaf 1: S = U, Mi = V | panel: T = U, Mi = V | update: T = U, S = V | leaf 2: S = V, Mi = U
  __shared__ double U[64][LS];
  __shared__ double V[64][LS];
  __shared__ double Tt[4][16][LT];
  unsigned long long* stamps = nullptr;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int c = tid & 63, rq = tid >> 6;
  __builtin_amdgcn_s_setprio(3);   // a serial chain: win issue arbitration against co-resident throughput waves
  const long ld = Ablk.ld;
  real* a = Ablk.p + b * Ablk.stride;
  real* iv = inv + b * inv_stride;
  // A21 is requested up front (it does not depend on the first leaf) and A22 right after the first leaf, BEFORE that
  // leaf's result stores: a load issued behind stores waits for them too (one in-order counter), and both blocks
  // are only needed a phase later.  (Requesting A22 up front as well costs 64 more live VGPRs - one workgroup per CU
  // instead of two, which large batches of small members notice.)
  real p21[16], p22[16];
  for (int q = 0; q < 16; ++q) {
    const int r = rq + 4 * q;
    U[r][c] = c <= r ? a[r * ld + c] : 0.0;
    V[r][c] = 0.0;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) p21[q] = a[(64 + rq + 4 * q) * ld + c];
  BAYENS_LDS_BARRIER();
  leaf_factor(U, V, Tt, b, tid, pivot_base, info, stamps);
#pragma unroll
  for (int q = 0; q < 16; ++q) p22[q] = a[(64 + rq + 4 * q) * ld + 64 + c];
#pragma unroll
  for (int q = 0; q < 16; ++q) {   // L11, inv(L11) out; A21 in (each thread replaces only its own elements of U)
    const int r = rq + 4 * q;
    a[r * ld + c] = c <= r ? U[r][c] : 0.0;
    iv[r * 64 + c] = c <= r ? V[r][c] : 0.0;
    U[r][c] = p21[q];
  }
  BAYENS_LDS_BARRIER();
  {   // L21 = A21 inv(L11)^T: wave w owns rows 16w..16w+15 (reads and writes only those rows of U)
    v4d acc[4];
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      acc[J] = v4d{0.0, 0.0, 0.0, 0.0};
      for (int K = 0; K <= J; ++K) acc[J] = tile_mma<true>(acc[J], U, 16 * wave, 16 * K, V, 16 * J, 16 * K, lane, 1.0);
    }
#pragma unroll
    for (int J = 0; J < 4; ++J) tile_store(U, 16 * wave, 16 * J, lane, acc[J]);
  }
  BAYENS_LDS_BARRIER();
#pragma unroll
  for (int q = 0; q < 16; ++q) {   // L21 out; A22 (lower) into V, whose inverse image is no longer needed
    const int r = rq + 4 * q;
    a[(64 + r) * ld + c] = U[r][c];
    V[r][c] = c <= r ? (double)p22[q] : 0.0;
  }
  BAYENS_LDS_BARRIER();
  {   // A22 -= L21 L21^T on the lower 16-block pairs
    int p = 0;
    for (int rb = 0; rb < 4; ++rb)
      for (int cb = 0; cb <= rb; ++cb, ++p)
        if ((p & 3) == wave) {
          v4d acc = tile_load(V, 16 * rb, 16 * cb, lane);
          for (int K = 0; K < 4; ++K) acc = tile_mma<true>(acc, U, 16 * rb, 16 * K, U, 16 * cb, 16 * K, lane, -1.0);
          tile_store(V, 16 * rb, 16 * cb, lane, acc);
        }
  }
  BAYENS_LDS_BARRIER();
  for (int q = 0; q < 16; ++q) U[rq + 4 * q][c] = 0.0;
  BAYENS_LDS_BARRIER();
  leaf_factor(V, U, Tt, b, tid, pivot_base + 64, info, stamps);
  for (int q = 0; q < 16; ++q) {
    const int r = rq + 4 * q;
    a[(64 + r) * ld + 64 + c] = c <= r ? V[r][c] : 0.0;
    iv[4096 + r * 64 + c] = c <= r ? U[r][c] : 0.0;
  }
}

// ------------------------------------------------------------------------------------------------
// matrix-vector kernels
// ------------------------------------------------------------------------------------------------
// out = beta out + alpha L^T (v/nu): one workgroup per 64 columns; lanes = columns (512-byte coalesced row
// segments), the 4 waves split the rows i >= j0 (L is lower triangular).
// (16 waves: the strip of column block 0 is n rows long and one workgroup's loads in flight bound its rate)
constexpr int LTD_WAVES = 16;
__global__ __launch_bounds__(64 * LTD_WAVES) void lt_dvec_kernel(int n, Mat L, const real* y, const real* nu, double alpha,
                                                                 double beta, real* theta1) {
  __shared__ double part[LTD_WAVES][64];
  const int b = blockIdx.y, j0 = blockIdx.x * 64;
  const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
  const real* l = L.p + b * L.stride;
  // rows j0 .. n-1 of the 64-column strip, 16 waves x 8 rows in flight (the strip of column block 0 is n rows long: the pass is
  // bound by how many loads a workgroup keeps in flight, so the row weights v / nu - one division per row, the same in every lane -
  // are formed once, in LDS, and the loop is loads and multiply-adds only)
  extern __shared__ double dvec[];        // [n - j0]
  const long vb = (long)b * n;
  for (int i = j0 + (int)threadIdx.x; i < n; i += 64 * LTD_WAVES)
    dvec[i - j0] = (double)y[vb + i] / (nu ? (double)nu[vb + i] : 1.0);
  __syncthreads();
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  constexpr int W = LTD_WAVES;
  const real* lc = l + j0 + c;
  int i = j0 + w;
  for (; i + 7 * W < n; i += 8 * W) {
    double a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = lc[(long)(i + q * W) * L.ld];
    s0 += a[0] * dvec[i - j0];
    s1 += a[1] * dvec[i + W - j0];
    s2 += a[2] * dvec[i + 2 * W - j0];
    s3 += a[3] * dvec[i + 3 * W - j0];
    s0 += a[4] * dvec[i + 4 * W - j0];
    s1 += a[5] * dvec[i + 5 * W - j0];
    s2 += a[6] * dvec[i + 6 * W - j0];
    s3 += a[7] * dvec[i + 7 * W - j0];
  }
  for (; i < n; i += W) s0 += lc[(long)i * L.ld] * dvec[i - j0];
  const double s = (s0 + s1) + (s2 + s3);
  part[w][c] = s;
  __syncthreads();
  if (w == 0) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < W; ++q) tot += part[q][c];
    real* t = theta1 + (long)b * n + j0 + c;
    *t = (beta != 0.0 ? beta * (*t) : 0.0) + alpha * tot;
  }
}

// out = S v, S symmetric, elements on and below the diagonal read only: one workgroup per 64 columns j0..j0+63 sums the
// column part sum_{i >= j} S_ij v_i (lanes = columns, the 4 waves split the rows, as in lt_dvec_kernel) and the row part
// sum_{k < j} S_jk v_k (16 rows per wave, lanes stride k): 64 n elements per workgroup wherever the strip lies.
constexpr int SYMV_WAVES = 16;
__global__ __launch_bounds__(64 * SYMV_WAVES) void symv_lower_kernel(int n, Mat S, const real* v, real* out) {
  constexpr int W = SYMV_WAVES;
  __shared__ double part[W][64];
  __shared__ double rowp[64];
  const int b = blockIdx.y, j0 = blockIdx.x * 64;
  const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
  const real* s = S.p + b * S.stride;
  const long vb = (long)b * n;
  // column part: rows i >= j0 of the strip, 16 waves x 8 rows in flight (as lt_dvec_kernel: the loads in flight bound the rate)
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const real* sc = s + j0 + c;
  int i = j0 + w;
  for (; i + 7 * W < n; i += 8 * W) {
    double a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = sc[(long)(i + q * W) * S.ld];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double t = (i + q * W >= j0 + c ? a[q] : 0.0) * (double)v[vb + i + q * W];
      if ((q & 3) == 0) s0 += t;
      else if ((q & 3) == 1) s1 += t;
      else if ((q & 3) == 2) s2 += t;
      else s3 += t;
    }
  }
  for (; i < n; i += W) s0 += (i >= j0 + c ? (double)sc[(long)i * S.ld] : 0.0) * (double)v[vb + i];
  part[w][c] = (s0 + s1) + (s2 + s3);
  // row part: rows j0 + 4 w .. + 3 of the strip, the four rows of a wave side by side (lanes stride k)
  {
    const int jb = j0 + 4 * w;
    const real* row = s + (long)jb * S.ld;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    for (int k = c; k < jb + 3; k += 64) {
      const double vk = (double)v[vb + k];
      const double r0 = row[k], r1 = row[S.ld + k], r2 = row[2L * S.ld + k], r3 = row[3L * S.ld + k];
      t0 += (k < jb ? r0 : 0.0) * vk;
      t1 += (k < jb + 1 ? r1 : 0.0) * vk;
      t2 += (k < jb + 2 ? r2 : 0.0) * vk;
      t3 += r3 * vk;
    }
    t0 = wave_sum(t0); t1 = wave_sum(t1); t2 = wave_sum(t2); t3 = wave_sum(t3);
    if (c == 0) { rowp[4 * w] = t0; rowp[4 * w + 1] = t1; rowp[4 * w + 2] = t2; rowp[4 * w + 3] = t3; }
  }
  __syncthreads();
  if (w == 0) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < W; ++q) tot += part[q][c];
    out[vb + j0 + c] = (real)(tot + rowp[c]);
  }
}

// out_i = sum_{k < klim(i)} A_ik x_k - (sub ? sub_i : 0); one wave per row, lanes stride k.
__global__ __launch_bounds__(256) void gemv_rows_kernel(int n, Mat A, const real* x, const real* sub,
                                                        real* out, int lower) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const real* a = A.p + b * A.stride + (long)i * A.ld;
  const int klim = lower ? i + 1 : n;
  double s = 0.0;
  for (int k = lane; k < klim; k += 64) s += a[k] * x[(long)b * n + k];
  s = wave_sum(s);
  if (lane == 0) out[(long)b * n + i] = s - (sub ? sub[(long)b * n + i] : 0.0);
}

// out_i = sum_k A_ik^2; one wave per row
__global__ __launch_bounds__(256) void row_sqsum_kernel(int n, Mat A, real* out) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const real* a = A.p + b * A.stride + (long)i * A.ld;
  double s = 0.0;
  for (int k = lane; k < n; k += 64) s += a[k] * a[k];
  s = wave_sum(s);
  if (lane == 0) out[(long)b * n + i] = s;
}

// prims.h elbo_combine: one workgroup per member, fixed-order reduction
__global__ __launch_bounds__(256) void elbo_combine_kernel(int N, int Np, const real* a, const real* f,
                                                           const real* nu, const real* mvec, Mat C,
                                                           const double* logdetC, const double* trS, double* trace,
                                                           int stride, int slot) {
  __shared__ double red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const long o = (long)b * Np;
  double s = 0.0;
  for (int t = tid; t < N; t += 256) {
    const double v = nu[o + t], at = a[o + t], mt = mvec[o + t];
    s += 0.5 * 1.8378770664093453 + 0.5 * log(v) + 0.5 * (at * at + f[o + t]) / v + 0.5 * mt * mt;
  }
  for (int t = N + tid; t < Np; t += 256) {       // the padded block of C^-1 is diag(1 / C_tt): not part of tr S
    const double ctt = C.p[b * C.stride + (long)t * C.ld + t];
    s -= 0.5 / (ctt * ctt);
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    const double data_and_mm = (red[0] + red[1]) + (red[2] + red[3]);
    trace[(long)b * stride + slot] = data_and_mm + 0.5 * (-(double)N + 2.0 * logdetC[b] + trS[b]);
  }
}

// the 64-row block appended to a right-hand side (prims.h aug_set / aug_get)
__global__ void aug_set_kernel(int n, Mat rows, const real* vec, int zero_rows) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  real* r = rows.p + b * rows.stride;
  r[j] = vec[(long)b * n + j];
  for (int q = 1; q <= zero_rows; ++q) r[(long)q * rows.ld + j] = 0.0;
}

__global__ void aug_get_kernel(int n, Mat rows, real* vec) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n) vec[(long)b * n + j] = rows.p[b * rows.stride + j];
}

// ------------------------------------------------------------------------------------------------
// LogLikelihoodWeight (weights.py:88-104): the MVN is evaluated at the constant vector x_t * 1_N,
// log N = -1/2 || x_t a - c ||^2 - logdet - N/2 log 2pi with a = Lc^-1 1, c = Lc^-1 mu.
// ------------------------------------------------------------------------------------------------
// ---- a caller-supplied lower factor (compact [N][N], any N) as the operand of the blocked solves ----------------
// tile (by, bx <= by) of the padded lower image: the compact factor's lower part, identity in the padding, zeros above
// the diagonal inside the diagonal blocks (the 64-blocks above the block diagonal are never read by the solves)
__global__ void pad_lower_kernel(int N, int Np, const double* src, Mat dst) {
  const int b = blockIdx.z;
  if (blockIdx.x > blockIdx.y) return;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6);
  const double* s = src + (long)b * N * N;
  real* d = dst.p + b * dst.stride;
  const int jc = j < N ? j : N - 1;
  double v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + 4 * q;
    v[q] = s[(long)(i < N ? i : N - 1) * N + jc];          // (clamped address: no load under a condition)
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + 4 * q;
    d[(long)i * dst.ld + j] = (i < N && j < N) ? (j <= i ? v[q] : 0.0) : (i == j ? 1.0 : 0.0);
  }
}
// inverse of every 64x64 diagonal block of the padded factor: one workgroup per (block, member).  The triangular block
// is inverted DIRECTLY by forward substitution (lane c solves T x = e_c in LDS; row i of T is a broadcast read), which is
// backward stable in cond(T).  (Until round 3 the block was re-factored from T T^T by the Cholesky leaf: that squares
// the condition number and could fail its pivot test on a valid, ill-conditioned factor.)  A diagonal entry that is not
// a positive finite number is recorded in info[b] (1-based row of the padded matrix) - callers turn it into
// BAYENS_E_NOTPD instead of returning garbage.
__global__ __launch_bounds__(256) void tri_inv64_kernel(Mat Lp, real* inv, long inv_stride, int* info) {
  __shared__ double T[64][LS];
  __shared__ double X[64][LS];
  const int blk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const real* a = Lp.p + b * Lp.stride + (long)(64 * blk) * Lp.ld + 64 * blk;
  {
    const int c = tid & 63, r0 = tid >> 6;
    for (int q = 0; q < 16; ++q) {
      const int r = r0 + 4 * q;
      T[r][c] = c <= r ? (double)a[(long)r * Lp.ld + c] : 0.0;
    }
  }
  __syncthreads();
  if (wave == 0) {
    const int c = lane;
    const double dc = T[c][c];
    if (!(dc > 0.0) || !(dc < 1.79e308)) atomicCAS(&info[b], 0, 64 * blk + c + 1);      // NaN / inf / <= 0
    for (int i = 0; i < 64; ++i) {
      double acc = i == c ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) acc -= T[i][k] * X[k][c];        // X[k][c] == 0 for k < c
      X[i][c] = c <= i ? acc / T[i][i] : 0.0;
    }
  }
  __syncthreads();
  real* iv = inv + b * inv_stride + (long)blk * 4096;
  {
    const int c = tid & 63, r0 = tid >> 6;
    for (int q = 0; q < 16; ++q) {
      const int r = r0 + 4 * q;
      iv[r * 64 + c] = (real)X[r][c];
    }
  }
}
// right-hand-side rows of a solve with the factor: rows[b][r][:] = x[b or 0][r][:] - mu[b][:] (mode 0: log_prob
// arguments, x_stride = 0 shares the arguments between the members) or the pair (1, mu) (mode 1: the log-likelihood
// basis); 64 rows per member, the unused ones and the padded columns zero
__global__ void solve_rows_set_kernel(int N, int Np, int nrows, int mode, const double* x, long x_stride, const double* mu,
                                      Mat rows) {
  const int b = blockIdx.z, r = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Np) return;
  double v = 0.0;
  if (j < N && r < nrows) {
    if (mode == 1) v = r == 0 ? 1.0 : mu[(long)b * N + j];
    else v = x[b * x_stride + (long)r * N + j] - mu[(long)b * N + j];
  }
  rows.p[b * rows.stride + (long)r * rows.ld + j] = v;
}
// log N(x_r; mu, L L^T) = -|z_r|^2 / 2 - logdet - N log(2 pi) / 2 from the solved rows z_r = L^-1 (x_r - mu)
__global__ __launch_bounds__(256) void rows_logprob_kernel(int N, int nrows, Mat rows, const double* logdet, double* out) {
  __shared__ double red[256];
  const int b = blockIdx.y, r = blockIdx.x;
  const real* z = rows.p + b * rows.stride + (long)r * rows.ld;
  double s = 0.0;
  for (int j = threadIdx.x; j < N; j += 256) s += (double)z[j] * (double)z[j];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[(long)b * nrows + r] = -0.5 * red[0] - logdet[b] - 0.5 * N * LOG2PI;
}

// rows appended to the covariance factorisation (prims.h basis_rows_set / _get)
__global__ void basis_rows_set_kernel(int N, int Np, Mat rows, const real* mu_pad, int in_padding) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Np) return;
  real* r = rows.p + b * rows.stride;
  double v0 = j < N ? 1.0 : 0.0, v1 = j < N ? mu_pad[(long)b * Np + j] : 0.0;
  if (in_padding) {            // rows N and N+1 of the matrix itself: huge pivots keep them positive definite
    if (j == N) v0 = 1e30;
    if (j == N + 1) v1 = 1e30;
  }
  r[j] = v0;
  r[(long)rows.ld + j] = v1;
  if (!in_padding)
    for (int q = 2; q < 64; ++q) r[(long)q * rows.ld + j] = 0.0;
}

__global__ void basis_rows_get_kernel(int N, Mat rows, real* basis) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  const real* r = rows.p + b * rows.stride;
  basis[((long)b * 2) * N + j] = r[j];
  basis[((long)b * 2 + 1) * N + j] = r[(long)rows.ld + j];
}

// (a.a, a.c, c.c) of the log-likelihood basis a = L_c^-1 1, c = L_c^-1 mu: one workgroup per member,
// fixed-order tree reduction (deterministic)
__global__ __launch_bounds__(256) void basis_dots_kernel(int N, const double* basis, double* dots) {
  __shared__ double red[3][256];
  const int b = blockIdx.x;
  const double* a = basis + ((long)b * 2) * N;
  const double* c = a + N;
  double aa = 0.0, ac = 0.0, cc = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) {
    const double ai = a[i], ci = c[i];
    aa += ai * ai; ac += ai * ci; cc += ci * ci;
  }
  red[0][threadIdx.x] = aa; red[1][threadIdx.x] = ac; red[2][threadIdx.x] = cc;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (threadIdx.x < s2)
      for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x < 3) dots[b * 3 + threadIdx.x] = red[threadIdx.x][0];
}

// weights.py:98-104 from the basis: ||x a - c||^2 = x^2 a.a - 2 x a.c + c.c ; mean over realisations
__global__ void loglik_basis_kernel(int N, int R_obs, const double* dots, const double* logdet, const double* obs_all,
                                    int n_groups, int b0, double* ll_mean) {
  const int b = b0 + blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  const double* obs = obs_all + (long)(b % n_groups) * R_obs * N;       // one observation set per group of posteriors
  const double aa = dots[b * 3], ac = dots[b * 3 + 1], cc = dots[b * 3 + 2];
  const double cst = -logdet[b] - 0.5 * N * LOG2PI;
  double acc = 0.0;
  for (int r = 0; r < R_obs; ++r) {
    const double x = obs[(long)r * N + t];
    acc += -0.5 * ((x * x) * aa - 2.0 * x * ac + cc) + cst;
  }
  ll_mean[(long)b * N + t] = acc / R_obs;
}

// dx.Normal branch (weights.py:95-96): elementwise log N(x; mu, scale), mean over realisations
__global__ void normal_loglik_kernel(int N, int R_obs, const double* mu, const double* scale, const double* obs,
                                     double* ll_mean) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  const double m = mu[(long)b * N + t], s = scale[(long)b * N + t];
  double acc = 0.0;
  for (int r = 0; r < R_obs; ++r) {
    const double z = (obs[(long)r * N + t] - m) / s;
    acc += -0.5 * z * z - log(s) - 0.5 * LOG2PI;
  }
  ll_mean[(long)b * N + t] = acc / R_obs;
}

// weights.py:107,118-123: w_raw = exp(c ll), w = w_raw / sum_m w_raw; thread per point, coalesced over points
// weights.py:107,118-123.  64 points per workgroup, the members dealt over the four waves (the sums over the members
// are latency chains of exp(): one thread per point made them the kernel's whole duration).
__global__ __launch_bounds__(256) void normalise_kernel(int M, int P, const double* ll_mean, double c_std, double* w,
                                                        double* w_raw) {
  __shared__ double part[4][64];
  const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tx;
  double s = 0.0;
  if (t < P)
    for (int m = g; m < M; m += 4) s += exp(c_std * ll_mean[(long)m * P + t]);
  part[g][tx] = s;
  __syncthreads();
  const double sum = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
  if (t >= P) return;
  for (int m = g; m < M; m += 4) {
    const double raw = exp(c_std * ll_mean[(long)m * P + t]);
    w[(long)m * P + t] = raw / sum;
    if (w_raw) w_raw[(long)m * P + t] = raw;
  }
}

// weights.py:302-325,331 with wasserstein.py:21-47 on 1x1 Normal(mean, scale=variance):
// D[i,j,t] = |mu_it - mu_jt| + (v_it - v_jt)^2 ; w[i,t] = mean_j D / sum_i mean_j D
__global__ void w2_temporal_kernel(int M, int P, const double* mean, const double* var, double* w) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P) return;
  double tot = 0.0;
  for (int i = 0; i < M; ++i) {
    const double mi = mean[(long)i * P + t], vi = var[(long)i * P + t];
    double d = 0.0;
    for (int j = 0; j < M; ++j) {
      const double dv = vi - var[(long)j * P + t];
      d += fabs(mi - mean[(long)j * P + t]) + dv * dv;
    }
    d /= M;
    w[(long)i * P + t] = d;
    tot += d;
  }
  for (int i = 0; i < M; ++i) w[(long)i * P + t] /= tot;
}

// models.py:175-182 at every grid point of a (realisation, time, *space) field: X = realisation_set^T, y_var = np.var over
// the realisations (ddof = 0, two passes like NumPy), y_mean = the mean over realisations (the documented stand-in for
// the DTW barycentre).  raw [M][R][N][G] as the caller's array lies in memory (grid points contiguous: coalesced reads);
// outputs in the fit's layout, fit index b = m * G + g: X [B][N][R], y_mean / y_var [B][N].
__global__ void field_prologue_kernel(int R, int N, int G, const double* raw, double* X, double* y_mean, double* y_var) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y, m = blockIdx.z;
  if (g >= G) return;
  const double* src = raw + ((long)m * R * N + n) * G + g;
  const long b = (long)m * G + g;
  double* xr = X + (b * N + n) * R;
  double sum = 0.0;
  for (int r = 0; r < R; ++r) {
    const double v = src[(long)r * N * G];
    xr[r] = v;
    sum += v;
  }
  const double mean = sum / R;
  double sq = 0.0;
  for (int r = 0; r < R; ++r) {
    const double d = src[(long)r * N * G] - mean;
    sq += d * d;
  }
  y_mean[b * N + n] = mean;
  y_var[b * N + n] = sq / R;
}

// [nb][A][B] -> [nb][B][A] through a padded 32 x 32 LDS tile (both sides coalesced): the per-gridpoint field keeps its
// tables fit-major ([member][grid point][time], what the fused fit kernel writes), the plug-in classes speak the data's
// (time, lat, lon) order
__global__ void transpose_rows_kernel(int A, int B, const double* src, double* dst) {
  __shared__ double tile[32][33];
  const long base = (long)blockIdx.z * A * B;
  const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8)
    if (a0 + r < A && b0 + tx < B) tile[r][tx] = src[base + (long)(a0 + r) * B + b0 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (b0 + r < B && a0 + tx < A) dst[base + (long)(b0 + r) * A + a0 + tx] = tile[tx][r];
}

// weights.py:267-300 (mode='spatial', dx.Normal branch) with wasserstein.py:21-47, full_cov=False: per grid point p the
// reference builds dx.Normal(mean_i[:, p], scale = variance_i[:, p]) over the T time steps, so
//   D[i,j,p] = || mean_i[:,p] - mean_j[:,p] ||_2  +  sum_t (V_i[t,p] - V_j[t,p])^2        (V = distribution.variance)
// and w[i,p] = nanmean_j D[i,j,p], normalised over i.  Tables are [M][T][G] (time-major per model, grid points
// contiguous: a wave reads 512 contiguous bytes of one time step).  Thread (p, i) forms row i's mean distance.
__global__ void w2_spatial_rows_kernel(int M, int T, int G, const double* mean, const double* V, double* w) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (p >= G) return;
  const double* mi = mean + (long)i * T * G + p;
  const double* vi = V + (long)i * T * G + p;
  double acc = 0.0;
  int n = 0;
  for (int j = 0; j < M; ++j) {
    const double* mj = mean + (long)j * T * G + p;
    const double* vj = V + (long)j * T * G + p;
    double dm = 0.0, dv = 0.0;
    for (int t = 0; t < T; ++t) {
      const double a = mi[(long)t * G] - mj[(long)t * G], b = vi[(long)t * G] - vj[(long)t * G];
      dm += a * a;
      dv += b * b;
    }
    const double d = sqrt(dm) + dv;
    if (d == d) { acc += d; ++n; }          // np.nanmean over j
  }
  w[(long)i * G + p] = n ? acc / n : NAN;
}
__global__ void column_normalise_kernel(int M, int P, double* w) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double tot = 0.0;
  for (int i = 0; i < M; ++i) tot += w[(long)i * P + p];
  for (int i = 0; i < M; ++i) w[(long)i * P + p] /= tot;
}

// ensemble_scheme.py:52-78 + wasserstein.py:61-100 (signed convergence test, <= 201 sweeps)
// ensemble_scheme.py:43-81 + wasserstein.py:61-100: per point, mu* = sum_m w mu and the fixed point
// v <- sum_m w sqrt(v) sigma_m with the reference's signed stopping test.  Same layout as normalise_kernel: 64 points
// per workgroup, members dealt over four waves, the block leaves the loop when all of its points have stopped.
__global__ __launch_bounds__(256) void barycentre_kernel(int M, int P, const double* w, const double* mean,
                                                         const double* var, double tol, double init_var, double* mu_out,
                                                         double* var_out, int* n_bad) {
  __shared__ double part[4][64];
  const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tx;
  double v = init_var;
  int n_iters = 0;
  bool done = t >= P;
  while (true) {
    double p = 0.0;
    if (!done) {
      const double sv = sqrt(v);
      for (int m = g; m < M; m += 4) p += w[(long)m * P + t] * sv * sqrt(var[(long)m * P + t]);
    }
    part[g][tx] = p;
    __syncthreads();
    const double cand = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
    if (!done) {
      if (cand - v < tol) { v = cand; done = true; }
      else {
        v = cand;
        ++n_iters;
        if (n_iters > 200) { if (g == 0) atomicAdd(n_bad, 1); done = true; }
      }
    }
    if (__syncthreads_and(done)) break;       // (also the barrier that protects `part` for the next round)
  }
  double p = 0.0;
  if (t < P)
    for (int m = g; m < M; m += 4) p += w[(long)m * P + t] * mean[(long)m * P + t];
  part[g][tx] = p;
  __syncthreads();
  if (g == 0 && t < P) {
    mu_out[t] = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
    const double sig = sqrt(v);
    var_out[t] = sig * sig;    // Barycentre passes bary_stds ** 2 (ensemble_scheme.py:76)
  }
}

// ensemble_scheme.py:136-160: mean and (std ** 2) across realisations (ddof = 0), weighted
__global__ void weighted_model_mean_kernel(int M, int R_max, const int* R, int P, const double* data,
                                           const double* w, double* mu_out, double* var_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P) return;
  double wm = 0.0, wv = 0.0;
  for (int m = 0; m < M; ++m) {
    const int Rm = R[m];
    const double* d = data + (long)m * R_max * P;
    double s = 0.0;
    for (int r = 0; r < Rm; ++r) s += d[(long)r * P + t];
    const double mean = s / Rm;
    double q = 0.0;
    for (int r = 0; r < Rm; ++r) { const double e = d[(long)r * P + t] - mean; q += e * e; }
    const double sd = sqrt(q / Rm);
    const double wt = w[(long)m * P + t];
    wv += (sd * sd) * (wt * wt);
    wm += mean * wt;
  }
  mu_out[t] = wm;
  var_out[t] = wv;
}

// ------------------------------------------------------------------------------------------------
// full-covariance Wasserstein distance (wasserstein.py:21-47, weights.py:240-265): reductions around the
// polar iteration of driver.cpp::nuclear_norms.  Two-phase deterministic reductions: 64x64 tile partials,
// then one small kernel per member.
// ------------------------------------------------------------------------------------------------
// mode 0: p0 = sum A^2 (Frobenius);  mode 1: X <- a X + b T, p0 = sum (new-old)^2, p1 = sum new^2;
// mode 2: p0 = sum A B
__global__ __launch_bounds__(256) void tile_reduce_kernel(int mode, Mat A, Mat B, double a, double bcoef, double* part) {
  __shared__ double red[2][4];
  const int b = blockIdx.z;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6);
  real* pa = A.p + b * A.stride;
  const real* pb = B.p ? B.p + b * B.stride : nullptr;
  double s0 = 0.0, s1 = 0.0;
  // (every load of the thread first - a load behind the store of the element before it waits for that store - then the sums, in
  //  the order they always had, then the stores)
  real xv[16], bv[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    xv[q] = pa[(long)(i0 + 4 * q) * A.ld + j];
    bv[q] = pb ? pb[(long)(i0 + 4 * q) * B.ld + j] : (real)0;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    if (mode == 0) { s0 += xv[q] * xv[q]; }
    else if (mode == 1) {
      const double nv = a * xv[q] + bcoef * bv[q];
      const double d = nv - xv[q];
      s0 += d * d; s1 += nv * nv;
      xv[q] = nv;
    } else { s0 += xv[q] * bv[q]; }
  }
  if (mode == 1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) pa[(long)(i0 + 4 * q) * A.ld + j] = xv[q];
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const long t = (long)blockIdx.y * gridDim.x + blockIdx.x, nt = (long)gridDim.x * gridDim.y;
    part[((long)b * nt + t) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[((long)b * nt + t) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// mode 0: out = sqrt(p0) ; mode 1: out = sqrt(p0 / p1) ; mode 2: out = p0
__global__ void tile_finish_kernel(int mode, int ntiles, const double* part, double* out) {
  __shared__ double red[2][256];
  const int b = blockIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int t = threadIdx.x; t < ntiles; t += 256) { s0 += part[((long)b * ntiles + t) * 2]; s1 += part[((long)b * ntiles + t) * 2 + 1]; }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (threadIdx.x < s2) { red[0][threadIdx.x] += red[0][threadIdx.x + s2]; red[1][threadIdx.x] += red[1][threadIdx.x + s2]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[b] = mode == 0 ? sqrt(red[0][0]) : (mode == 1 ? sqrt(red[0][0] / red[1][0]) : red[0][0]);
}

// norms[b] <- min(norms[b], cap_scale * cap[b]): a tighter bound of the two-norm than the Frobenius norm, when the caller has one
__global__ void cap_norms_kernel(int nb, double* norms, const double* cap, double cap_scale) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const double c = cap_scale * cap[b];
  if (c > 0.0 && c < norms[b]) norms[b] = c;
}

// max column sum / max row sum of |A| over the lower triangle of every member: the brackets
//   ||A||_2 <= sqrt(||A||_1 ||A||_inf)  of the polar iteration's starting interval.  out[b][0..1], zeroed by the caller
// (non-negative doubles order like their bit patterns, so the maximum is an integer atomicMax).
__global__ __launch_bounds__(256) void tri_absnorms_kernel(int n, Mat A, double* out) {
  __shared__ double red[4];
  const int b = blockIdx.z, which = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const real* pa = A.p + b * A.stride;
  double best = 0.0;
  if (which == 0) {                       // column sums: one thread per column, adjacent threads adjacent columns
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < n) {
      double s = 0.0;
      for (int r = c; r < n; ++r) s += fabs((double)pa[(long)r * A.ld + c]);
      best = s;
    }
  } else {                                // row sums: one wave per row, 64 rows per workgroup
    for (int q = 0; q < 64; ++q) {
      const int r = blockIdx.x * 256 + wave * 64 + q;
      if (r >= n) break;
      double s = 0.0;
      for (int c = lane; c <= r; c += 64) s += fabs((double)pa[(long)r * A.ld + c]);
      s = wave_sum(s);
      best = s > best ? s : best;
    }
  }
  best = wave_max(best);
  if (lane == 0) red[wave] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    atomicMax(reinterpret_cast<unsigned long long*>(out + (long)b * 2 + which), (unsigned long long)__double_as_longlong(m));
  }
}

__global__ void scale_by_norm_kernel(int n, Mat M, Mat X, const double* norms) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6);
  const double r = 1.0 / norms[b];
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + 4 * q;
    X.p[b * X.stride + (long)i * X.ld + j] = M.p[b * M.stride + (long)i * M.ld + j] * r;
  }
}

// one workgroup per ordered pair (i, j): ||mu_i - mu_j||_2 and, for diagonal posteriors, sum_t (sd_i - sd_j)^2
__global__ __launch_bounds__(256) void pair_terms_kernel(int M, int N, const double* mu, const double* sd, double* gap,
                                                         double* diagterm) {
  __shared__ double red[2][4];
  const int i = blockIdx.x / M, j = blockIdx.x % M;
  double s0 = 0.0, s1 = 0.0;
  for (int t = threadIdx.x; t < N; t += 256) {
    const double d = mu[(long)i * N + t] - mu[(long)j * N + t];
    s0 += d * d;
    if (sd) { const double e = sd[(long)i * N + t] - sd[(long)j * N + t]; s1 += e * e; }
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    gap[(long)i * M + j] = sqrt((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));   // un-squared norm (wasserstein.py:40)
    if (diagterm) diagterm[(long)i * M + j] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// D[i][j] = gap + tr_i + tr_j - 2 nuc_ij (full covariance; nuc holds i<j, D[i][i] = 0) or gap + diagterm;
// w_i = mean_j D[i][j] / sum_i mean_j D[i][j]      (weights.py:259-265,331)
__global__ void w2_single_finish_kernel(int M, const double* tr, const double* nuc, const double* gap,
                                        const double* diagterm, double* D, double* w) {
  __shared__ double rowmean[1024];
  for (int i = threadIdx.x; i < M; i += blockDim.x) {
    double s = 0.0;
    for (int j = 0; j < M; ++j) {
      double d;
      if (diagterm) d = gap[(long)i * M + j] + diagterm[(long)i * M + j];
      else if (i == j) d = 0.0;
      else {
        const double nn = i < j ? nuc[(long)i * M + j] : nuc[(long)j * M + i];
        d = gap[(long)i * M + j] + ((tr[i] + tr[j]) - 2.0 * nn);
      }
      if (D) D[(long)i * M + j] = d;
      s += d;
    }
    rowmean[i] = s / M;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < M; ++i) tot += rowmean[i];
    for (int i = 0; i < M; ++i) w[i] = rowmean[i] / tot;
  }
}

// ------------------------------------------------------------------------------------------------
// the remaining weight plug-ins, one thread per point, [M][P] tables
// ------------------------------------------------------------------------------------------------
// CRPSWeight (weights.py:444-515): properscoring.crps_gaussian against Normal(mean, scale = variance)
__global__ void crps_weights_kernel(int M, int P, int R_obs, const double* mean, const double* var, const double* obs,
                                    double* w, double* score) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P) return;
  const double inv_sqrt_pi = 0.5641895835477563, inv_sqrt_2pi = 0.3989422804014327, inv_sqrt2 = 0.7071067811865476;
  double tot = 0.0;
  for (int m = 0; m < M; ++m) {
    const double mu = mean[(long)m * P + t], sg = var[(long)m * P + t];
    double acc = 0.0;
    for (int r = 0; r < R_obs; ++r) {
      const double z = (obs[(long)r * P + t] - mu) / sg;
      const double cdf = 0.5 * erfc(-z * inv_sqrt2), pdf = inv_sqrt_2pi * exp(-0.5 * z * z);
      acc += sg * (z * (2.0 * cdf - 1.0) + 2.0 * pdf - inv_sqrt_pi);
    }
    const double sc = acc / R_obs;
    if (score) score[(long)m * P + t] = sc;
    w[(long)m * P + t] = 1.0 / sc;
    tot += 1.0 / sc;
  }
  for (int m = 0; m < M; ++m) w[(long)m * P + t] /= tot;
}

// KSDWeight (weights.py:336-441): inverse-multiquadric Stein kernel, c = 1, beta = -1/2, d = 1
__global__ void ksd_weights_kernel(int M, int P, int R_obs, const double* mean, const double* var, const double* obs,
                                   double* w, double* score) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P) return;
  const double beta = -0.5;
  double tot = 0.0;
  for (int m = 0; m < M; ++m) {
    const double mu = mean[(long)m * P + t], v = var[(long)m * P + t];
    const double iv2 = 1.0 / (v * v);
    double sum = 0.0;
    for (int a = 0; a < R_obs; ++a) {
      const double xa = obs[(long)a * P + t], ga = -(xa - mu) * iv2;
      for (int b = 0; b < R_obs; ++b) {
        const double xb = obs[(long)b * P + t], gb = -(xb - mu) * iv2;
        const double diff = xa - xb, q = 1.0 + diff * diff;
        const double qb = pow(q, beta), qb1 = qb / q, qb2 = qb1 / q;
        sum += ga * gb * qb - 2.0 * beta * ga * diff * qb1 + 2.0 * beta * gb * diff * qb1 - 2.0 * beta * qb1 -
               4.0 * beta * (beta - 1.0) * qb2 * diff * diff;
      }
    }
    const double ksd = sqrt(sum) / R_obs;
    if (score) score[(long)m * P + t] = ksd;
    w[(long)m * P + t] = 1.0 / ksd;
    tot += 1.0 / ksd;
  }
  for (int m = 0; m < M; ++m) w[(long)m * P + t] /= tot;
}

// InverseSquareWeight (weights.py:157-169)
__global__ void inverse_square_weights_kernel(int M, int R_max, const int* R, int P, const double* data, int R_obs,
                                              const double* obs, double* w) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P) return;
  double om = 0.0;
  for (int r = 0; r < R_obs; ++r) om += obs[(long)r * P + t];
  om /= R_obs;
  double tot = 0.0;
  for (int m = 0; m < M; ++m) {
    const double* d = data + (long)m * R_max * P;
    double s = 0.0;
    for (int r = 0; r < R[m]; ++r) s += d[(long)r * P + t];
    const double e = s / R[m] - om;
    const double raw = 1.0 / (e * e);
    w[(long)m * P + t] = raw;
    tot += raw;
  }
  for (int m = 0; m < M; ++m) w[(long)m * P + t] /= tot;
}

// MultiModelMean (ensemble_scheme.py:108-128): mean and std**2 over every realisation of every model
__global__ void multi_model_mean_kernel(int M, int R_max, const int* R, int P, const double* data, double* mu_out,
                                        double* var_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P) return;
  double s = 0.0;
  int n = 0;
  for (int m = 0; m < M; ++m)
    for (int r = 0; r < R[m]; ++r) { s += data[((long)m * R_max + r) * P + t]; ++n; }
  const double mean = s / n;
  double q = 0.0;
  for (int m = 0; m < M; ++m)
    for (int r = 0; r < R[m]; ++r) { const double e = data[((long)m * R_max + r) * P + t] - mean; q += e * e; }
  const double sd = sqrt(q / n);
  mu_out[t] = mean;
  var_out[t] = sd * sd;
}

inline dim3 tile_grid(int n, int nb) { return dim3(n / 64, n / 64, nb); }
inline dim3 tile_grid_up(int n, int nb) { return dim3((n + 63) / 64, (n + 63) / 64, nb); }

}  // namespace

#define LAUNCH_CHECK() BAYENS_HIP(hipGetLastError())

HipPrims::~HipPrims() {
  if (xs) (void)hipFree(xs);
  if (ssq) (void)hipFree(ssq);
  if (gpart) (void)hipFree(gpart);
  for (auto& e : prof.events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto& e : prof.pool) (void)hipEventDestroy(e);
  if (ev_fork) (void)hipEventDestroy(ev_fork);
  for (hipEvent_t e : ev_join) if (e) (void)hipEventDestroy(e);
  if (side_stream && owns_side) (void)hipStreamDestroy(side_stream);
}

void HipPrims::ensure_side_stream() {
  if (!side_stream) {
    // A plain stream: on this platform a stream created with a non-default priority or with a CU mask
    // (hipStreamCreateWithPriority / hipExtStreamCreateWithCUMask) ran every kernel ~1.5x slower (measured,
    // DESIGN.md section 6); the latency-bound side of the overlap raises its wave priority instead (s_setprio).
    BAYENS_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
    BAYENS_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    for (hipEvent_t& e : ev_join) BAYENS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
}
void HipPrims::borrow_side_stream(hipStream_t s) {
  if (side_stream || !s) return;
  side_stream = s;
  owns_side = false;
  BAYENS_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  for (hipEvent_t& e : ev_join) BAYENS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
}
// The per-launch accounting (prof.enabled) wants kernels one after another on one stream: no side stream then.
void HipPrims::side_begin() {
  // (BAYENS_NO_LOOKAHEAD=1: measurement switch - the kernel-trace and counter passes of tools/make_profiles.sh run with it, so that
  //  a kernel's recorded duration is its own and not the time it shared the device with the look-ahead stream's kernel)
  static const bool no_lookahead = std::getenv("BAYENS_NO_LOOKAHEAD") != nullptr;
  if (prof.enabled || in_side || no_lookahead) return;
  ensure_side_stream();
  BAYENS_HIP(hipEventRecord(ev_fork, stream));
  BAYENS_HIP(hipStreamWaitEvent(side_stream, ev_fork, 0));
  main_stream = stream;
  stream = side_stream;
  in_side = true;
}
void HipPrims::side_end(int ch) {
  if (!in_side) return;
  BAYENS_HIP(hipEventRecord(ev_join[ch], stream));
  stream = main_stream;
  in_side = false;
  side_pending[ch] = true;
}
void HipPrims::side_join(int ch) {
  if (!side_pending[ch]) return;
  BAYENS_HIP(hipStreamWaitEvent(stream, ev_join[ch], 0));
  side_pending[ch] = false;
}

void HipPrims::mark(const char* phase) {
  if (!prof.enabled) return;
  hipEvent_t e;
  BAYENS_HIP(hipEventCreate(&e));
  BAYENS_HIP(hipEventRecord(e, stream));
  prof.marks.emplace_back(phase, e);
}

void HipPrims::ensure_scratch(int nb, int Np, int R4) {
  size_t need_xs = (size_t)nb * Np * R4 * sizeof(double);
  size_t need_s = (size_t)nb * (Np + 2) * sizeof(double);   // squared norms, then (variance, lengthscale) per member
  size_t need_g = (size_t)nb * (Np / 64) * (Np / 64) * 2 * sizeof(double);
  if (need_xs > xs_cap) { if (xs) BAYENS_HIP(hipFree(xs)); BAYENS_HIP(hipMalloc(&xs, need_xs)); xs_cap = need_xs; }
  if (need_s > ssq_cap) { if (ssq) BAYENS_HIP(hipFree(ssq)); BAYENS_HIP(hipMalloc(&ssq, need_s)); ssq_cap = need_s; }
  if (need_g > gpart_cap) { if (gpart) BAYENS_HIP(hipFree(gpart)); BAYENS_HIP(hipMalloc(&gpart, need_g)); gpart_cap = need_g; }
}

void HipPrims::potrf_leaf(int nb, Mat Ablk, real* inv, long inv_stride, int pivot_base, int* info) {
  hipLaunchKernelGGL(potrf_leaf_kernel, dim3(nb), dim3(256), 0, stream, Ablk, inv, inv_stride, pivot_base, info,
                     leaf_stamps);
  LAUNCH_CHECK();
}
#ifndef BAYENS_F32
}  // namespace BAYENS_NS
#include "potrf256.h"
namespace BAYENS_NS {
#endif
bool HipPrims::potrf_diag256(int nb, Mat Ablk, real* inv, long inv_stride, int pivot_base, int* info) {
#ifdef BAYENS_F32
  (void)nb; (void)Ablk; (void)inv; (void)inv_stride; (void)pivot_base; (void)info;
  return false;                                   // (the fp32 instance keeps the four-launch composition)
#else
  static const bool off = getenv("BAYENS_NO_POTRF256") != nullptr;      // measurement switch
  if (off) return false;
  potrf256n::potrf_diag256(stream, nb, Ablk, inv, inv_stride, pivot_base, info);
  return true;
#endif
}

void HipPrims::potrf_leaf2(int nb, Mat Ablk, real* inv, long inv_stride, int pivot_base, int* info) {
  hipLaunchKernelGGL(potrf_leaf2_kernel, dim3(nb), dim3(256), 0, stream, Ablk, inv, inv_stride, pivot_base, info);
  LAUNCH_CHECK();
}
void HipPrims::copy(int nb, int n, Mat src, Mat dst) {
  hipLaunchKernelGGL(copy_kernel, tile_grid(n, nb), dim3(256), 0, stream, n, src, dst, 0);
  LAUNCH_CHECK();
}
void HipPrims::tri_copy(int nb, int n, Mat src, Mat dst) {
  hipLaunchKernelGGL(copy_kernel, tile_grid(n, nb), dim3(256), 0, stream, n, src, dst, 1);
  LAUNCH_CHECK();
}
void HipPrims::transpose_lower(int nb, int n, Mat W, Mat Wt) {
  hipLaunchKernelGGL(transpose_lower_kernel, tile_grid(n, nb), dim3(256), 0, stream, n, W, Wt);
  LAUNCH_CHECK();
}
void HipPrims::init_state(int nb, int N, int Np, Mat Lam, real* theta1, real* mvec, double* ustate,
                          const real* y_in, const real* nu_in, real* y_pad, real* nu_pad, int* info, bool lam_identity) {
  if (lam_identity) hipLaunchKernelGGL(fill_identity_kernel, tile_grid(Np, nb), dim3(256), 0, stream, Np, Lam);
  hipLaunchKernelGGL(init_vec_kernel, dim3((Np + 255) / 256, nb), dim3(256), 0, stream, nb, N, Np, theta1, mvec,
                     ustate, y_in, nu_in, y_pad, nu_pad, info);
  LAUNCH_CHECK();
}
void HipPrims::kmat(int nb, int N, int Np, int R, const real* X, const double* ustate, int kernel, double jitter,
                    Mat Lb, bool clear_upper, Mat Mb, double m_scale, const real* m_diag, const int* Rv, const long* xoff) {
  const int R4 = ((R + 3) / 4) * 4;
  ensure_scratch(nb, Np, R4);
  hipLaunchKernelGGL(xprep_kernel, dim3((Np + 255) / 256, nb), dim3(256), 0, stream, N, Np, R, R4, X, ustate, xs, ssq, Rv, xoff);
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, tile_grid(Np, nb), dim3(256), 0, stream, N, Np, R4, xs, ssq, ustate,
                       jitter, Lb, Mat{nullptr, 0, 0}, (double*)nullptr, (const real*)nullptr, 0.0, clear_upper ? 1 : 0, Mb,
                       m_scale, m_diag);
  };
  if (kernel == 0) launch(kmat_kernel<false, 0>);
  else if (kernel == 1) launch(kmat_kernel<false, 1>);
  else launch(kmat_kernel<false, 2>);
  LAUNCH_CHECK();
}
void HipPrims::scale_rows_rsqrt(int nb, int n, Mat src, const real* nu, Mat dst) {
  hipLaunchKernelGGL(scale_rows_kernel, tile_grid(n, nb), dim3(256), 0, stream, n, src, nu, dst);
  LAUNCH_CHECK();
}
void HipPrims::lt_dvec(int nb, int n, Mat L, const real* v, const real* nu, double alpha, double beta, real* out) {
  hipLaunchKernelGGL(lt_dvec_kernel, dim3(n / 64, nb), dim3(64 * LTD_WAVES), (size_t)n * sizeof(double), stream, n, L, v, nu, alpha, beta, out);
  LAUNCH_CHECK();
}
void HipPrims::symv_lower(int nb, int n, Mat S, const real* v, real* out) {
  hipLaunchKernelGGL(symv_lower_kernel, dim3(n / 64, nb), dim3(64 * SYMV_WAVES), 0, stream, n, S, v, out);
  LAUNCH_CHECK();
}
void HipPrims::lam_update(int nb, int n, Mat G, double gamma, Mat Lam, Mat Cb, Mat Gfull, int parts) {
  hipLaunchKernelGGL(lam_update_kernel, tile_grid(n, nb), dim3(256), 0, stream, n, G, gamma, Lam, Cb, Gfull, parts);
  LAUNCH_CHECK();
}
void HipPrims::aug_set(int nb, int n, Mat rows, const real* vec, int zero_rows) {
  hipLaunchKernelGGL(aug_set_kernel, dim3((n + 255) / 256, nb), dim3(256), 0, stream, n, rows, vec, zero_rows);
  LAUNCH_CHECK();
}
void HipPrims::aug_get(int nb, int n, Mat rows, real* vec) {
  hipLaunchKernelGGL(aug_get_kernel, dim3((n + 255) / 256, nb), dim3(256), 0, stream, n, rows, vec);
  LAUNCH_CHECK();
}
void HipPrims::basis_rows_set(int nb, int N, int Np, Mat rows, const real* mu_pad, int in_padding) {
  hipLaunchKernelGGL(basis_rows_set_kernel, dim3((Np + 255) / 256, nb), dim3(256), 0, stream, N, Np, rows, mu_pad,
                     in_padding);
  LAUNCH_CHECK();
}
void HipPrims::basis_rows_get(int nb, int N, int Np, Mat rows, real* basis) {
  (void)Np;
  hipLaunchKernelGGL(basis_rows_get_kernel, dim3((N + 255) / 256, nb), dim3(256), 0, stream, N, rows, basis);
  LAUNCH_CHECK();
}
void HipPrims::loglik_rows_basis(int B, int N, int R_obs, const double* basis, const double* logdet,
                                 const double* obs, double* ll_mean, double* dots, int n_groups) {
  hipLaunchKernelGGL(basis_dots_kernel, dim3(B), dim3(256), 0, stream, N, basis, dots);
  for (int b0 = 0; b0 < B; b0 += 32768) {          // (gridDim.y is limited to 65535)
    const int nb = B - b0 < 32768 ? B - b0 : 32768;
    hipLaunchKernelGGL(loglik_basis_kernel, dim3((N + 255) / 256, nb), dim3(256), 0, stream, N, R_obs, dots, logdet, obs,
                       n_groups, b0, ll_mean);
  }
  LAUNCH_CHECK();
}
void HipPrims::resid(int nb, int n, Mat L, const real* mvec, const real* y, real* a) {
  hipLaunchKernelGGL(gemv_rows_kernel, dim3(n / 4, nb), dim3(256), 0, stream, n, L, mvec, y, a, 1);
  LAUNCH_CHECK();
}
void HipPrims::form_phi(int nb, int n, Mat GS, const real* u, const real* mvec, Mat Phi, double scale, double diag_add,
                        double u_scale) {
  hipLaunchKernelGGL(form_phi_kernel, tile_grid(n, nb), dim3(256), 0, stream, n, GS, u, mvec, Phi, scale, diag_add, u_scale);
  LAUNCH_CHECK();
}
void HipPrims::grad_adam(int nb, int N, int Np, int R, const real* X, int kernel, Mat Kbt, double* ustate, double lr,
                         const real* pvec, double sym_gamma) {
  const int R4 = ((R + 3) / 4) * 4;
  ensure_scratch(nb, Np, R4);
  // xs / ssq still hold X / lengthscale of the hypers this iteration's kmat used (ustate not yet updated)
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, tile_grid(Np, nb), dim3(256), 0, stream, N, Np, R4, xs, ssq, ustate,
                       0.0, Mat{nullptr, 0, 0}, Kbt, gpart, pvec, sym_gamma, 1, Mat{nullptr, 0, 0}, 0.0, (const real*)nullptr);
  };
  if (kernel == 0) launch(kmat_kernel<true, 0>);
  else if (kernel == 1) launch(kmat_kernel<true, 1>);
  else launch(kmat_kernel<true, 2>);
  hipLaunchKernelGGL(adam_kernel, dim3(nb), dim3(256), 0, stream, (Np / 64) * (Np / 64), gpart, ustate, lr);
  LAUNCH_CHECK();
}
void HipPrims::gemv_n(int nb, int n, Mat A, const real* x, real* y) {
  hipLaunchKernelGGL(gemv_rows_kernel, dim3(n / 4, nb), dim3(256), 0, stream, n, A, x, (const real*)nullptr, y, 0);
  LAUNCH_CHECK();
}
void HipPrims::row_sqsum(int nb, int n, Mat A, real* out) {
  hipLaunchKernelGGL(row_sqsum_kernel, dim3(n / 4, nb), dim3(256), 0, stream, n, A, out);
  LAUNCH_CHECK();
}
void HipPrims::elbo_combine(int nb, int N, int Np, const real* a, const real* f, const real* nu,
                            const real* mvec, Mat C, const double* logdetC, const double* trS, double* trace,
                            int stride, int slot) {
  hipLaunchKernelGGL(elbo_combine_kernel, dim3(nb), dim3(256), 0, stream, N, Np, a, f, nu, mvec, C, logdetC, trS,
                     trace, stride, slot);
  LAUNCH_CHECK();
}
void HipPrims::identity(int nb, int n, Mat A) {
  hipLaunchKernelGGL(fill_identity_kernel, tile_grid(n, nb), dim3(256), 0, stream, n, A);
  LAUNCH_CHECK();
}
void HipPrims::at_from_linv(int nb, int n, Mat L, Mat Linv, double jitter, Mat At) {
  hipLaunchKernelGGL(at_from_linv_kernel, tile_grid(n, nb), dim3(256), 0, stream, n, L, Linv, jitter, At);
  LAUNCH_CHECK();
}
void HipPrims::diag_init(int nb, int n, const real* nu, double add, Mat dst) {
  hipLaunchKernelGGL(diag_init_kernel, tile_grid(n, nb), dim3(256), 0, stream, n, nu, add, dst);
  LAUNCH_CHECK();
}
void HipPrims::extract(int nb, int N, Mat src, real* dst, int mode) {
  hipLaunchKernelGGL(extract_kernel, tile_grid_up(N, nb), dim3(256), 0, stream, N, src, dst, mode);
  LAUNCH_CHECK();
}
void HipPrims::extract_vec(int nb, int N, int Np, const real* src, real* dst) {
  hipLaunchKernelGGL(extract_vec_kernel, dim3((N + 255) / 256, nb), dim3(256), 0, stream, N, Np, src, dst);
  LAUNCH_CHECK();
}
void HipPrims::extract_cols(int nb, int N, Mat src, real* dst, int c0, int c1) {
  if (c1 <= c0) return;
  hipLaunchKernelGGL(extract_cols_kernel, dim3((c1 - c0 + 63) / 64, (N + 63) / 64, nb), dim3(256), 0, stream, N, src, dst, c0, c1);
  LAUNCH_CHECK();
}
void HipPrims::extract_diagonal(int nb, int N, Mat src, real* dst) {
  hipLaunchKernelGGL(extract_diagonal_kernel, dim3((N + 255) / 256, nb), dim3(256), 0, stream, N, src, dst);
  LAUNCH_CHECK();
}
void HipPrims::info_fold(int nb, int* pair, int* info) {
  hipLaunchKernelGGL(info_fold_kernel, dim3((nb + 63) / 64), dim3(64), 0, stream, nb, pair, info);
  LAUNCH_CHECK();
}
void HipPrims::logdet(int nb, int N, Mat Lc, double* out) {
  hipLaunchKernelGGL(logdet_kernel, dim3(nb), dim3(256), 0, stream, N, Lc, out);
  LAUNCH_CHECK();
}
void HipPrims::hyp_out(int nb, const double* ustate, double* hyp) {
  hipLaunchKernelGGL(hyp_out_kernel, dim3((nb + 63) / 64), dim3(64), 0, stream, nb, ustate, hyp);
  LAUNCH_CHECK();
}
void HipPrims::tri_absnorms(int nb, int n, Mat A, double* out) {
  hipLaunchKernelGGL(tri_absnorms_kernel, dim3((n + 255) / 256, 2, nb), dim3(256), 0, stream, n, A, out);
  LAUNCH_CHECK();
}
void HipPrims::frob_normalise(int nb, int n, Mat M, Mat X, const double* cap, double cap_scale) {
  ensure_scratch(nb, n, 4);
  double* norms = ssq;    // [nb] (ssq holds >= nb * n doubles)
  hipLaunchKernelGGL(tile_reduce_kernel, tile_grid(n, nb), dim3(256), 0, stream, 0, M, Mat{nullptr, 0, 0}, 0.0, 0.0, gpart);
  hipLaunchKernelGGL(tile_finish_kernel, dim3(nb), dim3(256), 0, stream, 0, (n / 64) * (n / 64), gpart, norms);
  if (cap) hipLaunchKernelGGL(cap_norms_kernel, dim3((nb + 255) / 256), dim3(256), 0, stream, nb, norms, cap, cap_scale);
  hipLaunchKernelGGL(scale_by_norm_kernel, tile_grid(n, nb), dim3(256), 0, stream, n, M, X, norms);
  LAUNCH_CHECK();
}
void HipPrims::axpby_rel(int nb, int n, double a, Mat X, double b, Mat T, double* rel) {
  ensure_scratch(nb, n, 4);
  hipLaunchKernelGGL(tile_reduce_kernel, tile_grid(n, nb), dim3(256), 0, stream, 1, X, T, a, b, gpart);
  hipLaunchKernelGGL(tile_finish_kernel, dim3(nb), dim3(256), 0, stream, 1, (n / 64) * (n / 64), gpart, rel);
  LAUNCH_CHECK();
}
void HipPrims::dot_all(int nb, int n, Mat A, Mat B, double* out) {
  ensure_scratch(nb, n, 4);
  hipLaunchKernelGGL(tile_reduce_kernel, tile_grid(n, nb), dim3(256), 0, stream, 2, A, B, 0.0, 0.0, gpart);
  hipLaunchKernelGGL(tile_finish_kernel, dim3(nb), dim3(256), 0, stream, 2, (n / 64) * (n / 64), gpart, out);
  LAUNCH_CHECK();
}
void HipPrims::read_back(const double* dev, double* host, int n) {
  BAYENS_HIP(hipMemcpyAsync(host, dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream));
  BAYENS_HIP(hipStreamSynchronize(stream));
}
void HipPrims::pair_terms(int M, int N, const double* mu, const double* sd, double* gap, double* diagterm) {
  hipLaunchKernelGGL(pair_terms_kernel, dim3(M * M), dim3(256), 0, stream, M, N, mu, sd, gap, diagterm);
  LAUNCH_CHECK();
}
void HipPrims::w2_single_finish(int M, const double* tr, const double* nuc, const double* gap, const double* diagterm,
                                double* D, double* w) {
  if (M > 1024) throw HipError{-1, "w2_single: at most 1024 members"};
  hipLaunchKernelGGL(w2_single_finish_kernel, dim3(1), dim3(256), 0, stream, M, tr, nuc, gap, diagterm, D, w);
  LAUNCH_CHECK();
}
void HipPrims::pad_in(int nb, int N, int Np, const double* src, Mat dst) {
  hipLaunchKernelGGL(pad_in_kernel, tile_grid(Np, nb), dim3(256), 0, stream, N, Np, src, dst);
  LAUNCH_CHECK();
}

void HipPrims::pad_lower(int nb, int N, int Np, const double* src, Mat dst) {
  hipLaunchKernelGGL(pad_lower_kernel, tile_grid(Np, nb), dim3(256), 0, stream, N, Np, src, dst);
  LAUNCH_CHECK();
}
void HipPrims::tri_inv64(int nb, int Np, Mat Lp, real* inv, long inv_stride, int* info) {
  hipLaunchKernelGGL(tri_inv64_kernel, dim3(Np / 64, nb), dim3(256), 0, stream, Lp, inv, inv_stride, info);
  LAUNCH_CHECK();
}
void HipPrims::solve_rows_set(int nb, int N, int Np, int nrows, int mode, const double* x, long x_stride, const double* mu,
                              Mat rows) {
  hipLaunchKernelGGL(solve_rows_set_kernel, dim3((Np + 255) / 256, 64, nb), dim3(256), 0, stream, N, Np, nrows, mode, x,
                     x_stride, mu, rows);
  LAUNCH_CHECK();
}
void HipPrims::rows_logprob(int nb, int N, int nrows, Mat rows, const double* logdet, double* out) {
  hipLaunchKernelGGL(rows_logprob_kernel, dim3(nrows, nb), dim3(256), 0, stream, N, nrows, rows, logdet, out);
  LAUNCH_CHECK();
}

void HipPrims::fill_random(double* dst, size_t count, unsigned seed) {
  hipLaunchKernelGGL(fill_random_kernel, dim3(4096), dim3(256), 0, stream, dst, count, seed);
  LAUNCH_CHECK();
}
void HipPrims::pad_vec(int nb, int N, int Np, const double* src, double* dst) {
  hipLaunchKernelGGL(pad_vec_kernel, dim3((Np + 255) / 256, nb), dim3(256), 0, stream, N, Np, src, dst);
  LAUNCH_CHECK();
}

void HipPrims::normal_loglik_rows(int B, int N, int R_obs, const double* mu, const double* scale, const double* obs,
                                  double* ll_mean) {
  hipLaunchKernelGGL(normal_loglik_kernel, dim3((N + 255) / 256, B), dim3(256), 0, stream, N, R_obs, mu, scale, obs,
                     ll_mean);
  LAUNCH_CHECK();
}
void HipPrims::weights_normalise(int M, int P, const double* ll_mean, double c_std, double* w, double* w_raw) {
  hipLaunchKernelGGL(normalise_kernel, dim3((P + 63) / 64), dim3(256), 0, stream, M, P, ll_mean, c_std, w, w_raw);
  LAUNCH_CHECK();
}
void HipPrims::w2_temporal(int M, int P, const double* mean, const double* var, double* w) {
  hipLaunchKernelGGL(w2_temporal_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, M, P, mean, var, w);
  LAUNCH_CHECK();
}
void HipPrims::field_prologue(int M, int R, int N, int G, const double* raw, double* X, double* y_mean, double* y_var) {
  hipLaunchKernelGGL(field_prologue_kernel, dim3((G + 255) / 256, N, M), dim3(256), 0, stream, R, N, G, raw, X, y_mean, y_var);
  LAUNCH_CHECK();
}
void HipPrims::transpose_rows(int nb, int A, int B, const double* src, double* dst) {
  hipLaunchKernelGGL(transpose_rows_kernel, dim3((B + 31) / 32, (A + 31) / 32, nb), dim3(256), 0, stream, A, B, src, dst);
  LAUNCH_CHECK();
}
void HipPrims::w2_spatial(int M, int T, int G, const double* mean, const double* V, double* w) {
  hipLaunchKernelGGL(w2_spatial_rows_kernel, dim3((G + 255) / 256, M), dim3(256), 0, stream, M, T, G, mean, V, w);
  hipLaunchKernelGGL(column_normalise_kernel, dim3((G + 255) / 256), dim3(256), 0, stream, M, G, w);
  LAUNCH_CHECK();
}
void HipPrims::barycentre(int M, int P, const double* w, const double* mean, const double* var, double tol,
                          double init_var, double* mu_out, double* var_out, int* n_bad) {
  BAYENS_HIP(hipMemsetAsync(n_bad, 0, sizeof(int), stream));
  hipLaunchKernelGGL(barycentre_kernel, dim3((P + 63) / 64), dim3(256), 0, stream, M, P, w, mean, var, tol,
                     init_var, mu_out, var_out, n_bad);
  LAUNCH_CHECK();
}
void HipPrims::weighted_model_mean(int M, int R_max, const int* R, int P, const double* data, const double* w,
                                   double* mu_out, double* var_out) {
  hipLaunchKernelGGL(weighted_model_mean_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, M, R_max, R, P, data, w,
                     mu_out, var_out);
  LAUNCH_CHECK();
}
void HipPrims::score_weights(int kind, int M, int P, int R_obs, const double* mean, const double* var,
                             const double* obs, double* w, double* score) {
  if (kind == 0) hipLaunchKernelGGL(crps_weights_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, M, P, R_obs, mean, var, obs, w, score);
  else hipLaunchKernelGGL(ksd_weights_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, M, P, R_obs, mean, var, obs, w, score);
  LAUNCH_CHECK();
}
void HipPrims::inverse_square_weights(int M, int R_max, const int* R, int P, const double* data, int R_obs,
                                      const double* obs, double* w) {
  hipLaunchKernelGGL(inverse_square_weights_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, M, R_max, R, P, data,
                     R_obs, obs, w);
  LAUNCH_CHECK();
}
void HipPrims::multi_model_mean(int M, int R_max, const int* R, int P, const double* data, double* mu_out,
                                double* var_out) {
  hipLaunchKernelGGL(multi_model_mean_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, M, R_max, R, P, data, mu_out,
                     var_out);
  LAUNCH_CHECK();
}
void HipPrims::extract_diag(int B, int N, const double* cov, double* var) {
  hipLaunchKernelGGL(extract_diag_kernel, dim3((N + 255) / 256, B), dim3(256), 0, stream, N, cov, var);
  LAUNCH_CHECK();
}

}  // namespace BAYENS_NS
