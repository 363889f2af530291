// gemm.hip - batched fp64 GEMM on the gfx950 matrix cores (v_mfma_f64_16x16x4_f64).
//
// This is the one heavy primitive of the GP fit: the blocked Cholesky factorisations, the triangular
// solves and the symmetric rank-N updates of SURVEY.md Appendix A are all sequenced (driver.cpp) as calls
// of this kernel, so its rate bounds the fit (compute-bound on the fp64 matrix pipe).
//
// Tiling (CDNA4: 64-wide waves, 4 SIMDs/CU, 160 KiB LDS/CU):
//   workgroup = 256 threads = 4 waves, output tile 128 x 128, each wave a 64 x 64 sub-tile
//   = 4 x 4 MFMA tiles of 16 x 16 (16 accumulators x 4 f64 = 128 VGPR);
//   K is consumed in steps of 16, both operand tiles double-buffered in LDS, the global loads of step s+1 in flight
//   while step s runs 64 MFMAs per wave.  fp64: the tiles go STRAIGHT into LDS (global_load_lds_dwordx4, no staging
//   registers, no ds_write) in per-1-KB-block swizzled images whose operand reads are conflict-free (see
//   gemm_f64_mfma's fp64 definition); fp32: staged through registers into a k-major image [16][128+2].
//   64-66.6 KB LDS and <=256 VGPR per workgroup => two workgroups per CU.
//   All dimensions are multiples of 64; waves whose 64 x 64 sub-tile lies outside the matrix (or, with
//   GEMM_TRI_OUT, strictly above the block diagonal) skip their MFMAs and stores (wave-uniform).
//   Block ids are dealt so that each XCD (own L2) walks a contiguous range of tiles of few members.
#include <cstdlib>
#include "hip_prims.h"

namespace BAYENS_NS {

typedef real v4d __attribute__((ext_vector_type(4)));   // one 16x16 accumulator tile per lane (fp64 or fp32 `real`)
typedef real v2d __attribute__((ext_vector_type(2)));

// The 16x16x4 MFMA exists for both precisions with the same A/B operand layout (lane l: A[l&15][l>>4],
// B[l>>4][l&15]); the fp32 form issues in 32 cycles instead of 64.  Only the accumulator layout differs:
// fp64 keeps rows (l>>4) + 4r in register r, fp32 rows 4(l>>4) + r.
#ifdef BAYENS_F32
#define gemm_f64_mfma gemm_f32_mfma   /* the kernel symbol of the fp32 instance: bayens_f32::gemm_f32_mfma<TA,TB> */
#define BAYENS_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
__device__ __forceinline__ int acc_row(int lk, int r) { return 4 * lk + r; }
#else
#define BAYENS_MFMA16(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
__device__ __forceinline__ int acc_row(int lk, int r) { return lk + 4 * r; }
#endif

namespace {

// One K step stages 8 staging vectors of k per operand row.  VEC = 2 in both precisions (K step 16).  Measured for
// fp32 with VEC = 4 (16-byte loads, K step 32, row pad 4): 84 TFLOP/s against 91 with VEC = 2 - with 4-byte
// elements the k-major image is read with 4-way bank conflicts at either pad and the wider step only lengthens the
// exposed prologue of the short products; the fp32 instance needs operand-native LDS images, not a
// wider step.
constexpr int VEC = 2;                               // elements per staging vector
constexpr int BM = 128, BN = 128, BK = 8 * VEC;
// k-major LDS row stride: pad 2 in both precisions.  (fp32, measured: pad 16 - conflict-free operand reads, 4-way
// transposing stores - 87 TFLOP/s against 91 with pad 2; K step 32 with pad 16: 78.)
constexpr int LDT = BM + 2;
typedef real vst __attribute__((ext_vector_type(VEC)));

struct GemmArgs {
  const real* A;
  const real* B;
  real* C;
  long sA, sB, sC;
  int lda, ldb, ldc;
  int m, n, k;
  double alpha, beta, diag_add;
  int flags;
  int tiles_m, tiles_n, total, nb;
};

// Stage one 128 x BK operand tile into registers.  TR=true: memory is [row][k] (k contiguous) -> each thread
// takes 4 x (VEC consecutive k) of rows r, r+32, r+64, r+96; TR=false: memory is [k][row] -> each thread takes
// 4 x (VEC consecutive rows) of 4 k-rows.
template <bool TR>
__device__ __forceinline__ void stage_load(const real* base, int ld, int row0, int nrows,
                                           int k0, int tid, vst (&reg)[4]) {
  if (TR) {
    const int kp = tid & 7, r = tid >> 3;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int rr = row0 + r + 32 * p;
      rr = rr < nrows ? rr : nrows - 1;
      reg[p] = *reinterpret_cast<const vst*>(base + (long)rr * ld + k0 + VEC * kp);
    }
  } else {
    constexpr int CW = BM / VEC;                 // threads per k-row
    const int c = tid % CW, kr = tid / CW;
    int cc = row0 + VEC * c;
    cc = cc <= nrows - VEC ? cc : nrows - VEC;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      reg[p] = *reinterpret_cast<const vst*>(base + (long)(k0 + kr + (256 / CW) * p) * ld + cc);
  }
}

template <bool TR>
__device__ __forceinline__ void stage_store(real (*tile)[LDT], int tid, const vst (&reg)[4]) {
  if (TR) {
    const int kp = tid & 7, r = tid >> 3;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int e = 0; e < VEC; ++e) tile[VEC * kp + e][r + 32 * p] = reg[p][e];
  } else {
    constexpr int CW = BM / VEC;
    const int c = tid % CW, kr = tid / CW;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<vst*>(&tile[kr + (256 / CW) * p][VEC * c]) = reg[p];
  }
}

#ifdef BAYENS_F32
// The fp32 instance stages its operand tiles through registers (the fp64 instance below loads them straight into LDS).
// Register budget: two workgroups per CU (<=256 VGPR).  The fp32 instance needs 135 and runs three per CU; holding
// it to 128 VGPR for four per CU was measured slower (85 against 91 TFLOP/s).
template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_f64_mfma(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) real As[2][BK][LDT];
  __shared__ __attribute__((aligned(16))) real Bs[2][BK][LDT];

  // XCD-aware dealing: consecutive hardware block ids go round-robin over the 8 XCDs; give every XCD a
  // contiguous chunk of the (member, tile) space so that its L2 sees few members' panels.
  int bid = blockIdx.x;
  const int tiles = g.tiles_m * g.tiles_n;
  int batch, t;
  if ((g.flags & GEMM_TRI_OUT) && (g.nb & 7) == 0) {
    // triangular launches have unequal tiles (skipped ones, k-trimmed ones): inside an XCD's share walk the
    // tiles of all its members together (tile-major, member-minor) so that every member's long tiles start first
    const int mpc = g.nb >> 3, loc = bid >> 3;
    t = loc / mpc;
    batch = (bid & 7) * mpc + (loc - t * mpc);
  } else {
    if ((g.total & 7) == 0) bid = (bid & 7) * (g.total >> 3) + (bid >> 3);
    batch = bid / tiles;
    t = bid - batch * tiles;
  }
  const int tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
  const int row0 = tm * BM, col0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  if ((g.flags & GEMM_TRI_OUT) && (g.flags & GEMM_ZERO_SKIPPED)) {
    // 64x64 blocks strictly above the block diagonal are not computed but must read as zeros later
    const int r64 = row0 + wm * 64, c64 = col0 + wn * 64;
    if (c64 > r64 && r64 < g.m && c64 < g.n) {
      real* Cz = g.C + batch * g.sC + (long)r64 * g.ldc + c64;
      for (int e = lane; e < 64 * 64; e += 64) Cz[(long)(e >> 6) * g.ldc + (e & 63)] = 0.0;
    }
  }
  if ((g.flags & GEMM_TRI_OUT) && col0 > row0) return;   // whole tile above the block diagonal
  const real* A = g.A + batch * g.sA;
  const real* B = g.B + batch * g.sB;
  real* C = g.C + batch * g.sC;

  bool active = (row0 + wm * 64 < g.m) && (col0 + wn * 64 < g.n);
  if ((g.flags & GEMM_TRI_OUT) && (col0 + wn * 64 > row0 + wm * 64)) active = false;

  int kbeg = 0;
  if (g.flags & GEMM_KTRIM_TN_LOWER) kbeg = row0 > col0 ? row0 : col0;
  int kend = g.k;
  if ((g.flags & GEMM_KTRIM_A_LOWER) && row0 + BM < kend) kend = row0 + BM;   // A_ik = 0 for k > i: nothing beyond the tile's rows
  const int nk = (kend - kbeg) / BK;

  v4d acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = v4d{0, 0, 0, 0};

  vst ra[4], rb[4];
  if (nk > 0) {
    stage_load<!TA>(A, g.lda, row0, g.m, kbeg, tid, ra);
    stage_load<TB>(B, g.ldb, col0, g.n, kbeg, tid, rb);
    stage_store<!TA>(As[0], tid, ra);
    stage_store<TB>(Bs[0], tid, rb);
  }
  __syncthreads();

  const int lr = lane & 15, lk = lane >> 4;
  for (int s = 0; s < nk; ++s) {
    const int buf = s & 1;
    if (s + 1 < nk) {
      stage_load<!TA>(A, g.lda, row0, g.m, kbeg + (s + 1) * BK, tid, ra);
      stage_load<TB>(B, g.ldb, col0, g.n, kbeg + (s + 1) * BK, tid, rb);
    }
    if (active) {
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        real a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = As[buf][kk * 4 + lk][wm * 64 + i * 16 + lr];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Bs[buf][kk * 4 + lk][wn * 64 + j * 16 + lr];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = BAYENS_MFMA16(a[i], b[j], acc[i][j]);
      }
    }
    if (s + 1 < nk) {
      stage_store<!TA>(As[buf ^ 1], tid, ra);
      stage_store<TB>(Bs[buf ^ 1], tid, rb);
    }
    __syncthreads();
  }

  if (!active) return;
  // epilogue: lane holds rows (lane>>4)+4r, column lane&15 of each 16x16 tile.  The C values of one 16-row
  // strip (16 per lane) are all loaded before the first store of that strip: with a runtime ldc the compiler
  // cannot reorder a load past an earlier store, and one exposed memory round trip per element (64 per lane)
  // cost ~27 us per tile.  The strips are software-pipelined: the loads of strip i+1 are issued before the stores
  // of strip i, because a load queued behind stores waits for them too (one in-order memory counter).
  real cv[2][4][4];
  auto load_strip = [&](int i, real (&dst)[4][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dst[j][r] = C[(long)(row0 + wm * 64 + i * 16 + acc_row(lk, r)) * g.ldc + col0 + wn * 64 + j * 16 + lr];
  };
  const bool accumulate = g.beta != 0.0;
  if (accumulate) load_strip(0, cv[0]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (accumulate && i + 1 < 4) load_strip(i + 1, cv[(i + 1) & 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + wn * 64 + j * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wm * 64 + i * 16 + acc_row(lk, r);
        double v = g.alpha * acc[i][j][r];
        if (accumulate) v += g.beta * cv[i & 1][j][r];
        if (row == col) v += g.diag_add;
        C[(long)row * g.ldc + col] = v;
      }
    }
  }
}

#else   // fp64
// ------------------------------------------------------------------------------------------------
// fp64: the operand tiles are loaded STRAIGHT INTO LDS (global_load_lds_dwordx4: lane l's 16 bytes land
// at M0 + 16 l, no staging registers, no ds_write), fp64 only.  A wave instruction fills 1 KB of LDS, so padding is
// not available and the images are laid out per 1 KB block such that the MFMA operand reads (ds_read_b64: the two
// 32-lane halves are banked separately over 32 eight-byte banks) are conflict-free and every kk / sub-tile offset
// is an immediate:
//  * operand stored k-contiguous in memory (A of NN/NT, B of NT/TT) - "row image": block rb8 = 8 rows x 16 k,
//    element (row, k) at  rb8*128 + 16*((k>>1) ^ (rb8&1)) + 2*(row&7) + (k&1)   (doubles).  Lane j of the block's
//    load fetches row rb8*8 + (j&7), k-chunk (j>>3) ^ (rb8&1): eight lanes cover each 128-byte line.
//  * operand stored row-contiguous in memory (A of TN/TT, B of NN/TN) - "k image": block = one k row of 128,
//    element (k, row) at  k*128 + (row ^ 16*(k&1)).  Lane j fetches the two rows 2*(j ^ 8*(k&1)), +1: one 1 KB line run.
// MFMA operand of lane (lr = lane&15, lk = lane>>4), sub-tile i, step kk: row/col = w*64 + i*16 + lr, k = 4 kk + lk.
// One wave instruction: every lane's 16 bytes at `src` go to LDS at `dst_wave_base` + 16 * lane.  Issued as inline
// assembly: through the builtin the compiler must assume that the write may alias the LDS reads that follow and puts an
// s_waitcnt vmcnt(0) in front of them - the K loop then waits out the global latency of the NEXT tile before it touches
// the current one (measured: 59.7 instead of 68.5 TFLOP/s).  The caller waits (dl_wait) before the step's barrier.
__device__ __forceinline__ void lds_dma16(const double* src, const double* dst_wave_base) {
  const unsigned lds_off = (unsigned)(size_t)dst_wave_base;      // low half of a generic LDS address = the LDS offset
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_off) : "memory", "m0");
}
// before a step's barrier: this wave's DMA loads have landed AND its own operand reads of the buffer that the next DMA
// overwrites have returned (they were issued 48 MFMAs earlier; the wait is free, the guarantee is not optional)
__device__ __forceinline__ void dl_wait() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// one of the four wave instructions of a tile's load (p = 0 .. 3)
template <bool KIMG>
__device__ __forceinline__ void dl_issue_part(const double* base, int ld, int row0, int nrows, int k0, int lane, int wave,
                                              double* tile, int p) {
  if (!KIMG) {
    const int rb8 = 4 * p + wave;                         // blocks of this wave: wave, 4 + wave, 8 + wave, 12 + wave
    int rr = row0 + rb8 * 8 + (lane & 7);
    rr = rr < nrows ? rr : nrows - 1;
    const int c8 = (lane >> 3) ^ (rb8 & 1);
    lds_dma16(base + (long)rr * ld + k0 + 2 * c8, tile + rb8 * 128);
  } else {
    const int k = 4 * p + wave;
    int cc = row0 + 2 * (lane ^ (8 * (k & 1)));
    cc = cc <= nrows - 2 ? cc : nrows - 2;
    lds_dma16(base + (long)(k0 + k) * ld + cc, tile + k * 128);
  }
}
template <bool KIMG>
__device__ __forceinline__ void dl_issue(const double* base, int ld, int row0, int nrows, int k0, int lane, int wave,
                                         double* tile) {
#pragma unroll
  for (int p = 0; p < 4; ++p) dl_issue_part<KIMG>(base, ld, row0, nrows, k0, lane, wave, tile, p);
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_f64_mfma(GemmArgs g) {
  __shared__ __attribute__((aligned(1024))) double As[2][BK * BM];
  __shared__ __attribute__((aligned(1024))) double Bs[2][BK * BN];

  int bid = blockIdx.x;
  const int tiles = g.tiles_m * g.tiles_n;
  int batch, t;
  if ((g.flags & GEMM_TRI_OUT) && (g.nb & 7) == 0) {
    const int mpc = g.nb >> 3, loc = bid >> 3;
    t = loc / mpc;
    batch = (bid & 7) * mpc + (loc - t * mpc);
  } else {
    if ((g.total & 7) == 0) bid = (bid & 7) * (g.total >> 3) + (bid >> 3);
    batch = bid / tiles;
    t = bid - batch * tiles;
  }
  const int tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
  const int row0 = tm * BM, col0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  if ((g.flags & GEMM_TRI_OUT) && (g.flags & GEMM_ZERO_SKIPPED)) {
    const int r64 = row0 + wm * 64, c64 = col0 + wn * 64;
    if (c64 > r64 && r64 < g.m && c64 < g.n) {
      double* Cz = g.C + batch * g.sC + (long)r64 * g.ldc + c64;
      for (int e = lane; e < 64 * 64; e += 64) Cz[(long)(e >> 6) * g.ldc + (e & 63)] = 0.0;
    }
  }
  if ((g.flags & GEMM_TRI_OUT) && col0 > row0) return;
  const double* A = g.A + batch * g.sA;
  const double* B = g.B + batch * g.sB;
  double* C = g.C + batch * g.sC;

  bool active = (row0 + wm * 64 < g.m) && (col0 + wn * 64 < g.n);
  if ((g.flags & GEMM_TRI_OUT) && (col0 + wn * 64 > row0 + wm * 64)) active = false;

  int kbeg = 0;
  if (g.flags & GEMM_KTRIM_TN_LOWER) kbeg = row0 > col0 ? row0 : col0;
  int kend = g.k;
  if ((g.flags & GEMM_KTRIM_A_LOWER) && row0 + BM < kend) kend = row0 + BM;   // A_ik = 0 for k > i: nothing beyond the tile's rows
  const int nk = (kend - kbeg) / BK;

  v4d acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = v4d{0, 0, 0, 0};

  const int lr = lane & 15, lk = lane >> 4;
  // lane parts of the operand addresses (doubles); the sub-tile and kk parts are immediates
  //   row image: w*1024 + i*256 + kk*32 + [ (lr>>3)*128 + 16*((lk>>1) ^ (lr>>3)) + 2*(lr&7) + (lk&1) ]
  //   k image:   kk*512 + lk*128 + w*64 + lr + 16*((i ^ (lk&1)))        (i even: +16 s, i odd: -16 s, s = lk&1)
  const int row_part = (lr >> 3) * 128 + 16 * ((lk >> 1) ^ (lr >> 3)) + 2 * (lr & 7) + (lk & 1);
  const int k_part = lk * 128 + lr, ks = 16 * (lk & 1);
  const int a_lane = TA ? k_part + wm * 64 : row_part + wm * 1024;
  const int b_lane = TB ? row_part + wn * 1024 : k_part + wn * 64;

  // Operand reads run half a step ahead of their MFMAs, ACROSS the step's barrier: the last quarter of a step's MFMAs
  // (kk = 3, operands already in registers) is issued after the barrier, under the first operand reads of the next tile -
  // otherwise every step starts with a wave that has nothing to issue until its first ds_read returns.
  auto rd = [&](int buf, int kk, double (&a)[4], double (&b)[4]) {
    const double* pa = As[buf] + a_lane;
    const double* pb = Bs[buf] + b_lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = TA ? pa[kk * 512 + i * 16 + ((i & 1) ? -ks : ks)] : pa[i * 256 + kk * 32];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = TB ? pb[j * 256 + kk * 32] : pb[kk * 512 + j * 16 + ((j & 1) ? -ks : ks)];
  };
  auto mm = [&](const double (&a)[4], const double (&b)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = BAYENS_MFMA16(a[i], b[j], acc[i][j]);
  };

  // the first strip of C (beta != 0) is requested BEFORE the K loop: nobody else writes this tile during the launch, and the
  // epilogue then starts with its values in registers instead of with a memory round trip
  real cv[2][4][4];
  auto load_strip = [&](int i, real (&dst)[4][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dst[j][r] = C[(long)(row0 + wm * 64 + i * 16 + acc_row(lk, r)) * g.ldc + col0 + wn * 64 + j * 16 + lr];
  };
  const bool accumulate = g.beta != 0.0;
  // both buffers are free at the start: the first TWO tiles are requested together (the second used to be requested when the
  // first had landed, i.e. one memory round trip later - and the first step of a short product then ended waiting for it); the
  // first wait lets the second tile's eight loads stay in flight (loads return in the order they were issued)
  if (nk > 0) {
    dl_issue<TA>(A, g.lda, row0, g.m, kbeg, lane, wave, As[0]);
    dl_issue<!TB>(B, g.ldb, col0, g.n, kbeg, lane, wave, Bs[0]);
  }
  if (nk > 1) {
    dl_issue<TA>(A, g.lda, row0, g.m, kbeg + BK, lane, wave, As[1]);
    dl_issue<!TB>(B, g.ldb, col0, g.n, kbeg + BK, lane, wave, Bs[1]);
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
  } else {
    dl_wait();
  }
  __syncthreads();
  if (accumulate && active) load_strip(0, cv[0]);
  if (active) {
    double a0[4], b0[4], a1[4], b1[4], a2[4], b2[4], a3[4], b3[4];
    rd(0, 0, a0, b0);
    rd(0, 1, a1, b1);
    for (int s = 0; s < nk; ++s) {
      const int buf = s & 1;
      {
        const double* pa = As[buf] + a_lane;
        const double* pb = Bs[buf] + b_lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {        // the step's first sixteen MFMAs, one operand's four reads of k groups 2 / 3 behind every row
                                             // (as a burst of sixteen reads behind the first MFMA: 54.8 against 54.65 ms per step)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = BAYENS_MFMA16(a0[i], b0[j], acc[i][j]);
          if (i == 0) { for (int q = 0; q < 4; ++q) a2[q] = TA ? pa[2 * 512 + q * 16 + ((q & 1) ? -ks : ks)] : pa[q * 256 + 2 * 32]; }
          if (i == 1) { for (int q = 0; q < 4; ++q) b2[q] = TB ? pb[q * 256 + 2 * 32] : pb[2 * 512 + q * 16 + ((q & 1) ? -ks : ks)]; }
          if (i == 2) { for (int q = 0; q < 4; ++q) a3[q] = TA ? pa[3 * 512 + q * 16 + ((q & 1) ? -ks : ks)] : pa[q * 256 + 3 * 32]; }
          if (i == 3) { for (int q = 0; q < 4; ++q) b3[q] = TB ? pb[q * 256 + 3 * 32] : pb[3 * 512 + q * 16 + ((q & 1) ? -ks : ks)]; }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      mm(a1, b1);
      mm(a2, b2);
      __builtin_amdgcn_sched_barrier(0);
      dl_wait();                       // tile s + 1 has landed
      __syncthreads();                 // ... everywhere, and every wave has read the last of tile s
      __builtin_amdgcn_sched_barrier(0);
      // The step's last sixteen MFMAs (operands in registers since before the barrier) start at once; the next tile's first
      // operand reads go behind their first two rows and the eight load instructions of tile s + 2 behind each row.  (Issued in
      // one block in front of them, the loads - ~45 instructions of address arithmetic - and the reads were a stretch with no MFMA
      // of this wave in the pipe: 55.8 -> 55.2 ms per 32 x 2000 step with the loads between the rows, 55.0 -> 54.8 with the reads there
      // too; NT 2048 x 1024 x 1024 66.7 -> 69.1 TFLOP/s over the two.)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = BAYENS_MFMA16(a3[i], b3[j], acc[i][j]);
        if (i == 0) rd(buf ^ 1, 0, a0, b0);        // (after the last step: whatever the other buffer holds - never used)
        if (i == 1) rd(buf ^ 1, 1, a1, b1);
        if (s + 2 < nk) {
          dl_issue_part<TA>(A, g.lda, row0, g.m, kbeg + (s + 2) * BK, lane, wave, As[buf], i);
          dl_issue_part<!TB>(B, g.ldb, col0, g.n, kbeg + (s + 2) * BK, lane, wave, Bs[buf], i);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    for (int s = 0; s < nk; ++s) {     // a wave with no output block still moves its share of the tiles
      dl_wait();
      __syncthreads();
      if (s + 2 < nk) {
        dl_issue<TA>(A, g.lda, row0, g.m, kbeg + (s + 2) * BK, lane, wave, As[s & 1]);
        dl_issue<!TB>(B, g.ldb, col0, g.n, kbeg + (s + 2) * BK, lane, wave, Bs[s & 1]);
      }
    }
  }

  if (!active) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (accumulate && i + 1 < 4) load_strip(i + 1, cv[(i + 1) & 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + wn * 64 + j * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wm * 64 + i * 16 + acc_row(lk, r);
        double v = g.alpha * acc[i][j][r];
        if (accumulate) v += g.beta * cv[i & 1][j][r];
        if (row == col) v += g.diag_add;
        C[(long)row * g.ldc + col] = v;
      }
    }
  }
}
#endif

// ------------------------------------------------------------------------------------------------
// One-shot kernel for the narrow products of the blocked solves: C[m x 64] = alpha A[m x 64] op(B)[64 x 64] + beta C.
// ~80 % of the GEMM launches of a fit have this shape (multiplication by an inverted 64x64 diagonal block and
// the first-level update behind it); they carry ~2 % of the flop and are bound by HBM traffic and launch
// latency, so there is no K pipeline here: a workgroup takes 64 rows, stages its 64x64 A tile and the 64x64
// B block in LDS once, each wave multiplies 16 rows (64 fp64 MFMAs) and the tile is written back.  Reading the
// whole A tile before the first store also makes the in-place form (C aliases A) safe.
// ------------------------------------------------------------------------------------------------
constexpr int SK = 64, SLD = 66;

template <bool TB>
__global__ __launch_bounds__(256) void gemm_panel64(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) real As[SK][SLD];   // [row][k]
  __shared__ __attribute__((aligned(16))) real Bs[SK][SLD];   // [k][col]
  const int tiles_m = g.m / SK;
  const int batch = blockIdx.x / tiles_m;
  const int row0 = (blockIdx.x - batch * tiles_m) * SK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const real* A = g.A + batch * g.sA + (long)row0 * g.lda;
  const real* B = g.B + batch * g.sB;
  real* C = g.C + batch * g.sC + (long)row0 * g.ldc;
  {
    const int c2 = tid & 31, r = tid >> 5;
    v2d va[8], vb[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      va[p] = *reinterpret_cast<const v2d*>(A + (long)(r + 8 * p) * g.lda + 2 * c2);
      vb[p] = *reinterpret_cast<const v2d*>(B + (long)(r + 8 * p) * g.ldb + 2 * c2);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      *reinterpret_cast<v2d*>(&As[r + 8 * p][2 * c2]) = va[p];
      if (TB) {   // memory is [col][k]: transpose into the k-major image
        Bs[2 * c2][r + 8 * p] = vb[p].x;
        Bs[2 * c2 + 1][r + 8 * p] = vb[p].y;
      } else {
        *reinterpret_cast<v2d*>(&Bs[r + 8 * p][2 * c2]) = vb[p];
      }
    }
  }
  __syncthreads();
  const int lr = lane & 15, lk = lane >> 4;
  v4d acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = v4d{0, 0, 0, 0};
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const real a = As[wave * 16 + lr][kk * 4 + lk];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const real b = Bs[kk * 4 + lk][j * 16 + lr];
      acc[j] = BAYENS_MFMA16(a, b, acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = j * 16 + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + acc_row(lk, r);
      real* cp = C + (long)row * g.ldc + col;
      double v = g.alpha * acc[j][r];
      if (g.beta != 0.0) v += g.beta * (*cp);
      if (row0 + row == col) v += g.diag_add;
      *cp = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Small outputs: 64 x 64 tiles.  The last block columns of a factorisation and the inner levels of the blocked solves
// multiply into outputs of a few 128 x 128 tiles per member - 96 to 500 workgroups on 256 CUs, each a serial chain of K steps
// (profiles/r05_gemm_shapes.txt: 11-35 TFLOP/s below m = 768).  For those the same product runs on four times as many
// workgroups of a quarter of the work: a workgroup owns a 64 x 64 tile, wave w its rows 16 w .. 16 w + 15 (four MFMA tiles
// across), K in chunks of 32 staged through registers into padded LDS images (the next chunk's global loads fly under the
// current chunk's 32 MFMAs per wave).  Same flags, same arithmetic order along k as gemm_f64_mfma.
// ------------------------------------------------------------------------------------------------
constexpr int TK = 32, TLA = TK + 2, TLB = 64 + 2;

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_tile64(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) real As[64][TLA];   // [row][k]
  // [k][col]; with B stored n x k the chunk stays as it lies, [col][k], and the operand read swaps its indices (a 2-way conflicted
  // read instead of a transposing store, element by element, into the k-major image)
  __shared__ __attribute__((aligned(16))) real Bs[TB ? 64 : TK][TB ? TLA : TLB];
  const int tiles_m = g.m / 64, tiles_n = g.n / 64;
  const int tiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  const int total = tiles * g.nb;
  if ((total & 7) == 0) bid = (bid & 7) * (total >> 3) + (bid >> 3);      // an XCD walks a contiguous range of (member, tile)
  const int batch = bid / tiles;
  const int t = bid - batch * tiles;
  const int tm = t / tiles_n, tn = t - tm * tiles_n;
  const int row0 = tm * 64, col0 = tn * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  real* C = g.C + batch * g.sC + (long)row0 * g.ldc + col0;
  if ((g.flags & GEMM_TRI_OUT) && col0 > row0) {
    if (g.flags & GEMM_ZERO_SKIPPED)
      for (int e = tid; e < 64 * 64; e += 256) C[(long)(e >> 6) * g.ldc + (e & 63)] = 0.0;
    return;
  }
  const real* A = g.A + batch * g.sA;
  const real* B = g.B + batch * g.sB;
  int kbeg = 0, kend = g.k;
  if (g.flags & GEMM_KTRIM_TN_LOWER) kbeg = row0 > col0 ? row0 : col0;
  if ((g.flags & GEMM_KTRIM_A_LOWER) && row0 + 64 < kend) kend = row0 + 64;
  const int nchunk = kend > kbeg ? (kend - kbeg) / TK : 0;

  // the next chunk's operands are requested before the current chunk's MFMAs, and the tile of C before the first chunk (a second
  // chunk in flight was measured: no faster - these launches are bound by their fixed costs, not by the K loop)
  v2d ra[4], rb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (!TA) ra[p] = *reinterpret_cast<const v2d*>(A + (long)(row0 + (tid >> 4) + 16 * p) * g.lda + k0 + 2 * (tid & 15));
      else ra[p] = *reinterpret_cast<const v2d*>(A + (long)(k0 + (tid >> 5) + 8 * p) * g.lda + row0 + 2 * (tid & 31));
      if (!TB) rb[p] = *reinterpret_cast<const v2d*>(B + (long)(k0 + (tid >> 5) + 8 * p) * g.ldb + col0 + 2 * (tid & 31));
      else rb[p] = *reinterpret_cast<const v2d*>(B + (long)(col0 + (tid >> 4) + 16 * p) * g.ldb + k0 + 2 * (tid & 15));
    }
  };
  auto place = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (!TA) {
        *reinterpret_cast<v2d*>(&As[(tid >> 4) + 16 * p][2 * (tid & 15)]) = ra[p];
      } else {
        As[2 * (tid & 31)][(tid >> 5) + 8 * p] = ra[p].x;
        As[2 * (tid & 31) + 1][(tid >> 5) + 8 * p] = ra[p].y;
      }
      if (!TB) {
        *reinterpret_cast<v2d*>(&Bs[(tid >> 5) + 8 * p][2 * (tid & 31)]) = rb[p];
      } else {
        *reinterpret_cast<v2d*>(&Bs[(tid >> 4) + 16 * p][2 * (tid & 15)]) = rb[p];
      }
    }
  };
  v4d acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = v4d{0, 0, 0, 0};
  if (nchunk > 0) fetch(kbeg);
  const bool accumulate = g.beta != 0.0;
  real cv[4][4];
  if (accumulate) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cv[j][r] = C[(long)(wave * 16 + acc_row(lk, r)) * g.ldc + j * 16 + lr];
  }
  for (int c = 0; c < nchunk; ++c) {
    BAYENS_LDS_BARRIER();                    // the previous chunk's operand reads are done
    place();
    BAYENS_LDS_BARRIER();
    if (c + 1 < nchunk) fetch(kbeg + (c + 1) * TK);
#pragma unroll
    for (int kk = 0; kk < TK / 4; ++kk) {
      const real a = As[wave * 16 + lr][kk * 4 + lk];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] = BAYENS_MFMA16(a, TB ? Bs[j * 16 + lr][kk * 4 + lk] : Bs[kk * 4 + lk][j * 16 + lr], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + acc_row(lk, r), col = j * 16 + lr;
      double v = g.alpha * acc[j][r];
      if (accumulate) v += g.beta * cv[j][r];
      if (row0 + row == col0 + col) v += g.diag_add;
      C[(long)row * g.ldc + col] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fused triangular solve of a narrow diagonal block (n = 64 * nsub <= 256): B <- B L^-T (RLN=false) or
// B <- B L^-1 (RLN=true).  Rows of a right-side solve are independent, so a workgroup owns 64 rows and walks
// the column blocks itself:   X_j = (B_j - sum_k X_k op(L_jk)) op(M_j)   with M_j the inverted diagonal block.
// Every product is a 64x64x64 fp64 MFMA product (16 rows per wave); the accumulator becomes the A operand of
// the product with M_j through the wave's own rows of the LDS tile, finished blocks X_k come back from L2.
// One launch and one HBM pass over the panel replace the 2*nsub-1 launches of the recursion's bottom levels.
// ------------------------------------------------------------------------------------------------
struct PanelArgs {
  const real* L;     // n x n lower block
  const real* inv;   // inverted 64x64 diagonal blocks of that block, [nsub][64][64]
  real* B;           // m x n panel, in place
  long sL, sInv, sB;
  int ldl, ldb, m, nsub;
};

template <bool TR>
__device__ __forceinline__ void stage64_load(const real* src, int ld, int tid, v2d (&reg)[8]) {
  const int c2 = tid & 31, r = tid >> 5;
#pragma unroll
  for (int p = 0; p < 8; ++p) reg[p] = *reinterpret_cast<const v2d*>(src + (long)(r + 8 * p) * ld + 2 * c2);
}
// TR=false: tile[row][col] = src[row][col]; TR=true: tile[col][row] = src[row][col]
template <bool TR>
__device__ __forceinline__ void stage64_store(real (*tile)[SLD], int tid, const v2d (&reg)[8]) {
  const int c2 = tid & 31, r = tid >> 5;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    if (TR) {
      tile[2 * c2][r + 8 * p] = reg[p].x;
      tile[2 * c2 + 1][r + 8 * p] = reg[p].y;
    } else {
      *reinterpret_cast<v2d*>(&tile[r + 8 * p][2 * c2]) = reg[p];
    }
  }
}

// acc[jt] += As[wave rows][:] * Bs[:][16 jt ..] (Bs the k-major image B[k][col]) for the product with the inverted
// diagonal block: op(M_j) is triangular, so the 4x16 sub-tiles of Bs that are entirely zero are skipped (40 of 64
// MFMAs remain).  UPPER: Bs[k][col] != 0 only for k <= col.
// BT: Bs holds the transpose of the B operand (Bs[col][k]) and is read with swapped indices - a 2-way-conflicted read
// instead of the 4-way-conflicted scalar transposing store that would build the k-major image.
template <bool UPPER, bool BT>
__device__ __forceinline__ void panel_mma_tri(v4d (&acc)[4], const real (*As)[SLD], const real (*Bs)[SLD], int wave,
                                              int lane) {
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const real a = As[wave * 16 + lr][kk * 4 + lk];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool live = UPPER ? (kk < 4 * j + 4) : (kk >= 4 * j);
      if (live) acc[j] = BAYENS_MFMA16(a, BT ? Bs[j * 16 + lr][kk * 4 + lk] : Bs[kk * 4 + lk][j * 16 + lr], acc[j]);
    }
  }
}

template <bool RLN>
__global__ __launch_bounds__(256, 2) void trsm_panel_kernel(PanelArgs g) {
  // As: each wave's own 16 rows, only used to turn an accumulator (C layout) into an A operand; Bs: the staged
  // L block or inverted diagonal block, shared by the four waves.  Finished blocks X_k stay in registers as
  // (negated) A fragments, 16 doubles per lane and block, so a product only stages its L block; the next block's
  // global load is issued before the current product's MFMAs.
  __shared__ __attribute__((aligned(16))) real As[SK][SLD];
  __shared__ __attribute__((aligned(16))) real Bs[SK][SLD];
  const int tiles_m = g.m / SK;
  const int batch = blockIdx.x / tiles_m;
  const int row0 = (blockIdx.x - batch * tiles_m) * SK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const real* L = g.L + batch * g.sL;
  const real* inv = g.inv + batch * g.sInv;
  real* B = g.B + batch * g.sB + (long)row0 * g.ldb;
  const int nsub = g.nsub;

  // tile (s, t): t < s the L block coupling finished block t to block s (in solve order), t == s the inverse
  auto tile_load = [&](int s, int t, v2d (&reg)[8]) {
    const int j = RLN ? nsub - 1 - s : s;
    if (t == s) {
      stage64_load<false>(inv + (long)j * 4096, 64, tid, reg);
    } else {
      const int k = RLN ? nsub - 1 - t : t;
      // RLT: B-operand[kk][jj] = L[64j+jj][64k+kk] (transposed on the way into Bs); RLN: L[64k+kk][64j+jj]
      if (RLN) stage64_load<false>(L + (long)(64 * k) * g.ldl + 64 * j, g.ldl, tid, reg);
      else stage64_load<false>(L + (long)(64 * j) * g.ldl + 64 * k, g.ldl, tid, reg);
    }
  };

  real xf[3][16];   // -X_k as A fragments: xf[t][kk] = -X_t[wave rows lr][4 kk + lk]
  v2d rb[8];
  tile_load(0, 0, rb);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s < nsub) {
      const int j = RLN ? nsub - 1 - s : s;
      v4d acc[4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jt][r] = B[(long)(wave * 16 + acc_row(lk, r)) * g.ldb + 64 * j + jt * 16 + lr];
#pragma unroll
      for (int t = 0; t <= s; ++t) {
        BAYENS_LDS_BARRIER();                       // the previous product is done with Bs
        if (t == s) {
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) As[wave * 16 + acc_row(lk, r)][jt * 16 + lr] = acc[jt][r];
        }
        stage64_store<false>(Bs, tid, rb);       // as stored; RLT reads it transposed (see panel_mma_tri)
        BAYENS_LDS_BARRIER();
        // next tile's global load flies while this product runs
        if (t < s) tile_load(s, t + 1, rb);
        else if (s + 1 < nsub) tile_load(s + 1, 0, rb);
        if (t < s) {
#pragma unroll
          for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
              acc[jt] = BAYENS_MFMA16(xf[t][kk], RLN ? Bs[kk * 4 + lk][jt * 16 + lr] : Bs[jt * 16 + lr][kk * 4 + lk],
                                      acc[jt]);
        } else {
          // X_j = acc op(M_j); RLT: Bs[k][col] = M_j[col][k] (upper triangular image); RLN: Bs[k][col] = M_j[k][col]
          v4d x[4];
#pragma unroll
          for (int jt = 0; jt < 4; ++jt) x[jt] = v4d{0, 0, 0, 0};
          panel_mma_tri<!RLN, !RLN>(x, As, Bs, wave, lane);
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) B[(long)(wave * 16 + acc_row(lk, r)) * g.ldb + 64 * j + jt * 16 + lr] = x[jt][r];
          if (s < 3 && s + 1 < nsub) {
            // the wave's own rows of As: finished block in, A fragments out (same wave wrote and reads)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
              for (int r = 0; r < 4; ++r) As[wave * 16 + acc_row(lk, r)][jt * 16 + lr] = -x[jt][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) xf[s][kk] = As[wave * 16 + lr][kk * 4 + lk];
          }
        }
      }
    }
  }
}

// elements of an m x n output that are computed: all, or the lower trapezoid i >= j (m >= n; n^2/2 for m = n)
double out_elems(int m, int n, int flags) {
  if (!(flags & GEMM_TRI_OUT) || m < n) return (double)m * n;
  return (double)m * n - 0.5 * (double)n * n;
}
double algorithmic_flop(int nb, int m, int n, int k, int flags) {
  if ((flags & GEMM_TRI_OUT) && (flags & GEMM_KTRIM_TN_LOWER)) return 2.0 * m * (double)n * k * nb / 6.0;
  if (flags & GEMM_KTRIM_A_LOWER) {   // row i of A ends at column i: sum_i min(k, i + 1)
    const double kk = k < m ? k : m;
    return 2.0 * n * nb * (0.5 * kk * kk + (double)(m - kk) * k);
  }
  return 2.0 * k * out_elems(m, n, flags) * nb;
}
// operand bytes a launch has to move at least once: A and B read (half each when both are lower triangles),
// the computed part of C written, and read when beta != 0
double algorithmic_bytes(int nb, int m, int n, int k, int flags, double beta) {
  const double tri = (flags & GEMM_KTRIM_TN_LOWER) ? 0.5 : 1.0;
  const double c = out_elems(m, n, flags);
  return (double)sizeof(real) * nb * (tri * ((double)m * k + (double)n * k) + c * (beta != 0.0 ? 2.0 : 1.0));
}

}  // namespace

// does the 128 x 128 tiling leave most of the chip idle?  Measured per shape (profiles/r05_gemm_tile64.txt): the small tiles win
// up to ~400 workgroups of the large tiling (at 480 - 1024 x 256 x 256, 32 members - they lose again).  BAYENS_TILE64_MAX=n /
// bayens_set_gemm_tile64_max override the limit (0: never).
static bool small_output(const GemmArgs& g, int flags, int override_max) {
  static const int env_max = [] { const char* e = std::getenv("BAYENS_TILE64_MAX"); return e ? std::atoi(e) : -1; }();
  const int forced = override_max >= 0 ? override_max : env_max;
  const int limit = forced >= 0 ? forced : 400;
  long active = 0;
  for (int tm = 0; tm < g.tiles_m; ++tm) active += (flags & GEMM_TRI_OUT) ? (tm + 1 < g.tiles_n ? tm + 1 : g.tiles_n) : g.tiles_n;
  return active * g.nb <= limit;
}

void HipPrims::gemm(int nb, int m, int n, int k, double alpha, Mat A, Mat B, double beta, Mat C, int flags,
                    double diag_add) {
  if (nb <= 0 || m <= 0 || n <= 0) return;
  GemmArgs g;
  g.A = A.p; g.B = B.p; g.C = C.p;
  g.sA = A.stride; g.sB = B.stride; g.sC = C.stride;
  g.lda = A.ld; g.ldb = B.ld; g.ldc = C.ld;
  g.m = m; g.n = n; g.k = k;
  g.alpha = alpha; g.beta = beta; g.diag_add = diag_add;
  g.flags = flags;
  g.tiles_m = (m + BM - 1) / BM;
  g.tiles_n = (n + BN - 1) / BN;
  g.total = g.tiles_m * g.tiles_n * nb;
  g.nb = nb;
  if ((m % 64) || (n % 64) || (k % 64) || (A.ld % VEC) || (B.ld % VEC))
    throw HipError{-1, "gemm: dimensions must be multiples of 64 and leading dimensions 16-byte multiples"};
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof.enabled) {
    auto take = [&]() {
      hipEvent_t e;
      if (!prof.pool.empty()) { e = prof.pool.back(); prof.pool.pop_back(); }
      else BAYENS_HIP(hipEventCreate(&e));
      return e;
    };
    e0 = take(); e1 = take();
    BAYENS_HIP(hipEventRecord(e0, stream));
  }
  const bool ta = flags & GEMM_TA, tb = flags & GEMM_TB;
  dim3 grid(g.total), block(256);
  const int special = GEMM_KTRIM_TN_LOWER | GEMM_ZERO_SKIPPED;
  int fam = ta ? FAM_GEMM_TN : (tb ? FAM_GEMM_NT : FAM_GEMM_NN);
  if (n == 64 && k == 64 && !ta && !(flags & special)) {
    fam = FAM_PANEL64;
    // with n = 64 every output block lies on or below the block diagonal, so GEMM_TRI_OUT skips nothing
    dim3 pgrid((m / 64) * nb);
    if (tb) hipLaunchKernelGGL((gemm_panel64<true>), pgrid, block, 0, stream, g);
    else hipLaunchKernelGGL((gemm_panel64<false>), pgrid, block, 0, stream, g);
  } else if (small_output(g, flags, tile64_max)) {
    fam = FAM_TILE64;                        // (its own family: family 1-3 stay exactly the launches of gemm_f64_mfma<TA, TB>)
    dim3 tgrid((m / 64) * (n / 64) * nb);
    if (!ta && tb) hipLaunchKernelGGL((gemm_tile64<false, true>), tgrid, block, 0, stream, g);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_tile64<true, false>), tgrid, block, 0, stream, g);
    else if (!ta && !tb) hipLaunchKernelGGL((gemm_tile64<false, false>), tgrid, block, 0, stream, g);
    else hipLaunchKernelGGL((gemm_tile64<true, true>), tgrid, block, 0, stream, g);
  } else if (!ta && tb) hipLaunchKernelGGL((gemm_f64_mfma<false, true>), grid, block, 0, stream, g);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_f64_mfma<true, false>), grid, block, 0, stream, g);
  else if (!ta && !tb) hipLaunchKernelGGL((gemm_f64_mfma<false, false>), grid, block, 0, stream, g);
  else hipLaunchKernelGGL((gemm_f64_mfma<true, true>), grid, block, 0, stream, g);
  BAYENS_HIP(hipGetLastError());
  if (prof.enabled) {
    BAYENS_HIP(hipEventRecord(e1, stream));
    const double f = algorithmic_flop(nb, m, n, k, flags);
    prof.events.push_back(KernelProfile::Ev{e0, e1, fam, m, n, k, flags | (beta != 0.0 ? 256 : 0), f});
    prof.flop[FAM_ALL] += f; prof.flop[fam] += f;
    const double by = algorithmic_bytes(nb, m, n, k, flags, beta);
    prof.bytes[FAM_ALL] += by; prof.bytes[fam] += by;
    prof.launches[FAM_ALL] += 1; prof.launches[fam] += 1;
  }
}

void HipPrims::trsm_panel(int nb, int m, int n, Mat L, const real* inv, long inv_stride, Mat B, int mode) {
  if (nb <= 0 || m <= 0 || n <= 0) return;
  if ((m % 64) || (n % 64) || n > 256 || (L.ld & 1) || (B.ld & 1))
    throw HipError{-1, "trsm_panel: m, n must be multiples of 64, n <= 256"};
  PanelArgs g;
  g.L = L.p; g.inv = inv; g.B = B.p;
  g.sL = L.stride; g.sInv = inv_stride; g.sB = B.stride;
  g.ldl = L.ld; g.ldb = B.ld; g.m = m; g.nsub = n / 64;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof.enabled) {
    auto take = [&]() {
      hipEvent_t e;
      if (!prof.pool.empty()) { e = prof.pool.back(); prof.pool.pop_back(); }
      else BAYENS_HIP(hipEventCreate(&e));
      return e;
    };
    e0 = take(); e1 = take();
    BAYENS_HIP(hipEventRecord(e0, stream));
  }
  dim3 grid((m / 64) * nb), block(256);
  if (mode == 0) hipLaunchKernelGGL((trsm_panel_kernel<false>), grid, block, 0, stream, g);
  else hipLaunchKernelGGL((trsm_panel_kernel<true>), grid, block, 0, stream, g);
  BAYENS_HIP(hipGetLastError());
  if (prof.enabled) {
    BAYENS_HIP(hipEventRecord(e1, stream));
    const double f = (double)nb * m * (double)n * n;      // triangular solve: m n^2 flop
    prof.events.push_back(KernelProfile::Ev{e0, e1, FAM_TRSM_PANEL, m, n, n, mode, f});
    prof.flop[FAM_ALL] += f; prof.flop[FAM_TRSM_PANEL] += f;
    const double by = (double)sizeof(real) * nb * (2.0 * m * (double)n + 0.5 * (double)n * n);
    prof.bytes[FAM_ALL] += by; prof.bytes[FAM_TRSM_PANEL] += by;
    prof.launches[FAM_ALL] += 1; prof.launches[FAM_TRSM_PANEL] += 1;
  }
}

}  // namespace BAYENS_NS
