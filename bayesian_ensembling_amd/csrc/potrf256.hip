// potrf256.hip - Cholesky of a 256 x 256 diagonal block (+ the inverses of its four 64 x 64 diagonal blocks) in ONE launch,
// one workgroup of eight waves per member, the block's off-diagonal tiles RESIDENT IN REGISTERS (round 4).
//
// Where it sits: the blocked right-looking Cholesky of the general path (csrc/driver.cpp::potrf) factors one 256-wide block
// column per step; its diagonal block went through potrf_rec(256) = two-leaf kernel, 128-wide panel solve, rank-128 update,
// two-leaf kernel - four dependent launches, ~130 us alone and 115-160 us EACH leaf kernel beside the side stream's trailing
// update (profiles/r03_potrf_sequence.txt) - and that chain, eight times per factorisation, five factorisations per step, is
// what the factorisation-type stages of the step wait for (reference arithmetic: the Cholesky factors of K + J I, Lam and cov,
// SURVEY Appendix A; ensembles/models.py:185-220, data.py:38-39).
//
// How: the lower triangle of the block is 136 tiles of 16 x 16 = 272 KB - more than the CU's 160 KB of LDS, less than its
// 512 KB of vector registers.  Roles: wave 0 runs the serial chain and nothing else (per block column: the last update of the
// next diagonal tile, then its 16 x 16 factorisation + inverse, smalln::diag16); wave 4 - wave 0's partner on its SIMD, whose
// MFMAs would double the chain's run time (tools/diag16_bench.py) - only helps with input / output; the other SIX waves own the
// 120 strictly lower tiles, 20 each = 160 registers, each tile held TRANSPOSED in accumulator layout (T = A_ij^T): an
// accumulator tile is an MFMA B operand as it stands, so the panel step L_ik^T = inv(L_kk) A_ik^T takes its right-hand side
// from registers, and the trailing update T_ij -= L_jk L_ik^T takes both operands from the panel tiles of column k in LDS -
// no read-modify-write of an off-diagonal tile through memory anywhere.  The 16 diagonal tiles live in LDS (symmetric
// images), updated there by the workers and picked up by wave 0.  Per block column k: [panel: the owners of the tiles (i, k)]
// barrier [wave 0: next diagonal tile | workers: trailing update] barrier.  Which tile a wave owns is a constant expression
// (one specialisation of the schedule per wave, every loop over tiles unrolled at compile time): with one code path and
// run-time predicates every conditional update of a register tile became a merge of two values and the register allocator
// spilled 760 registers; with diag16 sharing a wave with 17 register tiles, 1400.
// LDS: two panel buffers (16 tiles each), the 16 inverted diagonal tiles, the 16 diagonal tiles, two factor tiles, a staging
// tile per wave: 148 KB.  The inverses of the four 64 x 64 diagonal blocks (what the fused panel solves of csrc/gemm.hip
// multiply by) are formed at the end, one pair of tile columns per wave, the finished tiles of a column kept in registers.
//
// fp64 only (the reference precision); the fp32 instance of the fit keeps the four-launch composition.
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

#define BAYENS_SMALL_HELPERS_ONLY
#include "small.hip"
#include "potrf256.h"

#ifdef BAYENS_F32
#error "potrf256.hip is built for the fp64 fit only"
#endif

namespace bayens {
namespace potrf256n {

using namespace smalln;

constexpr int NT = 16;                 // tiles per side
constexpr int PW = 8;                  // waves
constexpr int PTH = PW * 64;
constexpr int NWK = 6;                 // worker waves (1, 2, 3, 5, 6, 7)
constexpr int NOFF = NT * (NT - 1) / 2;// 120 strictly lower tiles
constexpr int TPW = NOFF / NWK;        // 20 per worker
static_assert(TPW * NWK == NOFF, "the strictly lower tiles deal evenly over the workers");
// LDS (doubles)
constexpr int L_PANEL = 0;                       // [2][NT] tiles
constexpr int L_DFAC = L_PANEL + 2 * NT * TS;    // [2] tiles: factor of the current diagonal tile
constexpr int L_DINV = L_DFAC + 2 * TS;          // [NT] tiles: inverses of the diagonal tiles
constexpr int L_DIAG = L_DINV + NT * TS;         // [NT] tiles: the diagonal tiles (symmetric images) until they are factored
constexpr int L_IO = L_DIAG + NT * TS;           // [PW] tiles: one staging tile per wave
constexpr int L_END = L_IO + PW * TS;
constexpr size_t LDS_BYTES = (size_t)L_END * sizeof(double);
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

// diagnostic: cycle-counter stamps of workgroup 0, [wave][16] (bayens_potrf256_stamps; a handful of stores per launch)
__device__ unsigned long long g_stamps[PW * 16];
#define PSTAMP(k) do { if (b == 0 && lane == 0) g_stamps[W * 16 + (k)] = __builtin_readcyclecounter(); } while (0)

#define LWAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// keeps the instruction scheduler from pulling the operand loads of LATER tile operations of an unrolled phase in front of the
// current one (20 tile updates in a row = 160 operand loads it would like to issue first: hundreds of spilled registers)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// No fp64 VECTOR instruction in a tile operation: such instructions share the matrix pipe and wait out the MFMAs of the SIMD's
// other wave (tools/diag16_bench.py: 71 cycles per dependent link beside MFMAs) - so the sign of an operand is flipped with an
// integer XOR and the four MFMAs of a tile product run as ONE chain into the accumulator (no second chain to add up afterwards).
__device__ __forceinline__ double flip(double x) { return __hiloint2double(__double2hiint(x) ^ (int)0x80000000, __double2loint(x)); }
// acc +- A B with A an LDS tile and B given in accumulator-layout registers (= the MFMA's B operand as it stands)
template <bool NEG>
__device__ __forceinline__ v4d mma_nb(v4d acc, const double* A, v4d b, const Lane& o) {
  double a[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = NEG ? flip(A[o.rc[q]]) : A[o.rc[q]];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc = MFMA(a[q], b[q], acc);
  return acc;
}
// acc -= A B^T, both LDS tiles in standard orientation
__device__ __forceinline__ v4d mma_sub_nt(v4d acc, const double* A, const double* B, const Lane& o) {
  double a[4], b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { a[q] = flip(A[o.rc[q]]); b[q] = B[o.rc[q]]; }
#pragma unroll
  for (int q = 0; q < 4; ++q) acc = MFMA(a[q], b[q], acc);
  return acc;
}
// LDS tile <- accumulator registers, transposed: element (row, col) of the registers lands at (col, row)
__device__ __forceinline__ void st_tile_t(double* t, const Lane& o, v4d a) {
#pragma unroll
  for (int r = 0; r < 4; ++r) t[o.rc[r]] = a[r];
}

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>());
}
constexpr int tile_row(int q) {          // (i, j), j < i, of the row-packed STRICTLY lower triangle, as constant expressions
  int i = 1;
  while ((i + 1) * i / 2 <= q) ++i;
  return i;
}
constexpr int tile_col(int q) { return q - tile_row(q) * (tile_row(q) - 1) / 2; }
constexpr int ltc(int i, int j) { return i * (i + 1) / 2 + j; }

// the serial 16 x 16 factorisation + inverse as a function of its own (LDS operands as offsets: a pointer that crosses a real
// call is a generic one); returns the first failed pivot (1-based) or `bad` as it came in.  A REAL call on purpose: the chain
// wave runs it 16 times, and 16 inlined copies are 45 KB of code that is executed once each - every call then waits for its
// instructions to arrive from memory (87 against 63 us per 32 x 256 factorisation)
__device__ __noinline__ int diag16_call(v4d acc, int fac_off, int inv_off, int pivot_base, int bad) {
  const int lane = threadIdx.x & 63;
  Lane o;
  lane_init(o, lane);
  fac_off = __builtin_amdgcn_readfirstlane(fac_off);
  inv_off = __builtin_amdgcn_readfirstlane(inv_off);
  pivot_base = __builtin_amdgcn_readfirstlane(pivot_base);
  diag16(acc, lds_base() + fac_off, lds_base() + inv_off, lane, pivot_base, bad, o);
  return bad;
}

// The factorisation as wave W runs it (see the head of the file for the roles).
template <int W>
__device__ __forceinline__ void factor_wave(double* a, long ld, double* inv, long inv_stride, int b, int pivot_base, int* info,
                                            int lane, const Lane& o) {
  constexpr bool CHAIN = W == 0, HELPER = W == 4, WORKER = !CHAIN && !HELPER;
  constexpr int WK = W < 4 ? W - 1 : W - 2;          // worker index 0 .. 5 (meaningless for waves 0 and 4)
  constexpr int NS = WORKER ? TPW : 1;
  double* lds = lds_base();
  double* panel = lds + L_PANEL;
  double* dfac = lds + L_DFAC;
  double* dinv = lds + L_DINV;
  double* diag = lds + L_DIAG;
  double* io = lds + L_IO + W * TS;
  v4d T[NS];
  PSTAMP(0);
  // ---- in.  Workers: T[s] = A_ij^T of tile q = WK + 6 s (accumulator layout), through the wave's staging tile (coalesced
  // 128-byte rows).  Waves 0 and 4: the diagonal tiles into LDS as symmetric images (only their lower triangles are trusted).
  if constexpr (WORKER) {
    // Coalesced (four rows x 128 bytes per instruction) through staging tiles in the still unused panel region, four tiles per
    // round: the requests of a round are in flight together and a round costs one global round trip.  (Straight into the
    // registers - 32-byte segments, all 80 requests at once - was tried: the CU's memory path took 12 us for them and the chain
    // wave's diagonal tiles queued behind; one tile per round through one staging tile serialised 20 round trips.)
    __builtin_amdgcn_s_sleep(8);          // (lets the chain wave's requests for the diagonal tiles go first)
    double* st5 = panel + WK * 4 * TS;
    static_for<TPW / 4>([&](auto RR) {
      constexpr int r5 = decltype(RR)::value;
      double v[4][4];
      static_for<4>([&](auto U) {
        constexpr int u = decltype(U)::value, s = 4 * r5 + u, q = WK + NWK * s, i = tile_row(q), j = tile_col(q);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int e = lane + 64 * t, r = e >> 4, c = e & 15;
          v[u][t] = a[(long)(16 * i + r) * ld + 16 * j + c];
        }
      });
      static_for<4>([&](auto U) {
        constexpr int u = decltype(U)::value;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int e = lane + 64 * t;
          st5[u * TS + swz(e >> 4, e & 15)] = v[u][t];
        }
      });
      LWAIT();
      static_for<4>([&](auto U) {
        constexpr int u = decltype(U)::value, s = 4 * r5 + u;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) T[s][q4] = st5[u * TS + o.rc[q4]];          // register (kq, lr) = A_ij[lr][kq]
      });
      LWAIT();                            // (the staging tiles are rewritten by the next round)
      FENCE();
    });
  } else {
    // (all eight tiles requested before the first is placed: one global round trip, not eight)
    double v[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = 2 * u + (CHAIN ? 0 : 1);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int e = lane + 64 * t, r = e >> 4, c = e & 15;
        v[u][t] = a[(long)(16 * k + r) * ld + 16 * k + c];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = 2 * u + (CHAIN ? 0 : 1);
      double* dk = diag + k * TS;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int e = lane + 64 * t, r = e >> 4, c = e & 15;
        if (c <= r) { dk[swz(r, c)] = v[u][t]; dk[swz(c, r)] = v[u][t]; }
      }
    }
  }
  PSTAMP(1);
  SYNC();
  PSTAMP(2);

  int bad = 0;
  auto write_diag = [&](int k, const double* fac) {       // L_kk out: lower triangle, zeros above
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = lane + 64 * t, r = e >> 4, c = e & 15;
      a[(long)(16 * k + r) * ld + 16 * k + c] = c <= r ? fac[swz(r, c)] : 0.0;
    }
  };
  auto report = [&]() {                 // the first failed pivot of the member (columns report in order: a barrier apart)
    if (lane == 0 && bad) atomicCAS(info + b, 0, bad);
  };
  if constexpr (CHAIN) {
    bad = diag16_call(ld_tile(diag, o), L_DFAC, L_DINV, pivot_base, bad);
    LWAIT();
    write_diag(0, dfac);
    report();
  }
  SYNC();
  PSTAMP(3);
  // The column loop is a RUN-TIME loop over compile-time slots (tile q = WK + 6 s has constant (i, j); "is it in column k" /
  // "is it right of column k" are scalar tests): fully unrolled over the 15 columns the kernel was 350 KB of straight-line code,
  // each instruction executed once, and the waves waited for their instructions.
  // Columns go in PAIRS (k, k + 1), k even: after the panel of column k only the tiles of column k + 1 take its rank-16 update;
  // every tile further right waits for the panel of column k + 1 and takes both in ONE operation (eight MFMAs behind sixteen
  // operand reads): half as many tile operations, each with twice the matrix work behind its fixed cost (a branch, an LDS round
  // trip) - a tile operation of four MFMAs took ~850 cycles where its MFMAs need 256.
  auto store_col = [&](int k, const double* pk, long ldo, int lane_o) {
    // the finished tiles of column k leave for memory off the chain's path: read back from the panel tile, so that a store
    // instruction covers four whole 128-byte rows (32-byte segments straight from the registers took ~700 cycles per
    // instruction to issue), and zeros go into the mirror image of a tile that lies inside a diagonal 64-block (the consumers
    // of a factor read whole 64 x 64 diagonal blocks: the triangle above the diagonal must hold zeros, and the input holds the
    // symmetric values there when it comes out of a GEMM)
    static_for<TPW>([&](auto S) {
      constexpr int s = decltype(S)::value, q = WK + NWK * s, i = tile_row(q), j = tile_col(q);
      if (j == k) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int e = lane_o + 64 * t, r = e >> 4, c = e & 15;
          a[(long)(16 * i + r) * ldo + 16 * j + c] = pk[i * TS + swz(r, c)];
        }
        if constexpr ((i >> 2) == (j >> 2)) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int e = lane_o + 64 * t;
            a[(long)(16 * j + (e >> 4)) * ldo + 16 * i + (e & 15)] = 0.0;
          }
        }
      }
      FENCE();
    });
  };
  auto panel_col = [&](int k, double* pk) {      // L_ik^T = inv(L_kk) A_ik^T (right-hand side from registers); tile -> LDS, standard orientation
    const double* ik = dinv + k * TS;
    static_for<TPW>([&](auto S) {
      constexpr int s = decltype(S)::value, q = WK + NWK * s, i = tile_row(q), j = tile_col(q);
      if (j == k) {
        const v4d zero = {0, 0, 0, 0};
        T[s] = mma_nb<false>(zero, ik, T[s], o);
        st_tile_t(pk + i * TS, o, T[s]);
      }
      FENCE();
    });
  };
  auto chain_step = [&](int kn, v4d d) {         // factor + invert diagonal tile kn (all its updates applied), L_kn,kn out
    bad = diag16_call(d, L_DFAC + (kn & 1) * TS, L_DINV + kn * TS, pivot_base + 16 * kn, bad);
    LWAIT();
    write_diag(kn, dfac + (kn & 1) * TS);
    report();
  };
#pragma unroll 1
  for (int k = 0; k < NT; k += 2) {
    double* p0 = panel;                   // panel of column k (k even)
    double* p1 = panel + NT * TS;         // panel of column k + 1
    // (output addresses are derived from an opaque copy of the leading dimension: otherwise the compiler recognises the row
    //  addresses of the input phase, keeps all 80 of them alive through the loop and spills 500 registers for it)
    long ldo = ld;
    asm volatile("" : "+s"(ldo));
    int lane_o = lane;                    // (likewise the lane-dependent row / column offsets of the stores: recomputed here - a few
    asm volatile("" : "+v"(lane_o));      //  integer operations - instead of hoisted out of the loop, spilled and reloaded from memory)
    // ---- 1: panel of column k
    if constexpr (WORKER) panel_col(k, p0);
    SYNC();
    // ---- 2: wave 0: diagonal tile k + 1 | workers: column k out, the rank-16 update of the tiles of column k + 1
    if constexpr (CHAIN) {
      v4d d = ld_tile(diag + (k + 1) * TS, o);
      d = mma_sub_nt(d, p0 + (k + 1) * TS, p0 + (k + 1) * TS, o);
      chain_step(k + 1, d);
    }
    if constexpr (WORKER) {
      store_col(k, p0, ldo, lane_o);
      static_for<TPW>([&](auto S) {
        constexpr int s = decltype(S)::value, q = WK + NWK * s, i = tile_row(q), j = tile_col(q);
        if (j == k + 1) T[s] = mma_sub_nt(T[s], p0 + j * TS, p0 + i * TS, o);        // T_ij -= L_jk L_ik^T  (T = A^T)
        FENCE();
      });
    }
    SYNC();
    // ---- 3: panel of column k + 1
    if constexpr (WORKER) panel_col(k + 1, p1);
    SYNC();
    // ---- 4: wave 0: diagonal tile k + 2 (both columns' updates) | workers: column k + 1 out, the rank-32 update of everything
    // right of it - register tiles, and the diagonal tiles beyond k + 2 in LDS (symmetric images)
    if (k + 2 < NT) {
      if constexpr (CHAIN) {
        v4d d = ld_tile(diag + (k + 2) * TS, o);
        d = mma_sub_nt(d, p0 + (k + 2) * TS, p0 + (k + 2) * TS, o);
        d = mma_sub_nt(d, p1 + (k + 2) * TS, p1 + (k + 2) * TS, o);
        chain_step(k + 2, d);
      }
    }
    if constexpr (WORKER) {
      store_col(k + 1, p1, ldo, lane_o);
      static_for<TPW>([&](auto S) {
        constexpr int s = decltype(S)::value, q = WK + NWK * s, i = tile_row(q), j = tile_col(q);
        if (j > k + 1) {
          T[s] = mma_sub_nt(T[s], p0 + j * TS, p0 + i * TS, o);
          T[s] = mma_sub_nt(T[s], p1 + j * TS, p1 + i * TS, o);
        }
        FENCE();
      });
      static_for<3>([&](auto JJ) {
        constexpr int jj = WK + NWK * decltype(JJ)::value;
        if constexpr (jj < NT) {
          if (jj > k + 2) {
            double* dj = diag + jj * TS;
            v4d t = ld_tile(dj, o);
            t = mma_sub_nt(t, p0 + jj * TS, p0 + jj * TS, o);
            t = mma_sub_nt(t, p1 + jj * TS, p1 + jj * TS, o);
            st_tile(dj, o, t);
          }
        }
        FENCE();
      });
    }
    SYNC();
    if (k == 0) PSTAMP(4);
    if (k == 6) PSTAMP(5);
  }
  PSTAMP(6);

  // ---- out: the strictly lower tiles (registers hold L_ij^T), zeros above the diagonal inside the diagonal 64-blocks; the
  // strictly lower tiles of the four diagonal 64-blocks are also left in LDS (panel region) for the inverses below
  double* stage = panel;               // [4 blocks][6 tiles]: (i % 4, j % 4) = (1,0) (2,0) (2,1) (3,0) (3,1) (3,2)
  if constexpr (WORKER) {
    static_for<TPW>([&](auto S) {
      constexpr int s = decltype(S)::value, q = WK + NWK * s, i = tile_row(q), j = tile_col(q);
      if constexpr ((i >> 2) == (j >> 2)) st_tile_t(stage + ((i >> 2) * 6 + ltc((i & 3) - 1, j & 3)) * TS, o, T[s]);
    });
  }
  PSTAMP(7);
  SYNC();
  PSTAMP(8);

  // ---- inverses of the four diagonal 64-blocks: X = L_JJ^-1 by block forward substitution, one tile column per task,
  // X_cc = inv_cc, X_ic = -inv_ii sum_{k = c}^{i-1} L_ik X_kc; wave W takes block W / 2 and columns {0, 3} or {1, 2}
  {
    constexpr int J = W >> 1;
    double* iv = inv + (long)b * inv_stride + (long)J * 4096;
    auto out_tile = [&](int r4, int c4, v4d x, bool zero) {      // tile (r4, c4) of the 64 x 64 row-major inverse
      if (!zero) { st_tile(io, o, x); LWAIT(); }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int e = lane + 64 * t, r = e >> 4, c = e & 15;
        iv[(16 * r4 + r) * 64 + 16 * c4 + c] = zero ? 0.0 : io[swz(r, c)];
      }
      LWAIT();
    };
    static_for<2>([&](auto H) {
      constexpr int half = decltype(H)::value;
      constexpr int c = (W & 1) ? 1 + half : 3 * half;           // {1, 2} or {0, 3}
      v4d x[4];
      x[0] = ld_tile(dinv + (4 * J + c) * TS, o);               // X_cc (slot 0 = row c, slot t = row c + t)
      out_tile(c, c, x[0], false);
      for (int r4 = 0; r4 < c; ++r4) out_tile(r4, c, x[0], true);
      static_for<3>([&](auto TT) {
        constexpr int t = decltype(TT)::value + 1, i = c + t;
        if constexpr (i < 4) {
          v4d acc = {0, 0, 0, 0};
          static_for<t>([&](auto U) {                            // k = c + u
            constexpr int u = decltype(U)::value;
            acc = mma_nb<false>(acc, stage + (J * 6 + ltc(i - 1, c + u)) * TS, x[u], o);
          });
          const v4d zero = {0, 0, 0, 0};
          x[t] = mma_nb<true>(zero, dinv + (4 * J + i) * TS, acc, o);
          out_tile(i, c, x[t], false);
        }
      });
    });
  }
  PSTAMP(9);
}

__global__ __launch_bounds__(PTH, 2) void potrf256_kernel(Mat A, double* inv, long inv_stride, int pivot_base, int* info) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Lane o;
  lane_init(o, lane);
  double* a = A.p + (long)b * A.stride;
  __builtin_amdgcn_s_setprio(2);       // a latency-bound chain beside the side stream's throughput kernels
  switch (wave) {                      // (every wave reaches the same barriers in the same order: the schedule is shared)
    case 0: factor_wave<0>(a, A.ld, inv, inv_stride, b, pivot_base, info, lane, o); break;
    case 1: factor_wave<1>(a, A.ld, inv, inv_stride, b, pivot_base, info, lane, o); break;
    case 2: factor_wave<2>(a, A.ld, inv, inv_stride, b, pivot_base, info, lane, o); break;
    case 3: factor_wave<3>(a, A.ld, inv, inv_stride, b, pivot_base, info, lane, o); break;
    case 4: factor_wave<4>(a, A.ld, inv, inv_stride, b, pivot_base, info, lane, o); break;
    case 5: factor_wave<5>(a, A.ld, inv, inv_stride, b, pivot_base, info, lane, o); break;
    case 6: factor_wave<6>(a, A.ld, inv, inv_stride, b, pivot_base, info, lane, o); break;
    default: factor_wave<7>(a, A.ld, inv, inv_stride, b, pivot_base, info, lane, o); break;
  }
}

void potrf256_stamps(unsigned long long* out128) {
  BAYENS_HIP(hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * PW * 16));
}

void potrf_diag256(hipStream_t stream, int nb, Mat A, double* inv, long inv_stride, int pivot_base, int* info) {
  static bool configured[64] = {};
  int dev = 0;
  BAYENS_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !configured[dev]) {
    BAYENS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(potrf256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)LDS_BYTES));
    if (dev >= 0 && dev < 64) configured[dev] = true;
  }
  hipLaunchKernelGGL(potrf256_kernel, dim3(nb), dim3(PTH), LDS_BYTES, stream, A, inv, inv_stride, pivot_base, info);
  BAYENS_HIP(hipGetLastError());
}

}  // namespace potrf256n
}  // namespace bayens
