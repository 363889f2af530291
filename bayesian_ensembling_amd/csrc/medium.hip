// medium.hip - the whole GP fit of a MID-SIZE member (94 < N <= 254 time points) in ONE kernel, one workgroup per
// member: the reference's real workload (historical annual series N = 165 fitted with n_optim_nits = 2000-2500,
// experiments/full_experiment_script.py:57,87-113; pre_fitting_cmip6models.py:76-77), which through the general path is
// ~30 dependent launches per optimiser iteration over matrices padded to 192 - latency bound at ~0.36 ms per iteration
// for 20 members and growing with the batch.
//
// Same arithmetic as the fused small-member kernel (small.hip, whose tile helpers this file includes: 16x16 tiles,
// tile-packed lower triangles, explicit triangular inverses riding along with the blocked factorisations, MFMA-based
// 16x16 diagonal factorisation) - reference ensembles/models.py:185-220 + data.py:38-39, SURVEY.md Appendix A.  What
// differs is where the matrices live.  At NT = 11 tiles per side a tile-packed triangle is 132 KB: the matrices cannot
// share the CU's 160 KB of LDS.  They live in a per-workgroup scratch in HBM that stays L2-resident (eight triangles,
// ~1.1 MB per member at N = 165), and LDS is the staging area between that scratch and the matrix cores:
//   * a tile in the scratch is stored in ACCUMULATOR-NATIVE order - lane (lr, lk)'s four accumulator elements
//     (rows lk + 4r of column lr) are 32 contiguous bytes - so every load / store of a tile is one fully coalesced 2 KB
//     wave access (the first version of this kernel read MFMA operands straight from the scratch, 8 scattered bytes per
//     lane and instruction: the CU's L1 saw 64 line accesses per tile and was the bottleneck, 1.06 M cycles per iteration);
//   * MFMA operands always come from LDS tiles in small.hip's conflict-free swizzled layout: every product phase is a
//     loop over the contraction index k whose step stages the tiles that carry k (a column / row / "cross" of a
//     triangle, <= 2 NT tiles) into LDS with coalesced loads, after which every wave updates the accumulators of the
//     output tiles it owns (<= 9 per wave and pass, in registers) - each scratch tile is read once per phase;
//   * the blocked factorisations keep the current block column, the current block row of the inverse and of the factor
//     in LDS, so a trailing / inverse / Gram update is one coalesced read-modify-write of its output tile.
// With no LDS limit on the matrices the tile count is a RUN-TIME value (one instantiation for every N).
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>
#ifdef BAYENS_MEDIUM_CHECK
// Debug build (BAYENS_EXTRA_DEFS=-DBAYENS_MEDIUM_CHECK python -m bayesian_ensembling_amd.build -> libbayens_debug.so, used through
// BAYENS_LIB): EVERY global access of the kernel - the tile scratch, X / y / nu, every output, the loss trace, the stamps - is
// checked before it is made against the regions medium_fit registered (the tile scratch against the launching workgroup's OWN
// region); an access outside them is counted, its address, source line and workgroup recorded, and NOT performed (loads return
// zeros): an indexing error shows up as a count instead of as a memory fault (or, worse, as a silent read of a neighbour's tiles).
// medium_fit prints the result after every launch.
namespace bayens { namespace mediumn {
constexpr int CHK_REGIONS = 16;
__device__ unsigned long long g_reg[2 * CHK_REGIONS];   // (base, bytes) of every array handed to the kernel; region 0 = the tile scratch
__device__ unsigned long long g_chk[8];                 // [1] scratch bytes per workgroup, [2] violations, [3] first address, [4] its line, [5] its workgroup
__device__ __forceinline__ bool chk_fail(unsigned long long a, int site) {
  if (atomicAdd(&g_chk[2], 1ull) == 0) { g_chk[3] = a; g_chk[4] = (unsigned long long)site; g_chk[5] = blockIdx.x; }
  return false;
}
__device__ __forceinline__ bool chk_g(const void* p, unsigned bytes, int site) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned long long lo = g_reg[0] + (unsigned long long)blockIdx.x * g_chk[1];
  if (a >= lo && a + bytes <= lo + g_chk[1]) return true;
  for (int k = 1; k < CHK_REGIONS; ++k)
    if (g_reg[2 * k + 1] && a >= g_reg[2 * k] && a + bytes <= g_reg[2 * k] + g_reg[2 * k + 1]) return true;
  return chk_fail(a, site);
}
template <class T> __device__ __forceinline__ T chk_ld(const T* p, int site) { return chk_g(p, sizeof(T), site) ? *p : T(0); }
template <class T, class V> __device__ __forceinline__ void chk_st(T* p, V v, int site) { if (chk_g(p, sizeof(T), site)) *p = (T)v; }
}}
#define GLD(lv) ::bayens::mediumn::chk_ld(&(lv), __LINE__)
#define GST(lv, v) ::bayens::mediumn::chk_st(&(lv), (v), __LINE__)
#else
#define GLD(lv) (lv)
#define GST(lv, v) ((lv) = (v))
#endif

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
#define LWAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// (diagnostic build -DBAYENS_MEDIUM_STOPS, g.tune bits 8..: the workgroup ends at phase boundary k of iteration `it_now` when the launch asks for it -
//  BAYENS_MEDIUM_TUNE = (1 + k + 16 it) << 8 - which is how a device fault is bracketed between two boundaries)
#ifdef BAYENS_MEDIUM_STOPS
#define MSTAMP(k) do { if (g.stamps && blockIdx.x == 0 && tid == 0) GST(g.stamps[k], __builtin_readcyclecounter()); \
                       if ((g.tune >> 8) == 1 + (k) + 16 * it_now) return; } while (0)
#else
#define MSTAMP(k) do { if (g.stamps && blockIdx.x == 0 && tid == 0) GST(g.stamps[k], __builtin_readcyclecounter()); } while (0)
#endif

constexpr int TPW = 9;                 // accumulator tiles a wave holds per pass of a product phase (72 VGPRs)
constexpr int PASS_TILES = TPW * NW;   // output tiles per pass

__device__ __forceinline__ void unpack_step(int& i, int& j, int step) {      // advance (i, j) by `step` packed positions
  j += step;
  while (j > i) { j -= i + 1; ++i; }
}
// accumulator-native tile in the scratch: element (row, col) at nidx(row, col); lane l owns doubles [4l, 4l + 4)
__device__ __forceinline__ int nidx(int row, int col) { return ((col + 16 * (row & 3)) << 2) + (row >> 2); }
#ifdef BAYENS_MEDIUM_CHECK
// a whole tile access: lane l touches the 32 bytes at t + 32 l of the workgroup's OWN scratch region
__device__ __forceinline__ bool chk_ok(const void* p, int lane, int site) {
  const unsigned long long a = (unsigned long long)p + 32ull * lane;
  const unsigned long long lo = g_reg[0] + (unsigned long long)blockIdx.x * g_chk[1];
  if (a >= lo && a + 32 <= lo + g_chk[1]) return true;
  return chk_fail(a, site);
}
#define CHK_LD(p) if (!chk_ok((const void*)(p), lane, __LINE__)) return v4d{0, 0, 0, 0}
#define CHK_ST(p) if (!chk_ok((const void*)(p), lane, __LINE__)) return
#else
#define CHK_LD(p)
#define CHK_ST(p)
#endif
__device__ __forceinline__ v4d ldn(const double* t, int lane) { CHK_LD(t); return reinterpret_cast<const v4d*>(t)[lane]; }
__device__ __forceinline__ void stn(double* t, int lane, v4d a) { CHK_ST(t); reinterpret_cast<v4d*>(t)[lane] = a; }
// the same through pointers that carry the GLOBAL address space in their type: a pointer that crosses a real function call
// is a generic one, its accesses FLAT ones (which also count as LDS operations: every operand read would wait for them)
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) v4d gv4d;
__device__ __forceinline__ v4d ldg(const gdouble* t, int lane) { CHK_LD((const double*)t); return reinterpret_cast<const gv4d*>(t)[lane]; }
__device__ __forceinline__ void stg(gdouble* t, int lane, v4d a) { CHK_ST((const double*)t); reinterpret_cast<gv4d*>(t)[lane] = a; }
// LDS tile <- accumulator registers, transposed (element (row, col) of the registers lands at (col, row))
__device__ __forceinline__ void st_tile_t(double* t, const Lane& o, v4d a) {
#pragma unroll
  for (int r = 0; r < 4; ++r) t[o.rc[r]] = a[r];
}
// acc += sa * A B with A an LDS tile and B given in accumulator-native registers (= the MFMA's B operand as it stands)
__device__ __forceinline__ v4d mma_nb(v4d acc, const double* A, v4d b, const Lane& o) {
  double a[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = A[o.rc[q]];
  v4d acc2 = {0, 0, 0, 0};
  acc = MFMA(a[0], b[0], acc);
  acc2 = MFMA(a[2], b[2], acc2);
  acc = MFMA(a[1], b[1], acc);
  acc2 = MFMA(a[3], b[3], acc2);
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
  return acc;
}
// acc += op(A) op(B), both LDS tiles, orientation chosen at run time (ta: use A^T, tb: use B^T)
__device__ __forceinline__ v4d mma_rt(v4d acc, const double* A, bool ta, const double* B, bool tb, const Lane& o) {
  double a[4], b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { a[q] = A[ta ? o.cr[q] : o.rc[q]]; b[q] = B[tb ? o.rc[q] : o.cr[q]]; }
  v4d acc2 = {0, 0, 0, 0};
  acc = MFMA(a[0], b[0], acc);
  acc2 = MFMA(a[2], b[2], acc2);
  acc = MFMA(a[1], b[1], acc);
  acc2 = MFMA(a[3], b[3], acc2);
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
  return acc;
}
// (T x)[lr] (ta = false) or (T^T x)[lr] (ta = true) of one LDS tile, x = 16 LDS doubles; valid in every lane after the shuffles
__device__ __forceinline__ double tile_mv(const double* T, bool ta, const double* x, const Lane& o) {
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) s = fma(T[ta ? o.cr[q] : o.rc[q]], x[o.kq(q)], s);
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}

// hides a wave-uniform value from loop-invariant code motion: without it the compiler hoists the LDS addresses of every
// (accumulator tile, operand, orientation) combination out of the k loops - ~150 VGPRs of addresses, i.e. hundreds of spills
__device__ __forceinline__ int opaque(int x) { x = __builtin_amdgcn_readfirstlane(x); asm volatile("" : "+s"(x)); return x; }

struct Ctx {            // what every phase needs
  int NT, NTT, wave, lane, tid;
  double* SA;           // LDS staging: slots [0, NT) and [NT, 2 NT), slot 2 NT = inverse of the current diagonal tile, 2 NT + 1 = work tile
  Lane o;
  __device__ __forceinline__ double* slot(int s) const { return SA + (size_t)opaque(s) * TS; }
};

// ---- blocked Cholesky of the scratch triangle A in place; WITH_INV: the inverse of the factor is formed alongside in Xb
// (block forward substitution).  Per block column k two phases, as in smalln::potrf_lds:
//   P  [every wave]   panel L_ik = A_ik inv_kk^T and X_ki = inv_kk T_ki; results go to the scratch AND stay in LDS
//                     (slot i: the panel tile for i > k, the inverse's tile for i <= k); MODE 1 also stages block row k
//                     of the finished factor (slots NT + i)
//   L  [wave 0]       next diagonal tile: update, 16x16 factorisation + inverse (diag16)
//      [waves 1..7]   trailing update A_ij -= L_ik L_jk^T, inverse update T_ji -= L_jk X_ki, extra(k): every operand is an LDS
//                     tile, every output one coalesced read-modify-write of a scratch tile
// MODE 0: nothing extra.  MODE 1 (the kernel matrix): G_ij += L_ki^T D_k L_kj into Gb and wv += L_k*^T (y / nu)_k.
// MODE 2 (Lam): S_ij += X_ki^T X_kj into Gb (the caller passes Sb).
// (NOT inlined: the fit kernel keeps dozens of pointers and sizes alive and the factorisation's own loop state on top of them
//  overflowed the scalar register file - ~500 scalar spills, every tile address recomputed through v_readlane chains.  As a
//  function of its own it sees only its arguments; LDS regions arrive as offsets and the scratch as an address-space-typed
//  pointer, so no access degrades to a flat one.)
struct PotrfArgs {
  gdouble* sc;                 // the workgroup's scratch
  long oA, oX, oG;             // offsets of the triangle, of its inverse (WITH_INV) and of the Gram triangle (MODE 1 / 2)
  int NT, NTT;
  int sa_off, rs_off, yd_off, wv_off;   // LDS offsets (doubles): staging tiles, 1 / sqrt(nu), y / nu, wv
  int w4idle;                  // measurement switch: wave 4 (wave 0's SIMD partner) takes no tile updates beside the serial 16x16
                               // factorisation - what the small-member kernel does (section 5b); slower here
};
// arguments of a real function arrive in vector registers: tell the compiler that these are wave-uniform
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ long uni(long x) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)x), hi = __builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long)x >> 32));
  return (long)(((unsigned long)hi << 32) | lo);
}
template <bool WITH_INV, int MODE>
__device__ __noinline__ int potrf_v2(PotrfArgs e_in, int bad) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Lane o;
  lane_init(o, lane);
  PotrfArgs e;
  e.sc = (gdouble*)uni((long)e_in.sc);
  e.oA = uni(e_in.oA); e.oX = uni(e_in.oX); e.oG = uni(e_in.oG);
  e.NT = uni(e_in.NT); e.NTT = uni(e_in.NTT);
  e.sa_off = uni(e_in.sa_off); e.rs_off = uni(e_in.rs_off); e.yd_off = uni(e_in.yd_off); e.wv_off = uni(e_in.wv_off);
  const bool w4idle = uni(e_in.w4idle) != 0;
  const int NT = e.NT;
  double* lds = lds_base();
  double* SA = lds + e.sa_off;
  const double* rs = lds + e.rs_off;
  const double* yd = lds + e.yd_off;
  double* wv = lds + e.wv_off;
  auto slot = [&](int sidx) { return SA + (size_t)opaque(sidx) * TS; };
  gdouble* A = e.sc + e.oA;
  gdouble* Xb = e.sc + e.oX;
  gdouble* Gb = e.sc + e.oG;
  double* Dk = slot(2 * NT);
  double* Wk = slot(2 * NT + 1);
  if (wave == 0) {
    st_tile(Wk, o, ldg(A, lane));
    LWAIT();
    diag16(ld_tile_sym64(Wk, o), Wk, Dk, lane, 0, bad, o);
    LWAIT();
    stg(A, lane, ld_tile(Wk, o));
  }
  GSYNC();
  auto extra = [&](int k) {                    // waves 1..7; block row k of the factor (MODE 1) / of its inverse (MODE 2) is in LDS
    if (MODE == 0) return;
    auto gram = [&](v4d acc, int i, int j) {
      if (MODE == 1) {
        const double* ta = slot(NT + i);
        const double* tb = slot(NT + j);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double scl = rs[16 * k + o.kq(q)];
          acc = MFMA(ta[o.cr[q]] * scl, tb[o.cr[q]] * scl, acc);
        }
        return acc;
      }
      return mma<true, false>(acc, slot(i), slot(j), o);
    };
    int i = 0, j = 0;
    unpack_step(i, j, wave - 1);
    const v4d zero = {0, 0, 0, 0};
#pragma unroll 1
    for (int p = wave - 1; p < e.NTT; p += 2 * (NW - 1)) {     // two of this wave's tiles per trip, both requested up front
      if (i > k) break;                        // packed order: every later tile has i > k too
      int i1 = i, j1 = j;
      unpack_step(i1, j1, NW - 1);
      const bool two = p + (NW - 1) < e.NTT && i1 <= k;
      gdouble* g0 = Gb + (size_t)p * TS;
      gdouble* g1 = Gb + (size_t)(two ? p + (NW - 1) : p) * TS;
      v4d a0 = ldg(g0, lane), a1 = ldg(g1, lane);
      a0 = gram(k > i ? a0 : zero, i, j);      // first touched by block row k == i
      stg(g0, lane, a0);
      if (two) {
        a1 = gram(k > i1 ? a1 : zero, i1, j1);
        stg(g1, lane, a1);
      }
      i = i1; j = j1;
      unpack_step(i, j, NW - 1);
    }
    if (MODE == 1) {                           // wv_j += L_kj^T (y / nu)_k, j <= k: every j has one owner wave
#pragma unroll 1
      for (int jj = wave - 1; jj <= k; jj += NW - 1) {
        const double s = tile_mv(slot(NT + jj), true, yd + 16 * k, o);
        if (o.lk() == 0) wv[16 * jj + o.lr()] += s;
      }
    }
  };
#pragma unroll 1
  for (int k = 0; k < NT; ++k) {
    const int below = NT - 1 - k;
    const int n_inv = WITH_INV ? k : 0, n_row = MODE == 1 ? k + 1 : 0;
#pragma unroll 1
    for (int q = wave; q < below + n_inv + n_row; q += NW) {
      if (q < below) {                                    // panel: L_ik = A_ik inv_kk^T
        const int i = k + 1 + q;
        gdouble* t = A + lt(i, k) * TS;
        double* s = slot(i);
        st_tile(s, o, ldg(t, lane));
        LWAIT();
        v4d acc = {0, 0, 0, 0};
        acc = mma<false, true>(acc, s, Dk, o);
        stg(t, lane, acc);
        LWAIT();                                          // (this wave's operand reads of the slot are done)
        st_tile(s, o, acc);
      } else if (q < below + n_inv) {                     // X_ki = inv_kk T_ki (T_ki is the B operand as it lies in the scratch)
        const int i = q - below;
        gdouble* t = Xb + lt(k, i) * TS;
        v4d acc = {0, 0, 0, 0};
        acc = mma_nb(acc, Dk, ldg(t, lane), o);
        stg(t, lane, acc);
        st_tile(slot(i), o, acc);
      } else {                                            // block row k of the finished factor -> slots NT + i
        const int i = q - below - n_inv;
        st_tile(slot(NT + i), o, ldg(A + lt(k, i) * TS, lane));
      }
    }
    if (WITH_INV && wave == NW - 1) {                     // X_kk = inv_kk
      const v4d d = ld_tile(Dk, o);
      stg(Xb + lt(k, k) * TS, lane, d);
      st_tile(slot(k), o, d);
    }
    GSYNC();
    if (below == 0) {
      if (wave != 0) extra(k);
      break;
    }
    if (wave == 0) {                                      // look-ahead: next diagonal tile, then its factorisation
      gdouble* t = A + lt(k + 1, k + 1) * TS;
      st_tile(Wk, o, ldg(t, lane));
      LWAIT();
      v4d acc = ld_tile_sym(Wk, o);
      acc = mma<false, true>(acc, slot(k + 1), slot(k + 1), o, -1);
      LWAIT();
      diag16(acc, Wk, Dk, lane, 16 * (k + 1), bad, o);   // (nobody reads inv_kk in this phase: inv_k+1,k+1 takes its slot)
      LWAIT();
      stg(t, lane, ld_tile(Wk, o));
    } else {
      // Two tile updates at a time: both output tiles are requested before either product runs.  The trailing tiles (i, j),
      // i = k+2 .. NT-1, j = k+1 .. i, and behind them the inverse's tiles (j, i), j = k+1 .. NT-1, i = 0 .. k, form ONE
      // sequence that is dealt round-robin over waves 1..7; a wave steps through its own positions directly (seven
      // positions at a time) instead of scanning the whole sequence with a modulo test - the scan was ~300 scalar cycles
      // per update.
      auto trailing2 = [&](int i0, int j0, int i1, int j1, bool two) {         // A_ij -= L_ik L_jk^T
        gdouble* t0 = A + lt(i0, j0) * TS;
        gdouble* t1 = A + lt(i1, j1) * TS;
        v4d a0 = ldg(t0, lane), a1 = ldg(t1, lane);
        a0 = mma<false, true>(a0, slot(i0), slot(j0), o, -1);
        stg(t0, lane, a0);
        if (two) {
          a1 = mma<false, true>(a1, slot(i1), slot(j1), o, -1);
          stg(t1, lane, a1);
        }
      };
      const int n_trail = (below - 1) * (below + 2) / 2;                       // sum_{i = k+2}^{NT-1} (i - k)
      // (wave 4 shares its SIMD with wave 0, whose fp64 chain waits out every MFMA the partner has in the pipe - section 5b:
      //  it takes no tile updates in this phase, the other six do)
      const int NWK = w4idle ? NW - 2 : NW - 1;
      int pos = (w4idle && wave > 4) ? wave - 2 : wave - 1;
      if (w4idle && wave == 4) pos = 1 << 20;
      {
        int i = k + 2, j = k + 1 + pos;                                        // row i holds i - k tiles
        while (i < NT && j > i) { j -= i - k; ++i; }
        int have = 0, hi = 0, hj = 0;
#pragma unroll 1
        while (pos < n_trail) {
          if (!have) { hi = i; hj = j; have = 1; }
          else { trailing2(hi, hj, i, j, true); have = 0; }
          pos += NWK;
          j += NWK;
          while (i < NT && j > i) { j -= i - k; ++i; }
        }
        if (have) trailing2(hi, hj, hi, hj, false);
      }
      if (WITH_INV && !(w4idle && wave == 4)) {
        auto inverse2 = [&](int j0, int i0, int j1, int i1, bool two) {        // T_ji -= L_jk X_ki   (first touched at k == i)
          gdouble* t0 = Xb + lt(j0, i0) * TS;
          gdouble* t1 = Xb + lt(j1, i1) * TS;
          v4d a0 = ldg(t0, lane), a1 = ldg(t1, lane);
          const v4d zero = {0, 0, 0, 0};
          a0 = mma<false, false>(k > i0 ? a0 : zero, slot(j0), slot(i0), o, -1);
          stg(t0, lane, a0);
          if (two) {
            a1 = mma<false, false>(k > i1 ? a1 : zero, slot(j1), slot(i1), o, -1);
            stg(t1, lane, a1);
          }
        };
        const int n_inv_ops = below * (k + 1);
        int q = pos - n_trail;                                                 // position inside the inverse's part of the sequence
        int j = k + 1, i = q;                                                  // (j, i) = (k + 1 + q / (k + 1), q % (k + 1)) without a division: a runtime
        while (i > k) { i -= k + 1; ++j; }                                     //  divisor costs a float reciprocal and a register that lives across the loop
        int have = 0, hi = 0, hj = 0;
#pragma unroll 1
        while (q < n_inv_ops) {
          if (!have) { hi = i; hj = j; have = 1; }
          else { inverse2(hj, hi, j, i, true); have = 0; }
          q += NWK;
          i += NWK;
          while (i > k) { i -= k + 1; ++j; }
        }
        if (have) inverse2(hj, hi, hj, hi, false);
      }
      extra(k);
    }
    GSYNC();
  }
  GSYNC();
  return bad;
}

// ---- product phases: out_p = sum_k contrib_k for the packed output tiles p of a triangle (or a full square), PASS_TILES at a
// time.  stage(k): the waves fill LDS slots for step k; use(k, i, j, acc): adds step k's contribution to the accumulator
// of output tile (i, j); done(p, i, j, acc): epilogue + store.  `square`: outputs are the NT x NT tiles (i, j) = (p / NT, p % NT).
template <class Stage, class Use, class Done>
__device__ __forceinline__ void staged_product(const Ctx& c, int n_out, bool square, int k0, int k1, Stage stage, Use use, Done done) {
#pragma unroll 1
  for (int base = 0; base < n_out; base += PASS_TILES) {
    v4d acc[TPW];
    int ti[TPW], tj[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      acc[s] = v4d{0, 0, 0, 0};
      const int p = base + c.wave + s * NW;
      int i = 0, j = 0;
      if (p < n_out) {
        if (square) { j = p; while (j >= c.NT) { j -= c.NT; ++i; } }        // (p / NT, p % NT) without a runtime division
        else unpack_step(i, j, p);
      }
      ti[s] = p < n_out ? i : -1;
      tj[s] = j;
    }
#pragma unroll 1
    for (int k = k0; k < k1; ++k) {
      stage(k);
      GSYNC();
#pragma unroll
      for (int s = 0; s < TPW; ++s)
        if (ti[s] >= 0) acc[s] = use(k, ti[s], tj[s], acc[s]);
      GSYNC();
    }
#pragma unroll
    for (int s = 0; s < TPW; ++s)
      if (ti[s] >= 0) done(base + c.wave + s * NW, ti[s], tj[s], acc[s]);
  }
  GSYNC();
}

// ---- the fit as a sequence of SEPARATELY COMPILED phases (round 5).  One kernel body with every phase inlined keeps dozens of
// pointers, offsets and constants alive across all of them: 276 spilled scalar and 60 spilled vector registers, 200 bytes of private
// segment per lane.  Each phase below is a real function (`__noinline__`) that rebuilds the few addresses it needs from the kernel
// argument block - read where it lies, in the kernarg segment, through scalar loads - and the member index; the kernel itself holds
// the member loop, the iteration counter and the pivot status.  (Calls are where the one code-generation defect of this compiler was
// met, section 5c: tools/check_exec_prologue.py scans every build.)
typedef const __attribute__((address_space(4))) MediumArgs* KArgs;
__device__ __forceinline__ KArgs uni_k(KArgs p) { return (KArgs)uni((long)p); }

#define MED_ENV(g_in, b_in) \
  const KArgs gk_ = uni_k(g_in); \
  const __attribute__((address_space(4))) MediumArgs& g = *gk_; \
  [[maybe_unused]] const int b = uni(b_in); \
  const int NT = g.NT, NTT = NT * (NT + 1) / 2, NP = 16 * NT; \
  double* lds = lds_base(); \
  double* ssq = lds; \
  double* misc = ssq + NP; \
  double* yv = misc + 64; \
  double* dv = yv + NP; \
  double* rs = dv + NP; \
  double* yd = rs + NP; \
  double* th = yd + NP; \
  double* mv = th + NP; \
  double* av = mv + NP; \
  double* wv = av + NP; \
  double* uv = wv + NP; \
  double* SA = uv + NP; \
  double* sc = g.scratch + (size_t)blockIdx.x * g.scratch_stride; \
  double* Lb = sc; \
  double* Cb = Lb + (size_t)NTT * TS; \
  double* Wb = Cb + (size_t)NTT * TS; \
  double* Gb = Wb + (size_t)NTT * TS; \
  double* LamB = Gb + (size_t)NTT * TS; \
  double* Sb = LamB + (size_t)NTT * TS; \
  double* K1b = Sb + (size_t)NTT * TS; \
  double* K2b = K1b + (size_t)NTT * TS; \
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); \
  Ctx c; \
  c.NT = NT; c.NTT = NTT; c.wave = wave; c.lane = lane; c.tid = tid; c.SA = SA; \
  lane_init(c.o, lane); \
  const Lane& o = c.o; \
  const int N = g.N; \
  const double J = g.jitter; \
  auto lower_matvec = [&](const double* Lmat, const double* x, double beta, double* out) { \
    double part[2] = {0.0, 0.0}; \
  _Pragma("unroll 1") \
    for (int k = 0; k < NT; ++k) { \
  _Pragma("unroll 1") \
      for (int i = k + wave; i < NT; i += NW) st_tile(c.slot(i), o, ldn(Lmat + lt(i, k) * TS, lane)); \
      GSYNC(); \
  _Pragma("unroll") \
      for (int u = 0; u < 2; ++u) { \
        const int i = wave + u * NW; \
        if (i < NT && i >= k) part[u] += tile_mv(c.slot(i), false, x + 16 * k, o); \
      } \
      GSYNC(); \
    } \
  _Pragma("unroll") \
    for (int u = 0; u < 2; ++u) { \
      const int i = wave + u * NW; \
      if (i < NT && o.lk() == 0) out[16 * i + o.lr()] = part[u] + beta * yv[16 * i + o.lr()]; \
    } \
    GSYNC(); \
  }; \
  auto stage_at_col = [&](int k, int slot0) { \
  _Pragma("unroll 1") \
    for (int i = wave; i < NT; i += NW) { \
      double* s = c.slot(slot0 + i); \
      if (i > k) st_tile(s, o, ldn(Lb + lt(i, k) * TS, lane)); \
      else { \
        v4d x = ldn(Wb + lt(k, i) * TS, lane); \
  _Pragma("unroll") \
        for (int r = 0; r < 4; ++r) x[r] *= -J; \
        st_tile_t(s, o, x); \
        if (i == k) { \
          const v4d l = ldn(Lb + lt(k, k) * TS, lane); \
          LWAIT(); \
  _Pragma("unroll") \
          for (int r = 0; r < 4; ++r) s[o.cr[r]] += l[r]; \
        } \
      } \
    } \
  }; \
  auto stage_cross = [&](const double* Sym, int k, int slot0) { \
    const int i0 = wave, i1 = wave + NW; \
    const bool h0 = i0 < NT, h1 = i1 < NT; \
    const int c0 = h0 ? i0 : 0, c1 = h1 ? i1 : c0; \
    const v4d a0 = ldn(Sym + (size_t)(c0 >= k ? lt(c0, k) : lt(k, c0)) * TS, lane); \
    const v4d a1 = ldn(Sym + (size_t)(c1 >= k ? lt(c1, k) : lt(k, c1)) * TS, lane); \
    if (h0) st_tile(c.slot(slot0 + i0), o, a0); \
    if (h1) st_tile(c.slot(slot0 + i1), o, a1); \
  }; \
  [[maybe_unused]] const int R = g.Rv ? g.Rv[b] : g.R; \
  [[maybe_unused]] const double* X = g.X + (g.xoff ? g.xoff[b] : (long)b * N * g.R); \
  (void)ssq; (void)misc; (void)yv; (void)dv; (void)rs; (void)yd; (void)th; (void)mv; (void)av; (void)wv; (void)uv; (void)SA; \
  (void)Lb; (void)Cb; (void)Wb; (void)Gb; (void)LamB; (void)Sb; (void)K1b; (void)K2b; (void)J; (void)o; \
  (void)lower_matvec; (void)stage_at_col; (void)stage_cross

__device__ __noinline__ void ph_setup(KArgs g_in, int b_in) {
  MED_ENV(g_in, b_in);
  // ---- member set-up ----
  for (int t = tid; t < NP; t += NTH) {
    const double nu = t < N ? GLD(g.nu[(long)b * N + t]) : 1.0;
    const double y = t < N ? GLD(g.y[(long)b * N + t]) : 0.0;
    yv[t] = y; dv[t] = 1.0 / nu; rs[t] = 1.0 / sqrt(nu); yd[t] = y / nu;
    th[t] = 0; mv[t] = 0;
    double sq = 0.0;
    if (t < N)
      for (int d = 0; d < R; ++d) { const double v = GLD(X[(long)t * R + d]); sq = fma(v, v, sq); }
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
      for (int r = 0; r < 4; ++r) id[r] = (i == j && o.kq(r) == o.lr()) ? 1.0 : 0.0;
      stn(LamB + (size_t)p * TS, lane, id);            // Lam = I: q_sqrt = I
      stn(Sb + (size_t)p * TS, lane, id);              // S = I until the first natural-gradient step (n_iters = 0)
      unpack_step(i, j, NW);
    }
  }
  GSYNC();

}

__device__ __noinline__ void ph_kmat(KArgs g_in, int b_in, int keep, int zero_wv) {
  MED_ENV(g_in, b_in);
  auto refresh_hypers = [&]() {
    if (tid == 0) {
      const double ls = softplus_d(misc[1]);
      misc[8] = softplus_d(misc[0]); misc[9] = ls; misc[10] = 1.0 / (ls * ls);
    }
    GSYNC();
  };
  // Lb <- tril(K0) + J I (identity in the padding)
  // `keep`: the unit-variance kernel values (k1, k2) of every tile are parked in the scratch for this iteration's hyper-gradient
  auto kmat = [&](bool keep) {
    const double var = misc[8], inv_ls2 = misc[10];
    int i = 0, j = 0;
    unpack_step(i, j, wave);
#pragma unroll 1
    for (int p = wave; p < NTT; p += NW) {
      const v4d r2 = r2_tile(X, N, R, inv_ls2, ssq, i, j, o);
      v4d kt, k1t, k2t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + o.kq(r), col = 16 * j + o.lr();
        double k1, k2;
        kern_unit<true>(g.kernel, r2[r], k1, k2);
        double K = var * k1;
        if (row >= N || col >= N) K = row == col ? 1.0 : 0.0;
        else K = col <= row ? K + (row == col ? J : 0.0) : 0.0;
        kt[r] = K;
        k1t[r] = (row < N && col < N) ? k1 : 0.0;         // (the padding contributes no gradient)
        k2t[r] = (row < N && col < N) ? k2 : 0.0;
      }
      stn(Lb + (size_t)p * TS, lane, kt);
      if (keep) { stn(K1b + (size_t)p * TS, lane, k1t); stn(K2b + (size_t)p * TS, lane, k2t); }
      unpack_step(i, j, NW);
    }
    GSYNC();
  };
  refresh_hypers();
  kmat(keep != 0);
  if (zero_wv) {
    for (int t = tid; t < NP; t += NTH) wv[t] = 0.0;
    GSYNC();
  }
}

__device__ __noinline__ void ph_neg_elbo(KArgs g_in, int b_in, int slot_in, int save_wv) {
  MED_ENV(g_in, b_in);
  // -ELBO sample (models.py:211-215) at the hypers whose factor L is in Lb, with the (m, S) of the last natural-gradient
  // step (S in Sb, log det C in misc[11]):
  //   sum_t [ log(2 pi)/2 + log(nu_t)/2 + ((y_t - (L m)_t)^2 + (L S L^T)_tt) / (2 nu_t) ] + (|m|^2 - N + 2 log det C + tr S) / 2
  // (L S L^T)_tt = sum_j (L S)_tj L_tj over the lower tiles of L S (a staged lower x symmetric product); uses av and wv
  auto neg_elbo = [&](int slot) {
    lower_matvec(Lb, mv, -1.0, av);                                      // av = L m - y
    for (int t = tid; t < NP; t += NTH) wv[t] = 0.0;
    GSYNC();
    staged_product(c, NTT, false, 0, NT,
      [&](int k) {
#pragma unroll 1
        for (int i = k + wave; i < NT; i += NW) st_tile(c.slot(i), o, ldn(Lb + lt(i, k) * TS, lane));
        stage_cross(Sb, k, NT);
      },
      [&](int k, int i, int j, v4d acc) {
        if (k > i) return acc;
        return mma_rt(acc, c.slot(i), false, c.slot(NT + j), j > k, o);   // L_ik S_kj ; S_kj = slot j as it lies for j <= k... see stage_cross
      },
      [&](int p, int i, int j, v4d acc) {
        // row sums of (L S)_ij o L_ij, one 16-vector per tile, parked in Cb (dead here: Phi of the previous iteration) and
        // added up below in a fixed order - no atomics, the sample is reproducible bit for bit
        const v4d lt_ = ldn(Lb + lt(i, j) * TS, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double v = acc[r] * lt_[r];
          v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
          if (o.lr() == 0) GST(Cb[(size_t)p * 16 + o.kq(r)], v);
        }
      });
    GSYNC();
    for (int t = tid; t < NP; t += NTH) {
      const int i = t >> 4;
      double f = 0.0;
      for (int j = 0; j <= i; ++j) f += GLD(Cb[(size_t)lt(i, j) * 16 + (t & 15)]);
      wv[t] = f;
    }
    GSYNC();
    double part = 0.0;
    for (int t = tid; t < N; t += NTH) {
      const double nu = 1.0 / dv[t], a = av[t], m = mv[t];
      const double stt = GLD(Sb[lt(t >> 4, t >> 4) * TS + nidx(t & 15, t & 15)]);
      part += 0.5 * 1.8378770664093453 + 0.5 * log(nu) + 0.5 * (a * a + wv[t]) * dv[t] + 0.5 * (m * m + stt);
    }
    part = wave_sum(part);
    if (lane == 0) misc[16 + wave] = part;
    GSYNC();
    if (tid == 0) {
      double tot = 0.0;
      for (int w = 0; w < NW; ++w) tot += misc[16 + w];
      GST(g.loss[(long)b * g.loss_stride + slot], tot - 0.5 * N + misc[11]);
    }
    GSYNC();
  };

  const int slot = uni(slot_in);
  if (save_wv) {                                                        // (the sample uses wv as scratch: keep L^T (y / nu) aside)
    for (int t = tid; t < NP; t += NTH) uv[t] = wv[t];
    GSYNC();
  }
  neg_elbo(slot);
  if (save_wv) {
    for (int t = tid; t < NP; t += NTH) wv[t] = uv[t];
    GSYNC();
  }
}

__device__ __noinline__ void ph_natgrad(KArgs g_in, int b_in) {
  MED_ENV(g_in, b_in);
  // ---- natural-gradient step: Lam <- (1-g) Lam + g (I + G) -> LamB and Cb; theta1 <- (1-g) theta1 + g L^T (y / nu)
  {
    int i = 0, j = 0;
    unpack_step(i, j, wave);
#pragma unroll 1
    for (int p = wave; p < NTT; p += NW) {
      v4d lam = ldn(LamB + (size_t)p * TS, lane);
      const v4d gt = ldn(Gb + (size_t)p * TS, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        lam[r] = (1.0 - g.gamma) * lam[r] + g.gamma * (gt[r] + ((i == j && o.kq(r) == o.lr()) ? 1.0 : 0.0));
      stn(LamB + (size_t)p * TS, lane, lam);
      stn(Cb + (size_t)p * TS, lane, lam);
      unpack_step(i, j, NW);
    }
  }
  for (int t = tid; t < NP; t += NTH) th[t] = (1.0 - g.gamma) * th[t] + g.gamma * wv[t];
  GSYNC();
}

__device__ __noinline__ void ph_logdet_c(KArgs g_in, int b_in) {
  MED_ENV(g_in, b_in);
  double s = 0.0;
  for (int t = tid; t < N; t += NTH) s += log(GLD(Cb[lt(t >> 4, t >> 4) * TS + nidx(t & 15, t & 15)]));
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

__device__ __noinline__ void ph_m(KArgs g_in, int b_in) {
  MED_ENV(g_in, b_in);
  // ---- m = C^-T (C^-1 theta1) through the inverse FACTOR (Lb), not S theta1: see small.hip - with a variance of ~1e-11 at
  // one time step the rounding of the explicit S puts 1e-5 into m, which u = G m - L^T (y / nu) multiplies by |G| ~ 1e11.
  lower_matvec(Lb, th, 0.0, uv);                                        // z = C^-1 theta1 (one pass over the block columns)
  {
    double part[2] = {0.0, 0.0};
#pragma unroll 1
    for (int k = 0; k < NT; ++k) {                                      // m_j = sum_{k >= j} X_kj^T z_k: one pass over the block rows
#pragma unroll 1
      for (int j = wave; j <= k; j += NW) st_tile(c.slot(j), o, ldn(Lb + lt(k, j) * TS, lane));
      GSYNC();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = wave + u * NW;
        if (j < NT && j <= k) part[u] += tile_mv(c.slot(j), true, uv + 16 * k, o);
      }
      GSYNC();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = wave + u * NW;
      if (i < NT && o.lk() == 0) mv[16 * i + o.lr()] = part[u];
    }
    GSYNC();
  }
}

__device__ __noinline__ void ph_phi(KArgs g_in, int b_in, int first_in) {
  MED_ENV(g_in, b_in);
  const int first = uni(first_in);
  // ---- Phi(L^T Lbar) = tril(u m^T + G S), half the diagonal -> Cb (C is dead).  u = L^T D (L m - y) = G m - L^T (y / nu) comes
  // out of the same pass (the crosses of G are in LDS anyway) and, like m, enters Phi only in the epilogue.
  if (first && g.gamma > 0.0) {
    // First iteration: Lam_old = I exactly, so Lam = I + gamma G, G S = (Lam - I) Lam^-1 / gamma = (I - S) / gamma and
    // u = G m - L^T (y / nu) = (Lam m - m) / gamma - theta1 / gamma = -m / gamma: no product, one pass over the tiles of S.
    // (not (theta1 - m) / gamma - L^T (y / nu): with |theta1| ~ 1e11 that difference rounds m away)
    for (int t = tid; t < NP; t += NTH) uv[t] = -mv[t] / g.gamma;
    GSYNC();
    int i = 0, j = 0;
    unpack_step(i, j, wave);
#pragma unroll 1
    for (int p = wave; p < NTT; p += NW) {
      v4d a = ldn(Sb + (size_t)p * TS, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = o.kq(r);
        double v = fma(uv[16 * i + rl], mv[16 * j + o.lr()], (((i == j && rl == o.lr()) ? 1.0 : 0.0) - a[r]) / g.gamma);
        if (i == j) v = rl < o.lr() ? 0.0 : (rl == o.lr() ? 0.5 * v : v);
        a[r] = v;
      }
      stn(Cb + (size_t)p * TS, lane, a);
      unpack_step(i, j, NW);
    }
    GSYNC();
  } else {
    double gm[2] = {0.0, 0.0};
#pragma unroll 1
    for (int base = 0; base < NTT; base += PASS_TILES) {
      v4d acc[TPW];
      int ti[TPW], tj[TPW];
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        acc[s] = v4d{0, 0, 0, 0};
        const int p = base + wave + s * NW;
        int i = 0, j = 0;
        if (p < NTT) unpack_step(i, j, p);
        ti[s] = p < NTT ? i : -1;
        tj[s] = j;
      }
#pragma unroll 1
      for (int k = 0; k < NT; ++k) {
        stage_cross(Gb, k, 0);
        stage_cross(Sb, k, NT);
        GSYNC();
#pragma unroll
        for (int s = 0; s < TPW; ++s)
          if (ti[s] >= 0)                                              // G_ik S_kj
            acc[s] = mma_rt(acc[s], c.slot(ti[s]), ti[s] < k, c.slot(NT + tj[s]), tj[s] > k, o);
        if (base == 0) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int i = wave + u * NW;
            if (i < NT) gm[u] += tile_mv(c.slot(i), i < k, mv + 16 * k, o);
          }
        }
        GSYNC();
      }
      if (base == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = wave + u * NW;
          if (i < NT && o.lk() == 0) uv[16 * i + o.lr()] = gm[u] - wv[16 * i + o.lr()];
        }
        GSYNC();
      }
#pragma unroll
      for (int s = 0; s < TPW; ++s)
        if (ti[s] >= 0) {
          const int i = ti[s], j = tj[s];
          v4d a = acc[s];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rl = o.kq(r);
            double v = fma(uv[16 * i + rl], mv[16 * j + o.lr()], a[r]);
            if (i == j) v = rl < o.lr() ? 0.0 : (rl == o.lr() ? 0.5 * v : v);
            a[r] = v;
          }
          stn(Cb + (size_t)(base + wave + s * NW) * TS, lane, a);
        }
    }
    GSYNC();
  }
}

__device__ __noinline__ void ph_w(KArgs g_in, int b_in) {
  MED_ENV(g_in, b_in);
  // ---- W = Phi L^-1 (lower; Phi in Cb, L^-1 in Wb) -> Lb (C^-1 is dead): W_ij = sum_{k = j..i} Phi_ik Linv_kj
  staged_product(c, NTT, false, 0, NT,
    [&](int k) {
#pragma unroll 1
      for (int q = wave; q < NT + 1; q += NW) {
        if (q < NT - k) st_tile(c.slot(k + q), o, ldn(Cb + lt(k + q, k) * TS, lane));               // Phi_ik, i >= k -> slot i
        else st_tile(c.slot(NT + (q - (NT - k))), o, ldn(Wb + lt(k, q - (NT - k)) * TS, lane));      // Linv_kj, j <= k -> slot NT + j
      }
    },
    [&](int k, int i, int j, v4d acc) {
      if (k < j || k > i) return acc;
      return mma_rt(acc, c.slot(i), false, c.slot(NT + j), false, o);
    },
    [&](int p, int, int, v4d acc) { stn(Lb + (size_t)p * TS, lane, acc); });
}

__device__ __noinline__ void ph_kbar_adam(KArgs g_in, int b_in) {
  MED_ENV(g_in, b_in);
  // ---- Kbar = L^-T W ; g = <Kbar, dK/dtheta> over the real part ; Adam   (models.py:210).  dK is symmetric: Kbar_ij and Kbar_ji^T
  // = sum_k W_ki^T Linv_kj are accumulated together and the kernel function is evaluated once per unordered pair
  {
    const double var = misc[8], ls = misc[9];
    double gv = 0.0, gl = 0.0;
    staged_product(c, NTT, false, 0, NT,
      [&](int k) {
#pragma unroll 1
        for (int q = wave; q < 2 * (k + 1); q += NW) {
          if (q <= k) st_tile(c.slot(q), o, ldn(Wb + lt(k, q) * TS, lane));                          // Linv_kj -> slot j
          else st_tile(c.slot(NT + q - (k + 1)), o, ldn(Lb + lt(k, q - (k + 1)) * TS, lane));        // W_kj -> slot NT + j
        }
      },
      [&](int k, int i, int j, v4d acc) {
        if (k < i) return acc;
        acc = mma_rt(acc, c.slot(i), true, c.slot(NT + j), false, o);                                // Linv_ki^T W_kj
        if (i != j) acc = mma_rt(acc, c.slot(NT + i), true, c.slot(j), false, o);                    // W_ki^T Linv_kj
        return acc;
      },
      [&](int p, int, int, v4d acc) {
        // <Kbar, k1>, <Kbar, k2> with the kernel values this iteration's kmat parked (zero in the padding)
        const v4d k1t = ldn(K1b + (size_t)p * TS, lane), k2t = ldn(K2b + (size_t)p * TS, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) { gv = fma(acc[r], k1t[r], gv); gl = fma(acc[r], k2t[r], gl); }
      });
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
}

__device__ __noinline__ void ph_mu_e(KArgs g_in, int b_in) {
  MED_ENV(g_in, b_in);
  // mu = A^T m
  {
    double part[2] = {0.0, 0.0};
#pragma unroll 1
    for (int k = 0; k < NT; ++k) {
      stage_at_col(k, 0);
      GSYNC();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = wave + u * NW;
        if (i < NT) part[u] += tile_mv(c.slot(i), false, mv + 16 * k, o);
      }
      GSYNC();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = wave + u * NW;
      if (i < NT && o.lk() == 0) av[16 * i + o.lr()] = part[u];              // av = mu (padded part 0)
    }
    GSYNC();
  }
  // E = A^T S, all NT x NT tiles -> the square buffer at Gb (G and Lam are dead): E_il = sum_k A^T(i, k) S_kl
  double* Eb = Gb;
  staged_product(c, NT * NT, true, 0, NT,
    [&](int k) { stage_at_col(k, 0); stage_cross(Sb, k, NT); },
    [&](int k, int i, int l, v4d acc) { return mma_rt(acc, c.slot(i), false, c.slot(NT + l), l > k, o); },
    [&](int p, int, int, v4d acc) { stn(Eb + (size_t)p * TS, lane, acc); });
}

__device__ __noinline__ void ph_cov(KArgs g_in, int b_in) {
  MED_ENV(g_in, b_in);
  double* Eb = Gb;
  // cov = diag(nu + J) - J^2 L^-T L^-1 + E A: lower tiles (i, j): -J^2 sum_{k >= i} Linv_ki^T Linv_kj, then sum_l E_il A^T(j, l)^T.
  // The finished tiles go, bordered by the rows (1^T, mu^T) in the padding, into Cb for the factorisation that follows.
#pragma unroll 1
  for (int base = 0; base < NTT; base += PASS_TILES) {
    v4d acc[TPW];
    int ti[TPW], tj[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      acc[s] = v4d{0, 0, 0, 0};
      const int p = base + wave + s * NW;
      int i = 0, j = 0;
      if (p < NTT) unpack_step(i, j, p);
      ti[s] = p < NTT ? i : -1;
      tj[s] = j;
    }
#pragma unroll 1
    for (int k = 0; k < NT; ++k) {                                       // block row k of L^-1 -> slots j <= k
#pragma unroll 1
      for (int q = wave; q <= k; q += NW) st_tile(c.slot(q), o, ldn(Wb + lt(k, q) * TS, lane));
      GSYNC();
#pragma unroll
      for (int s = 0; s < TPW; ++s)
        if (ti[s] >= 0 && k >= ti[s]) acc[s] = mma_rt(acc[s], c.slot(ti[s]), true, c.slot(tj[s]), false, o);
      GSYNC();
    }
#pragma unroll
    for (int s = 0; s < TPW; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[s][r] *= -J * J;
#pragma unroll 1
    for (int l = 0; l < NT; ++l) {                                       // block column l of E (slots i) and of A^T (slots NT + j)
#pragma unroll 1
      for (int i = wave; i < NT; i += NW) st_tile(c.slot(i), o, ldn(Eb + (size_t)(i * NT + l) * TS, lane));
      stage_at_col(l, NT);
      GSYNC();
#pragma unroll
      for (int s = 0; s < TPW; ++s)
        if (ti[s] >= 0) acc[s] = mma_rt(acc[s], c.slot(ti[s]), false, c.slot(NT + tj[s]), true, o);
      GSYNC();
    }
#pragma unroll
    for (int s = 0; s < TPW; ++s)
      if (ti[s] >= 0) {
        const int i = ti[s], j = tj[s];
        v4d a = acc[s], t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * i + o.kq(r), col = 16 * j + o.lr();
          if (row == col) a[r] = a[r] + (1.0 / dv[row] + J);
          t[r] = a[r];
          if (row < N && col < N) {
            // (a diagonal tile is a full 16x16 accumulation: its lower half is exported and mirrored, so that cov is exactly symmetric)
            if (g.cov && (i != j || col <= row)) {
              GST(g.cov[((long)b * N + row) * N + col], a[r]);
              GST(g.cov[((long)b * N + col) * N + row], a[r]);
            }
            if (g.var && row == col) GST(g.var[(long)b * N + row], a[r]);
          } else {
            // bordered matrix for chol(cov): rows N (1^T) and N+1 (mu^T) in the padding, huge pivots there, identity elsewhere
            double v = row == col ? ((row == N || row == N + 1) ? 1e30 : 1.0) : 0.0;
            if (col < N && row == N) v = 1.0;
            if (col < N && row == N + 1) v = av[col];
            t[r] = v;
          }
        }
        stn(Cb + (size_t)(base + wave + s * NW) * TS, lane, t);
      }
  }
  if (g.mu)
    for (int t = tid; t < N; t += NTH) GST(g.mu[(long)b * N + t], av[t]);
  GSYNC();
}

__device__ __noinline__ void ph_export(KArgs g_in, int b_in, int factored, int bad_in) {
  MED_ENV(g_in, b_in);
  if (factored) {
    if (g.chol) {
      // every tile (i, j) of the N x N factor, the tiles dealt over the waves: the lower ones are read as they lie in the scratch (one
      // coalesced 2 KB load), the upper ones are zeros; a wave instruction stores four 128-byte row segments
      int i = 0, j = wave;
      while (j >= NT) { j -= NT; ++i; }
  #pragma unroll 1
      for (int p = wave; p < NT * NT; p += NW) {
        v4d t = {0, 0, 0, 0};
        if (j <= i) t = ldn(Cb + (size_t)lt(i, j) * TS, lane);
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * i + o.kq(r), col = 16 * j + o.lr();
          if (row < N && col < N) GST(g.chol[((long)b * N + row) * N + col], col <= row ? t[r] : 0.0);
        }
        j += NW;
        while (j >= NT) { j -= NT; ++i; }
      }
    }
    if (g.basis) {
      for (int e = tid; e < 2 * N; e += NTH) {
        const int which = e >= N ? 1 : 0, cc = e - which * N, row = N + which;
        GST(g.basis[((long)b * 2 + which) * N + cc], GLD(Cb[lt(row >> 4, cc >> 4) * TS + nidx(row & 15, cc & 15)]));
      }
    }
    if (g.logdet) {
      double s = 0.0;
      for (int t = tid; t < N; t += NTH) s += log(GLD(Cb[lt(t >> 4, t >> 4) * TS + nidx(t & 15, t & 15)]));
      s = wave_sum(s);
      if (lane == 0) misc[16 + wave] = s;
      GSYNC();
      if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < NW; ++w) tot += misc[16 + w];
        GST(g.logdet[b], tot);
      }
    }
  }
  if (tid == 0 && g.hyp) { GST(g.hyp[(long)b * 2], misc[8]); GST(g.hyp[(long)b * 2 + 1], misc[9]); }
  if (wave == 0 && lane == 0) GST(g.info[b], uni(bad_in));
  GSYNC();                                                                // LDS and the scratch are re-initialised for the next member
}

__global__ __launch_bounds__(NTH, 2) void medium_fit_kernel(MediumArgs) {
  // the argument block where it lies (the kernel's one by-value parameter sits at the start of the kernarg segment)
  const KArgs gk = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  const __attribute__((address_space(4))) MediumArgs& g = *gk;
  const int tid = threadIdx.x;
  const int NT = g.NT, NTT = NT * (NT + 1) / 2, NP = 16 * NT;
  // what the factorisation (potrf_v2, a function of its own since round 3) is told: the workgroup's scratch, the triangles by
  // their index (0 Lb, 1 Cb, 2 Wb, 3 Gb, 4 LamB, 5 Sb) and the LDS offsets of the vectors it uses (layout: MED_ENV)
  auto pargs = [&](int a, int x, int gm) {
    PotrfArgs e;
    e.sc = (gdouble*)(g.scratch + (size_t)blockIdx.x * g.scratch_stride);
    e.oA = (long)a * NTT * TS; e.oX = (long)x * NTT * TS; e.oG = (long)gm * NTT * TS; e.NT = NT; e.NTT = NTT;
    e.rs_off = NP + 64 + 2 * NP; e.yd_off = e.rs_off + NP; e.wv_off = NP + 64 + 7 * NP; e.sa_off = NP + 64 + 9 * NP;
    e.w4idle = (g.tune & 1) ? 1 : 0;   // off: here the tile updates (through the scratch), not the chain, bound a step (288 against 279 us at N = 165)
    return e;
  };
#pragma unroll 1
  for (int b = blockIdx.x; b < g.B; b += gridDim.x) {
    int bad = 0;
    ph_setup(gk, b);
    [[maybe_unused]] int it_now = 0;
#pragma unroll 1
    for (int it = 0; it < g.n_iters; ++it) {
      it_now = it;
      MSTAMP(0);
      ph_kmat(gk, b, 1, 1);                                                // K0 + J I at the current hypers  (models.py:209); wv = 0
      MSTAMP(1);
      // L = chol(K0 + J I), L^-1 rides along into Wb; G = L^T D L and L^T (y / nu) accumulate block row by block row behind it
      bad = potrf_v2<true, 1>(pargs(0, 2, 3), bad);
      if (g.loss && it > 0 && (it - 1) % 25 == 0) ph_neg_elbo(gk, b, (it - 1) / 25, 1);   // after the Adam step = at these hypers
      MSTAMP(2);
      ph_natgrad(gk, b);
      MSTAMP(3);
      // C = chol(Lam) in Cb, C^-1 in Lb (L is dead); S = C^-T C^-1 accumulates into Sb behind it
      bad = potrf_v2<true, 2>(pargs(1, 0, 5), bad);
      MSTAMP(4);
      if (g.loss && it % 25 == 0) ph_logdet_c(gk, b);                       // log det C of this step, for its loss sample
      ph_m(gk, b);
      MSTAMP(5);
      ph_phi(gk, b, it == 0 ? 1 : 0);
      MSTAMP(6);
      ph_w(gk, b);
      MSTAMP(7);
      ph_kbar_adam(gk, b);
      MSTAMP(8);
    }

    // ---- prediction at the final hypers with the (m, S) of the last step    (models.py:217-220)
    it_now = g.n_iters;
    MSTAMP(9);
    ph_kmat(gk, b, 0, 0);
    bad = potrf_v2<true, 0>(pargs(0, 2, 3), bad);                              // Lb = L, Wb = L^-1, Sb = S
    if (g.loss && g.n_iters > 0 && (g.n_iters - 1) % 25 == 0) ph_neg_elbo(gk, b, (g.n_iters - 1) / 25, 0);
    MSTAMP(10);
    ph_mu_e(gk, b);
    ph_cov(gk, b);
    MSTAMP(11);
    const bool factor = g.chol || g.logdet || g.basis;
    if (factor) bad = potrf_v2<false, 0>(pargs(1, 1, 3), bad);                 // Distribution.__post_init__ (data.py:38-39)
    ph_export(gk, b, factor ? 1 : 0, bad);
    MSTAMP(12);
  }
}

int medium_tiles(int N) { return (N + 2 + 15) / 16; }

bool medium_fit_supported(int N, int R) { return N > smalln::SMALL_MAX_N && N <= MEDIUM_MAX_N && R >= 1; }

size_t medium_scratch_doubles(int N) {
  const int NT = medium_tiles(N), NTT = NT * (NT + 1) / 2;
  return (size_t)8 * NTT * TS;
}

static size_t medium_lds_bytes(int N) {
  const int NT = medium_tiles(N), NP = 16 * NT;
  return ((size_t)NP + 64 + 9 * (size_t)NP + (2 * (size_t)NT + 2) * TS) * sizeof(double);
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
  static const int tune = [] { const char* e = std::getenv("BAYENS_MEDIUM_TUNE"); return e ? std::atoi(e) : 0; }();
  g.tune = tune;
  const size_t lds = medium_lds_bytes(g.N);
  static bool configured[64] = {};
  int dev = 0;
  BAYENS_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !configured[dev]) {
    BAYENS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(medium_fit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   128 * 1024));
    if (dev >= 0 && dev < 64) configured[dev] = true;
  }
  // every array the kernel is handed, with its size in bytes (region 0 = the tile scratch, one stride per resident workgroup)
  const int grid = medium_grid(g.B, g.N, n_cus);
  const size_t BN = (size_t)g.B * g.N;
  struct Region { const char* name; const void* base; size_t bytes; };
  const Region regions[] = {
    {"scratch", g.scratch, (size_t)grid * g.scratch_stride * sizeof(double)},
    {"X", g.X, (g.x_elems ? (size_t)g.x_elems : BN * g.R) * sizeof(double)}, {"y", g.y, BN * sizeof(double)}, {"nu", g.nu, BN * sizeof(double)},
    {"mu", g.mu, BN * sizeof(double)}, {"cov", g.cov, BN * g.N * sizeof(double)}, {"var", g.var, BN * sizeof(double)},
    {"chol", g.chol, BN * g.N * sizeof(double)}, {"logdet", g.logdet, (size_t)g.B * sizeof(double)},
    {"hyp", g.hyp, (size_t)g.B * 2 * sizeof(double)}, {"basis", g.basis, BN * 2 * sizeof(double)},
    {"info", g.info, (size_t)g.B * sizeof(int)}, {"loss", g.loss, (size_t)g.B * g.loss_stride * sizeof(double)},
    {"stamps", g.stamps, 160 * sizeof(unsigned long long)}};
  constexpr int n_regions = (int)(sizeof(regions) / sizeof(regions[0]));
  // BAYENS_ALLOC_MAP=1: the map goes to stderr before the launch, so that the address of a device fault can be attributed
  static const bool alloc_map = std::getenv("BAYENS_ALLOC_MAP") != nullptr;
  if (alloc_map) {
    std::fprintf(stderr, "[bayens medium map] B=%d N=%d R=%d T=%d grid=%d lds=%zu scratch_stride=%zu bytes\n", g.B, g.N, g.R, g.n_iters, grid,
                 lds, (size_t)g.scratch_stride * sizeof(double));
    for (const Region& r : regions)
      if (r.base) std::fprintf(stderr, "[bayens medium map]   %-8s %p .. %p (%zu bytes)\n", r.name, r.base, (const void*)((const char*)r.base + r.bytes), r.bytes);
    std::fflush(stderr);
  }
#ifdef BAYENS_MEDIUM_CHECK
  {
    static_assert(n_regions <= CHK_REGIONS, "region table");
    unsigned long long hr[2 * CHK_REGIONS] = {};
    for (int k = 0; k < n_regions; ++k) { hr[2 * k] = (unsigned long long)regions[k].base; hr[2 * k + 1] = regions[k].base ? regions[k].bytes : 0; }
    unsigned long long h[8] = {0ull, (unsigned long long)g.scratch_stride * sizeof(double), 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    BAYENS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_reg), hr, sizeof(hr)));
    BAYENS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_chk), h, sizeof(h)));
  }
#endif
  // BAYENS_MEDIUM_GUARD=<MiB> (diagnostic): the launch runs on a private tile scratch with guard bands of that size in front of
  // and behind it, filled with NaN canaries that carry their own index; after the launch every canary is compared.  A stray
  // STORE shows up as a modified canary (offset and value printed), a stray LOAD whose value is used as NaNs in the outputs.
  static const long guard_mib = [] { const char* e = std::getenv("BAYENS_MEDIUM_GUARD"); return e ? std::atol(e) : 0L; }();
  char* guard_block = nullptr;
  const size_t guard_bytes = (size_t)(guard_mib > 0 ? guard_mib : 0) << 20, body_bytes = (size_t)grid * g.scratch_stride * sizeof(double);
  std::vector<unsigned long long> canary;
  if (guard_bytes) {
    BAYENS_HIP(hipMalloc((void**)&guard_block, 2 * guard_bytes + body_bytes));
    canary.resize(guard_bytes / 8);
    for (size_t i = 0; i < canary.size(); ++i) canary[i] = 0x7ff8dead00000000ull | i;
    BAYENS_HIP(hipMemcpy(guard_block, canary.data(), guard_bytes, hipMemcpyHostToDevice));
    BAYENS_HIP(hipMemcpy(guard_block + guard_bytes + body_bytes, canary.data(), guard_bytes, hipMemcpyHostToDevice));
    BAYENS_HIP(hipMemset(guard_block + guard_bytes, 0, body_bytes));
    g.scratch = reinterpret_cast<double*>(guard_block + guard_bytes);
    std::fprintf(stderr, "[bayens medium guard] scratch %p .. %p with %ld MiB of canaries on either side\n", (void*)g.scratch,
                 (void*)(guard_block + guard_bytes + body_bytes), guard_mib);
  }
  hipLaunchKernelGGL(medium_fit_kernel, dim3(grid), dim3(NTH), lds, stream, g);
  BAYENS_HIP(hipGetLastError());
  if (guard_bytes) {
    BAYENS_HIP(hipStreamSynchronize(stream));
    std::vector<unsigned long long> back(canary.size());
    for (int side = 0; side < 2; ++side) {
      BAYENS_HIP(hipMemcpy(back.data(), guard_block + (side ? guard_bytes + body_bytes : 0), guard_bytes, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < back.size(); ++i)
        if (back[i] != canary[i]) {
          if (bad < 24) {
            double v; std::memcpy(&v, &back[i], 8);
            const long off = side ? (long)(body_bytes + 8 * i) : (long)(8 * i) - (long)guard_bytes;
            std::fprintf(stderr, "[bayens medium guard]   scratch base%+ld bytes (tile %+.3f): 0x%016llx = %.17g\n", off, off / 2048.0, back[i], v);
          }
          ++bad;
        }
      std::fprintf(stderr, "[bayens medium guard] %s guard: %zu of %zu canaries modified\n", side ? "rear" : "front", bad, back.size());
    }
    BAYENS_HIP(hipFree(guard_block));
  }
#ifdef BAYENS_MEDIUM_CHECK
  {
    unsigned long long h[8];
    BAYENS_HIP(hipStreamSynchronize(stream));
    BAYENS_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_chk), sizeof(h)));
    std::fprintf(stderr, "[bayens medium check] B=%d N=%d grid=%d: %llu global accesses outside the registered regions%s",
                 g.B, g.N, grid, h[2], h[2] ? "" : "\n");
    if (h[2]) std::fprintf(stderr, ", first at %p = scratch base%+lld bytes (workgroup %llu, medium.hip:%llu; region per workgroup %llu bytes)\n",
                           (void*)h[3], (long long)(h[3] - (unsigned long long)g.scratch), h[5], h[4], h[1]);
  }
#endif
}

}  // namespace mediumn
}  // namespace bayens
