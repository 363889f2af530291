// dba.hip - the DTW-barycentre front-end of GPDTW1D, semantics of the reference's own DBA:
// ensembles/dtwa.py (performDBA :6-20, approximate_medoid_index :23-37, squared_DTW :48-78, DBA_update :87-143).
//
// Layout: series [B][R][N] (members x realisations x time).  One workgroup owns one (member, series pair): the
// N x N dynamic programme is swept by anti-diagonals, the three live diagonals sit in LDS, the 1-byte move codes of
// the update pass stream to HBM (B*R*N*N bytes).  Integer/compare work plus one sub, one mul, one add per cell in
// the reference's order and WITHOUT fma contraction, so costs, paths and centres are bit-identical to the
// reference, ties included.  Back-tracking is one thread per (member, series); the centre update then sums, per
// centre index, the aligned samples in the reference's accumulation order (series ascending, j descending).
#include "hip_prims.h"

namespace bayens {
namespace {

// cost(i,j) = min3(diag, left, top) + (x_i - y_j)^2 with the reference's tie order; code 0 diag, 1 left, 2 top
__device__ __forceinline__ double dtw_cell(double diag, double left, double top, double xi, double yj, int& code) {
  double res;
  if (diag <= left) {
    if (diag <= top) { res = diag; code = 0; } else { res = top; code = 2; }
  } else {
    if (left <= top) { res = left; code = 1; } else { res = top; code = 2; }
  }
  const double df = __dsub_rn(xi, yj);
  return __dadd_rn(res, __dmul_rn(df, df));
}

// mode 0 (medoid): pair p = (a, t) of member b: x = series[b][a], y = series[b][t] -> sq[b][a][t]
// mode 1 (update): pair p = r: x = center[b], y = series[b][r] -> path[b][r][N][N]
__global__ __launch_bounds__(1024) void dtw_kernel(int mode, int R, int N, const double* series, const double* center,
                                                   double* sq, signed char* path) {
  extern __shared__ double lds[];
  double* d0 = lds;            // diagonal d-2, indexed by i
  double* d1 = lds + N;        // diagonal d-1
  double* d2 = lds + 2 * N;    // diagonal d
  const int b = blockIdx.y, p = blockIdx.x;
  const double* x;
  const double* y;
  signed char* pm = nullptr;
  if (mode == 0) {
    x = series + ((long)b * R + p / R) * N;
    y = series + ((long)b * R + p % R) * N;
  } else {
    x = center + (long)b * N;
    y = series + ((long)b * R + p) * N;
    pm = path + ((long)b * R + p) * N * N;
  }
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  for (int d = 0; d <= 2 * N - 2; ++d) {
    const int lo = d - N + 1 > 0 ? d - N + 1 : 0, hi = d < N - 1 ? d : N - 1;
    for (int i = lo + threadIdx.x; i <= hi; i += blockDim.x) {
      const int j = d - i;
      double c;
      int code = -1;
      if (d == 0) {
        const double df = __dsub_rn(x[0], y[0]);
        c = __dmul_rn(df, df);
      } else {
        const double diag = (i > 0 && j > 0) ? d0[i - 1] : INF;
        const double left = j > 0 ? d1[i] : INF;
        const double top = i > 0 ? d1[i - 1] : INF;
        c = dtw_cell(diag, left, top, x[i], y[j], code);
      }
      d2[i] = c;
      if (pm) pm[(long)i * N + j] = (signed char)code;
    }
    __syncthreads();
    double* t = d0; d0 = d1; d1 = d2; d2 = t;     // rotate (uniform across the workgroup)
  }
  if (mode == 0 && threadIdx.x == 0) sq[((long)b * R + p / R) * R + p % R] = d1[N - 1];
}

// dtwa.py:23-37: first candidate with the smallest sum of squared DTW distances (sequential sums, strict <)
__global__ void medoid_kernel(int B, int R, int N, const double* sq, const double* series, int* medoid, double* center) {
  const int b = blockIdx.x;
  __shared__ int best_idx;
  if (threadIdx.x == 0) {
    int idx = -1;
    double best = 0.0;
    for (int a = 0; a < R; ++a) {
      double ss = 0.0;
      for (int t = 0; t < R; ++t) ss = __dadd_rn(ss, sq[((long)b * R + a) * R + t]);
      if (idx == -1 || ss < best) { best = ss; idx = a; }
    }
    best_idx = idx;
    medoid[b] = idx;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += blockDim.x) center[(long)b * N + i] = series[((long)b * R + best_idx) * N + i];
}

// walk the stored moves from (N-1, N-1) to (0, 0); per centre index record the aligned j range
__global__ void backtrack_kernel(int B, int R, int N, const signed char* path, int* jlo, int* jhi) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B * R) return;
  const signed char* pm = path + (long)p * N * N;
  int* lo = jlo + (long)p * N;
  int* hi = jhi + (long)p * N;
  int i = N - 1, j = N - 1;
  hi[i] = j;
  while (true) {
    const int code = pm[(long)i * N + j];
    if (code < 0) break;
    if (code == 0) { lo[i] = j; --i; --j; hi[i] = j; }
    else if (code == 1) { --j; }
    else { lo[i] = j; --i; hi[i] = j; }
  }
  lo[0] = 0;
}

// centre[i] = (sum over series, j from jhi down to jlo, of y[j]) / count      (dtwa.py:131-143)
__global__ void dba_accumulate_kernel(int R, int N, const double* series, const int* jlo, const int* jhi, double* center) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double acc = 0.0;
  long cnt = 0;
  for (int r = 0; r < R; ++r) {
    const double* y = series + ((long)b * R + r) * N;
    const long q = ((long)b * R + r) * N + i;
    for (int j = jhi[q]; j >= jlo[q]; --j) { acc = __dadd_rn(acc, y[j]); ++cnt; }
  }
  center[(long)b * N + i] = acc / (double)cnt;
}

}  // namespace

void HipPrims::dba_batch(int B, int R, int N, const double* series, int n_iterations, double* center, int* medoid,
                         double* sq, signed char* path, int* jlo, int* jhi) {
  if (R > 50) throw HipError{-1, "dba: more than 50 series per member (the reference samples a random subset there)"};
  const size_t lds = (size_t)3 * N * sizeof(double);
  if (lds > 150 * 1024) throw HipError{-1, "dba: N too large for the LDS-resident diagonals"};
  const int threads = N >= 1024 ? 1024 : (N >= 256 ? 256 : 64);
  if (lds > 48 * 1024)
    BAYENS_HIP(hipFuncSetAttribute((const void*)dtw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(dtw_kernel, dim3(R * R, B), dim3(threads), lds, stream, 0, R, N, series, (const double*)nullptr, sq,
                     (signed char*)nullptr);
  hipLaunchKernelGGL(medoid_kernel, dim3(B), dim3(256), 0, stream, B, R, N, sq, series, medoid, center);
  for (int it = 0; it < n_iterations; ++it) {
    hipLaunchKernelGGL(dtw_kernel, dim3(R, B), dim3(threads), lds, stream, 1, R, N, series, center, (double*)nullptr, path);
    hipLaunchKernelGGL(backtrack_kernel, dim3((B * R + 63) / 64), dim3(64), 0, stream, B, R, N, path, jlo, jhi);
    hipLaunchKernelGGL(dba_accumulate_kernel, dim3((N + 255) / 256, B), dim3(256), 0, stream, R, N, series, jlo, jhi, center);
  }
  BAYENS_HIP(hipGetLastError());
}

}  // namespace bayens
