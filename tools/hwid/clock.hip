// Shader clock under load: s_memtime (shader cycles) against s_memrealtime (100 MHz) around
//   mode 0: a register-only fp64 MFMA loop (16 independent accumulators per wave, 2 workgroups of 4 waves per CU)
//   mode 1: the same loop with a 1 KB global -> LDS DMA per wave every 16 MFMAs (the GEMM's operand traffic)
//   mode 2: the DMA alone
//   mode 7: a fifth wave per workgroup issues the workgroup's DMA (mode 3's traffic), the four MFMA waves none
//   mode 3: the MFMA loop with 2 KB of DMA per wave and 16 MFMAs - the ratio of the 128 x 128 x 16 GEMM step
// Prints the achieved TFLOP/s and the mean shader frequency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(320, 2) void k(int mode, int iters, const double* src, long span, unsigned long long* out, double* sink) {
  __shared__ __attribute__((aligned(1024))) double lds[4096];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v4d acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
  double keep[2] = {0, 0};
  const double* p = src + ((long)blockIdx.x * 4 + wave) * 128 + 2 * lane;
  if (wave == 4) {                       // (only launched in mode 7) the workgroup's loader: all its DMA, no MFMA
    if (mode != 7) return;
    for (int it = 0; it < iters; ++it) {
      const double* q = p + ((long)it * 131072) % span;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned off = (unsigned)(size_t)(lds + (u & 3) * 128) + (u >> 2) * 4096;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(q + u * 65536), "s"(off) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (mode >= 1 && mode <= 3) {
      const unsigned off = (unsigned)(size_t)(lds + wave * 128);
      const double* q = p + ((long)it * 131072) % span;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(q), "s"(off) : "memory");
      if (mode == 3) {      // the GEMM's ratio: 2 KB per wave and 16 MFMAs
        const unsigned off2 = off + 4096;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(q + 65536), "s"(off2) : "memory");
      }
    }
    if (mode == 4) {          // the same 2 KB per wave and 16 MFMAs, into registers instead of LDS
      const double* q = p + ((long)it * 131072) % span;
      v4d t0 = *reinterpret_cast<const v4d*>(q), t1 = *reinterpret_cast<const v4d*>(q + 65536);
      keep[it & 1] = t0[0] + t1[1];
    }
    if (mode == 5 && (it & 7) == 0) {      // mode 3's traffic issued as a burst of 16 loads every 8 iterations
      const double* q = p + ((long)it * 131072) % span;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const unsigned off = (unsigned)(size_t)(lds + wave * 128) + (u & 1) * 4096;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(q + u * 65536), "s"(off) : "memory");
      }
    }
    if (mode == 6 && wave == 0) {          // mode 3's traffic of the whole workgroup issued by wave 0 alone
      const double* q = p + ((long)it * 131072) % span;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned off = (unsigned)(size_t)(lds + (u & 3) * 128) + (u >> 2) * 4096;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(q + u * 65536), "s"(off) : "memory");
      }
    }
    if (mode <= 1 || mode >= 3) {   // (mode 7: waves 0-3 run the pure MFMA loop)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    if (mode >= 1 && mode <= 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if (mode == 5 && (it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (mode == 6 && wave == 0) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");   // keep eight 1 KB loads in flight per wave
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) sink[0] = s + lds[lane] + keep[0] + keep[1];
  if (lane == 0) { out[((long)blockIdx.x * 4 + wave) * 2] = t1 - t0; out[((long)blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}
int main() {
  const int nwg = 512, iters = 20000;
  const long span = 1L << 28;   // doubles: 2 GB window
  double *src, *sink; unsigned long long* out;
  hipMalloc(&src, (span + (1 << 20)) * sizeof(double)); hipMalloc(&sink, 8); hipMalloc(&out, nwg * 8 * sizeof(unsigned long long));
  hipMemset(src, 0, (span + (1 << 20)) * sizeof(double));
  std::vector<unsigned long long> h(nwg * 8);
  for (int mode = 0; mode < 8; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(mode == 7 ? 320 : 256), 0, 0, mode, 2000, src, span, out, sink);   // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(mode == 7 ? 320 : 256), 0, 0, mode, iters, src, span, out, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < nwg * 4; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    const double flop = (mode <= 1 || mode >= 3) ? (double)nwg * 4 * iters * 16 * 2048.0 : 0.0;
    const double bytes = mode >= 1 ? (double)nwg * 4 * iters * 1024.0 * (mode >= 3 ? 2 : 1) : 0.0;
    printf("mode %d: %.3f ms, %.2f TFLOP/s, %.2f TB/s DMA, shader clock %.3f GHz (memtime ticks per 100 MHz tick x 0.1)\n", mode, ms,
           flop / ms / 1e9, bytes / ms / 1e9, cyc / rt * 0.1);
  }
  return 0;
}
