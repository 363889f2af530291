// Where does global_load_lds_dwordx4 put a lane's 16 bytes?  (expected: M0 base + 16 * lane)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double* src, double* dst) {
  __shared__ __attribute__((aligned(16))) double tile[256];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 256; i += blockDim.x) tile[i] = -1.0;
  __syncthreads();
  // wave w, lane l: 16 bytes from src[2 * (64 w + (l ^ 5))] into LDS block w
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + 2 * (64 * wave + (lane ^ 5))),
                                   (void __attribute__((address_space(3)))*)&tile[128 * wave], 16, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x) dst[i] = tile[i];
}
int main() {
  double h[256], o[256], *d, *e;
  for (int i = 0; i < 256; ++i) h[i] = i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, e);
  hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 2; ++w)
    for (int l = 0; l < 64; ++l)
      for (int q = 0; q < 2; ++q) {
        const double want = 2 * (64 * w + (l ^ 5)) + q, got = o[128 * w + 2 * l + q];
        if (want != got) { if (bad < 8) printf("wave %d lane %d q %d: got %g want %g\n", w, l, q, got, want); ++bad; }
      }
  printf("mismatches %d\n", bad);
  return bad != 0;
}
