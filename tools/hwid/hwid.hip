// Diagnostic: where do workgroups land?  Reads HW_REG_HW_ID (and XCC_ID) per workgroup and prints the histogram of
// (xcc, se, sh, cu) for a grid that fills the chip.  Build: hipcc --offload-arch=gfx950 -O2 hwid.hip -o hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void k(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID, all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID (gfx940+)
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // keep the workgroup resident for a while so that the grid spreads over all CUs
  long t0 = clock64();
  while (clock64() - t0 < 2000000) {}
}
int main() {
  const int n = 1024;
  unsigned* d;
  hipMalloc(&d, n * 2 * sizeof(unsigned));
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
  std::vector<unsigned> h(2 * n);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::map<unsigned, int> cu, se, sh, xc, key;
  for (int i = 0; i < n; ++i) {
    unsigned hw = h[2 * i], x = h[2 * i + 1];
    cu[(hw >> 8) & 0xF]++; sh[(hw >> 12) & 1]++; se[(hw >> 13) & 0x7]++; xc[x & 0xF]++;
    key[((x & 0xF) << 16) | (hw & 0xFF00)]++;
  }
  printf("cu_id:"); for (auto& p : cu) printf(" %u:%d", p.first, p.second); printf("\n");
  printf("sh_id:"); for (auto& p : sh) printf(" %u:%d", p.first, p.second); printf("\n");
  printf("se_id:"); for (auto& p : se) printf(" %u:%d", p.first, p.second); printf("\n");
  printf("xcc_id:"); for (auto& p : xc) printf(" %u:%d", p.first, p.second); printf("\n");
  printf("distinct (xcc, se, sh, cu) = %zu\n", key.size());
  printf("first ids: "); for (int i = 0; i < 16; ++i) printf("%08x/%x ", h[2 * i], h[2 * i + 1]); printf("\n");
  return 0;
}
