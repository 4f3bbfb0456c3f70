// Probe: which physical CU does workgroup i of a 1-D launch land on?  512 co-resident blocks (256 threads, 60 KB LDS each: two per
// CU, like k_corr_build4t) record HW_ID / XCC_ID; the host prints, per CU, the block ids it received (diagnosis tool, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, int spin) {
  __shared__ float pad[15000];
  pad[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float a = pad[(threadIdx.x * 7) & 255];
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;          // stay resident so that all 512 blocks coexist
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  if (a == 12345.f) out[0] = 0;
}
int main() {
  const int nb = 1024;
  unsigned* d; hipMalloc(&d, nb * 8);
  hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * nb); hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  for (int i = 0; i < nb; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cu[(xcc << 12) | (se << 8) | (sh << 4) | cu_id].push_back(i);
  }
  printf("%zu distinct CUs\n", cu.size());
  int n = 0, pair_ok = 0;
  for (auto& kv : cu) {
    if (n++ < 24) { printf("xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 0xf, (kv.first >> 4) & 0xf, kv.first & 0xf); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    if (kv.second.size() >= 2 && kv.second[0] < 256 && kv.second[1] >= 256 && kv.second[1] < 512) ++pair_ok;
  }
  printf("CUs whose first two blocks are (one of 0..255, one of 256..511): %d\n", pair_ok);
  return 0;
}
