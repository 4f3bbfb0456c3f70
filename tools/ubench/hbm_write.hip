// Micro-benchmark: HBM write bandwidth for the P-matrix store patterns (diagnosis tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// pattern 0: fully linear 16 B per lane.  pattern 1: rows of 14336 B, a wave writes 4 rows x 256 B per store (the staged layout).
// pattern 2: a wave writes 32 rows x 16 B per store (8 B per lane, lanes l and l+32 adjacent) -- the accumulator layout.
template <int PAT>
__global__ __launch_bounds__(256) void k(char* __restrict__ P, long ld, int tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row0 = ((long)blockIdx.x * 4 + wave) * 32;
  u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
  for (int t = 0; t < tiles; ++t) {
    if (PAT == 0) {
      for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(P + ((row0 * tiles + (long)t * 32) * 256) + (i * 64 + lane) * 16) = v;
    } else if (PAT == 1) {
      for (int it = 0; it < 8; ++it) *reinterpret_cast<u32x4*>(P + (row0 + it * 4 + (lane >> 4)) * ld + (long)t * 256 + (lane & 15) * 16) = v;
    } else {
      for (int mt = 0; mt < 4; ++mt) for (int q = 0; q < 4; ++q)
        *reinterpret_cast<u32x2*>(P + (row0 + (lane & 31)) * ld + (long)t * 256 + (mt * 32 + 8 * q + 4 * (lane >> 5)) * 2) = u32x2{v.x, v.y};
    }
  }
}
template <int PAT> void run(char* P) {
  const long ld = 14336; const int tiles = 56; const int nblk = 114688 / 128;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PAT>), dim3(nblk), dim3(256), 0, 0, P, ld, tiles);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<PAT>), dim3(nblk), dim3(256), 0, 0, P, ld, tiles);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("pattern %d: %.3f ms  %.2f TB/s\n", PAT, ms, 114688.0 * 14336 / (ms * 1e-3) / 1e12);
}
int main() {
  char* P; hipMalloc(&P, (size_t)114688 * 14336 + 4096);
  run<0>(P); run<1>(P); run<2>(P);
  return 0;
}
