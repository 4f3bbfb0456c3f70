// Micro-benchmark: HBM read bandwidth of the P.V access pattern -- a block owns 128 rows of a [rows x 7168] fp16
// matrix (row stride 14336 B) and walks along K in tiles of SEG bytes per row (diagnosis tool, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SEG, int DEPTH>   // SEG bytes per row per tile; DEPTH tiles in flight
__global__ __launch_bounds__(256) void k(const char* __restrict__ P, long ld_bytes, int ncol_bytes, unsigned* out) {
  constexpr int TPR = SEG / 16;            // threads per row
  constexpr int RPP = 256 / TPR;           // rows per pass
  constexpr int NP = 128 / RPP;            // passes (loads per thread per tile)
  const int tid = threadIdx.x, c = tid % TPR, r0 = tid / TPR;
  const char* base = P + ((long)blockIdx.x * 128 + r0) * ld_bytes + c * 16;
  u32x4 acc = {0, 0, 0, 0};
  const int nt = ncol_bytes / SEG;
  for (int t = 0; t < nt; t += DEPTH) {
    u32x4 v[DEPTH][NP];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int i = 0; i < NP; ++i) v[d][i] = *reinterpret_cast<const u32x4*>(base + (long)(i * RPP) * ld_bytes + (long)(t + d) * SEG);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int i = 0; i < NP; ++i) acc ^= v[d][i];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int SEG, int DEPTH> void run(const char* P, unsigned* out, int nblk) {
  const long ld = 14336; const int ncol = 14336;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SEG, DEPTH>), dim3(nblk), dim3(256), 0, 0, P, ld, ncol, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<SEG, DEPTH>), dim3(nblk), dim3(256), 0, 0, P, ld, ncol, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)nblk * 128 * ncol;
  printf("SEG %4d B  depth %d  blocks %d: %.3f ms  %.2f TB/s\n", SEG, DEPTH, nblk, ms, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const int nblk = 896;
  char* P; unsigned* out;
  hipMalloc(&P, (size_t)1024 * 128 * 14336); hipMalloc(&out, 4);
  hipMemset(P, 1, (size_t)1024 * 128 * 14336);
  run<128, 1>(P, out, nblk); run<128, 2>(P, out, nblk); run<128, 4>(P, out, nblk);
  run<256, 1>(P, out, nblk); run<256, 2>(P, out, nblk); run<256, 4>(P, out, nblk);
  run<512, 1>(P, out, nblk); run<512, 2>(P, out, nblk);
  run<1024, 1>(P, out, nblk); run<1024, 2>(P, out, nblk);
  run<2048, 1>(P, out, nblk);
  run<256, 2>(P, out, 512); run<256, 2>(P, out, 768); run<256, 2>(P, out, 1024);
  return 0;
}
