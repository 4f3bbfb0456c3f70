// Micro-benchmark: how many bytes in flight per CU does the P.V row-streaming pattern need?  512 blocks x 224 rows, occupancy
// limited to 2 blocks per CU by a dummy LDS allocation, DEPTH tiles of 128 B per row in flight per thread (diagnosis tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(256) void k(const char* __restrict__ P, long ld_bytes, int ncol_bytes, unsigned* out) {
  __shared__ char pad[70 * 1024];
  const int tid = threadIdx.x, c = tid & 7, r0 = tid >> 3;
  const char* base = P + ((long)blockIdx.x * 224 + r0) * ld_bytes + c * 16;
  u32x4 acc = {0, 0, 0, 0};
  const int nt = ncol_bytes / 128;
  u32x4 v[DEPTH][7];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int i = 0; i < 7; ++i) v[d][i] = *reinterpret_cast<const u32x4*>(base + (long)(i * 32) * ld_bytes + (long)d * 128);
  for (int t = 0; t < nt; t += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int i = 0; i < 7; ++i) acc ^= v[d][i];
      const int tn = min(t + DEPTH + d, nt - 1);
#pragma unroll
      for (int i = 0; i < 7; ++i) v[d][i] = *reinterpret_cast<const u32x4*>(base + (long)(i * 32) * ld_bytes + (long)tn * 128);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) { out[0] = 1; pad[tid] = 1; }
}

template <int DEPTH> void run(const char* P, unsigned* out) {
  const long ld = 14336; const int ncol = 14336; const int nblk = 512;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<DEPTH>), dim3(nblk), dim3(256), 0, 0, P, ld, ncol, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<DEPTH>), dim3(nblk), dim3(256), 0, 0, P, ld, ncol, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)nblk * 224 * ncol;
  printf("depth %d (%.0f KB in flight per CU): %.3f ms  %.2f TB/s\n", DEPTH, DEPTH * 28.0 * 2, ms, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  char* P; unsigned* out;
  hipMalloc(&P, (size_t)512 * 224 * 14336); hipMalloc(&out, 4);
  hipMemset(P, 1, (size_t)512 * 224 * 14336);
  run<1>(P, out); run<2>(P, out); run<3>(P, out); run<4>(P, out); run<6>(P, out); run<8>(P, out);
  return 0;
}
