// Micro-benchmark: sustained issue rate of v_mfma_f32_32x32x16_f16 per SIMD (diagnosis tool, not part of the product).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        else {
          typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a), *reinterpret_cast<bf16x8*>(&b), acc[i], 0, 0, 0);
        }
      }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int KIND> void run(const char* name, int blocks_per_cu, float* d) {
  const int iters = 2000, nblk = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, KIND>), dim3(nblk), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, KIND>), dim3(nblk), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 4 * NACC * blocks_per_cu;     // one wave of each block per SIMD
  const double flops = (double)nblk * 4 * iters * 4 * NACC * 32768.0;
  printf("%-10s nacc %d waves/SIMD %d: %.3f ms  -> %.1f ns/MFMA/SIMD  %.0f TFLOP/s\n", name, NACC, blocks_per_cu, ms,
         ms * 1e6 / mfma_per_simd, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<4, 0>("f16", 1, d); run<4, 0>("f16", 2, d); run<4, 0>("f16", 4, d);
  run<1, 0>("f16", 1, d); run<2, 0>("f16", 2, d); run<8, 0>("f16", 1, d);
  run<4, 1>("bf16", 1, d); run<4, 1>("bf16", 2, d);
  return 0;
}
