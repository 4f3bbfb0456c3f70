// Micro-benchmark: do VALU / transcendental instructions of ONE wave issue in the shadow of ANOTHER wave's MFMAs on the
// same SIMD?  512-thread blocks, one per CU: waves 0-3 (one per SIMD) run an MFMA loop, waves 4-7 a VALU loop (fma + exp2
// mix like a softmax), and each role is also timed alone (the other role exits at once).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/mfma_valu_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: both roles, 1: MFMA waves only, 2: VALU waves only, 3: every wave alternates 16 MFMA / 64 VALU
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  const bool mf = wave < 4;
  float res = 0.f;
  if (MODE == 3 || (mf && MODE != 2)) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
    float v[16];
    for (int e = 0; e < 16; ++e) v[e] = threadIdx.x * 1e-3f + e;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
      if (MODE == 3) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = __builtin_amdgcn_exp2f(v[e] * 0.25f - 1.f) + v[e] * 0.5f;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) res += acc[i][e];
    for (int e = 0; e < 16; ++e) res += v[e];
  } else if (!mf && MODE != 1) {
    float v[16];
    for (int e = 0; e < 16; ++e) v[e] = threadIdx.x * 1e-3f + e;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = __builtin_amdgcn_exp2f(v[e] * 0.25f - 1.f) + v[e] * 0.5f;   // fma, exp, fma
    }
    for (int e = 0; e < 16; ++e) res += v[e];
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int MODE> float run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * sizeof(float));
  const int iters = 4000;
  const float both = run<0>(d, iters), m = run<1>(d, iters), v = run<2>(d, iters), alt = run<3>(d, iters);
  printf("per iteration and SIMD (16 MFMA 32x32x16 | 32 x (v_fma, v_exp, v_fma)):\n");
  printf("  MFMA wave alone      %.1f ns\n  VALU wave alone      %.1f ns\n  both (one wave each) %.1f ns   (sum %.1f, max %.1f)\n",
         m * 1e6 / iters, v * 1e6 / iters, both * 1e6 / iters, (m + v) * 1e6 / iters, (m > v ? m : v) * 1e6 / iters);
  printf("  8 waves each alternating 16 MFMA / 32 x (fma, exp, fma) in phase-free loops (2 per SIMD): %.1f ns per wave-iteration pair\n",
         alt * 1e6 / iters);
  return 0;
}
