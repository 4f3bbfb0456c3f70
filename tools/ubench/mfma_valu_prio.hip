// Micro-benchmark (follow-up of mfma_valu_overlap.hip): can the VALU instructions of one wave run beside the MFMAs of ANOTHER wave of the same
// SIMD when the MFMA wave does not hold the issue port while it waits for the matrix pipe?  Variants of the "both roles" run:
//   PRIO  : s_setprio -- the VALU wave at priority 3, the MFMA wave at 0 (and the reverse)
//   NOP   : the MFMA wave executes s_nop between its MFMAs (it sleeps through the pipe's busy time instead of stalling at issue)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_prio.hip -o tools/ubench/mfma_valu_prio.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ROLES: 0 both, 1 MFMA only, 2 VALU only.  VAR: 0 plain, 1 VALU prio 3 / MFMA prio 0, 2 VALU prio 0 / MFMA prio 3, 3 s_nop 5 x3 after every MFMA,
// 4 s_nop 7 x3, 5 = 1 + 3
template <int ROLES, int VAR, int NN = 0, int NP = 0>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  const bool mf = wave < 4;
  float res = 0.f;
  if ((ROLES == 3 && mf) || ROLES == 4) {      // ONE wave (3) or two waves (4) per SIMD, each interleaving 1 MFMA : 6 VALU (same totals per iteration)
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
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
          const int e0 = (r * 4 + i) * 2;
          v[e0 & 15] = __builtin_amdgcn_exp2f(v[e0 & 15] * 0.25f - 1.f) + v[e0 & 15] * 0.5f;
          v[(e0 + 1) & 15] = __builtin_amdgcn_exp2f(v[(e0 + 1) & 15] * 0.25f - 1.f) + v[(e0 + 1) & 15] * 0.5f;
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) res += acc[i][e];
    for (int e = 0; e < 16; ++e) res += v[e];
  } else if (mf && ROLES != 2) {
    if (VAR == 1 || VAR == 5) __builtin_amdgcn_s_setprio(0);
    if (VAR == 2) __builtin_amdgcn_s_setprio(3);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
          if (VAR == 3 || VAR == 5) { asm volatile("s_nop 5"); asm volatile("s_nop 5"); asm volatile("s_nop 5"); __builtin_amdgcn_sched_barrier(0); }
          if (VAR == 6) {
#pragma unroll
            for (int q = 0; q < NN; ++q) asm volatile("s_nop %0" :: "n"(NP));
            __builtin_amdgcn_sched_barrier(0);
          }
          if (VAR == 4) { asm volatile("s_nop 7"); asm volatile("s_nop 7"); asm volatile("s_nop 7"); __builtin_amdgcn_sched_barrier(0); }
        }
    }
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) res += acc[i][e];
  } else if (ROLES == 3 || ROLES == 4) {
  } else if (!mf && ROLES != 1) {
    if (VAR == 1 || VAR == 5) __builtin_amdgcn_s_setprio(3);
    if (VAR == 2) __builtin_amdgcn_s_setprio(0);
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

template <int ROLES, int VAR, int NN = 0, int NP = 0> float run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<ROLES, VAR, NN, NP>), dim3(256), dim3(512), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<ROLES, VAR, NN, NP>), dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters;
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * sizeof(float));
  const int it = 4000;
  printf("per iteration and SIMD (16 MFMA 32x32x16 | 32 x (v_fma, v_exp, v_fma)), ns:\n");
  printf("  MFMA alone %.1f | VALU alone %.1f | MFMA alone with s_nop 5x3 %.1f | with s_nop 7x3 %.1f\n", run<1, 0>(d, it), run<2, 0>(d, it), run<1, 3>(d, it), run<1, 4>(d, it));
  printf("  both, plain                          %.1f\n", run<0, 0>(d, it));
  printf("  both, VALU wave prio 3 / MFMA prio 0 %.1f\n", run<0, 1>(d, it));
  printf("  both, VALU wave prio 0 / MFMA prio 3 %.1f\n", run<0, 2>(d, it));
  printf("  both, MFMA wave s_nop 5 x3 per MFMA  %.1f\n", run<0, 3>(d, it));
  printf("  both, MFMA wave s_nop 7 x3 per MFMA  %.1f\n", run<0, 4>(d, it));
  printf("  both, prio + s_nop 5 x3              %.1f\n", run<0, 5>(d, it));
  printf("  ONE wave per SIMD interleaving 1 MFMA : 2 x (fma, exp, fma) (16 MFMA + 96 VALU per iteration): %.1f\n", run<3, 0>(d, it));
  printf("  TWO waves per SIMD, each doing that (32 MFMA + 192 VALU per iteration and SIMD):               %.1f\n", run<4, 0>(d, it));
#define SW(NN, NP) printf("  s_nop %d x%d per MFMA: MFMA alone %.1f | both %.1f\n", NP, NN, run<1, 6, NN, NP>(d, it), run<0, 6, NN, NP>(d, it));
  SW(1, 0) SW(1, 1) SW(1, 3) SW(1, 5) SW(1, 7) SW(2, 3) SW(2, 5) SW(2, 7) SW(1, 15) SW(2, 15)
  return 0;
}
