// Issue rate of the VALU instructions the softmax / aggregation epilogues are made of (diagnosis tool, not product): one block of
// 256 x WPS threads on one CU (WPS waves per SIMD), each wave runs N dependent-free copies of the instruction; cycles per wave64
// instruction per SIMD from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
#define REP 64
template <int OP>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  float a[8], b = out[threadIdx.x & 7] + 1.0f;
  float2v pa[4], pb = {b, b};
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = b + i;
#pragma unroll
  for (int i = 0; i < 4; ++i) pa[i] = float2v{b + i, b - i};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
        if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pa[i & 3]) : "v"(pb));
        if (OP == 4) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 5) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i & 3]) : "v"(pb));
        if (OP == 8) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a[i]));
        if (OP == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += pa[i].x + pa[i].y;
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP> void run(const char* nm, float* d, unsigned long long* c) {
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256 * wps), 0, 0, d, c, iters);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256 * wps), 0, 0, d, c, iters);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    // s_memtime / readcyclecounter counts at a fixed 100 MHz on gfx9: convert with the event time instead
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256 * wps), 0, 0, d, c, 20 * iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = 20.0 * iters * REP * wps;          // wave-instructions per SIMD
    printf("%-14s %d waves/SIMD: %.2f ns per wave64 instruction per SIMD (= %.1f cycles at 2.4 GHz)\n", nm, wps, ms * 1e6 / n, ms * 1e6 / n * 2.4);
  }
}
int main() {
  float* d; unsigned long long* c;
  hipMalloc(&d, 4096); hipMemset(d, 0, 4096); hipMalloc(&c, 64);
  run<0>("v_exp_f32", d, c); run<1>("v_rcp_f32", d, c); run<2>("v_fma_f32", d, c); run<3>("v_pk_fma_f32", d, c);
  run<4>("v_med3_f32", d, c); run<5>("v_max3_f32", d, c); run<6>("v_add_f32", d, c); run<7>("v_pk_mul_f32", d, c);
  run<8>("v_cvt_f16_f32", d, c); run<9>("v_mul_lo_u32", d, c);
  return 0;
}
