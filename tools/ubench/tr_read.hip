// Probe (not part of the product): semantics of gfx950's ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_v4f16) and of the
// 16-byte LDS-DMA load (__builtin_amdgcn_global_load_lds, size 16), as the packed-operand GEMM of kernels_gemm_pk.hip relies on them.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_read.hip -o tools/ubench/tr_read.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
#define LDS_ADDR __attribute__((address_space(3)))

// 1. every lane reads the 8 bytes at element offset 4*lane of an LDS array whose 16-bit element i holds the value i.
//    out[lane][e] = index of the element that arrived -> (source lane, source element).
__global__ void k_tr_linear(int* out) {
  __shared__ __attribute__((aligned(16))) unsigned short S[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) S[i] = (unsigned short)i;
  __syncthreads();
  auto p = reinterpret_cast<LDS_ADDR s16x4*>((LDS_ADDR unsigned short*)(S) + 4 * threadIdx.x);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (unsigned short)v[e];
}

// 2. the layout the GEMM wants: an LDS tile T[k][m] (m contiguous, row stride LDM elements); lane l of a 16-lane group g reads at
//    &T[k0 + (l & 3)... candidates below; print which (k, m) each lane receives so the right address formula can be read off.
__global__ void k_tr_tile(int* out, int variant) {
  constexpr int LDM = 64;
  __shared__ __attribute__((aligned(16))) unsigned short S[16 * LDM];
  for (int i = threadIdx.x; i < 16 * LDM; i += 64) S[i] = (unsigned short)i;      // value = k * 64 + m
  __syncthreads();
  const int l = threadIdx.x & 15, g = threadIdx.x >> 4;
  int k, m;
  if (variant == 0) { k = l >> 2; m = 16 * g + 4 * (l & 3); }          // lane -> row l/4, 4 contiguous m at 4*(l%4)
  else { k = l & 3; m = 16 * g + 4 * (l >> 2); }                        // lane -> row l%4, 4 contiguous m at 4*(l/4)
  auto p = reinterpret_cast<LDS_ADDR s16x4*>((LDS_ADDR unsigned short*)(S) + k * LDM + m);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (unsigned short)v[e];
}

// 3. LDS-DMA: each lane supplies a global address; the wave's 64 x 16 bytes land at LDS base + 16 * lane.
__global__ void k_dma(const uint32_t* src, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t S[512];
  for (int i = threadIdx.x; i < 512; i += 64) S[i] = 0xdeadbeefu;
  __syncthreads();
  // lane l fetches the 16 bytes at src + 4 * (63 - l) dwords (reversed) -> expect S[4 l + j] = src[4 (63 - l) + j]
  const uint32_t* g = src + 4 * (63 - threadIdx.x);
  __builtin_amdgcn_global_load_lds(g, (LDS_ADDR uint32_t*)(S), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = S[i];
}

int main() {
  int* d; hipMalloc(&d, 64 * 4 * sizeof(int));
  std::vector<int> h(256);
  hipLaunchKernelGGL(k_tr_linear, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h.data(), d, 256 * sizeof(int), hipMemcpyDeviceToHost);
  printf("== tr16_b64, lane reads elements 4*lane..4*lane+3: out[lane] = (source lane, source element) x 4\n");
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf(" (%2d,%d)", h[l * 4 + e] / 4, h[l * 4 + e] % 4);
    printf("\n");
  }
  for (int variant = 0; variant < 2; ++variant) {
    hipLaunchKernelGGL(k_tr_tile, dim3(1), dim3(64), 0, 0, d, variant);
    hipMemcpy(h.data(), d, 256 * sizeof(int), hipMemcpyDeviceToHost);
    printf("== tile T[k][m] (LDM 64), address variant %d: out[lane] = (k, m) x 4\n", variant);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int e = 0; e < 4; ++e) printf(" (%d,%2d)", h[l * 4 + e] / 64, h[l * 4 + e] % 64);
      printf("\n");
    }
  }
  uint32_t *src, *dst; hipMalloc(&src, 1024); hipMalloc(&dst, 1024);
  std::vector<uint32_t> hs(256), hd(256);
  for (int i = 0; i < 256; ++i) hs[i] = i;
  hipMemcpy(src, hs.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_dma, dim3(1), dim3(64), 0, 0, src, dst);
  hipMemcpy(hd.data(), dst, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) bad += hd[4 * l + j] != (uint32_t)(4 * (63 - l) + j);
  printf("== global_load_lds 16 B: LDS[16*lane] <- lane's own global address: %s (first dwords %u %u %u %u | %u)\n", bad ? "MISMATCH" : "ok",
         hd[0], hd[1], hd[2], hd[3], hd[4]);
  return 0;
}
