// Micro-benchmark for VERDICT r4 #6 (one bounded k_pv16 experiment): the P stream of k_pv16 -- 16 (b, m) slices of 7168 x 7168 fp16 in
// 32-row x 64-key tiles (CRAFT_P_TILED), a block owning BANDS consecutive 32-row bands and walking the 112 K-tiles -- read
//   (a) global -> VGPR with 16-byte loads two tiles ahead (what k_pv16 does: kernels_gemm.hip), one barrier per tile;
//   (b) global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), S stages, optional nt, one vmcnt wait + barrier per tile.
// Nothing is computed: this is the ceiling each staging form offers the kernel.  hipcc --offload-arch=gfx950 -O3 -o x hbm_rows_dma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDSA __attribute__((address_space(3)))

constexpr int N = 7168, NKT = N / 64, Z = 16;

template <int BANDS, bool NT>
__global__ __launch_bounds__(256) void k_regs(const char* __restrict__ P, unsigned* out) {
  __shared__ u32x4 S[2][BANDS * 256];
  const int tid = threadIdx.x;
  const int nblk = (N / 32) / BANDS;
  const int z = blockIdx.x / nblk, bx = blockIdx.x % nblk;
  const char* base = P + ((long)z * (N / 32) + (long)bx * BANDS) * (32L * N * 2) + tid * 16;
  u32x4 va[BANDS], vb[BANDS], acc = {0, 0, 0, 0};
  auto fetch = [&](int kt, u32x4 (&v)[BANDS]) {
    const int k = kt < NKT ? kt : NKT - 1;
#pragma unroll
    for (int i = 0; i < BANDS; ++i) {
      const u32x4* q = reinterpret_cast<const u32x4*>(base + (long)i * (32L * N * 2) + (long)k * 4096);
      v[i] = NT ? __builtin_nontemporal_load(q) : *q;
    }
  };
  auto step = [&](int kt, u32x4 (&vnear)[BANDS], u32x4 (&vfar)[BANDS]) {
    fetch(kt + 2, vfar);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < BANDS; ++i) S[(kt + 1) & 1][i * 256 + tid] = vnear[i];
    acc ^= S[kt & 1][tid];
    __syncthreads();
  };
  fetch(0, va);
#pragma unroll
  for (int i = 0; i < BANDS; ++i) S[0][i * 256 + tid] = va[i];
  fetch(1, vb);
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < NKT; kt += 2) { step(kt, vb, va); step(kt + 1, va, vb); }
  if ((acc.x ^ acc.y) == 0x12345678u) out[0] = 1;
}

template <int BANDS, int STAGES, bool NT>
__global__ __launch_bounds__(256) void k_dma(const char* __restrict__ P, unsigned* out) {
  __shared__ __attribute__((aligned(1024))) unsigned char S[STAGES * BANDS * 4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = (N / 32) / BANDS;
  const int z = blockIdx.x / nblk, bx = blockIdx.x % nblk;
  const unsigned long long a = reinterpret_cast<unsigned long long>(P + ((long)z * (N / 32) + (long)bx * BANDS) * (32L * N * 2));
  u32x4 d;
  d[0] = __builtin_amdgcn_readfirstlane((unsigned)a); d[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  d[2] = 0xffffffffu; d[3] = 0x00020000u;
  const unsigned lds0 = (unsigned)(size_t)(LDSA unsigned char*)(S);
  const unsigned voff = lane * 16 + wave * 1024;
  auto dma = [&](int kt, int st) {
    const int k = kt < NKT ? kt : NKT - 1;
#pragma unroll
    for (int i = 0; i < BANDS; ++i) {
      const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(i * (32 * N * 2) + k * 4096));
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((st * BANDS + i) * 4096 + wave * 1024));
      unsigned keep;
      if (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, %4 offen nt lds\n\ts_mov_b32 m0, %0"
                           : "=&s"(keep) : "v"(voff), "s"(dst), "s"(d), "s"(soff) : "memory");
      else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                        : "=&s"(keep) : "v"(voff), "s"(dst), "s"(d), "s"(soff) : "memory");
    }
  };
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) dma(s, s);
  int st = 0;
  for (int kt = 0; kt < NKT; ++kt) {
    int sn = st + STAGES - 1; if (sn >= STAGES) sn -= STAGES;
    dma(kt + STAGES - 1, sn);                                   // (the stage freed by the barrier of the previous iteration)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BANDS * (STAGES - 1)) : "memory");   // this wave's pieces of tile kt have landed
    __syncthreads();
    acc ^= *reinterpret_cast<const u32x4*>(&S[st * BANDS * 4096 + tid * 16]);
    __syncthreads();
    if (++st == STAGES) st = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((acc.x ^ acc.y) == 0x12345678u) out[0] = 1;
}

template <typename F> void timeit(const char* name, int nblk, F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  const double bytes = (double)Z * N * N * 2;
  printf("%-64s blocks %4d: %.3f ms  %.2f TB/s\n", name, nblk, ms, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  char* P; unsigned* out;
  const size_t bytes = (size_t)Z * N * N * 2;
  hipMalloc(&P, bytes); hipMalloc(&out, 4);
  hipMemset(P, 1, bytes);
  for (int rep = 0; rep < 2; ++rep) {
    timeit("regs: 16-B loads two tiles ahead, 7 bands, 2 blocks/CU", 512, [&] { hipLaunchKernelGGL((k_regs<7, false>), dim3(512), dim3(256), 0, 0, P, out); });
    timeit("regs: the same with __builtin_nontemporal_load (nt)", 512, [&] { hipLaunchKernelGGL((k_regs<7, true>), dim3(512), dim3(256), 0, 0, P, out); });
    timeit("dma : 7 bands, 2 stages (56 KB: 2 blocks/CU)", 512, [&] { hipLaunchKernelGGL((k_dma<7, 2, false>), dim3(512), dim3(256), 0, 0, P, out); });
    timeit("dma : 7 bands, 2 stages, nt", 512, [&] { hipLaunchKernelGGL((k_dma<7, 2, true>), dim3(512), dim3(256), 0, 0, P, out); });
    timeit("dma : 7 bands, 3 stages (84 KB: 1 block/CU, 2 rounds)", 512, [&] { hipLaunchKernelGGL((k_dma<7, 3, false>), dim3(512), dim3(256), 0, 0, P, out); });
    timeit("dma : 7 bands, 3 stages, nt", 512, [&] { hipLaunchKernelGGL((k_dma<7, 3, true>), dim3(512), dim3(256), 0, 0, P, out); });
    timeit("dma : 4 bands, 4 stages (64 KB: 2 blocks/CU, 896 blocks)", 896, [&] { hipLaunchKernelGGL((k_dma<4, 4, false>), dim3(896), dim3(256), 0, 0, P, out); });
    timeit("dma : 4 bands, 4 stages, nt", 896, [&] { hipLaunchKernelGGL((k_dma<4, 4, true>), dim3(896), dim3(256), 0, 0, P, out); });
    timeit("dma : 14 bands, 2 stages (112 KB: 1 block/CU, 256 blocks)", 256, [&] { hipLaunchKernelGGL((k_dma<14, 2, false>), dim3(256), dim3(256), 0, 0, P, out); });
    timeit("dma : 14 bands, 2 stages, nt", 256, [&] { hipLaunchKernelGGL((k_dma<14, 2, true>), dim3(256), dim3(256), 0, 0, P, out); });
  }
  return 0;
}
