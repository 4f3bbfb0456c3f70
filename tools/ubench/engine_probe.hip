// Micro-benchmark: where do the cycles of the 128 x 128 x 32 f16x3 K-tile of the GEMM engine go?  The engine's K loop
// (craft_amd/csrc/gemm_engine.hpp: global -> registers -> convert / split -> LDS -> fragments -> MFMA, one barrier per tile) is
// rebuilt here with each stage switchable, on operands that stay L2-resident (every block reads the same 128 x 4096 A and B), and
// timed per K-tile with 1 and 2 blocks per CU.  Stages: L = global loads, S = conversion + LDS stores, R = LDS fragment reads,
// M = MFMAs.  What the full loop costs beyond max(stage) is what in-order issue + barriers lose; which stage, removed, collapses the
// time is the bottleneck.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I include tools/ubench/engine_probe.hip -o tools/ubench/engine_probe.bin
#include "../../craft_amd/csrc/gemm_engine.hpp"
#include <cstdio>
#include <vector>

using namespace craft;

constexpr int PREC = CRAFT_PREC_F16X3, BM = 128, BN = 128, WM = 2, WN = 2, MT = 2, NT = 2;

template <bool L, bool S, bool R, bool M, int ORD = 0>
__global__ __launch_bounds__(NTHREADS) void k_probe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, int K) {
  typedef PrecT<PREC>::lds_t lds_t;
  typedef TileLds<PREC, BM, BN> TL;
  __shared__ __attribute__((aligned(16))) lds_t Sm[2 * (TL::A_ELEMS + TL::B_ELEMS)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  LoaderRowsF32<BM> la;
  LoaderRowsF32<BN> lb;
  la.init(A, K, 0, BM, K, tid);
  lb.init(B, K, 0, BN, K, tid);
  f32x16 acc[MT][NT];
  acc_zero(acc);
  RegsF32<BM> ra;
  RegsF32<BN> rb;
  la.fetch(0, ra);
  lb.fetch(0, rb);
  stage_store<PREC>(&Sm[0], ra, tid);
  stage_store<PREC>(&Sm[2 * TL::A_ELEMS], rb, tid);
  __syncthreads();
  const int nk = K / BK;
  constexpr int LD = PrecT<PREC>::LD;
  const int r = lane & 31, g = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int ao = (kt & 1) * TL::A_ELEMS, bo = 2 * TL::A_ELEMS + (kt & 1) * TL::B_ELEMS;
    const int an = TL::A_ELEMS - ao, bn = 2 * TL::A_ELEMS + TL::B_ELEMS - (kt & 1) * TL::B_ELEMS;
    if (L) {
      const int ktn = min(kt + 1, nk - 1);
      la.fetch(ktn, ra);
      lb.fetch(ktn, rb);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (R && M && ORD == 0) {
      constexpr int NSLOT = 2 * MT * NT, S0 = NSLOT / 3;
      auto piece = [&](int j) __attribute__((always_inline)) {
        if (!S) return;
        if (j < 4) stage_store_piece<PREC>(&Sm[an], ra, tid, j);
        else if (j < 8) stage_store_piece<PREC>(&Sm[bn], rb, tid, j - 4);
      };
      mma_tile<PREC, MT, NT, BM, BN>(&Sm[ao], &Sm[bo], wm0, wn0, lane, acc, [&](int i) __attribute__((always_inline)) {
        if (i >= S0) piece(i - S0);
        __builtin_amdgcn_sched_barrier(0);
      });
#pragma unroll
      for (int j = NSLOT - S0; j < 8; ++j) piece(j);
    } else {
      // stages in isolation
      f16x8 fa[2][2][MT], fb[2][2][NT];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if (R) fa[kk][pl][mt] = *reinterpret_cast<const f16x8*>(&Sm[ao + (pl * BM + wm0 + mt * 32 + r) * LD + kk * 16 + g * 8]);
            else for (int j = 0; j < 8; ++j) fa[kk][pl][mt][j] = (_Float16)(lane * 0.01f + j + kt);
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (R) fb[kk][pl][nt] = *reinterpret_cast<const f16x8*>(&Sm[bo + (pl * BN + wn0 + nt * 32 + r) * LD + kk * 16 + g * 8]);
            else for (int j = 0; j < 8; ++j) fb[kk][pl][nt][j] = (_Float16)(lane * 0.02f + j);
          }
        }
      if (M && ORD == 0) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk][1][mt], fb[kk][0][nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk][0][mt], fb[kk][1][nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk][0][mt], fb[kk][0][nt], acc[mt][nt], 0, 0, 0);
            }
      } else if (M) {      // same products, the three terms of one accumulator no longer back to back (per-accumulator order unchanged)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk][term == 0 ? 1 : 0][mt], fb[kk][term == 1 ? 1 : 0][nt], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        // keep the fragment reads alive
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][0][kk * 2 + pl] += (float)fa[kk][pl][mt][0];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[0][nt][4 + kk * 2 + pl] += (float)fb[kk][pl][nt][0];
          }
      }
      if (S) {
        stage_store<PREC>(&Sm[an], ra, tid);
        stage_store<PREC>(&Sm[bn], rb, tid);
      }
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt) for (int e = 0; e < 16; ++e) s += acc[mt][nt][e];
  s += ra.v[0].x + rb.v[0].x;
  if (s == 123.456f) out[blockIdx.x * NTHREADS + tid] = s;
}

template <bool L, bool S, bool R, bool M, int ORD = 0> void run(const char* name, const float* A, const float* B, float* out, int K) {
  for (int blocks : {256, 512, 768}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_probe<L, S, R, M, ORD>), dim3(blocks), dim3(NTHREADS), 0, 0, A, B, out, K);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_probe<L, S, R, M, ORD>), dim3(blocks), dim3(NTHREADS), 0, 0, A, B, out, K);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 5, tiles = (double)K / BK, rounds = blocks / 256.0;
    // cycles (at 2.4 GHz) per K-tile and CU = time / (tiles x blocks per CU)
    printf("%-28s blocks/CU %.0f: %8.1f us  -> %6.0f cycles per K-tile and block, %6.0f per K-tile and CU-slot\n", name, rounds, us,
           us * 2400.0 / tiles, us * 2400.0 / tiles / rounds);
  }
}

int main() {
  const int K = 4096;
  float *A, *B, *out;
  hipMalloc(&A, (size_t)BM * K * 4); hipMalloc(&B, (size_t)BN * K * 4); hipMalloc(&out, 768 * NTHREADS * 4);
  std::vector<float> h((size_t)BM * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<true, true, true, true>("full (L S R M)", A, B, out, K);
  run<false, true, true, true>("no global loads (S R M)", A, B, out, K);
  run<true, false, true, true>("no convert/store (L R M)", A, B, out, K);
  run<false, false, true, true>("frag reads + MFMA (R M)", A, B, out, K);
  run<false, false, false, true>("MFMA only (M)", A, B, out, K);
  run<false, false, false, true, 1>("MFMA only, interleaved accs", A, B, out, K);
  run<false, false, true, true, 1>("frag reads + MFMA interleaved", A, B, out, K);
  run<true, true, true, true, 1>("L S + (R M interleaved), S after", A, B, out, K);
  run<false, false, true, false>("frag reads only (R)", A, B, out, K);
  run<false, true, false, false>("convert/store only (S)", A, B, out, K);
  run<true, false, false, false>("global loads only (L)", A, B, out, K);
  run<true, true, false, false>("loads + convert/store (L S)", A, B, out, K);
  return 0;
}
