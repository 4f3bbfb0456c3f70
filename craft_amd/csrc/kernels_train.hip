// Training-side kernels of the hot path (include/craft_hip.h, "training" section): the element-wise / row-wise forward
// pieces that only exist in training form (softmax from materialised scores, mode pooling of the correlation scores,
// GRU gates as separate stages, dropout) and the backward of every non-GEMM operator.  All of them are HBM-bound streams:
// 16-byte accesses where the layout allows, one wave per token / query row for the row-wise reductions, replicated
// accumulation tables where thousands of blocks reduce into a handful of cells.
#include "launch.hpp"

namespace craft {

__device__ __forceinline__ float block_sum_256(float v, float* sh) {        // sh: >= 4 floats; all 256 threads call
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max_256(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__device__ __forceinline__ bool clamp_active(const unsigned* clamp_ord) {
  // craft_score_max leaves 0 when no score can exceed the clip threshold, else the ordered-uint global max
  if (!clamp_ord) return false;
  const unsigned u = *clamp_ord;
  return u != 0u && ord2f(u) > CRAFT_ATTN_CLIP;
}

// ---------------------------------------------------------------------------------------------
// column sums: out[c] += sum_r x[r][c]   (bias gradients)
// ---------------------------------------------------------------------------------------------
// block = 64 columns x 4 row-slices (one wave each); grid (row chunks, column groups): every wave reads whole 256-byte row
// segments and the four slices of a block are combined in LDS before ONE atomic per column.
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ x, long ld, long rows, int C, int rows_per_block,
                                                float* __restrict__ out) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lane;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    long r = r0 + wv;
    for (; r + 4 < r1; r += 8) { s0 += x[r * ld + c]; s1 += x[(r + 4) * ld + c]; }
    if (r < r1) s0 += x[r * ld + c];
  }
  part[wv][lane] = s0 + s1;
  __syncthreads();
  if (wv == 0 && c < C) unsafeAtomicAdd(out + c, part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
}
int launch_colsum(const float* x, long ld, long rows, int C, float* out, hipStream_t s) {
  if (rows <= 0 || C <= 0) return 0;
  const int cg = (C + 63) / 64;
  // enough blocks to fill the chip: ~2048 in total, at least 32 rows each
  long chunks = 2048 / cg;
  if (chunks < 1) chunks = 1;
  long rpb = (rows + chunks - 1) / chunks;
  if (rpb < 32) rpb = 32;
  hipLaunchKernelGGL(k_colsum, dim3((unsigned)((rows + rpb - 1) / rpb), cg), dim3(256), 0, s, x, ld, rows, C, (int)rpb, out);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// activations: forward in place-capable, backward from the OUTPUT (relu: y > 0, tanh: 1 - y^2, sigmoid: y (1 - y))
// ---------------------------------------------------------------------------------------------
__global__ void k_act_fwd(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy, long rows, int C, int act, float scale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C >> 2;
  if (i >= rows * c4) return;
  const long r = i / c4;
  const int c = (int)(i - r * c4) * 4;
  float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
  if (act == CRAFT_ACT_SIGMOID) { v.x = sigmoid_precise(v.x); v.y = sigmoid_precise(v.y); v.z = sigmoid_precise(v.z); v.w = sigmoid_precise(v.w); }
  else { v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act); }
  v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
  *reinterpret_cast<float4*>(y + r * ldy + c) = v;
}
// dy2 (optional): a second gradient contribution, dy + dy2 is what flows back; ztail: the last `ztail` of the C channels carry no
// gradient (the two pass-through flow channels of the motion encoder's output, update.py:93-94) -- dx = 0 there.
__global__ void k_act_bwd(const float* __restrict__ dy, long lddy, const float* __restrict__ dy2, long lddy2, const float* __restrict__ y,
                          long ldy, float* __restrict__ dx, long lddx, long rows, int C, int act, float scale, int ztail) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C >> 2;
  if (i >= rows * c4) return;
  const long r = i / c4;
  const int c = (int)(i - r * c4) * 4;
  float4 g = *reinterpret_cast<const float4*>(dy + r * lddy + c);
  if (dy2) {
    const float4 g2 = *reinterpret_cast<const float4*>(dy2 + r * lddy2 + c);
    g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
  }
  if (c + 3 >= C - ztail) {
    if (c >= C - ztail) g.x = 0.f;
    if (c + 1 >= C - ztail) g.y = 0.f;
    if (c + 2 >= C - ztail) g.z = 0.f;
    g.w = 0.f;
  }
  float4 o = {g.x * scale, g.y * scale, g.z * scale, g.w * scale};
  if (act != CRAFT_ACT_NONE) {
    const float4 v = *reinterpret_cast<const float4*>(y + r * ldy + c);
    auto d = [act](float yy) { return act == CRAFT_ACT_RELU ? (yy > 0.f ? 1.f : 0.f) : act == CRAFT_ACT_TANH ? 1.f - yy * yy : yy * (1.f - yy); };
    o.x *= d(v.x); o.y *= d(v.y); o.z *= d(v.z); o.w *= d(v.w);
  }
  *reinterpret_cast<float4*>(dx + r * lddx + c) = o;
}
int launch_act_fwd(const float* x, long ldx, float* y, long ldy, long rows, int C, int act, float scale, hipStream_t s) {
  if (rows <= 0 || C <= 0) return 0;
  if ((C & 3) || (ldx & 3) || (ldy & 3)) return CRAFT_ERR_ALIGN;
  const long n = rows * (C >> 2);
  hipLaunchKernelGGL(k_act_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ldx, y, ldy, rows, C, act, scale);
  return (int)hipGetLastError();
}
int launch_act_bwd(const float* dy, long lddy, const float* dy2, long lddy2, const float* y, long ldy, float* dx, long lddx, long rows, int C,
                   int act, float scale, int ztail, hipStream_t s) {
  if (rows <= 0 || C <= 0) return 0;
  if ((C & 3) || (lddy & 3) || (ldy & 3) || (lddx & 3) || (dy2 && (lddy2 & 3))) return CRAFT_ERR_ALIGN;
  if (ztail < 0 || ztail > C) return CRAFT_ERR_ARG;
  const long n = rows * (C >> 2);
  hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dy, lddy, dy2, lddy2, y, ldy, dx, lddx, rows, C, act, scale,
                     ztail);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// dropout (setrans.py:553-557 att_dropout, :791-795 vispos dropout): y = x * keep / (1 - p), keep = hash(seed, index) >= p.
// Counter-based, so the backward regenerates the mask by running the same kernel on the gradient.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned mix32(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return (unsigned)k;
}
__global__ void k_dropout(const float* __restrict__ x, float* __restrict__ y, long n, float p, unsigned long long seed) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned thr = (unsigned)fminf(p * 4294967296.f, 4294967040.f);
  const float inv = 1.f / (1.f - p);
  y[i] = mix32(seed * 0x9E3779B97F4A7C15ULL + (unsigned long long)i) >= thr ? x[i] * inv : 0.f;
}
int launch_dropout(const float* x, float* y, long n, float p, unsigned long long seed, hipStream_t s) {
  if (n <= 0) return 0;
  if (p < 0.f || p >= 1.f) return CRAFT_ERR_ARG;
  hipLaunchKernelGGL(k_dropout, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n, p, seed);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// backward of craft_tokens (token-major source): y = LN(act(x[c_off : c_off+C])) -> dx.  One wave per token, C <= 256.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tokens_bwd(const float* __restrict__ x, long ldx, const float* __restrict__ dy, long lddy,
                                                    float* __restrict__ dx, long lddx, long rows, int C, int act, int do_ln) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float a[4], g[4];
  float s1 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    a[i] = c < C ? act_apply(x[r * ldx + c], act) : 0.f;
    g[i] = c < C ? dy[r * lddy + c] : 0.f;
    s1 += a[i];
  }
  float da[4];
  if (do_ln) {
    const float mean = wave_sum(s1) / C;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int c = lane + 64 * i; const float d = c < C ? a[i] - mean : 0.f; s2 += d * d; }
    const float rstd = rsqrtf(wave_sum(s2) / C + CRAFT_LN_EPS);
    float sg = 0.f, sgy = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float yv = (a[i] - mean) * rstd; sg += g[i]; sgy += g[i] * yv; }
    const float mg = wave_sum(sg) / C, mgy = wave_sum(sgy) / C;
#pragma unroll
    for (int i = 0; i < 4; ++i) da[i] = rstd * (g[i] - mg - (a[i] - mean) * rstd * mgy);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) da[i] = g[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < C) {
      const float d = act == CRAFT_ACT_RELU ? (a[i] > 0.f ? 1.f : 0.f) : act == CRAFT_ACT_TANH ? 1.f - a[i] * a[i] : 1.f;
      dx[r * lddx + c] = da[i] * d;
    }
  }
}
int launch_tokens_bwd(const float* x, long ldx, const float* dy, long lddy, float* dx, long lddx, long rows, int C, int act, int do_ln,
                      hipStream_t s) {
  if (rows <= 0) return 0;
  if (C <= 0 || C > 256 || act == CRAFT_ACT_SIGMOID) return CRAFT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_tokens_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, ldx, dy, lddy, dx, lddx, rows, C, act, do_ln);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// softmax over keys from materialised scores (CrossAttFeatTrans, setrans.py:520-551), in place:
//   P[z][i][j] = softmax_j( clamp?(S[z][i][j]) + pos_w * pb(i, j) + mask(i, j) ),   z = b*M + m, columns [N, ld) = 0.
// One block per query row; the row lives in LDS between the passes.  clampbits (optional, [rows][ld/32] words): bit j of row
// = "this score was clamped" (its gradient is zero), written only when the clamp is active.
// ---------------------------------------------------------------------------------------------
// One block per query row, the row in REGISTERS: thread t holds the float4s t, t + 256, .. (ITER of them, every load issued before the
// first use), so the row crosses HBM -> registers -> HBM once with 16-byte accesses and the two block reductions are the only syncs
// (the first version staged the row in LDS with 4-byte loads and divided by W8 per element: 2.8 TB/s of the 8).
template <int ITER>
__global__ __launch_bounds__(256) void k_attn_softmax_fwd(float* __restrict__ S, long ld, int N, int H8, int W8,
                                                          const float* __restrict__ pos_tab, int R, float pos_w, int mask_radius,
                                                          const unsigned* __restrict__ clamp_ord, unsigned* __restrict__ clampbits,
                                                          float* __restrict__ Pdrop, float drop_p, unsigned long long seed,
                                                          unsigned short* __restrict__ Ppk, long pk_rows, int pk_np, int pk_prec, long nrows_grid) {
  __shared__ float red[4];
  __shared__ float stab[31 * 31];
  // Ppk: the (dropped) probabilities also as a packed MFMA operand (craft_gemm_pk): [plane][j / 32][z * pk_np + i][32], pk_np >= N rows
  // per batch -- the grid then covers the pk_np - N padding rows too (zeros: the contraction over i of dV = P^T dO runs over them)
  // Consecutive blocks go to different XCDs (8 L2s), but the packed output of a row is 64-byte pieces whose neighbours in memory belong
  // to rows i +- 1: runs of 8 consecutive rows are mapped to ONE XCD, back to back, so that its L2 merges their pieces into full lines
  const int rows_per = Ppk ? pk_np : N;
  const long seq = blockIdx.x >> 3;
  const long row = ((seq >> 3) * 8 + (blockIdx.x & 7)) * 8 + (seq & 7);
  if (row >= nrows_grid) return;
  const long z = row / rows_per;
  const int i = (int)(row - z * rows_per);
  const long pk_plane = (ld >> 5) * pk_rows * 32;
  unsigned short* const Pr = Ppk ? Ppk + (z * pk_np + i) * 32 : nullptr;
  const int tid = threadIdx.x;
  if (i >= N) {
    const f16x4 zero = {0, 0, 0, 0};
    for (int j = tid * 4; j < ld; j += 1024) {
      unsigned short* o = Pr + (long)(j >> 5) * pk_rows * 32 + (j & 31);
      *reinterpret_cast<f16x4*>(o) = zero;
      if (pk_prec == CRAFT_PREC_F16X3) *reinterpret_cast<f16x4*>(o + pk_plane) = zero;
    }
    return;
  }
  const long rr = z * N + i;                    // (z, i)
  const int hi = i / W8, wi = i - hi * W8;
  float* Sr = S + rr * ld;
  float4 v[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int j = (tid + 256 * it) * 4;
    v[it] = j < ld ? *reinterpret_cast<const float4*>(Sr + j) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int T = 2 * R + 1;
  if (pos_tab) {
    for (int t = tid; t < T * T; t += 256) stab[t] = pos_tab[t] * pos_w;
    __syncthreads();
  }
  const bool clamp = clamp_active(clamp_ord);
  float mx = -3.0e38f;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int j0 = (tid + 256 * it) * 4;
    int hj = j0 / W8, wj = j0 - hj * W8;
    float* e = reinterpret_cast<float*>(&v[it]);
    unsigned hit = 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x = e[c];
      if (j0 + c < N) {
        if (clamp) { if (x > CRAFT_ATTN_CLIP || x < -CRAFT_ATTN_CLIP) hit |= 1u << c; x = fminf(fmaxf(x, -CRAFT_ATTN_CLIP), CRAFT_ATTN_CLIP); }
        const int dh = hj - hi, dw = wj - wi;
        if (pos_tab && dh >= -R && dh <= R && dw >= -R && dw <= R) x += stab[(dh + R) * T + dw + R];
        if (mask_radius > 0 && max(abs(dh), abs(dw)) > mask_radius) x += -1e9f;
      } else {
        x = -3.0e38f;
      }
      e[c] = x;
      mx = fmaxf(mx, x);
      if (++wj == W8) { wj = 0; ++hj; }
    }
    if (clampbits && clamp) {
      // bit j of the row's words = "score j was clamped": the 8 lanes of a 32-column word OR their nibbles together
      unsigned w = hit << (4 * (tid & 7));
      w |= __shfl_xor(w, 1); w |= __shfl_xor(w, 2); w |= __shfl_xor(w, 4);
      if ((tid & 7) == 0 && j0 < ld) clampbits[rr * (ld >> 5) + (j0 >> 5)] = w;
    }
  }
  mx = block_max_256(mx, red);
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    float* e = reinterpret_cast<float*>(&v[it]);
#pragma unroll
    for (int c = 0; c < 4; ++c) { e[c] = expf(e[c] - mx); sum += e[c]; }      // (columns >= N: exp(-3e38 - mx) = 0)
  }
  sum = block_sum_256(sum, red);
  const float inv = 1.f / sum;
  // the dropout of the probabilities (setrans.py:553-557) in the same pass: P stays in S (the softmax backward needs it), the dropped
  // copy goes to Pdrop (fp32) and / or Ppk (packed) -- the mask of k_dropout over the same flat element index
  const bool drop = drop_p > 0.f && (Pdrop != nullptr || Ppk != nullptr);
  const unsigned thr = (unsigned)fminf(drop_p * 4294967296.f, 4294967040.f);
  const float dinv = drop ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int j = (tid + 256 * it) * 4;
    if (j >= ld) continue;
    float4 pv = make_float4(v[it].x * inv, v[it].y * inv, v[it].z * inv, v[it].w * inv);
    *reinterpret_cast<float4*>(Sr + j) = pv;
    if (drop) {
      float* e = reinterpret_cast<float*>(&pv);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        e[c] = mix32(seed * 0x9E3779B97F4A7C15ULL + (unsigned long long)(rr * ld + j + c)) >= thr ? e[c] * dinv : 0.f;
    }
    if (Pdrop) *reinterpret_cast<float4*>(Pdrop + rr * ld + j) = pv;
    if (Ppk) {
      unsigned short* o = Pr + (long)(j >> 5) * pk_rows * 32 + (j & 31);
      if (pk_prec == CRAFT_PREC_F16X3) {
        f16x4 h, l;
        split_f16x3(pv, h, l);
        *reinterpret_cast<f16x4*>(o) = h;
        *reinterpret_cast<f16x4*>(o + pk_plane) = l;
      } else if (pk_prec == CRAFT_PREC_F16) {
        f16x4 h;
        h[0] = (_Float16)pv.x; h[1] = (_Float16)pv.y; h[2] = (_Float16)pv.z; h[3] = (_Float16)pv.w;
        *reinterpret_cast<f16x4*>(o) = h;
      } else {
        bf16x4 h;
        h[0] = (__bf16)pv.x; h[1] = (__bf16)pv.y; h[2] = (__bf16)pv.z; h[3] = (__bf16)pv.w;
        *reinterpret_cast<bf16x4*>(o) = h;
      }
    }
  }
}
int launch_attn_softmax_fwd(float* S, long ld, int B, int M, int H8, int W8, const float* pos_tab, int R, float pos_w, int mask_radius,
                            const unsigned* clamp_ord, unsigned* clampbits, float* Pdrop, float drop_p, unsigned long long seed,
                            void* Ppk, long pk_rows, int pk_np, int pk_prec, hipStream_t s) {
  const int N = H8 * W8;
  if (N <= 0 || B <= 0) return 0;
  if (N > 16000 || (ld & 31) || ld < N) return CRAFT_ERR_UNSUPPORTED;
  if ((Pdrop != nullptr || Ppk != nullptr) && (drop_p < 0.f || drop_p >= 1.f)) return CRAFT_ERR_ARG;
  if (Ppk != nullptr) {
    if (pk_np < N || (pk_np & 31) || pk_rows < (long)B * M * pk_np || (reinterpret_cast<uintptr_t>(Ppk) & 15)) return CRAFT_ERR_ARG;
    if (pk_prec != CRAFT_PREC_F16X3 && pk_prec != CRAFT_PREC_F16 && pk_prec != CRAFT_PREC_BF16) return CRAFT_ERR_UNSUPPORTED;
  }
  if ((reinterpret_cast<uintptr_t>(S) & 15) || (Pdrop && (reinterpret_cast<uintptr_t>(Pdrop) & 15))) return CRAFT_ERR_ALIGN;
  if (R > 15) return CRAFT_ERR_UNSUPPORTED;
  const long nrows_grid = (long)B * M * (Ppk ? pk_np : N);
  const dim3 grid((unsigned)((nrows_grid + 63) / 64 * 64));
  const int iter = (int)((ld + 1023) / 1024);
#define GO(I) hipLaunchKernelGGL(k_attn_softmax_fwd<I>, grid, dim3(256), 0, s, S, ld, N, H8, W8, pos_tab, R, pos_w, mask_radius, clamp_ord, clampbits, \
                                 Pdrop, drop_p, seed, static_cast<unsigned short*>(Ppk), pk_rows, pk_np, pk_prec, nrows_grid)
  if (iter <= 1) GO(1); else if (iter <= 2) GO(2); else if (iter <= 3) GO(3); else if (iter <= 4) GO(4); else if (iter <= 5) GO(5);
  else if (iter <= 6) GO(6); else if (iter <= 8) GO(8); else GO(16);
#undef GO
  return (int)hipGetLastError();
}

// backward: dS = P * (dP - sum_j dP*P), zero where the score was clamped; dtab[rep][(dh+R)*(2R+1)+(dw+R)] += pos_w * dS' over
// the window (dS' = before the clamp mask: the bias is added after the clamp).  dP is overwritten with dS.
constexpr int SM_ROWS_PER_BLOCK = 8;
// rows in registers as in the forward (P and the incoming gradient: 2 ITER float4 per thread, all loads of a row issued together)
template <int ITER>
__global__ __launch_bounds__(256) void k_attn_softmax_bwd(const float* __restrict__ P, float* __restrict__ dP, long ld, int N, int H8,
                                                          int W8, int R, float pos_w, const unsigned* __restrict__ clamp_ord,
                                                          const unsigned* __restrict__ clampbits, float* __restrict__ dtab, long nrows,
                                                          float drop_p, unsigned long long seed,
                                                          unsigned short* __restrict__ dSpk, long pk_rows, int pk_np, int pk_prec) {
  __shared__ float red[4];
  __shared__ float tab[32 * 32];
  // dSpk: dS leaves as a packed operand of craft_gemm_pk ([plane][j / 32][z * pk_np + i][32], rows >= N of a batch zero) INSTEAD of fp32
  // over dP -- its only consumers are dQ = dS K and dK = dS^T Q.  The row index then runs over the padded rows (nrows = Z * pk_np).
  const int rows_per = dSpk ? pk_np : N;
  const long pk_plane = (ld >> 5) * pk_rows * 32;
  // drop_p > 0: dP is the gradient w.r.t. the DROPPED probabilities (k_attn_softmax_fwd's Pdrop / Ppk): the dropout backward (the same
  // mask) is applied while the row is read
  const unsigned thr = (unsigned)fminf(drop_p * 4294967296.f, 4294967040.f);
  const float dinv = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int T = 2 * R + 1, tid = threadIdx.x;
  const bool want_tab = dtab != nullptr && R >= 0;
  if (want_tab) for (int t = tid; t < T * T; t += 256) tab[t] = 0.f;
  const bool clamp = clamp_active(clamp_ord) && clampbits != nullptr;
  const long r0 = (long)blockIdx.x * SM_ROWS_PER_BLOCK;
  for (long rp = r0; rp < min(nrows, r0 + SM_ROWS_PER_BLOCK); ++rp) {
    const long z = rp / rows_per;
    const int i = (int)(rp - z * rows_per);
    unsigned short* const Dr = dSpk ? dSpk + (z * pk_np + i) * 32 : nullptr;
    if (i >= N) {                                                    // a padding row of the pack
      const f16x4 zero = {0, 0, 0, 0};
      for (int j = tid * 4; j < ld; j += 1024) {
        unsigned short* o = Dr + (long)(j >> 5) * pk_rows * 32 + (j & 31);
        *reinterpret_cast<f16x4*>(o) = zero;
        if (pk_prec == CRAFT_PREC_F16X3) *reinterpret_cast<f16x4*>(o + pk_plane) = zero;
      }
      continue;
    }
    const long rr = z * N + i;
    const int hi = i / W8, wi = i - hi * W8;
    const float* Pr = P + rr * ld;
    float* Gr = dP + rr * ld;
    float4 pv[ITER], gv[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int j = (tid + 256 * it) * 4;
      const bool in = j < ld;
      pv[it] = in ? *reinterpret_cast<const float4*>(Pr + j) : make_float4(0.f, 0.f, 0.f, 0.f);
      gv[it] = in ? *reinterpret_cast<const float4*>(Gr + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float dot = 0.f;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int j = (tid + 256 * it) * 4;
      float* g = reinterpret_cast<float*>(&gv[it]);
      const float* pp = reinterpret_cast<const float*>(&pv[it]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float x = j + c < N ? g[c] : 0.f;                            // (the padding columns of dP may hold anything)
        if (drop_p > 0.f) x = mix32(seed * 0x9E3779B97F4A7C15ULL + (unsigned long long)(rr * ld + j + c)) >= thr ? x * dinv : 0.f;
        g[c] = x;
        dot += pp[c] * x;
      }
    }
    dot = block_sum_256(dot, red);
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int j = (tid + 256 * it) * 4;
      if (j >= ld) continue;
      const float* g = reinterpret_cast<const float*>(&gv[it]);
      const float* pp = reinterpret_cast<const float*>(&pv[it]);
      float4 o;
      float* ds = reinterpret_cast<float*>(&o);
      int hj = j / W8, wj = j - hj * W8;
      unsigned bits = 0u;
      if (clamp) bits = clampbits[rr * (ld >> 5) + (j >> 5)] >> (j & 31);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ds[c] = pp[c] * (g[c] - dot);                                // (columns >= N: P = 0)
        if (want_tab && j + c < N) {
          const int dh = hj - hi, dw = wj - wi;
          if (dh >= -R && dh <= R && dw >= -R && dw <= R) atomicAdd(&tab[(dh + R) * T + dw + R], ds[c]);
        }
        if ((bits >> c) & 1u) ds[c] = 0.f;
        if (++wj == W8) { wj = 0; ++hj; }
      }
      if (Dr == nullptr) {
        *reinterpret_cast<float4*>(Gr + j) = o;
      } else {
        unsigned short* q = Dr + (long)(j >> 5) * pk_rows * 32 + (j & 31);
        if (pk_prec == CRAFT_PREC_F16X3) {
          f16x4 h, l;
          split_f16x3(o, h, l);
          *reinterpret_cast<f16x4*>(q) = h;
          *reinterpret_cast<f16x4*>(q + pk_plane) = l;
        } else if (pk_prec == CRAFT_PREC_F16) {
          f16x4 h;
          h[0] = (_Float16)o.x; h[1] = (_Float16)o.y; h[2] = (_Float16)o.z; h[3] = (_Float16)o.w;
          *reinterpret_cast<f16x4*>(q) = h;
        } else {
          bf16x4 h;
          h[0] = (__bf16)o.x; h[1] = (__bf16)o.y; h[2] = (__bf16)o.z; h[3] = (__bf16)o.w;
          *reinterpret_cast<bf16x4*>(q) = h;
        }
      }
    }
  }
  if (want_tab) {
    __syncthreads();
    float* rep = dtab + (long)(blockIdx.x % CRAFT_STATS_REPLICAS) * T * T;
    for (int t = tid; t < T * T; t += 256) if (tab[t] != 0.f) unsafeAtomicAdd(rep + t, tab[t] * pos_w);
  }
}
int launch_attn_softmax_bwd(const float* P, float* dP, long ld, int B, int M, int H8, int W8, int R, float pos_w,
                            const unsigned* clamp_ord, const unsigned* clampbits, float* dtab, float drop_p, unsigned long long seed,
                            void* dSpk, long pk_rows, int pk_np, int pk_prec, hipStream_t s) {
  const int N = H8 * W8;
  if (N <= 0 || B <= 0) return 0;
  if (R > 15 || (ld & 31)) return CRAFT_ERR_UNSUPPORTED;
  if (N > 16000 || ld < N) return CRAFT_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(P) & 15) || (reinterpret_cast<uintptr_t>(dP) & 15)) return CRAFT_ERR_ALIGN;
  if (dSpk != nullptr) {
    if (pk_np < N || (pk_np & 31) || pk_rows < (long)B * M * pk_np || (reinterpret_cast<uintptr_t>(dSpk) & 15)) return CRAFT_ERR_ARG;
    if (pk_prec != CRAFT_PREC_F16X3 && pk_prec != CRAFT_PREC_F16 && pk_prec != CRAFT_PREC_BF16) return CRAFT_ERR_UNSUPPORTED;
  }
  const long nrows = (long)B * M * (dSpk ? pk_np : N);
  const dim3 grid((unsigned)((nrows + SM_ROWS_PER_BLOCK - 1) / SM_ROWS_PER_BLOCK));
  const int iter = (int)((ld + 1023) / 1024);
#define GO(I) hipLaunchKernelGGL(k_attn_softmax_bwd<I>, grid, dim3(256), 0, s, P, dP, ld, N, H8, W8, R, pos_w, clamp_ord, clampbits, dtab, nrows, drop_p, seed, \
                                 static_cast<unsigned short*>(dSpk), pk_rows, pk_np, pk_prec)
  if (iter <= 1) GO(1); else if (iter <= 2) GO(2); else if (iter <= 3) GO(3); else if (iter <= 4) GO(4); else if (iter <= 5) GO(5);
  else if (iter <= 6) GO(6); else if (iter <= 8) GO(8); else GO(16);
#undef GO
  return (int)hipGetLastError();
}

// sum the replicas of an accumulation table: out[t] += sum_r rep[r][t]
__global__ void k_reduce_replicas(const float* __restrict__ rep, int nrep, int n, float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  float s = 0.f;
  for (int r = 0; r < nrep; ++r) s += rep[(long)r * n + t];
  out[t] += s;
}
int launch_reduce_replicas(const float* rep, int nrep, int n, float* out, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_reduce_replicas, dim3((n + 255) / 256), dim3(256), 0, s, rep, nrep, n, out);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// correlation scores -> pooled volume (TransCorrBlock.corr, corr.py:191-199; setrans.py:520-550) from materialised scores
// S [B][M][N][ld]:  s_m = clamp?(S_m) + pos_w*pb,  c = sum_m s_m softmax_m(w s_m)  -> level 0 [B*N][N] (row stride N),
// sums[b] += (sum c, sum c^2).  w: device pointer to attn_softaggr.feat2score.weight (one float).
// ---------------------------------------------------------------------------------------------
template <int M>
__device__ __forceinline__ float pool_modes(const float (&sm)[M], float w, float (&a)[M]) {
  float mx = w * sm[0];
#pragma unroll
  for (int m = 1; m < M; ++m) mx = fmaxf(mx, w * sm[m]);
  float den = 0.f;
#pragma unroll
  for (int m = 0; m < M; ++m) { a[m] = expf(w * sm[m] - mx); den += a[m]; }
  float c = 0.f;
  const float inv = 1.f / den;
#pragma unroll
  for (int m = 0; m < M; ++m) { a[m] *= inv; c += a[m] * sm[m]; }
  return c;
}
template <int M>
__global__ __launch_bounds__(256) void k_corr_pool_fwd(const float* __restrict__ S, long ld, int N, int H8, int W8,
                                                       const float* __restrict__ pos_tab, int R, float pos_w, const float* __restrict__ wp,
                                                       const unsigned* __restrict__ clamp_ord, float* __restrict__ c0,
                                                       double* __restrict__ sums) {
  __shared__ float red[4];
  const long q = blockIdx.x;                    // (b, i)
  const int b = (int)(q / N), i = (int)(q - (long)b * N);
  const int hi = i / W8, wi = i - hi * W8;
  const bool clamp = clamp_active(clamp_ord);
  const float w = M > 1 ? *wp : 1.f;
  float s1 = 0.f, s2 = 0.f;
  const int dq = 256 / W8, dr = 256 - dq * W8;          // (key row / column advance incrementally: no division per element)
  int hj = (int)threadIdx.x / W8, wj = (int)threadIdx.x - hj * W8;
  for (int j = threadIdx.x; j < N; j += 256, wj += dr, hj += dq) {
    if (wj >= W8) { wj -= W8; ++hj; }
    const float pb = pos_tab ? pos_w * pos_bias_at(pos_tab, R, hi, wi, hj, wj) : 0.f;
    float sm[M], a[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float v = S[(((long)b * M + m) * N + i) * ld + j];
      if (clamp) v = fminf(fmaxf(v, -CRAFT_ATTN_CLIP), CRAFT_ATTN_CLIP);
      sm[m] = v + pb;
    }
    const float c = M > 1 ? pool_modes<M>(sm, w, a) : sm[0];
    c0[q * N + j] = c;
    s1 += c; s2 += c * c;
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) { atomicAdd(&sums[2 * b], (double)s1); atomicAdd(&sums[2 * b + 1], (double)s2); }
}
// The same with four consecutive keys per thread (N % 4 == 0): 16-byte loads of the M score rows and one 16-byte store, i.e. four times the
// bytes in flight per thread -- the scalar form keeps 16 bytes per thread in flight and runs at 2.4 TB/s with eight blocks per CU.
constexpr int POOL_ROWS = 4;
template <int M>
__global__ __launch_bounds__(256) void k_corr_pool_fwd4(const float* __restrict__ S, long ld, int N, int H8, int W8,
                                                        const float* __restrict__ pos_tab, int R, float pos_w, const float* __restrict__ wp,
                                                        const unsigned* __restrict__ clamp_ord, float* __restrict__ c0,
                                                        double* __restrict__ sums) {
  __shared__ float red[4];
  // POOL_ROWS query rows per block (N % 4 == 0: all of one image): the two double atomics per block go to ONE address pair per image,
  // and same-address atomics retire at ~11 ns each -- with a block per row (22 816 at configs[3]) they alone were 0.5 ms, the whole kernel
  const bool clamp = clamp_active(clamp_ord);
  const float w = M > 1 ? *wp : 1.f;
  float s1 = 0.f, s2 = 0.f;
  const int b = (int)(((long)blockIdx.x * POOL_ROWS) / N);
  for (int rr = 0; rr < POOL_ROWS; ++rr) {
  const long q = (long)blockIdx.x * POOL_ROWS + rr;                    // (b, i)
  const int i = (int)(q - (long)b * N);
  const int hi = i / W8, wi = i - hi * W8;
  for (int j0 = threadIdx.x * 4; j0 < N; j0 += 1024) {
    float4 s4[M];
#pragma unroll
    for (int m = 0; m < M; ++m) s4[m] = *reinterpret_cast<const float4*>(S + (((long)b * M + m) * N + i) * ld + j0);
    int hj = j0 / W8, wj = j0 - hj * W8;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pb = pos_tab ? pos_w * pos_bias_at(pos_tab, R, hi, wi, hj, wj) : 0.f;
      float sm[M], a[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        float v = reinterpret_cast<const float*>(&s4[m])[e];
        if (clamp) v = fminf(fmaxf(v, -CRAFT_ATTN_CLIP), CRAFT_ATTN_CLIP);
        sm[m] = v + pb;
      }
      const float c = M > 1 ? pool_modes<M>(sm, w, a) : sm[0];
      o[e] = c;
      s1 += c; s2 += c * c;
      if (++wj == W8) { wj = 0; ++hj; }
    }
    *reinterpret_cast<float4*>(c0 + q * N + j0) = make_float4(o[0], o[1], o[2], o[3]);
  }
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) { atomicAdd(&sums[2 * b], (double)s1); atomicAdd(&sums[2 * b + 1], (double)s2); }
}
int launch_corr_pool_fwd(const float* S, long ld, int B, int M, int H8, int W8, const float* pos_tab, int R, float pos_w, const float* w,
                         const unsigned* clamp_ord, float* c0, double* sums, hipStream_t s) {
  const int N = H8 * W8;
  if (B <= 0 || N <= 0) return 0;
  dim3 grid((unsigned)((long)B * N));
  if ((N & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(S) & 15) == 0 && (reinterpret_cast<uintptr_t>(c0) & 15) == 0) {
    grid = dim3((unsigned)((long)B * N / POOL_ROWS));
    if (M == 4) { hipLaunchKernelGGL((k_corr_pool_fwd4<4>), grid, dim3(256), 0, s, S, ld, N, H8, W8, pos_tab, R, pos_w, w, clamp_ord, c0, sums); return (int)hipGetLastError(); }
    if (M == 1) { hipLaunchKernelGGL((k_corr_pool_fwd4<1>), grid, dim3(256), 0, s, S, ld, N, H8, W8, pos_tab, R, pos_w, w, clamp_ord, c0, sums); return (int)hipGetLastError(); }
  }
  if (M == 4) hipLaunchKernelGGL((k_corr_pool_fwd<4>), grid, dim3(256), 0, s, S, ld, N, H8, W8, pos_tab, R, pos_w, w, clamp_ord, c0, sums);
  else if (M == 1) hipLaunchKernelGGL((k_corr_pool_fwd<1>), grid, dim3(256), 0, s, S, ld, N, H8, W8, pos_tab, R, pos_w, w, clamp_ord, c0, sums);
  else if (M == 2) hipLaunchKernelGGL((k_corr_pool_fwd<2>), grid, dim3(256), 0, s, S, ld, N, H8, W8, pos_tab, R, pos_w, w, clamp_ord, c0, sums);
  else return CRAFT_ERR_UNSUPPORTED;
  return (int)hipGetLastError();
}

// backward: G [B*N][N] = gradient w.r.t. the NORMALISED level 0 (all pyramid levels folded in, k_corr_pyramid_bwd), gstats[b] =
// (sum G, sum G*c_hat).  dc = rstd (G - mean G - c_hat mean(G c_hat));  dS'_m = dc a_m (1 + w (s_m - c));  dw += dc sum_m a_m s_m
// (s_m - c);  dtab += pos_w sum_m dS'_m;  dS_m = clamped ? 0 : dS'_m  (written over S).
template <int M>
__global__ __launch_bounds__(256) void k_corr_pool_bwd(float* __restrict__ S, long ld, int N, int H8, int W8,
                                                       const float* __restrict__ pos_tab, int R, float pos_w, const float* __restrict__ wp,
                                                       const unsigned* __restrict__ clamp_ord, const float* __restrict__ c0,
                                                       const float* __restrict__ G, const float* __restrict__ mu_rstd,
                                                       const double* __restrict__ gstats, int do_norm, float* __restrict__ dtab,
                                                       double* __restrict__ dw, int rpb) {
  __shared__ float red[4];
  __shared__ float tab[32 * 32];
  const int T = 2 * R + 1;
  const bool want_tab = dtab != nullptr && pos_tab != nullptr;
  if (want_tab) for (int t = threadIdx.x; t < T * T; t += 256) tab[t] = 0.f;
  __syncthreads();
  // rpb query rows per block (all of one image): ONE double atomic on dw and one flush of the bias-table partials per block instead of
  // per row -- 22 816 same-address double atomics (~11 ns each) were half of this kernel's time
  const int b = (int)(((long)blockIdx.x * rpb) / N);
  const bool clamp = clamp_active(clamp_ord);
  const float w = M > 1 ? *wp : 1.f;
  const float mu = mu_rstd[2 * b], rstd = mu_rstd[2 * b + 1];
  const double cnt = (double)N * N;
  const float mg = do_norm ? (float)(gstats[2 * b] / cnt) : 0.f, mgc = do_norm ? (float)(gstats[2 * b + 1] / cnt) : 0.f;
  float dwl = 0.f;
  for (int rr = 0; rr < rpb; ++rr) {
  const long q = (long)blockIdx.x * rpb + rr;
  const int i = (int)(q - (long)b * N);
  const int hi = i / W8, wi = i - hi * W8;
  const int dq = 256 / W8, dr = 256 - dq * W8;          // (key row / column advance incrementally: no division per element)
  int hj = (int)threadIdx.x / W8, wj = (int)threadIdx.x - hj * W8;
  for (int j = threadIdx.x; j < N; j += 256, wj += dr, hj += dq) {
    if (wj >= W8) { wj -= W8; ++hj; }
    const int dh = hj - hi, dwd = wj - wi;
    const bool inwin = pos_tab && dh >= -R && dh <= R && dwd >= -R && dwd <= R;
    const float pb = inwin ? pos_w * pos_tab[(dh + R) * T + dwd + R] : 0.f;
    float sm[M], a[M];
    bool hit[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float v = S[(((long)b * M + m) * N + i) * ld + j];
      hit[m] = clamp && (v > CRAFT_ATTN_CLIP || v < -CRAFT_ATTN_CLIP);
      if (clamp) v = fminf(fmaxf(v, -CRAFT_ATTN_CLIP), CRAFT_ATTN_CLIP);
      sm[m] = v + pb;
    }
    const float g = G[q * N + j];
    const float chat = (c0[q * N + j] - mu) * rstd;
    const float dc = do_norm ? rstd * (g - mg - chat * mgc) : g;
    float tsum = 0.f;
    if (M > 1) {
      const float c = pool_modes<M>(sm, w, a);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const float ds = dc * a[m] * (1.f + w * (sm[m] - c));
        dwl += dc * a[m] * sm[m] * (sm[m] - c);
        tsum += ds;
        S[(((long)b * M + m) * N + i) * ld + j] = hit[m] ? 0.f : ds;
      }
    } else {
      tsum = dc;
      S[((long)b * N + i) * ld + j] = hit[0] ? 0.f : dc;
    }
    if (inwin && want_tab) atomicAdd(&tab[(dh + R) * T + dwd + R], tsum);
  }
  }
  dwl = block_sum_256(dwl, red);
  if (threadIdx.x == 0 && dw && M > 1) atomicAdd(dw, (double)dwl);
  if (want_tab) {
    __syncthreads();
    float* rep = dtab + (long)(blockIdx.x % CRAFT_STATS_REPLICAS) * T * T;
    for (int t = threadIdx.x; t < T * T; t += 256) if (tab[t] != 0.f) unsafeAtomicAdd(rep + t, tab[t] * pos_w);
  }
}
int launch_corr_pool_bwd(float* S, long ld, int B, int M, int H8, int W8, const float* pos_tab, int R, float pos_w, const float* w,
                         const unsigned* clamp_ord, const float* c0, const float* G, const float* mu_rstd, const double* gstats,
                         int do_norm, float* dtab, double* dw, hipStream_t s) {
  const int N = H8 * W8;
  if (B <= 0 || N <= 0) return 0;
  if (R > 15) return CRAFT_ERR_UNSUPPORTED;
  const int rpb = 1;       // (4 rows per block measured slower here, 578 vs 546 us: this kernel's single atomic per block hides behind its other work)
  dim3 grid((unsigned)((long)B * N / rpb));
#define GO(MM) hipLaunchKernelGGL((k_corr_pool_bwd<MM>), grid, dim3(256), 0, s, S, ld, N, H8, W8, pos_tab, R, pos_w, w, clamp_ord, c0, G, \
                                  mu_rstd, gstats, do_norm, dtab, dw, rpb)
  if (M == 4) GO(4); else if (M == 1) GO(1); else if (M == 2) GO(2); else return CRAFT_ERR_UNSUPPORTED;
#undef GO
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// pyramid backward: fold the gradients of levels 1..3 into level 0 (avg_pool2d 2x2, floor sizes: corr.py:186-189) and reduce
// the two sums the global-LayerNorm backward needs.  G0[q][y][x] += G1[q][y/2][x/2]/4 + G2[..]/16 + G3[..]/64 where the cell
// exists.  gstats[b] += (sum G0, sum G0 * c_hat).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_corr_pyramid_bwd(float* __restrict__ G0, const float* __restrict__ G1, const float* __restrict__ G2,
                                                          const float* __restrict__ G3, const float* __restrict__ c0,
                                                          const float* __restrict__ mu_rstd, int N, int H8, int W8,
                                                          double* __restrict__ gstats, int rpb) {
  __shared__ float red[4];
  // rpb query rows per block (all of one image): one pair of double atomics per block -- a pair per row (45 632 same-address atomics of
  // ~11 ns) was as long as the rest of the kernel (k_corr_pool_fwd4)
  const int b = (int)(((long)blockIdx.x * rpb) / N);
  const float mu = mu_rstd[2 * b], rstd = mu_rstd[2 * b + 1];
  const int h1 = H8 >> 1, w1 = W8 >> 1, h2 = h1 >> 1, w2 = w1 >> 1, h3 = h2 >> 1, w3 = w2 >> 1;
  float s1 = 0.f, s2 = 0.f;
  for (int rr = 0; rr < rpb; ++rr) {
  const long q = (long)blockIdx.x * rpb + rr;
  // Every load of an element is unconditional (clamped cell of the coarser level, the bounds as 0 / 1 weights; an absent level reads G0
  // with weight 0): under the bounds tests each was branch + load + vmcnt(0), four dependent round trips per element.  (y, x) advance
  // incrementally (a run-time division per element is ~30 VALU instructions).
  const float* L1 = G1 ? G1 + q * h1 * w1 : G0 + q * N;
  const float* L2 = G2 ? G2 + q * h2 * w2 : G0 + q * N;
  const float* L3 = G3 ? G3 + q * h3 * w3 : G0 + q * N;
  const float k1 = G1 ? 0.25f : 0.f, k2 = G2 ? 0.0625f : 0.f, k3 = G3 ? 0.015625f : 0.f;
  const int dq = 256 / W8, dr = 256 - dq * W8;
  int y = (int)threadIdx.x / W8, x = (int)threadIdx.x - y * W8;
  for (int j = threadIdx.x; j < N; j += 256) {
    const float g0 = G0[q * N + j];
    const float cc = c0[q * N + j];
    const float g1 = L1[min(y >> 1, max(h1 - 1, 0)) * w1 + min(x >> 1, max(w1 - 1, 0))];
    const float g2 = L2[min(y >> 2, max(h2 - 1, 0)) * w2 + min(x >> 2, max(w2 - 1, 0))];
    const float g3 = L3[min(y >> 3, max(h3 - 1, 0)) * w3 + min(x >> 3, max(w3 - 1, 0))];
    const bool in1 = G1 != nullptr && (y >> 1) < h1 && (x >> 1) < w1;
    const bool in2 = G2 != nullptr && (y >> 2) < h2 && (x >> 2) < w2 && (y >> 1) < 2 * h2 && (x >> 1) < 2 * w2;
    const bool in3 = G3 != nullptr && (y >> 3) < h3 && (x >> 3) < w3 && (y >> 2) < 2 * h3 && (x >> 2) < 2 * w3 && (y >> 1) < 2 * h2 && (x >> 1) < 2 * w2;
    float g = g0;
    g += in1 ? k1 * g1 : 0.f;
    g += in2 ? k2 * g2 : 0.f;
    g += in3 ? k3 * g3 : 0.f;
    G0[q * N + j] = g;
    s1 += g;
    s2 += g * (cc - mu) * rstd;
    x += dr; y += dq;
    if (x >= W8) { x -= W8; ++y; }
  }
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) { atomicAdd(&gstats[2 * b], (double)s1); atomicAdd(&gstats[2 * b + 1], (double)s2); }
}
int launch_corr_pyramid_bwd(float* G0, const float* G1, const float* G2, const float* G3, const float* c0, const float* mu_rstd, int B,
                            int H8, int W8, double* gstats, hipStream_t s) {
  const int N = H8 * W8;
  if (B <= 0 || N <= 0) return 0;
  const int rpb = (N % POOL_ROWS == 0) ? POOL_ROWS : 1;
  hipLaunchKernelGGL(k_corr_pyramid_bwd, dim3((unsigned)((long)B * N / rpb)), dim3(256), 0, s, G0, G1, G2, G3, c0, mu_rstd, N, H8, W8, gstats, rpb);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// lookup backward (CorrBlock.__call__, corr.py:47-71): w * g of every bilinear tap goes into the gradient of the NORMALISED
// pyramid level it sampled (zero padding: out-of-image taps have no gradient).  One wave per query.  The (2r+1)^2 taps of a level
// cover a (2r+2)^2 footprint of that query's own correlation image, each cell fed by up to 4 taps: a lane GATHERS the
// contributions of its cell and adds them with one plain read-modify-write (nothing else touches query q's images during a
// launch) -- 100 coalesced updates per level instead of 324 fp32 atomics (the scatter form took 246 us per call at configs[3]).
// ---------------------------------------------------------------------------------------------
template <int RADIUS>       // > 0: compile-time window radius (constant divisors, as k_corr_lookup); 0: run-time
__global__ __launch_bounds__(256) void k_corr_lookup_bwd(const float* __restrict__ dout, long ldo, const float* __restrict__ coords,
                                                         float* __restrict__ G0, float* __restrict__ G1, float* __restrict__ G2,
                                                         float* __restrict__ G3, int levels, int H8, int W8, int radius, int lvl_stride,
                                                         int col_off, long nq) {
  const int lane = threadIdx.x & 63;
  const long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const float cx = coords[2 * q], cy = coords[2 * q + 1];
  if (RADIUS) radius = RADIUS;
  const int win = RADIUS ? 2 * RADIUS + 1 : 2 * radius + 1, fp = win + 1;
  float* Gl[4] = {G0, G1, G2, G3};
  // As in k_corr_lookup: every load of the query is requested before any of them is used, from clamped addresses (a load under a
  // bounds test compiles to branch + load + s_waitcnt vmcnt(0): 40 dependent round trips per query in the first version), the bounds
  // live on as 0 / 1 weights and as the store's predicate.  NI = 64-lane passes over the (2r+2)^2 footprint.
  constexpr int NI = RADIUS ? ((2 * RADIUS + 2) * (2 * RADIUS + 2) + 63) / 64 : 4;
  if (!RADIUS && fp * fp > 64 * NI) return;                       // (the launcher bounds the run-time radius)
  float gv[4][NI][4], old[4][NI], wgt[4][NI][4];
  int offs[4][NI];
  unsigned okm = 0u;
  int x0s[4], y0s[4];
  float fxs[4], fys[4];
  {
    float sc = 1.f;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const float X = cx * sc, Y = cy * sc;
      const float x0f = floorf(X), y0f = floorf(Y);
      fxs[l] = X - x0f; fys[l] = Y - y0f;
      x0s[l] = (int)fminf(fmaxf(x0f, -100000.f), 100000.f) - radius;
      y0s[l] = (int)fminf(fmaxf(y0f, -100000.f), 100000.f) - radius;
      sc *= 0.5f;
    }
  }
  asm volatile("" : "+v"(x0s[0]), "+v"(x0s[1]), "+v"(x0s[2]), "+v"(x0s[3]), "+v"(y0s[0]), "+v"(y0s[1]), "+v"(y0s[2]), "+v"(y0s[3]));
  int h = H8, w = W8;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    if (l < levels) {
      const float* img = Gl[l] + q * (long)h * w;
      const float* g = dout + q * ldo + l * lvl_stride + col_off;      // g[a * win + bb]: x offset a, y offset bb
      const float wx[2] = {1.f - fxs[l], fxs[l]}, wy[2] = {1.f - fys[l], fys[l]};
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int c = lane + 64 * j;
        const int v = c / fp, u = c - v * fp;             // footprint cell: x = x0 + u, y = y0 + v (u fastest: coalesced rows)
        const int x = x0s[l] + u, y = y0s[l] + v;
        const bool ok = c < fp * fp && x >= 0 && x < w && y >= 0 && y < h;
        okm |= ok ? 1u << (l * NI + j) : 0u;
        offs[l][j] = min(max(y, 0), h - 1) * w + min(max(x, 0), w - 1);
        old[l][j] = img[offs[l][j]];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int a = u - (t & 1), bb = v - (t >> 1);
          const bool in = a >= 0 && a < win && bb >= 0 && bb < win;
          wgt[l][j][t] = in ? wx[t & 1] * wy[t >> 1] : 0.f;
          gv[l][j][t] = g[min(max(a, 0), win - 1) * win + min(max(bb, 0), win - 1)];
        }
      }
      h >>= 1; w >>= 1;
    }
  }
  h = H8; w = W8;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    if (l < levels) {
      float* img = Gl[l] + q * (long)h * w;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc += wgt[l][j][t] * gv[l][j][t];
        if ((okm >> (l * NI + j)) & 1u) img[offs[l][j]] = old[l][j] + acc;
      }
      h >>= 1; w >>= 1;
    }
  }
}
int launch_corr_lookup_bwd(const float* dout, long ldo, const float* coords, float* G0, float* G1, float* G2, float* G3, int levels, int B,
                           int H8, int W8, int radius, int lvl_stride, int col_off, hipStream_t s) {
  if (levels < 1 || levels > 4 || radius < 0 || 2 * radius + 2 > 16) return CRAFT_ERR_UNSUPPORTED;
  const int win2 = (2 * radius + 1) * (2 * radius + 1);
  if (lvl_stride <= 0) lvl_stride = win2;
  const long nq = (long)B * H8 * W8;
  if (nq <= 0) return 0;
  if ((H8 >> (levels - 1)) < 1 || (W8 >> (levels - 1)) < 1) return CRAFT_ERR_ARG;      // (an empty level: clamped addresses would be offset -1)
  if (radius == 4)
    hipLaunchKernelGGL(k_corr_lookup_bwd<4>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, dout, ldo, coords, G0, G1, G2, G3, levels, H8, W8,
                       radius, lvl_stride, col_off, nq);
  else
    hipLaunchKernelGGL(k_corr_lookup_bwd<0>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, dout, ldo, coords, G0, G1, G2, G3, levels, H8, W8,
                       radius, lvl_stride, col_off, nq);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// backward of craft_mode_pool_ln (ExpandedFeatTrans tail, setrans.py:395-407).  One wave per token, C <= 256.
// dw_agg / dskip: per-block partials -> replicated tables [CRAFT_STATS_REPLICAS][C + 1] (last cell: dskip).
// ---------------------------------------------------------------------------------------------
constexpr int MPL_TOK_PER_WAVE = 8;
// M (modes) and C (128 / 256 channels) are template parameters: with run-time values every load sat behind `c < C` / `m < M` tests (its own
// round trip each, 12 per token) and half of the 4 channel slots per lane were idle work at C = 128 (1.7 TB/s).  Here a token's loads
// are straight-line code: 2 * M + 4 (C = 128) requests in flight, then the arithmetic.
template <int M, int C>
__global__ __launch_bounds__(256) void k_mode_pool_ln_bwd(const float* __restrict__ O, const float* __restrict__ x, long ldx,
                                                          const float* __restrict__ w_agg, const float* __restrict__ skip_coeff,
                                                          const float* __restrict__ dy, long lddy, int N, long ntok,
                                                          float* __restrict__ dO, float* __restrict__ dx, long lddx,
                                                          float* __restrict__ dw_rep) {
  constexpr int NI = C / 64;                       // channel slots per lane: c = lane + 64 i
  __shared__ float acc_w[4][C + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float wl[NI], dwl[NI];
  float dskip = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) { wl[i] = w_agg[lane + 64 * i]; dwl[i] = 0.f; }
  const float skip = *skip_coeff;
  const long t0 = ((long)blockIdx.x * 4 + wv) * MPL_TOK_PER_WAVE;
  for (long tok = t0; tok < min(ntok, t0 + MPL_TOK_PER_WAVE); ++tok) {
    const long b = tok / N, n = tok - b * N;
    float o[M][NI], t[M], xv[NI], g[NI];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const float* Om = O + ((b * M + m) * N + n) * C;
#pragma unroll
      for (int i = 0; i < NI; ++i) o[m][i] = Om[lane + 64 * i];
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) { xv[i] = x[tok * ldx + lane + 64 * i]; g[i] = dy[tok * lddy + lane + 64 * i]; }
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) s += o[m][i] * wl[i];
      t[m] = wave_sum(s);
    }
    float mx = t[0];
#pragma unroll
    for (int m = 1; m < M; ++m) mx = fmaxf(mx, t[m]);
    float a[M], den = 0.f;
#pragma unroll
    for (int m = 0; m < M; ++m) { a[m] = expf(t[m] - mx); den += a[m]; }
#pragma unroll
    for (int m = 0; m < M; ++m) a[m] /= den;
    float u[NI], s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float p = 0.f;
#pragma unroll
      for (int m = 0; m < M; ++m) p += a[m] * o[m][i];
      u[i] = skip * xv[i] + p;
      s1 += u[i];
    }
    const float mean = wave_sum(s1) / C;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) { const float d = u[i] - mean; s2 += d * d; }
    const float rstd = rsqrtf(wave_sum(s2) / C + CRAFT_LN_EPS);
    float sg = 0.f, sgy = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) { sg += g[i]; sgy += g[i] * (u[i] - mean) * rstd; }
    const float mg = wave_sum(sg) / C, mgy = wave_sum(sgy) / C;
    float du[NI], sx = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      du[i] = rstd * (g[i] - mg - (u[i] - mean) * rstd * mgy);
      sx += du[i] * xv[i];
      dx[tok * lddx + lane + 64 * i] = skip * du[i];
    }
    dskip += sx;                                 // (lane partial; reduced at the end)
    float da[M], dsum = 0.f;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) s += du[i] * o[m][i];
      da[m] = wave_sum(s);
      dsum += a[m] * da[m];
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const float dt = a[m] * (da[m] - dsum);
      float* dOm = dO + ((b * M + m) * N + n) * C;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        dOm[lane + 64 * i] = a[m] * du[i] + dt * wl[i];
        dwl[i] += dt * o[m][i];
      }
    }
  }
  dskip = wave_sum(dskip);
#pragma unroll
  for (int i = 0; i < NI; ++i) acc_w[wv][lane + 64 * i] = dwl[i];
  if (lane == 0) acc_w[wv][C] = dskip;
  __syncthreads();
  float* rep = dw_rep + (long)(blockIdx.x % CRAFT_STATS_REPLICAS) * (C + 1);
  for (int c = threadIdx.x; c < C + 1; c += 256) unsafeAtomicAdd(rep + c, acc_w[0][c] + acc_w[1][c] + acc_w[2][c] + acc_w[3][c]);
}
int launch_mode_pool_ln_bwd(const float* O, const float* x, long ldx, const float* w_agg, const float* skip_coeff, const float* dy, long lddy,
                            int B, int N, int M, int C, float* dO, float* dx, long lddx, float* dw_rep, hipStream_t s) {
  if (B <= 0 || N <= 0) return 0;
  if ((C != 128 && C != 256) || M > 4 || M < 1) return CRAFT_ERR_UNSUPPORTED;
  const long ntok = (long)B * N;
  const dim3 grid((unsigned)((ntok + 4 * MPL_TOK_PER_WAVE - 1) / (4 * MPL_TOK_PER_WAVE)));
#define GO(MM, CC) hipLaunchKernelGGL((k_mode_pool_ln_bwd<MM, CC>), grid, dim3(256), 0, s, O, x, ldx, w_agg, skip_coeff, dy, lddy, N, ntok, dO, dx, lddx, dw_rep)
#define GOC(MM) do { if (C == 128) GO(MM, 128); else GO(MM, 256); } while (0)
  switch (M) { case 1: GOC(1); break; case 2: GOC(2); break; case 3: GOC(3); break; default: GOC(4); break; }
#undef GOC
#undef GO
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// backward of craft_convex_upsample (network.py:151-162).  One wave per low-resolution pixel, lane = sub-position (i, j).
// dmask [B*N][576] written; dflow [B*N][2] accumulated with atomics (zero it first).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_convex_upsample_bwd(const float* __restrict__ mask, long ldm, const float* __restrict__ flow,
                                                             const float* __restrict__ dup, int H8, int W8, long npix,
                                                             float* __restrict__ dmask, long lddm, float* __restrict__ dflow, long lddf) {
  const int lane = threadIdx.x & 63;
  const long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= npix) return;
  const int N = H8 * W8;
  const long b = p / N;
  const int n = (int)(p - b * N), y = n / W8, x = n - y * W8;
  const int si = lane >> 3, sj = lane & 7;
  const int H = 8 * H8, W = 8 * W8;
  const float g0 = dup[((b * 2 + 0) * H + 8 * y + si) * W + 8 * x + sj];
  const float g1 = dup[((b * 2 + 1) * H + 8 * y + si) * W + 8 * x + sj];
  float mk[9], mx = -3.0e38f;
#pragma unroll
  for (int k = 0; k < 9; ++k) { mk[k] = mask[p * ldm + k * 64 + lane]; mx = fmaxf(mx, mk[k]); }
  float den = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) { mk[k] = expf(mk[k] - mx); den += mk[k]; }
  float dm[9], dsum = 0.f;
  // (the 9 neighbour flows requested together from clamped addresses; `ok ? load : 0` compiled to a load under a branch each)
  float2 fl[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = min(max(y + k / 3 - 1, 0), H8 - 1), xx = min(max(x + k % 3 - 1, 0), W8 - 1);
    fl[k] = *reinterpret_cast<const float2*>(flow + 2 * (b * N + yy * W8 + xx));
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    mk[k] /= den;
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    const bool ok = yy >= 0 && yy < H8 && xx >= 0 && xx < W8;
    const long nb = b * N + (ok ? yy * W8 + xx : 0);
    const float f0 = ok ? 8.f * fl[k].x : 0.f, f1 = ok ? 8.f * fl[k].y : 0.f;
    dm[k] = g0 * f0 + g1 * f1;
    dsum += mk[k] * dm[k];
    const float d0 = wave_sum(8.f * mk[k] * g0), d1 = wave_sum(8.f * mk[k] * g1);
    if (lane == 0 && ok) { unsafeAtomicAdd(dflow + lddf * nb, d0); unsafeAtomicAdd(dflow + lddf * nb + 1, d1); }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) dmask[p * lddm + k * 64 + lane] = mk[k] * (dm[k] - dsum);
}
int launch_convex_upsample_bwd(const float* mask, long ldm, const float* flow, const float* dup, int B, int H8, int W8, float* dmask,
                               long lddm, float* dflow, long lddf, hipStream_t s) {
  const long npix = (long)B * H8 * W8;
  if (npix <= 0) return 0;
  if (lddf < 2) return CRAFT_ERR_ARG;
  hipLaunchKernelGGL(k_convex_upsample_bwd, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, s, mask, ldm, flow, dup, H8, W8, npix, dmask,
                     lddm, dflow, lddf);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The per-iteration coordinate bookkeeping of the training forward in one launch (network.py:232-234, :247): flow = coords1 - coords0
// as [rows][2] (the motion encoder's input) and zero-padded to [rows][32] (the 7x7 convolution's weight-gradient operand), and a copy
// of coords1 for the flow head to update in place.  One thread per row.
// ---------------------------------------------------------------------------------------------
__global__ void k_flow_tokens(const float* __restrict__ c1, const float* __restrict__ c0, long rows, float* __restrict__ flow,
                              float* __restrict__ flow32, float* __restrict__ c1copy) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const float2 a = *reinterpret_cast<const float2*>(c1 + 2 * i), b = *reinterpret_cast<const float2*>(c0 + 2 * i);
  const float2 f = make_float2(a.x - b.x, a.y - b.y);
  *reinterpret_cast<float2*>(flow + 2 * i) = f;
  if (c1copy) *reinterpret_cast<float2*>(c1copy + 2 * i) = a;
  if (flow32) {
    float4* d = reinterpret_cast<float4*>(flow32 + 32 * i);
    d[0] = make_float4(f.x, f.y, 0.f, 0.f);
#pragma unroll
    for (int k = 1; k < 8; ++k) d[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
int launch_flow_tokens(const float* c1, const float* c0, long rows, float* flow, float* flow32, float* c1copy, hipStream_t s) {
  if (rows <= 0) return 0;
  if (!c1 || !c0 || !flow) return CRAFT_ERR_ARG;
  hipLaunchKernelGGL(k_flow_tokens, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, c1, c0, rows, flow, flow32, c1copy);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// SepConvGRU gates as separate stages (update.py:49-64), C = 128 hidden channels, float4 streams.
//   zr stage : z = sigmoid(zr_pre[:, :C]), r = sigmoid(zr_pre[:, C:]), rh = r * h
//   out stage: q = tanh(q_pre), h' = (1 - z) h + z q
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
#define F4_MAP1(out, a, expr) { float A; A = a.x; out.x = (expr); A = a.y; out.y = (expr); A = a.z; out.z = (expr); A = a.w; out.w = (expr); }

__global__ void k_gru_zr_fwd(const float* __restrict__ zr, long ldzr, const float* __restrict__ h, long ldh, float* __restrict__ z,
                             float* __restrict__ r, float* __restrict__ rh, long rows, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C >> 2;
  if (i >= rows * c4) return;
  const long row = i / c4;
  const int c = (int)(i - row * c4) * 4;
  const float4 zp = ld4(zr + row * ldzr + c), rp = ld4(zr + row * ldzr + C + c), hv = ld4(h + row * ldh + c);
  float4 zo, ro, rho;
  F4_MAP1(zo, zp, sigmoid_precise(A));
  F4_MAP1(ro, rp, sigmoid_precise(A));
  rho.x = ro.x * hv.x; rho.y = ro.y * hv.y; rho.z = ro.z * hv.z; rho.w = ro.w * hv.w;
  st4(z + row * C + c, zo); st4(r + row * C + c, ro); st4(rh + row * C + c, rho);
}
__global__ void k_gru_out_fwd(const float* __restrict__ qp, long ldq, const float* __restrict__ z, const float* __restrict__ h, long ldh,
                              float* __restrict__ q, float* __restrict__ hn, long ldhn, long rows, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C >> 2;
  if (i >= rows * c4) return;
  const long row = i / c4;
  const int c = (int)(i - row * c4) * 4;
  const float4 qv = ld4(qp + row * ldq + c), zv = ld4(z + row * C + c), hv = ld4(h + row * ldh + c);
  float4 qo, ho;
  F4_MAP1(qo, qv, tanhf(A));
  ho.x = (1.f - zv.x) * hv.x + zv.x * qo.x; ho.y = (1.f - zv.y) * hv.y + zv.y * qo.y;
  ho.z = (1.f - zv.z) * hv.z + zv.z * qo.z; ho.w = (1.f - zv.w) * hv.w + zv.w * qo.w;
  st4(q + row * C + c, qo); st4(hn + row * ldhn + c, ho);
}
// dh' -> dq_pre = dh' z (1 - q^2), dz = dh' (q - h), dh = dh' (1 - z)
__global__ void k_gru_out_bwd(const float* __restrict__ dhn, long lddhn, const float* __restrict__ z, const float* __restrict__ q,
                              const float* __restrict__ h, long ldh, float* __restrict__ dqp, float* __restrict__ dz, float* __restrict__ dh,
                              long rows, int C, float* __restrict__ dqp_sum) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C >> 2;
  if (i >= rows * c4) return;
  const long row = i / c4;
  const int c = (int)(i - row * c4) * 4;
  const float4 g = ld4(dhn + row * lddhn + c), zv = ld4(z + row * C + c), qv = ld4(q + row * C + c), hv = ld4(h + row * ldh + c);
  float4 a, b2, d;
  a.x = g.x * zv.x * (1.f - qv.x * qv.x); a.y = g.y * zv.y * (1.f - qv.y * qv.y); a.z = g.z * zv.z * (1.f - qv.z * qv.z); a.w = g.w * zv.w * (1.f - qv.w * qv.w);
  b2.x = g.x * (qv.x - hv.x); b2.y = g.y * (qv.y - hv.y); b2.z = g.z * (qv.z - hv.z); b2.w = g.w * (qv.w - hv.w);
  d.x = g.x * (1.f - zv.x); d.y = g.y * (1.f - zv.y); d.z = g.z * (1.f - zv.z); d.w = g.w * (1.f - zv.w);
  st4(dqp + row * C + c, a); st4(dz + row * C + c, b2); st4(dh + row * C + c, d);
  if (dqp_sum) {                                     // running sum over the refinement iterations (the hoisted context's gradients)
    float4 t = ld4(dqp_sum + row * C + c);
    t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
    st4(dqp_sum + row * C + c, t);
  }
}
// (dz, d(rh)) -> dzr_pre = [dz z (1-z) | d(rh) h r (1-r)], dh += d(rh) r
__global__ void k_gru_zr_bwd(const float* __restrict__ dz, const float* __restrict__ drh, long lddrh, const float* __restrict__ z,
                             const float* __restrict__ r, const float* __restrict__ h, long ldh, float* __restrict__ dzr,
                             const float* __restrict__ dh, long rows, int C, float* __restrict__ dzr_sum, float* dh_out, long lddho) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C >> 2;
  if (i >= rows * c4) return;
  const long row = i / c4;
  const int c = (int)(i - row * c4) * 4;
  const float4 gz = ld4(dz + row * C + c), gr = ld4(drh + row * lddrh + c), zv = ld4(z + row * C + c), rv = ld4(r + row * C + c),
               hv = ld4(h + row * ldh + c);
  float4 a, b2, d = ld4(dh + row * C + c);
  a.x = gz.x * zv.x * (1.f - zv.x); a.y = gz.y * zv.y * (1.f - zv.y); a.z = gz.z * zv.z * (1.f - zv.z); a.w = gz.w * zv.w * (1.f - zv.w);
  b2.x = gr.x * hv.x * rv.x * (1.f - rv.x); b2.y = gr.y * hv.y * rv.y * (1.f - rv.y); b2.z = gr.z * hv.z * rv.z * (1.f - rv.z); b2.w = gr.w * hv.w * rv.w * (1.f - rv.w);
  d.x += gr.x * rv.x; d.y += gr.y * rv.y; d.z += gr.z * rv.z; d.w += gr.w * rv.w;
  st4(dzr + row * 2 * C + c, a); st4(dzr + row * 2 * C + C + c, b2);
  st4(dh_out + row * lddho + c, d);                 // (dh_out may be drh itself: each element is read before it is written, by this thread)
  if (dzr_sum) {
    float4 t = ld4(dzr_sum + row * 2 * C + c), u = ld4(dzr_sum + row * 2 * C + C + c);
    t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w; u.x += b2.x; u.y += b2.y; u.z += b2.z; u.w += b2.w;
    st4(dzr_sum + row * 2 * C + c, t); st4(dzr_sum + row * 2 * C + C + c, u);
  }
}
#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, s
int launch_gru_zr_fwd(const float* zr, long ldzr, const float* h, long ldh, float* z, float* r, float* rh, long rows, int C, hipStream_t s) {
  if (rows <= 0) return 0;
  if ((C & 3) || (ldzr & 3) || (ldh & 3)) return CRAFT_ERR_ALIGN;
  hipLaunchKernelGGL(k_gru_zr_fwd, GRID1(rows * (C >> 2)), zr, ldzr, h, ldh, z, r, rh, rows, C);
  return (int)hipGetLastError();
}
int launch_gru_out_fwd(const float* qp, long ldq, const float* z, const float* h, long ldh, float* q, float* hn, long ldhn, long rows, int C,
                       hipStream_t s) {
  if (rows <= 0) return 0;
  if ((C & 3) || (ldq & 3) || (ldh & 3) || (ldhn & 3)) return CRAFT_ERR_ALIGN;
  hipLaunchKernelGGL(k_gru_out_fwd, GRID1(rows * (C >> 2)), qp, ldq, z, h, ldh, q, hn, ldhn, rows, C);
  return (int)hipGetLastError();
}
int launch_gru_out_bwd(const float* dhn, long lddhn, const float* z, const float* q, const float* h, long ldh, float* dqp, float* dz, float* dh,
                       long rows, int C, float* dqp_sum, hipStream_t s) {
  if (rows <= 0) return 0;
  if ((C & 3) || (lddhn & 3) || (ldh & 3)) return CRAFT_ERR_ALIGN;
  hipLaunchKernelGGL(k_gru_out_bwd, GRID1(rows * (C >> 2)), dhn, lddhn, z, q, h, ldh, dqp, dz, dh, rows, C, dqp_sum);
  return (int)hipGetLastError();
}
int launch_gru_zr_bwd(const float* dz, const float* drh, long lddrh, const float* z, const float* r, const float* h, long ldh, float* dzr,
                      float* dh, long rows, int C, float* dzr_sum, float* dh_out, long lddho, hipStream_t s) {
  if (rows <= 0) return 0;
  if (dh_out == nullptr) { dh_out = dh; lddho = C; }
  if ((C & 3) || (lddrh & 3) || (ldh & 3) || (lddho & 3)) return CRAFT_ERR_ALIGN;
  hipLaunchKernelGGL(k_gru_zr_bwd, GRID1(rows * (C >> 2)), dz, drh, lddrh, z, r, h, ldh, dzr, dh, rows, C, dzr_sum, dh_out, lddho);
  return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// RelPosEmb scores of gma.Attention (gma.py:21-50, :84-98) on MATERIALISED scores: query i = (x, y) (x = row), key j = (u, v):
//   S[z][i][j] += w * (Hs[z][i][u - x + H8 - 1] + Ws[z][i][v - y + W8 - 1])
// Hs / Ws = (scaled) q . E_h / E_w rows (two small GEMMs of the caller), row strides ldh / ldw.  One block per query row.
// backward: dHs[z][i][d] = w * sum_v dS[i][(u, v)], d = u - x + H8 - 1 (unreachable offsets: 0); dWs likewise over u.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_relpos_add(float* __restrict__ S, long ld, int N, int H8, int W8, const float* __restrict__ Hs, long ldh,
                                                    const float* __restrict__ Ws, long ldw, float w) {
  const long row = blockIdx.x;                       // z * N + i
  const int i = (int)(row % N), x = i / W8, y = i - x * W8;
  float* s = S + row * ld;
  const float* hs = Hs + row * ldh + (H8 - 1 - x);
  const float* ws = Ws + row * ldw + (W8 - 1 - y);
  for (int j = threadIdx.x; j < N; j += 256) {
    const int u = j / W8, v = j - u * W8;
    s[j] += w * (hs[u] + ws[v]);
  }
}
__global__ __launch_bounds__(256) void k_relpos_bwd(const float* __restrict__ dS, long ld, int N, int H8, int W8, float* __restrict__ dHs, long ldh,
                                                    int nh, float* __restrict__ dWs, long ldw, int nw, float w) {
  extern __shared__ float sm[];                      // [H8] row sums | [W8] column sums
  const long row = blockIdx.x;
  const int i = (int)(row % N), x = i / W8, y = i - x * W8;
  const float* g = dS + row * ld;
  for (int t = threadIdx.x; t < H8 + W8; t += 256) sm[t] = 0.f;
  __syncthreads();
  for (int j = threadIdx.x; j < N; j += 256) {
    const int u = j / W8, v = j - u * W8;
    const float d = g[j];
    atomicAdd(&sm[u], d);
    atomicAdd(&sm[H8 + v], d);
  }
  __syncthreads();
  float* dh = dHs + row * ldh;
  float* dw = dWs + row * ldw;
  for (int t = threadIdx.x; t < nh; t += 256) { const int u = t - (H8 - 1 - x); dh[t] = (u >= 0 && u < H8) ? w * sm[u] : 0.f; }
  for (int t = threadIdx.x; t < nw; t += 256) { const int v = t - (W8 - 1 - y); dw[t] = (v >= 0 && v < W8) ? w * sm[H8 + v] : 0.f; }
}
int launch_relpos_add(float* S, long ld, int BZ, int H8, int W8, const float* Hs, long ldh, const float* Ws, long ldw, float w, hipStream_t s) {
  const int N = H8 * W8;
  if (BZ <= 0 || N <= 0) return 0;
  hipLaunchKernelGGL(k_relpos_add, dim3((unsigned)((long)BZ * N)), dim3(256), 0, s, S, ld, N, H8, W8, Hs, ldh, Ws, ldw, w);
  return (int)hipGetLastError();
}
int launch_relpos_bwd(const float* dS, long ld, int BZ, int H8, int W8, float* dHs, long ldh, int nh, float* dWs, long ldw, int nw, float w,
                      hipStream_t s) {
  const int N = H8 * W8;
  if (BZ <= 0 || N <= 0) return 0;
  if (nh < 2 * H8 - 1 || nw < 2 * W8 - 1) return CRAFT_ERR_ARG;
  hipLaunchKernelGGL(k_relpos_bwd, dim3((unsigned)((long)BZ * N)), dim3(256), (H8 + W8) * sizeof(float), s, dS, ld, N, H8, W8, dHs, ldh, nh, dWs, ldw,
                     nw, w);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// craft_multi_copy: n contiguous fp32 tensors -> their places in one flat buffer, ONE launch (the parameter gradients of a step into the
// optimizer's flat gradient buffer: torch._foreach_copy_ on this build issues one copyBuffer per tensor, 131 per configs[3] step).
// The descriptor table travels in the kernel arguments; block -> (tensor, 8 KiB chunk) through a prefix table.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int MC_MAX = 160;                       // tensors per launch (the argument block stays under 4 KiB)
constexpr int MC_CHUNK = 2048;                    // floats per block
// cl[i] != 0: the source is a conv-weight gradient in the layout the weight-gradient kernels write, [cout][taps][cin] (cl = cin * 1024 + taps),
// and lands as nn.Conv2d's [cout][cin][taps]
struct MultiCopy { const float* src[MC_MAX]; unsigned off[MC_MAX]; unsigned n[MC_MAX]; unsigned cl[MC_MAX]; unsigned first[MC_MAX + 1]; int count; };
__global__ __launch_bounds__(256) void k_multi_copy(MultiCopy m, float* __restrict__ dst) {
  int lo = 0, hi = m.count;                       // the tensor whose chunks contain this block
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (blockIdx.x >= m.first[mid]) lo = mid; else hi = mid; }
  const unsigned c0 = (blockIdx.x - m.first[lo]) * MC_CHUNK, n = m.n[lo];
  const unsigned len = min((unsigned)MC_CHUNK, n - c0);
  float* d = dst + m.off[lo] + c0;
  if (m.cl[lo]) {
    const unsigned cin = m.cl[lo] >> 10, T = m.cl[lo] & 1023u, ct = cin * T;
    const float* s = m.src[lo];
    for (unsigned i = threadIdx.x; i < len; i += 256) {
      const unsigned j = c0 + i, co = j / ct, rem = j - co * ct, ci = rem / T, t = rem - ci * T;
      d[i] = s[(co * T + t) * cin + ci];
    }
    return;
  }
  const float* s = m.src[lo] + c0;
  if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
    for (unsigned i = threadIdx.x * 4; i + 3 < len; i += 1024) *reinterpret_cast<float4*>(d + i) = *reinterpret_cast<const float4*>(s + i);
    for (unsigned i = (len & ~3u) + threadIdx.x; i < len; i += 256) d[i] = s[i];
  } else {
    for (unsigned i = threadIdx.x; i < len; i += 256) d[i] = s[i];
  }
}
int launch_multi_copy(const void* const* src, const long* n, const long* dst_off, const long* chlast, int count, float* dst, hipStream_t s) {
  for (int i0 = 0; i0 < count; i0 += MC_MAX) {
    MultiCopy m = {};
    m.count = count - i0 < MC_MAX ? count - i0 : MC_MAX;
    long blocks = 0;
    for (int i = 0; i < m.count; ++i) {
      const long ni = n[i0 + i], oi = dst_off[i0 + i], cl = chlast ? chlast[i0 + i] : 0;
      if (ni < 0 || oi < 0 || ni >= (1L << 32) || oi >= (1L << 32) || (ni > 0 && src[i0 + i] == nullptr)) return CRAFT_ERR_ARG;
      if (cl < 0 || cl >= (1L << 32) || (cl && ((cl & 1023) == 0 || (cl >> 10) == 0 || ni % ((cl >> 10) * (cl & 1023)) != 0))) return CRAFT_ERR_ARG;
      m.src[i] = static_cast<const float*>(src[i0 + i]); m.n[i] = (unsigned)ni; m.off[i] = (unsigned)oi; m.cl[i] = (unsigned)cl;
      m.first[i] = (unsigned)blocks;
      blocks += (ni + MC_CHUNK - 1) / MC_CHUNK;
    }
    if (blocks >= (1L << 31)) return CRAFT_ERR_UNSUPPORTED;
    m.first[m.count] = (unsigned)blocks;
    if (blocks > 0) hipLaunchKernelGGL(k_multi_copy, dim3((unsigned)blocks), dim3(256), 0, s, m, dst);
    const hipError_t e = hipGetLastError();          // (read ONCE: the call clears the sticky error)
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

}  // namespace craft
