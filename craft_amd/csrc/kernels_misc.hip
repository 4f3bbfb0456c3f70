// HBM-bound kernels of the hot path: token LayerNorm, mode pooling + skip + LayerNorm, pyramid pooling,
// correlation lookup (radius-r bilinear gather), the two tiny convolutions that are not worth a GEMM,
// convex upsampling and coordinate bookkeeping.  All activations are channels-last ("tokens": [B, N, C]).
#include <atomic>
#include "launch.hpp"

namespace craft {

// ---------------------------------------------------------------------------------------------
// tokens: (NCHW | tokens) -> activation -> optional LayerNorm over C -> tokens  (K1; network.py:209-212)
// block = 256 threads handles 32 pixels x C channels (C <= 256) through a padded LDS tile.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tokens(const float* __restrict__ src, int src_nchw, int Ctot, int c_off, int C,
                                                int HW, long src_ld, int act, int do_ln, float* __restrict__ dst, long dst_ld) {
  __shared__ float tile[256 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, p0 = blockIdx.x * 32;
  if (src_nchw) {
    const int px = tid & 31;
    const bool ok = p0 + px < HW;
    for (int c = tid >> 5; c < C; c += 8) {
      float v = ok ? src[((long)b * Ctot + c_off + c) * HW + p0 + px] : 0.f;
      tile[c * 33 + px] = act_apply(v, act);
    }
  } else {
    for (int idx = tid; idx < 32 * C; idx += 256) {
      const int px = idx / C, c = idx - px * C;
      float v = (p0 + px < HW) ? src[((long)b * HW + p0 + px) * src_ld + c_off + c] : 0.f;
      tile[c * 33 + px] = act_apply(v, act);
    }
  }
  __syncthreads();
  for (int px = wave; px < 32; px += 4) {
    if (p0 + px >= HW) break;
    float v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      v[i] = (c < C) ? tile[c * 33 + px] : 0.f;
      s += v[i];
    }
    float* o = dst + ((long)b * HW + p0 + px) * dst_ld;
    if (do_ln) {
      const float mean = wave_sum(s) / (float)C;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        const float d = (c < C) ? v[i] - mean : 0.f;
        q += d * d;
      }
      const float rstd = 1.f / sqrtf(wave_sum(q) / (float)C + CRAFT_LN_EPS);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < C) o[c] = (v[i] - mean) * rstd;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < C) o[c] = v[i];
      }
    }
  }
}

int launch_tokens(const float* src, int src_nchw, int B, int Ctot, int c_off, int C, int HW, long src_ld, int act,
                  int do_ln, float* dst, long dst_ld, hipStream_t s) {
  if (C > 256 || C < 1) return CRAFT_ERR_UNSUPPORTED;
  dim3 grid((HW + 31) / 32, B);
  hipLaunchKernelGGL(k_tokens, grid, dim3(256), 0, s, src, src_nchw, Ctot, c_off, C, HW, src_ld, act, do_ln, dst, dst_ld);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// ExpandedFeatTrans tail (setrans.py:395-407): a_m = softmax_m(<O_m, w>), y = sum_m a_m O_m,
// out = LN(c_skip * x + y).  One wave per token; C <= 256, M <= 8.
// ---------------------------------------------------------------------------------------------
// C / 4 lanes per token (float4 per lane and mode, everything in registers), 64 / (C/4) tokens per wave.
template <int M, int C>
__global__ __launch_bounds__(256) void k_mode_pool_ln(const float* __restrict__ O, const float* __restrict__ x, long ldx,
                                                      const float* __restrict__ w_agg, const float* __restrict__ skip_coeff,
                                                      int N, float* __restrict__ out, long ldo, long ntok) {
  constexpr int LPT = C / 4, TPB = 256 / LPT;     // lanes per token, tokens per block
  const int sub = threadIdx.x % LPT;
  const long tok = (long)blockIdx.x * TPB + threadIdx.x / LPT;
  const bool live = tok < ntok;
  const long tk = live ? tok : ntok - 1;          // clamped: all lanes take part in the shuffles
  const int b = (int)(tk / N), n = (int)(tk - (long)b * N);
  auto group_sum = [&](float v) __attribute__((always_inline)) {
#pragma unroll
    for (int o = LPT / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  const float4 w = *reinterpret_cast<const float4*>(w_agg + sub * 4);
  float4 o[M];
  float sc[M];
#pragma unroll
  for (int m = 0; m < M; ++m) o[m] = *reinterpret_cast<const float4*>(O + (((long)b * M + m) * N + n) * C + sub * 4);
  const float4 xv = *reinterpret_cast<const float4*>(x + tk * ldx + sub * 4);
#pragma unroll
  for (int m = 0; m < M; ++m) sc[m] = group_sum(o[m].x * w.x + o[m].y * w.y + o[m].z * w.z + o[m].w * w.w);
  float mxs = sc[0];
#pragma unroll
  for (int m = 1; m < M; ++m) mxs = fmaxf(mxs, sc[m]);
  float den = 0.f;
#pragma unroll
  for (int m = 0; m < M; ++m) { sc[m] = expf(sc[m] - mxs); den += sc[m]; }
  const float cs = skip_coeff[0];
  float4 t = make_float4(cs * xv.x, cs * xv.y, cs * xv.z, cs * xv.w);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const float a = sc[m] / den;
    t.x += o[m].x * a; t.y += o[m].y * a; t.z += o[m].z * a; t.w += o[m].w * a;
  }
  const float mean = group_sum(t.x + t.y + t.z + t.w) / (float)C;
  const float dx = t.x - mean, dy = t.y - mean, dz = t.z - mean, dw = t.w - mean;
  const float rstd = 1.f / sqrtf(group_sum(dx * dx + dy * dy + dz * dz + dw * dw) / (float)C + CRAFT_LN_EPS);
  if (live) *reinterpret_cast<float4*>(out + tok * ldo + sub * 4) = make_float4(dx * rstd, dy * rstd, dz * rstd, dw * rstd);
}

template <int M, int C> static void launch_mpl_t(const float* O, const float* x, long ldx, const float* w_agg, const float* skip_coeff,
                                                  int N, float* out, long ldo, long ntok, hipStream_t s) {
  constexpr int TPB = 256 / (C / 4);
  hipLaunchKernelGGL((k_mode_pool_ln<M, C>), dim3((unsigned)((ntok + TPB - 1) / TPB)), dim3(256), 0, s, O, x, ldx, w_agg, skip_coeff, N,
                     out, ldo, ntok);
}

int launch_mode_pool_ln(const float* O, const float* x, long ldx, const float* w_agg, const float* skip_coeff, int B, int N,
                        int M, int C, float* out, long ldo, hipStream_t s) {
  if ((C != 128 && C != 256) || M < 1 || M > 8 || (ldx & 3) || (ldo & 3)) return CRAFT_ERR_UNSUPPORTED;
  const long ntok = (long)B * N;
#define GO(MM) do { if (C == 128) launch_mpl_t<MM, 128>(O, x, ldx, w_agg, skip_coeff, N, out, ldo, ntok, s); \
                    else launch_mpl_t<MM, 256>(O, x, ldx, w_agg, skip_coeff, N, out, ldo, ntok, s); } while (0)
  switch (M) {
    case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break;
    case 5: GO(5); break; case 6: GO(6); break; case 7: GO(7); break; default: GO(8); break;
  }
#undef GO
  return (int)hipGetLastError();
}

// gma.Aggregate tail (gma.py:138): out = fmap + gamma * O
__global__ void k_gma_residual(const float* __restrict__ mf, long ldm, const float* __restrict__ O,
                               const float* __restrict__ gamma, long ntok, int C, float* __restrict__ out, long ldo) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;      // 4 channels per thread
  if (i >= ntok * C) return;
  const long tok = i / C;
  const int c = (int)(i - tok * C);
  const float g = gamma[0];
  const float4 a = *reinterpret_cast<const float4*>(mf + tok * ldm + c), o = *reinterpret_cast<const float4*>(O + i);
  *reinterpret_cast<float4*>(out + tok * ldo + c) = make_float4(a.x + g * o.x, a.y + g * o.y, a.z + g * o.z, a.w + g * o.w);
}
int launch_gma_residual(const float* mf, long ldm, const float* O, const float* gamma, int B, int N, int C, float* out,
                        long ldo, hipStream_t s) {
  if ((C & 3) || (ldm & 3) || (ldo & 3)) return CRAFT_ERR_ALIGN;
  const long tot = (long)B * N * C / 4;
  hipLaunchKernelGGL(k_gma_residual, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, mf, ldm, O, gamma, (long)B * N, C,
                     out, ldo);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// correlation pyramid: levels 1..3 by repeated 2x2 average pooling (corr.py:186-189); one block per
// (sample, query) image.  Level l+1 is computed from level l held in LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_corr_pyramid(const float* __restrict__ l0, float* __restrict__ l1,
                                                      float* __restrict__ l2, float* __restrict__ l3, int H8, int W8) {
  extern __shared__ float sm[];
  const long img = blockIdx.x;
  const int h1 = H8 / 2, w1 = W8 / 2, h2 = h1 / 2, w2 = w1 / 2, h3 = h2 / 2, w3 = w2 / 2;
  float* s1 = sm;
  float* s2 = sm + h1 * w1;
  const float* src = l0 + img * (long)H8 * W8;
  for (int i = threadIdx.x; i < h1 * w1; i += 256) {
    const int y = i / w1, x = i - y * w1;
    const float* r0 = src + (2 * y) * W8 + 2 * x;
    const float* r1 = r0 + W8;
    const float v = (((r0[0] + r0[1]) + r1[0]) + r1[1]) * 0.25f;
    s1[i] = v;
    l1[img * (long)h1 * w1 + i] = v;
  }
  __syncthreads();
  if (l2 == nullptr) return;
  for (int i = threadIdx.x; i < h2 * w2; i += 256) {
    const int y = i / w2, x = i - y * w2;
    const float* r0 = s1 + (2 * y) * w1 + 2 * x;
    const float* r1 = r0 + w1;
    const float v = (((r0[0] + r0[1]) + r1[0]) + r1[1]) * 0.25f;
    s2[i] = v;
    l2[img * (long)h2 * w2 + i] = v;
  }
  __syncthreads();
  if (l3 == nullptr) return;
  for (int i = threadIdx.x; i < h3 * w3; i += 256) {
    const int y = i / w3, x = i - y * w3;
    const float* r0 = s2 + (2 * y) * w2 + 2 * x;
    const float* r1 = r0 + w2;
    l3[img * (long)h3 * w3 + i] = (((r0[0] + r0[1]) + r1[0]) + r1[1]) * 0.25f;
  }
}

int launch_corr_pyramid(const float* l0, float* l1, float* l2, float* l3, long nimg, int H8, int W8, hipStream_t s) {
  if (H8 < 2 || W8 < 2) return CRAFT_ERR_UNSUPPORTED;
  const int h1 = H8 / 2, w1 = W8 / 2, h2 = h1 / 2, w2 = w1 / 2;
  const size_t lds = sizeof(float) * ((size_t)h1 * w1 + (size_t)h2 * w2 + 4);
  if (lds > 150 * 1024) return CRAFT_ERR_UNSUPPORTED;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_corr_pyramid), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k_corr_pyramid, dim3((unsigned)nimg), dim3(256), lds, s, l0, l1, l2, l3, H8, W8);
  return (int)hipGetLastError();
}

// (sum, sum^2) -> (mean, 1/sqrt(var+eps)) per sample (corr.py:200-204); identity when !do_norm
__global__ void k_corr_stats(const double* __restrict__ sums, float* __restrict__ mu_rstd, int B, double count, int do_norm) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (!do_norm) { mu_rstd[2 * b] = 0.f; mu_rstd[2 * b + 1] = 1.f; return; }
  const double mu = sums[2 * b] / count;
  double var = sums[2 * b + 1] / count - mu * mu;
  if (var < 0.0) var = 0.0;
  mu_rstd[2 * b] = (float)mu;
  mu_rstd[2 * b + 1] = (float)(1.0 / sqrt(var + (double)CRAFT_LN_EPS));
}
int launch_corr_stats(const double* sums, float* mu_rstd, int B, double count, int do_norm, hipStream_t s) {
  hipLaunchKernelGGL(k_corr_stats, dim3((B + 63) / 64), dim3(64), 0, s, sums, mu_rstd, B, count, do_norm);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// correlation lookup (CorrBlock.__call__, corr.py:47-71; bilinear_sampler utils.py:65-79).
// One wave per query pixel.  Per level: gather the (2r+2)^2 patch around floor(coords/2^l) into LDS,
// normalising in-bounds taps with the lazy global LayerNorm ((v-mu)*rstd; out of bounds = 0, which is
// exactly grid_sample's zero padding applied to the normalised volume), then each lane blends 4 patch
// values per output channel k = l*(2r+1)^2 + a*(2r+1) + b  with  X = x/2^l + a - r,  Y = y/2^l + b - r
// (the x offset runs along the FIRST window axis).
// ---------------------------------------------------------------------------------------------
// RADIUS > 0: the window radius as a compile-time constant (the reference's radius 4).  The patch gather and the blend index their
// elements by `idx / PS`, `k / win^2`, `rem / win`: with run-time divisors those are ~30 VALU instructions each, 34 of them per lane -- the
// kernel was VALU-bound on address arithmetic (52 us at 448x1024, batch 4), not on its gathers.  RADIUS = 0 keeps the run-time form.
#define LOOKUP_MAXP 16
template <int RADIUS>
__global__ __launch_bounds__(256) void k_corr_lookup(const float* __restrict__ l0, const float* __restrict__ l1,
                                                     const float* __restrict__ l2, const float* __restrict__ l3, int levels,
                                                     const float* __restrict__ mu_rstd, const float* __restrict__ coords,
                                                     int N, int H8, int W8, int radius, float* __restrict__ out, long ldo,
                                                     int lvl_stride, int col_off, long nq, int tiled) {
  __shared__ float patch[4][4][LOOKUP_MAXP * LOOKUP_MAXP];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long q0 = (long)blockIdx.x * 4 + wv;
  const bool valid = q0 < nq;
  const long q = valid ? q0 : nq - 1;      // idle waves shadow the last query (no early return before the barrier)
  const int b = (int)(q / N);
  const float cx = coords[2 * q], cy = coords[2 * q + 1];
  const float mu = mu_rstd[2 * b], rstd = mu_rstd[2 * b + 1];
  if (RADIUS) radius = RADIUS;
  const int PS = RADIUS ? 2 * RADIUS + 2 : 2 * radius + 2, win = RADIUS ? 2 * RADIUS + 1 : 2 * radius + 1;
  float fxs[4], fys[4];
  const float* lv[4] = {l0, l1, l2, l3};
  // Phase 1 requests the patch elements of ALL levels before any of them is used (one HBM round trip per query instead of one per
  // level: the level loop with its load -> normalise -> LDS store body made four dependent round trips, ~2 us each under load), phase 2
  // normalises and publishes them.  NI = 64-lane passes over a (2r+2)^2 patch.
  constexpr int NI = RADIUS ? ((2 * RADIUS + 2) * (2 * RADIUS + 2) + 63) / 64 : (LOOKUP_MAXP * LOOKUP_MAXP) / 64;
  float vals[4][NI];
  unsigned okm = 0u;                       // bit l * NI + j: element j of level l is inside the image
  // (everything that reads the coordinates first, for all levels: a wait for them placed between the gathers of two levels would --
  // the vector-memory counter retires in order -- also drain the gathers already in flight)
  int x0s[4], y0s[4];
  {
    float sc = 1.f;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const float X = cx * sc, Y = cy * sc;
      const float x0f = floorf(X), y0f = floorf(Y);
      fxs[l] = X - x0f;
      fys[l] = Y - y0f;
      // clamp the integer base far outside the image so int conversion can not overflow
      x0s[l] = (int)fminf(fmaxf(x0f, -100000.f), 100000.f) - radius;
      y0s[l] = (int)fminf(fmaxf(y0f, -100000.f), 100000.f) - radius;
      sc *= 0.5f;
    }
  }
  asm volatile("" : "+v"(x0s[0]), "+v"(x0s[1]), "+v"(x0s[2]), "+v"(x0s[3]), "+v"(y0s[0]), "+v"(y0s[1]), "+v"(y0s[2]), "+v"(y0s[3]));   // (materialised here)
  int h = H8, w = W8;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    if (l < levels) {
      const int x0 = x0s[l], y0 = y0s[l];
      // CRAFT_PYR_TILED (craft_corr_build_pyramid's layout): level 0 in 8 x 16 tiles of 128 floats, level 1 in 4 x 8 tiles of 32
      const bool tl = tiled && l < 2;
      const int tsy = l == 0 ? 3 : 2, tsx = l == 0 ? 4 : 3;                    // log2 of the tile height / width
      const int ntx = (w + (1 << tsx) - 1) >> tsx, nty = (h + (1 << tsy) - 1) >> tsy;
      const float* img = lv[l] + q * (tl ? ((long)nty * ntx) << (tsy + tsx) : (long)h * w);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int idx = lane + 64 * j;
        const int py = idx / PS, px = idx - py * PS;
        const int y = y0 + py, x = x0 + px;
        // unconditional load from a clamped address, validity kept as a mask bit: a load under the bounds test compiles to
        // branch + load + s_waitcnt vmcnt(0) -- eight dependent round trips per query
        const bool ok = idx < PS * PS && y >= 0 && y < h && x >= 0 && x < w;
        const int yc = min(max(y, 0), h - 1), xc = min(max(x, 0), w - 1);
        const int off_t = ((((yc >> tsy) * ntx + (xc >> tsx)) << (tsy + tsx)) + ((yc & ((1 << tsy) - 1)) << tsx) + (xc & ((1 << tsx) - 1)));
        const int off = tl ? off_t : yc * w + xc;            // (one query's level: < 2^31 elements)
        okm |= ok ? 1u << (l * NI + j) : 0u;
        vals[l][j] = img[off];
      }
      h >>= 1; w >>= 1;
    }
  }
#pragma unroll
  for (int l = 0; l < 4; ++l)
    if (l < levels) {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int idx = lane + 64 * j;
        const int py = idx / PS, px = idx - py * PS;
        if (idx < PS * PS) patch[wv][l][py * LOOKUP_MAXP + px] = ((okm >> (l * NI + j)) & 1u) ? (vals[l][j] - mu) * rstd : 0.f;
      }
    }
  __syncthreads();
  if (!valid) return;
  const int nch = levels * win * win;
  float* o = out + q * ldo;
  for (int k = lane; k < nch; k += 64) {
    const int l = k / (win * win), rem = k - l * win * win;
    const int a = rem / win, bb = rem - a * win;
    const float fx = fxs[l], fy = fys[l];
    const float* P = &patch[wv][l][bb * LOOKUP_MAXP + a];
    const float nw = (1.f - fx) * (1.f - fy), ne = fx * (1.f - fy), sw = (1.f - fx) * fy, se = fx * fy;
    o[l * lvl_stride + col_off + rem] = ((P[0] * nw + P[1] * ne) + P[LOOKUP_MAXP] * sw) + P[LOOKUP_MAXP + 1] * se;
  }
}

int launch_corr_lookup(const float* l0, const float* l1, const float* l2, const float* l3, int levels, const float* mu_rstd,
                       const float* coords, int B, int H8, int W8, int radius, float* out, long ldo, int lvl_stride, int col_off,
                       int tiled, hipStream_t s) {
  if (levels < 1 || levels > 4 || radius < 0 || 2 * radius + 2 > LOOKUP_MAXP) return CRAFT_ERR_UNSUPPORTED;
  const int win2 = (2 * radius + 1) * (2 * radius + 1);
  if (lvl_stride <= 0) lvl_stride = win2;            // default: one volume, levels back to back
  if (lvl_stride < win2 || col_off < 0 || col_off + win2 > lvl_stride) return CRAFT_ERR_ARG;
  // every level must hold at least one pixel: the kernel loads unconditionally from CLAMPED addresses (min(max(y, 0), h - 1) ...), which
  // on an empty level (h or w == 0) is offset -1 of a null / empty buffer
  if ((H8 >> (levels - 1)) < 1 || (W8 >> (levels - 1)) < 1) return CRAFT_ERR_ARG;
  const long nq = (long)B * H8 * W8;
  if (radius == 4)
    hipLaunchKernelGGL(k_corr_lookup<4>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, l0, l1, l2, l3, levels, mu_rstd, coords,
                       H8 * W8, H8, W8, radius, out, ldo, lvl_stride, col_off, nq, tiled);
  else
    hipLaunchKernelGGL(k_corr_lookup<0>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, l0, l1, l2, l3, levels, mu_rstd, coords,
                       H8 * W8, H8, W8, radius, out, ldo, lvl_stride, col_off, nq, tiled);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// convf1: 7x7 conv 2 -> 128 + ReLU on the flow field (update.py:75,82).  w packed [7*7*2][128]
// (tap-major: ((ky*7+kx)*2 + c), output channel contiguous).  block = 2 pixels x 128 channels.
// ---------------------------------------------------------------------------------------------
// A block owns CF1_SEG consecutive pixels of one image row; thread = (output channel, half of the segment).  The 98
// weights of the thread's channel live in registers for the whole segment and the 7 x (CF1_SEG + 6) x 2 flow patch in
// LDS (broadcast reads), so the weight table is read once per block instead of once per pixel.
constexpr int CF1_SEG = 32, CF1_PW = CF1_SEG + 6;
__global__ __launch_bounds__(256) void k_convf1(const float* __restrict__ flow, const float* __restrict__ w,
                                                const float* __restrict__ bias, int H8, int W8,
                                                float* __restrict__ out, long ldo) {
  __shared__ __attribute__((aligned(16))) float pat[7][CF1_PW * 2];
  const int tid = threadIdx.x, co = tid & 127, half = tid >> 7;
  const int segs = (W8 + CF1_SEG - 1) / CF1_SEG;
  int bid = blockIdx.x;
  const int sx = bid % segs; bid /= segs;
  const int y = bid % H8;
  const int b = bid / H8;
  const int x0 = sx * CF1_SEG;
  const long img = (long)b * H8 * W8;
  for (int i = tid; i < 7 * CF1_PW * 2; i += 256) {
    const int ky = i / (CF1_PW * 2), rem = i - ky * (CF1_PW * 2);
    const int px = rem >> 1, c = rem & 1;
    const int yy = y + ky - 3, xx = x0 + px - 3;
    pat[ky][rem] = (yy >= 0 && yy < H8 && xx >= 0 && xx < W8) ? flow[(img + (long)yy * W8 + xx) * 2 + c] : 0.f;
  }
  float wr[98];
#pragma unroll
  for (int k = 0; k < 98; ++k) wr[k] = w[k * 128 + co];
  const float bco = bias[co];
  __syncthreads();
  constexpr int PER = CF1_SEG / 2;
  for (int i = 0; i < PER; ++i) {
    const int px = half * PER + i;            // pixel within the segment (uniform per wave)
    if (x0 + px >= W8) break;
    float acc = bco;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const float2* row = reinterpret_cast<const float2*>(&pat[ky][px * 2]);
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const float2 f = row[kx];
        acc += f.x * wr[(ky * 7 + kx) * 2] + f.y * wr[(ky * 7 + kx) * 2 + 1];
      }
    }
    out[(img + (long)y * W8 + x0 + px) * ldo + co] = fmaxf(acc, 0.f);
  }
}
int launch_convf1(const float* flow, const float* w, const float* bias, int B, int H8, int W8, float* out, long ldo,
                  hipStream_t s) {
  const int segs = (W8 + CF1_SEG - 1) / CF1_SEG;
  hipLaunchKernelGGL(k_convf1, dim3((unsigned)(B * H8 * segs)), dim3(256), 0, s, flow, w, bias, H8, W8, out, ldo);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// FlowHead.conv2: 3x3 conv 256 -> 2 (update.py:12,16) fused with the coordinate update
// coords1 += delta (network.py:247) and flow = coords1 - coords0.  One wave per pixel; lane owns 4 channels.
// w packed [2][3*3][256].
// ---------------------------------------------------------------------------------------------
// A wave owns FH2_SEG consecutive pixels of one image row; a lane owns 4 of the 256 channels.  The lane's 72 weights
// stay in registers, and the 3x3 window slides along x: per pixel only the new column (3 float4) is loaded.
constexpr int FH2_SEG = 16;
__global__ __launch_bounds__(256) void k_flow_head2(const float* __restrict__ hid, const float* __restrict__ w,
                                                    const float* __restrict__ bias, int H8, int W8, int nseg_total,
                                                    float* __restrict__ coords1, const float* __restrict__ coords0,
                                                    float* __restrict__ flow, float* __restrict__ delta) {
  const int lane = threadIdx.x & 63;
  const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seg >= nseg_total) return;
  const int segs = (W8 + FH2_SEG - 1) / FH2_SEG;
  int t = seg;
  const int sx = t % segs; t /= segs;
  const int y = t % H8;
  const int b = t / H8;
  const int x0 = sx * FH2_SEG;
  const long img = (long)b * H8 * W8;
  float4 w0[9], w1[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    w0[k] = *reinterpret_cast<const float4*>(w + (0 * 9 + k) * 256 + lane * 4);
    w1[k] = *reinterpret_cast<const float4*>(w + (1 * 9 + k) * 256 + lane * 4);
  }
  const float b0 = bias[0], b1 = bias[1];
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  struct Col3 { float4 r[3]; };
  auto col = [&](int xx) __attribute__((always_inline)) {     // column xx of the 3 rows (zero padding), by value (arrays passed by
    Col3 c;                                                    // reference to the lambda ended up in scratch memory: 48 bytes per lane)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int yy = y + r - 1;
      const bool ok = yy >= 0 && yy < H8 && xx >= 0 && xx < W8;
      const float4 v = *reinterpret_cast<const float4*>(hid + (img + (long)min(max(yy, 0), H8 - 1) * W8 + min(max(xx, 0), W8 - 1)) * 256 + lane * 4);
      c.r[r] = ok ? v : zero;
    }
    return c;
  };
  Col3 cl = col(x0 - 1), cm = col(x0), cr;
  const int n = min(FH2_SEG, W8 - x0);
  // Lane i keeps pixel i's coordinates and, as the window slides, its delta: the segment's coordinates are read once before the loop and
  // written once after it (coalesced), instead of a load -> wait -> store sequence under `lane == 0` per pixel (16 dependent round trips).
  const long pix0 = img + (long)y * W8 + x0;
  float2 c1v = make_float2(0.f, 0.f), c0v = make_float2(0.f, 0.f);
  if (lane < n) {
    c1v = *reinterpret_cast<const float2*>(coords1 + 2 * (pix0 + lane));
    c0v = *reinterpret_cast<const float2*>(coords0 + 2 * (pix0 + lane));
  }
  float mdx = 0.f, mdy = 0.f;
  for (int i = 0; i < n; ++i) {
    cr = col(x0 + i + 1);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float4 h0 = cl.r[r], h1 = cm.r[r], h2 = cr.r[r];
      const float4 u0 = w0[r * 3], u1 = w0[r * 3 + 1], u2 = w0[r * 3 + 2];
      const float4 v0 = w1[r * 3], v1 = w1[r * 3 + 1], v2 = w1[r * 3 + 2];
      a0 += h0.x * u0.x + h0.y * u0.y + h0.z * u0.z + h0.w * u0.w + h1.x * u1.x + h1.y * u1.y + h1.z * u1.z + h1.w * u1.w +
            h2.x * u2.x + h2.y * u2.y + h2.z * u2.z + h2.w * u2.w;
      a1 += h0.x * v0.x + h0.y * v0.y + h0.z * v0.z + h0.w * v0.w + h1.x * v1.x + h1.y * v1.y + h1.z * v1.z + h1.w * v1.w +
            h2.x * v2.x + h2.y * v2.y + h2.z * v2.z + h2.w * v2.w;
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    mdx = lane == i ? a0 + b0 : mdx;
    mdy = lane == i ? a1 + b1 : mdy;
    cl = cm; cm = cr;
  }
  if (lane < n) {
    const long pix = pix0 + lane;
    const float nx = c1v.x + mdx, ny = c1v.y + mdy;
    *reinterpret_cast<float2*>(coords1 + 2 * pix) = make_float2(nx, ny);
    *reinterpret_cast<float2*>(flow + 2 * pix) = make_float2(nx - c0v.x, ny - c0v.y);
    if (delta) *reinterpret_cast<float2*>(delta + 2 * pix) = make_float2(mdx, mdy);
  }
}
int launch_flow_head2(const float* hid, const float* w, const float* bias, int B, int H8, int W8, float* coords1,
                      const float* coords0, float* flow, float* delta, hipStream_t s) {
  const int nseg = B * H8 * ((W8 + FH2_SEG - 1) / FH2_SEG);
  hipLaunchKernelGGL(k_flow_head2, dim3((unsigned)((nseg + 3) / 4)), dim3(256), 0, s, hid, w, bias, H8, W8, nseg, coords1,
                     coords0, flow, delta);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// convex upsampling (CRAFT.upsample_flow, network.py:151-162).  One wave per low-res pixel, lane = the
// 8x8 sub-pixel: softmax over the 9 neighbours of mask[k*64 + lane], weighted sum of 8*flow (zero pad).
// mask: tokens [B*N, 576] (already 0.25*(conv+bias)); flow tokens [B*N, 2]; up: NCHW [B, 2, 8*H8, 8*W8].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_convex_upsample(const float* __restrict__ mask, const float* __restrict__ flow,
                                                         int H8, int W8, long npix, float* __restrict__ up) {
  const int lane = threadIdx.x & 63;
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= npix) return;
  const int hw = H8 * W8;
  const int b = (int)(pix / hw), rem = (int)(pix - (long)b * hw);
  const int y = rem / W8, x = rem - y * W8;
  float mk[9];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 9; ++k) { mk[k] = mask[pix * 576 + k * 64 + lane]; mx = fmaxf(mx, mk[k]); }
  float den = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) { mk[k] = expf(mk[k] - mx); den += mk[k]; }
  float ox = 0.f, oy = 0.f;
  // (the 9 neighbour flows from clamped addresses, zero padding applied to the values: loads under the bounds test were nine
  // dependent round trips)
  float2 fl[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = min(max(y + k / 3 - 1, 0), H8 - 1), xx = min(max(x + k % 3 - 1, 0), W8 - 1);
    fl[k] = *reinterpret_cast<const float2*>(flow + ((long)b * hw + yy * W8 + xx) * 2);
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    const bool ok = yy >= 0 && yy < H8 && xx >= 0 && xx < W8;
    const float fx = ok ? 8.f * fl[k].x : 0.f, fy = ok ? 8.f * fl[k].y : 0.f;
    const float pk = mk[k] / den;
    ox += pk * fx;
    oy += pk * fy;
  }
  const int i = lane >> 3, j = lane & 7;
  const long H = 8L * H8, W = 8L * W8;
  up[(((long)b * 2 + 0) * H + 8 * y + i) * W + 8 * x + j] = ox;
  up[(((long)b * 2 + 1) * H + 8 * y + i) * W + 8 * x + j] = oy;
}
int launch_convex_upsample(const float* mask, const float* flow, int B, int H8, int W8, float* up, hipStream_t s) {
  const long npix = (long)B * H8 * W8;
  hipLaunchKernelGGL(k_convex_upsample, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, s, mask, flow, H8, W8, npix, up);
  return (int)hipGetLastError();
}

// coords_grid + flow_init (utils.py:82-85, network.py:219-222): tokens [B*N, 2] with (x, y) order
__global__ void k_coords_init(const float* __restrict__ flow_init, int H8, int W8, long npix, float* __restrict__ c0,
                              float* __restrict__ c1, float* __restrict__ flow) {
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int hw = H8 * W8;
  const int b = (int)(pix / hw), rem = (int)(pix - (long)b * hw);
  const int y = rem / W8, x = rem - y * W8;
  float fx = 0.f, fy = 0.f;
  if (flow_init) { fx = flow_init[((long)b * 2 + 0) * hw + rem]; fy = flow_init[((long)b * 2 + 1) * hw + rem]; }
  c0[2 * pix] = (float)x; c0[2 * pix + 1] = (float)y;
  const float nx = (float)x + fx, ny = (float)y + fy;
  c1[2 * pix] = nx; c1[2 * pix + 1] = ny;
  flow[2 * pix] = nx - (float)x; flow[2 * pix + 1] = ny - (float)y;
}
int launch_coords_init(const float* flow_init_nchw, int B, int H8, int W8, float* coords0, float* coords1, float* flow,
                       hipStream_t s) {
  const long npix = (long)B * H8 * W8;
  hipLaunchKernelGGL(k_coords_init, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, flow_init_nchw, H8, W8, npix,
                     coords0, coords1, flow);
  return (int)hipGetLastError();
}

// (sum, sum^2) -> (mean, 1/sqrt(var + eps)) for n independent populations of `count` samples each
// (lazy InstanceNorm of the HIP encoders: one population per (image, channel)).
// One wave per population: its 64 lanes fetch the CRAFT_STATS_REPLICAS = 64 partial sums in one go (a thread per population walked
// them serially: 11 us per call, nine calls on the feature encoder's critical path).
__global__ __launch_bounds__(256) void k_stats_finalize(const double* __restrict__ sums, long n, double count, float eps, float* __restrict__ mr) {
  static_assert(CRAFT_STATS_REPLICAS == 64, "one replica per lane");
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  double a = sums[(lane * n + i) * 2], q = sums[(lane * n + i) * 2 + 1];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
  if (lane) return;
  const double mu = a / count;
  double var = q / count - mu * mu;
  if (var < 0.0) var = 0.0;
  mr[2 * i] = (float)mu;
  mr[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}
int launch_stats_finalize(const double* sums, long n, double count, float eps, float* mean_rstd, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_stats_finalize, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, sums, n, count, eps, mean_rstd);
  return (int)hipGetLastError();
}

// ResidualBlock tail (extractor.py:58-64) with lazy norms: out = relu( fx(x) + fy(y) ),
//   fx(x) = x, (x - mean)*rstd (downsample branch: norm3, no ReLU) or relu((x - mean)*rstd) (flag bit 1 of
//           y_relu: the stem's norm1 + ReLU applied lazily to the identity branch of layer1.0)
//   fy(y) = y, relu(y) or relu((y - mean)*rstd)   (norm2 + ReLU of conv2's raw output)
// tokens [B, HW, C]; norm tables [B][C][2].
__global__ void k_residual_relu(const float* __restrict__ x, long ldx, const float* __restrict__ xn, const float* __restrict__ y,
                                long ldy, const float* __restrict__ yn, int y_relu, int HW, int C4, long tot,
                                float* __restrict__ out, long ldo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one float4 of channels per thread
  if (i >= tot) return;
  const long tok = i / C4;
  const int c = (int)(i - tok * C4) * 4;
  const long b = tok / HW;
  float4 vx = *reinterpret_cast<const float4*>(x + tok * ldx + c);
  float4 vy = *reinterpret_cast<const float4*>(y + tok * ldy + c);
  const int C = C4 * 4;
  if (xn) {
    const float* t = xn + (b * C + c) * 2;
    const float4 t0 = *reinterpret_cast<const float4*>(t), t1 = *reinterpret_cast<const float4*>(t + 4);
    vx.x = (vx.x - t0.x) * t0.y; vx.y = (vx.y - t0.z) * t0.w; vx.z = (vx.z - t1.x) * t1.y; vx.w = (vx.w - t1.z) * t1.w;
  }
  if (y_relu & 2) { vx.x = fmaxf(vx.x, 0.f); vx.y = fmaxf(vx.y, 0.f); vx.z = fmaxf(vx.z, 0.f); vx.w = fmaxf(vx.w, 0.f); }
  if (yn) {
    const float* t = yn + (b * C + c) * 2;
    const float4 t0 = *reinterpret_cast<const float4*>(t), t1 = *reinterpret_cast<const float4*>(t + 4);
    vy.x = (vy.x - t0.x) * t0.y; vy.y = (vy.y - t0.z) * t0.w; vy.z = (vy.z - t1.x) * t1.y; vy.w = (vy.w - t1.z) * t1.w;
  }
  if (y_relu & 1) { vy.x = fmaxf(vy.x, 0.f); vy.y = fmaxf(vy.y, 0.f); vy.z = fmaxf(vy.z, 0.f); vy.w = fmaxf(vy.w, 0.f); }
  float4 o;
  o.x = fmaxf(vx.x + vy.x, 0.f); o.y = fmaxf(vx.y + vy.y, 0.f); o.z = fmaxf(vx.z + vy.z, 0.f); o.w = fmaxf(vx.w + vy.w, 0.f);
  *reinterpret_cast<float4*>(out + tok * ldo + c) = o;
}
int launch_residual_relu(const float* x, long ldx, const float* xnorm, const float* y, long ldy, const float* ynorm, int y_relu,
                         int B, int HW, int C, float* out, long ldo, hipStream_t s) {
  if (C % 4 || (ldx & 3) || (ldy & 3) || (ldo & 3)) return CRAFT_ERR_ALIGN;
  const long tot = (long)B * HW * (C / 4);
  hipLaunchKernelGGL(k_residual_relu, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, x, ldx, xnorm, y, ldy, ynorm, y_relu, HW,
                     C / 4, tot, out, ldo);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Encoder stem: 7x7 / stride 2 / pad 3 convolution 3 -> 64 (extractor.py:139, :181) fused with the input
// normalisation 2*(x/255)-1 of CRAFT.forward (network.py:169-173), bias, optional ReLU (cnet: BatchNorm folded
// into w / bias) and optional per-(image, channel) (sum, sum^2) for a lazy InstanceNorm (fnet).
// Direct fp32 FMA kernel (K = 147 is too ragged for the MFMA engine and the stem is 3.5 % of the encoder
// flops): a block owns 16 x 32 output pixels, stages the 37 x 69 x 3 input patch and the [147][64] weights in
// LDS; each thread accumulates 2 pixels x 64 channels, so every weight read from LDS feeds 2 FMAs.
// image: NCHW [B][3][H][W] raw 0..255;  w: [147 = (ky*7+kx)*3+c][64];  out: tokens [B][(H/2)*(W/2)][64].
// ---------------------------------------------------------------------------------------------
#define STEM_TH 16
#define STEM_TW 32
#define STEM_PH (2 * STEM_TH + 5)
#define STEM_PW (2 * STEM_TW + 5)
#define STEM_PLD (STEM_PW + 1)
__global__ __launch_bounds__(256) void k_stem7x7(const float* __restrict__ img, const float* __restrict__ w,
                                                 const float* __restrict__ bias, int act, int H, int W, float* __restrict__ out,
                                                 double* __restrict__ stats) {
  extern __shared__ float sm[];
  float* sw = sm;                                   // [147][64]
  float* sp = sm + 147 * 64;                        // [3][STEM_PH][STEM_PLD]
  float* sred = sp + 3 * STEM_PH * STEM_PLD;        // [4 waves][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H2 = H / 2, W2 = W / 2;
  const int tiles_x = (W2 + STEM_TW - 1) / STEM_TW, tiles_y = (H2 + STEM_TH - 1) / STEM_TH;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int oy0 = ty * STEM_TH, ox0 = tx * STEM_TW;
  for (int i = tid; i < 147 * 64 / 4; i += 256) reinterpret_cast<float4*>(sw)[i] = reinterpret_cast<const float4*>(w)[i];
  const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
  for (int i = tid; i < 3 * STEM_PH * STEM_PW; i += 256) {
    const int c = i / (STEM_PH * STEM_PW), rem = i - c * (STEM_PH * STEM_PW);
    const int r = rem / STEM_PW, q = rem - r * STEM_PW;
    const int y = iy0 + r, x = ix0 + q;
    float v = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) v = 2.f * (img[(((long)b * 3 + c) * H + y) * W + x] / 255.f) - 1.f;
    sp[(c * STEM_PH + r) * STEM_PLD + q] = v;
  }
  __syncthreads();
  const int py = tid >> 4, px = tid & 15;           // pixels (py, px) and (py, px + 16) of the tile
  float a0[64], a1[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
  for (int ky = 0; ky < 7; ++ky)
    for (int kx = 0; kx < 7; ++kx)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* prow = sp + (c * STEM_PH + 2 * py + ky) * STEM_PLD + kx;
        const float v0 = prow[2 * px], v1 = prow[2 * (px + 16)];
        const float4* wr = reinterpret_cast<const float4*>(sw + ((ky * 7 + kx) * 3 + c) * 64);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float4 ww = wr[j];
          a0[4 * j + 0] += v0 * ww.x; a0[4 * j + 1] += v0 * ww.y; a0[4 * j + 2] += v0 * ww.z; a0[4 * j + 3] += v0 * ww.w;
          a1[4 * j + 0] += v1 * ww.x; a1[4 * j + 1] += v1 * ww.y; a1[4 * j + 2] += v1 * ww.z; a1[4 * j + 3] += v1 * ww.w;
        }
      }
  const int oy = oy0 + py, ox_a = ox0 + px, ox_b = ox0 + px + 16;
  const bool ok_a = oy < H2 && ox_a < W2, ok_b = oy < H2 && ox_b < W2;
  float* o_a = out + (((long)b * H2 + oy) * W2 + ox_a) * 64;
  float* o_b = out + (((long)b * H2 + oy) * W2 + ox_b) * 64;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float4 bb = reinterpret_cast<const float4*>(bias)[j];
    float4 r0 = make_float4(a0[4 * j] + bb.x, a0[4 * j + 1] + bb.y, a0[4 * j + 2] + bb.z, a0[4 * j + 3] + bb.w);
    float4 r1 = make_float4(a1[4 * j] + bb.x, a1[4 * j + 1] + bb.y, a1[4 * j + 2] + bb.z, a1[4 * j + 3] + bb.w);
    if (act == CRAFT_ACT_RELU) {
      r0.x = fmaxf(r0.x, 0.f); r0.y = fmaxf(r0.y, 0.f); r0.z = fmaxf(r0.z, 0.f); r0.w = fmaxf(r0.w, 0.f);
      r1.x = fmaxf(r1.x, 0.f); r1.y = fmaxf(r1.y, 0.f); r1.z = fmaxf(r1.z, 0.f); r1.w = fmaxf(r1.w, 0.f);
    }
    a0[4 * j] = r0.x; a0[4 * j + 1] = r0.y; a0[4 * j + 2] = r0.z; a0[4 * j + 3] = r0.w;
    a1[4 * j] = r1.x; a1[4 * j + 1] = r1.y; a1[4 * j + 2] = r1.z; a1[4 * j + 3] = r1.w;
    if (ok_a) reinterpret_cast<float4*>(o_a)[j] = r0;
    if (ok_b) reinterpret_cast<float4*>(o_b)[j] = r1;
  }
  if (stats) {
    // per-channel (sum, sum^2) over the tile's valid pixels: wave reduce, then across the 4 waves through LDS
#pragma unroll
    for (int ch = 0; ch < 64; ++ch) {
      const float v0 = ok_a ? a0[ch] : 0.f, v1 = ok_b ? a1[ch] : 0.f;
      const float s1 = wave_sum(v0 + v1), s2 = wave_sum(v0 * v0 + v1 * v1);
      if (lane == 0) { sred[wave * 128 + ch] = s1; sred[wave * 128 + 64 + ch] = s2; }
    }
    __syncthreads();
    if (tid < 128) {
      const float t = sred[tid] + sred[128 + tid] + sred[256 + tid] + sred[384 + tid];
      const int ch = tid & 63, which = tid >> 6;
      atomicAdd(&stats[(((long)(blockIdx.x % CRAFT_STATS_REPLICAS) * (gridDim.x / (tiles_x * tiles_y)) + b) * 64 + ch) * 2 + which], (double)t);
    }
  }
}
int launch_stem7x7(const float* img, const float* w, const float* bias, int act, int B, int H, int W, float* out, double* stats,
                   hipStream_t s) {
  if ((H & 1) || (W & 1)) return CRAFT_ERR_ALIGN;
  const int H2 = H / 2, W2 = W / 2;
  const int tiles = ((W2 + STEM_TW - 1) / STEM_TW) * ((H2 + STEM_TH - 1) / STEM_TH) * B;
  const size_t lds = sizeof(float) * (147 * 64 + 3 * STEM_PH * STEM_PLD + 4 * 128);
  // the attribute is per device: one bit per device ordinal (set once each; a benign race sets it twice)
  static std::atomic<unsigned long long> attr_done{0ull};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_relaxed) & bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_stem7x7), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_done.fetch_or(bit, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(k_stem7x7, dim3(tiles), dim3(256), lds, s, img, w, bias, act, H, W, out, stats);
  return (int)hipGetLastError();
}

// tokens [B, HW, ld] (first C columns) -> NCHW [B, C, HW]
__global__ void k_tokens_to_nchw(const float* __restrict__ src, long ld, int C, int HW, long tot, float* __restrict__ dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tot) return;
  const int p = (int)(i % HW);
  const long bc = i / HW;
  const int c = (int)(bc % C);
  const long b = bc / C;
  dst[i] = src[(b * HW + p) * ld + c];
}
int launch_tokens_to_nchw(const float* src, long ld, int B, int C, int HW, float* dst, hipStream_t s) {
  const long tot = (long)B * C * HW;
  hipLaunchKernelGGL(k_tokens_to_nchw, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, ld, C, HW, tot, dst);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Evaluation metrics of the harness (evaluate.py:529, :578-598, :833-841, :911): per valid pixel
//   epe = |pred - gt|_2, mag = |gt + gt_offset|_2;  out[16] (doubles, += : zero them first):
//   [0] sum epe  [1] #valid  [2..4] #(epe < 1, 3, 5)  [5] #(epe > 3 and epe / mag > 0.05)   (KITTI Fl outliers)
//   [6..10] sum epe by magnitude bin [0,1) [1,10) [10,20) [20,30) [30,inf)   [11..15] pixel counts of the bins
// pred / gt NCHW [B][2][H][W]; valid [B][H][W] (>= 0.5 counts) or null (all pixels).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_metrics(const float* __restrict__ pred, const float* __restrict__ gt,
                                                      const float* __restrict__ valid, long hw, long tot, float offx, float offy,
                                                      float max_mag, double* __restrict__ out) {
  __shared__ double s_acc[4][16];
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    if (valid && valid[i] < 0.5f) continue;
    const long b = i / hw, p = i - b * hw;
    const float gx = gt[(2 * b) * hw + p], gy = gt[(2 * b + 1) * hw + p];
    if (max_mag > 0.f && !(sqrtf(gx * gx + gy * gy) < max_mag)) continue;        // train.py:53 (MAX_FLOW)
    const float dx = pred[(2 * b) * hw + p] - gx, dy = pred[(2 * b + 1) * hw + p] - gy;
    const float epe = sqrtf(dx * dx + dy * dy);
    const float mx = gx + offx, my = gy + offy;
    const float mag = sqrtf(mx * mx + my * my);
    a[0] += epe; a[1] += 1.f;
    a[2] += epe < 1.f; a[3] += epe < 3.f; a[4] += epe < 5.f;
    a[5] += (epe > 3.f && epe / mag > 0.05f);
    const int bin = mag < 1.f ? 0 : mag < 10.f ? 1 : mag < 20.f ? 2 : mag < 30.f ? 3 : 4;
#pragma unroll
    for (int k = 0; k < 5; ++k) { a[6 + k] += bin == k ? epe : 0.f; a[11 + k] += bin == k; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float v = wave_sum(a[i]);
    if (lane == 0) s_acc[wave][i] = (double)v;
  }
  __syncthreads();
  if (threadIdx.x < 16) atomicAdd(&out[threadIdx.x], s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x] + s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x]);
}
int launch_flow_metrics(const float* pred, const float* gt, const float* valid, int B, int H, int W, float offx, float offy,
                        float max_mag, double* out, hipStream_t s) {
  const long hw = (long)H * W, tot = hw * B;
  if (tot <= 0) return 0;
  // <= 4096 pixels per thread keeps the fp32 per-thread partial sums exact enough (counts exact below 2^24)
  const long nb = (tot + 255) / 256;
  const unsigned blocks = (unsigned)(nb < 4096 ? nb : 4096);
  hipLaunchKernelGGL(k_flow_metrics, dim3(blocks), dim3(256), 0, s, pred, gt, valid, hw, tot, offx, offy, max_mag, out);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Training-step pieces around the path.
// k_flow_l1: one term of sequence_loss (train.py:44-61): loss += weight * sum(valid * |pred - gt|) / (B*2*H*W), valid =
//   (valid >= 0.5) & (|gt| < max_flow); optional gradient d loss / d pred = weight * valid * sign(pred - gt) / (B*2*H*W).
// k_sumsq: sum of squares (the global gradient norm of clip_grad_norm_, train.py:234).
// k_adamw: torch.optim.AdamW (decoupled weight decay) over a flat buffer, with the clip coefficient min(1, max_norm /
//   (norm + 1e-6)) and the 1/world_size of a summed all-reduce folded in.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_l1(const float* __restrict__ pred, const float* __restrict__ gt,
                                                 const float* __restrict__ valid, long hw, long tot, float weight, float max_flow,
                                                 double* __restrict__ loss, float* __restrict__ grad) {
  __shared__ double s_acc[4];
  const float inv = weight / (float)(2 * tot);
  float a = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const long b = i / hw, p = i - b * hw;
    const float gx = gt[(2 * b) * hw + p], gy = gt[(2 * b + 1) * hw + p];
    const bool ok = (valid == nullptr || valid[i] >= 0.5f) && sqrtf(gx * gx + gy * gy) < max_flow;
    const float dx = pred[(2 * b) * hw + p] - gx, dy = pred[(2 * b + 1) * hw + p] - gy;
    if (ok) a += fabsf(dx) + fabsf(dy);
    if (grad) {
      grad[(2 * b) * hw + p] = ok ? (dx > 0.f ? inv : dx < 0.f ? -inv : 0.f) : 0.f;
      grad[(2 * b + 1) * hw + p] = ok ? (dy > 0.f ? inv : dy < 0.f ? -inv : 0.f) : 0.f;
    }
  }
  const float v = wave_sum(a);
  if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6] = (double)v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (s_acc[0] + s_acc[1] + s_acc[2] + s_acc[3]) * (double)weight / (double)(2 * tot));
}
int launch_flow_l1(const float* pred, const float* gt, const float* valid, int B, int H, int W, float weight, float max_flow,
                   double* loss, float* grad, hipStream_t s) {
  const long hw = (long)H * W, tot = hw * B;
  if (tot <= 0) return 0;
  const long nb = (tot + 255) / 256;
  // (512 blocks: every block ends with one double atomic on the same address; 4 096 of them cost more than the 35 MB pass itself)
  hipLaunchKernelGGL(k_flow_l1, dim3((unsigned)(nb < 512 ? nb : 512)), dim3(256), 0, s, pred, gt, valid, hw, tot, weight, max_flow,
                     loss, grad);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ x, long n, double* __restrict__ out) {
  __shared__ double s_acc[4];
  double a = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const double v = x[i]; a += v * v; }
  // wave reduction in double (two 32-bit shuffles per step)
  for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
  if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, s_acc[0] + s_acc[1] + s_acc[2] + s_acc[3]);
}
int launch_sumsq(const float* x, long n, double* out, hipStream_t s) {
  if (n <= 0) return 0;
  const long nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_sumsq, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, s, x, n, out);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, long n, float lr, float beta1, float beta2, float eps,
                                               float wd, float bc1, float bc2_sqrt, float grad_mul, const double* __restrict__ sumsq,
                                               float max_norm) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float gm = grad_mul;
  if (sumsq != nullptr) {                          // clip_grad_norm_: the norm is that of the (already averaged) gradient
    const float norm = (float)sqrt(*sumsq) * grad_mul;
    if (!(norm < 3.0e38f)) return;                  // non-finite gradient: skip the update whether or not clipping is on (GradScaler.step)
    if (max_norm > 0.f) gm *= fminf(1.f, max_norm / (norm + 1e-6f));
  }
  const float gi = g[i] * gm;
  float pi = p[i] * (1.f - lr * wd);
  const float mi = beta1 * m[i] + (1.f - beta1) * gi;
  const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = pi - (lr / bc1) * (mi / denom);
}
int launch_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float wd,
                 int step, float grad_mul, const double* sumsq, float max_norm, hipStream_t s) {
  if (n <= 0) return 0;
  if (step < 1) return CRAFT_ERR_ARG;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(k_adamw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, wd, bc1,
                     bc2_sqrt, grad_mul, sumsq, max_norm);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Dynamic loss scaling on the device (train.py:215, 231-238: torch.cuda.amp.GradScaler -- scale halves on an overflow, the optimizer
// step is skipped and its step count does not advance, the scale doubles after `growth_interval` clean steps).  The whole decision
// lives in one 32-byte state record so that no host read-back sits on the step: k_scaler_update (one thread) turns the gradient's
// sum of squares into {found_inf, the final gradient multiplier (un-scale x 1/world x clip coefficient), AdamW's bias corrections for
// the count of APPLIED steps} and updates {scale, growth tracker, applied / skipped counters}; k_adamw_dyn reads the record.
// ---------------------------------------------------------------------------------------------
struct ScalerState { float scale; int growth_tracker; int opt_step; int skipped; int found_inf; float gm; float bc1; float bc2_sqrt; };
static_assert(sizeof(ScalerState) == 32, "ScalerState is the 8-word record of include/craft_hip.h");

__global__ void k_scaler_update(const double* __restrict__ sumsq, ScalerState* __restrict__ st, float grad_mul, float max_norm, float beta1,
                                float beta2, float growth, float backoff, int growth_interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  ScalerState s = *st;
  const double ss = *sumsq;
  const float inv = grad_mul / s.scale;                        // the flat gradient holds scale x (sum over ranks of) the gradient
  const float norm = (float)sqrt(ss) * inv;
  if (!(ss >= 0.0) || !(norm < 3.0e38f)) {                     // NaN or inf anywhere in the gradient
    s.found_inf = 1; s.skipped += 1; s.gm = 0.f; s.growth_tracker = 0;
    s.scale = fmaxf(s.scale * backoff, 9.5367431640625e-7f);   // (2^-20 floor: the scale never reaches 0)
  } else {
    s.found_inf = 0; s.opt_step += 1;
    s.gm = inv * (max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f);
    s.bc1 = 1.f - powf(beta1, (float)s.opt_step);
    s.bc2_sqrt = sqrtf(1.f - powf(beta2, (float)s.opt_step));
    s.growth_tracker += 1;
    if (growth_interval > 0 && s.growth_tracker >= growth_interval) { s.scale = fminf(s.scale * growth, 1.8446744e19f); s.growth_tracker = 0; }
  }
  *st = s;
}
int launch_scaler_update(const double* sumsq, void* state, float grad_mul, float max_norm, float beta1, float beta2, float growth,
                         float backoff, int growth_interval, hipStream_t s) {
  if (sumsq == nullptr || state == nullptr || !(backoff > 0.f) || !(growth > 0.f)) return CRAFT_ERR_ARG;
  hipLaunchKernelGGL(k_scaler_update, dim3(1), dim3(64), 0, s, sumsq, (ScalerState*)state, grad_mul, max_norm, beta1, beta2, growth, backoff,
                     growth_interval);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void k_adamw_dyn(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float beta1, float beta2, float eps, float wd,
                                                   const ScalerState* __restrict__ st) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (st->found_inf) return;                                   // skipped step: weights, moments and the step count stay
  const float gi = g[i] * st->gm;
  float pi = p[i] * (1.f - lr * wd);
  const float mi = beta1 * m[i] + (1.f - beta1) * gi;
  const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / st->bc2_sqrt + eps;
  p[i] = pi - (lr / st->bc1) * (mi / denom);
}
int launch_adamw_dyn(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float wd,
                     const void* state, hipStream_t s) {
  if (n <= 0) return 0;
  if (state == nullptr) return CRAFT_ERR_ARG;
  hipLaunchKernelGGL(k_adamw_dyn, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, wd,
                     (const ScalerState*)state);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Warm start: forward_interpolate (core/utils/utils.py:34-62).  out[:, y0, x0] = flow of the source pixel whose warped
// position (x + dx, y + dy) is nearest to (x0, y0), among the sources that land strictly inside (0, W) x (0, H).
// Brute force in float64 (the reference's griddata works on float64 coordinates): one thread per target pixel, the
// sources of the image stream through LDS in chunks; an invalid source is parked at +inf.  Ties: lowest source index.
// ---------------------------------------------------------------------------------------------
constexpr int FI_CHUNK = 1024;
__global__ __launch_bounds__(256) void k_forward_interpolate(const float* __restrict__ flow, int H, int W, float* __restrict__ out) {
  __shared__ double sx[FI_CHUNK], sy[FI_CHUNK];
  const int b = blockIdx.y, HW = H * W;
  const float* fx = flow + (long)b * 2 * HW;
  const float* fy = fx + HW;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int ty = t / W, tx = t - ty * W;
  const double gx = tx, gy = ty;
  double best = INFINITY;
  int bi = -1;
  for (int c0 = 0; c0 < HW; c0 += FI_CHUNK) {
    __syncthreads();
    for (int i = threadIdx.x; i < FI_CHUNK; i += 256) {
      const int j = c0 + i;
      double x1 = INFINITY, y1 = INFINITY;
      if (j < HW) {
        const int y = j / W, x = j - y * W;
        const double ax = (double)x + (double)fx[j], ay = (double)y + (double)fy[j];
        if (ax > 0.0 && ax < (double)W && ay > 0.0 && ay < (double)H) { x1 = ax; y1 = ay; }
      }
      sx[i] = x1; sy[i] = y1;
    }
    __syncthreads();
    const int n = min(FI_CHUNK, HW - c0);
    for (int i = 0; i < n; ++i) {
      const double ex = sx[i] - gx, ey = sy[i] - gy;
      const double d2 = ex * ex + ey * ey;          // (inf for parked sources: never < best)
      if (d2 < best) { best = d2; bi = c0 + i; }
    }
  }
  if (t < HW) {
    out[(long)b * 2 * HW + t] = bi >= 0 ? fx[bi] : 0.f;
    out[(long)b * 2 * HW + HW + t] = bi >= 0 ? fy[bi] : 0.f;
  }
}

int launch_forward_interpolate(const float* flow, int B, int H, int W, float* out, hipStream_t s) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  dim3 grid((H * W + 255) / 256, B);
  hipLaunchKernelGGL(k_forward_interpolate, grid, dim3(256), 0, s, flow, H, W, out);
  return (int)hipGetLastError();
}

}  // namespace craft
