// Training-mode pieces of the CNN encoders (BasicEncoder / ResidualBlock, extractor.py:6-64, 124-196) that the inference
// path folds away: the normalisation layers as real operators with a backward, and the stem's weight gradient.
//
//   out = tail(act((x - mean) * rstd * gamma + beta))            tail(t) = relu(res + t) when a residual input is given
//
// covers nn.InstanceNorm2d (mean / rstd per image and channel, no affine), nn.BatchNorm2d in training (per channel over the
// batch) and in eval / freeze_bn mode (running statistics), followed by the block's ReLU and, for the last norm of a residual
// block, by relu(x + y) (extractor.py:56-64).  Tensors are channels-last tokens [B][N][C] (C % 4 == 0), one float4 of channels
// per thread; the statistics come from the producing convolution's epilogue (craft_conv2d_nhwc_ex `stats`).
//
// Backward of y = norm(x) over a population P (an image's pixels, or the batch's):  with dz = dL/dz and x^ = (x - mean) * rstd
//   dx = gamma * rstd * (dz - mean_P(dz) - x^ * mean_P(dz * x^)),   dgamma = sum dz * x^,   dbeta = sum dz
// (eval-mode BatchNorm: the two means are dropped).  Two passes: a column reduction of (dz, dz * x^) per image and channel,
// then the elementwise pass; both recompute dz from (dy, out, x) instead of storing it.
#include "launch.hpp"

namespace craft {

struct Quad { float4 xh, sc, z; };        // x^, rstd * gamma, normalised + affine value

__device__ __forceinline__ Quad norm_quad(const NormActParams& p, int b, long row, int c) {
  const float4 x = *reinterpret_cast<const float4*>(p.x + row * p.ldx + c);
  const float* m = p.mr + ((long)b * p.mr_bs + c) * 2;
  const float4 m01 = *reinterpret_cast<const float4*>(m), m23 = *reinterpret_cast<const float4*>(m + 4);
  float4 g = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
  if (p.gamma) g = *reinterpret_cast<const float4*>(p.gamma + c);
  if (p.beta) be = *reinterpret_cast<const float4*>(p.beta + c);
  Quad q;
  q.xh = make_float4((x.x - m01.x) * m01.y, (x.y - m01.z) * m01.w, (x.z - m23.x) * m23.y, (x.w - m23.z) * m23.w);
  q.sc = make_float4(m01.y * g.x, m01.w * g.y, m23.y * g.z, m23.w * g.w);
  q.z = make_float4(q.xh.x * g.x + be.x, q.xh.y * g.y + be.y, q.xh.z * g.z + be.z, q.xh.w * g.w + be.w);
  return q;
}

// dz (gradient at the normalised value) and, when there is a residual tail, the gradient that flows to the residual input
__device__ __forceinline__ float4 norm_dz(const NormActParams& p, const Quad& q, long row, int c, float4* dres) {
  float4 g = *reinterpret_cast<const float4*>(p.dy + row * p.ldg + c);
  if (p.has_res) {
    const float4 o = *reinterpret_cast<const float4*>(p.out + row * p.ldo + c);
    g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
    if (dres) *dres = g;
  }
  if (p.act == CRAFT_ACT_RELU) {
    g.x = q.z.x > 0.f ? g.x : 0.f; g.y = q.z.y > 0.f ? g.y : 0.f; g.z = q.z.z > 0.f ? g.z : 0.f; g.w = q.z.w > 0.f ? g.w : 0.f;
  }
  return g;
}

__global__ void k_norm_act_fwd(NormActParams p) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = p.C >> 2;
  if (i >= (long)p.B * p.N * c4) return;
  const long row = i / c4;
  const int c = (int)(i - row * c4) * 4, b = (int)(row / p.N);
  const Quad q = norm_quad(p, b, row, c);
  float4 t = q.z;
  if (p.act == CRAFT_ACT_RELU) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
  if (p.res) {
    const float4 r = *reinterpret_cast<const float4*>(p.res + row * p.ldr + c);
    t.x = fmaxf(t.x + r.x, 0.f); t.y = fmaxf(t.y + r.y, 0.f); t.z = fmaxf(t.z + r.z, 0.f); t.w = fmaxf(t.w + r.w, 0.f);
  }
  *reinterpret_cast<float4*>(p.out + row * p.ldo + c) = t;
}

// grid (row chunks, B); a block owns NORM_ROWS rows of one image; thread -> channel quad (tid % c4) and row lane (tid / c4)
constexpr int NORM_ROWS = 512;
// RES / RELU: p.has_res / p.act == ReLU as compile-time constants.  The per-(image, channel) constants (mean, rstd, gamma, beta) are read
// once per thread and the row loop is straight-line code unrolled by four, so four rows' loads are in flight per thread: with norm_quad /
// norm_dz inside the loop every row re-read the constants and waited for its own round trip behind their uniform branches (2.9 TB/s).
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void k_norm_act_bwd_reduce(NormActParams p) {
  __shared__ float sh[256 * 8];
  const int tid = threadIdx.x, c4 = p.C >> 2, b = blockIdx.y;
  const int rl = tid / c4, nrl = 256 / c4, cq = tid - rl * c4;
  float4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  if (rl < nrl) {
    const int c = cq * 4;
    const float* m = p.mr + ((long)b * p.mr_bs + c) * 2;
    const float4 m01 = *reinterpret_cast<const float4*>(m), m23 = *reinterpret_cast<const float4*>(m + 4);
    float4 g = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
    if (p.gamma) g = *reinterpret_cast<const float4*>(p.gamma + c);
    if (p.beta) be = *reinterpret_cast<const float4*>(p.beta + c);
    const int r0 = blockIdx.x * NORM_ROWS, r1 = min(p.N, r0 + NORM_ROWS);
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += nrl) {
      const long row = (long)b * p.N + r;
      const float4 x = *reinterpret_cast<const float4*>(p.x + row * p.ldx + c);
      float4 dz = *reinterpret_cast<const float4*>(p.dy + row * p.ldg + c);
      if constexpr (RES) {
        const float4 o = *reinterpret_cast<const float4*>(p.out + row * p.ldo + c);
        dz.x = o.x > 0.f ? dz.x : 0.f; dz.y = o.y > 0.f ? dz.y : 0.f; dz.z = o.z > 0.f ? dz.z : 0.f; dz.w = o.w > 0.f ? dz.w : 0.f;
      }
      const float4 xh = make_float4((x.x - m01.x) * m01.y, (x.y - m01.z) * m01.w, (x.z - m23.x) * m23.y, (x.w - m23.z) * m23.w);
      if constexpr (RELU) {
        dz.x = xh.x * g.x + be.x > 0.f ? dz.x : 0.f; dz.y = xh.y * g.y + be.y > 0.f ? dz.y : 0.f;
        dz.z = xh.z * g.z + be.z > 0.f ? dz.z : 0.f; dz.w = xh.w * g.w + be.w > 0.f ? dz.w : 0.f;
      }
      s1.x += dz.x; s1.y += dz.y; s1.z += dz.z; s1.w += dz.w;
      s2.x += dz.x * xh.x; s2.y += dz.y * xh.y; s2.z += dz.z * xh.z; s2.w += dz.w * xh.w;
    }
  }
  float* my = sh + tid * 8;
  my[0] = s1.x; my[1] = s1.y; my[2] = s1.z; my[3] = s1.w; my[4] = s2.x; my[5] = s2.y; my[6] = s2.z; my[7] = s2.w;
  __syncthreads();
  // t < C * 2: channel t / 2, which = t & 1 -> sum over the row lanes
  for (int t = tid; t < p.C * 2; t += 256) {
    const int ch = t >> 1, which = t & 1;
    const int q = ch >> 2, j = ch & 3;
    double acc = 0.0;
    for (int l = 0; l < nrl; ++l) acc += (double)sh[(l * c4 + q) * 8 + which * 4 + j];
    unsafeAtomicAdd(p.sums + ((long)b * p.C + ch) * 2 + which, acc);
  }
}

__global__ void k_norm_act_bwd_apply(NormActParams p) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = p.C >> 2;
  if (i >= (long)p.B * p.N * c4) return;
  const long row = i / c4;
  const int c = (int)(i - row * c4) * 4, b = (int)(row / p.N);
  const Quad q = norm_quad(p, b, row, c);
  float4 dr;
  const float4 dz = norm_dz(p, q, row, c, &dr);
  if (p.has_res && p.dres) *reinterpret_cast<float4*>(p.dres + row * p.lddr + c) = dr;
  float4 m1 = {0.f, 0.f, 0.f, 0.f}, m2 = {0.f, 0.f, 0.f, 0.f};
  if (p.red) {
    const float* m = p.red + ((long)b * p.red_bs + c) * 2;
    const float4 a = *reinterpret_cast<const float4*>(m), bb = *reinterpret_cast<const float4*>(m + 4);
    m1 = make_float4(a.x, a.z, bb.x, bb.z);
    m2 = make_float4(a.y, a.w, bb.y, bb.w);
  }
  float4 dx;
  dx.x = q.sc.x * (dz.x - m1.x - q.xh.x * m2.x);
  dx.y = q.sc.y * (dz.y - m1.y - q.xh.y * m2.y);
  dx.z = q.sc.z * (dz.z - m1.z - q.xh.z * m2.z);
  dx.w = q.sc.w * (dz.w - m1.w - q.xh.w * m2.w);
  *reinterpret_cast<float4*>(p.dx + row * p.lddx + c) = dx;
}

static int check_norm(const NormActParams& p) {
  if (p.B <= 0 || p.N <= 0 || p.C <= 0) return 1;       // empty
  if ((p.C & 3) || p.C > 512 || (p.ldx & 3) || (p.mr_bs != 0 && p.mr_bs != p.C)) return CRAFT_ERR_ALIGN;
  return 0;
}

int launch_norm_act_fwd(const NormActParams& p, hipStream_t s) {
  const int rc = check_norm(p);
  if (rc) return rc == 1 ? 0 : rc;
  if ((p.ldo & 3) || (p.res && (p.ldr & 3))) return CRAFT_ERR_ALIGN;
  const long n = (long)p.B * p.N * (p.C >> 2);
  hipLaunchKernelGGL(k_norm_act_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
  return (int)hipGetLastError();
}
int launch_norm_act_bwd_reduce(const NormActParams& p, hipStream_t s) {
  const int rc = check_norm(p);
  if (rc) return rc == 1 ? 0 : rc;
  if ((p.ldg & 3) || (p.has_res && (p.ldo & 3))) return CRAFT_ERR_ALIGN;
  const dim3 grid((unsigned)((p.N + NORM_ROWS - 1) / NORM_ROWS), (unsigned)p.B);
  const bool relu = p.act == CRAFT_ACT_RELU;
  if (p.has_res) { if (relu) hipLaunchKernelGGL((k_norm_act_bwd_reduce<true, true>), grid, dim3(256), 0, s, p); else hipLaunchKernelGGL((k_norm_act_bwd_reduce<true, false>), grid, dim3(256), 0, s, p); }
  else { if (relu) hipLaunchKernelGGL((k_norm_act_bwd_reduce<false, true>), grid, dim3(256), 0, s, p); else hipLaunchKernelGGL((k_norm_act_bwd_reduce<false, false>), grid, dim3(256), 0, s, p); }
  return (int)hipGetLastError();
}
int launch_norm_act_bwd_apply(const NormActParams& p, hipStream_t s) {
  const int rc = check_norm(p);
  if (rc) return rc == 1 ? 0 : rc;
  if ((p.ldg & 3) || (p.lddx & 3) || (p.has_res && ((p.ldo & 3) || (p.dres && (p.lddr & 3)))) || (p.red && p.red_bs != 0 && p.red_bs != p.C))
    return CRAFT_ERR_ALIGN;
  const long n = (long)p.B * p.N * (p.C >> 2);
  hipLaunchKernelGGL(k_norm_act_bwd_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Stem weight gradient: the 7x7 / stride-2 / 3-channel convolution's input patches as a matrix,
//   cols[p][(ky*7 + kx)*3 + c] = 2 * image[b][c][2*oy + ky - 3][2*ox + kx - 3] / 255 - 1   (0 outside the image, and for k >= 147)
// with p = (b, oy, ox) and ld = 160, so that dW = dY^T . cols is one k-major x k-major craft_gemm (K = all output pixels).
// ---------------------------------------------------------------------------------------------
__global__ void k_stem_im2col(const float* __restrict__ img, int B, int H, int W, float* __restrict__ cols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int Ho = H / 2, Wo = W / 2;
  if (i >= (long)B * Ho * Wo * 160) return;
  const long pix = i / 160;
  const int k = (int)(i - pix * 160);
  float v = 0.f;
  if (k < 147) {
    const int tap = k / 3, c = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
    const int b = (int)(pix / ((long)Ho * Wo));
    const int rem = (int)(pix - (long)b * Ho * Wo), oy = rem / Wo, ox = rem - oy * Wo;
    const int y = 2 * oy + ky - 3, x = 2 * ox + kx - 3;
    if (y >= 0 && y < H && x >= 0 && x < W) v = 2.f * (img[(((long)b * 3 + c) * H + y) * W + x] / 255.f) - 1.f;
  }
  cols[i] = v;
}
int launch_stem_im2col(const float* img, int B, int H, int W, float* cols, hipStream_t s) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  if ((H & 1) || (W & 1)) return CRAFT_ERR_ALIGN;
  const long n = (long)B * (H / 2) * (W / 2) * 160;
  hipLaunchKernelGGL(k_stem_im2col, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, img, B, H, W, cols);
  return (int)hipGetLastError();
}

// zero-stuffing of a stride-2 layer's output gradient: g [B][Ho*Wo][C] -> gf [B][Hin*Win][C], gf(2oy, 2ox) = g(oy, ox), 0 elsewhere.
// The backward of a stride-2 convolution is then the backward of the stride-1 convolution it subsamples.
__global__ void k_zero_stuff2(const float* __restrict__ g, long ldg, int B, int Hin, int Win, int C, float* __restrict__ gf, long ldf) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C >> 2;
  if (i >= (long)B * Hin * Win * c4) return;
  const long row = i / c4;
  const int c = (int)(i - row * c4) * 4;
  const int b = (int)(row / ((long)Hin * Win));
  const int rem = (int)(row - (long)b * Hin * Win), y = rem / Win, x = rem - y * Win;
  float4 v = {0.f, 0.f, 0.f, 0.f};
  const int Ho = Hin / 2, Wo = Win / 2;
  if (!(y & 1) && !(x & 1) && (y >> 1) < Ho && (x >> 1) < Wo)
    v = *reinterpret_cast<const float4*>(g + (((long)b * Ho + (y >> 1)) * Wo + (x >> 1)) * ldg + c);
  *reinterpret_cast<float4*>(gf + row * ldf + c) = v;
}
int launch_zero_stuff2(const float* g, long ldg, int B, int Hin, int Win, int C, float* gf, long ldf, hipStream_t s) {
  if (B <= 0 || Hin <= 0 || Win <= 0 || C <= 0) return 0;
  if ((C & 3) || (ldg & 3) || (ldf & 3) || (Hin & 1) || (Win & 1)) return CRAFT_ERR_ALIGN;
  const long n = (long)B * Hin * Win * (C >> 2);
  hipLaunchKernelGGL(k_zero_stuff2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, ldg, B, Hin, Win, C, gf, ldf);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Small per-channel finalisers (one thread per channel; they replace a dozen tiny tensor ops per layer).
// craft_bn_finalize: BatchNorm2d statistics for craft_norm_act_fwd.  stats != NULL (training): the conv epilogue's
//   [CRAFT_STATS_REPLICAS][B][C][2] (sum, sum^2) -> batch mean / biased variance over n = B * count samples -> mean_rstd [C][2], and
//   running_mean / running_var (momentum update with the UNBIASED variance, nn.BatchNorm2d semantics).  stats == NULL (eval /
//   freeze_bn): mean_rstd from the running statistics.
// craft_norm_bwd_finalize: sums [B][C][2] of craft_norm_act_bwd_reduce -> red = population means for craft_norm_act_bwd_apply
//   ([B][C][2] per image, [C][2] over the batch; skipped when population == 0) and dgamma / dbeta (batch sums; may be NULL).
// ---------------------------------------------------------------------------------------------
// one WAVE per channel: the 64 lanes split the CRAFT_STATS_REPLICAS x B partial sums (a thread per channel walked 1 024 dependent
// double loads: 124 us per call, 15 calls per training step)
__global__ __launch_bounds__(256) void k_bn_finalize(const double* __restrict__ stats, int B, int C, double count, float eps, float momentum,
                                                     float* __restrict__ mr, float* __restrict__ rmean, float* __restrict__ rvar) {
  const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  if (!stats) {
    if (lane == 0) {
      mr[2 * c] = rmean[c];
      mr[2 * c + 1] = (float)(1.0 / sqrt((double)rvar[c] + (double)eps));
    }
    return;
  }
  double s0 = 0.0, s1 = 0.0;
  for (int i = lane; i < CRAFT_STATS_REPLICAS * B; i += 64) { s0 += stats[((long)i * C + c) * 2]; s1 += stats[((long)i * C + c) * 2 + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
  if (lane) return;
  const double n = (double)B * count, mean = s0 / n, var = fmax(s1 / n - mean * mean, 0.0);
  mr[2 * c] = (float)mean;
  mr[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * (n / fmax(n - 1.0, 1.0)));
}
int launch_bn_finalize(const double* stats, int B, int C, double count, float eps, float momentum, float* mr, float* rmean, float* rvar,
                       hipStream_t s) {
  if (C <= 0) return 0;
  if (!stats && (!rmean || !rvar)) return CRAFT_ERR_ARG;
  hipLaunchKernelGGL(k_bn_finalize, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, stats, B, C, count, eps, momentum, mr, rmean, rvar);
  return (int)hipGetLastError();
}

__global__ void k_norm_bwd_finalize(const double* __restrict__ sums, int B, int C, double population, int per_image,
                                    float* __restrict__ red, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double t0 = 0.0, t1 = 0.0;
  for (int b = 0; b < B; ++b) {
    const double a0 = sums[((long)b * C + c) * 2], a1 = sums[((long)b * C + c) * 2 + 1];
    t0 += a0; t1 += a1;
    if (red && per_image && population > 0.0) { red[((long)b * C + c) * 2] = (float)(a0 / population); red[((long)b * C + c) * 2 + 1] = (float)(a1 / population); }
  }
  if (red && !per_image && population > 0.0) { red[2 * c] = (float)(t0 / population); red[2 * c + 1] = (float)(t1 / population); }
  if (dbeta) dbeta[c] = (float)t0;
  if (dgamma) dgamma[c] = (float)t1;
}
int launch_norm_bwd_finalize(const double* sums, int B, int C, double population, int per_image, float* red, float* dgamma, float* dbeta,
                             hipStream_t s) {
  if (C <= 0 || B <= 0) return 0;
  hipLaunchKernelGGL(k_norm_bwd_finalize, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, sums, B, C, population, per_image, red, dgamma, dbeta);
  return (int)hipGetLastError();
}

}  // namespace craft
