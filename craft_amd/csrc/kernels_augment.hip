// Input pipeline on the GPU (reference core/utils/augmentor.py): the per-sample augmentation of the
// trainers -- photometric jitter, eraser, random scale / stretch / flip / crop, and the random-shift augmentation -- as a few
// gather / element-wise kernels over HWC images that stay in HBM.  The host (craft_amd/augment.py) draws the random parameters
// in the reference's order; these kernels apply them.
//   * k_aug_spatial : resize (cv2.INTER_LINEAR convention: src = (dst + 0.5) / f - 0.5, replicate border) -> h-flip -> v-flip ->
//                     crop as ONE gather (augmentor.py:148-193): only the crop is ever produced.  Images are rounded back to
//                     integer levels (the reference works on uint8 arrays), the flow is scaled by (fx, fy) and sign-flipped.
//   * k_aug_photo   : one ColorJitter step (torchvision semantics on a uint8 image, augmentor.py:106-123): brightness /
//                     contrast / saturation / hue, result rounded to integer levels.
//   * k_aug_erase   : eraser rectangles filled with the mean colour (augmentor.py:125-139).
//   * k_aug_blur    : cv2.GaussianBlur((K, K), sigma) of the cropped frames (augmentor.py:195-198): separable Gaussian weights
//                     exp(-(i - (K-1)/2)^2 / (2 sigma^2)) normalised to 1 (cv2.getGaussianKernel for sigma > 0), BORDER_REFLECT_101,
//                     applied as ONE K x K gather per output value (rows of a 368 x 496 crop: the whole image is L2-resident), rounded
//                     to integer levels.
//   * k_aug_shift   : random_shift (augmentor.py:16-78): crop both frames against each other by (dx, dy), subtract the shift from
//                     the flow, zero-pad back to the input size, emit the valid mask.
#include "launch.hpp"

namespace craft {

__device__ __forceinline__ float bilinear_hwc(const float* __restrict__ src, int H, int W, int C, float sy, float sx, int c) {
  const float fy = floorf(sy), fx = floorf(sx);
  const float wy = sy - fy, wx = sx - fx;
  const int y0 = min(max((int)fy, 0), H - 1), y1 = min(max((int)fy + 1, 0), H - 1);
  const int x0 = min(max((int)fx, 0), W - 1), x1 = min(max((int)fx + 1, 0), W - 1);
  const float a = src[((long)y0 * W + x0) * C + c], b = src[((long)y0 * W + x1) * C + c];
  const float d = src[((long)y1 * W + x0) * C + c], e = src[((long)y1 * W + x1) * C + c];
  return (a * (1.f - wx) + b * wx) * (1.f - wy) + (d * (1.f - wx) + e * wx) * wy;
}

// out [ch][cw][C]; the scaled image has size (Hs, Ws) = (round(H*fy), round(W*fx)) when do_resize, else (H, W)
__global__ void k_aug_spatial(const float* __restrict__ src, int H, int W, int C, int do_resize, float fx, float fy, int Hs, int Ws,
                              int hflip, int vflip, int y0, int x0, int ch, int cw, int is_flow, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)ch * cw * C) return;
  const int c = (int)(i % C);
  const long p = i / C;
  const int x = (int)(p % cw), y = (int)(p / cw);
  const int ys = vflip ? Hs - 1 - (y0 + y) : y0 + y;
  const int xs = hflip ? Ws - 1 - (x0 + x) : x0 + x;
  float v;
  if (do_resize) {
    const float sy = ((float)ys + 0.5f) / fy - 0.5f, sx = ((float)xs + 0.5f) / fx - 0.5f;
    v = bilinear_hwc(src, H, W, C, sy, sx, c);
  } else {
    v = src[((long)ys * W + xs) * C + c];
  }
  if (is_flow) {
    if (do_resize) v *= (c == 0 ? fx : fy);
    if ((c == 0 && hflip) || (c == 1 && vflip)) v = -v;
  } else if (do_resize) {
    v = fminf(fmaxf(rintf(v), 0.f), 255.f);
  }
  out[i] = v;
}

__device__ __forceinline__ float q255(float v) { return fminf(fmaxf(rintf(v), 0.f), 255.f); }

// ---- ColorJitter on 8-bit images = Pillow's arithmetic (torchvision's PIL path wraps ImageEnhance and the "HSV" mode; augmentor.py:104,
// :111-123).  Restated from libImaging (Blend.c ImagingBlend, Convert.c rgb2l / rgb2hsv_row / hsv2rgb) and pinned bit for bit by
// tests/golden/photo_pil.npz, which Pillow itself produced (tools/make_golden_photo.py; oracle/augment_oracle.py is the numpy form).
__device__ __forceinline__ int gray_L(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }   // Convert.c rgb2l
// ImagingBlend(degenerate d, image i, alpha): float d + alpha * (i - d), product and sum rounded separately (no FMA contraction: Pillow is
// built for baseline x86-64); alpha inside [0, 1]: (UINT8) truncation, outside: clipped to [0, 255] first
__device__ __forceinline__ float blend8(int d, int i, float alpha, bool inside) {
#pragma clang fp contract(off)
  const float prod = alpha * (float)(i - d);
  const float t = (float)d + prod;
  if (inside) return (float)(int)t;
  return t <= 0.f ? 0.f : (t >= 255.f ? 255.f : (float)(int)t);
}

// op: 0 brightness (blend with black), 1 contrast (blend with the grey level `aux` = int(mean of the "L" image + 0.5), which the caller
// reduces), 2 saturation (blend with the pixel's own "L"), 3 hue (8-bit HSV round trip, hue + `aux` mod 256 where aux = the integer shift
// int32(hue_factor * 255) & 255 that torchvision adds to the uint8 hue plane).  img [npix][3] integer levels stored as float, in place.
__global__ void k_aug_photo(float* __restrict__ img, long npix, int op, float factor, float aux) {
#pragma clang fp contract(off)                     // (every product and sum rounded on its own, like the host library)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const int r = (int)rintf(img[3 * i]), g = (int)rintf(img[3 * i + 1]), b = (int)rintf(img[3 * i + 2]);
  float ro, go, bo;
  const bool inside = factor >= 0.f && factor <= 1.f;
  if (op == 0) {
    ro = blend8(0, r, factor, inside); go = blend8(0, g, factor, inside); bo = blend8(0, b, factor, inside);
  } else if (op == 1) {
    const int m = (int)aux;
    ro = blend8(m, r, factor, inside); go = blend8(m, g, factor, inside); bo = blend8(m, b, factor, inside);
  } else if (op == 2) {
    const int gr = gray_L(r, g, b);
    ro = blend8(gr, r, factor, inside); go = blend8(gr, g, factor, inside); bo = blend8(gr, b, factor, inside);
  } else {
    // Convert.c rgb2hsv_row: float quotients, sums with the double literals in double, h = fmod(h / 6.0 + 1.0, 1.0) stored to float,
    // (int)(x * 255.0) truncation
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    int uh = 0, us = 0;
    const int uv = maxc;
    if (minc != maxc) {
      const float cr = (float)(maxc - minc);
      const float s = cr / (float)maxc;
      const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
      float h;
      if (r == maxc) h = bc - gc;
      else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
      else h = (float)(4.0 + (double)gc - (double)rc);
      h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
      uh = min(max((int)((double)h * 255.0), 0), 255);
      us = min(max((int)((double)s * 255.0), 0), 255);
    }
    uh = (uh + (int)aux) & 255;                      // the uint8 hue plane wraps around
    if (us == 0) {
      ro = go = bo = (float)uv;
    } else {
      // Convert.c hsv2rgb: i = floor(h * 6.0 / 255.0); f, fs stored to float; p / q / t = round(...) in double (arguments >= 0)
      const double hh = (double)(float)uh * 6.0 / 255.0;
      const int sect = (int)floor(hh);
      const float f = (float)(hh - (double)(float)sect);
      const float fs = (float)((double)(float)us / 255.0);
      const double v = (double)uv;
      const int pp = min(max((int)floor(v * (1.0 - (double)fs) + 0.5), 0), 255);
      const int qq = min(max((int)floor(v * (1.0 - (double)fs * (double)f) + 0.5), 0), 255);
      const int tt = min(max((int)floor(v * (1.0 - (double)fs * (1.0 - (double)f)) + 0.5), 0), 255);
      int R, G, B;
      switch (sect % 6) {
        case 0: R = uv; G = tt; B = pp; break;
        case 1: R = qq; G = uv; B = pp; break;
        case 2: R = pp; G = uv; B = tt; break;
        case 3: R = pp; G = qq; B = uv; break;
        case 4: R = tt; G = pp; B = uv; break;
        default: R = uv; G = pp; B = qq; break;
      }
      ro = (float)R; go = (float)G; bo = (float)B;
    }
  }
  img[3 * i] = ro; img[3 * i + 1] = go; img[3 * i + 2] = bo;
}

// rects [n][4] = (x0, y0, dx, dy); img [H][W][3]
__global__ void k_aug_erase(float* __restrict__ img, int H, int W, const int* __restrict__ rects, int nrect, float mr, float mg, float mb) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)H * W) return;
  const int x = (int)(i % W), y = (int)(i / W);
  bool hit = false;
  for (int k = 0; k < nrect; ++k) {
    const int x0 = rects[4 * k], y0 = rects[4 * k + 1];
    hit |= x >= x0 && x < x0 + rects[4 * k + 2] && y >= y0 && y < y0 + rects[4 * k + 3];
  }
  if (hit) { img[3 * i] = mr; img[3 * i + 1] = mg; img[3 * i + 2] = mb; }
}

// random_shift (augmentor.py:16-78) for even (dx, dy): frame 1 keeps rows [T1, T1 + h), cols [L1, L1 + w), frame 2 [T2..), [L2..) with
// h = H - |dy|, w = W - |dx|; both land at offset (|dy|/2, |dx|/2) of a zero image of the input size; flow -= (dx, dy).
__global__ void k_aug_shift(const float* __restrict__ img1, const float* __restrict__ img2, const float* __restrict__ flow, int H, int W,
                            int dx, int dy, float* __restrict__ o1, float* __restrict__ o2, float* __restrict__ oflow,
                            float* __restrict__ valid) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)H * W) return;
  const int x = (int)(i % W), y = (int)(i / W);
  const int ax = abs(dx), ay = abs(dy), h = H - ay, w = W - ax;
  const int T1 = dy >= 0 ? 0 : -dy, L1 = dx >= 0 ? 0 : -dx, T2 = dy >= 0 ? dy : 0, L2 = dx >= 0 ? dx : 0;
  const int yy = y - ay / 2, xx = x - ax / 2;
  const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;
  const long s1 = in ? (long)(T1 + yy) * W + L1 + xx : 0, s2 = in ? (long)(T2 + yy) * W + L2 + xx : 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) { o1[3 * i + c] = in ? img1[3 * s1 + c] : 0.f; o2[3 * i + c] = in ? img2[3 * s2 + c] : 0.f; }
  oflow[2 * i] = in ? flow[2 * s1] - (float)dx : 0.f;
  oflow[2 * i + 1] = in ? flow[2 * s1 + 1] - (float)dy : 0.f;
  valid[i] = in ? 1.f : 0.f;
}

// SparseFlowAugmentor.resize_sparse_flow_map (augmentor.py:249-281): every valid source pixel (x, y) lands on (round(x fx),
// round(y fy)) if that is strictly inside (0, wd1) x (0, ht1); numpy's fancy assignment lets the LAST source pixel (row-major
// order) win a contested target.  Pass 1 records the largest source index per target (atomicMax), pass 2 gathers: deterministic
// and identical to the reference's order.  Then flips / crop like the dense path (no interpolation: values are moved, not mixed).
__global__ void k_aug_sparse_claim(const float* __restrict__ valid, int H, int W, float fx, float fy, int Hs, int Ws, int* __restrict__ owner) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)H * W) return;
  if (valid[i] < 1.f) return;
  const int x = (int)(i % W), y = (int)(i / W);
  const int xx = (int)rintf((float)x * fx), yy = (int)rintf((float)y * fy);
  if (xx > 0 && xx < Ws && yy > 0 && yy < Hs) atomicMax(&owner[(long)yy * Ws + xx], (int)i);
}
__global__ void k_aug_sparse_gather(const float* __restrict__ flow, const int* __restrict__ owner, int Hs, int Ws, float fx, float fy,
                                    int hflip, int y0, int x0, int ch, int cw, float* __restrict__ oflow, float* __restrict__ ovalid) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)ch * cw) return;
  const int x = (int)(i % cw), y = (int)(i / cw);
  const int ys = y0 + y, xs = hflip ? Ws - 1 - (x0 + x) : x0 + x;
  const int src = owner[(long)ys * Ws + xs];
  float u = 0.f, v = 0.f, ok = 0.f;
  if (src >= 0) { u = flow[2 * (long)src] * fx; v = flow[2 * (long)src + 1] * fy; ok = 1.f; if (hflip) u = -u; }
  oflow[2 * i] = u; oflow[2 * i + 1] = v; ovalid[i] = ok;
}

// Gaussian blur of an HWC image, K x K taps (K odd, <= 31), cv2's default border (BORDER_REFLECT_101: ... 2 1 | 0 1 2 ... ).  The row pass
// is accumulated first for each tap row (cv2's separable order: horizontal filter, then vertical), in float.
struct BlurTaps { float w[32]; };
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}
__global__ void k_aug_blur(const float* __restrict__ src, int H, int W, int C, int K, BlurTaps t, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)H * W * C) return;
  const int c = (int)(i % C);
  const long p = i / C;
  const int x = (int)(p % W), y = (int)(p / W), r = K / 2;
  float acc = 0.f;
  for (int dy = 0; dy < K; ++dy) {
    const float* row = src + (long)reflect101(y + dy - r, H) * W * C + c;
    float a = 0.f;
    for (int dx = 0; dx < K; ++dx) a += t.w[dx] * row[(long)reflect101(x + dx - r, W) * C];
    acc += t.w[dy] * a;
  }
  out[i] = q255(acc);
}

#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, s
int launch_aug_sparse(const float* flow, const float* valid, int H, int W, float fx, float fy, int hflip, int y0, int x0, int ch, int cw,
                      int* owner, float* oflow, float* ovalid, hipStream_t s) {
  const int Hs = (int)rint((double)H * fy), Ws = (int)rint((double)W * fx);
  if (y0 < 0 || x0 < 0 || y0 + ch > Hs || x0 + cw > Ws) return CRAFT_ERR_ARG;
  hipError_t e = hipMemsetAsync(owner, 0xff, sizeof(int) * (size_t)Hs * Ws, s);       // -1
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_aug_sparse_claim, GRID1((long)H * W), valid, H, W, fx, fy, Hs, Ws, owner);
  hipLaunchKernelGGL(k_aug_sparse_gather, GRID1((long)ch * cw), flow, owner, Hs, Ws, fx, fy, hflip, y0, x0, ch, cw, oflow, ovalid);
  return (int)hipGetLastError();
}
int launch_aug_spatial(const float* src, int H, int W, int C, int do_resize, float fx, float fy, int hflip, int vflip, int y0, int x0, int ch,
                       int cw, int is_flow, float* out, hipStream_t s) {
  if (H <= 0 || W <= 0 || ch <= 0 || cw <= 0 || C <= 0) return CRAFT_ERR_ARG;
  const int Hs = do_resize ? (int)rint((double)H * fy) : H, Ws = do_resize ? (int)rint((double)W * fx) : W;
  if (y0 < 0 || x0 < 0 || y0 + ch > Hs || x0 + cw > Ws) return CRAFT_ERR_ARG;
  if (is_flow && C != 2) return CRAFT_ERR_ARG;
  hipLaunchKernelGGL(k_aug_spatial, GRID1((long)ch * cw * C), src, H, W, C, do_resize, fx, fy, Hs, Ws, hflip, vflip, y0, x0, ch, cw, is_flow, out);
  return (int)hipGetLastError();
}
int launch_aug_photo(float* img, long npix, int op, float factor, float mean, hipStream_t s) {
  if (npix <= 0) return 0;
  if (op < 0 || op > 3) return CRAFT_ERR_ARG;
  hipLaunchKernelGGL(k_aug_photo, GRID1(npix), img, npix, op, factor, mean);
  return (int)hipGetLastError();
}
int launch_aug_erase(float* img, int H, int W, const int* rects, int nrect, float mr, float mg, float mb, hipStream_t s) {
  if (nrect <= 0) return 0;
  hipLaunchKernelGGL(k_aug_erase, GRID1((long)H * W), img, H, W, rects, nrect, mr, mg, mb);
  return (int)hipGetLastError();
}
int launch_aug_blur(const float* src, int H, int W, int C, int K, float sigma, float* out, hipStream_t s) {
  if (H <= 0 || W <= 0 || C <= 0 || !src || !out || src == out) return CRAFT_ERR_ARG;
  if (K < 1 || !(K & 1) || !(sigma > 0.f)) return CRAFT_ERR_ARG;
  if (K > 31) return CRAFT_ERR_UNSUPPORTED;
  BlurTaps t = {};
  double sum = 0.0, w[32];
  for (int i = 0; i < K; ++i) { const double x = i - 0.5 * (K - 1); w[i] = exp(-x * x / (2.0 * (double)sigma * (double)sigma)); sum += w[i]; }
  for (int i = 0; i < K; ++i) t.w[i] = (float)(w[i] / sum);
  hipLaunchKernelGGL(k_aug_blur, GRID1((long)H * W * C), src, H, W, C, K, t, out);
  return (int)hipGetLastError();
}
int launch_aug_shift(const float* img1, const float* img2, const float* flow, int H, int W, int dx, int dy, float* o1, float* o2, float* oflow,
                     float* valid, hipStream_t s) {
  if ((dx & 1) || (dy & 1) || abs(dx) >= W || abs(dy) >= H) return CRAFT_ERR_ARG;
  hipLaunchKernelGGL(k_aug_shift, GRID1((long)H * W), img1, img2, flow, H, W, dx, dy, o1, o2, oflow, valid);
  return (int)hipGetLastError();
}

}  // namespace craft
