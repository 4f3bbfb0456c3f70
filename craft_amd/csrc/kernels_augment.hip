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

__device__ __forceinline__ float gray_of(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }   // PIL "L"
__device__ __forceinline__ float q255(float v) { return fminf(fmaxf(rintf(v), 0.f), 255.f); }

// op: 0 brightness (blend with black), 1 contrast (blend with the mean grey `mean`), 2 saturation (blend with the pixel's grey),
// 3 hue (shift of the HSV hue by `factor` turns, PIL's integer HSV).  img [npix][3], in place.
__global__ void k_aug_photo(float* __restrict__ img, long npix, int op, float factor, float mean) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  float r = img[3 * i], g = img[3 * i + 1], b = img[3 * i + 2];
  if (op == 0) {
    r = q255(r * factor); g = q255(g * factor); b = q255(b * factor);
  } else if (op == 1) {
    r = q255((r - mean) * factor + mean); g = q255((g - mean) * factor + mean); b = q255((b - mean) * factor + mean);
  } else if (op == 2) {
    const float gr = q255(gray_of(r, g, b));
    r = q255((r - gr) * factor + gr); g = q255((g - gr) * factor + gr); b = q255((b - gr) * factor + gr);
  } else {
    // RGB -> HSV (h in [0,1)), h += factor, HSV -> RGB; 8-bit channels like PIL's 'HSV' mode
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    const float d = mx - mn;
    float h = 0.f;
    if (d > 0.f) {
      if (mx == r) h = (g - b) / d; else if (mx == g) h = 2.f + (b - r) / d; else h = 4.f + (r - g) / d;
      h /= 6.f;
      h -= floorf(h);
    }
    const float s = mx > 0.f ? d / mx : 0.f, v = mx;
    float h8 = rintf(h * 255.f) + rintf(factor * 255.f);          // uint8 wrap-around of the hue channel
    h8 -= 256.f * floorf(h8 / 256.f);
    const float hh = h8 / 255.f * 6.f;
    const int sect = (int)floorf(hh) % 6;
    const float f = hh - floorf(hh);
    const float s8 = rintf(s * 255.f) / 255.f;
    const float p = v * (1.f - s8), q = v * (1.f - s8 * f), t = v * (1.f - s8 * (1.f - f));
    switch (sect) {
      case 0: r = v; g = t; b = p; break;
      case 1: r = q; g = v; b = p; break;
      case 2: r = p; g = v; b = t; break;
      case 3: r = p; g = q; b = v; break;
      case 4: r = t; g = p; b = v; break;
      default: r = v; g = p; b = q; break;
    }
    r = q255(r); g = q255(g); b = q255(b);
  }
  img[3 * i] = r; img[3 * i + 1] = g; img[3 * i + 2] = b;
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
