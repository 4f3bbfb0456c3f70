// General strided batched GEMM and the convolution weight gradient: the contractions of the BACKWARD pass
// (include/craft_hip.h, "training" section).  Same MFMA engine as the forward kernels (gemm_engine.hpp: 128 x BN block
// tiles, 4 waves, K-tile 32, double-buffered LDS, fp32 / bf16 / fp16 / f16x3 operand modes), with one addition: an operand
// may be stored K-MAJOR (element (row, k) at base[k*ld + row]) -- a transposed operand read in place instead of through a
// transposed copy.  The backward pass is made of exactly such products:
//     dX = dY . W          A = dY rows,        B = W  k-major        (nn.Linear input gradient)
//     dW = dY^T . X        A = dY k-major,     B = X  k-major        (weight gradients: K = rows, split over blocks)
//     dP = dO . V^T        A = dO rows,        B = V  rows
//     dV = P^T . dO        A = P  k-major,     B = dO k-major
//     dQ = dS . K, dK = dS^T . Q               likewise
// A k-major tile is fetched as float4s ALONG THE ROWS (coalesced: 8 lanes x 16 B per k) and scattered into the [row][k]
// LDS layout the fragment reads expect.
#include "launch.hpp"

namespace craft {

// ---------------------------------------------------------------------------------------------
// k-major staging registers: thread -> k = tid >> 3 (0..31), row chunks (tid & 7) + 8*i, 4 rows each
// ---------------------------------------------------------------------------------------------
template <int ROWS> struct RegsF32T { float4 v[ROWS / 32]; unsigned zmask; };

template <int PREC, int ROWS>
__device__ __forceinline__ void stage_store_piece(typename PrecT<PREC>::lds_t* S, const RegsF32T<ROWS>& r, int tid, int i) {
  constexpr int LD = PrecT<PREC>::LD;
  typedef typename PrecT<PREC>::lds_t lds_t;
  const int k = tid >> 3, row = ((tid & 7) + 8 * i) * 4;
  const bool z = (r.zmask >> i) & 1u;
  float4 v;
  v.x = z ? 0.f : r.v[i].x; v.y = z ? 0.f : r.v[i].y; v.z = z ? 0.f : r.v[i].z; v.w = z ? 0.f : r.v[i].w;
  if constexpr (PREC == CRAFT_PREC_F16X3) {
    f16x4 h, l;
    split_f16x3(v, h, l);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      S[(row + j) * LD + k] = h[j];
      S[(ROWS + row + j) * LD + k] = l[j];
    }
  } else {
    S[(row + 0) * LD + k] = (lds_t)v.x;
    S[(row + 1) * LD + k] = (lds_t)v.y;
    S[(row + 2) * LD + k] = (lds_t)v.z;
    S[(row + 3) * LD + k] = (lds_t)v.w;
  }
}
template <int ROWS> __device__ __forceinline__ constexpr int stage_pieces(const RegsF32T<ROWS>&) { return ROWS / 32; }
template <int PREC, int ROWS>
__device__ __forceinline__ void stage_store(typename PrecT<PREC>::lds_t* S, const RegsF32T<ROWS>& r, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) stage_store_piece<PREC, ROWS>(S, r, tid, i);
}

// K-major fp32 operand: element (row, k) at base[k*ld + row], k in [k0, k1) (split-K range), rows >= nrows and k >= k1 read
// as zero.  ld % 4 == 0, base 16-B aligned.  A chunk of 4 rows that straddles nrows is loaded element-wise from clamped
// addresses (its out-of-range rows only reach output rows / columns the epilogue drops).
template <int ROWS> struct LoaderColsF32 {
  typedef RegsF32T<ROWS> Regs;
  const float* base;
  long ld;
  int row[ROWS / 32];
  int nrows, k0, k1, kk;
  __device__ __forceinline__ void init(const float* base_, long ld_, int row0, int nrows_, int k0_, int k1_, int tid) {
    base = base_; ld = ld_; nrows = nrows_; k0 = k0_; k1 = k1_;
    kk = tid >> 3;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) row[i] = row0 + ((tid & 7) + 8 * i) * 4;
  }
  __device__ __forceinline__ void fetch(int kt, Regs& r) const {
    const int k = k0 + kt * BK + kk;
    const bool kok = k < k1;
    const float* p = base + (long)(kok ? k : k1 - 1) * ld;
    unsigned zm = kok ? 0u : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int r0 = row[i];
      if (r0 + 4 <= nrows) {
        r.v[i] = *reinterpret_cast<const float4*>(p + r0);
      } else {
        float4 t;
        t.x = p[min(r0, nrows - 1)]; t.y = p[min(r0 + 1, nrows - 1)]; t.z = p[min(r0 + 2, nrows - 1)]; t.w = p[min(r0 + 3, nrows - 1)];
        r.v[i] = t;
      }
    }
    r.zmask = zm;
  }
};

// Rows loader restricted to a k range (split-K): a thin wrapper over LoaderRowsF32 semantics.
template <int ROWS> struct LoaderRowsRangeF32 {
  typedef RegsF32<ROWS> Regs;
  const float* p[ROWS / 32];
  int k0, k1, kcol;
  __device__ __forceinline__ void init(const float* base, long ld, int row0, int nrows, int k0_, int k1_, int tid) {
    k0 = k0_; k1 = k1_;
    kcol = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) p[i] = base + (long)min(row0 + (tid >> 3) + 32 * i, nrows - 1) * ld;
  }
  __device__ __forceinline__ void fetch(int kt, Regs& r) const {
    const int k = k0 + kt * BK + kcol;
    const bool kok = k < k1;                      // K ranges are multiples of 4 (checked by the launcher)
    const int kc = kok ? k : k1 - 4;
    r.zmask = kok ? 0u : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) r.v[i] = *reinterpret_cast<const float4*>(p[i] + kc);
  }
};

struct GenGemmParams {
  const float* A; const float* B; float* C;
  long a_sm, a_sk, a_bs0, a_bs1;       // A(z, m, k) at A + z0*a_bs0 + z1*a_bs1 + m*a_sm + k*a_sk; one of a_sm / a_sk is 1
  long b_sn, b_sk, b_bs0, b_bs1;
  long ldc, c_bs0, c_bs1;
  int zdiv, batch, M, N, K;
  float alpha;
  int accumulate;                      // 1: C += (plain read-modify-write when ksplit == 1, atomics otherwise)
  int ksplit, kchunk;                  // K range per block = kchunk (multiple of 32)
};

template <int PREC, int BN, bool AT, bool BT>
__global__ __launch_bounds__(NTHREADS) void k_gemm_gen(GenGemmParams p) {
  constexpr int BM = 128, WM = 2, WN = 2, MT = BM / WM / 32, NT = BN / WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int z = blockIdx.z / p.ksplit, ks = blockIdx.z - z * p.ksplit;
  const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
  const int k0 = ks * p.kchunk, k1 = min(p.K, k0 + p.kchunk);
  if (k0 >= k1) return;
  const float* A = p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
  const float* B = p.B + z0 * p.b_bs0 + z1 * p.b_bs1;
  f32x16 acc[MT][NT];
  acc_zero(acc);
  const int nk = (k1 - k0 + BK - 1) / BK;
  typename std::conditional<AT, LoaderColsF32<BM>, LoaderRowsRangeF32<BM>>::type la;
  typename std::conditional<BT, LoaderColsF32<BN>, LoaderRowsRangeF32<BN>>::type lb;
  la.init(A, AT ? p.a_sk : p.a_sm, m0, p.M, k0, k1, tid);
  lb.init(B, BT ? p.b_sk : p.b_sn, n0, p.N, k0, k1, tid);
  gemm_mainloop<PREC, BM, BN, WM, WN>(la, lb, nk, acc, NoFold());
  float* C = p.C + z0 * p.c_bs0 + z1 * p.c_bs1;
  const int rb = m0 + (wave / WN) * (BM / WM), cb = n0 + (wave % WN) * (BN / WN);
  const bool atomic = p.ksplit > 1;
  acc_foreach(acc, lane, [&](int r, int c, float v, int, int, int) {
    const int row = rb + r, col = cb + c;
    if (row < p.M && col < p.N) {
      float* d = C + (long)row * p.ldc + col;
      const float t = v * p.alpha;
      if (atomic) unsafeAtomicAdd(d, t);
      else *d = p.accumulate ? *d + t : t;
    }
  });
}

template <int PREC, bool AT, bool BT> static int launch_gen_t(const GenGemmParams& p, hipStream_t s) {
  const int bn = (p.N % 128 == 0 || p.N > 96) ? 128 : 64;
  dim3 grid((p.M + 127) / 128, (p.N + bn - 1) / bn, p.batch * p.ksplit);
  if (bn == 128) hipLaunchKernelGGL((k_gemm_gen<PREC, 128, AT, BT>), grid, dim3(NTHREADS), 0, s, p);
  else hipLaunchKernelGGL((k_gemm_gen<PREC, 64, AT, BT>), grid, dim3(NTHREADS), 0, s, p);
  return (int)hipGetLastError();
}

int launch_gemm_gen(const float* A, long a_sm, long a_sk, long a_bs0, long a_bs1, const float* B, long b_sn, long b_sk, long b_bs0,
                    long b_bs1, float* C, long ldc, long c_bs0, long c_bs1, int zdiv, int batch, int M, int N, int K, float alpha,
                    int accumulate, int ksplit, int prec, hipStream_t s) {
  if (M <= 0 || N <= 0 || batch <= 0) return 0;
  if (K <= 0) return accumulate ? 0 : CRAFT_ERR_ARG;
  if (zdiv <= 0 || ksplit < 0) return CRAFT_ERR_ARG;
  const bool at = a_sm == 1 && a_sk != 1, bt = b_sn == 1 && b_sk != 1;
  if (!at && a_sk != 1) return CRAFT_ERR_ARG;
  if (!bt && b_sk != 1) return CRAFT_ERR_ARG;
  // 16-byte vector loads: leading dimensions and batch strides in multiples of 4 floats, bases 16-B aligned
  const long lda = at ? a_sk : a_sm, ldb = bt ? b_sk : b_sn;
  if ((lda & 3) || (ldb & 3) || (a_bs0 & 3) || (a_bs1 & 3) || (b_bs0 & 3) || (b_bs1 & 3)) return CRAFT_ERR_ALIGN;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return CRAFT_ERR_ALIGN;
  if ((!at || !bt) && (K & 3)) return CRAFT_ERR_ALIGN;       // a k-contiguous operand is read 4 k at a time
  GenGemmParams p = {};
  p.A = A; p.B = B; p.C = C;
  p.a_sm = a_sm; p.a_sk = a_sk; p.a_bs0 = a_bs0; p.a_bs1 = a_bs1;
  p.b_sn = b_sn; p.b_sk = b_sk; p.b_bs0 = b_bs0; p.b_bs1 = b_bs1;
  p.ldc = ldc; p.c_bs0 = c_bs0; p.c_bs1 = c_bs1;
  p.zdiv = zdiv; p.batch = batch; p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.accumulate = accumulate;
  if (ksplit == 0) {      // auto: enough blocks to fill 256 CUs twice, K chunks of at least 256
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128) * batch;
    long want = (512 + tiles - 1) / tiles;
    const long maxs = (K + 255) / 256;
    ksplit = (int)(want < 1 ? 1 : (want > maxs ? maxs : want));
  }
  if (ksplit > 1 && !accumulate) return CRAFT_ERR_ARG;      // split-K adds into C: the caller zero-fills (or accumulates)
  p.ksplit = ksplit;
  p.kchunk = (((K + ksplit - 1) / ksplit) + BK - 1) / BK * BK;
#define GO(PR) do { if (at && bt) return launch_gen_t<PR, true, true>(p, s); if (at) return launch_gen_t<PR, true, false>(p, s); \
                    if (bt) return launch_gen_t<PR, false, true>(p, s); return launch_gen_t<PR, false, false>(p, s); } while (0)
  if (prec == CRAFT_PREC_F32) GO(CRAFT_PREC_F32);
  if (prec == CRAFT_PREC_BF16) GO(CRAFT_PREC_BF16);
  if (prec == CRAFT_PREC_F16) GO(CRAFT_PREC_F16);
  if (prec == CRAFT_PREC_F16X3) GO(CRAFT_PREC_F16X3);
#undef GO
  return CRAFT_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------
// Convolution weight gradient (stride 1, "same" padding):
//   dW[co][ky][kx][ci] += sum_pix dY[pix][co] * X[pix + (ky - padH, kx - padW)][ci]        (zero outside the image)
// = for every tap a (cout x cin) product over K = all pixels: A = dY k-major, B = X k-major read at the tap's offset.
// grid (ceil(cout / 128), taps * ceil(cin / BN), ksplit); split-K partial sums are added with atomics.
// ---------------------------------------------------------------------------------------------
template <int ROWS> struct LoaderShiftColsF32 {
  typedef RegsF32T<ROWS> Regs;
  const float* base;
  long ld;
  int row[ROWS / 32];
  int nrows, k0, k1, kk, H, W, dy, dx;
  __device__ __forceinline__ void init(const float* base_, long ld_, int row0, int nrows_, int k0_, int k1_, int H_, int W_, int dy_,
                                       int dx_, int tid) {
    base = base_; ld = ld_; nrows = nrows_; k0 = k0_; k1 = k1_; H = H_; W = W_; dy = dy_; dx = dx_;
    kk = tid >> 3;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) row[i] = row0 + ((tid & 7) + 8 * i) * 4;
  }
  __device__ __forceinline__ void fetch(int kt, Regs& r) const {
    const int k = k0 + kt * BK + kk;
    const int kc = k < k1 ? k : k1 - 1;
    const int hw = H * W;
    const int b = kc / hw, rem = kc - b * hw;
    const int y = rem / W, x = rem - y * W;
    const int yy = y + dy, xx = x + dx;
    const bool ok = k < k1 && yy >= 0 && yy < H && xx >= 0 && xx < W;
    const float* p = base + (ok ? ((long)b * hw + (long)yy * W + xx) : 0L) * ld;       // unconditional load, valid address
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int r0 = row[i];
      if (r0 + 4 <= nrows) {
        r.v[i] = *reinterpret_cast<const float4*>(p + r0);
      } else {
        float4 t;
        t.x = p[min(r0, nrows - 1)]; t.y = p[min(r0 + 1, nrows - 1)]; t.z = p[min(r0 + 2, nrows - 1)]; t.w = p[min(r0 + 3, nrows - 1)];
        r.v[i] = t;
      }
    }
    r.zmask = ok ? 0u : 0xffffffffu;
  }
};

struct WgradParams {
  const float* X; const float* dY; float* dW;
  long ldx, ldy;
  int cin, cout, KH, KW, B, H, W;
  int ksplit, kchunk, ntile_n;
};

template <int PREC, int BN>
__global__ __launch_bounds__(NTHREADS) void k_conv_wgrad(WgradParams p) {
  constexpr int BM = 128, WM = 2, WN = 2, MT = BM / WM / 32, NT = BN / WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tap = blockIdx.y / p.ntile_n, nt_i = blockIdx.y - tap * p.ntile_n;
  const int m0 = blockIdx.x * BM, n0 = nt_i * BN;
  const int npix = p.B * p.H * p.W;
  const int k0 = blockIdx.z * p.kchunk, k1 = min(npix, k0 + p.kchunk);
  if (k0 >= k1) return;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  f32x16 acc[MT][NT];
  acc_zero(acc);
  LoaderColsF32<BM> la;
  la.init(p.dY, p.ldy, m0, p.cout, k0, k1, tid);
  LoaderShiftColsF32<BN> lb;
  lb.init(p.X, p.ldx, n0, p.cin, k0, k1, p.H, p.W, ky - p.KH / 2, kx - p.KW / 2, tid);
  gemm_mainloop<PREC, BM, BN, WM, WN>(la, lb, (k1 - k0 + BK - 1) / BK, acc, NoFold());
  const int rb = m0 + (wave / WN) * (BM / WM), cb = n0 + (wave % WN) * (BN / WN);
  const long taps = (long)p.KH * p.KW;
  acc_foreach(acc, lane, [&](int r, int c, float v, int, int, int) {
    const int co = rb + r, ci = cb + c;
    if (co < p.cout && ci < p.cin) unsafeAtomicAdd(p.dW + ((long)co * taps + tap) * p.cin + ci, v);
  });
}

int launch_conv_wgrad(const float* x, long ldx, int cin, const float* dy, long ldy, int cout, int KH, int KW, int B, int H, int W,
                      float* dW, int prec, hipStream_t s) {
  if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return 0;
  if ((ldx & 3) || (ldy & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(dy) & 15)) return CRAFT_ERR_ALIGN;
  WgradParams p = {};
  p.X = x; p.dY = dy; p.dW = dW; p.ldx = ldx; p.ldy = ldy; p.cin = cin; p.cout = cout; p.KH = KH; p.KW = KW; p.B = B; p.H = H; p.W = W;
  const int bn = cin > 96 ? 128 : 64;
  p.ntile_n = (cin + bn - 1) / bn;
  const long npix = (long)B * H * W;
  const long tiles = (long)((cout + 127) / 128) * p.ntile_n * KH * KW;
  long want = (768 + tiles - 1) / tiles;
  const long maxs = (npix + 511) / 512;
  p.ksplit = (int)(want < 1 ? 1 : (want > maxs ? maxs : want));
  p.kchunk = (int)((((npix + p.ksplit - 1) / p.ksplit) + BK - 1) / BK * BK);
  dim3 grid((cout + 127) / 128, p.ntile_n * KH * KW, p.ksplit);
#define GO(PR) do { if (bn == 128) hipLaunchKernelGGL((k_conv_wgrad<PR, 128>), grid, dim3(NTHREADS), 0, s, p); \
                    else hipLaunchKernelGGL((k_conv_wgrad<PR, 64>), grid, dim3(NTHREADS), 0, s, p); return (int)hipGetLastError(); } while (0)
  if (prec == CRAFT_PREC_F32) GO(CRAFT_PREC_F32);
  if (prec == CRAFT_PREC_BF16) GO(CRAFT_PREC_BF16);
  if (prec == CRAFT_PREC_F16) GO(CRAFT_PREC_F16);
  if (prec == CRAFT_PREC_F16X3) GO(CRAFT_PREC_F16X3);
#undef GO
  return CRAFT_ERR_ARG;
}

}  // namespace craft
