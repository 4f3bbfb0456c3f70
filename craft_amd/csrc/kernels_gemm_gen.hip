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
// A k-major tile is fetched with lane = row (a wave reads 256 contiguous bytes of one k-row per load), four consecutive k per
// thread and group, so that every group is one vector write into the [row][k] LDS layout the fragment reads expect.
#include "launch.hpp"

namespace craft {

// ---------------------------------------------------------------------------------------------
// k-major operands.  Thread -> ONE row of the tile (lane = row: the 64 lanes of a wave read 256 contiguous bytes of a k-row)
// and NG groups of 4 consecutive k; a group is 4 scalar loads (each coalesced across the wave) that land in one float4 =
// (row, k..k+3), i.e. exactly what the [row][k] LDS layout stores as ONE 8-byte (16-bit modes) / 16-byte (fp32) write.
// (The first version fetched float4s along the rows and scattered them with four 2-byte LDS writes per plane: 64 ds_write_b16
// per thread and k-tile made the weight-gradient kernel LDS-write-bound at 25 % of the MFMA rate.)
// zmask: 4 bits per group, bit 4*i + j = "k component j of group i is out of range / padding -> store zero".
// ---------------------------------------------------------------------------------------------
template <int ROWS> struct RegsF32K { float4 v[ROWS / 32]; unsigned zmask; };     // NG = 32 k / 4 / (256 / ROWS) = ROWS / 32

template <int PREC, int ROWS>
__device__ __forceinline__ void stage_store_piece(typename PrecT<PREC>::lds_t* S, const RegsF32K<ROWS>& r, int tid, int i) {
  constexpr int LD = PrecT<PREC>::LD, NG = ROWS / 32;
  const int row = tid % ROWS, kg = (tid / ROWS) * NG + i;
  const unsigned z = (r.zmask >> (4 * i)) & 15u;                // wave-uniform (see the loaders): a scalar test, not per-lane selects
  float4 v = r.v[i];
  if (z) { v.x = (z & 1u) ? 0.f : v.x; v.y = (z & 2u) ? 0.f : v.y; v.z = (z & 4u) ? 0.f : v.z; v.w = (z & 8u) ? 0.f : v.w; }
  if constexpr (PREC == CRAFT_PREC_F32) {
    *reinterpret_cast<float4*>(&S[row * LD + kg * 4]) = v;
  } else if constexpr (PREC == CRAFT_PREC_BF16) {
    bf16x4 h;
    h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4*>(&S[row * LD + kg * 4]) = h;
  } else if constexpr (PREC == CRAFT_PREC_F16) {
    f16x4 h;
    h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
    *reinterpret_cast<f16x4*>(&S[row * LD + kg * 4]) = h;
  } else {
    f16x4 h, l;
    split_f16x3(v, h, l);
    *reinterpret_cast<f16x4*>(&S[row * LD + kg * 4]) = h;
    *reinterpret_cast<f16x4*>(&S[(ROWS + row) * LD + kg * 4]) = l;
  }
}
template <int ROWS> __device__ __forceinline__ constexpr int stage_pieces(const RegsF32K<ROWS>&) { return ROWS / 32; }
template <int PREC, int ROWS>
__device__ __forceinline__ void stage_store(typename PrecT<PREC>::lds_t* S, const RegsF32K<ROWS>& r, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) stage_store_piece<PREC, ROWS>(S, r, tid, i);
}

// K-major fp32 operand: element (row, k) at base[k*ld + row], k in [k0, k1) (split-K range); rows >= nrows re-read the last valid
// row (their products only reach output rows / columns the epilogue drops), k >= k1 is stored as zero.
// Everything about k is WAVE-UNIFORM here (a wave holds 64 rows of the same k groups: tid / ROWS is constant over a wave), so
// the k-row addresses, the range tests and the zero mask live on the scalar unit (readfirstlane makes that visible to the
// compiler) and the loads are BUFFER loads: descriptor base = the tile's first k-row (scalar), soffset = k-row byte offset
// (scalar, one s_add per element), voffset = this lane's row (a loop-invariant VGPR) -- no vector address arithmetic at all.
// The first version computed a 64-bit address per lane and element: 16 VALU instructions per MFMA, a quarter of them quarter-
// rate integer multiplies, and the weight-gradient products ran at a quarter of the MFMA rate because of it.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t krow_rsrc(const float* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)0xffffffffu, 0x00020000);      // raw dword buffer, no range limit
}
__device__ __forceinline__ float krow_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}

template <int ROWS> struct LoaderColsF32 {
  typedef RegsF32K<ROWS> Regs;
  static constexpr int NG = ROWS / 32, NE = 4 * NG;
  const float* base;       // uniform
  unsigned rowoff;         // per lane: byte offset of this lane's row
  long ld;
  int k0, k1, kg0;
  __device__ __forceinline__ void init(const float* base_, long ld_, int row0, int nrows, int k0_, int k1_, int tid) {
    ld = ld_; k0 = k0_; k1 = k1_;
    base = base_;
    rowoff = 4u * (unsigned)min(row0 + tid % ROWS, nrows - 1);
    kg0 = __builtin_amdgcn_readfirstlane((tid / ROWS) * NG);
  }
  __device__ __forceinline__ void fetch(int kt, Regs& r) const {
    const int kb = k0 + kt * BK + kg0 * 4;
    const unsigned ld4 = (unsigned)ld * 4u;                      // launcher: ld < 2^26
    float t[NE];
    unsigned zm = 0u;
    if (kb + NE <= k1) {
      const __amdgpu_buffer_rsrc_t rs = krow_rsrc(base + (long)kb * ld);
      unsigned so = 0u;
#pragma unroll
      for (int e = 0; e < NE; ++e) { t[e] = krow_load(rs, rowoff, so); so += ld4; }
    } else {                                                     // the tail of the K range: clamp (a valid address) and zero
      const int kc = min(kb, k1 - 1);
      const __amdgpu_buffer_rsrc_t rs = krow_rsrc(base + (long)kc * ld);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const bool ok = kb + e < k1;
        zm |= ok ? 0u : (1u << e);
        t[e] = krow_load(rs, rowoff, (unsigned)(min(kb + e, k1 - 1) - kc) * ld4);
      }
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) r.v[i] = make_float4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
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
  // The store mode is decided ONCE, outside the element loops (a per-element `accumulate ? *d + t : t` made hipcc branch around a
  // load and wait for it 64 times per lane: the N x N score products, two K-tiles deep, spent most of their time there), and a
  // tile that lies inside the matrix skips the bounds tests.
  const int mode = p.ksplit > 1 ? 2 : (p.accumulate ? 1 : 0);
  const float alpha = p.alpha;
  const long ldc = p.ldc;
  const int M = p.M, N = p.N;
  const bool inside = rb + MT * 32 <= M && cb + NT * 32 <= N;
  const int c = lane & 31, rh = 4 * (lane >> 5);
  auto store_all = [&](auto&& put, auto inb) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = cb + nt * 32 + c;
        float* d0 = C + (long)(rb + mt * 32 + rh) * ldc + col;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = (e & 3) + 8 * (e >> 2);
          if (decltype(inb)::value || (rb + mt * 32 + rh + r < M && col < N)) put(d0 + (long)r * ldc, acc[mt][nt][e] * alpha);
        }
      }
  };
  auto go = [&](auto inb) __attribute__((always_inline)) {
    if (mode == 0) store_all([](float* d, float t) __attribute__((always_inline)) { *d = t; }, inb);
    else if (mode == 1) store_all([](float* d, float t) __attribute__((always_inline)) { *d += t; }, inb);
    else store_all([](float* d, float t) __attribute__((always_inline)) { unsafeAtomicAdd(d, t); }, inb);
  };
  if (inside && mode == 1) {
    // C += : the 16 old values of a fragment are requested together (`*d += t` per element is load, wait, add, store: 64 dependent
    // round trips per lane)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float* d0 = C + (long)(rb + mt * 32 + rh) * ldc + cb + nt * 32 + c;
        float old[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) old[e] = d0[(long)((e & 3) + 8 * (e >> 2)) * ldc];
#pragma unroll
        for (int e = 0; e < 16; ++e) d0[(long)((e & 3) + 8 * (e >> 2)) * ldc] = old[e] + acc[mt][nt][e] * alpha;
      }
  } else if (inside) go(std::true_type());
  else go(std::false_type());
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
  if ((at && (a_sk < 0 || a_sk >= (1L << 26))) || (bt && (b_sk < 0 || b_sk >= (1L << 26)))) return CRAFT_ERR_UNSUPPORTED;   // 32-bit k-row offsets
  if (!bt && b_sk != 1) return CRAFT_ERR_ARG;
  // a k-contiguous ("rows") operand is read with 16-byte vector loads: its leading dimension and batch strides are multiples of 4
  // floats, its base is 16-byte aligned and K is a multiple of 4; a k-major operand is read with scalar loads (lane = row) and has
  // no alignment requirement (the convolution weight gradient uses batch strides of ONE element: the tap shift)
  if (!at && ((a_sm & 3) || (a_bs0 & 3) || (a_bs1 & 3) || (reinterpret_cast<uintptr_t>(A) & 15) || (K & 3))) return CRAFT_ERR_ALIGN;
  if (!bt && ((b_sn & 3) || (b_bs0 & 3) || (b_bs1 & 3) || (reinterpret_cast<uintptr_t>(B) & 15) || (K & 3))) return CRAFT_ERR_ALIGN;
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
__device__ __forceinline__ unsigned bits_range(int lo, int hi) {       // bits lo .. hi-1 (hi <= 16)
  lo = max(lo, 0);
  return hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
}

template <int ROWS> struct LoaderShiftColsF32 {
  typedef RegsF32K<ROWS> Regs;
  static constexpr int NG = ROWS / 32, NE = 4 * NG;
  const float* base;       // uniform
  unsigned rowoff;         // per lane: byte offset of this lane's row (input channel)
  long ld;
  int k0, k1, kg0, H, W, dy, dx, npix;
  unsigned long long magic_hw;         // ceil(2^64 / (H*W)): k / (H*W) = umul64hi(k, magic_hw), exact for every 32-bit k
  unsigned magic_w;                    // ceil(2^32 / W): r / W = umulhi(r, magic_w) for r * W < 2^32 (r < H*W; launcher checks)
  __device__ __forceinline__ void init(const float* base_, long ld_, int row0, int nrows, int k0_, int k1_, int npix_, int H_, int W_,
                                       int dy_, int dx_, unsigned long long magic_hw_, unsigned magic_w_, int tid) {
    init_row(base_, ld_, min(row0 + tid % ROWS, nrows - 1), k0_, k1_, npix_, H_, W_, dy_, dx_, magic_hw_, magic_w_, tid);
  }
  // `row` = this lane's channel, (dy_, dx_) = the tap: both may differ between WAVES (a wave = 64 consecutive tile rows), not lanes
  __device__ __forceinline__ void init_row(const float* base_, long ld_, int row, int k0_, int k1_, int npix_, int H_, int W_, int dy_,
                                           int dx_, unsigned long long magic_hw_, unsigned magic_w_, int tid) {
    ld = ld_; k0 = k0_; k1 = k1_; npix = npix_; H = H_; W = W_; magic_hw = magic_hw_; magic_w = magic_w_;
    dy = __builtin_amdgcn_readfirstlane(dy_); dx = __builtin_amdgcn_readfirstlane(dx_);
    base = base_;
    rowoff = 4u * (unsigned)row;
    kg0 = __builtin_amdgcn_readfirstlane((tid / ROWS) * NG);
  }
  // A wave's NE pixels of one K-tile are consecutive (k = kb .. kb + NE - 1) and wave-uniform (see LoaderColsF32): ONE pair of
  // divisions per fetch gives (y, x) of the first, the validity of the tap is a bit mask built from at most two row segments,
  // and the NE k-rows are read unconditionally at consecutive addresses kb + shift + e (inside the tensor away from its two ends).
  __device__ __forceinline__ void fetch(int kt, Regs& r) const {
    const int hw = H * W;
    const int shift = dy * W + dx;                    // a valid tap of pixel p is pixel p + shift of the same image
    const int kb = k0 + kt * BK + kg0 * 4;
    const unsigned ld4 = (unsigned)ld * 4u;
    float t[NE];
    unsigned zm = 0u;
    if (kb + NE <= k1 && kb + shift >= 0 && kb + NE - 1 + shift < npix && W >= NE) {
      const int rem = kb - (int)__umul64hi((unsigned long long)kb, magic_hw) * hw;
      const int y = (int)__umulhi((unsigned)rem, magic_w), x = rem - y * W;
      const int n1 = min(NE, W - x);                  // pixels left in row y; the rest start row y + 1 (or row 0 of the next image)
      const int x0 = x + dx;
      unsigned m = (unsigned)(y + dy) < (unsigned)H ? bits_range(-x0, min(n1, W - x0)) : 0u;
      const int y2 = (y + 1 == H ? 0 : y + 1) + dy;
      if ((unsigned)y2 < (unsigned)H) m |= bits_range(n1 + max(0, -dx), min((int)NE, n1 + W - dx));
      zm = ~m & ((1u << NE) - 1u);
      const __amdgpu_buffer_rsrc_t rs = krow_rsrc(base + (long)(kb + shift) * ld);
      unsigned so = 0u;
#pragma unroll
      for (int e = 0; e < NE; ++e) { t[e] = krow_load(rs, rowoff, so); so += ld4; }
    } else {                                          // the ends of the tensor / of the K range, images narrower than NE
      const __amdgpu_buffer_rsrc_t rs = krow_rsrc(base);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int k = kb + e, kc = min(k, k1 - 1);
        const int rem = kc - (int)__umul64hi((unsigned long long)kc, magic_hw) * hw;
        const int y = (int)__umulhi((unsigned)rem, magic_w), x = rem - y * W;
        const bool ok = k < k1 && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
        zm |= ok ? 0u : (1u << e);
        const float* pk = base + (long)(ok ? kc + shift : 0) * ld;
        t[e] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pk) + rowoff);
      }
      (void)rs;
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) r.v[i] = make_float4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
    r.zmask = zm;
  }
};

struct WgradParams {
  const float* X; const float* dY; float* dW;
  float* db;                           // non-null: db[co] += sum_pix dY[pix][co] (bias gradient), added by the blocks of tap 0 / input tile 0
  float* ws;                           // non-null: split z writes its partial tile to ws + z * (cout*taps*cin) with plain stores
  long ldx, ldy;
  int cin, cout, KH, KW, B, H, W;
  int ksplit, kchunk, ntile_n, ntile_m;
  unsigned long long magic_hw;
  unsigned magic_w;
};

// Bias gradient folded into the weight-gradient launch: the blocks that own (tap 0, first input-channel tile) of a pixel range
// also add that range's column sums of dY (their A operand, L2-resident after the K loop) into db.  One launch less per conv and
// iteration (132 craft_colsum launches per training step at 12 iterations).
__device__ __forceinline__ void wgrad_bias_tail(const WgradParams& p, int m0, int rows, int k0, int k1, int tid) {
  const int co = m0 + tid % rows, lane_k = tid / rows, nk = NTHREADS / rows;
  if (lane_k >= nk || co >= p.cout) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int k = k0 + lane_k;
  for (; k + 3 * nk < k1; k += 4 * nk) {
    a0 += p.dY[(long)k * p.ldy + co]; a1 += p.dY[(long)(k + nk) * p.ldy + co];
    a2 += p.dY[(long)(k + 2 * nk) * p.ldy + co]; a3 += p.dY[(long)(k + 3 * nk) * p.ldy + co];
  }
  for (; k < k1; k += nk) a0 += p.dY[(long)k * p.ldy + co];
  unsafeAtomicAdd(p.db + co, (a0 + a1) + (a2 + a3));
}

template <int PREC, int BN, bool SB>
__global__ __launch_bounds__(NTHREADS) void k_conv_wgrad(WgradParams p) {
  constexpr int BM = 128, WM = 2, WN = 2, MT = BM / WM / 32, NT = BN / WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware block -> (output tile, pixel range) map.  Every output tile of one pixel range re-reads the same rows of dY and X:
  // if they are spread over the 8 XCDs (block b runs on XCD b % 8) each private L2 fetches the range from the Infinity Cache on
  // its own and the kernel is bound by that fabric.  So: XCD x owns the pixel ranges x, x + 8, ... and walks all output tiles
  // of one range back to back (their working set, kchunk x (cout + cin) x 4 B, stays in that XCD's 4 MB L2).
  const int ntile = p.ntile_m * p.ntile_n * p.KH * p.KW;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int split = xcd + 8 * (j / ntile), tile = j - (j / ntile) * ntile;
  const int mt_i = tile % p.ntile_m, yy = tile / p.ntile_m;
  const int tap = yy / p.ntile_n, nt_i = yy - tap * p.ntile_n;
  const int m0 = mt_i * BM, n0 = nt_i * BN;
  const int npix = p.B * p.H * p.W;
  if (split >= p.ksplit) return;
  const int k0 = split * p.kchunk, k1 = min(npix, k0 + p.kchunk);
  if (k0 >= k1) return;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  f32x16 acc[MT][NT];
  acc_zero(acc);
  LoaderColsF32<BM> la;
  la.init(p.dY, p.ldy, m0, p.cout, k0, k1, tid);
  LoaderShiftColsF32<BN> lb;
  lb.init(p.X, p.ldx, n0, p.cin, k0, k1, npix, p.H, p.W, ky - p.KH / 2, kx - p.KW / 2, p.magic_hw, p.magic_w, tid);
  if constexpr (SB) gemm_mainloop_sb<PREC, BM, BN, WM, WN>(la, lb, (k1 - k0 + BK - 1) / BK, acc);
  else gemm_mainloop<PREC, BM, BN, WM, WN>(la, lb, (k1 - k0 + BK - 1) / BK, acc, NoFold());
  const int rb = m0 + (wave / WN) * (BM / WM), cb = n0 + (wave % WN) * (BN / WN);
  const long taps = (long)p.KH * p.KW;
  if (p.ws) {
    float* dst = p.ws + (long)split * p.cout * taps * p.cin;
    acc_foreach(acc, lane, [&](int r, int c, float v, int, int, int) {
      const int co = rb + r, ci = cb + c;
      if (co < p.cout && ci < p.cin) dst[((long)co * taps + tap) * p.cin + ci] = v;
    });
  } else {
    acc_foreach(acc, lane, [&](int r, int c, float v, int, int, int) {
      const int co = rb + r, ci = cb + c;
      if (co < p.cout && ci < p.cin) unsafeAtomicAdd(p.dW + ((long)co * taps + tap) * p.cin + ci, v);
    });
  }
  if (p.db && tap == 0 && nt_i == 0) wgrad_bias_tail(p, m0, BM, k0, k1, tid);
}

// cout <= 64 and cin == 64 (the encoders' 64-channel layers): a 128-row M tile would be half padding.  Here the block tile is
// 64 (cout) x 128 = TWO taps x 64 input channels, 4 waves side by side along N (each 64 x 32): the B tile's rows 0..63 are X at tap
// 2t, rows 64..127 X at tap 2t + 1 -- a wave's 64 rows share one tap, which is all the shifted loader needs.
template <int PREC>
__global__ __launch_bounds__(NTHREADS) void k_conv_wgrad64(WgradParams p) {
  constexpr int BM = 64, BN = 128, WM = 1, WN = 4, MT = 2, NT = 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int taps = p.KH * p.KW, ntile = (taps + 1) / 2;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int split = xcd + 8 * (j / ntile), tile = j - (j / ntile) * ntile;
  const int npix = p.B * p.H * p.W;
  if (split >= p.ksplit) return;
  const int k0 = split * p.kchunk, k1 = min(npix, k0 + p.kchunk);
  if (k0 >= k1) return;
  f32x16 acc[MT][NT];
  acc_zero(acc);
  LoaderColsF32<BM> la;
  la.init(p.dY, p.ldy, 0, p.cout, k0, k1, tid);
  const int brow = tid % BN;                                   // tile row of the B operand this lane stages
  const int tap_l = 2 * tile + brow / 64;
  const bool tap_ok = tap_l < taps;
  const int ky = tap_l / p.KW, kx = tap_l - ky * p.KW;
  LoaderShiftColsF32<BN> lb;
  // a tap that does not exist (odd tap count) reads as "always outside the image": dy = H
  lb.init_row(p.X, p.ldx, brow % 64, k0, k1, npix, p.H, p.W, tap_ok ? ky - p.KH / 2 : p.H, tap_ok ? kx - p.KW / 2 : 0, p.magic_hw,
              p.magic_w, tid);
  gemm_mainloop<PREC, BM, BN, WM, WN>(la, lb, (k1 - k0 + BK - 1) / BK, acc, NoFold());
  const int cb = wave * 32;
  acc_foreach(acc, lane, [&](int r, int c, float v, int, int, int) {
    const int co = r, n = cb + c, tap = 2 * tile + (n >> 6), ci = n & 63;
    if (co < p.cout && tap < taps) {
      float* d = (p.ws ? p.ws + (long)split * p.cout * taps * 64 : p.dW) + ((long)co * taps + tap) * 64 + ci;
      if (p.ws) *d = v; else unsafeAtomicAdd(d, v);
    }
  });
  if (p.db && tile == 0) wgrad_bias_tail(p, 0, BM, k0, k1, tid);
}

int launch_reduce_replicas(const float* rep, int nrep, int n, float* out, hipStream_t s);

int launch_conv_wgrad(const float* x, long ldx, int cin, const float* dy, long ldy, int cout, int KH, int KW, int B, int H, int W,
                      float* dW, float* db, float* ws, long ws_floats, int prec, hipStream_t s) {
  if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return 0;
  if (ldx >= (1L << 26) || ldy >= (1L << 26)) return CRAFT_ERR_UNSUPPORTED;
  if ((ldx & 3) || (ldy & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(dy) & 15)) return CRAFT_ERR_ALIGN;
  WgradParams p = {};
  p.X = x; p.dY = dy; p.dW = dW; p.db = db; p.ldx = ldx; p.ldy = ldy; p.cin = cin; p.cout = cout; p.KH = KH; p.KW = KW; p.B = B; p.H = H; p.W = W;
  const int bn = cin > 96 ? 128 : 64;
  p.ntile_n = (cin + bn - 1) / bn;
  p.ntile_m = (cout + 127) / 128;
  if ((double)B * H * W >= 2147483648.0 || (double)H * W * W >= 4294967296.0 || H * W < 2 || W < 2) return CRAFT_ERR_UNSUPPORTED;   // division ranges
  p.magic_hw = ~0ULL / (unsigned long long)(H * W) + 1ULL;
  p.magic_w = (unsigned)((4294967296ULL + (unsigned long long)W - 1) / (unsigned long long)W);
  const long npix = (long)B * H * W;
  const long tiles = (cout <= 64 && cin == 64 && !tuning().no_wgrad64) ? (KH * KW + 1) / 2 : (long)((cout + 127) / 128) * p.ntile_n * KH * KW;
  long want = (768 + tiles - 1) / tiles;
  const long maxs = (npix + 511) / 512;
  p.ksplit = (int)(want < 1 ? 1 : (want > maxs ? maxs : want));
  // split-K partial sums: plain stores into the caller's scratch + one reduction pass when it is large enough (millions of
  // same-line fp32 atomics in L2 cost more than the products); atomics straight into dW otherwise
  const long nw = (long)cout * KH * KW * cin;
  p.ws = (ws != nullptr && p.ksplit > 1 && ws_floats >= nw * p.ksplit) ? ws : nullptr;
  p.kchunk = (int)((((npix + p.ksplit - 1) / p.ksplit) + BK - 1) / BK * BK);
  const int nz = (int)((npix + p.kchunk - 1) / p.kchunk);          // splits that actually own pixels (the others would leave holes)
  p.ksplit = nz;
  const bool pair64 = cout <= 64 && cin == 64 && !tuning().no_wgrad64;
  const bool sb = tuning().wgrad_sb;
  const int ntile = pair64 ? (KH * KW + 1) / 2 : p.ntile_m * p.ntile_n * KH * KW;
  dim3 grid((unsigned)(8 * ntile * ((p.ksplit + 7) / 8)), 1, 1);
#define GO(PR) do { if (pair64) hipLaunchKernelGGL((k_conv_wgrad64<PR>), grid, dim3(NTHREADS), 0, s, p); \
                    else if (bn == 128 && sb) hipLaunchKernelGGL((k_conv_wgrad<PR, 128, true>), grid, dim3(NTHREADS), 0, s, p); \
                    else if (bn == 128) hipLaunchKernelGGL((k_conv_wgrad<PR, 128, false>), grid, dim3(NTHREADS), 0, s, p); \
                    else hipLaunchKernelGGL((k_conv_wgrad<PR, 64, false>), grid, dim3(NTHREADS), 0, s, p); \
                    { const hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } \
                    return p.ws ? launch_reduce_replicas(p.ws, p.ksplit, (int)nw, dW, s) : 0; } while (0)
  if (prec == CRAFT_PREC_F32) GO(CRAFT_PREC_F32);
  if (prec == CRAFT_PREC_BF16) GO(CRAFT_PREC_BF16);
  if (prec == CRAFT_PREC_F16) GO(CRAFT_PREC_F16);
  if (prec == CRAFT_PREC_F16X3) GO(CRAFT_PREC_F16X3);
#undef GO
  return CRAFT_ERR_ARG;
}

}  // namespace craft
