// Tiled MFMA "NT" GEMM engine for gfx950:  C[m,n] = sum_k A[m,k] * B[n,k]   (both operands K-contiguous).
//
// Everything dense on CRAFT's hot path is expressed in this one form:
//   * nn.Linear projections                       A = tokens [rows, Cin],          B = W [Cout, Cin]
//   * NHWC implicit-GEMM convolutions             A = gathered pixels [B*H*W, taps*Cin], B = packed W [Cout, taps*Cin]
//   * attention scores / correlation volume       A = Q rows, B = K rows (mode m = column slice m*d..)
//   * attention apply  O = P V                    A = P [N, N],  B = V^T [Dv, N]
//
// Block = 256 threads = 4 waves (WM x WN); block tile BM x BN, K-tile 32; each wave owns
// (BM/WM) x (BN/WN) as MT x NT MFMA tiles of 32x32.  Operands are staged global -> registers -> LDS
// (double-buffered, one barrier per K-tile; the next tile's global loads are in flight during the
// MFMAs) and read back as 16-byte fragments.
//
// Precision (template PREC): F32 uses v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain);
// BF16 / F16 convert while staging into LDS and use v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulate.
//
// K-slot trick: an NT product is invariant under any permutation of k applied to both operands, so
// each lane reads ONE 16-byte vector per operand per MFMA group and feeds its components to
// successive MFMAs; which k lands in which hardware k-slot is irrelevant as long as A and B agree.
// LDS rows are padded (fp32: 36 floats = 144 B, 16-bit: 40 halves = 80 B) so the 16-lane groups of
// ds_read_b128 hit distinct 16-B bank slots (MI355X_MICROARCH §LDS).
#pragma once
#include "common.hpp"

namespace craft {

constexpr int BK = 32;
constexpr int NTHREADS = 256;

template <int PREC> struct PrecT;
template <> struct PrecT<CRAFT_PREC_F32> { typedef float lds_t; static constexpr int LD = 36; };
template <> struct PrecT<CRAFT_PREC_BF16> { typedef __bf16 lds_t; static constexpr int LD = 40; };
template <> struct PrecT<CRAFT_PREC_F16> { typedef _Float16 lds_t; static constexpr int LD = 40; };
// F16X3: every fp32 operand is split into two fp16 planes x = hi + lo (hi = fp16(x), lo = fp16(x - hi), 22
// significant bits) and a product is three fp16 MFMAs hi*hi + hi*lo + lo*hi with fp32 accumulation: fp32-class
// results (relative error ~2^-22 per product) at 3/16 of the fp32-MFMA cost.  gfx950 has no TF32/xf32 path.
template <> struct PrecT<CRAFT_PREC_F16X3> { typedef _Float16 lds_t; static constexpr int LD = 40; };

template <int PREC> struct Planes { static constexpr int N = (PREC == CRAFT_PREC_F16X3) ? 2 : 1; };

template <int PREC, int BM, int BN> struct TileLds {
  typedef typename PrecT<PREC>::lds_t lds_t;
  static constexpr int LD = PrecT<PREC>::LD;
  static constexpr int A_ELEMS = Planes<PREC>::N * BM * LD;
  static constexpr int B_ELEMS = Planes<PREC>::N * BN * LD;
  static constexpr int BYTES = 2 * (A_ELEMS + B_ELEMS) * (int)sizeof(lds_t);
};

// ---------------------------------------------------------------------------------------------
// staging registers
// ---------------------------------------------------------------------------------------------
// zmask: bit i set -> chunk i must be stored as zeros (K tail / conv padding).  The zeroing is applied when the
// registers are written to LDS, NOT right after the load: a select next to the load would make the compiler
// wait for the data immediately and lose the overlap of the global-load latency with the MFMAs.
template <int ROWS> struct RegsF32 { float4 v[ROWS / 32]; unsigned zmask; };   // thread: k-chunk (tid&7)*4, rows (tid>>3)+32*i
template <int ROWS> struct RegsH16 { uint4 v[ROWS / 64]; unsigned zmask; };    // thread: k-chunk (tid&3)*8, rows (tid>>2)+64*i

// one register (float4 / uint4) of a staged tile -> LDS; stage_store = all pieces
template <int PREC, int ROWS>
__device__ __forceinline__ void stage_store_piece(typename PrecT<PREC>::lds_t* S, const RegsF32<ROWS>& r, int tid, int i) {
  constexpr int LD = PrecT<PREC>::LD;
  const int c4 = tid & 7, r0 = tid >> 3;
  {
    const int row = r0 + 32 * i;
    const bool z = (r.zmask >> i) & 1u;
    float4 v;
    v.x = z ? 0.f : r.v[i].x; v.y = z ? 0.f : r.v[i].y; v.z = z ? 0.f : r.v[i].z; v.w = z ? 0.f : r.v[i].w;
    if constexpr (PREC == CRAFT_PREC_F32) {
      *reinterpret_cast<float4*>(&S[row * LD + c4 * 4]) = v;
    } else if constexpr (PREC == CRAFT_PREC_BF16) {
      bf16x4 h;
      h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
      *reinterpret_cast<bf16x4*>(&S[row * LD + c4 * 4]) = h;
    } else if constexpr (PREC == CRAFT_PREC_F16) {
      f16x4 h;
      h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
      *reinterpret_cast<f16x4*>(&S[row * LD + c4 * 4]) = h;
    } else {   // F16X3: hi plane, then lo plane at + ROWS*LD
      f16x4 h, l;
      split_f16x3(v, h, l);
      *reinterpret_cast<f16x4*>(&S[row * LD + c4 * 4]) = h;
      *reinterpret_cast<f16x4*>(&S[(ROWS + row) * LD + c4 * 4]) = l;
    }
  }
}
template <int ROWS> __device__ __forceinline__ constexpr int stage_pieces(const RegsF32<ROWS>&) { return ROWS / 32; }
template <int ROWS> __device__ __forceinline__ constexpr int stage_pieces(const RegsH16<ROWS>&) { return ROWS / 64; }
template <int PREC, int ROWS>
__device__ __forceinline__ void stage_store(typename PrecT<PREC>::lds_t* S, const RegsF32<ROWS>& r, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) stage_store_piece<PREC, ROWS>(S, r, tid, i);
}
template <int PREC, int ROWS>
__device__ __forceinline__ void stage_store_piece(typename PrecT<PREC>::lds_t* S, const RegsH16<ROWS>& r, int tid, int i) {
  static_assert(PREC == CRAFT_PREC_BF16 || PREC == CRAFT_PREC_F16, "16-bit operands need a plain 16-bit MFMA mode");
  constexpr int LD = PrecT<PREC>::LD;
  const int c8 = tid & 3, r0 = tid >> 2;
  {
    const int row = r0 + 64 * i;
    const bool z = (r.zmask >> i) & 1u;
    uint4 v;
    v.x = z ? 0u : r.v[i].x; v.y = z ? 0u : r.v[i].y; v.z = z ? 0u : r.v[i].z; v.w = z ? 0u : r.v[i].w;
    *reinterpret_cast<uint4*>(&S[row * LD + c8 * 8]) = v;
  }
}
template <int PREC, int ROWS>
__device__ __forceinline__ void stage_store(typename PrecT<PREC>::lds_t* S, const RegsH16<ROWS>& r, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS / 64; ++i) stage_store_piece<PREC, ROWS>(S, r, tid, i);
}

// ---------------------------------------------------------------------------------------------
// loaders (global -> registers).  fetch(kt, regs) loads K-tile kt (k in [32kt, 32kt+32)).
// ---------------------------------------------------------------------------------------------
// Plain fp32 rows: element (row, k) at base[row*ld + k]; rows >= nrows and k >= K read as zero.
// Requirements: base 16-B aligned, ld % 4 == 0, K % 4 == 0.
// NOTE on predication: a load guarded by a per-lane condition ("valid ? *p : 0") makes hipcc branch around
// the load and wait vmcnt(0) right behind it, serialising every load of the tile.  All loaders therefore load
// UNCONDITIONALLY from a clamped, always-valid address; out-of-range rows re-read the last valid row (their
// products only reach output rows / columns that the epilogues discard) and the K tail is zeroed by a value
// select after the load.
template <int ROWS> struct LoaderRowsF32 {
  typedef RegsF32<ROWS> Regs;
  const float* p[ROWS / 32];
  int K, kcol;
  __device__ __forceinline__ void init(const float* base, long ld, int row0, int nrows, int K_, int tid) {
    K = K_;
    kcol = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int row = min(row0 + (tid >> 3) + 32 * i, nrows - 1);
      p[i] = base + (long)row * ld;
    }
  }
  __device__ __forceinline__ void fetch(int kt, Regs& r) const {
    const int k = kt * BK + kcol;
    const bool kok = k < K;
    const int kc = kok ? k : K - 4;
    r.zmask = kok ? 0u : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) r.v[i] = *reinterpret_cast<const float4*>(p[i] + kc);
  }
};

// 16-bit rows (attention probabilities P): ld % 8 == 0, K % 8 == 0.
template <int ROWS> struct LoaderRowsH16 {
  typedef RegsH16<ROWS> Regs;
  const uint16_t* p[ROWS / 64];
  int K, kcol;
  __device__ __forceinline__ void init(const uint16_t* base, long ld, int row0, int nrows, int K_, int tid) {
    K = K_;
    kcol = (tid & 3) * 8;
#pragma unroll
    for (int i = 0; i < ROWS / 64; ++i) {
      const int row = min(row0 + (tid >> 2) + 64 * i, nrows - 1);
      p[i] = base + (long)row * ld;
    }
  }
  __device__ __forceinline__ void fetch(int kt, Regs& r) const {
    const int k = kt * BK + kcol;
    const bool kok = k < K;
    const int kc = kok ? k : K - 8;
    r.zmask = kok ? 0u : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < ROWS / 64; ++i) r.v[i] = *reinterpret_cast<const uint4*>(p[i] + kc);
  }
};

// NHWC implicit-GEMM gather: row = pixel (b, y, x); k = (tap, channel) over up to two concatenated
// channel segments (a virtual torch.cat: seg0 has c0 channels at row stride ld0, seg1 c1 at ld1).
// Each segment's channel count must be a multiple of 32, so a K-tile never straddles a tap or segment.
struct ConvGeom {
  const float* seg0; const float* seg1;
  int ld0, ld1, c0, c1;
  int H, W, KH, KW, padH, padW;              // H, W: OUTPUT size
  int npix;                                  // B*H*W (output pixels)
  int stride, Hin, Win;                      // input size (== H, W for stride 1); stride 2 only via k_gemm_conv
  const float* in_norm;                      // optional [B][c0][2] (mean, rstd): input is relu((x-mean)*rstd)
                                             // (lazy InstanceNorm + ReLU of the producing conv; k_conv_halo only)
};
template <int ROWS> struct LoaderConvF32 {
  typedef RegsF32<ROWS> Regs;
  ConvGeom g;
  int py[ROWS / 32], px[ROWS / 32];
  long pimg[ROWS / 32];
  int kcol, ctot;
  __device__ __forceinline__ void init(const ConvGeom& g_, int row0, int tid) {
    g = g_;
    kcol = (tid & 7) * 4;
    ctot = g.c0 + g.c1;
    const int hw = g.H * g.W;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int row = row0 + (tid >> 3) + 32 * i;
      if (row < g.npix) {
        const int b = row / hw, rem = row - b * hw;
        py[i] = (rem / g.W) * g.stride;                      // input coordinates of the window origin
        px[i] = (rem - (rem / g.W) * g.W) * g.stride;
        pimg[i] = (long)b * g.Hin * g.Win;
      } else {
        py[i] = -100000; px[i] = 0; pimg[i] = 0;
      }
    }
  }
  __device__ __forceinline__ void fetch(int kt, Regs& r) const {
    const int k0 = kt * BK;
    const int tap = k0 / ctot, cb = k0 - tap * ctot;
    const int ty = tap / g.KW;
    const int dy = ty - g.padH, dx = (tap - ty * g.KW) - g.padW;
    const float* sp; int ld, c;
    if (cb < g.c0) { sp = g.seg0; ld = g.ld0; c = cb; } else { sp = g.seg1; ld = g.ld1; c = cb - g.c0; }
    unsigned zm = 0u;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int yy = py[i] + dy, xx = px[i] + dx;
      const bool ok = yy >= 0 && yy < g.Hin && xx >= 0 && xx < g.Win;
      const long pix = ok ? pimg[i] + (long)yy * g.Win + xx : 0;    // unconditional load from a valid address
      zm |= ok ? 0u : (1u << i);
      r.v[i] = *reinterpret_cast<const float4*>(sp + pix * ld + c + kcol);
    }
    r.zmask = zm;
  }
};

// ---------------------------------------------------------------------------------------------
// LDS fragments -> MFMA
// ---------------------------------------------------------------------------------------------
struct NoSlot { __device__ __forceinline__ void operator()(int) const {} };
// `slot(i)`, i = 0 .. 2*MT*NT-1, runs after the MFMAs of accumulator tile (kk, mt, nt) in the 16-bit modes: the caller
// places slices of other work there, each pinned behind those MFMAs (VALU hides only behind MFMAs of the same wave).
template <int PREC, int MT, int NT, int BM, int BN, class SLOT = NoSlot>
__device__ __forceinline__ void mma_tile(const typename PrecT<PREC>::lds_t* As, const typename PrecT<PREC>::lds_t* Bs,
                                         int wm0, int wn0, int lane, f32x16 (&acc)[MT][NT], SLOT&& slot = NoSlot()) {
  constexpr int LD = PrecT<PREC>::LD;
  const int r = lane & 31, g = lane >> 5;
  if constexpr (PREC == CRAFT_PREC_F32) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float4 a[MT], b[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4*>(&As[(wm0 + mt * 32 + r) * LD + kk * 8 + g * 4]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const float4*>(&Bs[(wn0 + nt * 32 + r) * LD + kk * 8 + g * 4]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, b[nt].x, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, b[nt].y, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].z, b[nt].z, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].w, b[nt].w, acc[mt][nt], 0, 0, 0);
        }
    }
  } else if constexpr (PREC == CRAFT_PREC_BF16) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a[MT], b[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(&As[(wm0 + mt * 32 + r) * LD + kk * 16 + g * 8]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const bf16x8*>(&Bs[(wn0 + nt * 32 + r) * LD + kk * 16 + g * 8]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
        {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
          slot((kk * MT + mt) * NT + nt);
        }
    }
  } else if constexpr (PREC == CRAFT_PREC_F16) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f16x8 a[MT], b[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const f16x8*>(&As[(wm0 + mt * 32 + r) * LD + kk * 16 + g * 8]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f16x8*>(&Bs[(wn0 + nt * 32 + r) * LD + kk * 16 + g * 8]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
        {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
          slot((kk * MT + mt) * NT + nt);
        }
    }
  } else {   // F16X3
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        ah[mt] = *reinterpret_cast<const f16x8*>(&As[(wm0 + mt * 32 + r) * LD + kk * 16 + g * 8]);
        al[mt] = *reinterpret_cast<const f16x8*>(&As[(BM + wm0 + mt * 32 + r) * LD + kk * 16 + g * 8]);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bh[nt] = *reinterpret_cast<const f16x8*>(&Bs[(wn0 + nt * 32 + r) * LD + kk * 16 + g * 8]);
        bl[nt] = *reinterpret_cast<const f16x8*>(&Bs[(BN + wn0 + nt * 32 + r) * LD + kk * 16 + g * 8]);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
          slot((kk * MT + mt) * NT + nt);
        }
    }
  }
}

template <int MT, int NT> __device__ __forceinline__ void acc_zero(f32x16 (&acc)[MT][NT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
}

// Visit every accumulator element of this lane: f(row_in_wave_tile, col_in_wave_tile, value, mt, nt, reg).
// C/D layout of the 32x32 MFMA family: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
template <int MT, int NT, class F>
__device__ __forceinline__ void acc_foreach(f32x16 (&acc)[MT][NT], int lane, F&& f) {
  const int c = lane & 31, rh = 4 * (lane >> 5);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) f(mt * 32 + (e & 3) + 8 * (e >> 2) + rh, nt * 32 + c, acc[mt][nt][e], mt, nt, e);
}

struct NoFold { __device__ __forceinline__ void operator()(int) const {} };

// The K loop.  `fold(kt)` runs after K-tile kt has been accumulated (used by the correlation build to
// close a mode every d/32 tiles).  Ends with a barrier, so it can be called repeatedly.
template <int PREC, int BM, int BN, int WM, int WN, class LA, class LB, class FOLD>
__device__ __forceinline__ void gemm_mainloop(const LA& la, const LB& lb, int nk,
                                              f32x16 (&acc)[BM / WM / 32][BN / WN / 32], FOLD&& fold) {
  typedef typename PrecT<PREC>::lds_t lds_t;
  typedef TileLds<PREC, BM, BN> L;
  constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
  // The tile buffers live HERE (one static LDS array, indexed by integer offsets) so that every access is
  // provably in the LDS address space: handing the engine a generic pointer made hipcc emit flat_load
  // instead of ds_read_b128 for the fragments.  Layout: A0 | A1 | B0 | B1.
  __shared__ __attribute__((aligned(16))) lds_t S[2 * (L::A_ELEMS + L::B_ELEMS)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  typename LA::Regs ra;
  typename LB::Regs rb;
  la.fetch(0, ra);
  lb.fetch(0, rb);
  stage_store<PREC>(&S[0], ra, tid);
  stage_store<PREC>(&S[2 * L::A_ELEMS], rb, tid);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int ao = (kt & 1) * L::A_ELEMS, bo = 2 * L::A_ELEMS + (kt & 1) * L::B_ELEMS;
    const int an = L::A_ELEMS - ao, bn = 2 * L::A_ELEMS + L::B_ELEMS - (kt & 1) * L::B_ELEMS;
    // Straight-line body: the next tile is always fetched (index clamped; the duplicate of the last tile lands in
    // the idle buffer) and the issue order is pinned -- left alone, hipcc sinks the global loads below the
    // MFMAs to save registers and then waits on them at once, exposing an L2 round trip in every K-tile.
    const int ktn = min(kt + 1, nk - 1);
    la.fetch(ktn, ra);
    lb.fetch(ktn, rb);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PREC == CRAFT_PREC_F32) {
      mma_tile<PREC, MT, NT, BM, BN>(&S[ao], &S[bo], wm0, wn0, lane, acc);
      fold(kt);
      __builtin_amdgcn_sched_barrier(0);
      stage_store<PREC>(&S[an], ra, tid);
      stage_store<PREC>(&S[bn], rb, tid);
    } else {
      // the conversion / split of the next tile is cut into pieces (one staged register each) placed behind the MFMAs of
      // this tile, starting after the first third (the fetch is still in flight); what does not fit follows the last MFMA
      constexpr int NSLOT = 2 * MT * NT, S0 = NSLOT / 3;
      constexpr int NPA = stage_pieces(typename LA::Regs()), NPB = stage_pieces(typename LB::Regs());
      auto piece = [&](int j) __attribute__((always_inline)) {
        if (j < NPA) stage_store_piece<PREC>(&S[an], ra, tid, j);
        else if (j < NPA + NPB) stage_store_piece<PREC>(&S[bn], rb, tid, j - NPA);
      };
      mma_tile<PREC, MT, NT, BM, BN>(&S[ao], &S[bo], wm0, wn0, lane, acc, [&](int i) __attribute__((always_inline)) {
        if (i >= S0) piece(i - S0);
        __builtin_amdgcn_sched_barrier(0);
      });
      fold(kt);
#pragma unroll
      for (int j = NSLOT - S0; j < NPA + NPB; ++j) piece(j);
    }
    __syncthreads();
  }
}

// The same K loop with ONE LDS tile buffer (half the LDS: twice the resident blocks where LDS is the occupancy limit).  The next
// tile still travels global -> registers during the MFMAs; it is converted and stored after a barrier (nobody reads the buffer any
// more) and published by a second one.  The store phase is not hidden behind this block's own MFMAs -- the extra resident waves
// are what covers it.
template <int PREC, int BM, int BN, int WM, int WN, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop_sb(const LA& la, const LB& lb, int nk, f32x16 (&acc)[BM / WM / 32][BN / WN / 32]) {
  typedef typename PrecT<PREC>::lds_t lds_t;
  typedef TileLds<PREC, BM, BN> L;
  constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
  __shared__ __attribute__((aligned(16))) lds_t S[L::A_ELEMS + L::B_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  typename LA::Regs ra;
  typename LB::Regs rb;
  la.fetch(0, ra);
  lb.fetch(0, rb);
  stage_store<PREC>(&S[0], ra, tid);
  stage_store<PREC>(&S[L::A_ELEMS], rb, tid);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int ktn = min(kt + 1, nk - 1);
    la.fetch(ktn, ra);
    lb.fetch(ktn, rb);
    __builtin_amdgcn_sched_barrier(0);
    mma_tile<PREC, MT, NT, BM, BN>(&S[0], &S[L::A_ELEMS], wm0, wn0, lane, acc);
    __syncthreads();
    if (kt + 1 < nk) {
      stage_store<PREC>(&S[0], ra, tid);
      stage_store<PREC>(&S[L::A_ELEMS], rb, tid);
    }
    __syncthreads();
  }
}

}  // namespace craft
