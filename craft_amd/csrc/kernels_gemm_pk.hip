// Packed-operand contraction engine of the backward pass (include/craft_hip.h: craft_pack_operand, craft_wgrad_pk).
//
// The weight gradients  dW[co][tap][ci] = sum_pix dY[pix][co] * X[pix + tap][ci]  (autograd of update.py:49-64, :79-87, extractor.py)
// are products over K = ALL pixels whose operands are both "k-major": memory is contiguous along the channel, the contraction runs
// down the rows.  The first engine (kernels_gemm_gen.hip) read fp32 rows with scalar loads, split every element into its two fp16
// planes and transposed through LDS inside the K loop: the conversion VALU was additive to the MFMA time (profiles/r2/
// engine_probe.txt: 770 of 1 950 cycles per K-tile) and the 4-wave 128 x 128 block was latency-bound.  Here:
//
//   * operands are PACKED once per tensor (k_pack_operand, HBM-bound): fp32 -> 1 (fp16 / bf16) or 2 (f16x3: hi, lo) 16-bit planes
//     in channel-group-major order  P[plane][C/32][rows_p][32]  -- a 32-channel x 32-row tile is one contiguous 2 KiB.  The
//     spatial form lays the rows out as a zero-padded pixel grid [B][H + 2 padH][W + 2 padW] (plus zero guard rows), so that a
//     convolution tap is ONE constant row shift and the padding of the convolution needs no masks anywhere;
//   * the K loop of the GEMM is a pure copy: tiles travel global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave
//     instruction, no VGPRs, no VALU), double-buffered, one barrier per K-tile;
//   * the transposition is done by the LDS read: ds_read_b64_tr_b16 hands every lane 4 consecutive k of its column straight
//     in MFMA operand order (a [4 k][16 m] block per 16 lanes; the four 64-byte rows of a 32-lane group cover all 64 banks once);
//   * 8 waves per block, 256-wide tiles (wave tile up to 128 x 64: 48 MFMAs per 48 LDS reads and K-tile);
//   * the N dimension is (tap, ci) flattened in 32-channel groups, each group with its own row shift: layers with few input
//     channels share one dY tile between several taps.
#include "launch.hpp"

namespace craft {

#define CRAFT_LDS __attribute__((address_space(3)))
typedef __fp16 fp16x4_raw __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __bf16 bf16x4_raw __attribute__((__vector_size__(4 * sizeof(__bf16))));

// ---------------------------------------------------------------------------------------------------------------------
// pack
// ---------------------------------------------------------------------------------------------------------------------
struct PackParams {
  const float* x; long ldx; int C; long rows;
  int B, H, W, padH, padW;       // B > 0: spatial form
  int tail;                      // spatial form: extra (zero) columns behind every grid row: Wp = W + 2 padW + tail (H = 1, padW = 0: B batches of W
                                 // rows, each padded to W + tail rows -- the per-batch operands of craft_gemm_pk)
  long guard, rows_p;
  unsigned short* out; int prec;
  float* colsum;                 // optional [C]: += sum over rows (bias gradient of a convolution: its dY is packed anyway)
  int ncg;                       // channel groups of THIS source
  int cg_off, ncg_total;         // where they go: groups [cg_off, cg_off + ncg) of a pack of ncg_total groups (a virtual torch.cat)
};

// block = 256 threads; it packs PACK_ROWS rows x one 32-channel group: thread -> (row = tid >> 2 (+ 64 per pass), 8 channels at
// (tid & 3) * 8), PACK_ROWS / 64 passes with every load issued before the first store.
// Column sums: per-thread partial sums over the passes, reduced over the 16 row lanes by shuffles and over the 4 waves through LDS:
// 32 atomics per block (the first version had every wave add its 16-row sums: 3 000 same-address atomics per channel cost 20 x the copy).
// One launch packs up to PACK_MAX_DESC tensors (a training iteration packs ~14 conv inputs of 2 - 12 MB each: as separate launches they
// were overhead, not bandwidth -- 13 us for 5 us of traffic): block -> (descriptor, row block, channel group) through a prefix table.
constexpr int PACK_ROWS = 256;
constexpr int PACK_MAX_DESC = 16;
struct PackBatch { PackParams d[PACK_MAX_DESC]; int first_block[PACK_MAX_DESC + 1]; int n; };

template <int PREC>
__device__ __forceinline__ void pack_block(const PackParams& p, int rb, int cg, bool sum, float (*red)[32]) {
  constexpr int NP = PACK_ROWS / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cq = (tid & 3) * 8, c = cg * 32 + cq;
  const long plane = (long)p.ncg_total * p.rows_p * 32;
  const int Hp = p.H + 2 * p.padH, Wp = p.W + 2 * p.padW + p.tail;
  float4 v0[NP], v1[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const long r = (long)rb * PACK_ROWS + i * 64 + (tid >> 2);
    long src = -1;
    const long q = r - p.guard;
    if (p.B > 0) {
      if (q >= 0 && q < (long)p.B * Hp * Wp) {
        const int qi = (int)q, b = qi / (Hp * Wp), rem = qi - b * Hp * Wp;
        const int y = rem / Wp - p.padH, x = rem % Wp - p.padW;
        if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) src = ((long)b * p.H + y) * p.W + x;
      }
    } else if (q >= 0 && q < p.rows) src = q;
    v0[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    v1[i] = v0[i];
    if (src >= 0) {                                  // (C % 4 == 0: a float4 is in or out as a whole)
      const float* xr = p.x + src * p.ldx + c;
      if (c < p.C) v0[i] = *reinterpret_cast<const float4*>(xr);
      if (c + 4 < p.C) v1[i] = *reinterpret_cast<const float4*>(xr + 4);
    }
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const long r = (long)rb * PACK_ROWS + i * 64 + (tid >> 2);
    if (r >= p.rows_p) continue;
    unsigned short* o = p.out + ((long)(p.cg_off + cg) * p.rows_p + r) * 32 + cq;
    if constexpr (PREC == CRAFT_PREC_F16X3) {
      f16x4 h0, l0, h1, l1;
      split_f16x3(v0[i], h0, l0);
      split_f16x3(v1[i], h1, l1);
      f16x8 h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) { h[j] = h0[j]; h[4 + j] = h1[j]; l[j] = l0[j]; l[4 + j] = l1[j]; }
      *reinterpret_cast<f16x8*>(o) = h;
      *reinterpret_cast<f16x8*>(o + plane) = l;
    } else if constexpr (PREC == CRAFT_PREC_F16) {
      f16x8 h;
      h[0] = (_Float16)v0[i].x; h[1] = (_Float16)v0[i].y; h[2] = (_Float16)v0[i].z; h[3] = (_Float16)v0[i].w;
      h[4] = (_Float16)v1[i].x; h[5] = (_Float16)v1[i].y; h[6] = (_Float16)v1[i].z; h[7] = (_Float16)v1[i].w;
      *reinterpret_cast<f16x8*>(o) = h;
    } else {
      bf16x8 h;
      h[0] = (__bf16)v0[i].x; h[1] = (__bf16)v0[i].y; h[2] = (__bf16)v0[i].z; h[3] = (__bf16)v0[i].w;
      h[4] = (__bf16)v1[i].x; h[5] = (__bf16)v1[i].y; h[6] = (__bf16)v1[i].z; h[7] = (__bf16)v1[i].w;
      *reinterpret_cast<bf16x8*>(o) = h;
    }
  }
  if (sum) {                                         // (block-uniform)
    float sm[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      sm[0] += v0[i].x; sm[1] += v0[i].y; sm[2] += v0[i].z; sm[3] += v0[i].w;
      sm[4] += v1[i].x; sm[5] += v1[i].y; sm[6] += v1[i].z; sm[7] += v1[i].w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int o2 = 4; o2 < 64; o2 <<= 1) sm[j] += __shfl_xor(sm[j], o2);
    }
    if (lane < 4) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wave][cq + j] = sm[j];
    }
    __syncthreads();
    if (tid < 32 && cg * 32 + tid < p.C) unsafeAtomicAdd(p.colsum + cg * 32 + tid, (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
  }
}

__global__ __launch_bounds__(256) void k_pack_operands(PackBatch pb) {
  __shared__ float red[4][32];
  int i = 0;
#pragma unroll 1
  while (i + 1 < pb.n && (int)blockIdx.x >= pb.first_block[i + 1]) ++i;
  const PackParams& p = pb.d[i];
  const int local = blockIdx.x - pb.first_block[i];
  const int rb = local / p.ncg, cg = local - rb * p.ncg;
  const bool sum = p.colsum != nullptr;
  if (p.prec == CRAFT_PREC_F16X3) pack_block<CRAFT_PREC_F16X3>(p, rb, cg, sum, red);
  else if (p.prec == CRAFT_PREC_F16) pack_block<CRAFT_PREC_F16>(p, rb, cg, sum, red);
  else pack_block<CRAFT_PREC_BF16>(p, rb, cg, sum, red);
}

static int fill_pack_params(PackParams& p, const float* x, long ldx, int C, long rows, int B, int H, int W, int padH, int padW, long guard, long rows_p,
                            int prec, void* out, int cg_off, int ncg_total, float* colsum, int tail) {
  if (rows_p <= 0 || C <= 0) return CRAFT_ERR_ARG;
  if (cg_off < 0 || cg_off + (C + 31) / 32 > ncg_total) return CRAFT_ERR_ARG;
  if ((C & 3) || (ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return CRAFT_ERR_ALIGN;
  if (prec != CRAFT_PREC_F16X3 && prec != CRAFT_PREC_F16 && prec != CRAFT_PREC_BF16) return CRAFT_ERR_UNSUPPORTED;
  if (tail < 0 || (B <= 0 && tail != 0)) return CRAFT_ERR_ARG;
  if (B > 0 && (double)B * (H + 2 * padH) * (W + 2 * padW + tail) >= 2147483648.0) return CRAFT_ERR_UNSUPPORTED;
  p = PackParams{};
  p.x = x; p.ldx = ldx; p.C = C; p.rows = rows; p.B = B; p.H = H; p.W = W; p.padH = padH; p.padW = padW; p.guard = guard; p.rows_p = rows_p;
  p.out = static_cast<unsigned short*>(out); p.prec = prec; p.colsum = colsum; p.ncg = (C + 31) / 32;
  p.cg_off = cg_off; p.ncg_total = ncg_total; p.tail = tail;
  return 0;
}

// descs: n x 17 longs (x, ldx, C, rows, B, H, W, padH, padW, guard, rows_p, prec, out, cg_off, ncg_total, colsum, tail), see craft_pack_operands
int launch_pack_operands(const long* descs, int n, hipStream_t s) {
  for (int i0 = 0; i0 < n; i0 += PACK_MAX_DESC) {
    PackBatch pb = {};
    pb.n = n - i0 < PACK_MAX_DESC ? n - i0 : PACK_MAX_DESC;
    int blocks = 0;
    for (int i = 0; i < pb.n; ++i) {
      const long* d = descs + (long)(i0 + i) * 17;
      const int rc = fill_pack_params(pb.d[i], reinterpret_cast<const float*>(d[0]), d[1], (int)d[2], d[3], (int)d[4], (int)d[5], (int)d[6], (int)d[7],
                                      (int)d[8], d[9], d[10], (int)d[11], reinterpret_cast<void*>(d[12]), (int)d[13], (int)d[14],
                                      reinterpret_cast<float*>(d[15]), (int)d[16]);
      if (rc) return rc;
      pb.first_block[i] = blocks;
      blocks += (int)((d[10] + PACK_ROWS - 1) / PACK_ROWS) * pb.d[i].ncg;
    }
    pb.first_block[pb.n] = blocks;
    hipLaunchKernelGGL(k_pack_operands, dim3((unsigned)blocks), dim3(256), 0, s, pb);
    { const hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; }
  }
  return 0;
}

int launch_pack_operand(const float* x, long ldx, int C, long rows, int B, int H, int W, int padH, int padW, long guard, long rows_p,
                        int prec, void* out, int cg_off, int ncg_total, float* colsum, int tail, hipStream_t s) {
  if (rows_p <= 0 || C <= 0) return 0;
  const long d[17] = {(long)reinterpret_cast<uintptr_t>(x), ldx, C, rows, B, H, W, padH, padW, guard, rows_p, prec, (long)reinterpret_cast<uintptr_t>(out),
                      cg_off, ncg_total, (long)reinterpret_cast<uintptr_t>(colsum), tail};
  return launch_pack_operands(d, 1, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// C[m][g][n'] += sum_k A[k][m] * B[k + shift(g)][g, n']   over packed operands, split-K
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PK_MAX_SEG = 16;
struct PkParams {
  // K is the concatenation of nseg segments (the calls of one layer in the 12 refinement iterations: ONE launch, one atomic epilogue
  // per pass instead of twelve): segment s = packs (A[s], B[s]) of identical geometry, seg_splits K ranges each
  const unsigned char* A[PK_MAX_SEG]; const unsigned char* B[PK_MAX_SEG]; float* C;
  const unsigned char* B1[PK_MAX_SEG];       // optional second pack of the B operand: channel groups >= ncg_b0 come from it (a virtual cat)
  int ncg_b0; unsigned b1_plane;
  int nseg, seg_splits;
  unsigned a_plane, a_cg, b_plane, b_cg;     // byte strides of the packs (cg stride = rows_p * 64)
  int ncg_a, ncg_b;                          // channel groups of A (M / 32) and of B per tap (cin / 32)
  long a_row0, b_row0;                       // row of k = 0 in each pack (guard rows in front)
  int KH, KW, Wp;                            // tap t: B rows shifted by (t / KW - KH / 2) * Wp + (t % KW - KW / 2)
  long ldc;                                  // C[m * ldc + t * cin + ci]   (cin = 32 * ncg_b)
  int ksplit, kchunk;                        // K rows per split (multiple of 32)
  long K;
  int ntile_m, ntile_n, ngroups;             // ngroups = taps * ncg_b: 32-wide column groups of the flattened N
  int mode;                                  // developer ablations (CRAFT_PK_MODE), 0 in production
};

// PLANES: 2 = f16x3 (hi, lo planes of both operands: al.bh + ah.bl + ah.bh), 1 = one 16-bit plane each.  PLANES_B = 1 with PLANES = 2: the B
// operand (the activation X of a weight gradient) is a single fp16 plane -- X rounded to fp16, dY exact: al.b + ah.b, two MFMAs per product
// (relative error of dW 2e-4 against 2e-5: the library's "mixed" training policy, craft_amd.hip.Precision role `wgx`).
template <int PLANES, int BM, int BN, int WM, int WN, bool BF16, int PLANES_B = PLANES>
__global__ __launch_bounds__(512) void k_gemm_pk(PkParams p) {
  constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
  constexpr int A_CH = PLANES * (BM / 32), B_CH = PLANES_B * (BN / 32);        // 2 KiB chunks per stage
  constexpr int STAGE = (A_CH + B_CH) * 2048;
  constexpr int NDMA = 2 * (A_CH + B_CH), DPW = (NDMA + 7) / 8;                 // 1 KiB DMA pieces per stage / per wave
  __shared__ __attribute__((aligned(1024))) unsigned char S[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware map (as k_conv_wgrad): XCD x owns the K ranges x, x + 8, ... and walks all output tiles of one range back to back
  const int ntile = p.ntile_m * p.ntile_n;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int split = xcd + 8 * (jj / ntile), tile = jj - (jj / ntile) * ntile;
  if (split >= p.ksplit) return;
  const int tm = tile % p.ntile_m, tn = tile / p.ntile_m;
  const int seg = split / p.seg_splits;
  const long k0 = (long)(split - seg * p.seg_splits) * p.kchunk;
  const unsigned char* const Aseg = p.A[seg];
  const unsigned char* const Bseg = p.B[seg];
  const unsigned char* const B1seg = p.B1[seg];
  const int nk = (int)((min(p.K, k0 + p.kchunk) - k0) / 32);
  if (nk <= 0) return;

  // ---- DMA plan of this wave: DPW pieces per stage, each (operand, plane, channel group, half) -> scalar byte offset.
  // The LDS-DMA loads are issued from inline asm: through the builtin hipcc orders every later ds_read behind the pending LDS write
  // (s_waitcnt vmcnt(0) in front of the first fragment read of each K-tile, i.e. no overlap at all); an asm load is invisible to
  // its counters, the waits are placed by hand (vmcnt(0) + barrier at the top of the K loop).
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // one buffer descriptor per piece: its 48-bit base already carries (operand, plane, channel group, half, tap shift), the common
  // scalar offset is the K-tile (a negative tap shift stays inside the pack thanks to the guard rows)
  unsigned blo[DPW], bhi[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int id = wave * DPW + i, ch = id >> 1, half = id & 1;
    const unsigned char* base;
    if (ch < A_CH) {
      const int pl = ch / (BM / 32), cg = min(tm * (BM / 32) + ch % (BM / 32), p.ncg_a - 1);
      base = Aseg + (p.a_row0 + k0) * 64 + (long)pl * p.a_plane + (long)cg * p.a_cg + half * 1024;
    } else {
      const int c2 = min(ch - A_CH, B_CH - 1);
      const int pl = c2 / (BN / 32), g = min(tn * (BN / 32) + c2 % (BN / 32), p.ngroups - 1);
      const int tap = g / p.ncg_b, cg = g - tap * p.ncg_b;
      const int shift = (tap / p.KW - p.KH / 2) * p.Wp + (tap % p.KW - p.KW / 2);
      base = cg < p.ncg_b0 ? Bseg + (long)pl * p.b_plane + (long)cg * p.b_cg : B1seg + (long)pl * p.b1_plane + (long)(cg - p.ncg_b0) * p.b_cg;
      base += (p.b_row0 + k0 + shift) * 64 + half * 1024;
    }
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    blo[i] = __builtin_amdgcn_readfirstlane((unsigned)a);
    bhi[i] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);              // stride field 0
  }
  const unsigned voff = lane * 16;
  const unsigned lds0 = (unsigned)(size_t)(CRAFT_LDS unsigned char*)(S);                 // LDS byte address of the tile buffers
  auto dma = [&](int stage, int kt) __attribute__((always_inline)) {
    const unsigned soff = (unsigned)kt * 2048u;
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int id = wave * DPW + i;
      if (NDMA % 8 != 0 && id >= NDMA) break;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(stage * STAGE + id * 1024));
      u32x4 d;
      d[0] = blo[i]; d[1] = bhi[i]; d[2] = 0xffffffffu; d[3] = 0x00020000u;                // raw dword buffer, no range limit
      unsigned keep;
      // M0 = LDS destination of the wave's 1 KiB (written and read in ONE statement: hipcc owns M0 otherwise); s_nop 4: an SGPR
      // operand fresh from v_readfirstlane needs 5 wait states before a buffer instruction reads it
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(dst), "s"(d), "s"(soff) : "memory");
    }
  };

  // ---- fragment addresses.  MFMA 32x32x16 operand: lane -> column (lane & 31), k half h = lane >> 5.  One ds_read_b64_tr_b16 gives
  // a lane 4 consecutive k of its column: the lanes of a 16-lane group supply the addresses of a [4 k][16 columns] block (lane j:
  // row j >> 2, columns 4 (j & 3) ..) and receive column j.  k-slot assignment (the same for A and B, so any order is legal):
  // read 0 -> k = 4 h .. 4 h + 3, read 1 -> k = 8 + 4 h ..
  const int wm = wave / WN, wn = wave - wm * WN;
  const int j16 = lane & 15, mh = (lane >> 4) & 1, h = lane >> 5;
  const unsigned lane_off = (unsigned)((4 * h + (j16 >> 2)) * 64 + (16 * mh + 4 * (j16 & 3)) * 2);
  const unsigned a_base = lane_off + (unsigned)(wm * MT) * 2048u;                          // + plane * (BM/32) * 2048 + mt * 2048
  const unsigned b_base = lane_off + (unsigned)(A_CH + wn * NT) * 2048u;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  auto frag = [&](unsigned addr) __attribute__((always_inline)) {
    if constexpr (BF16) {
      const bf16x4_raw r0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((CRAFT_LDS bf16x4_raw*)((CRAFT_LDS unsigned char*)(S) + addr));
      const bf16x4_raw r1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((CRAFT_LDS bf16x4_raw*)((CRAFT_LDS unsigned char*)(S) + addr + 512));
      bf16x8 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = r0[i]; v[4 + i] = r1[i]; }
      return v;
    } else {
      const fp16x4_raw r0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((CRAFT_LDS fp16x4_raw*)((CRAFT_LDS unsigned char*)(S) + addr));
      const fp16x4_raw r1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((CRAFT_LDS fp16x4_raw*)((CRAFT_LDS unsigned char*)(S) + addr + 512));
      f16x8 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = (_Float16)r0[i]; v[4 + i] = (_Float16)r1[i]; }
      return v;
    }
  };
  typedef typename std::conditional<BF16, bf16x8, f16x8>::type frag_t;
  auto mma = [&](const frag_t& a, const frag_t& b, f32x16& c) __attribute__((always_inline)) {
    if constexpr (BF16) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  };

  dma(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt have landed ...
    __syncthreads();                     // ... and every other wave's; nobody reads the other stage any more
    if (kt + 1 < nk && !(p.mode & 1)) dma(st ^ 1, kt + 1);
    const unsigned sb = (unsigned)st * STAGE;
    if (p.mode & 4) continue;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag_t ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        ah[mt] = frag(sb + a_base + mt * 2048u + ks * 1024u);
        if constexpr (PLANES == 2) al[mt] = frag(sb + a_base + (BM / 32 + mt) * 2048u + ks * 1024u);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bh[nt] = frag(sb + b_base + nt * 2048u + ks * 1024u);
        if constexpr (PLANES_B == 2) bl[nt] = frag(sb + b_base + (BN / 32 + nt) * 2048u + ks * 1024u);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (PLANES == 2) mma(al[mt], bh[nt], acc[mt][nt]);
          if constexpr (PLANES_B == 2) mma(ah[mt], bl[nt], acc[mt][nt]);
          mma(ah[mt], bh[nt], acc[mt][nt]);
        }
    }
  }

  // ---- epilogue: split-K partial tile added into C with fp32 atomics (C/D layout: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 h)
  const int cin = p.ncg_b * 32, M = p.ncg_a * 32;
  if (p.mode & 2) { if (acc[0][0][0] == 123.456f) p.C[0] = 1.f; return; }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int g = tn * (BN / 32) + wn * NT + nt;
    if (g >= p.ngroups) continue;
    const int tap = g / p.ncg_b, cg = g - tap * p.ncg_b;
    float* cbase = p.C + (long)tap * cin + cg * 32 + (lane & 31);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int mrow = tm * BM + (wm * MT + mt) * 32 + 4 * h;
      if (mrow >= M) continue;                     // (whole 32-row groups are in or out: M % 32 == 0)
#pragma unroll
      for (int e = 0; e < 16; ++e) unsafeAtomicAdd(cbase + (long)(mrow + (e & 3) + 8 * (e >> 2)) * p.ldc, acc[mt][nt][e]);
    }
  }
}

static int pick_tile(int m) { return m > 128 ? 256 : (m > 64 ? 128 : 64); }

int launch_wgrad_pk(const void* const* dYp, const void* const* Xp, const void* const* Xp1, int cin0, int nseg, long dy_rows_p, int cout, long x_rows_p,
                    int cin, long guard, long K, int KH, int KW, int Wp, float* dW, int prec, int prec_x, hipStream_t s) {
  if (cout <= 0 || cin <= 0 || K <= 0 || nseg <= 0) return 0;
  if (prec_x != prec && !(prec == CRAFT_PREC_F16X3 && prec_x == CRAFT_PREC_F16)) return CRAFT_ERR_UNSUPPORTED;
  if ((cout & 31) || (cin & 31) || (K & 31) || KH < 1 || KW < 1) return CRAFT_ERR_ARG;
  if (Xp1 == nullptr) cin0 = cin;
  if ((cin0 & 31) || cin0 <= 0 || cin0 > cin || (Xp1 != nullptr && cin0 == cin)) return CRAFT_ERR_ARG;
  if (prec != CRAFT_PREC_F16X3 && prec != CRAFT_PREC_F16 && prec != CRAFT_PREC_BF16) return CRAFT_ERR_UNSUPPORTED;
  if (nseg > PK_MAX_SEG) {                              // more segments than one launch carries: in groups
    for (int i = 0; i < nseg; i += PK_MAX_SEG) {
      const int n = nseg - i < PK_MAX_SEG ? nseg - i : PK_MAX_SEG;
      const int rc = launch_wgrad_pk(dYp + i, Xp + i, Xp1 ? Xp1 + i : nullptr, cin0, n, dy_rows_p, cout, x_rows_p, cin, guard, K, KH, KW, Wp, dW, prec, prec_x, s);
      if (rc) return rc;
    }
    return 0;
  }
  const int planes = prec == CRAFT_PREC_F16X3 ? 2 : 1, planes_x = prec_x == CRAFT_PREC_F16X3 ? 2 : 1;
  const long max_shift = (long)(KH / 2) * Wp + KW / 2;
  if (guard < max_shift || guard + K + max_shift > x_rows_p || guard + K > dy_rows_p) return CRAFT_ERR_ARG;
  if ((double)planes * (cout / 32) * dy_rows_p * 64.0 >= 4294967296.0 || (double)planes_x * (cin / 32) * x_rows_p * 64.0 >= 4294967296.0)
    return CRAFT_ERR_UNSUPPORTED;                     // 32-bit byte offsets inside a pack
  PkParams p = {};
  for (int i = 0; i < nseg; ++i) {
    p.A[i] = static_cast<const unsigned char*>(dYp[i]);
    p.B[i] = static_cast<const unsigned char*>(Xp[i]);
    p.B1[i] = Xp1 ? static_cast<const unsigned char*>(Xp1[i]) : p.B[i];
    if (!p.A[i] || !p.B[i] || !p.B1[i]) return CRAFT_ERR_ARG;
  }
  p.C = dW; p.nseg = nseg;
  p.ncg_a = cout / 32; p.ncg_b = cin / 32;
  p.a_cg = (unsigned)(dy_rows_p * 64); p.a_plane = (unsigned)(p.ncg_a * dy_rows_p * 64);
  p.ncg_b0 = cin0 / 32;
  p.b_cg = (unsigned)(x_rows_p * 64); p.b_plane = (unsigned)(p.ncg_b0 * x_rows_p * 64); p.b1_plane = (unsigned)((p.ncg_b - p.ncg_b0) * x_rows_p * 64);
  p.a_row0 = guard; p.b_row0 = guard; p.KH = KH; p.KW = KW; p.Wp = Wp;
  p.ldc = (long)KH * KW * cin; p.K = K;
  p.ngroups = KH * KW * p.ncg_b;
  const int bm = pick_tile(cout);
  const int ntot = p.ngroups * 32;
  int bn = ntot > 128 ? 256 : (ntot > 64 ? 128 : 64);
  if (bm == 64 && bn < 128) bn = 128;                  // smallest instantiation: 64 x 128
  p.ntile_m = (cout + bm - 1) / bm;
  p.ntile_n = (ntot + bn - 1) / bn;
  const long tiles = (long)p.ntile_m * p.ntile_n;
  // Blocks of one K range share an XCD (block map of the kernel), an XCD has 32 CUs and a CU holds `bpc` blocks (LDS: two stages of
  // planes * (bm + bn) * 64 bytes per K-tile): the largest split count whose per-XCD share still runs in ONE round -- one split too many
  // and two XCDs run a second round (26 splits x 10 tiles measured 2 x the time of 24)
  const int lds = 2 * (planes * bm + planes_x * bn) * 64;
  const int bpc = lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : (lds > 40 * 1024 ? 3 : 4));
  long per_xcd = (32L * bpc) / tiles;                   // K ranges per XCD
  if (per_xcd < 1) per_xcd = 1;
  const long maxs = (K + 255) / 256;
  long ks = (8 * per_xcd) / nseg;                       // K ranges per segment
  if (ks > maxs) ks = maxs;
  if (ks < 1) ks = 1;
  p.kchunk = (int)((((K + ks - 1) / ks) + 31) / 32 * 32);
  p.seg_splits = (int)((K + p.kchunk - 1) / p.kchunk);
  p.ksplit = p.seg_splits * nseg;
  p.mode = tuning().pk_mode;
  dim3 grid((unsigned)(8 * tiles * ((p.ksplit + 7) / 8)));
#define GO2(PL, BM_, BN_, WM_, WN_, BF, PB) hipLaunchKernelGGL((k_gemm_pk<PL, BM_, BN_, WM_, WN_, BF, PB>), grid, dim3(512), 0, s, p)
#define GO(PL, BF, PB) do { \
    if (bm == 256 && bn == 256) GO2(PL, 256, 256, 2, 4, BF, PB); \
    else if (bm == 256 && bn == 128) GO2(PL, 256, 128, 4, 2, BF, PB); \
    else if (bm == 256) GO2(PL, 256, 64, 8, 1, BF, PB); \
    else if (bm == 128 && bn == 256) GO2(PL, 128, 256, 2, 4, BF, PB); \
    else if (bm == 128 && bn == 128) GO2(PL, 128, 128, 2, 4, BF, PB); \
    else if (bm == 128) GO2(PL, 128, 64, 4, 2, BF, PB); \
    else if (bn == 256) GO2(PL, 64, 256, 1, 8, BF, PB); \
    else GO2(PL, 64, 128, 2, 4, BF, PB); } while (0)
  if (prec == CRAFT_PREC_F16X3 && prec_x == CRAFT_PREC_F16) GO(2, false, 1);
  else if (prec == CRAFT_PREC_F16X3) GO(2, false, 2);
  else if (prec == CRAFT_PREC_F16) GO(1, false, 1);
  else GO(1, true, 1);
#undef GO
#undef GO2
  return (int)hipGetLastError();
}

}  // namespace craft
