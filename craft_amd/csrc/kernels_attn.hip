// Multi-mode attention scores on the GEMM engine (CrossAttFeatTrans, setrans.py:501-566):
//   k_corr_build<MAXONLY=false> : inter-frame correlation volume  c(i,j) = sum_m s_m softmax_m(w s_m) + pw*pb(i,j)
//                                 written straight into pyramid level 0, plus per-sample (sum, sum^2) for the
//                                 lazy global LayerNorm (corr.py:200-204).  The 4-mode score tensor never exists.
//   k_corr_build<MAXONLY=true>  : global max of the raw scaled scores -> clamp decision (setrans.py:520-529)
//   (k_attn_probs, the softmax branch, lives in attn_probs.inc.hpp)
#include "gemm_engine.hpp"
#include "launch.hpp"

namespace craft {

// ---------------------------------------------------------------------------------------------
// correlation build / score max.  grid (query tiles of 128, key tiles of 64, B)
// ---------------------------------------------------------------------------------------------
template <int PREC, bool MAXONLY>
__global__ __launch_bounds__(NTHREADS) void k_corr_build(ScoreParams p, float w_aggr, float* __restrict__ pyr0,
                                                        double* __restrict__ sums, unsigned* __restrict__ max_ord) {
  constexpr int BM = 128, BN = 64, WM = 2, WN = 2, MT = 2, NT = 1;
  __shared__ int s_rh[BM], s_rw[BM];
  __shared__ float s_tab[961];
  __shared__ float s_red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, b = blockIdx.z;
  const int N = p.N;
  const int K = p.M * p.d, tpm = p.d / BK, nk = K / BK;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  const int c_lane = lane & 31, rh4 = 4 * (lane >> 5);
  const int col = n0 + wn0 + c_lane;
  const bool clamp = (!MAXONLY) && p.clamp_ord != nullptr && ord2f(*p.clamp_ord) > CRAFT_ATTN_CLIP;
  if (MAXONLY) {
    // Cauchy-Schwarz bound from k_norm_bound: max_ij q_m(i).k_m(j) <= max_i|q_m(i)| * max_j|k_m(j)|.  If even that
    // can not exceed the clip threshold, no score can: leave max_ord = 0 ("no clamp") and skip the exact pass.
    float bound = 0.f;
    for (int m = 0; m < p.M; ++m) bound = fmaxf(bound, __uint_as_float(max_ord[1 + m]) * __uint_as_float(max_ord[1 + 8 + m]));
    if (bound * p.scale * 1.0001f <= CRAFT_ATTN_CLIP) return;
  }

  if (!MAXONLY) {
    if (tid < BM) { const int r = m0 + tid; s_rh[tid] = r / p.W8; s_rw[tid] = r - (r / p.W8) * p.W8; }
    if (p.pos_tab) { const int T = (2 * p.R + 1) * (2 * p.R + 1); for (int i = tid; i < T; i += NTHREADS) s_tab[i] = p.pos_tab[i]; }
  }

  LoaderRowsF32<BM> la;
  la.init(p.Q + (long)b * p.q_bs, p.ldq, m0, N, K, tid);
  LoaderRowsF32<BN> lb;
  lb.init(p.Kf + (long)b * p.k_bs, p.ldk, n0, N, K, tid);

  f32x16 acc[MT][NT];
  acc_zero(acc);
  float mx[MT][16], den[MT][16], num[MT][16];
  float vmax = -INFINITY;
  constexpr bool kExact = (PREC == CRAFT_PREC_F32);
  const float wl2 = w_aggr * 1.4426950408889634f;

  auto fold = [&](int kt) {
    if ((kt + 1) % tpm != 0) return;
    const bool first = (kt + 1 == tpm);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float s = acc[mt][0][e] * p.scale;
        acc[mt][0][e] = 0.f;
        if (MAXONLY) {
          const int row = m0 + wm0 + mt * 32 + (e & 3) + 8 * (e >> 2) + rh4;
          if (row < N && col < N) vmax = fmaxf(vmax, s);
        } else {
          if (clamp) s = fminf(fmaxf(s, -CRAFT_ATTN_CLIP), CRAFT_ATTN_CLIP);
          // online softmax over modes of t = w_aggr * s.  Non-fp32 modes track it in the base-2 domain (w2 = w_aggr *
          // log2 e, v_exp_f32 directly): ~12 VALU per (element, mode) instead of ~50 with two libm expf -- this
          // epilogue, not the MFMAs, bounds the kernel.
          const float t = (kExact ? w_aggr : wl2) * s;
          if (first) { mx[mt][e] = t; den[mt][e] = 1.f; num[mt][e] = s; }
          else {
            const float nm = fmaxf(mx[mt][e], t);
            const float e0 = kExact ? expf(mx[mt][e] - nm) : __builtin_amdgcn_exp2f(mx[mt][e] - nm);
            const float e1 = kExact ? expf(t - nm) : __builtin_amdgcn_exp2f(t - nm);
            den[mt][e] = den[mt][e] * e0 + e1;
            num[mt][e] = num[mt][e] * e0 + s * e1;
            mx[mt][e] = nm;
          }
        }
      }
  };
  gemm_mainloop<PREC, BM, BN, WM, WN>(la, lb, nk, acc, fold);

  if (MAXONLY) {
    vmax = wave_max(vmax);
    if (lane == 0) s_red[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
      const float v = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
      if (v > -INFINITY) atomicMax(max_ord, f2ord(v));
    }
    return;
  }

  const int h2 = col / p.W8, w2 = col - h2 * p.W8;
  float s1 = 0.f, s2 = 0.f;
  float* out = pyr0 + (long)b * N * N;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rl = wm0 + mt * 32 + (e & 3) + 8 * (e >> 2) + rh4;
      const int row = m0 + rl;
      if (row < N && col < N) {
        float c = num[mt][e] / den[mt][e];
        if (p.pos_tab) {
          const int dh = h2 - s_rh[rl], dw = w2 - s_rw[rl];
          if (dh >= -p.R && dh <= p.R && dw >= -p.R && dw <= p.R) c += p.pos_w * s_tab[(dh + p.R) * (2 * p.R + 1) + dw + p.R];
        }
        out[(long)row * N + col] = c;
        s1 += c;
        s2 += c * c;
      }
    }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) { s_red[wave] = s1; s_red[4 + wave] = s2; }
  __syncthreads();
  if (tid == 0) {
    const double a = (double)s_red[0] + (double)s_red[1] + (double)s_red[2] + (double)s_red[3];
    const double q = (double)s_red[4] + (double)s_red[5] + (double)s_red[6] + (double)s_red[7];
    atomicAdd(&sums[2 * b], a);
    atomicAdd(&sums[2 * b + 1], q);
  }
}

// ---------------------------------------------------------------------------------------------
// correlation build, M = 4 modes (the released configuration): every mode keeps its own accumulator tile, so the
// K loop carries no per-mode fold and the softmax-over-modes pooling c = sum_m s_m softmax_m(w s_m) is computed once
// per element from the four scores (max, 4 x exp2, one reciprocal: ~28 VALU per element instead of 4 x 12 for
// the online form plus its state).  The positional bias comes from a clamped (dh, dw) table with a zero border.
// ---------------------------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(NTHREADS) void k_corr_build4(ScoreParams p, float w_aggr, float* __restrict__ pyr0,
                                                         double* __restrict__ sums) {
  constexpr int BM = 128, BN = 64, WM = 2, WN = 2, MT = 2, NT = 1;
  constexpr bool kExact = (PREC == CRAFT_PREC_F32);
  __shared__ int s_rh[BM], s_rw[BM];
  __shared__ float s_tab[33 * 33];
  __shared__ float s_red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, b = blockIdx.z;
  const int N = p.N, d = p.d;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  const int c_lane = lane & 31, rh4 = 4 * (lane >> 5);
  const int col = n0 + wn0 + c_lane;
  const bool clamp = p.clamp_ord != nullptr && ord2f(*p.clamp_ord) > CRAFT_ATTN_CLIP;
  const int R = p.pos_tab ? p.R : 0, TW = 2 * R + 3;
  if (tid < BM) { const int r = m0 + tid; s_rh[tid] = r / p.W8; s_rw[tid] = r - (r / p.W8) * p.W8; }
  for (int i = tid; i < TW * TW; i += NTHREADS) {
    const int dh = i / TW - R - 1, dw = i - (i / TW) * TW - R - 1;
    s_tab[i] = (p.pos_tab && abs(dh) <= R && abs(dw) <= R) ? p.pos_w * p.pos_tab[(dh + R) * (2 * R + 1) + dw + R] : 0.f;
  }
  f32x16 acc[4][MT][NT];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    acc_zero(acc[m]);
    LoaderRowsF32<BM> la;
    la.init(p.Q + (long)b * p.q_bs + m * d, p.ldq, m0, N, d, tid);
    LoaderRowsF32<BN> lb;
    lb.init(p.Kf + (long)b * p.k_bs + m * d, p.ldk, n0, N, d, tid);
    gemm_mainloop<PREC, BM, BN, WM, WN>(la, lb, d / BK, acc[m], NoFold());     // ends with a barrier
  }
  const float clipv = clamp ? CRAFT_ATTN_CLIP : 3.0e38f;
  const float wl = kExact ? w_aggr : w_aggr * 1.4426950408889634f;
  const int h2 = col / p.W8, w2 = col - h2 * p.W8;
  const int ch = R + 1 + h2, cw = R + 1 + w2;           // (unsigned)(ch - rh) = dh + R + 1, clamped to [0, 2R+2]
  const unsigned umax = 2 * R + 2;
  float s1 = 0.f, s2 = 0.f;
  float* out = pyr0 + (long)b * N * N;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rl = wm0 + mt * 32 + (e & 3) + 8 * (e >> 2) + rh4;
      const int row = m0 + rl;
      float sv[4], tv[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        sv[m] = __builtin_amdgcn_fmed3f(acc[m][mt][0][e] * p.scale, -clipv, clipv);
        tv[m] = wl * sv[m];
      }
      const float mx = fmaxf(fmaxf(tv[0], tv[1]), fmaxf(tv[2], tv[3]));
      float den = 0.f, num = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float ex = kExact ? expf(tv[m] - mx) : __builtin_amdgcn_exp2f(tv[m] - mx);
        den += ex;
        num += sv[m] * ex;
      }
      float c = kExact ? num / den : num * __builtin_amdgcn_rcpf(den);
      const unsigned u = min((unsigned)(ch - s_rh[rl]), umax), v = min((unsigned)(cw - s_rw[rl]), umax);
      c += s_tab[u * TW + v];
      if (row < N && col < N) {
        out[(long)row * N + col] = c;
        s1 += c;
        s2 += c * c;
      }
    }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) { s_red[wave] = s1; s_red[4 + wave] = s2; }
  __syncthreads();
  if (tid == 0) {
    const double a = (double)s_red[0] + (double)s_red[1] + (double)s_red[2] + (double)s_red[3];
    const double q = (double)s_red[4] + (double)s_red[5] + (double)s_red[6] + (double)s_red[7];
    atomicAdd(&sums[2 * b], a);
    atomicAdd(&sums[2 * b + 1], q);
  }
}

static int check_score(const ScoreParams& p) {
  if (p.d % BK || p.M < 1 || (p.ldq & 3) || (p.ldk & 3) || (p.q_bs & 3) || (p.k_bs & 3)) return CRAFT_ERR_ALIGN;
  if (p.pos_tab && p.R > 15) return CRAFT_ERR_UNSUPPORTED;
  return 0;
}

// per-mode max row norms of Q (slots 1..8) and K (slots 9..16) as float bits (non-negative floats order like uints)
__global__ __launch_bounds__(256) void k_norm_bound(const float* __restrict__ Q, long ldq, const float* __restrict__ Kf, long ldk,
                                                    long ntok, int M, int d, unsigned* __restrict__ ws) {
  __shared__ unsigned s_max[16];
  if (threadIdx.x < 16) s_max[threadIdx.x] = 0u;
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;    // one thread per (which, token, mode)
  const long per = ntok * M;
  if (i < 2 * per) {
    const int which = i >= per;
    const long j = which ? i - per : i;
    const long tok = j / M;
    const int m = (int)(j - tok * M);
    const float* p = (which ? Kf + tok * ldk : Q + tok * ldq) + (long)m * d;
    float ss = 0.f;
    for (int c = 0; c < d; c += 4) {
      const float4 v = *reinterpret_cast<const float4*>(p + c);
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    atomicMax(&s_max[which * 8 + m], __float_as_uint(sqrtf(ss) * 1.00001f));
  }
  __syncthreads();
  if (threadIdx.x < 16 && s_max[threadIdx.x]) atomicMax(&ws[1 + threadIdx.x], s_max[threadIdx.x]);
}

int launch_score_max(const ScoreParams& p, unsigned* max_ord, int prec, hipStream_t s) {
  if (int e = check_score(p)) return e;
  if (p.M > 8) return CRAFT_ERR_UNSUPPORTED;
  dim3 grid((p.N + 127) / 128, (p.N + 63) / 64, p.B);
  hipError_t me = hipMemsetAsync(max_ord, 0, 32 * sizeof(unsigned), s);
  if (me != hipSuccess) return (int)me;
  {
    const long ntok = (long)p.B * p.N, tot = 2 * ntok * p.M;
    hipLaunchKernelGGL(k_norm_bound, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, p.Q, p.ldq, p.Kf, p.ldk, ntok, p.M, p.d, max_ord);
  }
  if (prec == CRAFT_PREC_F32) hipLaunchKernelGGL((k_corr_build<CRAFT_PREC_F32, true>), grid, dim3(NTHREADS), 0, s, p, 0.f, nullptr, nullptr, max_ord);
  else if (prec == CRAFT_PREC_BF16) hipLaunchKernelGGL((k_corr_build<CRAFT_PREC_BF16, true>), grid, dim3(NTHREADS), 0, s, p, 0.f, nullptr, nullptr, max_ord);
  else if (prec == CRAFT_PREC_F16) hipLaunchKernelGGL((k_corr_build<CRAFT_PREC_F16, true>), grid, dim3(NTHREADS), 0, s, p, 0.f, nullptr, nullptr, max_ord);
  else if (prec == CRAFT_PREC_F16X3) hipLaunchKernelGGL((k_corr_build<CRAFT_PREC_F16X3, true>), grid, dim3(NTHREADS), 0, s, p, 0.f, nullptr, nullptr, max_ord);
  else return CRAFT_ERR_ARG;
  return (int)hipGetLastError();
}

int launch_corr_build(const ScoreParams& p, float w_aggr, float* pyr0, double* sums, int prec, hipStream_t s) {
  if (int e = check_score(p)) return e;
  dim3 grid((p.N + 127) / 128, (p.N + 63) / 64, p.B);
  hipError_t me = hipMemsetAsync(sums, 0, sizeof(double) * 2 * p.B, s);
  if (me != hipSuccess) return (int)me;
  if (p.M == 4) {
    if (prec == CRAFT_PREC_F32) hipLaunchKernelGGL((k_corr_build4<CRAFT_PREC_F32>), grid, dim3(NTHREADS), 0, s, p, w_aggr, pyr0, sums);
    else if (prec == CRAFT_PREC_BF16) hipLaunchKernelGGL((k_corr_build4<CRAFT_PREC_BF16>), grid, dim3(NTHREADS), 0, s, p, w_aggr, pyr0, sums);
    else if (prec == CRAFT_PREC_F16) hipLaunchKernelGGL((k_corr_build4<CRAFT_PREC_F16>), grid, dim3(NTHREADS), 0, s, p, w_aggr, pyr0, sums);
    else if (prec == CRAFT_PREC_F16X3) hipLaunchKernelGGL((k_corr_build4<CRAFT_PREC_F16X3>), grid, dim3(NTHREADS), 0, s, p, w_aggr, pyr0, sums);
    else return CRAFT_ERR_ARG;
    return (int)hipGetLastError();
  }
  if (prec == CRAFT_PREC_F32) hipLaunchKernelGGL((k_corr_build<CRAFT_PREC_F32, false>), grid, dim3(NTHREADS), 0, s, p, w_aggr, pyr0, sums, nullptr);
  else if (prec == CRAFT_PREC_BF16) hipLaunchKernelGGL((k_corr_build<CRAFT_PREC_BF16, false>), grid, dim3(NTHREADS), 0, s, p, w_aggr, pyr0, sums, nullptr);
  else if (prec == CRAFT_PREC_F16) hipLaunchKernelGGL((k_corr_build<CRAFT_PREC_F16, false>), grid, dim3(NTHREADS), 0, s, p, w_aggr, pyr0, sums, nullptr);
  else if (prec == CRAFT_PREC_F16X3) hipLaunchKernelGGL((k_corr_build<CRAFT_PREC_F16X3, false>), grid, dim3(NTHREADS), 0, s, p, w_aggr, pyr0, sums, nullptr);
  else return CRAFT_ERR_ARG;
  return (int)hipGetLastError();
}

// attention probabilities: kernels live in attn_probs.inc.hpp, one translation unit per mode width d
int launch_attn_probs(const ScoreParams& p, void* P, long ldp, int p_prec, int prec, hipStream_t s) {
  if (int e = check_score(p)) return e;
  if (ldp % 32 || ldp < p.N) return CRAFT_ERR_ALIGN;
  if (p.d == 32) return launch_attn_probs_d<32>(p, P, ldp, p_prec, prec, s);
  if (p.d == 64) return launch_attn_probs_d<64>(p, P, ldp, p_prec, prec, s);
  if (p.d == 128) return launch_attn_probs_d<128>(p, P, ldp, p_prec, prec, s);
  return CRAFT_ERR_UNSUPPORTED;
}

}  // namespace craft
