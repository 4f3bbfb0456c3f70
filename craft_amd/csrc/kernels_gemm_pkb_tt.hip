// craft_gemm_pk, operand kinds (ROWS, ROWS) -- dV = P^T dO, dK = dS^T Q -- and the dispatcher (see gemm_pkb.inc.hpp)
#include "gemm_pkb.inc.hpp"
#include "../../include/craft_hip.h"

namespace craft {

int launch_gemm_pkb_tt(PkbParams& p, int prec, hipStream_t s) { return launch_gemm_pkb_kind<0, 0>(p, prec, s); }

static int fill_operand(PkbOperand& X, const void* ptr, const long* d, int K, int ext) {
  // d = {kind, rows_p, ncg, row0, row_outer, row_inner, cg0, cg_outer, cg_inner}
  if (ptr == nullptr || (d[0] != 0 && d[0] != 1) || d[1] <= 0 || d[2] <= 0 || d[3] < 0 || d[6] < 0) return CRAFT_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(ptr) & 15) return CRAFT_ERR_ALIGN;
  if ((double)d[1] * 64.0 >= 4294967296.0) return CRAFT_ERR_UNSUPPORTED;       // the channel-group stride is a 32-bit K step
  X.base = static_cast<const unsigned char*>(ptr);
  X.cg = (unsigned)(d[1] * 64);
  X.plane = d[2] * d[1] * 64;
  X.row0 = d[3]; X.row_outer = d[4]; X.row_inner = d[5];
  X.cg0 = (int)d[6]; X.cg_outer = (int)d[7]; X.cg_inner = (int)d[8];
  // 32-bit offsets inside a K walk
  if (d[0] == 0 ? (double)K * 64.0 >= 4294967296.0 : (double)(K / 32) * d[1] * 64.0 >= 4294967296.0) return CRAFT_ERR_UNSUPPORTED;
  (void)ext;
  return 0;
}

int launch_gemm_pk(const void* A, const long* a_desc, const void* B, const long* b_desc, float* C, long ldc, long c_outer, long c_inner,
                   int inner, int nbatch, int M, int N, int K, float alpha, int prec, hipStream_t s) {
  if (M <= 0 || N <= 0 || nbatch <= 0) return 0;
  const int cshift = (prec >> CRAFT_PK_CBLK_SHIFT) & 31;          // CRAFT_PK_CBLK(s)
  prec &= (1 << CRAFT_PK_CBLK_SHIFT) - 1;
  if (cshift && (cshift < 5 || c_inner != (1L << cshift))) return CRAFT_ERR_ARG;
  if (K <= 0 || (K & 31) || inner <= 0 || C == nullptr || a_desc == nullptr || b_desc == nullptr) return CRAFT_ERR_ARG;
  if (prec != CRAFT_PREC_F16X3 && prec != CRAFT_PREC_F16 && prec != CRAFT_PREC_BF16) return CRAFT_ERR_UNSUPPORTED;
  PkbParams p = {};
  int rc = fill_operand(p.A, A, a_desc, K, M);
  if (rc) return rc;
  rc = fill_operand(p.B, B, b_desc, K, N);
  if (rc) return rc;
  p.C = C; p.ldc = ldc; p.c_outer = c_outer; p.c_inner = c_inner; p.inner = inner; p.nbatch = nbatch; p.M = M; p.N = N; p.K = K; p.alpha = alpha;
  p.c_blk_shift = cshift; p.c_blk_stride = (long)inner << cshift;
  const int ak = (int)a_desc[0], bk = (int)b_desc[0];
  if (ak == 0 && bk == 0) return launch_gemm_pkb_tt(p, prec, s);
  if (ak == 1 && bk == 0) return launch_gemm_pkb_ct(p, prec, s);
  if (ak == 1 && bk == 1) return launch_gemm_pkb_cc(p, prec, s);
  return CRAFT_ERR_UNSUPPORTED;                     // (ROWS, CH): swap the operands and transpose the output instead
}

}  // namespace craft
