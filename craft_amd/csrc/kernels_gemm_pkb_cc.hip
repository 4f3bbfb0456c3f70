// craft_gemm_pk, operand kinds (CH, CH) -- dP = dO V^T, S = Q K^T (see gemm_pkb.inc.hpp)
#include "gemm_pkb.inc.hpp"

namespace craft {
int launch_gemm_pkb_cc(PkbParams& p, int prec, hipStream_t s) { return launch_gemm_pkb_kind<1, 1>(p, prec, s); }
}  // namespace craft
