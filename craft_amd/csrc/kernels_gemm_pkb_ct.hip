// craft_gemm_pk, operand kinds (CH, ROWS) -- O = P V, dQ = dS K (see gemm_pkb.inc.hpp)
#include "gemm_pkb.inc.hpp"

namespace craft {
int launch_gemm_pkb_ct(PkbParams& p, int prec, hipStream_t s) { return launch_gemm_pkb_kind<1, 0>(p, prec, s); }
}  // namespace craft
