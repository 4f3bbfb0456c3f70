// attention-probability kernels for per-mode width d = 64 (see attn_probs.inc.hpp)
#include "attn_probs.inc.hpp"
namespace craft {
template int launch_attn_probs_d<64>(const ScoreParams&, void*, long, int, int, hipStream_t);
}
