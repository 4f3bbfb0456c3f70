// attention-probability kernels for per-mode width d = 128 (see attn_probs.inc.hpp)
#include "attn_probs.inc.hpp"
namespace craft {
template int launch_attn_probs_d<128>(const ScoreParams&, void*, long, int, int, hipStream_t);
}
