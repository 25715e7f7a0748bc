// msda_bf16.hip — C-ABI entry points msda_fwd_bf16 / msda_bwd_bf16 (storage type __bf16).
#include "msda_launch.hpp"

MSDA_DEFINE_ENTRY_POINTS(bf16, __bf16)
