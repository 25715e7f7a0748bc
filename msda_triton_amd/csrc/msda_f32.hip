// msda_f32.hip — C-ABI entry points msda_fwd_f32 / msda_bwd_f32 (storage type float).
#include "msda_launch.hpp"

MSDA_DEFINE_ENTRY_POINTS(f32, float)
