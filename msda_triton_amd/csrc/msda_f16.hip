// msda_f16.hip — C-ABI entry points msda_fwd_f16 / msda_bwd_f16 (storage type _Float16).
#include "msda_launch.hpp"

MSDA_DEFINE_ENTRY_POINTS(f16, _Float16)
