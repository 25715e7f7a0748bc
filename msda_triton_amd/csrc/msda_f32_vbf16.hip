// msda_f32_vbf16.hip — C-ABI entry points msda_*_f32_vbf16: value / grad_value stored as __bf16, everything else float.
#include "msda_launch.hpp"

MSDA_DEFINE_ENTRY_POINTS2(f32_vbf16, float, __bf16)
