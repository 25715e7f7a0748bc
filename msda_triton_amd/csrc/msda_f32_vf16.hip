// msda_f32_vf16.hip — C-ABI entry points msda_*_f32_vf16: value / grad_value stored as _Float16, everything else float.
#include "msda_launch.hpp"

MSDA_DEFINE_ENTRY_POINTS2(f32_vf16, float, _Float16)
