// msda_f32_sf16.hip — C-ABI entry points msda_{fwd,bwd}_fused_f32_sf16: as msda_f32_sbf16.hip with _Float16 storage.
#include "msda_launch.hpp"

MSDA_DEFINE_FUSED_STORAGE_ENTRY_POINTS(f32_sf16, float, _Float16)
