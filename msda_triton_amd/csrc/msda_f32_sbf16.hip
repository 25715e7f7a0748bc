// msda_f32_sbf16.hip — C-ABI entry points msda_{fwd,bwd}_fused_f32_sbf16: the module's kernels with value, projection,
// out and their gradients stored as __bf16, reference points (and their gradient partials) and all arithmetic in float.
#include "msda_launch.hpp"

MSDA_DEFINE_FUSED_STORAGE_ENTRY_POINTS(f32_sbf16, float, __bf16)
