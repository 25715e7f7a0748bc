// msda_f32.hip — C-ABI entry points msda_fwd_f32 / msda_bwd_f32 (storage type float).
#include "msda_launch.hpp"

MSDA_DEFINE_ENTRY_POINTS(f32, float)

// size of the backward workspace (shared by every dtype: the accumulate type decides the record sizes)
extern "C" __attribute__((visibility("hidden"))) int64_t msda_bwd_workspace_bytes_impl(
    int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q, int64_t L, int64_t P, int elem_size)
{
    return (int64_t)msda::sorted_ws_layout(B, I, H, D, Q, L, P, elem_size == 8 ? 8 : 4).total;
}
