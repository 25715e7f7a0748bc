// msda_f32.hip — C-ABI entry points msda_fwd_f32 / msda_bwd_f32 (storage type float).
#include "msda_launch.hpp"

MSDA_DEFINE_ENTRY_POINTS(f32, float)

// size of the backward workspace (shared by every dtype: the accumulate type decides the record sizes)
extern "C" __attribute__((visibility("hidden"))) int64_t msda_bwd_workspace_bytes_impl(
    int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q, int64_t L, int64_t P, int elem_size, int records_in_grads,
    int value_elem_size, int64_t max_level_cells, int passes)
{
    // problems the single-launch kernel takes need no workspace at all
    const msda::Dims d{B, I, H, D, Q, L, P, max_level_cells > 0 ? max_level_cells : 0};
    const bool small = elem_size == 8 ? msda::small_path_chosen<double>(d) : msda::small_path_chosen<float>(d);
    const size_t acc = elem_size == 8 ? 8 : 4;
    if (small) return 0;
    // passes over the batch (MSDA_WS_PASSES): the workspace of ceil(B / passes) batch elements, used once per group
    if (passes > 1 && B > 1) B = (B + passes - 1) / passes;
    // the larger of the 16-byte-vector and the scalar layout: which one a call takes depends on the alignment of its
    // grad_out / grad_value pointers (a slice of a shard's buffers can be misaligned), and a workspace that is too
    // small would be rejected (MSDA_ERR_BAD_ARG)
    const bool rg = records_in_grads != 0 && msda::option_records_in_grads() != 0;
    const size_t vec = msda::sorted_ws_layout(B, I, H, D, Q, L, P, acc, (size_t)elem_size, true, rg, (size_t)value_elem_size).total;
    const size_t sca = msda::sorted_ws_layout(B, I, H, D, Q, L, P, acc, (size_t)elem_size, false, rg, (size_t)value_elem_size).total;
    return (int64_t)(vec > sca ? vec : sca);
}

// can grad_value be produced for these sizes at all (include/msda_hip.h: msda_bwd_supported)
extern "C" __attribute__((visibility("hidden"))) int msda_bwd_supported_impl(int64_t B, int64_t I, int64_t H, int64_t D,
                                                                           int64_t Q, int64_t L, int64_t P, int elem_size)
{
    const msda::Dims d{B, I, H, D, Q, L, P};
    if (L > MSDA_MAX_LEVELS) return 0;
    if (B * Q * H * D == 0 || L * P == 0 || I == 0) return 1;  // all-zero gradients
    switch (elem_size) {
    case 8: return msda::sorted_fits<double>(d) || msda::small_fits<double>(d);
    case 2: return msda::sorted_fits<_Float16>(d) || msda::small_fits<_Float16>(d);
    default: return msda::sorted_fits<float>(d) || msda::small_fits<float>(d);
    }
}

// largest L*P the fused-prologue kernels (msda_fwd_fused / msda_bwd_fused) take for this head dimension and
// element size: all records of a unit must sit in LDS at once (plan_gather); the 16-byte vector path and the
// backward's larger records give the smaller bound
extern "C" __attribute__((visibility("hidden"))) int64_t msda_fused_lp_limit_impl(int64_t D, int elem_size)
{
    if (D <= 0 || elem_size <= 0) return 0;
    const size_t acc = elem_size == 8 ? 8 : 4;
    int64_t best = INT64_MAX;
    for (int vec : {16 / elem_size, 1}) {
        const int NU = msda::kBlock / msda::pick_group((int)((D + vec - 1) / vec));
        int sc;
        size_t lds;
        msda::plan_gather(NU, 1 << 22, acc, sc, lds, true);
        if (sc < best) best = sc;
    }
    return best;
}
