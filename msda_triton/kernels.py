"""Shim for ``msda_triton.kernels`` — the launcher pair that is the native seam
(reference: src/msda_triton/kernels.py:351-379 and :556-592)."""
from msda_triton_amd.functional import msda_hip_bwd, msda_hip_fwd

triton_multi_scale_deformable_attention_fwd = msda_hip_fwd


def triton_multi_scale_deformable_attention_bwd(out_grad, img, img_shapes, sampling_points, attention_weights,
                                                padding_mode, align_corners):
    return msda_hip_bwd(out_grad, img, img_shapes, sampling_points, attention_weights, padding_mode, align_corners)
