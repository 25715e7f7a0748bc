"""Adapter for Hugging Face Deformable-DETR / Grounding-DINO style models.

Those models call a parameter-free ``MultiScaleDeformableAttention`` module
(``transformers/models/{deformable_detr,grounding_dino,...}/modeling_*.py``) with

    forward(value, value_spatial_shapes, value_spatial_shapes_list, level_start_index,
            sampling_locations, attention_weights, im2col_step) -> [batch, queries, heads * head_dim]

which is this operator with ``padding_mode="zeros"``, ``align_corners=False`` followed by a flatten of
the (head, channel) axes — the parity demo of the reference's README (README.md:25-37).
``level_start_index`` and ``im2col_step`` are accepted and ignored (level starts are derived from the
shapes in-kernel, exactly as the reference does, kernels.py:58-62).
"""
from __future__ import annotations

import torch
from torch import nn

from .functional import _autocast_on, multiscale_deformable_attention


class MultiScaleDeformableAttention(nn.Module):
    """Drop-in for the HF module of the same name; runs the MI355X HIP kernels on GPU tensors."""

    def forward(self, value: torch.Tensor, value_spatial_shapes: torch.Tensor, value_spatial_shapes_list=None,
                level_start_index=None, sampling_locations: torch.Tensor = None,
                attention_weights: torch.Tensor = None, im2col_step: int = 64) -> torch.Tensor:
        shapes = value_spatial_shapes
        if not torch.is_tensor(shapes):
            shapes = torch.as_tensor(shapes if shapes is not None else value_spatial_shapes_list,
                                     dtype=torch.int64, device=value.device)
        elif shapes.device != value.device:
            shapes = shapes.to(value.device)
        # the level sizes as host numbers, when the model carries them (transformers >= 4.46 passes
        # `spatial_shapes_list`): lets the backward size its single-launch grad_value kernel for the real levels
        # (decoder layers at image size: 1.4x faster there); never read back from the device
        level_shapes = value_spatial_shapes_list if isinstance(value_spatial_shapes_list, (list, tuple)) else None
        dtype = value.dtype
        if value.device.type == "cuda" and dtype in (torch.bfloat16, torch.float16) and \
                sampling_locations.dtype == torch.float32 and attention_weights.dtype == torch.float32:
            # what autocast hands this module: a 16-bit value pyramid (from the value projection) next to fp32
            # sampling locations / softmaxed weights.  The mixed-storage kernels read the pyramid as it is and keep
            # the coordinates in fp32 (casting them to 16 bits would cost a quarter pixel on a 64-px level; casting
            # everything to fp32, which autocast's policy for the plain operator does, copies the pyramid).
            autocast = _autocast_on()
            with torch.autocast("cuda", enabled=False):
                out = multiscale_deformable_attention(value, shapes, sampling_locations, attention_weights, "zeros", False,
                                                      level_shapes=level_shapes)
            return (out if autocast else out.to(dtype)).flatten(2)
        if sampling_locations.dtype != dtype:
            sampling_locations = sampling_locations.to(dtype)
        if attention_weights.dtype != dtype:
            attention_weights = attention_weights.to(dtype)
        out = multiscale_deformable_attention(value, shapes, sampling_locations, attention_weights, "zeros", False,
                                              level_shapes=level_shapes)
        return out.flatten(2)


def replace_hf_msda(model: nn.Module) -> int:
    """Swap every HF ``MultiScaleDeformableAttention`` submodule of ``model`` for the adapter.
    Returns the number of modules replaced."""
    count = 0
    for parent in model.modules():
        for name, child in list(parent.named_children()):
            if type(child).__name__ == "MultiScaleDeformableAttention" and not isinstance(child, MultiScaleDeformableAttention):
                setattr(parent, name, MultiScaleDeformableAttention())
                count += 1
    return count
