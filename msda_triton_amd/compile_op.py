"""``torch.library`` registration of the operator.

The default path (``functional._HipMultiscaleDeformableAttentionFunction``) is an
``autograd.Function`` around ctypes calls, which ``torch.compile`` cannot look into.  Here the two
launchers are registered as custom ops with fake (meta) kernels and an autograd formula, so the
operator survives ``torch.compile(fullgraph=True)`` / ``torch.export`` as one opaque node per direction:

    torch.ops.msda_amd.forward(img, img_shapes, sampling_points, attention_weights, zeros, align_corners)
    torch.ops.msda_amd.backward(out_grad, img, img_shapes, sampling_points, attention_weights, zeros,
                                align_corners, need_value, need_sample)

``compiled_multiscale_deformable_attention`` is the functional entry that uses them;
``functional.multiscale_deformable_attention`` switches to it automatically while being traced.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import functional as F

_PAD = {True: "zeros", False: "border"}


@torch.library.custom_op("msda_amd::forward", mutates_args=(), device_types="cuda")
def msda_forward(img: torch.Tensor, img_shapes: torch.Tensor, sampling_points: torch.Tensor,
                 attention_weights: torch.Tensor, zeros: bool, align_corners: bool) -> torch.Tensor:
    return F.msda_hip_fwd(img, img_shapes, sampling_points, attention_weights, _PAD[zeros], align_corners)


@msda_forward.register_fake
def _(img, img_shapes, sampling_points, attention_weights, zeros, align_corners):
    B, _, H, D = img.shape
    return img.new_empty((B, sampling_points.shape[1], H, D))


@torch.library.custom_op("msda_amd::backward", mutates_args=(), device_types="cuda")
def msda_backward(out_grad: torch.Tensor, img: torch.Tensor, img_shapes: torch.Tensor, sampling_points: torch.Tensor,
                  attention_weights: torch.Tensor, zeros: bool, align_corners: bool, need_value: bool,
                  need_sample: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    g_img, g_pts, g_att = F.msda_hip_bwd(out_grad, img, img_shapes, sampling_points, attention_weights, _PAD[zeros],
                                         align_corners, (need_value, need_sample, need_sample))
    # custom ops cannot return None: unneeded gradients come back as empty tensors
    return (g_img if g_img is not None else img.new_empty(0),
            g_pts if g_pts is not None else img.new_empty(0),
            g_att if g_att is not None else img.new_empty(0))


@msda_backward.register_fake
def _(out_grad, img, img_shapes, sampling_points, attention_weights, zeros, align_corners, need_value, need_sample):
    return (torch.empty_like(img, memory_format=torch.contiguous_format) if need_value else img.new_empty(0),
            torch.empty_like(sampling_points, memory_format=torch.contiguous_format) if need_sample else img.new_empty(0),
            torch.empty_like(attention_weights, memory_format=torch.contiguous_format) if need_sample else img.new_empty(0))


def _setup_context(ctx, inputs, output):
    img, img_shapes, sampling_points, attention_weights, zeros, align_corners = inputs
    ctx.save_for_backward(img, img_shapes, sampling_points, attention_weights)
    ctx.zeros, ctx.align_corners = zeros, align_corners


def _backward(ctx, out_grad):
    img, img_shapes, sampling_points, attention_weights = ctx.saved_tensors
    need_value = ctx.needs_input_grad[0]
    need_sample = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
    g_img, g_pts, g_att = msda_backward(out_grad.contiguous(), img, img_shapes, sampling_points, attention_weights,
                                        ctx.zeros, ctx.align_corners, need_value, need_sample)
    return (g_img if need_value else None, None, g_pts if ctx.needs_input_grad[2] else None,
            g_att if ctx.needs_input_grad[3] else None, None, None)


msda_forward.register_autograd(_backward, setup_context=_setup_context)


def compiled_multiscale_deformable_attention(img, img_shapes, sampling_points, attention_weights, padding_mode,
                                             align_corners) -> torch.Tensor:
    """Same contract as ``hip_multiscale_deformable_attention`` but through the registered custom ops
    (traceable).  Like the default path, the op computes in fp32 under autocast."""
    F._padding_code(padding_mode)
    if torch.is_autocast_enabled("cuda"):
        with torch.autocast("cuda", enabled=False):
            return msda_forward(img.float(), img_shapes, sampling_points.float(), attention_weights.float(),
                                padding_mode == "zeros", bool(align_corners))
    return msda_forward(img, img_shapes, sampling_points, attention_weights, padding_mode == "zeros", bool(align_corners))
