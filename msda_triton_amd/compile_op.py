"""``torch.library`` registration of the operator.

The default path (``functional._HipMultiscaleDeformableAttentionFunction``) is an
``autograd.Function`` around ctypes calls, which ``torch.compile`` cannot look into.  Here the two
launchers are registered as custom ops with fake (meta) kernels and an autograd formula, so the
operator survives ``torch.compile(fullgraph=True)`` / ``torch.export`` as one opaque node per direction:

    torch.ops.msda_amd.forward(img, img_shapes, sampling_points, attention_weights, zeros, align_corners, level_cells)
    torch.ops.msda_amd.backward(out_grad, img, img_shapes, sampling_points, attention_weights, zeros,
                                align_corners, need_value, need_sample, level_cells)

``level_cells`` is ``functional.level_cells_of(level_shapes)`` — a host integer (0: unknown), a constant of the traced
graph — which only the backward reads.

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
                 attention_weights: torch.Tensor, zeros: bool, align_corners: bool, level_cells: int = 0) -> torch.Tensor:
    return F.msda_hip_fwd(img, img_shapes, sampling_points, attention_weights, _PAD[zeros], align_corners)


@msda_forward.register_fake
def _(img, img_shapes, sampling_points, attention_weights, zeros, align_corners, level_cells=0):
    B, _, H, D = img.shape
    return sampling_points.new_empty((B, sampling_points.shape[1], H, D))  # (mixed storage: the result is fp32)


@torch.library.custom_op("msda_amd::backward", mutates_args=(), device_types="cuda")
def msda_backward(out_grad: torch.Tensor, img: torch.Tensor, img_shapes: torch.Tensor, sampling_points: torch.Tensor,
                  attention_weights: torch.Tensor, zeros: bool, align_corners: bool, need_value: bool,
                  need_sample: bool, level_cells: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    g_img, g_pts, g_att = F.msda_hip_bwd(out_grad, img, img_shapes, sampling_points, attention_weights, _PAD[zeros],
                                         align_corners, (need_value, need_sample, need_sample), level_cells=level_cells)
    # custom ops cannot return None: unneeded gradients come back as empty tensors
    return (g_img if g_img is not None else img.new_empty(0),
            g_pts if g_pts is not None else img.new_empty(0),
            g_att if g_att is not None else img.new_empty(0))


@msda_backward.register_fake
def _(out_grad, img, img_shapes, sampling_points, attention_weights, zeros, align_corners, need_value, need_sample,
      level_cells=0):
    return (torch.empty_like(img, memory_format=torch.contiguous_format) if need_value else img.new_empty(0),
            torch.empty_like(sampling_points, memory_format=torch.contiguous_format) if need_sample else img.new_empty(0),
            torch.empty_like(attention_weights, memory_format=torch.contiguous_format) if need_sample else img.new_empty(0))


def _setup_context(ctx, inputs, output):
    img, img_shapes, sampling_points, attention_weights, zeros, align_corners, level_cells = inputs
    ctx.save_for_backward(img, img_shapes, sampling_points, attention_weights)
    ctx.zeros, ctx.align_corners, ctx.level_cells = zeros, align_corners, level_cells


def _backward(ctx, out_grad):
    img, img_shapes, sampling_points, attention_weights = ctx.saved_tensors
    need_value = ctx.needs_input_grad[0]
    need_sample = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
    g_img, g_pts, g_att = msda_backward(out_grad.contiguous(), img, img_shapes, sampling_points, attention_weights,
                                        ctx.zeros, ctx.align_corners, need_value, need_sample, ctx.level_cells)
    return (g_img if need_value else None, None, g_pts if ctx.needs_input_grad[2] else None,
            g_att if ctx.needs_input_grad[3] else None, None, None, None)


msda_forward.register_autograd(_backward, setup_context=_setup_context)


def compiled_multiscale_deformable_attention(img, img_shapes, sampling_points, attention_weights, padding_mode,
                                             align_corners, level_cells: int = 0) -> torch.Tensor:
    """Same contract as ``hip_multiscale_deformable_attention`` but through the registered custom ops
    (traceable).  Like the default path, the op computes in fp32 under autocast."""
    F._padding_code(padding_mode)
    if torch.is_autocast_enabled("cuda"):
        with torch.autocast("cuda", enabled=False):
            return msda_forward(img.float(), img_shapes, sampling_points.float(), attention_weights.float(),
                                padding_mode == "zeros", bool(align_corners), int(level_cells))
    return msda_forward(img, img_shapes, sampling_points, attention_weights, padding_mode == "zeros", bool(align_corners),
                        int(level_cells))


# ---------------------------------------------------------------------------------------------
# the module core with its prologue fused in (msda_fwd_fused_ / msda_bwd_fused_<dtype>) as custom ops, so that a
# compiled MultiscaleDeformableAttention module keeps the fused kernels
# ---------------------------------------------------------------------------------------------
@torch.library.custom_op("msda_amd::fused_forward", mutates_args=(), device_types="cuda")
def msda_fused_forward(img: torch.Tensor, img_shapes: torch.Tensor, proj: torch.Tensor, reference_points: torch.Tensor,
                       zeros: bool, align_corners: bool, level_cells: int = 0) -> torch.Tensor:
    out = F.msda_hip_fwd_fused(img, img_shapes, proj, reference_points, _PAD[zeros], align_corners)
    if out is None:  # (callers check fused_lp_ok first; kept for safety)
        # as the eager fallback (_HipFusedModuleCoreFunction): with 16-bit storage next to fp32 reference points the
        # prologue runs in fp32 over the mixed-storage operator and the result returns to the projection's dtype —
        # what the fake kernel promises
        pts, att = F.module_sampling_inputs(proj.to(reference_points.dtype), img_shapes, reference_points)
        out = F.msda_hip_fwd(img, img_shapes, pts, att, _PAD[zeros], align_corners).to(proj.dtype)
    return out


@msda_fused_forward.register_fake
def _(img, img_shapes, proj, reference_points, zeros, align_corners, level_cells=0):
    B, _, H, D = img.shape
    return proj.new_empty((B, proj.shape[1], H, D))


@torch.library.custom_op("msda_amd::fused_backward", mutates_args=(), device_types="cuda")
def msda_fused_backward(out_grad: torch.Tensor, img: torch.Tensor, img_shapes: torch.Tensor, proj: torch.Tensor,
                        reference_points: torch.Tensor, zeros: bool, align_corners: bool,
                        need_value: bool, level_cells: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    res = F.msda_hip_bwd_fused(out_grad, img, img_shapes, proj, reference_points, _PAD[zeros], align_corners, need_value,
                               level_cells=level_cells)
    if res is None:
        raise RuntimeError("msda_amd::fused_backward: L*P too large for the fused kernels (check fused_lp_ok first)")
    g_img, g_proj, g_ref = res
    return (g_img if g_img is not None else img.new_empty(0)), g_proj, g_ref


@msda_fused_backward.register_fake
def _(out_grad, img, img_shapes, proj, reference_points, zeros, align_corners, need_value, level_cells=0):
    return (torch.empty_like(img, memory_format=torch.contiguous_format) if need_value else img.new_empty(0),
            torch.empty_like(proj, memory_format=torch.contiguous_format),
            torch.empty_like(reference_points, memory_format=torch.contiguous_format))


def _fused_setup_context(ctx, inputs, output):
    img, img_shapes, proj, reference_points, zeros, align_corners, level_cells = inputs
    ctx.save_for_backward(img, img_shapes, proj, reference_points)
    ctx.zeros, ctx.align_corners, ctx.level_cells = zeros, align_corners, level_cells


def _fused_backward(ctx, out_grad):
    img, img_shapes, proj, reference_points = ctx.saved_tensors
    g_img, g_proj, g_ref = msda_fused_backward(out_grad.contiguous(), img, img_shapes, proj, reference_points, ctx.zeros,
                                               ctx.align_corners, ctx.needs_input_grad[0], ctx.level_cells)
    return (g_img if ctx.needs_input_grad[0] else None, None, g_proj if ctx.needs_input_grad[2] else None,
            g_ref if ctx.needs_input_grad[3] else None, None, None, None)


msda_fused_forward.register_autograd(_fused_backward, setup_context=_fused_setup_context)


@torch._dynamo.assume_constant_result
def _fused_lp_limit(head_dim: int, elem_size: int) -> int:
    from . import _lib
    return int(_lib.load().msda_fused_lp_limit(int(head_dim), int(elem_size)))


def fused_lp_ok(img: torch.Tensor, proj: torch.Tensor, reference_points=None) -> bool:
    """Do the fused kernels take this L*P for this head dimension / dtype?  (Static shapes: decided at trace time.)"""
    storage = reference_points is not None and F.fused_storage_dtypes(img.dtype, proj.dtype, reference_points.dtype)
    elem = 4 if storage else proj.element_size()  # (16-bit storage: the kernels' records are those of fp32 arithmetic)
    return int(proj.shape[3]) * int(proj.shape[4]) <= _fused_lp_limit(int(img.shape[-1]), elem)


def compiled_fused_module_core(img, img_shapes, proj, reference_points, padding_mode, align_corners,
                               level_cells: int = 0) -> torch.Tensor:
    """``fused_module_core`` through the registered custom ops (traceable); fp32 under autocast like the eager path."""
    F._padding_code(padding_mode)
    if F.fused_storage_dtypes(img.dtype, proj.dtype, reference_points.dtype):
        # 16-bit value pyramid and projection next to fp32 reference points: the storage kernels as they are (fp32 arithmetic)
        with torch.autocast("cuda", enabled=False):
            return msda_fused_forward(img, img_shapes, proj, reference_points, padding_mode == "zeros",
                                      bool(align_corners), int(level_cells))
    if torch.is_autocast_enabled("cuda"):
        with torch.autocast("cuda", enabled=False):
            return msda_fused_forward(img.float(), img_shapes, proj.float(), reference_points.float(),
                                      padding_mode == "zeros", bool(align_corners), int(level_cells))
    return msda_fused_forward(img, img_shapes, proj, reference_points, padding_mode == "zeros", bool(align_corners),
                              int(level_cells))
