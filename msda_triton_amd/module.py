"""``MultiscaleDeformableAttention`` — the reference's nn.Module interface
(/root/reference/src/msda_triton/frontend.py:175-292) over the HIP operator.

Parameter names (``img_input_proj``, ``query_input_proj``, ``query_output_proj``) and shapes are
the reference's, so state dicts interchange.  The projections are ``nn.Linear`` layers (library
GEMMs; ``_linear.py`` only re-shapes their weight gradient for tall inputs); the deformable-attention core is custom.
"""
from __future__ import annotations

from typing import Literal, Optional

import torch
from torch import nn

from ._linear import projection
from .functional import fused_module_core, module_sampling_inputs


class MultiscaleDeformableAttention(nn.Module):
    """Multiscale deformable attention with input/output projections (Deformable-DETR, fig. 2).

    Args:
        emb_dim: feature dimension of ``img`` and ``queries``.
        hidden_dim: projected feature dimension; must be divisible by ``num_heads``.
        num_levels: number of pyramid levels.
        num_heads: number of attention heads.
        num_points: sampling points per level.
        padding_mode: ``"border"`` or ``"zeros"``.
        align_corners: grid alignment.
        value_dtype: (not in the reference; GPU only) ``torch.bfloat16`` / ``torch.float16``: keep the projected value
            pyramid — the largest tensor the attention core reads, and the one autograd saves — in 16 bits, in the
            kernel's ``[B, I, H, D]`` layout, while sampling offsets, attention logits, reference points and the
            result stay fp32 (the kernels' mixed-storage entry points; arithmetic is fp32 either way).  Under
            ``torch.autocast`` to that dtype the projection GEMM writes it directly; otherwise its output is cast
            once.  ``None`` (default): the reference's behaviour — everything in the parameters' dtype, fp32 under
            autocast.

    Raises:
        ValueError: if ``hidden_dim`` is not divisible by ``num_heads`` (frontend.py:211-212).
    """

    def __init__(self, emb_dim: int, hidden_dim: int, num_levels: int, num_heads: int, num_points: int,
                 padding_mode: Literal["border", "zeros"], align_corners: bool,
                 value_dtype: Optional[torch.dtype] = None):
        super().__init__()
        if hidden_dim % num_heads != 0:
            raise ValueError(
                f"Hidden dimension ({hidden_dim=}) should be divisible by number of heads ({num_heads=}).")
        if value_dtype not in (None, torch.bfloat16, torch.float16):
            raise ValueError(f"`value_dtype` should be None, torch.bfloat16 or torch.float16, but got {value_dtype}.")
        self.value_dtype = value_dtype
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.hidden_dim = hidden_dim
        self.padding_mode = padding_mode
        self.align_corners = align_corners
        # one fused query projection: (x offset, y offset, attention logit) per (head, level, point)
        self.img_input_proj = nn.Linear(emb_dim, hidden_dim)
        self.query_input_proj = nn.Linear(emb_dim, num_heads * num_levels * num_points * 3)
        self.query_output_proj = nn.Linear(hidden_dim, emb_dim)

    def sampling_inputs(self, img_shapes: torch.Tensor, queries: torch.Tensor, reference_points: torch.Tensor):
        """Query projection -> (sampling_points [B,N,H,L,P,2], attention_weights [B,N,H,L,P])."""
        B, N, _ = queries.shape
        proj = self.query_input_proj(queries).reshape(B, N, self.num_heads, self.num_levels, self.num_points, 3)
        return module_sampling_inputs(proj, img_shapes, reference_points)

    def forward(self, img: torch.Tensor, img_shapes: torch.Tensor, queries: torch.Tensor,
                reference_points: torch.Tensor, level_shapes=None) -> torch.Tensor:
        """
        Args:
            img: flattened pyramid ``[batch, num_image, emb_dim]``.
            img_shapes: ``[num_levels, 2]`` (height, width).
            queries: ``[batch, num_queries, emb_dim]``.
            reference_points: ``[batch, num_queries, 2]`` (x, y) or ``[batch, num_queries, 4]``
                (cx, cy, w, h), normalised to [0, 1].
            level_shapes: optional (not in the reference): the same (height, width) pairs as host numbers; see
                ``msda_triton_amd.functional.level_cells_of``.

        Returns:
            ``[batch, num_queries, emb_dim]``.
        """
        B, I, _ = img.shape  # noqa: E741
        N = queries.shape[1]
        H, L, P = self.num_heads, self.num_levels, self.num_points
        if reference_points.shape[-1] not in (2, 4):
            raise ValueError(
                f"`reference_points` should have the last dim either 2 or 4, but got {reference_points.shape[-1]}.")
        # one projection holds (x offset, y offset, attention logit) per (head, level, point); on the GPU the softmax
        # and the offset -> sampling-point math run inside the attention kernel's prologue
        proj = projection(self.query_input_proj, queries).reshape(B, N, H, L, P, 3)
        # (the value pyramid is this module's own tensor: for large fp32 calls its pixels' rows are written one 128-byte line
        #  apart, which takes them off the vector L1's tag-RAM skew — functional.value_row_pad; the kernels read that layout
        #  in place.  Measured at the c2 module shape, fp32, tools/module_pad_ab.py: step 1.05 -> 1.02 ms at 10 000 queries,
        #  no gain at 2 500, and a LOSS at 900 — a host-bound step that the two extra small launches lengthen — so only
        #  from 32 768 (b, q) rows on)
        value = projection(self.img_input_proj, img, pad_rows=B * N >= 32768).reshape(B, I, H, self.hidden_dim // H)
        if value.device.type == "cuda" and proj.dtype in (torch.bfloat16, torch.float16) and \
                self.value_dtype in (None, proj.dtype) and value.dtype == proj.dtype:
            # 16-bit projections (autocast's GEMMs, or 16-bit parameters) with fp32 reference points: the kernels read
            # and write the 16-bit tensors as they are and compute in fp32 — what the reference's core does under
            # autocast (it casts every input to fp32, frontend.py:111) without the fp32 copies of value, projection,
            # result and their gradients; rounding happens at the same places (the GEMMs' outputs / inputs)
            with torch.autocast("cuda", enabled=False):
                attended = fused_module_core(value, img_shapes, proj, reference_points.float(), self.padding_mode,
                                             self.align_corners, level_shapes)
        elif self.value_dtype is not None and value.device.type == "cuda":
            # 16-bit value pyramid next to fp32 sampling inputs: the mixed-storage kernels read it as it is (no fp32
            # copy, which is what autocast's cast_inputs would make), so the call sits outside autocast
            value = value.to(self.value_dtype)  # nothing to do when autocast already produced this dtype
            with torch.autocast("cuda", enabled=False):
                attended = fused_module_core(value, img_shapes, proj.float(), reference_points.float(),
                                             self.padding_mode, self.align_corners, level_shapes).to(proj.dtype)
        else:
            attended = fused_module_core(value, img_shapes, proj, reference_points, self.padding_mode,
                                         self.align_corners, level_shapes)
        return projection(self.query_output_proj, attended.reshape(B, N, self.hidden_dim))
