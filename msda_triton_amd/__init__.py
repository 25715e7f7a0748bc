"""msda_triton_amd — MI355X (gfx950) native multi-scale deformable attention.

Hand-written HIP kernels behind the operator interface of rziga/msda-triton:

    from msda_triton_amd import multiscale_deformable_attention, MultiscaleDeformableAttention

or, as a drop-in under the reference's own import path, ``import msda_triton`` (the shim
package at the repository root re-exports the same names).
"""
from .functional import (  # noqa: F401
    hip_multiscale_deformable_attention,
    msda_hip_bwd,
    msda_hip_fwd,
    multiscale_deformable_attention,
    native_multiscale_deformable_attention,
)
from .module import MultiscaleDeformableAttention  # noqa: F401

__version__ = "0.1.0"

__all__ = [
    "multiscale_deformable_attention",
    "MultiscaleDeformableAttention",
    "hip_multiscale_deformable_attention",
    "native_multiscale_deformable_attention",
    "msda_hip_fwd",
    "msda_hip_bwd",
]
