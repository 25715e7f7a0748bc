"""Drop-in shim: the import path of rziga/msda-triton, served by the MI355X HIP kernels.

``msda_triton.multiscale_deformable_attention`` and ``msda_triton.MultiscaleDeformableAttention``
(reference: src/msda_triton/__init__.py:2,7-10) resolve to ``msda_triton_amd``.
"""
from msda_triton_amd import MultiscaleDeformableAttention, multiscale_deformable_attention
from msda_triton_amd import __version__

__all__ = [
    "multiscale_deformable_attention",
    "MultiscaleDeformableAttention",
]
