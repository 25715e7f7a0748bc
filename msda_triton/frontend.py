"""Shim for ``msda_triton.frontend`` — the names the reference's own tests and benchmark import
(tests/test_msda.py:8-12, scripts/benchmark.py:4-7 of rziga/msda-triton)."""
from msda_triton_amd.functional import (  # noqa: F401
    hip_multiscale_deformable_attention,
    multiscale_deformable_attention,
    native_multiscale_deformable_attention,
)
from msda_triton_amd.module import MultiscaleDeformableAttention  # noqa: F401

# the reference's GPU entry point; here it launches the HIP kernels
triton_multiscale_deformable_attention = hip_multiscale_deformable_attention
