"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes front-end for the CPU oracle (``oracle/msda_oracle.c``), a scalar C
restatement of the reference algorithm (reference file:line citations live in
``msda_oracle_impl.h``).  Importers are limited to ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg; the
product package (``msda_triton_amd``) never imports this module.

Parity status: pinned against ``tests/golden/*.npz`` (outputs of the reference's
own ``native_multiscale_deformable_attention`` generated in the build container
by ``tests/golden/make_golden.py``) — see ``tests/test_oracle.py``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmsda_oracle.so")
_lib = None

PADDING = {"border": 0, "zeros": 1}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (idempotent)."""
    srcs = [os.path.join(_HERE, f) for f in ("msda_oracle.c", "msda_oracle_impl.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        i64 = ctypes.c_int64
        vp = ctypes.c_void_p
        for suf in ("f32", "f64"):
            f = getattr(_lib, f"msda_oracle_fwd_{suf}")
            f.restype = ctypes.c_int
            f.argtypes = [vp, vp, vp, vp, vp] + [i64] * 7 + [ctypes.c_int, ctypes.c_int]
            g = getattr(_lib, f"msda_oracle_bwd_{suf}")
            g.restype = ctypes.c_int
            g.argtypes = [vp] * 8 + [i64] * 7 + [ctypes.c_int, ctypes.c_int]
        _lib.msda_oracle_num_threads.restype = ctypes.c_int
        _lib.msda_oracle_set_num_threads.argtypes = [ctypes.c_int]
    return _lib


def num_threads() -> int:
    return int(_load().msda_oracle_num_threads())


def set_num_threads(n: int) -> None:
    _load().msda_oracle_set_num_threads(int(n))


def _prep(a, dtype):
    a = np.ascontiguousarray(np.asarray(a), dtype=dtype)
    return a


def _dims(value, loc):
    B, I, H, D = value.shape
    B2, Q, H2, L, P, two = loc.shape
    assert (B, H, two) == (B2, H2, 2), "inconsistent shapes"
    return B, I, H, D, Q, L, P


def _suffix(dtype):
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"oracle computes in float32/float64 only, got {dtype}")


def forward(value, shapes, loc, attn, padding_mode: str, align_corners: bool):
    """numpy in, numpy out.  dtype follows ``value`` (float32 or float64)."""
    value = np.asarray(value)
    dt = value.dtype.type
    suf = _suffix(dt)
    value, loc, attn = _prep(value, dt), _prep(loc, dt), _prep(attn, dt)
    shapes = _prep(shapes, np.int64)
    B, I, H, D, Q, L, P = _dims(value, loc)
    assert shapes.shape == (L, 2) and attn.shape == (B, Q, H, L, P)
    out = np.empty((B, Q, H, D), dtype=dt)
    rc = getattr(_load(), f"msda_oracle_fwd_{suf}")(
        value.ctypes.data, shapes.ctypes.data, loc.ctypes.data, attn.ctypes.data, out.ctypes.data,
        B, I, H, D, Q, L, P, PADDING[padding_mode], int(bool(align_corners)))
    if rc:
        raise ValueError(f"msda_oracle_fwd_{suf} failed with code {rc}")
    return out


def backward(grad_out, value, shapes, loc, attn, padding_mode: str, align_corners: bool):
    """Returns (grad_value, grad_loc, grad_attn) as numpy arrays."""
    value = np.asarray(value)
    dt = value.dtype.type
    suf = _suffix(dt)
    grad_out, value, loc, attn = (_prep(t, dt) for t in (grad_out, value, loc, attn))
    shapes = _prep(shapes, np.int64)
    B, I, H, D, Q, L, P = _dims(value, loc)
    assert grad_out.shape == (B, Q, H, D)
    g_value = np.empty_like(value)
    g_loc = np.empty_like(loc)
    g_attn = np.empty_like(attn)
    rc = getattr(_load(), f"msda_oracle_bwd_{suf}")(
        grad_out.ctypes.data, value.ctypes.data, shapes.ctypes.data, loc.ctypes.data, attn.ctypes.data,
        g_value.ctypes.data, g_loc.ctypes.data, g_attn.ctypes.data,
        B, I, H, D, Q, L, P, PADDING[padding_mode], int(bool(align_corners)))
    if rc:
        raise ValueError(f"msda_oracle_bwd_{suf} failed with code {rc}")
    return g_value, g_loc, g_attn
