"""ctypes binding of ``libmsda_hip.so`` — the C ABI declared in ``include/msda_hip.h``.

The library is built in-tree by ``msda_triton_amd/csrc/Makefile`` (``__graft_entry__.build()``)
and lives next to this file.  There is no fallback: if it is missing, every GPU call raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libmsda_hip.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)
CSRC_DIR = os.path.join(_HERE, "csrc")

ABI_VERSION = 11
PADDING_MODES = {"border": 0, "zeros": 1}
WS_RECORDS_IN_GRADS = 1  # msda_bwd_workspace_bytes flag (include/msda_hip.h)


def ws_passes(n: int) -> int:
    """msda_bwd_workspace_bytes flag MSDA_WS_PASSES(n): the size for n passes over the batch (include/msda_hip.h)."""
    return (int(n) & 0xFF) << 8


# one storage type for every tensor, then the mixed ones: value / grad_value in 16 bits, everything else fp32
DTYPE_SUFFIXES = ("f32", "f16", "bf16", "f64", "f32_vbf16", "f32_vf16")
# module storage (fused entry points only): value, projection, out and their gradients in 16 bits, reference points fp32
FUSED_STORAGE_SUFFIXES = ("f32_sbf16", "f32_sf16")

# every symbol include/msda_hip.h declares
EXPORTED_SYMBOLS = tuple(
    [f"msda_{d}_{s}" for d in ("fwd", "bwd", "fwd_fused", "bwd_fused") for s in DTYPE_SUFFIXES]
    + [f"msda_{d}_{s}" for d in ("fwd_fused", "bwd_fused") for s in FUSED_STORAGE_SUFFIXES]
    + ["msda_abi_version", "msda_last_error", "msda_set_option", "msda_get_option", "msda_bwd_workspace_bytes",
       "msda_bwd_fused_workspace_bytes", "msda_bwd_supported", "msda_fused_lp_limit", "msda_profile_read",
       "msda_last_launch_info"]
)

_lib = None
_lock = threading.Lock()


class MSDALibraryError(RuntimeError):
    """libmsda_hip.so is missing/unloadable, or a call into it failed."""


def build(verbose: bool = False, jobs: int = 5) -> str:
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, f"-j{jobs}"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise MSDALibraryError(f"building {LIB_NAME} failed (exit {res.returncode})")
    return LIB_PATH


def load():
    """Load (once) and return the ctypes handle; raises MSDALibraryError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise MSDALibraryError(
                f"{LIB_PATH} not found: the HIP extension has not been built. Run "
                f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C {CSRC_DIR}`. "
                "There is no CPU fallback for GPU tensors."
            )
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 missing
            raise MSDALibraryError(f"cannot load {LIB_PATH}: {e}") from e
        i64, vp, ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
        for suf in DTYPE_SUFFIXES:
            f = getattr(lib, f"msda_fwd_{suf}")
            f.restype = ci
            # (..., padding_mode, align_corners, value_row_stride, stream)
            f.argtypes = [vp] * 5 + [i64] * 7 + [ci, ci, i64, vp]
            ff = getattr(lib, f"msda_fwd_fused_{suf}")
            ff.restype = ci
            ff.argtypes = [vp] * 5 + [i64] * 7 + [ci, ci, ci, i64, vp]
            # the backward: (..., max_level_cells, value_row_stride, workspace, workspace_bytes, stream)
            g = getattr(lib, f"msda_bwd_{suf}")
            g.restype = ci
            g.argtypes = [vp] * 8 + [i64] * 7 + [ci, ci, i64, i64, vp, i64, vp]
            gf = getattr(lib, f"msda_bwd_fused_{suf}")
            gf.restype = ci
            gf.argtypes = [vp] * 8 + [i64] * 7 + [ci, ci, ci, i64, i64, vp, i64, vp]
        for suf in FUSED_STORAGE_SUFFIXES:
            ff = getattr(lib, f"msda_fwd_fused_{suf}")
            ff.restype = ci
            ff.argtypes = [vp] * 5 + [i64] * 7 + [ci, ci, ci, i64, vp]
            gf = getattr(lib, f"msda_bwd_fused_{suf}")
            gf.restype = ci
            gf.argtypes = [vp] * 8 + [i64] * 7 + [ci, ci, ci, i64, i64, vp, i64, vp]
        lib.msda_bwd_workspace_bytes.restype = i64
        lib.msda_bwd_workspace_bytes.argtypes = [i64] * 7 + [ci, ci, i64, ci]
        lib.msda_bwd_fused_workspace_bytes.restype = i64
        lib.msda_bwd_fused_workspace_bytes.argtypes = [i64] * 7 + [ci, ci, i64, ci]
        lib.msda_bwd_supported.restype = ci
        lib.msda_bwd_supported.argtypes = [i64] * 7 + [ci]
        lib.msda_profile_read.restype = ci
        lib.msda_profile_read.argtypes = [ctypes.c_char_p, ci]
        lib.msda_fused_lp_limit.restype = i64
        lib.msda_fused_lp_limit.argtypes = [i64, ci]
        lib.msda_abi_version.restype = ci
        lib.msda_last_error.restype = ctypes.c_char_p
        lib.msda_set_option.restype = ci
        lib.msda_set_option.argtypes = [ctypes.c_char_p, ci]
        lib.msda_get_option.restype = ci
        lib.msda_get_option.argtypes = [ctypes.c_char_p]
        lib.msda_last_launch_info.restype = ci
        lib.msda_last_launch_info.argtypes = [ctypes.c_char_p]
        got = lib.msda_abi_version()
        if got != ABI_VERSION:
            raise MSDALibraryError(f"{LIB_NAME} has ABI version {got}, this package expects {ABI_VERSION}; rebuild it")
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().msda_last_error().decode(errors="replace")
        if rc < 0:
            raise ValueError(f"{what}: rejected arguments ({rc}): {msg}")
        raise MSDALibraryError(f"{what}: HIP error {rc}: {msg}")


OPTION_EPOCH = 0  # bumped by set_option: callers that cache option-dependent values (workspace sizes) key on it


def set_option(key: str, value: int) -> None:
    global OPTION_EPOCH
    check(load().msda_set_option(key.encode(), int(value)), f"msda_set_option({key})")
    OPTION_EPOCH += 1


def get_option(key: str) -> int:
    return int(load().msda_get_option(key.encode()))


LAUNCH_INFO_KEYS = ("fwd_variant", "fwd_lds_level_bytes", "fwd_lds_planes", "fwd_workgroups", "sample_variant",
                    "sample_lds_level_bytes", "value_path", "value_passes")


def last_launch_info() -> dict:
    """Which variants the most recent launches took (measurement only; include/msda_hip.h msda_last_launch_info)."""
    lib = load()
    return {k: int(lib.msda_last_launch_info(k.encode())) for k in LAUNCH_INFO_KEYS}


def profile_read() -> dict:
    """{kernel name: (launches, mean microseconds)} of the kernels launched (by any thread) since the last read while
    option "profile" was 1 (measurement only; synchronises on the recorded events)."""
    lib = load()
    buf = ctypes.create_string_buffer(1 << 16)  # (one call: the read consumes the records)
    lib.msda_profile_read(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, launches, total = line.rsplit(" ", 2)
        out[name] = (int(launches), float(total) / max(int(launches), 1))
    return out
