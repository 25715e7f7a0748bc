"""Optional C++ autograd glue (``csrc/msda_torch_ext.cpp``) over the same C ABI as the ctypes route.

It exists for one reason: host time.  For Grounding-DINO-sized calls the kernels take ~60 us while the Python
autograd Function + ctypes marshalling cost ~100 us per forward+backward; the C++ Function costs a fraction of
that.  No kernels, no numerics live here; when the extension is not built (or its ABI version differs) the
package silently uses the ctypes route — ``libmsda_hip.so`` itself stays mandatory (``_lib.load``).
"""
from __future__ import annotations

import importlib
import os
import subprocess
import sysconfig

from . import _lib

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
SOURCE = os.path.join(PKG_DIR, "csrc", "msda_torch_ext.cpp")
MODULE = "msda_torch_ext"
EXT_PATH = os.path.join(PKG_DIR, MODULE + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))

_mod = None
_tried = False


def build(verbose: bool = False) -> str:
    """Compile the binding with g++ against the installed PyTorch (host code only; ~40 s)."""
    import torch
    from torch.utils import cpp_extension

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    newest_src = max(os.path.getmtime(p) for p in (SOURCE, os.path.join(os.path.dirname(PKG_DIR), "include", "msda_hip.h")))
    if os.path.exists(EXT_PATH) and os.path.getmtime(EXT_PATH) >= newest_src:
        return EXT_PATH
    libs = cpp_extension.library_paths()
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = (["g++", "-O2", "-fPIC", "-shared", "-std=c++17", SOURCE, f"-DTORCH_EXTENSION_NAME={MODULE}",
            "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
            f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
            "-I" + sysconfig.get_paths()["include"], f"-I{rocm}/include"]
           + ["-I" + i for i in cpp_extension.include_paths()]
           + ["-L" + d for d in libs] + ["-L" + PKG_DIR, "-lmsda_hip", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch",
                                         "-ltorch_hip", "-ltorch_python", f"-L{rocm}/lib", "-lamdhip64",
                                         "-Wl,-rpath,$ORIGIN"]
           + ["-Wl,-rpath," + d for d in libs] + ["-o", EXT_PATH])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise _lib.MSDALibraryError(f"building {MODULE} failed (exit {res.returncode})")
    return EXT_PATH


def load():
    """The extension module, or None when it is absent, disabled (MSDA_NO_TORCH_EXT=1) or of another ABI version."""
    global _mod, _tried
    if _tried:
        return _mod
    _tried = True
    if os.environ.get("MSDA_NO_TORCH_EXT") == "1" or not os.path.exists(EXT_PATH):
        return None
    try:
        import torch  # noqa: F401  (libtorch must be loaded before the extension)
        _lib.load()
        mod = importlib.import_module(f"{__package__}.{MODULE}")
        if int(mod.abi_version()) == _lib.ABI_VERSION:
            _mod = mod
    except Exception:  # a stale or unloadable binding is not an error: the ctypes route serves every call
        _mod = None
    return _mod
