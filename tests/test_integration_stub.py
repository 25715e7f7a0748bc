"""The reference-side binding INTEGRATION.md shows a maintainer (Option B: a ctypes stub replacing
/root/reference/src/msda_triton/kernels.py:351-379, 556-592) is EXECUTED here, straight from the document: a signature
that drifts from include/msda_hip.h (as ABI 10 -> 11 changed argument lists under existing names) fails this test instead of
corrupting a reader's memory.  CPU: the stub loads the library, passes its ABI check and binds every function it names.
GPU: its launcher pair against the oracle."""
import os
import re
import types

import numpy as np
import pytest

from conftest import ROOT


def _load_stub():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    code = next(b for b in blocks if "reference-side stub" in b)
    from msda_triton_amd import _lib
    _lib.build() if not os.path.exists(_lib.LIB_PATH) else None
    assert 'ctypes.CDLL("libmsda_hip.so")' in code
    code = code.replace('ctypes.CDLL("libmsda_hip.so")', f"ctypes.CDLL({_lib.LIB_PATH!r})")
    mod = types.ModuleType("integration_stub")
    exec(compile(code, "INTEGRATION.md:option-B", "exec"), mod.__dict__)
    return mod


def test_stub_loads_checks_the_abi_and_binds_its_functions():
    from msda_triton_amd import _lib
    stub = _load_stub()
    assert stub._lib.msda_abi_version() == _lib.ABI_VERSION  # the document's literal is the library's version
    assert callable(stub.triton_multi_scale_deformable_attention_fwd)
    assert callable(stub.triton_multi_scale_deformable_attention_bwd)
    # the stub's argument lists are the header's: same arity as the package's own binding
    lib = _lib.load()
    for suf in ("f32", "f16", "bf16", "f64"):
        for name in (f"msda_fwd_{suf}", f"msda_bwd_{suf}"):
            assert len(getattr(stub._lib, name).argtypes) == len(getattr(lib, name).argtypes), name
    assert len(stub._lib.msda_bwd_workspace_bytes.argtypes) == len(lib.msda_bwd_workspace_bytes.argtypes)


@pytest.mark.gpu
@pytest.mark.parametrize("pm,ac", [("zeros", False), ("border", True)])
@pytest.mark.parametrize("Q", [60, 1500])  # the single-launch grad_value kernel (no workspace) / the sorted pipeline
def test_stub_launcher_pair_matches_the_oracle(oracle, pm, ac, Q):
    import torch
    from test_gpu_parity import BWD_TOL, FWD_TOL, rand_case
    from conftest import kink_mask
    stub = _load_stub()
    c = rand_case(np.random.default_rng(5 + Q), 2, Q, 4, 32, [(9, 7), (5, 4), (2, 3)], 3)
    dev = torch.device("cuda", 0)
    v, l, a, g = (torch.from_numpy(c[k]).to(dev) for k in ("value", "loc", "attn", "grad_out"))
    s = torch.from_numpy(c["shapes"]).to(dev)
    out = stub.triton_multi_scale_deformable_attention_fwd(v, s, l, a, pm, ac)
    gv, gl, ga = stub.triton_multi_scale_deformable_attention_bwd(g, v, s, l, a, pm, ac)
    torch.cuda.synchronize()
    host = (c["value"], c["shapes"], c["loc"], c["attn"])
    np.testing.assert_allclose(out.cpu().numpy(), oracle.forward(*host, pm, ac), **FWD_TOL[torch.float32])
    r_gv, r_gl, r_ga = oracle.backward(c["grad_out"], *host, pm, ac)
    np.testing.assert_allclose(gv.cpu().numpy(), r_gv, **BWD_TOL[torch.float32])
    np.testing.assert_allclose(ga.cpu().numpy(), r_ga, **BWD_TOL[torch.float32])
    keep = ~kink_mask(c["loc"], c["shapes"], ac)
    np.testing.assert_allclose(np.where(keep, gl.cpu().numpy(), 0), np.where(keep, r_gl, 0), **BWD_TOL[torch.float32])
