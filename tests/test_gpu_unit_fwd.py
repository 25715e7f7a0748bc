"""The forward for small problems — one wave per (b, q, h) unit (``msda_fwd_unit_kernel``; ``msda_set_option("unit_fwd", …)``:
0 never, 1 up to 12 288 units (default), 2 wherever it exists) — against the oracle and the general kernel."""
import zlib

import numpy as np
import pytest
import torch

from conftest import MODES, mode_key
from test_gpu_parity import DEV, FWD_TOL, SHAPE_MATRIX, rand_case

pytestmark = pytest.mark.gpu


def _forward(c, td, pm, ac, opt):
    from msda_triton_amd import _lib, multiscale_deformable_attention
    old = _lib.get_option("unit_fwd")
    _lib.set_option("unit_fwd", opt)
    try:
        v = torch.from_numpy(c["value"]).to(DEV, td)
        l = torch.from_numpy(c["loc"]).to(DEV, td)
        a = torch.from_numpy(c["attn"]).to(DEV, td)
        s = torch.from_numpy(c["shapes"]).to(DEV)
        with torch.no_grad():
            out = multiscale_deformable_attention(v, s, l, a, pm, ac)
        torch.cuda.synchronize()
        return out
    finally:
        _lib.set_option("unit_fwd", old)


@pytest.mark.parametrize("name", list(SHAPE_MATRIX), ids=list(SHAPE_MATRIX))
@pytest.mark.parametrize("pm,ac", MODES, ids=[mode_key(*m) for m in MODES])
def test_unit_forward_vs_oracle_f32(oracle, name, pm, ac):
    """every shape of the parity matrix (the kernel serves the 16-byte vector shapes; the others must fall through to the
    general kernel unharmed)"""
    B, Q, H, D, levels, P = SHAPE_MATRIX[name]
    c = rand_case(np.random.default_rng(zlib.crc32(name.encode()) + 7), B, Q, H, D, levels, P)
    unit = _forward(c, torch.float32, pm, ac, 2)
    ref = oracle.forward(c["value"], c["shapes"], c["loc"], c["attn"], pm, ac)
    np.testing.assert_allclose(unit.cpu().numpy(), ref, **FWD_TOL[torch.float32])
    general = _forward(c, torch.float32, pm, ac, 0)
    np.testing.assert_allclose(unit.cpu().numpy(), general.cpu().numpy(), atol=2e-5, rtol=1e-5)
    again = _forward(c, torch.float32, pm, ac, 2)
    assert torch.equal(unit, again)  # (a fixed summation order: bitwise reproducible)


@pytest.mark.parametrize("td,atol,rtol", [(torch.float16, 2e-2, 2e-2), (torch.bfloat16, 4e-2, 2e-2)], ids=["fp16", "bf16"])
@pytest.mark.parametrize("name", ["d32_vec_g8", "d64_vec_g16", "d8_vec_g4"])
def test_unit_forward_low_precision(oracle, td, atol, rtol, name):
    B, Q, H, D, levels, P = SHAPE_MATRIX[name]
    c = rand_case(np.random.default_rng(zlib.crc32(name.encode()) + 9), B, Q, H, D, levels, P)
    r = {k: (v if k == "shapes" else torch.from_numpy(v).to(td).float().numpy()) for k, v in c.items()}
    for pm, ac in (("zeros", False), ("border", True)):
        out = _forward(c, td, pm, ac, 2).float().cpu().numpy()
        ref = oracle.forward(r["value"], r["shapes"], r["loc"], r["attn"], pm, ac)
        np.testing.assert_allclose(out, ref, atol=atol, rtol=rtol)


def test_unit_forward_is_the_default_for_decoder_calls_and_not_for_large_ones():
    """c4 (B = 8, Q = 900: 57 600 units) stays with the general kernel, a 300-query call (B = 2: 4 800 units) takes the unit
    kernel — read back from the library's own kernel profile"""
    from msda_triton_amd import _lib, synth
    from msda_triton_amd.functional import msda_hip_fwd
    old = _lib.get_option("unit_fwd")
    _lib.set_option("unit_fwd", 1)  # (the default; the suite may run under MSDA_TEST_OPTS=unit_fwd=…)
    seen = {}
    import dataclasses
    for wl_name in ("c1_readme", "c4_gdino_dec"):
        wl = synth.WORKLOADS[wl_name]
        if wl_name == "c1_readme":
            wl = dataclasses.replace(wl, Q=300)
        d = synth.make_inputs_torch(wl, DEV, seed=1)
        _lib.set_option("profile", 1)
        try:
            _lib.profile_read()
            msda_hip_fwd(d["value"], d["shapes"], d["loc"], d["attn"], wl.padding_mode, wl.align_corners)
            torch.cuda.synchronize()
            seen[wl_name] = set(_lib.profile_read())
        finally:
            _lib.set_option("profile", 0)
    _lib.set_option("unit_fwd", old)
    assert seen["c1_readme"] == {"msda_fwd_unit_kernel"}, seen
    assert seen["c4_gdino_dec"] == {"msda_fwd_kernel"}, seen


@pytest.mark.parametrize("name", list(SHAPE_MATRIX), ids=list(SHAPE_MATRIX))
def test_two_units_per_wave_agree_with_one(name):
    """``unit_waves`` 2: each half of a wave serves a unit (rows of a unit over 32 lanes instead of 64: other partial sums, so
    not bit-identical in general — but within rounding of the one-unit kernel, reproducible, and correct for an odd number
    of units, where the last wave's second half has nothing to do)"""
    from msda_triton_amd import _lib
    B, Q, H, D, levels, P = SHAPE_MATRIX[name]
    c = rand_case(np.random.default_rng(zlib.crc32(name.encode()) + 11), B, Q, H, D, levels, P)
    old = _lib.get_option("unit_waves")
    try:
        outs = {}
        for uw in (1, 2, 2):
            _lib.set_option("unit_waves", uw)
            outs.setdefault(uw, []).append(_forward(c, torch.float32, "zeros", False, 2))
    finally:
        _lib.set_option("unit_waves", old)
    assert torch.equal(outs[2][0], outs[2][1])
    np.testing.assert_allclose(outs[2][0].cpu().numpy(), outs[1][0].cpu().numpy(), atol=2e-5, rtol=1e-5)
