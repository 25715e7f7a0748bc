"""The gather kernels that keep the coarsest pyramid levels in LDS (``msda_set_option("lds_levels", …)``: 0 never,
1 where the launcher's plan says it pays, 2 wherever the variant exists) against the plain kernels and the oracle.
The variant runs the same arithmetic in the same order, so its results must be BIT-identical to the plain kernels'."""
import zlib

import numpy as np
import pytest
import torch

from conftest import MODES, mode_key
from test_gpu_parity import BWD_TOL, DEV, FWD_TOL, rand_case

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_unit_forward():
    """these shapes are small: without this they would all take the one-wave-per-unit forward (test_gpu_unit_fwd.py)"""
    from msda_triton_amd import _lib
    old = _lib.get_option("unit_fwd")
    _lib.set_option("unit_fwd", 0)
    yield
    _lib.set_option("unit_fwd", old)

# name: (B, Q, H, D, levels, P, dtype)
CASES = {
    "c2_like_f32": (2, 700, 4, 32, [(16, 16), (8, 8), (4, 4), (2, 2)], 4, torch.float32),
    "all_levels_fit": (1, 300, 2, 32, [(6, 5), (3, 3)], 4, torch.float32),
    "no_level_fits_but_last": (1, 260, 2, 32, [(40, 40), (33, 31), (2, 2)], 2, torch.float32),
    "nothing_fits": (1, 200, 1, 32, [(64, 64), (40, 40)], 2, torch.float32),
    "coarse_first_order": (1, 150, 2, 16, [(2, 2), (4, 4), (9, 7)], 3, torch.float32),
    "odd_points_boundary": (2, 333, 3, 32, [(9, 9), (5, 4), (3, 2)], 3, torch.float32),
    "many_samples_two_trips": (1, 140, 2, 32, [(8, 8), (4, 4), (2, 2), (1, 1), (3, 3)], 8, torch.float32),
    "bf16_g4": (2, 520, 4, 32, [(20, 17), (10, 9), (5, 4)], 4, torch.bfloat16),
    "fp16_d64": (1, 300, 4, 64, [(16, 16), (8, 8), (4, 4)], 8, torch.float16),
    "f64": (1, 130, 2, 8, [(7, 6), (4, 3)], 3, torch.float64),
    "d64_f32_g16": (1, 200, 2, 64, [(12, 12), (6, 6), (3, 3)], 2, torch.float32),
}


def _forward(c, td, pm, ac, opt):
    from msda_triton_amd import _lib, multiscale_deformable_attention
    old = _lib.get_option("lds_levels")
    _lib.set_option("lds_levels", opt)
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
        _lib.set_option("lds_levels", old)


@pytest.mark.parametrize("name", list(CASES), ids=list(CASES))
@pytest.mark.parametrize("pm,ac", MODES, ids=[mode_key(*m) for m in MODES])
def test_lds_served_levels_forward_is_bit_identical(oracle, name, pm, ac):
    B, Q, H, D, levels, P, td = CASES[name]
    c = rand_case(np.random.default_rng(zlib.crc32(name.encode())), B, Q, H, D, levels, P,
                  dtype=np.float64 if td == torch.float64 else np.float32)
    plain = _forward(c, td, pm, ac, 0)
    lds = _forward(c, td, pm, ac, 2)
    assert torch.equal(plain, lds), f"max diff {(plain.double() - lds.double()).abs().max().item():.3e}"
    if td in FWD_TOL:
        ref = oracle.forward(c["value"], c["shapes"], c["loc"], c["attn"], pm, ac)
        np.testing.assert_allclose(lds.cpu().numpy(), ref, **FWD_TOL[td])


def _backward(c, td, pm, ac, opt):
    from msda_triton_amd import _lib, multiscale_deformable_attention
    old = _lib.get_option("lds_levels")
    _lib.set_option("lds_levels", opt)
    try:
        v = torch.from_numpy(c["value"]).to(DEV, td).requires_grad_(True)
        l = torch.from_numpy(c["loc"]).to(DEV, td).requires_grad_(True)
        a = torch.from_numpy(c["attn"]).to(DEV, td).requires_grad_(True)
        s = torch.from_numpy(c["shapes"]).to(DEV)
        out = multiscale_deformable_attention(v, s, l, a, pm, ac)
        out.backward(torch.from_numpy(c["grad_out"]).to(DEV, td))
        torch.cuda.synchronize()
        return out.detach(), v.grad, l.grad, a.grad
    finally:
        _lib.set_option("lds_levels", old)


@pytest.mark.parametrize("name", list(CASES), ids=list(CASES))
@pytest.mark.parametrize("pm,ac", MODES, ids=[mode_key(*m) for m in MODES])
def test_lds_served_levels_backward_is_bit_identical(oracle, name, pm, ac):
    """grad_loc / grad_attn from the sample-gradient kernel with LDS-served levels (the units of 4 / 8 lanes with float
    accumulation have the variant; the others run the plain kernel under either option)"""
    B, Q, H, D, levels, P, td = CASES[name]
    c = rand_case(np.random.default_rng(zlib.crc32(name.encode()) + 1), B, Q, H, D, levels, P,
                  dtype=np.float64 if td == torch.float64 else np.float32)
    plain = _backward(c, td, pm, ac, 0)
    lds = _backward(c, td, pm, ac, 2)
    for nm, x, y in zip(("out", "grad_value", "grad_loc", "grad_attn"), plain, lds):
        assert torch.equal(x, y), f"{nm}: max diff {(x.double() - y.double()).abs().max().item():.3e}"
    if td in BWD_TOL:
        from conftest import kink_mask
        _, r_gl, r_ga = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], pm, ac)
        np.testing.assert_allclose(lds[3].cpu().numpy(), r_ga, **BWD_TOL[td])
        keep = ~kink_mask(c["loc"], c["shapes"], ac)
        np.testing.assert_allclose(np.where(keep, lds[2].cpu().numpy(), 0), np.where(keep, r_gl, 0), **BWD_TOL[td])


def test_lds_served_levels_are_chosen_by_default_at_c2_size():
    """the launcher's own plan (option 1, the default) takes the variant at the headline shape — and the result is
    still bit-identical to the plain kernel's"""
    from msda_triton_amd import _lib, synth
    assert _lib.get_option("lds_levels") in (1, 2)
    wl = synth.WORKLOADS["c2_q10k"]
    d = synth.make_inputs_numpy(wl, seed=3)
    c = {k: (v if k == "shapes" else v.astype(np.float32)) for k, v in d.items()}
    a = _forward(c, torch.float32, "border", True, 1)
    b = _forward(c, torch.float32, "border", True, 0)
    assert torch.equal(a, b)


# ---- two planes per workgroup (msda_set_option("lds_planes", 2): neighbouring heads share a workgroup, its waves take
#      slices of whichever plane has more left) — the same arithmetic per unit, so still bit-identical ----
@pytest.mark.parametrize("name", ["c2_like_f32", "all_levels_fit", "no_level_fits_but_last", "odd_points_boundary",
                                  "many_samples_two_trips", "d64_f32_g16"])
@pytest.mark.parametrize("pm,ac", MODES, ids=[mode_key(*m) for m in MODES])
def test_two_planes_per_workgroup_forward_is_bit_identical(name, pm, ac):
    from msda_triton_amd import _lib
    B, Q, H, D, levels, P, td = CASES[name]
    c = rand_case(np.random.default_rng(zlib.crc32(name.encode()) + 1), B, Q, H, D, levels, P)
    plain = _forward(c, td, pm, ac, 0)
    old = _lib.get_option("lds_planes")
    try:
        res = {}
        for planes in (1, 2):  # (H odd: 2 falls back to one plane per workgroup)
            _lib.set_option("lds_planes", planes)
            res[planes] = _forward(c, td, pm, ac, 2)
    finally:
        _lib.set_option("lds_planes", old)
    assert torch.equal(plain, res[1]) and torch.equal(plain, res[2])


@pytest.mark.parametrize("ref_dim", [2, 4])
def test_two_planes_per_workgroup_fused_forward_and_backward_are_bit_identical(ref_dim):
    from msda_triton_amd import _lib
    from msda_triton_amd.functional import fused_module_core
    g = torch.Generator().manual_seed(ref_dim)
    levels = [(16, 16), (8, 8), (4, 4), (2, 2)]
    B, Q, H, D, P = 2, 650, 4, 32, 4
    I = sum(h * w for h, w in levels)  # noqa: E741
    value = torch.randn(B, I, H, D, generator=g).to(DEV)
    proj = (torch.randn(B, Q, H, len(levels), P, 3, generator=g) * 2).to(DEV)
    ref = torch.rand(B, Q, ref_dim, generator=g).to(DEV)
    go = torch.randn(B, Q, H, D, generator=g).to(DEV)
    shapes = torch.tensor(levels, device=DEV)
    olds = {k: _lib.get_option(k) for k in ("lds_levels", "lds_planes")}
    res = {}
    try:
        for key, (lv, pl) in {"plain": (0, 1), "one": (2, 1), "two": (2, 2)}.items():
            _lib.set_option("lds_levels", lv)
            _lib.set_option("lds_planes", pl)
            v, pr = value.clone().requires_grad_(), proj.clone().requires_grad_()
            out = fused_module_core(v, shapes, pr, ref, "border", True)
            out.backward(go)
            res[key] = (out.detach(), pr.grad, v.grad)
    finally:
        for k, v_ in olds.items():
            _lib.set_option(k, v_)
    for key in ("one", "two"):
        for a, b in zip(res["plain"], res[key]):
            assert torch.equal(a, b), key


@pytest.mark.parametrize("lds", [0, 2], ids=["plain", "lds_levels"])
@pytest.mark.parametrize("name", ["c2_like_f32", "nothing_fits", "odd_points_boundary", "bf16_g4", "fp16_d64", "f64", "d64_f32_g16"])
def test_touched_rows_do_not_change_the_forward(name, lds):
    """``touch``: the workgroups request one dword of every row of their plane at the start (a cache warm-up for forwards of
    few queries, msda_launch.hpp ``touch_plan``); the values are dropped, so forcing it (2) must give the bits of never (0) —
    also with two planes per workgroup, where each half of the workgroup touches its own plane, with rows beyond the
    touched range (I > 4 rows per thread x threads) and with pyramids shorter than one round of touches."""
    from msda_triton_amd import _lib
    B, Q, H, D, levels, P, td = CASES[name]
    c = rand_case(np.random.default_rng(zlib.crc32(name.encode()) + 3), B, Q, H, D, levels, P,
                  dtype=np.float64 if td == torch.float64 else np.float32)
    old = {k: _lib.get_option(k) for k in ("touch", "lds_planes")}
    try:
        outs = {}
        for planes in (1, 2):
            _lib.set_option("lds_planes", planes)
            for t in (0, 2, 1):
                _lib.set_option("touch", t)
                outs[planes, t] = _forward(c, td, "zeros", False, lds)
        first = outs[1, 0]
        for k, o in outs.items():
            assert torch.equal(first, o), k
    finally:
        for k, v in old.items():
            _lib.set_option(k, v)
