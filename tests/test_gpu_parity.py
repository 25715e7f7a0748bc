"""GPU parity tests proper: the HIP path (through the C ABI in libmsda_hip.so) against the CPU
oracle, the reference's golden vectors and size-independent properties.  Run with ``-m gpu``.

Tolerances follow the reference's tests (/root/reference/tests/test_msda.py:15-27):
fp32 fwd atol 1e-4 rtol 1e-3, bwd atol 1e-3 rtol 1e-2; fp64 1e-8; fp16 1e-1.  The north-star
bar (1e-4 fp32 vs the native CPU fallback) is what the fp32 forward checks use.
"""
import zlib

import os

import numpy as np
import pytest
import torch

from conftest import MODES, case_id, digest_cases, golden_cases, kink_mask, mode_key

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
FWD_TOL = {torch.float32: dict(atol=1e-4, rtol=1e-3), torch.float64: dict(atol=1e-8, rtol=1e-8)}
BWD_TOL = {torch.float32: dict(atol=1e-3, rtol=1e-2), torch.float64: dict(atol=1e-8, rtol=1e-8)}


def _ops():
    import msda_triton_amd
    return msda_triton_amd


def run_hip(value, shapes, loc, attn, grad_out, pm, ac, dtype=None, needs_grad=True):
    """numpy in -> numpy out through the public API (autograd -> ctypes -> C ABI -> HIP)."""
    ops = _ops()
    td = dtype or torch.from_numpy(np.asarray(value)).dtype
    v = torch.from_numpy(np.asarray(value)).to(DEV, td)
    pad = int(os.environ.get("MSDA_TEST_VALUE_PAD", "0"))  # (tools/fuzz_parity.py: the same cases over padded value rows)
    if pad:
        from msda_triton_amd.functional import padded_value_rows
        vp = padded_value_rows(*v.shape, v.dtype, v.device, pad_bytes=pad * v.element_size())
        vp.copy_(v)
        v = vp
    v.requires_grad_(needs_grad)
    l = torch.from_numpy(np.asarray(loc)).to(DEV, td).requires_grad_(needs_grad)
    a = torch.from_numpy(np.asarray(attn)).to(DEV, td).requires_grad_(needs_grad)
    s = torch.from_numpy(np.asarray(shapes)).to(DEV)
    out = ops.multiscale_deformable_attention(v, s, l, a, pm, ac)
    if not needs_grad:
        return out.detach().float().cpu().numpy() if td in (torch.float16, torch.bfloat16) else out.detach().cpu().numpy()
    out.backward(torch.from_numpy(np.asarray(grad_out)).to(DEV, td))
    conv = (lambda t: t.detach().float().cpu().numpy()) if td in (torch.float16, torch.bfloat16) else (
        lambda t: t.detach().cpu().numpy())
    return conv(out), conv(v.grad), conv(l.grad), conv(a.grad)


def rand_case(rng, B, Q, H, D, levels, P, lo=-0.3, hi=1.3, dtype=np.float32):
    I = sum(h * w for h, w in levels)  # noqa: E741
    L = len(levels)
    return dict(
        value=rng.standard_normal((B, I, H, D)).astype(dtype),
        shapes=np.asarray(levels, dtype=np.int64),
        loc=rng.uniform(lo, hi, size=(B, Q, H, L, P, 2)).astype(dtype),
        attn=rng.uniform(0, 1, size=(B, Q, H, L, P)).astype(dtype),
        grad_out=rng.uniform(0, 1, size=(B, Q, H, D)).astype(dtype),
    )


def check_against_oracle(oracle, c, pm, ac, fwd_tol, bwd_tol, mask_kinks=True):
    out, gv, gl, ga = run_hip(c["value"], c["shapes"], c["loc"], c["attn"], c["grad_out"], pm, ac)
    r_out = oracle.forward(c["value"], c["shapes"], c["loc"], c["attn"], pm, ac)
    r_gv, r_gl, r_ga = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], pm, ac)
    np.testing.assert_allclose(out, r_out, err_msg="out", **fwd_tol)
    np.testing.assert_allclose(gv, r_gv, err_msg="grad_value", **bwd_tol)
    np.testing.assert_allclose(ga, r_ga, err_msg="grad_attn", **bwd_tol)
    if mask_kinks:
        keep = ~kink_mask(c["loc"], c["shapes"], ac)
        gl, r_gl = np.where(keep, gl, 0), np.where(keep, r_gl, 0)
    np.testing.assert_allclose(gl, r_gl, err_msg="grad_loc", **bwd_tol)


# ------------------------------------------------------------------------------------------
# golden vectors produced by the reference itself
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", golden_cases(), ids=case_id)
def test_hip_matches_reference_golden(path):
    z = np.load(path)
    td = torch.float32 if path.endswith("_f32.npz") else torch.float64
    for pm, ac in MODES:
        k = mode_key(pm, ac)
        out, gv, gl, ga = run_hip(z["value"], z["shapes"], z["loc"], z["attn"], z["grad_out"], pm, ac)
        np.testing.assert_allclose(out, z[f"out_{k}"], err_msg=f"out {k}", **FWD_TOL[td])
        np.testing.assert_allclose(gv, z[f"grad_value_{k}"], err_msg=f"grad_value {k}", **BWD_TOL[td])
        np.testing.assert_allclose(ga, z[f"grad_attn_{k}"], err_msg=f"grad_attn {k}", **BWD_TOL[td])
        r_gl = z[f"grad_loc_{k}"]
        if td == torch.float32:
            keep = ~kink_mask(z["loc"], z["shapes"], ac)
            gl, r_gl = np.where(keep, gl, 0), np.where(keep, r_gl, 0)
        np.testing.assert_allclose(gl, r_gl, err_msg=f"grad_loc {k}", **BWD_TOL[td])


@pytest.mark.parametrize("path", digest_cases(), ids=case_id)
def test_hip_matches_reference_digest_fullsize(path):
    """BASELINE-sized workloads (c1, c2 at Q = 1k / 5k / 10k, c4, the c3 shape in fp32) against digests of the
    REFERENCE's outputs (tests/golden/make_golden.py): 512 strided samples at the small cases' tolerances
    (out 1e-4, the north-star bar) plus sum / abs-sum of every tensor."""
    from msda_triton_amd import synth
    z = np.load(path)
    wl = synth.WORKLOADS[str(z["workload"])]
    d = synth.make_inputs_numpy(wl, seed=int(z["seed"]), loc_lo=float(z["loc_lo"]), loc_hi=float(z["loc_hi"]))
    f32 = {k: (v if k == "shapes" else v.astype(np.float32)) for k, v in d.items()}
    for pm, ac in MODES:
        k = mode_key(pm, ac)
        res = run_hip(f32["value"], f32["shapes"], f32["loc"], f32["attn"], f32["grad_out"], pm, ac)
        for nm, arr in zip(("out", "grad_value", "grad_loc", "grad_attn"), res):
            dg = synth.digest(arr)
            got, want = dg["samples"], z[f"{nm}_{k}_samples"].astype(np.float64)
            if nm == "grad_loc":  # not comparable where the pixel coordinate sits on a grid kink (conftest.kink_mask)
                kinks = kink_mask(f32["loc"], f32["shapes"], ac).reshape(-1)[::dg["step"]][:got.size]
                assert kinks.mean() < 0.02, f"{kinks.mean():.3%} of the sampled grad_loc entries sit on a grid kink"
                got, want = np.where(kinks, 0, got), np.where(kinks, 0, want)
            tol = FWD_TOL[torch.float32] if nm == "out" else BWD_TOL[torch.float32]
            np.testing.assert_allclose(got, want, err_msg=f"{nm} {k}", **tol)
            scale = max(1.0, float(z[f"{nm}_{k}_abs_sum"]))
            assert abs(dg["sum"] - float(z[f"{nm}_{k}_sum"])) <= 1e-4 * scale, (nm, k)
            assert abs(dg["abs_sum"] - float(z[f"{nm}_{k}_abs_sum"])) <= 1e-4 * scale, (nm, k)


# ------------------------------------------------------------------------------------------
# oracle on seeded random inputs: shape / dtype / mode matrix, incl. every kernel variant
# ------------------------------------------------------------------------------------------
SHAPE_MATRIX = {
    # name: (B, Q, H, D, levels, P)
    "d32_vec_g8": (2, 70, 8, 32, [(16, 16), (8, 8), (4, 4), (2, 2)], 4),
    "d64_vec_g16": (1, 33, 4, 64, [(9, 7), (5, 4)], 3),
    "d128_vec_g32": (1, 9, 2, 128, [(6, 6)], 2),
    "d256_vec_g64": (1, 5, 1, 256, [(4, 5), (2, 3)], 2),
    "d512_two_channel_chunks": (1, 3, 1, 512, [(3, 3)], 2),
    "d8_vec_g4": (2, 19, 3, 8, [(7, 9), (3, 4)], 5),
    "d5_scalar": (2, 13, 3, 5, [(6, 4), (3, 2), (2, 5)], 3),
    "d36_scalar_g64": (1, 7, 2, 36, [(5, 5), (2, 2)], 2),
    "d1": (1, 11, 2, 1, [(5, 6)], 3),
    "pairs_not_multiple_of_8": (3, 21, 5, 16, [(8, 8), (4, 4)], 2),
    "many_samples_chunked": (1, 6, 2, 8, [(6, 6), (3, 3), (2, 2), (1, 1), (4, 2), (2, 4), (5, 1), (1, 5)], 64),
    "one_query": (1, 1, 1, 32, [(4, 4)], 1),
    "big_level_pixel_ranges": (1, 50, 1, 3, [(210, 200), (10, 10)], 2),
}


@pytest.mark.parametrize("name", list(SHAPE_MATRIX), ids=list(SHAPE_MATRIX))
@pytest.mark.parametrize("pm,ac", MODES, ids=[mode_key(*m) for m in MODES])
def test_hip_vs_oracle_f32(oracle, name, pm, ac):
    B, Q, H, D, levels, P = SHAPE_MATRIX[name]
    c = rand_case(np.random.default_rng(zlib.crc32(name.encode())), B, Q, H, D, levels, P)
    check_against_oracle(oracle, c, pm, ac, FWD_TOL[torch.float32], BWD_TOL[torch.float32])


@pytest.mark.parametrize("name", ["d32_vec_g8", "d5_scalar", "d8_vec_g4", "big_level_pixel_ranges", "d512_two_channel_chunks"])
@pytest.mark.parametrize("pm,ac", MODES, ids=[mode_key(*m) for m in MODES])
def test_hip_vs_oracle_f64(oracle, name, pm, ac):
    B, Q, H, D, levels, P = SHAPE_MATRIX[name]
    c = rand_case(np.random.default_rng(7), B, Q, H, D, levels, P, dtype=np.float64)
    check_against_oracle(oracle, c, pm, ac, FWD_TOL[torch.float64], BWD_TOL[torch.float64], mask_kinks=False)


@pytest.mark.parametrize("td,atol,rtol", [(torch.float16, 2e-2, 2e-2), (torch.bfloat16, 4e-2, 2e-2)],
                         ids=["fp16", "bf16"])
@pytest.mark.parametrize("name", ["d32_vec_g8", "d64_vec_g16", "d5_scalar", "d8_vec_g4"])
@pytest.mark.parametrize("pm,ac", [("zeros", False), ("border", True)], ids=["zeros_0", "border_1"])
def test_hip_low_precision_vs_fp32_oracle_on_rounded_inputs(oracle, td, atol, rtol, name, pm, ac):
    """fp16/bf16 oracle = fp32 math on inputs rounded to the low dtype; the reference's
    own fp16 tolerance is (1e-1, 1e-1) (tests/test_msda.py:16-18)."""
    B, Q, H, D, levels, P = SHAPE_MATRIX[name]
    c = rand_case(np.random.default_rng(11), B, Q, H, D, levels, P, lo=-0.1, hi=1.1)
    rounded = {k: (v if k == "shapes" else torch.from_numpy(v).to(td).float().numpy()) for k, v in c.items()}
    out, gv, gl, ga = run_hip(rounded["value"], rounded["shapes"], rounded["loc"], rounded["attn"],
                              rounded["grad_out"], pm, ac, dtype=td)
    r_out = oracle.forward(rounded["value"], rounded["shapes"], rounded["loc"], rounded["attn"], pm, ac)
    r_gv, r_gl, r_ga = oracle.backward(rounded["grad_out"], rounded["value"], rounded["shapes"], rounded["loc"],
                                       rounded["attn"], pm, ac)
    np.testing.assert_allclose(out, r_out, atol=atol, rtol=rtol, err_msg="out")
    np.testing.assert_allclose(ga, r_ga, atol=atol * 4, rtol=rtol, err_msg="grad_attn")
    np.testing.assert_allclose(gv, r_gv, atol=atol * 4, rtol=rtol, err_msg="grad_value")
    keep = ~kink_mask(rounded["loc"], rounded["shapes"], ac, tol=2e-2)
    scale = max(1.0, float(np.abs(r_gl).max()))
    np.testing.assert_allclose(np.where(keep, gl, 0), np.where(keep, r_gl, 0), atol=atol * scale, rtol=rtol * 2,
                               err_msg="grad_loc")


def test_far_oob_and_nonfinite_free(oracle):
    rng = np.random.default_rng(3)
    c = rand_case(rng, 1, 40, 2, 32, [(8, 8), (3, 5)], 4, lo=-50.0, hi=50.0)
    c["loc"].reshape(-1)[::9] = 1e30
    c["loc"].reshape(-1)[1::13] = -1e30
    for pm, ac in MODES:
        check_against_oracle(oracle, c, pm, ac, FWD_TOL[torch.float32], BWD_TOL[torch.float32])
        out = run_hip(c["value"], c["shapes"], c["loc"], c["attn"], None, pm, ac, needs_grad=False)
        assert np.isfinite(out).all()


def test_zeros_padding_never_reads_masked_corners():
    """tl.where semantics (kernels.py:220-231): a masked corner contributes exactly 0 even when the
    clamped pixel it would alias holds nan (a weight-zero multiply would give nan)."""
    ops = _ops()
    shapes = torch.tensor([(4, 4)], device=DEV)
    att = torch.ones(1, 2, 1, 1, 1, device=DEV, requires_grad=True)
    value = torch.full((1, 16, 1, 32), float("nan"), device=DEV)
    value[:, 15] = 3.0
    # align_corners=True on 4x4: loc 1.2 -> pixel coordinate 3.6 (x0 = 3 valid, x1 = 4 masked)
    #   query 0: (3.6, 3.6): only corner (3,3) is valid, weight 0.4*0.4
    #   query 1: (3.6, 9.0): every corner is masked (y0 = 9) but all of them clamp onto nan pixels' row 3
    loc = torch.tensor([[1.2, 1.2], [1.2, 3.0]], device=DEV).reshape(1, 2, 1, 1, 1, 2).requires_grad_(True)
    out = ops.multiscale_deformable_attention(value, shapes, loc, att, "zeros", True)
    assert torch.allclose(out[0, 0], torch.full_like(out[0, 0], 3.0 * 0.16), atol=1e-6)
    assert torch.count_nonzero(out[0, 1]) == 0 and torch.isfinite(out).all()
    value[:, 15] = float("nan")
    value[:, 11] = float("nan")
    out = ops.multiscale_deformable_attention(value, shapes, loc[:, 1:], att[:, 1:], "zeros", True)
    assert torch.count_nonzero(out) == 0
    out.sum().backward()
    assert torch.count_nonzero(loc.grad[:, 1]) == 0 and torch.count_nonzero(att.grad[:, 1]) == 0


# ------------------------------------------------------------------------------------------
# properties at BASELINE.json's full sizes
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("wl_name", ["c2_q10k", "c4_gdino_dec"])
def test_fullsize_properties(wl_name):
    """Linearity in value and in the attention weights, query-permutation equivariance, and
    sum(grad_value) == sum_q attn-weighted grad_out mass (border mode: bilinear weights sum to 1)."""
    from msda_triton_amd import synth
    ops = _ops()
    wl = synth.WORKLOADS[wl_name]
    d = synth.make_inputs_torch(wl, DEV, seed=0)
    v, s, l, a, g = d["value"], d["shapes"], d["loc"], d["attn"], d["grad_out"]
    pm, ac = wl.padding_mode, wl.align_corners
    f = lambda vv, aa, ll=l: ops.multiscale_deformable_attention(vv, s, ll, aa, pm, ac)  # noqa: E731
    out = f(v, a)
    assert out.shape == (wl.B, wl.Q, wl.H, wl.D) and torch.isfinite(out).all()
    v2 = torch.randn_like(v)
    torch.testing.assert_close(f(v + 2 * v2, a), out + 2 * f(v2, a), atol=2e-4, rtol=1e-4)
    torch.testing.assert_close(f(v, 3 * a), 3 * out, atol=2e-4, rtol=1e-4)
    perm = torch.randperm(wl.Q, device=DEV)
    torch.testing.assert_close(f(v, a[:, perm], l[:, perm]), out[:, perm], atol=0, rtol=0)
    # mass conservation of the scatter in border mode (every sample's 4 weights sum to 1)
    vb = v.clone().requires_grad_(True)
    ob = ops.multiscale_deformable_attention(vb, s, l, a, "border", True)
    ob.backward(g)
    expect = torch.einsum("bqhd,bqh->bhd", g.double(), a.double().sum((-1, -2)))
    got = vb.grad.double().sum(1)
    torch.testing.assert_close(got, expect, atol=1e-2, rtol=1e-4)


def test_c3_encoder_shape_bf16_and_c5_slice_fp16(oracle):
    """BASELINE configs[2] at full size in bf16 and a query slice of configs[4] in fp16, each against
    the fp32 oracle on a strided subset of queries."""
    from msda_triton_amd import synth
    ops = _ops()
    for wl_name, q_take, td, atol in (("c3_ddetr_enc", 17821, torch.bfloat16, 6e-2), ("c5_stress", 2000, torch.float16, 2e-2)):
        wl = synth.WORKLOADS[wl_name]
        d = synth.make_inputs_torch(wl, "cpu", seed=0, q_end=q_take, loc_lo=-0.05, loc_hi=1.05)
        dd = {k: t.to(DEV) for k, t in d.items()}
        out = ops.multiscale_deformable_attention(dd["value"], dd["shapes"], dd["loc"], dd["attn"],
                                                  wl.padding_mode, wl.align_corners)
        sel = slice(0, q_take, max(1, q_take // 64))
        r = oracle.forward(d["value"].float().numpy(), d["shapes"].numpy(), d["loc"][:, sel].float().numpy(),
                           d["attn"][:, sel].float().numpy(), wl.padding_mode, wl.align_corners)
        np.testing.assert_allclose(out[:, sel].float().cpu().numpy(), r, atol=atol, rtol=2e-2)


# ------------------------------------------------------------------------------------------
# API behaviour on the GPU
# ------------------------------------------------------------------------------------------
def test_gradcheck_fp64():
    ops = _ops()
    rng = np.random.default_rng(5)
    levels = [(5, 4), (3, 2)]
    c = rand_case(rng, 1, 3, 2, 4, levels, 2, lo=-0.4, hi=1.4, dtype=np.float64)
    # keep samples away from the pixel-grid kinks so finite differences are valid
    shapes = torch.tensor(levels, device=DEV)
    for pm, ac in MODES:
        k = kink_mask(c["loc"], c["shapes"], ac, tol=5e-3)
        loc = np.where(k, c["loc"] + 0.031, c["loc"])
        v = torch.from_numpy(c["value"]).to(DEV).requires_grad_(True)
        l = torch.from_numpy(loc).to(DEV).requires_grad_(True)
        a = torch.from_numpy(c["attn"]).to(DEV).requires_grad_(True)
        assert torch.autograd.gradcheck(
            lambda vv, ll, aa: ops.multiscale_deformable_attention(vv, shapes, ll, aa, pm, ac),
            (v, l, a), eps=1e-6, atol=1e-5, rtol=1e-4, nondet_tol=1e-12)


def test_noncontiguous_inputs_and_int32_shapes(oracle):
    ops = _ops()
    rng = np.random.default_rng(9)
    c = rand_case(rng, 2, 17, 4, 32, [(6, 6), (3, 3)], 2)
    v = torch.from_numpy(c["value"]).to(DEV).permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)  # dense, permuted
    l = torch.from_numpy(c["loc"]).to(DEV)
    l_big = torch.zeros(2, 17, 4, 2, 2, 4, device=DEV)
    l_big[..., :2] = l
    a = torch.from_numpy(c["attn"]).to(DEV)
    s32 = torch.from_numpy(c["shapes"]).to(DEV, torch.int32)
    assert not v.is_contiguous() and not l_big[..., :2].is_contiguous()
    v.requires_grad_(True)
    out = ops.multiscale_deformable_attention(v, s32, l_big[..., :2], a, "zeros", False)
    out.backward(torch.from_numpy(c["grad_out"]).to(DEV))
    r = oracle.forward(c["value"], c["shapes"], c["loc"], c["attn"], "zeros", False)
    r_gv, _, _ = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], "zeros", False)
    np.testing.assert_allclose(out.detach().cpu().numpy(), r, **FWD_TOL[torch.float32])
    np.testing.assert_allclose(v.grad.cpu().numpy(), r_gv, **BWD_TOL[torch.float32])


def test_unaligned_views_take_the_scalar_path(oracle):
    """A storage offset that breaks 16-byte alignment must still give right answers."""
    ops = _ops()
    rng = np.random.default_rng(10)
    c = rand_case(rng, 1, 9, 2, 32, [(4, 4)], 2)
    flat = torch.zeros(c["value"].size + 1, device=DEV)
    flat[1:] = torch.from_numpy(c["value"]).to(DEV).reshape(-1)
    v = flat[1:].reshape(c["value"].shape)  # contiguous but only 4-byte aligned
    assert v.data_ptr() % 16 != 0 and v.is_contiguous()
    out = ops.multiscale_deformable_attention(v, torch.from_numpy(c["shapes"]).to(DEV), torch.from_numpy(c["loc"]).to(DEV),
                                              torch.from_numpy(c["attn"]).to(DEV), "border", False)
    r = oracle.forward(c["value"], c["shapes"], c["loc"], c["attn"], "border", False)
    np.testing.assert_allclose(out.cpu().numpy(), r, **FWD_TOL[torch.float32])


def test_partial_needs_input_grad(oracle):
    ops = _ops()
    c = rand_case(np.random.default_rng(12), 1, 10, 2, 32, [(5, 5)], 3)
    s = torch.from_numpy(c["shapes"]).to(DEV)
    g = torch.from_numpy(c["grad_out"]).to(DEV)
    r_gv, r_gl, r_ga = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], "zeros", True)
    for which in ((True, False, False), (False, True, False), (False, False, True), (False, True, True)):
        v, l, a = (torch.from_numpy(c[k]).to(DEV).requires_grad_(w) for k, w in zip(("value", "loc", "attn"), which))
        ops.multiscale_deformable_attention(v, s, l, a, "zeros", True).backward(g)
        for t, w, r in ((v, which[0], r_gv), (l, which[1], r_gl), (a, which[2], r_ga)):
            assert (t.grad is not None) == w
            if w:
                np.testing.assert_allclose(t.grad.cpu().numpy(), r, **BWD_TOL[torch.float32])


def test_forward_is_bitwise_deterministic_and_backward_stable():
    from msda_triton_amd import synth
    ops = _ops()
    d = synth.make_inputs_torch(synth.WORKLOADS["c1_readme"], DEV, seed=3)
    outs, gvs, gls = [], [], []
    for _ in range(2):
        v, l, a = (d[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
        o = ops.multiscale_deformable_attention(v, d["shapes"], l, a, "zeros", False)
        o.backward(d["grad_out"])
        outs.append(o.detach()); gvs.append(v.grad); gls.append((l.grad, a.grad))
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(gls[0][0], gls[1][0]) and torch.equal(gls[0][1], gls[1][1])  # private per sample: exact
    torch.testing.assert_close(gvs[0], gvs[1], atol=1e-4, rtol=1e-4)  # the order of the records inside a cell list may vary (last bit)


def test_xcd_map_on_off_same_results():
    from msda_triton_amd import _lib, synth
    ops = _ops()
    d = synth.make_inputs_torch(synth.WORKLOADS["c1_readme"], DEV, seed=4)
    res = []
    try:
        for flag in (1, 0):
            _lib.set_option("xcd_map", flag)
            v = d["value"].clone().requires_grad_(True)
            o = ops.multiscale_deformable_attention(v, d["shapes"], d["loc"], d["attn"], "border", True)
            o.backward(d["grad_out"])
            res.append((o.detach(), v.grad))
    finally:
        _lib.set_option("xcd_map", 1)
    assert torch.equal(res[0][0], res[1][0])
    torch.testing.assert_close(res[0][1], res[1][1], atol=1e-4, rtol=1e-4)


def test_empty_and_degenerate_inputs():
    ops = _ops()
    s = torch.tensor([[3, 3]], device=DEV)
    v = torch.randn(2, 9, 2, 8, device=DEV, requires_grad=True)
    out = ops.multiscale_deformable_attention(v, s, torch.rand(2, 0, 2, 1, 2, 2, device=DEV),
                                              torch.rand(2, 0, 2, 1, 2, device=DEV), "zeros", False)
    assert out.shape == (2, 0, 2, 8)
    out.sum().backward()
    assert torch.count_nonzero(v.grad) == 0
    # zero points per level: empty sum -> zeros
    out = ops.multiscale_deformable_attention(v.detach(), s, torch.rand(2, 5, 2, 1, 0, 2, device=DEV),
                                              torch.rand(2, 5, 2, 1, 0, device=DEV), "border", True)
    assert out.shape == (2, 5, 2, 8) and torch.count_nonzero(out) == 0


def test_errors_match_reference_contract():
    ops = _ops()
    s = torch.tensor([[2, 2]], device=DEV)
    v = torch.randn(1, 4, 1, 4, device=DEV)
    l = torch.rand(1, 1, 1, 1, 1, 2, device=DEV)
    a = torch.rand(1, 1, 1, 1, 1, device=DEV)
    with pytest.raises(ValueError):  # unsupported dtype (frontend.py:84-90)
        ops.hip_multiscale_deformable_attention(v.to(torch.int32), s, l, a, "zeros", False)
    with pytest.raises(ValueError):  # inputs not on the gpu (frontend.py:93-95)
        ops.hip_multiscale_deformable_attention(v.cpu(), s, l, a, "zeros", False)
    with pytest.raises(ValueError):  # shapes tensor on the host while the rest is on the gpu
        ops.multiscale_deformable_attention(v, s.cpu(), l, a, "zeros", False)
    with pytest.raises(ValueError):
        ops.multiscale_deformable_attention(v, s, l, a, "reflection", False)
    with pytest.raises(ValueError):  # mixed dtypes
        ops.multiscale_deformable_attention(v, s, l.double(), a, "zeros", False)
    with pytest.raises(ValueError):  # too many levels for the C ABI (MSDA_ERR_TOO_MANY_LEVELS)
        L = 33
        ops.multiscale_deformable_attention(torch.randn(1, L, 1, 4, device=DEV), torch.ones(L, 2, dtype=torch.int64, device=DEV),
                                            torch.rand(1, 1, 1, L, 1, 2, device=DEV), torch.rand(1, 1, 1, L, 1, device=DEV),
                                            "zeros", False)


def test_launchers_and_module_core_validate_devices():
    """A host or foreign-device pointer must never reach a kernel: the launcher pair and the fused
    module core raise ValueError for host tensors (reference: frontend.py:93-95), and the nn.Module accepts a
    host-resident img_shapes next to GPU tensors (reference: works through its fallback)."""
    from msda_triton_amd.functional import (fused_module_core, msda_hip_bwd, msda_hip_bwd_fused, msda_hip_fwd,
                                            msda_hip_fwd_fused)
    ops = _ops()
    c = rand_case(np.random.default_rng(8), 1, 6, 2, 8, [(4, 3), (2, 2)], 2)
    v, l, a, g = (torch.from_numpy(c[k]).to(DEV) for k in ("value", "loc", "attn", "grad_out"))
    s = torch.from_numpy(c["shapes"]).to(DEV)
    proj = torch.randn(1, 6, 2, 2, 2, 3, device=DEV)
    ref = torch.rand(1, 6, 2, device=DEV)
    with pytest.raises(ValueError):
        msda_hip_fwd(v, s.cpu(), l, a, "zeros", False)
    with pytest.raises(ValueError):
        msda_hip_fwd(v, s, l.cpu(), a, "zeros", False)
    with pytest.raises(ValueError):
        msda_hip_bwd(g.cpu(), v, s, l, a, "zeros", False)
    with pytest.raises(ValueError):
        msda_hip_fwd_fused(v, s.cpu(), proj, ref, "zeros", False)
    with pytest.raises(ValueError):
        msda_hip_bwd_fused(g, v, s, proj, ref.cpu(), "zeros", False)
    with pytest.raises(ValueError):
        fused_module_core(v, s, proj.cpu(), ref, "zeros", False)
    want = fused_module_core(v, s, proj, ref, "zeros", False)
    torch.testing.assert_close(fused_module_core(v, s.cpu(), proj, ref, "zeros", False), want, atol=0, rtol=0)
    torch.manual_seed(1)
    m = ops.MultiscaleDeformableAttention(16, 16, 2, 2, 2, "border", True).to(DEV)
    img, q = torch.randn(1, 16, 16, device=DEV), torch.randn(1, 5, 16, device=DEV)
    for coords in (2, 4):
        r = torch.rand(1, 5, coords, device=DEV)
        torch.testing.assert_close(m(img, s.cpu(), q, r), m(img, s, q, r), atol=0, rtol=0)


@pytest.mark.parametrize("td", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_autocast_runs_in_fp32(td):
    """Reference: custom_fwd(cast_inputs=float32) (frontend.py:111) and tests/test_msda.py:171-182."""
    ops = _ops()
    c = rand_case(np.random.default_rng(2), 1, 12, 2, 32, [(4, 4), (2, 2)], 2)
    v, l, a = (torch.from_numpy(c[k]).to(DEV, torch.float16) for k in ("value", "loc", "attn"))
    s = torch.from_numpy(c["shapes"]).to(DEV)
    with torch.amp.autocast(device_type="cuda", dtype=td):
        out = ops.multiscale_deformable_attention(v, s, l, a, "zeros", False)
    assert out.dtype == torch.float32
    ref = ops.multiscale_deformable_attention(v.float(), s, l.float(), a.float(), "zeros", False)
    torch.testing.assert_close(out, ref, atol=0, rtol=0)


@pytest.mark.parametrize("coords", [2, 4])
def test_module_gpu_matches_cpu_native(coords):
    """Reference smoke test tests/test_msda.py:154-168, plus a numeric check against the host path."""
    ops = _ops()
    torch.manual_seed(0)
    emb, heads, levels, points = 256, 8, 4, 8
    shapes = [(64 // 2**i, 64 // 2**i) for i in range(levels)]
    I = sum(h * w for h, w in shapes)  # noqa: E741
    m = ops.MultiscaleDeformableAttention(emb, emb // heads, levels, heads, points, "border", True)
    img, q, ref = torch.randn(2, I, emb), torch.randn(2, 100, emb), torch.randn(2, 100, coords)
    s = torch.tensor(shapes)
    out_cpu = m(img, s, q, ref)
    m_gpu = m.to(DEV)
    out_gpu = m_gpu(img.to(DEV), s.to(DEV), q.to(DEV), ref.to(DEV))
    assert out_gpu.shape == (2, 100, emb)
    torch.testing.assert_close(out_gpu.cpu(), out_cpu, atol=2e-3, rtol=1e-3)
    out_gpu.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m_gpu.parameters())


def test_reference_import_path_shim_on_gpu():
    import msda_triton
    from msda_triton.frontend import triton_multiscale_deformable_attention
    s = torch.tensor([[4, 4]], device=DEV)
    v = torch.randn(1, 16, 2, 32, device=DEV)
    l = torch.rand(1, 3, 2, 1, 2, 2, device=DEV)
    a = torch.rand(1, 3, 2, 1, 2, device=DEV)
    o1 = msda_triton.multiscale_deformable_attention(v, s, l, a, "zeros", False)
    o2 = triton_multiscale_deformable_attention(v, s, l, a, "zeros", False)
    assert torch.equal(o1, o2)


def test_native_library_is_loaded_in_process():
    """The HIP path must be the one that ran: libmsda_hip.so is mapped into this process."""
    from msda_triton_amd import _lib
    _lib.load()
    with open("/proc/self/maps") as f:
        assert _lib.LIB_NAME in f.read()


def test_graph_capture_replays():
    """No allocation / sync inside the C ABI calls: a fwd+bwd pair captures into a hipGraph."""
    ops = _ops()
    c = rand_case(np.random.default_rng(21), 2, 64, 8, 32, [(8, 8), (4, 4)], 4)
    v, l, a, g = (torch.from_numpy(c[k]).to(DEV) for k in ("value", "loc", "attn", "grad_out"))
    s = torch.from_numpy(c["shapes"]).to(DEV)
    eager = ops.msda_hip_fwd(v, s, l, a, "zeros", False)
    eager_g = ops.msda_hip_bwd(g, v, s, l, a, "zeros", False)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.msda_hip_fwd(v, s, l, a, "zeros", False)  # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = ops.msda_hip_fwd(v, s, l, a, "zeros", False)
        grads = ops.msda_hip_bwd(g, v, s, l, a, "zeros", False)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    torch.testing.assert_close(grads[0], eager_g[0], atol=1e-4, rtol=1e-4)
    assert torch.equal(grads[1], eager_g[1]) and torch.equal(grads[2], eager_g[2])


@pytest.mark.parametrize("wl_name,force", [("c4_gdino_dec", True), ("c2_q10k", False)], ids=["c4_forced", "c2_q10k_default"])
def test_graph_capture_replays_with_the_side_stream_fork(wl_name, force):
    """The backward with the sample-gradient kernel forked onto the side stream still captures into one hipGraph (the
    fork / join is event-based); replays give the eager results.  At the Grounding-DINO decoder size (c4: 921 600
    samples, single-launch grad_value kernel) the fork is forced — since round 3 it no longer pays there and is off by
    default; at c2 @ 10k (5.1 M samples, sorted pipeline with its caller-side workspace) it is on by default."""
    from msda_triton_amd import _lib, synth
    ops = _ops()
    assert _lib.get_option("overlap") == -1  # automatic
    wl = synth.WORKLOADS[wl_name]
    d = synth.make_inputs_torch(wl, DEV, seed=3)
    v, l, a, g, s = d["value"], d["loc"], d["attn"], d["grad_out"], d["shapes"]
    try:
        if force:
            _lib.set_option("overlap", 1)
        eager = ops.msda_hip_fwd(v, s, l, a, wl.padding_mode, wl.align_corners)
        eager_g = ops.msda_hip_bwd(g, v, s, l, a, wl.padding_mode, wl.align_corners)  # (also creates this thread's side stream)
        torch.cuda.synchronize()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            ops.msda_hip_bwd(g, v, s, l, a, wl.padding_mode, wl.align_corners)  # warm-up on the capture stream
        torch.cuda.current_stream().wait_stream(cap)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = ops.msda_hip_fwd(v, s, l, a, wl.padding_mode, wl.align_corners)
            grads = ops.msda_hip_bwd(g, v, s, l, a, wl.padding_mode, wl.align_corners)
        for _ in range(3):
            out.zero_()
            for t in grads:
                t.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, eager)
            torch.testing.assert_close(grads[0], eager_g[0], atol=1e-4, rtol=1e-4)
            assert torch.equal(grads[1], eager_g[1]) and torch.equal(grads[2], eager_g[2])
    finally:
        _lib.set_option("overlap", -1)


def test_two_host_threads_run_backwards_on_one_device():
    """Autograd worker threads / user threads with their own streams: each host thread has its own side stream and
    event pair (msda_api.hip), so concurrent backwards cannot wait on each other's fork records.  Two threads, each on
    its own stream, c4-sized problems (fork/join forced on) with different data: both get their own eager results."""
    import threading
    from msda_triton_amd import _lib, synth
    ops = _ops()
    _lib.set_option("overlap", 1)
    wl = synth.WORKLOADS["c4_gdino_dec"]
    data = [synth.make_inputs_torch(wl, DEV, seed=20 + i) for i in range(2)]
    want = [ops.msda_hip_bwd(d["grad_out"], d["value"], d["shapes"], d["loc"], d["attn"], wl.padding_mode, wl.align_corners)
            for d in data]
    torch.cuda.synchronize()
    got, errs = [None, None], []

    def work(i):
        try:
            st = torch.cuda.Stream()
            d = data[i]
            with torch.cuda.stream(st):
                for _ in range(20):
                    # the producer of grad_out runs on this thread's stream right before the backward: a missed
                    # fork dependency would read a stale / half-written buffer
                    go = d["grad_out"] * 1.0
                    res = ops.msda_hip_bwd(go, d["value"], d["shapes"], d["loc"], d["attn"], wl.padding_mode, wl.align_corners)
                st.synchronize()
            got[i] = res
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    try:
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        _lib.set_option("overlap", -1)
    assert not errs, errs
    for i in range(2):
        torch.testing.assert_close(got[i][0], want[i][0], atol=1e-4, rtol=1e-4)
        assert torch.equal(got[i][1], want[i][1]) and torch.equal(got[i][2], want[i][2])


# ------------------------------------------------------------------------------------------
# backward grad_value: both implementations, and the shapes that stress the sorted gather
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("value_path", [2, 3], ids=["sorted_gather", "single_launch"])
@pytest.mark.parametrize("pm,ac", MODES, ids=[mode_key(*m) for m in MODES])
def test_grad_value_both_paths_and_hot_pixels(oracle, value_path, pm, ac):
    """Every query samples the same few pixels (lists far longer than one work-item chunk, so pixels are
    split into many chunks and summed by the finish kernel) plus a uniform background."""
    from msda_triton_amd import _lib
    rng = np.random.default_rng(31)
    c = rand_case(rng, 2, 300, 2, 32, [(9, 7), (4, 4), (1, 1)], 4, lo=-0.2, hi=1.2)
    hot = c["loc"][:, :, :, 0]                       # level 0: concentrate 3/4 of the points around one spot
    hot[:, :, :, :3] = 0.43 + 0.02 * rng.standard_normal(hot[:, :, :, :3].shape)
    try:
        _lib.set_option("value_path", value_path)
        check_against_oracle(oracle, c, pm, ac, FWD_TOL[torch.float32], BWD_TOL[torch.float32])
    finally:
        _lib.set_option("value_path", 0)


@pytest.mark.parametrize("D,Q,td", [(32, 3000, torch.float32), (512, 700, torch.float32), (32, 3000, torch.float64),
                                    (32, 3000, torch.bfloat16)],
                         ids=["f32_d32", "f32_d512_two_channel_chunks", "f64_d32", "bf16_d32"])
@pytest.mark.parametrize("value_path", [2, 3], ids=["sorted_gather", "single_launch"])
def test_grad_value_cell_flood_spans_gather_workgroups(oracle, D, Q, td, value_path):
    """A 1x1 level: every sample falls into one of four cells, so a cell's list is cut into far more work items
    than one gather workgroup holds — items are merged inside workgroups, the finish kernel adds one row per
    workgroup, and the cell-scan kernel writes these records with the whole block."""
    from msda_triton_amd import _lib
    rng = np.random.default_rng(77)
    npdt = np.float64 if td == torch.float64 else np.float32
    c = rand_case(rng, 1, Q, 2, D, [(1, 1), (3, 2)], 4, lo=0.0, hi=1.0, dtype=npdt)
    c["grad_out"] = (c["grad_out"] / Q).astype(npdt)   # keep the sums O(1)
    if td == torch.bfloat16:
        for k in ("value", "loc", "attn", "grad_out"):
            c[k] = torch.from_numpy(c[k]).to(td).float().numpy()
    try:
        _lib.set_option("value_path", value_path)
        for pm, ac in (("zeros", False), ("border", True)):
            _, gv, _, _ = run_hip(c["value"], c["shapes"], c["loc"], c["attn"], c["grad_out"], pm, ac, dtype=td)
            r_gv, _, _ = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], pm, ac)
            tol = dict(atol=2e-2, rtol=2e-2) if td == torch.bfloat16 else BWD_TOL[td]
            np.testing.assert_allclose(gv, r_gv, err_msg=f"{pm} {ac}", **tol)
    finally:
        _lib.set_option("value_path", 0)


def test_grad_value_without_workspace(oracle):
    """The C ABI accepts workspace == NULL for problems the single-launch kernel takes (its inverted index lives in
    LDS); a larger problem without workspace is an argument error for grad_value — nothing is launched, the message
    names the size — while grad_loc / grad_attn alone never need one."""
    from msda_triton_amd import _lib
    lib = _lib.load()
    c = rand_case(np.random.default_rng(32), 1, 40, 2, 32, [(6, 6), (3, 3)], 3)
    v, l, a, g = (torch.from_numpy(c[k]).to(DEV) for k in ("value", "loc", "attn", "grad_out"))
    s = torch.from_numpy(c["shapes"]).to(DEV)
    gv, gl, ga = torch.empty_like(v), torch.empty_like(l), torch.empty_like(a)
    B, I, H, D = v.shape
    _, Q, _, L, P, _ = l.shape
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.msda_bwd_f32(g.data_ptr(), v.data_ptr(), s.data_ptr(), l.data_ptr(), a.data_ptr(),
                          gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), B, I, H, D, Q, L, P, 1, 0, 0, 0, None, 0, st)
    assert rc == 0
    torch.cuda.synchronize()
    r_gv, r_gl, r_ga = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], "zeros", False)
    np.testing.assert_allclose(gv.cpu().numpy(), r_gv, **BWD_TOL[torch.float32])
    np.testing.assert_allclose(ga.cpu().numpy(), r_ga, **BWD_TOL[torch.float32])
    # a problem beyond the single-launch kernel (Q * P > 4096 samples per plane and level)
    c = rand_case(np.random.default_rng(33), 1, 3000, 2, 32, [(6, 6), (3, 3)], 3)
    v, l, a, g = (torch.from_numpy(c[k]).to(DEV) for k in ("value", "loc", "attn", "grad_out"))
    gv, gl, ga = torch.empty_like(v), torch.empty_like(l), torch.empty_like(a)
    B, I, H, D = v.shape
    _, Q, _, L, P, _ = l.shape
    need = lib.msda_bwd_workspace_bytes(B, I, H, D, Q, L, P, 4, 4, 0, 0)
    assert need > 0
    args = (g.data_ptr(), v.data_ptr(), s.data_ptr(), l.data_ptr(), a.data_ptr())
    dims = (B, I, H, D, Q, L, P, 1, 0, 0, 0)  # (..., padding_mode, align_corners, max_level_cells, value_row_stride)
    assert lib.msda_bwd_f32(*args, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), *dims, None, 0, st) == -1
    assert str(need) in lib.msda_last_error().decode()
    small_ws = torch.empty(need // 8, dtype=torch.uint8, device=DEV)
    assert lib.msda_bwd_f32(*args, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), *dims, small_ws.data_ptr(), need // 8, st) == -1
    # with all three gradients in one call the sorted records may live in the grad_loc / grad_attn buffers: the size
    # msda_bwd_workspace_bytes reports for that (MSDA_WS_RECORDS_IN_GRADS) is smaller and suffices — for that call only
    lean = lib.msda_bwd_workspace_bytes(B, I, H, D, Q, L, P, 4, 4, 0, _lib.WS_RECORDS_IN_GRADS)
    assert 0 < lean < need
    lean_ws = torch.empty(lean, dtype=torch.uint8, device=DEV)
    assert lib.msda_bwd_f32(*args, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), *dims, lean_ws.data_ptr(), lean, st) == 0
    torch.cuda.synchronize()
    r_gv2, r_gl2, r_ga2 = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], "zeros", False)
    np.testing.assert_allclose(gv.cpu().numpy(), r_gv2, **BWD_TOL[torch.float32])
    np.testing.assert_allclose(ga.cpu().numpy(), r_ga2, **BWD_TOL[torch.float32])
    assert lib.msda_bwd_f32(*args, gv.data_ptr(), None, None, *dims, lean_ws.data_ptr(), lean, st) == -1  # grad_value alone needs the full size
    assert lib.msda_bwd_f32(*args, None, gl.data_ptr(), ga.data_ptr(), *dims, None, 0, st) == 0  # no grad_value: no workspace
    torch.cuda.synchronize()
    _, r_gl, r_ga = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], "zeros", False)
    np.testing.assert_allclose(ga.cpu().numpy(), r_ga, **BWD_TOL[torch.float32])
    # a pyramid large enough for the third home of the records: plane 0's in grad_loc, plane 1's in grad_value itself
    # (written only by the finish kernel, behind the gather) — no record region left in the workspace
    c = rand_case(np.random.default_rng(34), 1, 2500, 2, 32, [(40, 40), (20, 20)], 2, lo=-0.05, hi=1.05)
    v, l, a, g = (torch.from_numpy(c[k]).to(DEV) for k in ("value", "loc", "attn", "grad_out"))
    s = torch.from_numpy(c["shapes"]).to(DEV)
    B, I, H, D = v.shape
    _, Q, _, L, P, _ = l.shape
    gvb = torch.empty(v.numel() + 4, device=DEV)
    gv, gl, ga = gvb[:v.numel()].view_as(v), torch.empty_like(l), torch.empty_like(a)
    args = (g.data_ptr(), v.data_ptr(), s.data_ptr(), l.data_ptr(), a.data_ptr())
    dims = (B, I, H, D, Q, L, P, 1, 0, 0, 0)  # (..., padding_mode, align_corners, max_level_cells, value_row_stride)
    full = lib.msda_bwd_workspace_bytes(B, I, H, D, Q, L, P, 4, 4, 0, 0)
    lean = lib.msda_bwd_workspace_bytes(B, I, H, D, Q, L, P, 4, 4, 0, _lib.WS_RECORDS_IN_GRADS)
    assert 0 < lean <= full - 2 * Q * L * P * 16  # both planes' records are gone from it
    lean_ws = torch.empty(lean, dtype=torch.uint8, device=DEV)
    assert lib.msda_bwd_f32(*args, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), *dims, lean_ws.data_ptr(), lean, st) == 0
    torch.cuda.synchronize()
    r_gv, r_gl, r_ga = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], "zeros", False)
    np.testing.assert_allclose(gv.cpu().numpy(), r_gv, **BWD_TOL[torch.float32])
    np.testing.assert_allclose(ga.cpu().numpy(), r_ga, **BWD_TOL[torch.float32])
    full_ws = torch.empty(full, dtype=torch.uint8, device=DEV)
    gv2 = torch.empty_like(v)
    assert lib.msda_bwd_f32(*args, gv2.data_ptr(), None, None, *dims, full_ws.data_ptr(), full, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(gv2, gv)  # same records, same order, wherever they were kept
    # a grad_value buffer that is not 16-byte aligned holds no records (the call keeps them in the workspace: either the
    # lean size happens to cover that layout, or the call is rejected before anything is launched); the full size always works
    gv_off = gvb[1:1 + v.numel()].view_as(v)
    for ws_t, nbytes, may_reject in ((lean_ws, lean, True), (full_ws, full, False)):
        gv_off.fill_(float("nan"))
        rc = lib.msda_bwd_f32(*args, gv_off.data_ptr(), gl.data_ptr(), ga.data_ptr(), *dims, ws_t.data_ptr(), nbytes, st)
        assert rc == 0 or (may_reject and rc == -1)
        torch.cuda.synchronize()
        if rc == 0:
            np.testing.assert_allclose(gv_off.cpu().numpy(), r_gv, **BWD_TOL[torch.float32])


def test_c_abi_rejects_bad_arguments_without_launching():
    from msda_triton_amd import _lib
    lib = _lib.load()
    x = torch.zeros(64, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    p = x.data_ptr()
    assert lib.msda_fwd_f32(p, p, p, p, p, 1, 4, 1, 4, 1, 1, 1, 7, 0, 0, st) == -1          # unknown padding mode
    assert lib.msda_fwd_f32(p, p, p, p, p, 1, 33, 1, 4, 1, 33, 1, 0, 0, 0, st) == -2        # too many levels
    assert lib.msda_fwd_f32(p, p, p, p, p, 1, 1 << 28, 8, 64, 1, 1, 1, 0, 0, 0, st) == -3    # plane offsets overflow
    assert lib.msda_fwd_f32(p + 2, p, p, p, p, 1, 4, 1, 4, 1, 1, 1, 0, 0, 0, st) == -4      # misaligned
    assert lib.msda_fwd_f32(None, p, p, p, p, 1, 4, 1, 4, 1, 1, 1, 0, 0, 0, st) == -1    # null buffer
    assert lib.msda_fwd_f32(p, p, p, p, p, 1, 4, 1, 4, 1, 1, 1, 0, 0, 8, st) == -1       # value_row_stride below H*D*sizeof
    assert lib.msda_fwd_f32(p, p, p, p, p, 1, 4, 1, 4, 1, 1, 1, 0, 0, 18, st) == -1      # ... not a multiple of the element size
    assert lib.msda_bwd_workspace_bytes(4, 5440, 8, 32, 10000, 4, 4, 4, 4, 0, 0) > 0


def test_torch_compile_fullgraph_uses_registered_custom_op():
    """§8f-3: the operator survives torch.compile(fullgraph=True) as an opaque custom op, forward and backward."""
    ops = _ops()
    import msda_triton_amd.compile_op  # noqa: F401  (registers torch.ops.msda_amd.*)
    c = rand_case(np.random.default_rng(41), 2, 33, 4, 32, [(8, 6), (4, 3)], 3)
    s = torch.from_numpy(c["shapes"]).to(DEV)
    g = torch.from_numpy(c["grad_out"]).to(DEV)

    def fn(v, l, a):
        return ops.multiscale_deformable_attention(v * 1.0, s, l, a, "zeros", False) * 2.0

    outs = []
    for compiled in (False, True):
        v, l, a = (torch.from_numpy(c[k]).to(DEV).requires_grad_(True) for k in ("value", "loc", "attn"))
        f = torch.compile(fn, fullgraph=True, backend="aot_eager") if compiled else fn
        o = f(v, l, a)
        o.backward(g)
        outs.append((o.detach(), v.grad, l.grad, a.grad))
    for x, y in zip(*outs):
        torch.testing.assert_close(x, y, atol=1e-5, rtol=1e-5)
    torch.library.opcheck(torch.ops.msda_amd.forward.default,
                          (torch.from_numpy(c["value"]).to(DEV), s, torch.from_numpy(c["loc"]).to(DEV),
                           torch.from_numpy(c["attn"]).to(DEV), True, False),
                          test_utils=("test_schema", "test_faketensor"))


# ------------------------------------------------------------------------------------------
# §8f-1: module prologue (softmax + offsets -> sampling points) fused into the forward kernel
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("coords", [2, 4])
@pytest.mark.parametrize("name", ["d32_vec_g8", "d5_scalar", "d8_vec_g4", "d64_vec_g16"])
@pytest.mark.parametrize("pm,ac", [("zeros", False), ("border", True)], ids=["zeros_0", "border_1"])
def test_fused_module_core_matches_unfused(coords, name, pm, ac):
    """fused kernel == operator(prologue in PyTorch), forward and every gradient (non-square levels: Q6 order)."""
    from msda_triton_amd.functional import fused_module_core, module_sampling_inputs
    ops = _ops()
    B, Q, H, D, levels, P = SHAPE_MATRIX[name]
    L = len(levels)
    g = torch.Generator(device="cpu").manual_seed(zlib.crc32(name.encode()) + coords)
    value = torch.randn(B, sum(h * w for h, w in levels), H, D, generator=g)
    proj = torch.randn(B, Q, H, L, P, 3, generator=g) * 1.5
    ref = torch.rand(B, Q, coords, generator=g)
    gout = torch.rand(B, Q, H, D, generator=g)
    s = torch.tensor(levels, device=DEV)
    res = []
    for fused in (True, False):
        v, pr, rf = (t.clone().to(DEV).requires_grad_(True) for t in (value, proj, ref))
        if fused:
            out = fused_module_core(v, s, pr, rf, pm, ac)
        else:
            pts, att = module_sampling_inputs(pr, s, rf)
            out = ops.multiscale_deformable_attention(v, s, pts, att, pm, ac)
        out.backward(gout.to(DEV))
        res.append((out.detach(), v.grad, pr.grad, rf.grad))
    torch.testing.assert_close(res[0][0], res[1][0], atol=2e-5, rtol=1e-4)
    for a, b in zip(res[0][1:], res[1][1:]):
        torch.testing.assert_close(a, b, atol=1e-3, rtol=1e-3)


def test_fused_module_core_is_the_module_path_and_handles_large_lp():
    """The nn.Module uses the fused kernel on the GPU (same numbers as its host path); L*P too large for one
    LDS pass falls back to the unfused GPU route without error."""
    from msda_triton_amd import _lib
    from msda_triton_amd.functional import KernelTimer, fused_module_core, module_sampling_inputs
    ops = _ops()
    torch.manual_seed(3)
    m = ops.MultiscaleDeformableAttention(64, 64, 3, 4, 4, "zeros", False)
    shapes = [(7, 5), (4, 3), (2, 2)]
    I = sum(h * w for h, w in shapes)  # noqa: E741
    img, q, ref = torch.randn(2, I, 64), torch.randn(2, 50, 64), torch.rand(2, 50, 4)
    out_cpu = m(img, torch.tensor(shapes), q, ref)
    m = m.to(DEV)
    with KernelTimer() as kt:
        out_gpu = m(img.to(DEV), torch.tensor(shapes, device=DEV), q.to(DEV), ref.to(DEV))
        torch.cuda.synchronize()
    assert "msda_fwd_fused" in kt.summary()
    torch.testing.assert_close(out_gpu.cpu(), out_cpu, atol=1e-4, rtol=1e-3)
    # L*P = 2 * 640 = 1280 samples per unit do not fit one pass
    levels = [(3, 3), (2, 2)]
    v = torch.randn(1, 13, 1, 8, device=DEV)
    pr = torch.randn(1, 3, 1, 2, 640, 3, device=DEV)
    rf = torch.rand(1, 3, 2, device=DEV)
    s = torch.tensor(levels, device=DEV)
    got = fused_module_core(v, s, pr, rf, "border", True)
    pts, att = module_sampling_inputs(pr, s, rf)
    want = ops.multiscale_deformable_attention(v, s, pts, att, "border", True)
    torch.testing.assert_close(got, want, atol=2e-5, rtol=1e-4)
    assert _lib.load().msda_abi_version() == _lib.ABI_VERSION


@pytest.mark.parametrize("coords", [2, 4])
@pytest.mark.parametrize("pm,ac", MODES, ids=[mode_key(*m) for m in MODES])
def test_fused_backward_fp64_matches_autograd_through_the_prologue(coords, pm, ac):
    """msda_bwd_fused_f64 (softmax / offset chain rule inside the kernel, per-head partials of the reference
    points' gradient) against PyTorch autograd through module_sampling_inputs + the plain operator, at fp64
    tolerance; the fused kernels are the ones that ran."""
    from msda_triton_amd.functional import KernelTimer, fused_module_core, module_sampling_inputs
    ops = _ops()
    levels = [(7, 5), (4, 6), (2, 3)]           # non-square: the (h, w) division order matters
    B, Q, H, D, L, P = 2, 37, 3, 16, len(levels), 4
    g = torch.Generator(device="cpu").manual_seed(1234 + coords)
    value = torch.randn(B, sum(h * w for h, w in levels), H, D, generator=g, dtype=torch.float64)
    proj = torch.randn(B, Q, H, L, P, 3, generator=g, dtype=torch.float64) * 1.5
    ref = torch.rand(B, Q, coords, generator=g, dtype=torch.float64)
    gout = torch.rand(B, Q, H, D, generator=g, dtype=torch.float64)
    s = torch.tensor(levels, device=DEV)
    res = []
    for fused in (True, False):
        v, pr, rf = (t.clone().to(DEV).requires_grad_(True) for t in (value, proj, ref))
        with KernelTimer() as kt:
            if fused:
                out = fused_module_core(v, s, pr, rf, pm, ac)
            else:
                pts, att = module_sampling_inputs(pr, s, rf)
                out = ops.multiscale_deformable_attention(v, s, pts, att, pm, ac)
            out.backward(gout.to(DEV))
            torch.cuda.synchronize()
        if fused:
            assert set(kt.summary()) == {"msda_fwd_fused", "msda_bwd_fused"}, kt.summary()
        res.append((out.detach(), v.grad, pr.grad, rf.grad))
    for name, a, b in zip(("out", "grad_value", "grad_proj", "grad_ref"), res[0], res[1]):
        torch.testing.assert_close(a, b, atol=1e-9, rtol=1e-8, msg=lambda m, n=name: f"{n}: {m}")


def test_fused_backward_partial_needs_and_large_lp_fallback():
    """value without grad: the fused backward skips the grad_value passes; L*P too large for one LDS pass: the
    module core still differentiates (PyTorch prologue around the plain operator)."""
    from msda_triton_amd.functional import KernelTimer, fused_module_core, module_sampling_inputs
    ops = _ops()
    torch.manual_seed(11)
    levels = [(5, 4), (2, 2)]
    s = torch.tensor(levels, device=DEV)
    v = torch.randn(2, 24, 2, 8, device=DEV)
    pr = torch.randn(2, 9, 2, 2, 3, 3, device=DEV, requires_grad=True)
    rf = torch.rand(2, 9, 4, device=DEV, requires_grad=True)
    fused_module_core(v, s, pr, rf, "zeros", False).sum().backward()
    pr2, rf2 = pr.detach().clone().requires_grad_(True), rf.detach().clone().requires_grad_(True)
    pts, att = module_sampling_inputs(pr2, s, rf2)
    ops.multiscale_deformable_attention(v, s, pts, att, "zeros", False).sum().backward()
    torch.testing.assert_close(pr.grad, pr2.grad, atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(rf.grad, rf2.grad, atol=1e-4, rtol=1e-3)
    # 1280 samples per unit
    v = torch.randn(1, 13, 1, 8, device=DEV, requires_grad=True)
    pr = torch.randn(1, 3, 1, 2, 640, 3, device=DEV, requires_grad=True)
    rf = torch.rand(1, 3, 2, device=DEV, requires_grad=True)
    s = torch.tensor([(3, 3), (2, 2)], device=DEV)
    with KernelTimer() as kt:
        fused_module_core(v, s, pr, rf, "border", True).sum().backward()
        torch.cuda.synchronize()
    assert "msda_bwd_fused" not in kt.summary() and "msda_bwd_sample" in kt.summary()
    assert pr.grad is not None and rf.grad is not None and v.grad is not None
    assert torch.isfinite(pr.grad).all()


@pytest.mark.parametrize("value_path", [2, 3], ids=["sorted_gather", "single_launch"])
@pytest.mark.parametrize("seed", list(range(24)))
def test_random_shapes_against_oracle(oracle, seed, value_path):
    """Differential test over random shapes / modes / coordinate ranges (fixed seeds), with the sorted-gather
    grad_value pipeline forced so that its cell / work-item bookkeeping sees odd pyramids, empty cells, cells on
    the border and out-of-range samples."""
    from msda_triton_amd import _lib
    rng = np.random.default_rng(9000 + seed)
    B, Q, H = int(rng.integers(1, 4)), int(rng.integers(1, 160)), int(rng.integers(1, 5))
    D = int(rng.choice([1, 3, 8, 16, 32, 40, 64]))
    L, P = int(rng.integers(1, 5)), int(rng.integers(1, 6))
    levels = [(int(rng.integers(1, 13)), int(rng.integers(1, 13))) for _ in range(L)]
    lo, hi = [(-0.4, 1.4), (0.0, 1.0), (0.3, 0.6), (-3.0, 4.0)][seed % 4]
    pm, ac = MODES[int(rng.integers(0, len(MODES)))]
    f64 = seed % 3 == 0
    c = rand_case(rng, B, Q, H, D, levels, P, lo=lo, hi=hi, dtype=np.float64 if f64 else np.float32)
    td = torch.float64 if f64 else torch.float32
    try:
        _lib.set_option("value_path", value_path)
        check_against_oracle(oracle, c, pm, ac, FWD_TOL[td], BWD_TOL[td])
    finally:
        _lib.set_option("value_path", 0)


@pytest.mark.parametrize("small_ns", [1, 2, 3, 4, 7])
def test_single_launch_kernel_with_several_workgroups_per_level(oracle, small_ns):
    """The single-launch grad_value kernel with its (plane, level) work split over small_ns workgroups (chosen by the
    launcher when few planes would leave CUs idle; forced here): every 2 x 2-pixel block is served by exactly one of
    them whatever order each workgroup walks its blocks in — levels with fewer blocks than workgroups, odd sizes, a
    hot cell that makes one level split its blocks over lane groups."""
    from msda_triton_amd import _lib
    rng = np.random.default_rng(515 + small_ns)
    levels = [(13, 9), (6, 7), (3, 2), (1, 1)]
    c = rand_case(rng, 2, 97, 2, 32, levels, 3, lo=-0.1, hi=1.1)
    c["loc"][:, ::3, :, 0, :, :] = (0.41 + rng.normal(0, 0.004, size=c["loc"][:, ::3, :, 0, :, :].shape)).astype(np.float32)
    try:
        _lib.set_option("value_path", 3)
        _lib.set_option("small_ns", small_ns)
        for pm, ac in (("zeros", False), ("border", True)):
            check_against_oracle(oracle, c, pm, ac, FWD_TOL[torch.float32], BWD_TOL[torch.float32])
    finally:
        _lib.set_option("small_ns", 0)
        _lib.set_option("value_path", 0)


def test_level_shapes_hint_opens_the_single_launch_kernel_on_image_sized_pyramids(oracle):
    """`level_shapes=` (the pyramid's sizes as host numbers, e.g. Hugging Face's spatial_shapes_list) promises the
    library a bound on the largest level's bilinear cells (the max_level_cells argument): a decoder-sized call over a pyramid
    whose 2 I + 2 L worst case does not fit the single-launch grad_value kernel's LDS then takes that kernel (no
    workspace) instead of the sorted pipeline — same results, both routes (C++ binding and ctypes); the bound is an
    argument of the call and of the size query, nothing outlives it."""
    from msda_triton_amd import _lib
    from msda_triton_amd.functional import level_cells_of, msda_hip_bwd
    ops = _ops()
    levels = [(100, 134), (50, 67), (25, 34), (13, 17)]  # an 800 x 1066 image; I = 17 821
    B, Q, H, D, P = 1, 60, 2, 32, 4
    rng = np.random.default_rng(77)
    c = rand_case(rng, B, Q, H, D, levels, P, lo=-0.05, hi=1.05)
    lib = _lib.load()
    dims = (B, sum(h * w for h, w in levels), H, D, Q, len(levels), P, 4, 4)
    cells = level_cells_of(levels)
    assert cells == 101 * 135
    assert lib.msda_bwd_workspace_bytes(*dims, 0, 0) > 0        # worst case from I: the sorted pipeline
    assert lib.msda_bwd_workspace_bytes(*dims, cells, 0) == 0   # with the bound: the single-launch kernel
    assert lib.msda_bwd_workspace_bytes(*dims, 0, 0) > 0        # (an argument: nothing is remembered)
    r_gv, r_gl, r_ga = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], "zeros", False)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in c.items()}
    # launcher API
    gv, gl, ga = msda_hip_bwd(t["grad_out"], t["value"], t["shapes"], t["loc"], t["attn"], "zeros", False,
                              level_cells=cells)
    np.testing.assert_allclose(gv.cpu().numpy(), r_gv, **BWD_TOL[torch.float32])
    np.testing.assert_allclose(ga.cpu().numpy(), r_ga, **BWD_TOL[torch.float32])
    # public API (the C++ autograd node when the binding is built) and the Python Function over ctypes
    from msda_triton_amd.functional import _HipMultiscaleDeformableAttentionFunction as PyFn
    for route in ("public", "py"):
        v, l, a = (t[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
        if route == "public":
            out = ops.multiscale_deformable_attention(v, t["shapes"], l, a, "zeros", False, level_shapes=levels)
        else:
            out = PyFn.apply(v, t["shapes"], l, a, "zeros", False, cells)
        out.backward(t["grad_out"])
        np.testing.assert_allclose(v.grad.cpu().numpy(), r_gv, err_msg=route, **BWD_TOL[torch.float32])
        np.testing.assert_allclose(a.grad.cpu().numpy(), r_ga, err_msg=route, **BWD_TOL[torch.float32])


def test_level_shapes_reach_the_backward_of_the_compiled_op(oracle):
    """torch.compile: `level_shapes` becomes the custom ops' `level_cells` constant, so the compiled
    operator's backward takes the single-launch grad_value kernel on an image-sized pyramid like the eager one — read
    off the library's own kernel records."""
    from msda_triton_amd import _lib
    ops = _ops()
    import msda_triton_amd.compile_op  # noqa: F401
    levels = [(100, 134), (50, 67), (25, 34), (13, 17)]
    c = rand_case(np.random.default_rng(78), 1, 60, 2, 32, levels, 4, lo=-0.05, hi=1.05)
    r_gv, _, r_ga = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], "zeros", False)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in c.items()}

    def with_hint(v, l, a):
        return ops.multiscale_deformable_attention(v, t["shapes"], l, a, "zeros", False, level_shapes=levels)

    def without(v, l, a):
        return ops.multiscale_deformable_attention(v, t["shapes"], l, a, "zeros", False)

    for fn, kernel in ((with_hint, "msda_value_small_kernel"), (without, "msda_value_gather_kernel")):
        f = torch.compile(fn, fullgraph=True, backend="aot_eager")
        v, l, a = (t[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
        out = f(v, l, a)
        _lib.set_option("profile", 1)
        try:
            _lib.profile_read()
            out.backward(t["grad_out"])
            torch.cuda.synchronize()
            ran = _lib.profile_read()
        finally:
            _lib.set_option("profile", 0)
        assert kernel in ran, (fn.__name__, sorted(ran))
        np.testing.assert_allclose(v.grad.cpu().numpy(), r_gv, **BWD_TOL[torch.float32])
        np.testing.assert_allclose(a.grad.cpu().numpy(), r_ga, **BWD_TOL[torch.float32])


def test_level_shapes_hint_through_the_fused_module_core_and_the_module():
    """The same promise through fused_module_core (C++ node or Python Function) and MultiscaleDeformableAttention's
    forward(level_shapes=...): same outputs and gradients as without it (the route differs, the numbers may in the
    last bits: different summation order)."""
    from msda_triton_amd.functional import fused_module_core
    ops = _ops()
    levels = [(100, 134), (50, 67)]  # 2 I + 2 L = 33 504 cells do not fit the kernel's LDS, the promised 101 * 135 do
    B, Q, H, D, P = 2, 70, 2, 32, 4
    g = torch.Generator(device="cpu").manual_seed(42)
    I = sum(h * w for h, w in levels)
    value = torch.randn(B, I, H, D, generator=g)
    proj = torch.randn(B, Q, H, len(levels), P, 3, generator=g)
    ref = torch.rand(B, Q, 2, generator=g)
    gout = torch.rand(B, Q, H, D, generator=g)
    shapes = torch.tensor(levels, device=DEV)
    res = []
    for ls in (None, levels):
        v, pr, rf = (t.clone().to(DEV).requires_grad_(True) for t in (value, proj, ref))
        out = fused_module_core(v, shapes, pr, rf, "zeros", False, level_shapes=ls)
        out.backward(gout.to(DEV))
        res.append((out.detach(), v.grad, pr.grad, rf.grad))
    for name, a, b in zip(("out", "grad_value", "grad_proj", "grad_ref"), res[0], res[1]):
        torch.testing.assert_close(a, b, atol=2e-5, rtol=1e-4, msg=lambda m, n=name: f"{n}: {m}")
    torch.manual_seed(0)
    mod = ops.MultiscaleDeformableAttention(emb_dim=32, hidden_dim=64, num_levels=len(levels), num_heads=2, num_points=4,
                                            padding_mode="border", align_corners=True).to(DEV)
    img = torch.randn(B, I, 32, device=DEV)
    qs = torch.randn(B, Q, 32, device=DEV)
    refs = torch.rand(B, Q, 4, device=DEV)
    outs = []
    for ls in (None, levels):
        mod.zero_grad()
        o = mod(img, shapes, qs, refs, level_shapes=ls)
        o.square().sum().backward()
        outs.append((o.detach(), mod.img_input_proj.weight.grad.clone()))
    torch.testing.assert_close(outs[0][0], outs[1][0], atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(outs[0][1], outs[1][1], atol=2e-3, rtol=2e-3)


def test_broken_level_shapes_promise_gives_nan_rows_not_wrong_ones():
    """A level larger than promised cannot be served by the table sized on the promise and cannot be reported from the
    kernel: its grad_value rows come back NaN, the other levels' rows are right."""
    from msda_triton_amd.functional import msda_hip_bwd
    levels = [(40, 50), (9, 7)]
    B, Q, H, D, P = 1, 30, 2, 16, 2
    rng = np.random.default_rng(5)
    c = rand_case(rng, B, Q, H, D, levels, P, lo=0.0, hi=1.0)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in c.items()}
    good, _, _ = msda_hip_bwd(t["grad_out"], t["value"], t["shapes"], t["loc"], t["attn"], "border", True)
    bad, gl, ga = msda_hip_bwd(t["grad_out"], t["value"], t["shapes"], t["loc"], t["attn"], "border", True,
                               level_cells=20 * 20)  # level 0 has 41 * 51 cells
    n0 = 40 * 50
    assert torch.isnan(bad[:, :n0]).all()
    torch.testing.assert_close(bad[:, n0:], good[:, n0:], atol=1e-5, rtol=1e-5)
    assert torch.isfinite(gl).all() and torch.isfinite(ga).all()


@pytest.mark.parametrize("td", [torch.float32, torch.float64, torch.bfloat16], ids=["f32", "f64", "bf16"])
def test_sorted_grad_value_in_query_rounds(oracle, td):
    """Very large Q is served in rounds over the queries (a plane's grad_out rows stay in L2; running sums in the
    accumulate type between rounds).  Forced here with tiny rounds: first / middle / last round paths, ragged last
    round, every storage class of the running sums."""
    from msda_triton_amd import _lib
    rng = np.random.default_rng(4242)
    npdt = np.float64 if td == torch.float64 else np.float32
    c = rand_case(rng, 2, 53, 3, 32, [(9, 7), (4, 5), (2, 2)], 3, lo=-0.2, hi=1.2, dtype=npdt)
    if td == torch.bfloat16:
        for k in ("value", "loc", "attn", "grad_out"):
            c[k] = torch.from_numpy(c[k]).to(td).float().numpy()
    try:
        _lib.set_option("value_path", 2)
        for q_round in (7, 20, 52):
            _lib.set_option("q_round", q_round)
            for pm, ac in (("zeros", False), ("border", True)):
                _, gv, _, _ = run_hip(c["value"], c["shapes"], c["loc"], c["attn"], c["grad_out"], pm, ac, dtype=td)
                r_gv, _, _ = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], pm, ac)
                tol = dict(atol=5e-2, rtol=2e-2) if td == torch.bfloat16 else BWD_TOL[td]
                np.testing.assert_allclose(gv, r_gv, err_msg=f"q_round {q_round} {pm} {ac}", **tol)
    finally:
        _lib.set_option("q_round", 0)
        _lib.set_option("value_path", 0)


def test_sorted_pipeline_is_bitwise_reproducible_by_default():
    """Round 4: the level-major place pass lets its waves take their cursor atomics in turns, so wherever it runs — the
    sorted pipeline on problems with at least as many samples as cell-table entries: c2 @ 5k / 10k, c3, c5 — grad_value
    is bitwise reproducible WITHOUT any option, at no measurable cost.  Repeated calls, other work
    in flight, the other workgroup -> plane mapping."""
    from msda_triton_amd import _lib, synth
    ops = _ops()
    wl = synth.WORKLOADS["c2_q5k"]
    d = synth.make_inputs_torch(wl, DEV, seed=8)
    args = (d["grad_out"], d["value"], d["shapes"], d["loc"], d["attn"])
    try:
        runs = []
        for k in range(4):
            if k == 2:
                _lib.set_option("xcd_map", 0)
                noise = torch.randn(1 << 22, device=DEV).sin_()  # noqa: F841
            runs.append(ops.msda_hip_bwd(*args, wl.padding_mode, wl.align_corners))
            torch.cuda.synchronize()
            _lib.set_option("xcd_map", 1)
        for r in runs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(r, runs[0]))
    finally:
        _lib.set_option("xcd_map", 1)


@pytest.mark.parametrize("case", ["long_lists", "c4", "many_points"])
def test_single_launch_kernel_is_bitwise_reproducible_by_default(case):
    """Round 4: the single-launch grad_value kernel places its LDS records in turns too (msda_value_small.hpp), so
    the small problems are bitwise reproducible with no option either: repeated calls, other work in flight, another
    workgroup -> plane mapping, 1 / 2 / 4 workgroups per (plane, level) each against itself."""
    from msda_triton_amd import _lib, synth
    ops = _ops()
    if case == "c4":
        wl = synth.WORKLOADS["c4_gdino_dec"]
        d = synth.make_inputs_torch(wl, DEV, seed=3)
        args, pm, ac = (d["grad_out"], d["value"], d["shapes"], d["loc"], d["attn"]), wl.padding_mode, wl.align_corners
    else:
        rng = np.random.default_rng(516)
        if case == "long_lists":   # a 1 x 1 level takes a quarter of all samples; more than four samples per thread
            c = rand_case(rng, 2, 700, 4, 32, [(9, 7), (4, 4), (1, 1)], 8, lo=-0.2, hi=1.2)
        else:                      # P does not divide the workgroup: whole waves have nothing to place
            c = rand_case(rng, 1, 90, 2, 16, [(6, 5), (3, 3)], 24, lo=-0.2, hi=1.2)
        args, pm, ac = tuple(torch.from_numpy(c[k]).to(DEV) for k in ("grad_out", "value", "shapes", "loc", "attn")), "zeros", False
    try:
        _lib.set_option("value_path", 3)
        for ns in (0, 1, 2, 4):
            _lib.set_option("small_ns", ns)
            runs = []
            for k in range(4):
                if k == 2:
                    _lib.set_option("xcd_map", 0)
                    noise = torch.randn(1 << 22, device=DEV).sin_()  # noqa: F841
                runs.append(ops.msda_hip_bwd(*args, pm, ac)[0])
                torch.cuda.synchronize()
                _lib.set_option("xcd_map", 1)
            for r in runs[1:]:
                assert torch.equal(r, runs[0]), (case, ns)
    finally:
        _lib.set_option("small_ns", 0)
        _lib.set_option("value_path", 0)
        _lib.set_option("xcd_map", 1)


@pytest.mark.parametrize("place_path", [0, 3], ids=["auto", "256_threads"])
def test_level_larger_than_the_place_pass_lds_table(oracle, place_path):
    """The level-major place pass keeps ONE level's cell cursors in LDS, capped at 19 456 cells so that two workgroups
    share a CU; a larger level (here 150 x 160 = 24 311 cells, next to a small one) is walked in several trips over
    its samples.  Sorted pipeline, both workgroup sizes, zeros and border."""
    from msda_triton_amd import _lib
    c = rand_case(np.random.default_rng(2024), 1, 2600, 2, 32, [(150, 160), (7, 9)], 4, lo=-0.03, hi=1.03)
    try:
        _lib.set_option("value_path", 2)
        _lib.set_option("place_path", place_path)
        for pm, ac in (("zeros", False), ("border", True)):
            check_against_oracle(oracle, c, pm, ac, FWD_TOL[torch.float32], BWD_TOL[torch.float32])
    finally:
        _lib.set_option("value_path", 0)
        _lib.set_option("place_path", 0)


@pytest.mark.parametrize("value_path", [2, 3], ids=["sorted_gather", "single_launch"])
def test_more_points_per_level_than_a_workgroup_has_threads(oracle, value_path):
    """P = 1100 > 1024: the single-launch kernel walks a query's points in strides of the workgroup (its ordered placing
    takes one turn per stride), the sorted pipeline falls back to the plane-major place pass (the level-major one maps a
    thread to one point)."""
    from msda_triton_amd import _lib
    c = rand_case(np.random.default_rng(1100), 1, 3, 2, 8, [(5, 4), (2, 3)], 1100, lo=-0.2, hi=1.2)
    try:
        _lib.set_option("value_path", value_path)
        for pm, ac in (("zeros", False), ("border", True)):
            check_against_oracle(oracle, c, pm, ac, FWD_TOL[torch.float32], BWD_TOL[torch.float32])
    finally:
        _lib.set_option("value_path", 0)


def _boundary_coordinates(n):
    """float32 coordinates next to the cell boundaries of an n-pixel axis for which x * n - 0.5 lands in different
    cells when it is computed with one rounding (fused multiply-add) and with two"""
    x = ((np.arange(n) + 0.5) / n).astype(np.float32)
    for _ in range(4):
        x = np.nextafter(x, np.float32(-2))
    out = []
    for _ in range(9):
        p = x.astype(np.float64) * n
        two = (p.astype(np.float32) - np.float32(0.5)).astype(np.float32)
        one = (p - 0.5).astype(np.float32)
        out.extend(x[np.floor(two) != np.floor(one)].tolist())
        x = np.nextafter(x, np.float32(2))
    return np.array(out, dtype=np.float32)


@pytest.mark.parametrize("value_path", [2, 3], ids=["sorted_gather", "single_launch"])
@pytest.mark.parametrize("pm", ["zeros", "border"])
def test_samples_within_an_ulp_of_a_cell_boundary(oracle, value_path, pm):
    """Found by tools/fuzz_parity.py (seed 1000117, round 4): the count pass, the place pass (twice: before and after its
    turn) and the single-launch kernel's walks each inline the sample -> cell arithmetic, and the compiler fused
    `x * W - 0.5` in one copy and not in the other; a coordinate within an ulp of a cell boundary was then placed in one
    cell's list carrying the other cell's word.  Here most samples sit on exactly such coordinates (align_corners=False:
    the expression in question)."""
    from msda_triton_amd import _lib
    levels = [(25, 25), (40, 100)]
    assert len(_boundary_coordinates(25)) >= 3 and len(_boundary_coordinates(100)) >= 3
    rng = np.random.default_rng(1000117)
    c = rand_case(rng, 2, 900 if value_path == 3 else 2500, 2, 32, levels, 4, lo=-0.02, hi=1.02)
    for l, (h, w) in enumerate(levels):
        for axis, n in ((0, w), (1, h)):
            adv = _boundary_coordinates(n)
            m = rng.uniform(size=c["loc"].shape[:3] + (c["loc"].shape[4],)) < 0.7
            c["loc"][:, :, :, l, :, axis][m] = rng.choice(adv, size=int(m.sum()))
    try:
        _lib.set_option("value_path", value_path)
        check_against_oracle(oracle, c, pm, False, FWD_TOL[torch.float32], BWD_TOL[torch.float32])
    finally:
        _lib.set_option("value_path", 0)


def test_grad_value_is_bitwise_reproducible_everywhere(oracle):
    """No option needed (round 4): a problem of the single-launch kernel with long cell lists (a
    1 x 1 level takes a quarter of all samples), a c2-sized problem of the sorted pipeline, and a decoder call over an
    image-sized pyramid (more cells than samples: the shape the plane-major place pass used to take).  Across repeated
    calls, with other work interleaved on the device, and against a run under a different workgroup -> plane mapping;
    the result still matches the oracle.  Under msda_set_option("strict", 1) these shapes run all the same (they ARE
    reproducible); a shape that is not (P > 1024 points per level) is refused there — test_strict_refuses_…"""
    from msda_triton_amd import _lib, synth
    ops = _ops()
    rng = np.random.default_rng(515)
    c = rand_case(rng, 2, 700, 4, 32, [(9, 7), (4, 4), (1, 1)], 4, lo=-0.2, hi=1.2)
    small = tuple(torch.from_numpy(c[k]).to(DEV) for k in ("grad_out", "value", "shapes", "loc", "attn"))
    wl = synth.WORKLOADS["c2_q5k"]
    d = synth.make_inputs_torch(wl, DEV, seed=8)
    big = (d["grad_out"], d["value"], d["shapes"], d["loc"], d["attn"])
    cs = rand_case(rng, 2, 300, 2, 32, [(100, 134), (50, 67), (25, 34), (13, 17)], 4, lo=-0.05, hi=1.05)
    sparse = tuple(torch.from_numpy(cs[k]).to(DEV) for k in ("grad_out", "value", "shapes", "loc", "attn"))
    lib = _lib.load()
    assert lib.msda_bwd_workspace_bytes(2, 17821, 2, 32, 300, 4, 4, 4, 4, 0, 0) > 0  # (the sorted pipeline)
    try:
        for det in (0, 1):
            _lib.set_option("strict", det)
            assert lib.msda_get_option(b"strict") == det
            for args, pm, ac in ((small, "zeros", False), (small, "border", True), (big, wl.padding_mode, wl.align_corners),
                                 (sparse, "zeros", False)):
                runs = []
                for k in range(4):
                    if k == 2:  # different scheduling: other kernels in flight, plain block mapping
                        _lib.set_option("xcd_map", 0)
                        noise = torch.randn(1 << 22, device=DEV).sin_()  # noqa: F841
                    runs.append(ops.msda_hip_bwd(*args, pm, ac))
                    torch.cuda.synchronize()
                    _lib.set_option("xcd_map", 1)
                for r in runs[1:]:
                    assert all(torch.equal(a, b) for a, b in zip(r, runs[0]))
        gv = ops.msda_hip_bwd(*small, "zeros", False)[0]
        r_gv, _, _ = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], "zeros", False)
        np.testing.assert_allclose(gv.cpu().numpy(), r_gv, **BWD_TOL[torch.float32])
        gv = ops.msda_hip_bwd(*sparse, "zeros", False)[0]
        r_gv, _, _ = oracle.backward(cs["grad_out"], cs["value"], cs["shapes"], cs["loc"], cs["attn"], "zeros", False)
        np.testing.assert_allclose(gv.cpu().numpy(), r_gv, **BWD_TOL[torch.float32])
    finally:
        _lib.set_option("strict", 0)
        _lib.set_option("xcd_map", 1)


def test_strict_refuses_a_backward_that_would_not_be_reproducible():
    """P > 1024 points per level takes the plane-major place pass, whose record order follows free-running atomics: with
    msda_set_option("strict", 1) the call is refused (MSDA_ERR_UNSUPPORTED -> ValueError) instead of silently giving a
    grad_value whose last bit may change from run to run; without it the call runs.  The same for a forced
    place_path = 1."""
    from msda_triton_amd import _lib
    ops = _ops()
    rng = np.random.default_rng(5)
    c = rand_case(rng, 1, 12, 2, 8, [(6, 5)], 1100)
    many = tuple(torch.from_numpy(c[k]).to(DEV) for k in ("grad_out", "value", "shapes", "loc", "attn"))
    c2 = rand_case(rng, 1, 3000, 2, 32, [(12, 10), (6, 5)], 4)
    usual = tuple(torch.from_numpy(c2[k]).to(DEV) for k in ("grad_out", "value", "shapes", "loc", "attn"))
    try:
        _lib.set_option("value_path", 2)  # (the sorted pipeline for both)
        ops.msda_hip_bwd(*many, "zeros", False)  # runs
        _lib.set_option("strict", 1)
        with pytest.raises(ValueError, match="strict"):
            ops.msda_hip_bwd(*many, "zeros", False)
        ops.msda_hip_bwd(*usual, "zeros", False)  # a reproducible shape is not affected
        _lib.set_option("place_path", 1)
        with pytest.raises(ValueError, match="strict"):
            ops.msda_hip_bwd(*usual, "zeros", False)
    finally:
        _lib.set_option("place_path", 0)
        _lib.set_option("strict", 0)
        _lib.set_option("value_path", 0)
    torch.cuda.synchronize()


def test_make_graphed_callables_replays_forward_and_backward():
    """torch.cuda.make_graphed_callables captures the operator's forward and backward (no host sync, no allocation
    outside PyTorch's capture pool) — the way to take the launch overhead off small problems."""
    from msda_triton_amd import synth
    ops = _ops()
    wl = synth.WORKLOADS["c1_readme"]
    d = synth.make_inputs_torch(wl, DEV, seed=4)
    v, l, a = (d[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
    s, g = d["shapes"], d["grad_out"]

    def fn(v_, l_, a_):
        return ops.multiscale_deformable_attention(v_, s, l_, a_, wl.padding_mode, wl.align_corners)

    def eager():  # (no reference to the eager autograd graph may survive into the capture: PyTorch would crash)
        out = fn(v, l, a)
        out.backward(g)
        return out.detach().clone(), v.grad.clone(), l.grad.clone(), a.grad.clone()

    want = eager()
    v.grad = l.grad = a.grad = None
    graphed = torch.cuda.make_graphed_callables(fn, (v, l, a))
    for _ in range(2):  # replay twice: static buffers are reused
        v.grad = l.grad = a.grad = None
        out = graphed(v, l, a)
        out.backward(g)
        torch.cuda.synchronize()
        torch.testing.assert_close(out.detach(), want[0], atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(l.grad, want[2], atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(a.grad, want[3], atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(v.grad, want[1], atol=1e-4, rtol=1e-4)


def test_cpp_binding_and_ctypes_routes_agree():
    """The public entry point goes through the C++ autograd glue (msda_torch_ext) when it is built; the Python
    autograd Function over ctypes is the other route to the same C ABI.  Same kernels, same numbers."""
    from msda_triton_amd import _ext, synth
    from msda_triton_amd.functional import _HipMultiscaleDeformableAttentionFunction as PyFn
    ops = _ops()
    ext = _ext.load()
    if ext is None:
        pytest.skip("msda_torch_ext is not built (optional)")
    wl = synth.WORKLOADS["c1_readme"]
    d = synth.make_inputs_torch(wl, DEV, seed=9, loc_lo=-0.1, loc_hi=1.1)
    res = []
    for route in ("ext", "py"):
        v, l, a = (d[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
        if route == "ext":
            out = ops.multiscale_deformable_attention(v, d["shapes"], l, a, "zeros", False)
            assert "MSDAFunction" in out.grad_fn.name()  # the C++ autograd node, not the Python Function
        else:
            out = PyFn.apply(v, d["shapes"], l, a, "zeros", False)
        out.backward(d["grad_out"])
        res.append((out.detach(), v.grad, l.grad, a.grad))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])
    torch.testing.assert_close(res[0][1], res[1][1], atol=1e-5, rtol=1e-5)
    # partial needs and errors behave alike
    v = d["value"].clone()
    l = d["loc"].clone().requires_grad_(True)
    out = ops.multiscale_deformable_attention(v, d["shapes"], l, d["attn"], "zeros", False)
    out.sum().backward()
    assert l.grad is not None and torch.isfinite(l.grad).all()
    with pytest.raises(ValueError):
        ops.multiscale_deformable_attention(v, d["shapes"], l[:, :, :, :1], d["attn"], "zeros", False)


def test_torch_compile_module_keeps_the_fused_kernels():
    """A compiled MultiscaleDeformableAttention traces to torch.ops.msda_amd.fused_forward / fused_backward (the
    fused kernels stay in the graph as opaque custom ops) and matches the eager module, gradients included."""
    import msda_triton_amd.compile_op  # noqa: F401
    ops = _ops()
    torch.manual_seed(5)
    m = ops.MultiscaleDeformableAttention(32, 32, 2, 4, 3, "border", True).to(DEV)
    levels = [(6, 5), (3, 4)]
    s = torch.tensor(levels, device=DEV)
    img = torch.randn(2, sum(h * w for h, w in levels), 32, device=DEV)
    q = torch.randn(2, 21, 32, device=DEV)
    ref = torch.rand(2, 21, 2, device=DEV)
    compiled = torch.compile(m, fullgraph=True, backend="aot_eager")
    res = []
    for f in (m, compiled):
        m.zero_grad()
        i_, q_ = img.clone().requires_grad_(True), q.clone().requires_grad_(True)
        out = f(i_, s, q_, ref)
        out.square().sum().backward()
        res.append((out.detach(), i_.grad, q_.grad, m.query_input_proj.weight.grad.clone()))
    for a, b in zip(*res):
        torch.testing.assert_close(a, b, atol=1e-4, rtol=1e-4)
    v = torch.randn(2, 42, 4, 8, device=DEV)
    pr = torch.randn(2, 21, 4, 2, 3, 3, device=DEV)
    torch.library.opcheck(torch.ops.msda_amd.fused_forward.default, (v, s, pr, ref, False, True),
                          test_utils=("test_schema", "test_faketensor"))


def test_seventeen_levels_beyond_the_single_launch_kernel(oracle):
    """L = 17 with Q*P > 4096 (too many samples for the single-launch kernel, more levels than the round-3
    record format's 4 level bits) had no grad_value route.  The cell word now carries 5 level bits."""
    rng = np.random.default_rng(17)
    levels = [(6, 5), (4, 7), (3, 3), (8, 2), (2, 9), (5, 5), (1, 6), (7, 1), (4, 4), (3, 6), (2, 2), (6, 3), (1, 1), (5, 2),
              (2, 7), (3, 4), (9, 9)]
    c = rand_case(rng, 1, 1100, 2, 8, levels, 4)
    assert 1100 * 4 > 4096 and len(levels) == 17
    for pm, ac in MODES:
        check_against_oracle(oracle, c, pm, ac, FWD_TOL[torch.float32], BWD_TOL[torch.float32])




@pytest.mark.parametrize("slots", [1, 6, 40])
def test_block_mapping_may_differ_between_the_kernels_of_one_call(oracle, slots):
    """`linear_slots`: launches with many workgroups per plane take the linear block order, the others the XCD-aware one —
    inside ONE backward the count, place, gather and finish passes may therefore disagree about the mapping, and each grid
    has to travel with the mapping it was built for (a full-size c5 run caught a stale one in round 5)."""
    from msda_triton_amd import _lib
    c = rand_case(np.random.default_rng(100 + slots), 2, 1500, 8, 32, [(16, 16), (8, 8), (4, 4)], 4)
    old = {k: _lib.get_option(k) for k in ("linear_slots", "value_path")}
    _lib.set_option("linear_slots", slots)
    _lib.set_option("value_path", 2)  # the sorted pipeline (five launches)
    try:
        for pm, ac in (MODES[0], MODES[3]):
            check_against_oracle(oracle, c, pm, ac, FWD_TOL[torch.float32], BWD_TOL[torch.float32])
    finally:
        for k, v in old.items():
            _lib.set_option(k, v)
