"""Mixed storage (the value pyramid kept in 16 bits in the kernel's layout): `img` / its gradient in
bfloat16 or float16, sampling points, attention weights, output and the other gradients in float32 — the
``msda_*_f32_vbf16`` / ``msda_*_f32_vf16`` entry points.  Run with ``-m gpu``.

The oracle gets the SAME 16-bit-rounded value pyramid (as float32), so out / grad_loc / grad_attn are compared at the
fp32 tolerances (out 1e-4: nothing but `img`'s storage differs from the fp32 path) and only grad_value carries its
storage rounding (bf16: 8 significant bits, fp16: 11).
"""
import zlib

import numpy as np
import pytest
import torch

from conftest import MODES, kink_mask
from test_gpu_parity import BWD_TOL, DEV, FWD_TOL, SHAPE_MATRIX, rand_case

pytestmark = pytest.mark.gpu

F32 = torch.float32
GV_TOL = {torch.bfloat16: dict(atol=2e-2, rtol=1e-2), torch.float16: dict(atol=3e-3, rtol=2e-3)}


def _ops():
    import msda_triton_amd
    return msda_triton_amd


def _round(value: np.ndarray, vdt: torch.dtype) -> np.ndarray:
    return torch.from_numpy(value).to(vdt).to(F32).numpy()


def run_mixed(c, pm, ac, vdt, fn=None):
    ops = _ops()
    v = torch.from_numpy(c["value"]).to(DEV, vdt).requires_grad_(True)
    l = torch.from_numpy(c["loc"]).to(DEV, F32).requires_grad_(True)
    a = torch.from_numpy(c["attn"]).to(DEV, F32).requires_grad_(True)
    s = torch.from_numpy(c["shapes"]).to(DEV)
    out = (fn or ops.multiscale_deformable_attention)(v, s, l, a, pm, ac)
    assert out.dtype == F32
    out.backward(torch.from_numpy(c["grad_out"]).to(DEV, F32))
    assert v.grad.dtype == vdt and l.grad.dtype == F32 and a.grad.dtype == F32
    return tuple(t.detach().float().cpu().numpy() for t in (out, v.grad, l.grad, a.grad))


def check_mixed(oracle, c, pm, ac, vdt):
    c = dict(c, value=_round(c["value"], vdt))
    out, gv, gl, ga = run_mixed(c, pm, ac, vdt)
    r_out = oracle.forward(c["value"], c["shapes"], c["loc"], c["attn"], pm, ac)
    r_gv, r_gl, r_ga = oracle.backward(c["grad_out"], c["value"], c["shapes"], c["loc"], c["attn"], pm, ac)
    np.testing.assert_allclose(out, r_out, err_msg="out", **FWD_TOL[F32])
    np.testing.assert_allclose(ga, r_ga, err_msg="grad_attn", **BWD_TOL[F32])
    keep = ~kink_mask(c["loc"], c["shapes"], ac)
    np.testing.assert_allclose(np.where(keep, gl, 0), np.where(keep, r_gl, 0), err_msg="grad_loc", **BWD_TOL[F32])
    np.testing.assert_allclose(gv, r_gv, err_msg="grad_value", **GV_TOL[vdt])


@pytest.mark.parametrize("vdt", [torch.bfloat16, torch.float16], ids=["vbf16", "vf16"])
@pytest.mark.parametrize("name", sorted(SHAPE_MATRIX))
def test_mixed_storage_matches_oracle_shape_matrix(oracle, name, vdt):
    B, Q, H, D, levels, P = SHAPE_MATRIX[name]
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    c = rand_case(rng, B, Q, H, D, levels, P)
    for pm, ac in MODES:
        check_mixed(oracle, c, pm, ac, vdt)


@pytest.mark.parametrize("value_path", [2, 3], ids=["sorted", "small"])
def test_mixed_storage_every_grad_value_path(oracle, value_path):
    """Both grad_value paths store the 16-bit rows (sorted pipeline's finish kernel, single-launch kernel)."""
    from msda_triton_amd import _lib
    rng = np.random.default_rng(77 + value_path)
    c = rand_case(rng, 2, 300, 4, 32, [(12, 10), (6, 5), (3, 3)], 4)
    old = _lib.get_option("value_path")
    _lib.set_option("value_path", value_path)
    try:
        for pm, ac in MODES:
            check_mixed(oracle, c, pm, ac, torch.bfloat16)
    finally:
        _lib.set_option("value_path", old)


def test_mixed_storage_out_equals_fp32_path_on_the_rounded_values():
    """Nothing but the storage of `img` differs: the forward is bit-identical to the fp32 kernels fed the rounded
    pyramid, and so are grad_loc / grad_attn (same arithmetic, same order); grad_value is that of the fp32 kernels
    rounded to the storage type."""
    ops = _ops()
    rng = np.random.default_rng(5)
    c = rand_case(rng, 2, 500, 8, 32, [(16, 16), (8, 8), (4, 4), (2, 2)], 4)
    c["value"] = _round(c["value"], torch.bfloat16)
    for pm, ac in MODES:
        out, gv, gl, ga = run_mixed(c, pm, ac, torch.bfloat16)
        v = torch.from_numpy(c["value"]).to(DEV).requires_grad_(True)
        l = torch.from_numpy(c["loc"]).to(DEV).requires_grad_(True)
        a = torch.from_numpy(c["attn"]).to(DEV).requires_grad_(True)
        o32 = ops.multiscale_deformable_attention(v, torch.from_numpy(c["shapes"]).to(DEV), l, a, pm, ac)
        o32.backward(torch.from_numpy(c["grad_out"]).to(DEV))
        np.testing.assert_array_equal(out, o32.detach().cpu().numpy())
        np.testing.assert_array_equal(gl, l.grad.cpu().numpy())
        np.testing.assert_array_equal(ga, a.grad.cpu().numpy())
        # grad_value: the fp32 sums rounded once (RNE); the sort's record order is not deterministic, so a sum may
        # differ in its last fp32 bit between two runs and land on the neighbouring bf16 value
        np.testing.assert_allclose(gv, v.grad.to(torch.bfloat16).float().cpu().numpy(), rtol=1e-2, atol=1e-6)


def test_mixed_storage_python_and_cpp_routes_agree(monkeypatch):
    from msda_triton_amd import _ext, functional
    if _ext.load() is None:
        pytest.skip("C++ autograd glue not built")
    rng = np.random.default_rng(6)
    c = rand_case(rng, 1, 200, 4, 32, [(10, 10), (5, 5)], 4)
    a = run_mixed(c, "zeros", False, torch.float16)
    fn = functional._HipMultiscaleDeformableAttentionFunction.apply
    b = run_mixed(c, "zeros", False, torch.float16, fn=fn)
    for i, (x, y) in enumerate(zip(a, b)):
        if i == 1:  # grad_value: record order, hence the last bit of a sum, may differ between two runs
            np.testing.assert_allclose(x, y, rtol=2e-3, atol=1e-6)
        else:
            np.testing.assert_array_equal(x, y)


def test_mixed_storage_rejects_other_combinations():
    ops = _ops()
    s = torch.tensor([[4, 4]], device=DEV)
    v = torch.zeros(1, 16, 2, 8, device=DEV)
    pts = torch.rand(1, 3, 2, 1, 2, 2, device=DEV)
    att = torch.rand(1, 3, 2, 1, 2, device=DEV)
    with pytest.raises(ValueError, match="share one dtype"):
        ops.multiscale_deformable_attention(v, s, pts.half(), att.half(), "border", True)       # fp32 img, fp16 inputs
    with pytest.raises(ValueError, match="share one dtype"):
        ops.multiscale_deformable_attention(v.bfloat16(), s, pts.half(), att.half(), "border", True)
    with pytest.raises(ValueError, match="share one dtype"):
        ops.multiscale_deformable_attention(v.half(), s, pts, att.half(), "border", True)
    with pytest.raises(ValueError, match="share one dtype"):
        ops.multiscale_deformable_attention(v.half(), s, pts.double(), att.double(), "border", True)
    out = ops.multiscale_deformable_attention(v.half(), s, pts, att, "border", True)
    assert out.dtype == torch.float32


@pytest.mark.parametrize("ref_dim", [2, 4])
def test_mixed_storage_fused_core_matches_composition(ref_dim):
    """fused_module_core with a bf16 pyramid and fp32 projection == prologue in PyTorch + the mixed operator."""
    from msda_triton_amd.functional import fused_module_core, module_sampling_inputs, multiscale_deformable_attention
    torch.manual_seed(3)
    B, Q, H, D, L, P = 2, 150, 4, 32, 3, 4
    shapes = torch.tensor([[12, 9], [6, 5], [3, 3]], device=DEV)
    I = int((shapes[:, 0] * shapes[:, 1]).sum())  # noqa: E741
    res = []
    for fused in (True, False):
        g = torch.Generator(device=DEV).manual_seed(11)
        img = torch.randn(B, I, H, D, device=DEV, generator=g).bfloat16().requires_grad_(True)
        proj = (0.5 * torch.randn(B, Q, H, L, P, 3, device=DEV, generator=g)).requires_grad_(True)
        ref = torch.rand(B, Q, ref_dim, device=DEV, generator=g)
        if ref_dim == 4:
            ref[..., 2:] = 0.1 + 0.3 * ref[..., 2:]
        ref.requires_grad_(True)
        if fused:
            out = fused_module_core(img, shapes, proj, ref, "zeros", False)
        else:
            pts, att = module_sampling_inputs(proj, shapes, ref)
            out = multiscale_deformable_attention(img, shapes, pts, att, "zeros", False)
        assert out.dtype == torch.float32
        go = torch.rand(out.shape, device=DEV, generator=g)
        out.backward(go)
        assert img.grad.dtype == torch.bfloat16 and proj.grad.dtype == torch.float32
        res.append([t.detach().float().cpu().numpy() for t in (out, img.grad, proj.grad, ref.grad)])
    for nm, x, y in zip(("out", "g_img", "g_proj", "g_ref"), *res):
        tol = dict(atol=2e-2, rtol=1e-2) if nm == "g_img" else dict(atol=2e-4, rtol=1e-3) if nm == "out" else dict(atol=1e-3, rtol=1e-2)
        np.testing.assert_allclose(x, y, err_msg=nm, **tol)


@pytest.mark.parametrize("autocast", [False, True], ids=["plain", "autocast"])
def test_module_value_dtype_option(autocast):
    """MultiscaleDeformableAttention(value_dtype=torch.bfloat16): same parameters, output and parameter gradients
    within bf16 storage error of the default module; the attention core saw a bf16 pyramid and fp32 sampling inputs."""
    from msda_triton_amd import MultiscaleDeformableAttention, functional
    torch.manual_seed(0)
    kw = dict(emb_dim=64, hidden_dim=64, num_levels=3, num_heads=4, num_points=4, padding_mode="border", align_corners=True)
    base = MultiscaleDeformableAttention(**kw).to(DEV)
    mixed = MultiscaleDeformableAttention(**kw, value_dtype=torch.bfloat16).to(DEV)
    mixed.load_state_dict(base.state_dict())
    shapes = torch.tensor([[10, 12], [5, 6], [3, 3]], device=DEV)
    I = int((shapes[:, 0] * shapes[:, 1]).sum())  # noqa: E741
    img = torch.randn(2, I, 64, device=DEV)
    queries = torch.randn(2, 90, 64, device=DEV)
    ref = torch.rand(2, 90, 2, device=DEV)
    seen = {}
    orig = functional.msda_hip_fwd_fused

    def spy(img_, shapes_, proj_, ref_, *a, **k):
        seen["dtypes"] = (img_.dtype, proj_.dtype, ref_.dtype)
        return orig(img_, shapes_, proj_, ref_, *a, **k)

    outs, grads = [], []
    for m in (base, mixed):
        m.zero_grad()
        if m is mixed:
            functional.msda_hip_fwd_fused = spy
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                if m is mixed:
                    with functional.KernelTimer():  # (the timer keeps the call on the Python launchers, where the spy sits)
                        out = m(img, shapes, queries, ref)
                else:
                    out = m(img, shapes, queries, ref)
        finally:
            functional.msda_hip_fwd_fused = orig
        out.float().sum().backward()
        outs.append(out.detach().float().cpu().numpy())
        grads.append({k: p.grad.detach().float().cpu().numpy() for k, p in m.named_parameters()})
    # (under autocast the projection arrives in bf16 as well: the 16-bit storage kernels, fp32 reference points)
    assert seen["dtypes"] == (torch.bfloat16, torch.bfloat16 if autocast else torch.float32, torch.float32)
    # relative error in the Frobenius norm (under autocast the default module forms its sampling points from bf16
    # offsets, this one from their fp32 copies: single entries next to a grid kink may differ visibly, the tensors not)
    bound = 5e-2 if autocast else 1e-2

    def rel(x, y):
        return float(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-12))

    assert rel(outs[1], outs[0]) < bound, rel(outs[1], outs[0])
    for k in grads[0]:
        assert rel(grads[1][k], grads[0][k]) < bound, (k, rel(grads[1][k], grads[0][k]))


def test_mixed_storage_compiled_op():
    """The registered custom ops carry the mixed dtypes through torch.compile (fake kernels: fp32 result, 16-bit
    value gradient)."""
    ops = _ops()
    rng = np.random.default_rng(9)
    c = rand_case(rng, 1, 64, 2, 16, [(6, 6), (3, 3)], 2)
    f = torch.compile(ops.multiscale_deformable_attention, fullgraph=True, backend="aot_eager")
    eager = run_mixed(c, "border", False, torch.float16)
    comp = run_mixed(c, "border", False, torch.float16, fn=f)
    for i, (x, y) in enumerate(zip(eager, comp)):
        if i == 1:
            np.testing.assert_allclose(x, y, rtol=2e-3, atol=1e-6)
        else:
            np.testing.assert_array_equal(x, y)


def test_default_module_under_autocast_takes_the_fused_kernels():
    """bf16 projections next to fp32 reference points (what autocast hands the attention core) run the fused prologue
    kernels — fp32 arithmetic like everything under autocast (frontend.py:111), since round 5 on the bf16 tensors as they
    are (16-bit storage entry points) instead of fp32 copies — forward and backward."""
    from msda_triton_amd import MultiscaleDeformableAttention, functional
    torch.manual_seed(0)
    m = MultiscaleDeformableAttention(64, 64, 3, 4, 4, "zeros", False).to(DEV)
    shapes = torch.tensor([[10, 12], [5, 6], [3, 3]], device=DEV)
    I = int((shapes[:, 0] * shapes[:, 1]).sum())  # noqa: E741
    img, queries = torch.randn(2, I, 64, device=DEV), torch.randn(2, 90, 64, device=DEV)
    ref = torch.rand(2, 90, 4, device=DEV)
    want = m(img, shapes, queries, ref)
    calls = []
    orig_f, orig_b = functional.msda_hip_fwd_fused, functional.msda_hip_bwd_fused
    functional.msda_hip_fwd_fused = lambda *a, **k: (calls.append(("fwd", a[0].dtype, a[2].dtype)), orig_f(*a, **k))[1]
    functional.msda_hip_bwd_fused = lambda *a, **k: (calls.append(("bwd", a[1].dtype, a[3].dtype)), orig_b(*a, **k))[1]
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got = m(img, shapes, queries, ref)
        got.float().sum().backward()
    finally:
        functional.msda_hip_fwd_fused, functional.msda_hip_bwd_fused = orig_f, orig_b
    assert calls == [("fwd", torch.bfloat16, torch.bfloat16), ("bwd", torch.bfloat16, torch.bfloat16)]
    err = float((got.float() - want).norm() / want.norm())
    assert err < 3e-2, err
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
