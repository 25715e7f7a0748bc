"""The module kernels with 16-bit STORAGE (``msda_{fwd,bwd}_fused_f32_sbf16`` / ``_sf16``): value pyramid, projection,
result and their gradients in bfloat16 / float16, reference points and all arithmetic in float32 — what the reference's
module core computes under autocast (it casts every input to fp32, frontend.py:111), without the fp32 copies.

Checked against the fp32 fused kernels on the SAME 16-bit-rounded inputs: the arithmetic is the same, so the results
agree to the storage rounding of the outputs (one 16-bit ulp).  Run with ``-m gpu``."""
import zlib

import pytest
import torch

from msda_triton_amd import MultiscaleDeformableAttention, functional
from msda_triton_amd.functional import fused_module_core

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
ULP = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}
LEVELS = [(20, 24), (10, 12), (5, 6), (3, 3)]


def make(B, Q, H, D, levels, P, ref_dim, sdt, seed, spread=0.6):
    g = torch.Generator(device="cpu").manual_seed(seed)
    L = len(levels)
    I = sum(h * w for h, w in levels)  # noqa: E741
    value = torch.randn(B, I, H, D, generator=g).to(sdt)
    proj = (torch.randn(B, Q, H, L, P, 3, generator=g) * torch.tensor([spread * 8, spread * 8, 1.0])).to(sdt)
    ref = torch.rand(B, Q, ref_dim, generator=g)
    if ref_dim == 4:
        ref[..., 2:] = ref[..., 2:] * 0.4 + 0.05
    go = torch.randn(B, Q, H, D, generator=g).to(sdt)
    shapes = torch.tensor(levels)
    return [t.to(DEV) for t in (value, shapes, proj, ref, go)]


def run(value, shapes, proj, ref, go, pm, ac, level_shapes=None):
    value, proj, ref = value.detach().requires_grad_(), proj.detach().requires_grad_(), ref.detach().requires_grad_()
    out = fused_module_core(value, shapes, proj, ref, pm, ac, level_shapes)
    out.backward(go.to(out.dtype))
    return out.detach(), value.grad, proj.grad, ref.grad


def close(a, b32, sdt, what, scale_tol=2.0):
    """`a` (16-bit) against the fp32 result rounded the same way: within `scale_tol` 16-bit ulps of the value's scale."""
    assert a.dtype == sdt, (what, a.dtype)
    a, b = a.float(), b32.float()
    tol = ULP[sdt] * scale_tol
    err = (a - b).abs()
    bound = tol * b.abs().clamp_min(b.abs().max() * 1e-3)
    assert bool((err <= bound + 1e-30).all()), (what, float((err / bound.clamp_min(1e-30)).max()))


@pytest.mark.parametrize("sdt", [torch.bfloat16, torch.float16], ids=["sbf16", "sf16"])
@pytest.mark.parametrize("pm,ac", [("border", True), ("zeros", False), ("border", False), ("zeros", True)])
@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("shape", ["small", "sorted", "d64"])
def test_storage_kernels_match_the_fp32_fused_kernels_on_rounded_inputs(sdt, pm, ac, ref_dim, shape):
    B, Q, H, D, P = {"small": (2, 90, 4, 32, 4), "sorted": (2, 2600, 8, 32, 4), "d64": (1, 300, 2, 64, 2)}[shape]
    value, shapes, proj, ref, go = make(B, Q, H, D, LEVELS, P, ref_dim, sdt, seed=zlib.crc32(f"{shape}{ref_dim}".encode()) % 1000)
    assert functional.fused_storage_dtypes(value.dtype, proj.dtype, ref.dtype)
    out, gv, gp, gr = run(value, shapes, proj, ref, go, pm, ac)
    o32, gv32, gp32, gr32 = run(value.float(), shapes, proj.float(), ref, go.float(), pm, ac)
    assert gr.dtype == torch.float32
    close(out, o32, sdt, "out")
    close(gp, gp32, sdt, "grad_proj")
    close(gv, gv32, sdt, "grad_value", scale_tol=3.0)
    torch.testing.assert_close(gr, gr32, rtol=2e-4, atol=2e-4 * float(gr32.abs().max()))


def oracle_module_core(oracle, value, shapes, proj, ref, go, pm, ac):
    """The module core on the host in float64: the reference module's prologue (softmax over L * P, offsets -> sampling
    points; frontend.py:253-284) in plain PyTorch, the operator and its three gradients from the C oracle
    (oracle/msda_oracle.c: the restatement pinned to the reference's fixtures), the prologue's chain rule by autograd."""
    import numpy as np
    v64, p64, r64, g64 = (t.detach().cpu().double() for t in (value, proj, ref, go))
    p64.requires_grad_(True)
    r64.requires_grad_(True)
    sh = shapes.cpu()
    pts, att = functional.module_sampling_inputs(p64, sh, r64)
    args = (v64.numpy(), sh.numpy(), pts.detach().numpy(), att.detach().numpy())
    out = oracle.forward(*args, pm, ac)
    gv, gl, ga = oracle.backward(g64.numpy(), *args, pm, ac)
    gp, gr = torch.autograd.grad([pts, att], [p64, r64], [torch.from_numpy(np.ascontiguousarray(gl)),
                                                          torch.from_numpy(np.ascontiguousarray(ga))])
    return torch.from_numpy(out), torch.from_numpy(gv), gp, gr


@pytest.mark.parametrize("sdt", [torch.bfloat16, torch.float16], ids=["sbf16", "sf16"])
@pytest.mark.parametrize("pm,ac", [("border", True), ("zeros", False), ("border", False), ("zeros", True)])
@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("shape", ["small", "sorted"])
def test_storage_kernels_match_the_oracle_on_rounded_inputs(oracle, sdt, pm, ac, ref_dim, shape):
    """the 16-bit storage kernels held to the ORACLE, not to another HIP kernel — the inputs as the
    kernels see them (value, projection, grad_out rounded to 16 bits; reference points fp32), the module core composed on
    the host in float64; results within the storage rounding of each output (the kernels compute in fp32 and round
    once).  ``small``: the single-launch grad_value kernel, ``sorted``: the sorted pipeline."""
    B, Q, H, D, P = {"small": (2, 90, 4, 32, 4), "sorted": (2, 2600, 8, 32, 4)}[shape]
    value, shapes, proj, ref, go = make(B, Q, H, D, LEVELS, P, ref_dim, sdt, seed=zlib.crc32(f"o{shape}{ref_dim}".encode()) % 1000,
                                        spread=0.3)
    out, gv, gp, gr = run(value, shapes, proj, ref, go, pm, ac)
    o64, gv64, gp64, gr64 = oracle_module_core(oracle, value, shapes, proj, ref, go, pm, ac)
    close(out.cpu(), o64, sdt, "out")
    close(gv.cpu(), gv64, sdt, "grad_value", scale_tol=3.0)
    # grad_proj: the offsets' gradients jump where a sample sits on a pixel boundary (a kink of the bilinear surface);
    # 16-bit projections put samples ON such boundaries more often than fp32 ones do — compare away from them
    gp_h, gp_o = gp.cpu().float(), gp64.float()
    err = (gp_h - gp_o).abs()
    bound = ULP[sdt] * 3.0 * gp_o.abs().clamp_min(float(gp_o.abs().max()) * 1e-3)
    bad = err > bound
    assert float(bad.float().mean()) < 0.02, float(bad.float().mean())
    assert bool((~bad[..., 2]).all()) or float(bad[..., 2].float().mean()) < 2e-3  # (logit gradients: smooth)
    assert gr.dtype == torch.float32
    # reference points collect every sample's location gradient of the unit: kinks included, so a norm-wise bound
    rel = float((gr.cpu().double() - gr64).norm() / gr64.norm().clamp_min(1e-30))
    assert rel < 2e-2, rel


def test_storage_path_needs_fp32_reference_points_and_matching_16_bit_dtypes():
    value, shapes, proj, ref, go = make(1, 20, 2, 32, LEVELS, 2, 2, torch.bfloat16, 3)
    assert not functional.fused_storage_dtypes(torch.bfloat16, torch.float16, torch.float32)
    assert not functional.fused_storage_dtypes(torch.float32, torch.float32, torch.float32)
    with pytest.raises(ValueError, match="share one dtype"):
        functional.msda_hip_fwd_fused(value, shapes, proj, ref.double(), "border", True)


def test_storage_path_falls_back_to_the_unfused_operator_when_the_prologue_does_not_fit():
    # L * P beyond the fused kernels' limit: the prologue in PyTorch (fp32) around the mixed-storage operator
    levels = [(6, 6)] * 8
    value, shapes, proj, ref, go = make(1, 40, 2, 32, levels, 160, 2, torch.bfloat16, 5)
    out, gv, gp, gr = run(value, shapes, proj, ref, go, "border", True)
    o32, gv32, gp32, gr32 = run(value.float(), shapes, proj.float(), ref, go.float(), "border", True)
    close(out, o32, torch.bfloat16, "out")
    close(gp, gp32, torch.bfloat16, "grad_proj", scale_tol=3.0)
    close(gv, gv32, torch.bfloat16, "grad_value", scale_tol=3.0)


@pytest.mark.parametrize("value_dtype", [None, torch.bfloat16])
def test_module_under_autocast_takes_the_storage_kernels_and_matches_the_fp32_core(value_dtype, monkeypatch):
    torch.manual_seed(4)
    levels = [(32, 32), (16, 16), (8, 8), (4, 4)]
    I = sum(h * w for h, w in levels)  # noqa: E741
    B, Q, E = 2, 700, 128
    m = MultiscaleDeformableAttention(E, E, 4, 8, 4, "border", True, value_dtype=value_dtype).to(DEV)
    img = torch.randn(B, I, E, device=DEV, requires_grad=True)
    q = torch.randn(B, Q, E, device=DEV, requires_grad=True)
    ref = torch.rand(B, Q, 2, device=DEV)
    shapes = torch.tensor(levels, device=DEV)
    go = torch.randn(B, Q, E, device=DEV)
    seen = []
    real = functional.msda_hip_fwd_fused

    def spy(img_, shapes_, proj_, ref_, *a, **k):
        seen.append((img_.dtype, proj_.dtype, ref_.dtype))
        return real(img_, shapes_, proj_, ref_, *a, **k)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(img, shapes, q, ref)
        return [out] + list(torch.autograd.grad(out, [img, q] + list(m.parameters()), go.to(out.dtype)))

    monkeypatch.setattr(functional, "msda_hip_fwd_fused", spy)
    got = step()
    assert seen == [(torch.bfloat16, torch.bfloat16, torch.float32)], seen
    # the reference's behaviour: the core in fp32 on fp32 copies (custom_fwd cast_inputs)
    monkeypatch.setattr(functional, "fused_storage_dtypes", lambda *a: False)
    seen.clear()
    m.value_dtype = None
    orig_forward = MultiscaleDeformableAttention.forward

    def fp32_core_forward(self, img_, shapes_, q_, ref_, level_shapes=None):
        from msda_triton_amd._linear import projection
        B_, I_, _ = img_.shape
        N_ = q_.shape[1]
        proj = projection(self.query_input_proj, q_).reshape(B_, N_, self.num_heads, self.num_levels, self.num_points, 3)
        value = projection(self.img_input_proj, img_).reshape(B_, I_, self.num_heads, -1)
        attended = fused_module_core(value, shapes_, proj, ref_, self.padding_mode, self.align_corners, level_shapes)
        return projection(self.query_output_proj, attended.reshape(B_, N_, self.hidden_dim))

    monkeypatch.setattr(MultiscaleDeformableAttention, "forward", fp32_core_forward)
    want = step()
    assert seen and seen[0] == (torch.float32, torch.float32, torch.float32), seen
    monkeypatch.setattr(MultiscaleDeformableAttention, "forward", orig_forward)
    for a, b in zip(got, want):
        assert a.dtype == b.dtype and a.shape == b.shape
        scale = b.float().abs().max().clamp_min(1e-6)
        torch.testing.assert_close(a.float() / scale, b.float() / scale, rtol=3e-2, atol=2e-2)


def test_compiled_module_under_autocast_keeps_the_storage_kernels():
    """torch.compile(fullgraph) of the module under bf16 autocast: the 16-bit storage kernels stay in the graph as the
    registered custom ops (bf16 value / projection, fp32 reference points) and match the eager module."""
    import msda_triton_amd.compile_op  # noqa: F401
    torch.manual_seed(6)
    m = MultiscaleDeformableAttention(32, 32, 2, 4, 3, "zeros", False).to(DEV)
    levels = [(6, 5), (3, 4)]
    s = torch.tensor(levels, device=DEV)
    img = torch.randn(2, sum(h * w for h, w in levels), 32, device=DEV)
    q = torch.randn(2, 21, 32, device=DEV)
    ref = torch.rand(2, 21, 4, device=DEV)
    seen = []
    real = functional.msda_hip_fwd_fused

    def spy(img_, shapes_, proj_, ref_, *a, **k):
        seen.append((img_.dtype, proj_.dtype, ref_.dtype))
        return real(img_, shapes_, proj_, ref_, *a, **k)

    functional.msda_hip_fwd_fused = spy
    try:
        compiled = torch.compile(m, fullgraph=True, backend="aot_eager")
        res = []
        for f in (m, compiled):
            m.zero_grad()
            i_, q_ = img.clone().requires_grad_(True), q.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = f(i_, s, q_, ref)
            out.float().square().sum().backward()
            res.append((out.detach(), i_.grad, q_.grad, m.query_input_proj.weight.grad.clone()))
    finally:
        functional.msda_hip_fwd_fused = real
    assert seen == [(torch.bfloat16, torch.bfloat16, torch.float32)] * 2, seen
    for a, b in zip(*res):
        assert a.dtype == b.dtype
        torch.testing.assert_close(a.float(), b.float(), atol=2e-2, rtol=2e-2)
    v = torch.randn(2, 42, 4, 8, device=DEV).bfloat16()
    pr = torch.randn(2, 21, 4, 2, 3, 3, device=DEV).bfloat16()
    torch.library.opcheck(torch.ops.msda_amd.fused_forward.default, (v, s, pr, ref, False, True),
                          test_utils=("test_schema", "test_faketensor"))
