"""GPU parity at BASELINE.json's full sizes and for the module path, against comparators that are NOT the HIP
kernels: the CPU oracle (oracle/), the reference's digests (tests/golden/digest_*.npz) and the package's own
host-tensor path (plain PyTorch autograd, frontend.py:15-68 / :253-289 formulation).  Run with ``-m gpu``.

Covers what round 1 left open (VERDICT r01 "parity gaps"):
  * c3 (Deformable-DETR encoder shape, bf16) forward AND backward at full size,
  * c5 (stress: Q = 100k, D = 64, L = 5, P = 8, fp16) forward AND backward at full size,
  * the fused module core and the nn.Module's parameter gradients against the CPU host path in fp64.
"""
import numpy as np
import pytest
import torch

from conftest import MODES, kink_mask, mode_key

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    import msda_triton_amd
    return msda_triton_amd


def _np32(t):
    return t.detach().float().cpu().numpy()


def assert_close_lowp(got, ref, rel, what, abs_frac=None):
    """|got - ref| <= rel * |ref| + abs_frac * max|ref|  (abs_frac defaults to rel: a 16-bit result rounds
    relative to its own magnitude, a cancelling sum relative to its terms')."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    scale = float(np.abs(ref).max()) or 1.0
    atol = (rel if abs_frac is None else abs_frac) * scale
    bad = np.abs(got - ref) > rel * np.abs(ref) + atol
    assert not bad.any(), (f"{what}: {int(bad.sum())} of {bad.size} elements off, worst abs err "
                           f"{float(np.abs(got - ref).max()):.4g} (scale {scale:.4g}, rel {rel}, atol {atol:.4g})")


# ------------------------------------------------------------------------------------------
# c3: Deformable-DETR encoder shape, bf16, full size, forward + backward vs the fp32 oracle on the rounded inputs
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("wl_name", ["c3_ddetr_enc", "c3_ddetr_enc_local"])
def test_c3_encoder_bf16_full_size_forward_and_backward(oracle, wl_name):
    """... with uniformly random sampling points, and with what an encoder layer produces: every query sits
    on a pixel, its samples are that pixel's centre + N(0, 2 px) offsets — neighbouring queries hit the same few cells
    (cell lists of very unequal length; ~5 % of the samples fall outside the image)."""
    from msda_triton_amd import synth
    ops = _ops()
    wl = synth.WORKLOADS[wl_name]
    d = synth.make_inputs_torch(wl, "cpu", seed=0, loc_lo=-0.02, loc_hi=1.02)  # bf16 tensors
    v, l, a = (d[k].to(DEV).requires_grad_(True) for k in ("value", "loc", "attn"))
    out = ops.multiscale_deformable_attention(v, d["shapes"].to(DEV), l, a, wl.padding_mode, wl.align_corners)
    out.backward(d["grad_out"].to(DEV))
    f32 = {k: (t.numpy() if k == "shapes" else t.float().numpy()) for k, t in d.items()}
    r_out = oracle.forward(f32["value"], f32["shapes"], f32["loc"], f32["attn"], wl.padding_mode, wl.align_corners)
    r_gv, r_gl, r_ga = oracle.backward(f32["grad_out"], f32["value"], f32["shapes"], f32["loc"], f32["attn"],
                                       wl.padding_mode, wl.align_corners)
    bf16 = 2.0 ** -8  # half an ulp of an 8-bit significand, doubled for the accumulate-order slack
    assert_close_lowp(_np32(out), r_out, bf16, "out")
    assert_close_lowp(_np32(v.grad), r_gv, bf16, "grad_value")
    assert_close_lowp(_np32(a.grad), r_ga, bf16, "grad_attn")
    keep = ~kink_mask(f32["loc"], f32["shapes"], wl.align_corners, tol=1e-3)
    assert keep.mean() > 0.99
    assert_close_lowp(np.where(keep, _np32(l.grad), 0), np.where(keep, r_gl, 0), bf16, "grad_loc")


# ------------------------------------------------------------------------------------------
# c5: stress shape at FULL size (Q = 100 000), fp16, forward + backward
# ------------------------------------------------------------------------------------------
def _valid_weight_sum(loc, shapes, align_corners):
    """Per sample: the sum of the bilinear weights of its in-image corners ("zeros" padding) — an independent
    torch statement used for the mass-conservation property.  loc [B,Q,H,L,P,2] float32 on the GPU."""
    out = torch.empty(loc.shape[:-1], dtype=torch.float64, device=loc.device)
    for lvl, (h, w) in enumerate(shapes):
        per_axis = []
        for coord, size in ((loc[:, :, :, lvl, :, 0].double(), w), (loc[:, :, :, lvl, :, 1].double(), h)):
            pix = coord * (size - 1) if align_corners else coord * size - 0.5
            p0 = pix.floor()
            frac = pix - p0
            ok0 = (p0 >= 0) & (p0 <= size - 1)
            ok1 = (p0 + 1 >= 0) & (p0 + 1 <= size - 1)
            per_axis.append((1 - frac) * ok0 + frac * ok1)
        out[:, :, :, lvl] = per_axis[0] * per_axis[1]
    return out


def test_c5_stress_full_size_fp16_forward_and_backward(oracle):
    """BASELINE configs[4] in full: the sorted-gather grad_value pipeline at 4 M samples per plane with its
    multi-GB workspace, fp16 storage.  out / grad_loc / grad_attn: fp32 oracle on a strided query subset;
    grad_value: the mass-conservation property at full size, and the oracle on a 5 000-query slice."""
    from msda_triton_amd import synth
    ops = _ops()
    wl = synth.WORKLOADS["c5_stress"]
    B, Q, H, D, L, P, I = wl.B, wl.Q, wl.H, wl.D, wl.L, wl.P, wl.I  # noqa: E741
    pm, ac = wl.padding_mode, wl.align_corners
    g = torch.Generator(device=DEV).manual_seed(55)
    value = torch.randn(B, I, H, D, device=DEV, generator=g).half()
    loc = (torch.rand(B, Q, H, L, P, 2, device=DEV, generator=g) * 1.1 - 0.05).half()
    attn = torch.softmax(torch.randn(B, Q, H, L * P, device=DEV, generator=g), -1).reshape(B, Q, H, L, P).half()
    gout = torch.rand(B, Q, H, D, device=DEV, generator=g).half()
    shapes = torch.tensor(wl.levels, device=DEV)
    v, l, a = value.clone().requires_grad_(True), loc.clone().requires_grad_(True), attn.clone().requires_grad_(True)
    out = ops.multiscale_deformable_attention(v, shapes, l, a, pm, ac)
    out.backward(gout)
    torch.cuda.synchronize()
    assert out.shape == (B, Q, H, D)
    for t in (out, v.grad, l.grad, a.grad):
        assert torch.isfinite(t).all()

    fp16 = 2.0 ** -10
    np_shapes = np.asarray(wl.levels, dtype=np.int64)
    # (a) per-query results on a strided subset of the 100k queries (every batch element, every head)
    sel = torch.arange(0, Q, Q // 101, device=DEV)
    sub = dict(value=_np32(value), loc=_np32(loc[:, sel]), attn=_np32(attn[:, sel]), gout=_np32(gout[:, sel]))
    r_out = oracle.forward(sub["value"], np_shapes, sub["loc"], sub["attn"], pm, ac)
    _, r_gl, r_ga = oracle.backward(sub["gout"], sub["value"], np_shapes, sub["loc"], sub["attn"], pm, ac)
    assert_close_lowp(_np32(out[:, sel]), r_out, fp16, "out (query subset)")
    assert_close_lowp(_np32(a.grad[:, sel]), r_ga, fp16, "grad_attn (query subset)")
    # power-of-two levels: fp16 coordinates often land exactly on a pixel centre (1 in 16 at the 128-wide level)
    keep = ~kink_mask(sub["loc"], np_shapes, ac, tol=1e-3)
    assert keep.mean() > 0.9
    assert_close_lowp(np.where(keep, _np32(l.grad[:, sel]), 0), np.where(keep, r_gl, 0), fp16, "grad_loc (query subset)")

    # (b) grad_value at full size: sum over the pixels of a plane == sum_q grad_out * sum_{l,p} attn * (weight of
    # the in-image corners)
    wsum = _valid_weight_sum(loc.float(), wl.levels, ac)                         # [B,Q,H,L,P]
    per_unit = (attn.double() * wsum).sum((-1, -2))                               # [B,Q,H]
    expect = torch.einsum("bqhd,bqh->bhd", gout.double(), per_unit)
    got = v.grad.double().sum(1)
    torch.testing.assert_close(got, expect, atol=0.0, rtol=2e-3)
    # per level too (a level's pixels are a contiguous range): catches mass moved between levels
    start = 0
    for lvl, (h, w) in enumerate(wl.levels):
        per_unit_l = (attn[:, :, :, lvl].double() * wsum[:, :, :, lvl]).sum(-1)
        expect_l = torch.einsum("bqhd,bqh->bhd", gout.double(), per_unit_l)
        got_l = v.grad[:, start:start + h * w].double().sum(1)
        torch.testing.assert_close(got_l, expect_l, atol=0.0, rtol=2e-3, msg=lambda m, k=lvl: f"level {k}: {m}")
        start += h * w

    # (c) grad_value against the oracle on a 5 000-query slice (a separate, smaller call of the same operator)
    qs = 5000
    v2 = value.clone().requires_grad_(True)
    out2 = ops.multiscale_deformable_attention(v2, shapes, loc[:, :qs].contiguous(), attn[:, :qs].contiguous(), pm, ac)
    out2.backward(gout[:, :qs].contiguous())
    r_gv, _, _ = oracle.backward(_np32(gout[:, :qs]), sub["value"], np_shapes, _np32(loc[:, :qs]), _np32(attn[:, :qs]),
                                 pm, ac)
    assert_close_lowp(_np32(v2.grad), r_gv, fp16, "grad_value (5000-query slice)")
    torch.testing.assert_close(out2, out[:, :qs], atol=0, rtol=0)  # a query's output does not depend on the others

    # (d) additivity in the queries ties the full-size grad_value to such slices: the full gradient equals the sum
    # of the gradients of 20 disjoint 5 000-query calls (fp32 sums of fp16-rounded parts)
    acc = torch.zeros_like(value, dtype=torch.float32)
    for q0 in range(0, Q, qs):
        v3 = value.clone().requires_grad_(True)
        o3 = ops.multiscale_deformable_attention(v3, shapes, loc[:, q0:q0 + qs].contiguous(),
                                                 attn[:, q0:q0 + qs].contiguous(), pm, ac)
        o3.backward(gout[:, q0:q0 + qs].contiguous())
        acc += v3.grad.float()
    assert_close_lowp(_np32(v.grad), acc.cpu().numpy(), 2 * fp16, "grad_value (full) vs sum of 20 slices")


# ------------------------------------------------------------------------------------------
# module path: fused kernels against the CPU host path, fp64
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("coords", [2, 4])
@pytest.mark.parametrize("pm,ac", MODES, ids=[mode_key(*m) for m in MODES])
def test_fused_module_core_gradients_match_cpu_host_path_fp64(coords, pm, ac):
    """msda_fwd_fused_f64 / msda_bwd_fused_f64 against autograd through the package's plain-PyTorch host path
    (module_sampling_inputs + native_multiscale_deformable_attention on CPU tensors): out and the gradients of
    value, the raw projection and the reference points.  Non-square levels (the (h, w) normaliser order, Q6)."""
    from msda_triton_amd.functional import KernelTimer, fused_module_core
    levels = [(7, 5), (4, 6), (2, 3)]
    B, Q, H, D, L, P = 2, 41, 3, 16, len(levels), 4
    g = torch.Generator(device="cpu").manual_seed(4321 + coords)
    value = torch.randn(B, sum(h * w for h, w in levels), H, D, generator=g, dtype=torch.float64)
    proj = torch.randn(B, Q, H, L, P, 3, generator=g, dtype=torch.float64) * 1.5
    ref = torch.rand(B, Q, coords, generator=g, dtype=torch.float64)
    gout = torch.rand(B, Q, H, D, generator=g, dtype=torch.float64)
    s = torch.tensor(levels)
    res = {}
    for dev in ("cpu", DEV):
        v, pr, rf = (t.clone().to(dev).requires_grad_(True) for t in (value, proj, ref))
        with KernelTimer() as kt:
            out = fused_module_core(v, s.to(dev), pr, rf, pm, ac)
            out.backward(gout.to(dev))
            if dev != "cpu":
                torch.cuda.synchronize()
        if dev != "cpu":
            assert set(kt.summary()) == {"msda_fwd_fused", "msda_bwd_fused"}, kt.summary()
        else:
            assert not kt.summary()  # nothing of the HIP library ran for the comparator
        res[dev] = [t.detach().cpu() for t in (out, v.grad, pr.grad, rf.grad)]
    for name, a, b in zip(("out", "grad_value", "grad_proj", "grad_ref"), res[DEV], res["cpu"]):
        torch.testing.assert_close(a, b, atol=1e-9, rtol=1e-8, msg=lambda m, n=name: f"{n}: {m}")


@pytest.mark.parametrize("coords", [2, 4])
def test_module_parameter_gradients_match_cpu_host_path_fp64(coords):
    """MultiscaleDeformableAttention (reference frontend.py:175-292): every parameter's gradient and the input
    gradients on the GPU (fused kernels inside) against the same module on CPU tensors, fp64, non-square levels."""
    ops = _ops()
    torch.manual_seed(7 + coords)
    emb, hidden, heads, points = 48, 64, 4, 3
    shapes = [(9, 6), (5, 3), (2, 4)]
    I = sum(h * w for h, w in shapes)  # noqa: E741
    m = ops.MultiscaleDeformableAttention(emb, hidden, len(shapes), heads, points, "zeros", False).double()
    img = torch.randn(2, I, emb, dtype=torch.float64)
    q = torch.randn(2, 23, emb, dtype=torch.float64)
    ref = torch.rand(2, 23, coords, dtype=torch.float64)
    w = torch.randn(2, 23, emb, dtype=torch.float64)
    s = torch.tensor(shapes)
    grads = {}
    for dev in ("cpu", DEV):
        mm = ops.MultiscaleDeformableAttention(emb, hidden, len(shapes), heads, points, "zeros", False).double()
        mm.load_state_dict(m.state_dict())
        mm = mm.to(dev)
        i_, q_, r_ = (t.clone().to(dev).requires_grad_(True) for t in (img, q, ref))
        out = mm(i_, s.to(dev), q_, r_)
        (out * w.to(dev)).sum().backward()
        grads[dev] = {"out": out.detach().cpu(), "img": i_.grad.cpu(), "queries": q_.grad.cpu(), "ref": r_.grad.cpu(),
                      **{n: p.grad.cpu() for n, p in mm.named_parameters()}}
    assert set(grads[DEV]) == set(grads["cpu"]) and len(grads[DEV]) == 4 + 6
    for name in grads["cpu"]:
        torch.testing.assert_close(grads[DEV][name], grads["cpu"][name], atol=1e-9, rtol=1e-7,
                                   msg=lambda m_, n=name: f"{n}: {m_}")


# ------------------------------------------------------------------------------------------
# module path against the REFERENCE's nn.Module: committed fixtures (tests/golden/module_*.npz, written by
# tests/golden/make_golden.py from /root/reference/src/msda_triton/frontend.py:175-292 in the build container)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", __import__("conftest").module_cases(), ids=__import__("conftest").case_id)
def test_gpu_module_matches_reference_module_fixtures(path):
    """The module on the GPU — fused prologue kernels inside (asserted) — DIRECTLY against the reference module's
    outputs: out, the input gradients and all six parameter gradients.  fp64 fixtures at 1e-9, fp32 at fp32 round-off
    (1e-4, the north-star bar, on out)."""
    from conftest import load_module_case
    from msda_triton_amd.functional import KernelTimer
    m, x, want = load_module_case(path, DEV)
    f64 = x["img"].dtype == torch.float64
    tol = dict(atol=1e-9, rtol=1e-8) if f64 else dict(atol=1e-4, rtol=1e-3)
    img, q, ref = (x[k].clone().requires_grad_(True) for k in ("img", "queries", "reference_points"))
    with KernelTimer() as kt:
        out = m(img, x["shapes"], q, ref)
        out.backward(x["grad_out"])
        torch.cuda.synchronize()
    assert set(kt.summary()) == {"msda_fwd_fused", "msda_bwd_fused"}, kt.summary()
    torch.testing.assert_close(out.detach().cpu(), want["out"], **tol)
    torch.testing.assert_close(img.grad.cpu(), want["grad_img"], **tol)
    torch.testing.assert_close(q.grad.cpu(), want["grad_queries"], **tol)
    torch.testing.assert_close(ref.grad.cpu(), want["grad_reference_points"], **tol)
    got = dict(m.named_parameters())
    assert len(got) == 6
    for name, prm in got.items():
        torch.testing.assert_close(prm.grad.cpu(), want["grad__" + name], msg=lambda s, n=name: f"{n}: {s}", **tol)
