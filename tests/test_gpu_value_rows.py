"""ABI 11: padded value rows (``value_row_stride``) and passes over the batch (``MSDA_WS_PASSES`` / option ``ws_passes``).

Neither exists in the reference (its launcher copies every input dense, /root/reference/src/msda_triton/kernels.py:367-370,
and its backward needs no workspace); both must leave every RESULT untouched, so each case is held BIT-EXACT to the dense /
one-pass run of the same kernels — which the oracle-backed parity tests pin — and one case per entry point also to the
oracle directly.  Run with ``-m gpu``."""
import numpy as np
import pytest
import torch

from msda_triton_amd import _lib, functional
from msda_triton_amd.functional import (_HipMultiscaleDeformableAttentionFunction, fused_module_core,
                                        multiscale_deformable_attention, padded_value_rows, value_row_pad)

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
LEVELS = [(20, 24), (10, 12), (5, 6), (3, 3)]


def _case(B, Q, H, D, P, dtype, seed, levels=LEVELS):
    g = torch.Generator().manual_seed(seed)
    L, I = len(levels), sum(h * w for h, w in levels)  # noqa: E741
    value = torch.randn(B, I, H, D, generator=g).to(dtype).to(DEV)
    loc = (torch.rand(B, Q, H, L, P, 2, generator=g) * 1.2 - 0.1).to(dtype).to(DEV)
    attn = torch.softmax(torch.randn(B, Q, H, L * P, generator=g), -1).reshape(B, Q, H, L, P).to(dtype).to(DEV)
    go = torch.randn(B, Q, H, D, generator=g).to(dtype).to(DEV)
    return value, torch.tensor(levels, device=DEV), loc, attn, go


def _padded(value, pad_bytes=None):
    B, I, H, D = value.shape  # noqa: E741
    v = padded_value_rows(B, I, H, D, value.dtype, value.device, pad_bytes)
    v.copy_(value)
    return v


def _run(op, value, shapes, loc, attn, go, pm, ac):
    v, l, a = value.detach().requires_grad_(), loc.detach().requires_grad_(), attn.detach().requires_grad_()
    out = op(v, shapes, l, a, pm, ac)
    out.backward(go)
    torch.cuda.synchronize()
    return out.detach(), v.grad, l.grad, a.grad


def _py_route(v, s, l, a, pm, ac):
    return _HipMultiscaleDeformableAttentionFunction.apply(v, s, l, a, pm, ac, 0)


def test_value_row_pad_rule_and_layout():
    assert value_row_pad(1024) == 128 and value_row_pad(512) == 128 and value_row_pad(256) == 128
    assert value_row_pad(1152) == 0 and value_row_pad(640) == 0 and value_row_pad(144) == 0
    v = padded_value_rows(2, 7, 8, 32, torch.float32, DEV)
    assert tuple(v.shape) == (2, 7, 8, 32) and v.stride() == (7 * 288, 288, 32, 1) and not v.is_contiguous()
    t, row = functional._value_rows(v)
    assert t is v and row == 1152
    t, row = functional._value_rows(v.permute(0, 1, 3, 2))  # any other layout: a dense copy, as the reference makes
    assert t.is_contiguous() and row == 0


@pytest.mark.parametrize("route", ["cpp", "python"])
@pytest.mark.parametrize("dtype,pad", [(torch.float32, None), (torch.float32, 384), (torch.bfloat16, None), (torch.float16, 64),
                                       (torch.float64, 256)])
@pytest.mark.parametrize("pm,ac", [("zeros", False), ("border", True)])
@pytest.mark.parametrize("B,Q", [(2, 70), (3, 1700)])  # the one-wave-per-unit forward + single-launch grad_value; the general kernels + sorted pipeline
def test_padded_rows_are_bit_identical_to_dense(route, dtype, pad, pm, ac, B, Q):
    value, shapes, loc, attn, go = _case(B, Q, 8, 32, 4, dtype, seed=B * 1000 + Q)
    op = multiscale_deformable_attention if route == "cpp" else _py_route
    dense = _run(op, value, shapes, loc, attn, go, pm, ac)
    vp = _padded(value, pad)
    assert functional._value_rows(vp)[1] == 8 * 32 * value.element_size() + (128 if pad is None else pad)
    got = _run(op, vp, shapes, loc, attn, go, pm, ac)
    for name, a, b in zip(("out", "grad_value", "grad_loc", "grad_attn"), got, dense):
        assert a.shape == b.shape and torch.equal(a, b), name
    assert got[1].is_contiguous()  # grad_value is dense whatever the pyramid's layout


def test_padded_rows_with_lds_levels_two_planes_and_touch():
    """The variants that build their own descriptors per plane / per slice: LDS-served levels with one and two planes
    per workgroup, the row touches, the linear block order."""
    value, shapes, loc, attn, go = _case(4, 2600, 8, 32, 4, torch.float32, seed=77, levels=[(32, 32), (16, 16), (8, 8), (4, 4)])
    vp = _padded(value)
    keep = {k: _lib.get_option(k) for k in ("lds_levels", "lds_planes", "touch", "xcd_map", "unit_fwd")}
    try:
        for opts in ({"lds_levels": 2, "lds_planes": 2}, {"lds_levels": 2, "lds_planes": 1}, {"lds_levels": 0, "touch": 2},
                     {"xcd_map": 0}, {"xcd_map": 2}, {"unit_fwd": 2}):
            for k, v in keep.items():
                _lib.set_option(k, v)
            for k, v in opts.items():
                _lib.set_option(k, v)
            # (the same variant on both layouts: the one-wave-per-unit forward sums in another order than the general kernel)
            dense = _run(multiscale_deformable_attention, value, shapes, loc, attn, go, "border", True)
            got = _run(multiscale_deformable_attention, vp, shapes, loc, attn, go, "border", True)
            for a, b in zip(got, dense):
                assert torch.equal(a, b), opts
    finally:
        for k, v in keep.items():
            _lib.set_option(k, v)


def test_c_abi_takes_any_element_multiple_as_the_stride(oracle):
    """Straight through the C ABI: a stride that is a multiple of the element size but not of 16 bytes (the scalar kernels
    then run: rows no longer start on 16-byte boundaries) — forward and sample gradients against the oracle."""
    value, shapes, loc, attn, go = _case(2, 150, 4, 32, 3, torch.float32, seed=12)
    B, I, H, D = value.shape  # noqa: E741
    Q, L, P = 150, len(LEVELS), 3
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    host = [t.cpu().numpy() for t in (value, shapes, loc, attn)]
    want = oracle.forward(*host, "border", False)
    _, want_gl, want_ga = oracle.backward(go.cpu().numpy(), *host, "border", False)
    for pad_elems in (1, 3, 4, 33):
        buf = torch.zeros(B, I, H * D + pad_elems, device=DEV)
        buf[:, :, :H * D] = value.reshape(B, I, H * D)
        out, gl, ga = torch.empty(B, Q, H, D, device=DEV), torch.empty_like(loc), torch.empty_like(attn)
        stride = (H * D + pad_elems) * 4
        assert lib.msda_fwd_f32(buf.data_ptr(), shapes.data_ptr(), loc.data_ptr(), attn.data_ptr(), out.data_ptr(),
                                B, I, H, D, Q, L, P, 0, 0, stride, st) == 0
        assert lib.msda_bwd_f32(go.data_ptr(), buf.data_ptr(), shapes.data_ptr(), loc.data_ptr(), attn.data_ptr(), None,
                                gl.data_ptr(), ga.data_ptr(), B, I, H, D, Q, L, P, 0, 0, 0, stride, None, 0, st) == 0
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ga.cpu().numpy(), want_ga, rtol=1e-3, atol=1e-4)
        keep = ~_kinks(host[2], host[1])
        np.testing.assert_allclose(np.where(keep, gl.cpu().numpy(), 0), np.where(keep, want_gl, 0), rtol=1e-3, atol=1e-4)


def _kinks(loc, shapes):
    from conftest import kink_mask
    return kink_mask(loc, shapes, False)


def test_padded_rows_against_the_oracle(oracle):
    value, shapes, loc, attn, go = _case(2, 300, 4, 32, 3, torch.float32, seed=5)
    out, gv, gl, ga = _run(multiscale_deformable_attention, _padded(value), shapes, loc, attn, go, "zeros", False)
    host = [t.cpu().numpy() for t in (value, shapes, loc, attn)]
    np.testing.assert_allclose(out.cpu().numpy(), oracle.forward(*host, "zeros", False), rtol=1e-4, atol=1e-5)
    r_gv, _, r_ga = oracle.backward(go.cpu().numpy(), *host, "zeros", False)
    np.testing.assert_allclose(gv.cpu().numpy(), r_gv, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(ga.cpu().numpy(), r_ga, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("kind", ["f32", "mixed_bf16", "storage_bf16"])
@pytest.mark.parametrize("ref_dim", [2, 4])
def test_fused_module_core_reads_padded_rows(kind, ref_dim):
    g = torch.Generator().manual_seed(9)
    B, Q, H, D, P, L = 2, 1500, 8, 32, 4, len(LEVELS)
    I = sum(h * w for h, w in LEVELS)  # noqa: E741
    vdt = torch.float32 if kind == "f32" else torch.bfloat16
    pdt = torch.bfloat16 if kind == "storage_bf16" else torch.float32
    value = torch.randn(B, I, H, D, generator=g).to(vdt).to(DEV)
    proj = (torch.randn(B, Q, H, L, P, 3, generator=g) * torch.tensor([2.0, 2.0, 1.0])).to(pdt).to(DEV)
    ref = torch.rand(B, Q, ref_dim, generator=g).to(DEV)
    if ref_dim == 4:
        ref[..., 2:] = ref[..., 2:] * 0.4 + 0.05
    go = torch.randn(B, Q, H, D, generator=g).to(pdt).to(DEV)
    shapes = torch.tensor(LEVELS, device=DEV)

    def run(v):
        v, p, r = v.detach().requires_grad_(), proj.detach().requires_grad_(), ref.detach().requires_grad_()
        out = fused_module_core(v, shapes, p, r, "border", True)
        out.backward(go)
        torch.cuda.synchronize()
        return out.detach(), v.grad, p.grad, r.grad

    dense, got = run(value), run(_padded(value))
    for name, a, b in zip(("out", "grad_value", "grad_proj", "grad_ref"), got, dense):
        assert torch.equal(a, b), (kind, name)


def test_module_writes_its_value_projection_padded(monkeypatch):
    """The nn.Module owns the value pyramid: its projection GEMM writes the pixels' rows 128 bytes apart (training through
    the row-split Function, inference through the plain GEMM) and the kernels are handed that view — no dense copy —;
    outputs and parameter gradients equal the dense module's."""
    from msda_triton_amd import MultiscaleDeformableAttention
    torch.manual_seed(3)
    levels = [(48, 48), (24, 24), (12, 12), (6, 6)]
    I = sum(h * w for h, w in levels)  # noqa: E741
    B, Q, E = 4, 8192, 256  # (>= 32 768 rows: the size from which the module pads)
    m = MultiscaleDeformableAttention(E, E, 4, 8, 4, "border", True).to(DEV)
    img = torch.randn(B, I, E, device=DEV, requires_grad=True)
    q = torch.randn(B, Q, E, device=DEV, requires_grad=True)
    ref = torch.rand(B, Q, 2, device=DEV)
    shapes = torch.tensor(levels, device=DEV)
    go = torch.randn(B, Q, E, device=DEV)
    seen = []
    real = functional._value_rows

    def spy(v):
        t, row = real(v)
        seen.append((row, t.data_ptr() == v.data_ptr()))
        return t, row

    monkeypatch.setattr(functional, "_value_rows", spy)

    def step(train=True):
        if not train:
            with torch.no_grad():
                return [m(img, shapes, q, ref)]
        out = m(img, shapes, q, ref)
        return [out] + list(torch.autograd.grad(out, [img, q] + list(m.parameters()), go))

    ext = functional._ext.load
    monkeypatch.setattr(functional._ext, "load", lambda: None)  # (the Python launchers: the spy sees every call)
    got, got_inf = step(), step(False)
    assert seen and all(row == 1024 + 128 and same for row, same in seen), seen
    monkeypatch.setattr(functional, "value_row_pad", lambda n: 0)
    seen.clear()
    want, want_inf = step(), step(False)
    assert seen and all(row == 0 for row, _ in seen), seen
    monkeypatch.setattr(functional._ext, "load", ext)
    for a, b in zip(got + got_inf, want + want_inf):
        scale = float(b.abs().max())
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5 * scale)  # (GEMMs with another leading dimension may sum in another order)


# ------------------------------------------------------------------------------------------
# passes over the batch
# ------------------------------------------------------------------------------------------
def _same_to_the_last_bit(a, b):
    """grad_value of two pass counts: sums of the same terms, possibly in another order (a few ulps of the largest term)."""
    assert a.shape == b.shape and a.dtype == b.dtype
    if torch.equal(a, b):
        return
    scale = float(b.float().abs().max())
    tol = {torch.float32: 4e-6, torch.float64: 1e-14}.get(a.dtype, 1.6e-2)  # (16-bit results: one rounding of the fp32 sum)
    torch.testing.assert_close(a.float(), b.float(), rtol=tol, atol=tol * scale)


def test_workspace_shrinks_with_the_passes():
    lib = _lib.load()
    c2 = (4, 5440, 8, 32, 10000, 4, 4)
    rg = _lib.WS_RECORDS_IN_GRADS
    keep = _lib.get_option("ws_passes")
    try:
        _lib.set_option("ws_passes", 1)  # (the suite may run under MSDA_TEST_OPTS=ws_passes=n)
        one = lib.msda_bwd_workspace_bytes(*c2, 4, 4, 0, rg)
        two = lib.msda_bwd_workspace_bytes(*c2, 4, 4, 0, rg | _lib.ws_passes(2))
        four = lib.msda_bwd_workspace_bytes(*c2, 4, 4, 0, rg | _lib.ws_passes(4))
        # (the partial rows halve with the planes; the per-slice cell tables do not — fewer planes are cut into more slices)
        assert one > 1.7 * two and two > 1.5 * four > 0
        assert lib.msda_bwd_workspace_bytes(*c2, 4, 4, 0, rg | _lib.ws_passes(64)) == four  # (one batch element per pass at least)
        _lib.set_option("ws_passes", 2)
        assert lib.msda_bwd_workspace_bytes(*c2, 4, 4, 0, rg) == two                      # the option: the query's default
        assert lib.msda_bwd_workspace_bytes(*c2, 4, 4, 0, rg | _lib.ws_passes(1)) == one  # (a flag wins)
        assert lib.msda_bwd_fused_workspace_bytes(*c2, 4, 4, 0, 0) < lib.msda_bwd_fused_workspace_bytes(*c2, 4, 4, 0, _lib.ws_passes(1))
    finally:
        _lib.set_option("ws_passes", keep)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,passes", [(4, 2), (4, 4), (3, 2), (5, 8)])
@pytest.mark.parametrize("need_sample", [True, False])
def test_passes_over_the_batch_are_bit_identical(dtype, B, passes, need_sample):
    """ws_passes = n: the size queries return the n-pass workspace, the backward given it runs the sorted pipeline once per
    group of ceil(B / n) batch elements (B = 3 in 2 passes: groups of 2 + 1; B = 5 asked for 8: one element per pass) —
    grad_loc / grad_attn bit-identical to the one-pass run and grad_value too wherever the groups keep the slice count
    (else equal to the last bit's rounding), with the records in the gradient buffers (all three gradients) and in the
    workspace (grad_value alone)."""
    value, shapes, loc, attn, go = _case(B, 1700, 4, 32, 4, dtype, seed=B * 31 + passes)

    def run():
        v, l, a = value.detach().requires_grad_(), loc.detach().requires_grad_(need_sample), attn.detach().requires_grad_(need_sample)
        out = multiscale_deformable_attention(v, shapes, l, a, "zeros", False)
        out.backward(go)
        torch.cuda.synchronize()
        return (v.grad, l.grad, a.grad) if need_sample else (v.grad,)

    keep = _lib.get_option("ws_passes")
    try:
        _lib.set_option("ws_passes", 1)
        one = run()
        assert _lib.last_launch_info()["value_passes"] == 1 and _lib.last_launch_info()["value_path"] == 2
        _lib.set_option("ws_passes", passes)
        got = run()
        info = _lib.last_launch_info()
        assert info["value_passes"] == -(-B // -(-B // min(passes, B))), info
        # grad_loc / grad_attn: the same kernel on the same inputs — bit-exact.  grad_value: each pass count is bitwise
        # reproducible by itself, but a group of fewer planes is cut into more query slices (sorted_ws_layout: enough
        # workgroups to fill the chip), a cell's records then sit in another order and its fp32 sum may round differently
        # in the last bit (B = 5: 13 slices per plane in one pass, 14 with one batch element per pass)
        for a, b in zip(got[1:], one[1:]):
            assert torch.equal(a, b)
        _same_to_the_last_bit(got[0], one[0])
    finally:
        _lib.set_option("ws_passes", keep)


def test_passes_in_the_fused_backward_and_through_the_python_route():
    g = torch.Generator().manual_seed(21)
    B, Q, H, D, P, L = 4, 1500, 8, 32, 4, len(LEVELS)
    I = sum(h * w for h, w in LEVELS)  # noqa: E741
    value = torch.randn(B, I, H, D, generator=g).to(DEV)
    proj = (torch.randn(B, Q, H, L, P, 3, generator=g) * torch.tensor([2.0, 2.0, 1.0])).to(DEV)
    ref = torch.rand(B, Q, 2, generator=g).to(DEV)
    go = torch.randn(B, Q, H, D, generator=g).to(DEV)
    shapes = torch.tensor(LEVELS, device=DEV)
    loc, attn = functional.module_sampling_inputs(proj, shapes, ref)

    def fused():
        v, p = value.detach().requires_grad_(), proj.detach().requires_grad_()
        fused_module_core(v, shapes, p, ref, "border", True).backward(go)
        torch.cuda.synchronize()
        return v.grad, p.grad

    def python_route():
        return _run(_py_route, value, shapes, loc, attn, go, "border", True)[1:]

    keep = _lib.get_option("ws_passes")
    try:
        for fn in (fused, python_route):
            _lib.set_option("ws_passes", 1)
            one = fn()
            _lib.set_option("ws_passes", 2)
            two = fn()
            assert _lib.last_launch_info()["value_passes"] == 2
            _same_to_the_last_bit(two[0], one[0])  # grad_value
            for a, b in zip(two[1:], one[1:]):
                assert torch.equal(a, b)
    finally:
        _lib.set_option("ws_passes", keep)


def test_last_launch_info_names_the_variants():
    value, shapes, loc, attn, go = _case(2, 40, 8, 32, 4, torch.float32, seed=1)
    with torch.no_grad():
        multiscale_deformable_attention(value, shapes, loc, attn, "border", True)
    info = _lib.last_launch_info()
    assert info["fwd_variant"] == 2 and info["fwd_lds_level_bytes"] == 0  # 640 units: one wave per unit
    lib = _lib.load()
    assert lib.msda_last_launch_info(b"no_such_key") == -1
