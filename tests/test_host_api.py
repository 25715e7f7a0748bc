"""CPU-only: the host-side mirror of the reference interface, the synthetic-workload generator and
the C-ABI library's export table (no compute calls into the library without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import MODES, ROOT, case_id, golden_cases, mode_key


# ---------------------------------------------------------------- C ABI -----------------------
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "msda_hip.h")).read()
    names = set(re.findall(r"MSDA_API\s+(?:const\s+)?\w+\s*\*?\s*(msda_\w+)\s*\(", text))
    names = {n for n in names if "##" not in n}
    # the per-dtype entry points: every `#define MSDA_DECLARE...(SUF)` macro declares its stems for each suffix it is used with
    macros = re.findall(r"#define\s+(MSDA_DECLARE\w*)\(SUF\)((?:.*\\\n)*.*\n)", text)
    assert len(macros) == 2, [m for m, _ in macros]
    for macro, body in macros:
        stems = re.findall(r"MSDA_API\s+int\s+(msda_\w+_)##SUF\s*\(", body)
        assert len(stems) >= 2, (macro, stems)
        for suf in re.findall(macro + r"\((\w+)\)", text):
            if suf != "SUF":
                names |= {stem + suf for stem in stems}
    return names


def test_library_builds_loads_and_exports_every_declared_symbol():
    from msda_triton_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert _lib.load().msda_abi_version() == _lib.ABI_VERSION
    # argument validation happens before anything touches a device
    assert _lib.load().msda_set_option(b"no_such_option", 1) < 0
    assert b"no_such_option" in _lib.load().msda_last_error()
    assert _lib.get_option("xcd_map") in (0, 1)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from msda_triton_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libmsda_hip.so"))
    with pytest.raises(_lib.MSDALibraryError, match="no CPU fallback"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "msda_triton_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "msda_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


# ---------------------------------------------------------------- functional API --------------
@pytest.mark.parametrize("path", golden_cases(), ids=case_id)
def test_native_host_path_matches_reference_golden(path):
    from msda_triton_amd import multiscale_deformable_attention
    z = np.load(path)
    f64 = path.endswith("_f64.npz")
    tol = dict(atol=1e-10, rtol=1e-9) if f64 else dict(atol=2e-5, rtol=1e-4)
    for pm, ac in MODES:
        k = mode_key(pm, ac)
        v = torch.from_numpy(z["value"]).requires_grad_(True)
        l = torch.from_numpy(z["loc"]).requires_grad_(True)
        a = torch.from_numpy(z["attn"]).requires_grad_(True)
        out = multiscale_deformable_attention(v, torch.from_numpy(z["shapes"]), l, a, pm, ac)  # CPU tensors -> native
        np.testing.assert_allclose(out.detach().numpy(), z[f"out_{k}"], **tol)
        if f64:  # float32 kinks are covered by the oracle tests
            out.backward(torch.from_numpy(z["grad_out"]))
            np.testing.assert_allclose(v.grad.numpy(), z[f"grad_value_{k}"], **tol)
            np.testing.assert_allclose(l.grad.numpy(), z[f"grad_loc_{k}"], **tol)
            np.testing.assert_allclose(a.grad.numpy(), z[f"grad_attn_{k}"], **tol)


def test_readme_functional_example_on_cpu():
    """/root/reference/README.md:121-147 with device='cpu', through the reference's import path."""
    from msda_triton import multiscale_deformable_attention
    batch, head_dim, num_queries, num_heads, num_points = 2, 32, 900, 8, 4
    img_shapes = [(64, 64), (32, 32), (16, 16), (8, 8)]
    num_pixels = sum(h * w for h, w in img_shapes)
    img = torch.randn(batch, num_pixels, num_heads, head_dim)
    out = multiscale_deformable_attention(
        img, torch.tensor(img_shapes), torch.rand(batch, num_queries, num_heads, len(img_shapes), num_points, 2),
        torch.rand(batch, num_queries, num_heads, len(img_shapes), num_points), "zeros", False)
    assert out.shape == (batch, num_queries, num_heads, head_dim)


def test_gpu_entry_point_rejects_host_tensors_and_bad_dtypes():
    from msda_triton.frontend import triton_multiscale_deformable_attention as gpu_only
    from msda_triton_amd import multiscale_deformable_attention
    v, s = torch.randn(1, 4, 1, 4), torch.tensor([[2, 2]])
    l, a = torch.rand(1, 1, 1, 1, 1, 2), torch.rand(1, 1, 1, 1, 1)
    with pytest.raises(ValueError, match="gpu"):
        gpu_only(v, s, l, a, "zeros", False)
    with pytest.raises(ValueError, match="Dtype"):
        gpu_only(v.to(torch.int32), s, l, a, "zeros", False)
    with pytest.raises(ValueError, match="padding_mode"):
        multiscale_deformable_attention(v, s, l, a, "reflection", False)
    with pytest.raises(ValueError):
        multiscale_deformable_attention(v, torch.tensor([[2, 2], [1, 1]]), l, a, "zeros", False)


def test_shim_exposes_reference_names():
    import msda_triton
    import msda_triton.frontend as fe
    import msda_triton.kernels as ke
    assert set(msda_triton.__all__) == {"multiscale_deformable_attention", "MultiscaleDeformableAttention"}
    for name in ("triton_multiscale_deformable_attention", "native_multiscale_deformable_attention",
                 "MultiscaleDeformableAttention", "multiscale_deformable_attention"):
        assert hasattr(fe, name)
    for name in ("triton_multi_scale_deformable_attention_fwd", "triton_multi_scale_deformable_attention_bwd"):
        assert hasattr(ke, name)


# ---------------------------------------------------------------- nn.Module -------------------
@pytest.mark.parametrize("coords", [2, 4])
def test_module_cpu(coords):
    """Reference smoke test tests/test_msda.py:154-168 (cpu leg) + state-dict contract."""
    from msda_triton_amd import MultiscaleDeformableAttention
    channels, heads, levels, points = 64, 8, 4, 8
    shapes = [(16 // 2**i, 16 // 2**i) for i in range(levels)]
    I = sum(h * w for h, w in shapes)  # noqa: E741
    m = MultiscaleDeformableAttention(channels, channels // heads * heads, levels, heads, points, "border", True)
    assert set(m.state_dict()) == {f"{n}.{p}" for n in ("img_input_proj", "query_input_proj", "query_output_proj")
                                   for p in ("weight", "bias")}
    assert m.query_input_proj.weight.shape == (heads * levels * points * 3, channels)
    out = m(torch.randn(2, I, channels), torch.tensor(shapes), torch.randn(2, 10, channels), torch.randn(2, 10, coords))
    assert out.shape == (2, 10, channels) and torch.isfinite(out).all()
    out.sum().backward()
    assert all(p.grad is not None for p in m.parameters())


def test_module_errors_and_q6_normaliser_order():
    from msda_triton_amd import MultiscaleDeformableAttention
    with pytest.raises(ValueError, match="divisible"):
        MultiscaleDeformableAttention(32, 30, 2, 4, 2, "zeros", False)
    m = MultiscaleDeformableAttention(8, 8, 1, 1, 1, "zeros", False)
    with pytest.raises(ValueError, match="2 or 4"):
        m(torch.randn(1, 16, 8), torch.tensor([[2, 8]]), torch.randn(1, 1, 8), torch.randn(1, 1, 3))
    # reference quirk (frontend.py:275): (x, y) offsets are divided by img_shapes in (h, w) order
    with torch.no_grad():
        m.query_input_proj.weight.zero_()
        m.query_input_proj.bias.copy_(torch.tensor([1.0, 1.0, 0.0]))
    pts, _ = m.sampling_inputs(torch.tensor([[2, 8]]), torch.zeros(1, 1, 8), torch.zeros(1, 1, 2))
    assert torch.allclose(pts.flatten(), torch.tensor([0.5, 0.125]))


# ---------------------------------------------------------------- synth -----------------------
def test_synth_shards_are_slices_of_the_whole_and_bytes_match_baseline_md():
    from msda_triton_amd import synth
    wl = synth.Workload("t", 2, 24, 2, 4, ((4, 4), (2, 2)), 3)
    full = synth.make_inputs_numpy(wl, seed=5)
    part = synth.make_inputs_numpy(wl, seed=5, q_begin=7, q_end=19)
    for k in ("loc", "attn", "grad_out"):
        assert np.array_equal(part[k], full[k][:, 7:19])
    assert np.array_equal(part["value"], full["value"])
    assert abs(full["attn"].sum(-1) - 1).max() < 1e-12
    c2 = synth.WORKLOADS["c2_q10k"]
    assert (c2.alg_fwd_bytes, c2.alg_fwd_bytes + c2.alg_bwd_bytes) == (124_682_304, 333_086_784)  # 124.68 / 333.09 MB
    assert c2.gather_fwd_bytes == 2_621_440_000
    assert synth.WORKLOADS["c3_ddetr_enc"].I == 17821 and synth.WORKLOADS["c5_stress"].I == 21824


def test_optional_cpp_binding_builds_and_matches_the_abi():
    """csrc/msda_torch_ext.cpp (host-only C++ autograd glue) compiles against the installed PyTorch, loads without a
    GPU and reports the ABI version of the library it is linked to."""
    from msda_triton_amd import _ext, _lib
    path = _ext.build()
    assert os.path.exists(path)
    mod = _ext.load()
    assert mod is not None and int(mod.abi_version()) == _lib.ABI_VERSION


def test_backward_workspace_sizes_stay_within_budget():
    """msda_bwd_workspace_bytes (no GPU needed): the sorted pipeline's scratch is sized from the shapes alone —
    round 1 asked for 319 MB at c2 @ 10k and 5.6 GB at c5; the pixel-indexed partial rows brought that down, and
    problems the single-launch kernel takes need none at all."""
    from msda_triton_amd import _lib, synth
    lib = _lib.load()

    def ws(name):
        wl = synth.WORKLOADS[name]
        return int(lib.msda_bwd_workspace_bytes(wl.B, wl.I, wl.H, wl.D, wl.Q, wl.L, wl.P, wl.elem_size, wl.elem_size, 0, 0))

    assert ws("c2_q10k") <= 200 * 2**20          # entries 82 MB + partial rows 89 MB + cell tables
    # two rounds over the queries: records 1.02 GB + partial rows 0.72 GB + running sums; sized for the scalar
    # (misaligned-pointer) layout too, whose continuation rows add ~70 MB here
    assert ws("c5_stress") <= int(2.1 * 2**30)
    assert ws("c1_readme") == 0 and ws("c4_gdino_dec") == 0 and ws("c2_q1k") == 0
    wl = synth.WORKLOADS["c2_q10k"]
    assert ws("c2_q10k") >= wl.B * wl.H * wl.Q * wl.L * wl.P * 16  # at least the sorted records


def test_module_feeds_the_projection_output_to_the_kernel_in_place(monkeypatch):
    """SURVEY 8f-4 (frontend.py:264-267): the value projection's `[B, I, H*D]` output IS the kernel's `[B, I, H, D]`
    layout — the module hands it over as a view, no reshape copy, no transpose."""
    import torch
    import msda_triton_amd.module as mod
    seen = {}
    real = mod.fused_module_core

    def spy(value, *a, **k):
        seen["value"] = value
        return real(value, *a, **k)

    monkeypatch.setattr(mod, "fused_module_core", spy)
    m = mod.MultiscaleDeformableAttention(16, 32, 2, 4, 2, "zeros", False)
    m.img_input_proj.register_forward_hook(lambda _m, _i, out: seen.__setitem__("proj", out))
    shapes = torch.tensor([(3, 4), (2, 2)])
    m(torch.randn(2, 16, 16), shapes, torch.randn(2, 5, 16), torch.rand(2, 5, 2))
    assert seen["value"].shape == (2, 16, 4, 8) and seen["value"].is_contiguous()
    assert seen["value"].data_ptr() == seen["proj"].data_ptr()          # same storage: a view
    assert seen["value"].untyped_storage().data_ptr() == seen["proj"].untyped_storage().data_ptr()


def test_host_path_with_a_16_bit_pyramid_promotes_to_float32():
    """Mixed storage on host tensors (the GPU entry points msda_*_f32_vbf16 / _vf16 have the same contract): a bf16
    pyramid next to fp32 sampling inputs gives an fp32 result equal to the fp32 computation on the rounded pyramid."""
    import msda_triton_amd as ops
    torch.manual_seed(0)
    shapes = torch.tensor([[5, 4], [3, 2]])
    img = torch.randn(2, 26, 3, 8).bfloat16()
    pts = torch.rand(2, 7, 3, 2, 4, 2)
    att = torch.rand(2, 7, 3, 2, 4)
    out = ops.multiscale_deformable_attention(img, shapes, pts, att, "zeros", False)
    assert out.dtype == torch.float32
    want = ops.multiscale_deformable_attention(img.float(), shapes, pts, att, "zeros", False)
    torch.testing.assert_close(out, want, atol=1e-6, rtol=1e-6)


def test_module_value_dtype_is_validated_and_ignored_on_host_tensors():
    import msda_triton_amd as ops
    kw = dict(emb_dim=16, hidden_dim=16, num_levels=2, num_heads=2, num_points=2, padding_mode="border", align_corners=True)
    with pytest.raises(ValueError, match="value_dtype"):
        ops.MultiscaleDeformableAttention(**kw, value_dtype=torch.float64)
    torch.manual_seed(0)
    a = ops.MultiscaleDeformableAttention(**kw)
    b = ops.MultiscaleDeformableAttention(**kw, value_dtype=torch.bfloat16)
    b.load_state_dict(a.state_dict())
    shapes = torch.tensor([[4, 3], [2, 2]])
    img, q, ref = torch.randn(1, 16, 16), torch.randn(1, 5, 16), torch.rand(1, 5, 2)
    torch.testing.assert_close(a(img, shapes, q, ref), b(img, shapes, q, ref))


# ---------------------------------------------------------------- module fixtures ----------------
@pytest.mark.parametrize("path", __import__("conftest").module_cases(), ids=__import__("conftest").case_id)
def test_host_module_matches_reference_module_fixtures(path):
    """The nn.Module on host tensors against the REFERENCE's MultiscaleDeformableAttention (frontend.py:175-292)
    run in the build container (tests/golden/make_golden.py): same state dict, same inputs -> out, the input
    gradients and all six parameter gradients.  2-d and 4-d reference points, non-square levels, both padding modes."""
    from conftest import load_module_case
    m, x, want = load_module_case(path)
    f64 = x["img"].dtype == torch.float64
    tol = dict(atol=1e-11, rtol=1e-9) if f64 else dict(atol=2e-5, rtol=1e-4)
    img, q, ref = (x[k].clone().requires_grad_(True) for k in ("img", "queries", "reference_points"))
    out = m(img, x["shapes"], q, ref)
    out.backward(x["grad_out"])
    torch.testing.assert_close(out.detach(), want["out"], **tol)
    torch.testing.assert_close(img.grad, want["grad_img"], **tol)
    torch.testing.assert_close(q.grad, want["grad_queries"], **tol)
    torch.testing.assert_close(ref.grad, want["grad_reference_points"], **tol)
    for name, prm in m.named_parameters():
        torch.testing.assert_close(prm.grad, want["grad__" + name], msg=lambda s, n=name: f"{n}: {s}", **tol)


def test_host_path_and_gather_formulation_agree():
    """Two statements of the operator on host tensors — per-level grid_sample + contraction (the product's host path)
    and index arithmetic + gather (no grid_sample) — agree in fp64 on out and all three gradients, all four modes,
    non-square levels, out-of-range points (location gradients compared away from the pixel-grid kinks)."""
    from conftest import MODES, kink_mask
    from msda_triton_amd.functional import _gather_multiscale_deformable_attention, native_multiscale_deformable_attention
    g = torch.Generator().manual_seed(99)
    levels = [(7, 5), (3, 4), (1, 2)]
    B, Q, H, D, P = 2, 19, 3, 6, 3
    shapes = torch.tensor(levels)
    value = torch.randn(B, sum(h * w for h, w in levels), H, D, generator=g, dtype=torch.float64)
    loc = torch.rand(B, Q, H, len(levels), P, 2, generator=g, dtype=torch.float64) * 1.6 - 0.3
    attn = torch.rand(B, Q, H, len(levels), P, generator=g, dtype=torch.float64)
    gout = torch.rand(B, Q, H, D, generator=g, dtype=torch.float64)
    for pm, ac in MODES:
        res = []
        for fn in (native_multiscale_deformable_attention, _gather_multiscale_deformable_attention):
            v, l, a = (t.clone().requires_grad_(True) for t in (value, loc, attn))
            out = fn(v, shapes, l, a, pm, ac)
            out.backward(gout)
            res.append((out.detach(), v.grad, l.grad, a.grad))
        keep = torch.from_numpy(~kink_mask(loc.numpy(), shapes.numpy(), ac, tol=1e-9))
        for name, x, y in zip(("out", "grad_value", "grad_loc", "grad_attn"), res[0], res[1]):
            if name == "grad_loc":
                x, y = x * keep, y * keep
            torch.testing.assert_close(x, y, atol=1e-11, rtol=1e-9, msg=lambda m, n=name: f"{pm} {ac} {n}: {m}")


def test_level_cells_of_host_shapes():
    """The bound msda_bwd_<dtype> takes as max_level_cells, from host numbers only (a device tensor would need a synchronisation)."""
    from msda_triton_amd.functional import level_cells_of
    assert level_cells_of(None) == 0
    assert level_cells_of([]) == 0
    assert level_cells_of([(100, 134), (50, 67)]) == 101 * 135
    assert level_cells_of(torch.tensor([[8, 8], [64, 3]])) == 65 * 4
    assert level_cells_of(((1, 1),)) == 4


def test_backward_support_query_and_forward_time_rejection():
    """shapes beyond the sorted pipeline's record format used to surface as MSDA_ERR_UNSUPPORTED from the
    BACKWARD, mid-training.  The library now answers at forward time (host arithmetic, no GPU): L = 17..32 is served
    (5 level bits), a plane whose partial rows would not fit 32-bit offsets is refused with a clear ValueError."""
    import torch
    from msda_triton_amd import _lib
    from msda_triton_amd.functional import check_backward_supported
    lib = _lib.load()
    assert lib.msda_bwd_supported(4, 5440, 8, 32, 10000, 4, 4, 4) == 1          # c2 @ 10k
    assert lib.msda_bwd_supported(1, 2000, 2, 8, 5000, 17, 4, 4) == 1           # 17 levels, Q*P > 4096: sorted pipeline
    assert lib.msda_bwd_supported(1, 2000, 2, 8, 5000, 32, 4, 4) == 1
    assert lib.msda_bwd_supported(1, 2000, 2, 8, 5000, 33, 4, 4) == 0           # > MSDA_MAX_LEVELS
    assert lib.msda_bwd_supported(1, 600000, 1, 256, 5000, 1, 4, 4) == 0        # I*4*D*4 bytes >= 2^31
    assert lib.msda_bwd_supported(1, 1 << 22, 1, 8, 5000, 1, 4, 4) == 0         # 2^22 pixels in one plane
    img = torch.empty(1, 600000, 1, 256, device="meta")
    pts = torch.empty(1, 5000, 1, 1, 4, 2, device="meta")
    with pytest.raises(ValueError, match="grad_value is not available"):
        check_backward_supported(img, pts)
    check_backward_supported(torch.empty(4, 5440, 8, 32, device="meta"), torch.empty(4, 10000, 8, 4, 4, 2, device="meta"))


def test_workspace_size_queries_are_consistent():
    """Host arithmetic only (no GPU): the lean backward workspace (MSDA_WS_RECORDS_IN_GRADS: sorted records inside the
    caller's gradient buffers) is never larger than the full one and, at c2 @ 10k, smaller by the whole record array;
    problems of the single-launch kernel need none; the level-size bound turns a decoder call over an image-sized
    pyramid into such a problem; several rounds over the queries (c5) keep grad_value out of the record homes."""
    from msda_triton_amd import _lib
    lib = _lib.load()
    flag = _lib.WS_RECORDS_IN_GRADS
    c2 = (4, 5440, 8, 32, 10000, 4, 4)
    full, lean = lib.msda_bwd_workspace_bytes(*c2, 4, 4, 0, 0), lib.msda_bwd_workspace_bytes(*c2, 4, 4, 0, flag)
    records = 4 * 8 * 10000 * 16 * 16  # planes x samples per plane x 16 bytes
    assert 0 < lean <= full - records and lean < 110e6
    for dims, es, ves in (((4, 5440, 8, 32, 5000, 4, 4), 4, 4), ((2, 17821, 8, 32, 17821, 4, 4), 2, 2),
                          ((2, 17821, 8, 32, 17821, 4, 4), 4, 2), ((4, 21824, 8, 64, 100000, 5, 8), 2, 2)):
        f, le = lib.msda_bwd_workspace_bytes(*dims, es, ves, 0, 0), lib.msda_bwd_workspace_bytes(*dims, es, ves, 0, flag)
        assert 0 < le < f, dims
    c5 = (4, 21824, 8, 64, 100000, 5, 8)  # two rounds: grad_loc / grad_attn hold records, grad_value does not
    per_round = 4 * 8 * 50000 * 40 * 16
    assert lib.msda_bwd_workspace_bytes(*c5, 2, 2, 0, 0) - lib.msda_bwd_workspace_bytes(*c5, 2, 2, 0, flag) < per_round
    assert lib.msda_bwd_workspace_bytes(2, 5440, 8, 32, 900, 4, 4, 4, 4, 0, flag) == 0          # c1: single launch
    dec = (8, 17821, 8, 32, 900, 4, 4)
    assert lib.msda_bwd_workspace_bytes(*dec, 4, 4, 0, 0) > 0
    assert lib.msda_bwd_workspace_bytes(*dec, 4, 4, 101 * 135, 0) == 0                           # with the bound


def test_every_documented_option_exists_with_its_documented_default():
    """include/msda_hip.h lists the msda_set_option keys as `"key"  <default> (default)`: each must be known to the library
    and read back that default in a fresh process (no GPU involved), and a key the header does not list must be refused."""
    import re
    import subprocess
    import sys
    text = open(os.path.join(ROOT, "include", "msda_hip.h")).read()
    documented = {k: int(v) for k, v in re.findall(r'^ \*   "(\w+)"\s+(-?\d+) \(default\)', text, flags=re.M)}
    assert {"xcd_map", "lds_levels", "unit_fwd", "unit_waves", "touch", "value_path", "strict", "profile"} <= set(documented), documented
    code = ("import json; from msda_triton_amd import _lib; "
            f"print(json.dumps({{k: _lib.get_option(k) for k in {sorted(documented)!r}}}))")
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("MSDA_TEST_OPTS", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    import json
    got = json.loads(out.strip().splitlines()[-1])
    assert got == documented, {k: (got[k], documented[k]) for k in documented if got[k] != documented[k]}


def test_no_kernel_spills_vector_registers_or_uses_scratch(tmp_path):
    """every kernel of every translation unit of the shipped library has .vgpr_spill_count 0 and no
    private (scratch) segment — read from the code objects' metadata notes (no GPU needed).  A gather kernel that spills in
    its hot loop pays a scratch round trip per trip; round 5 shipped 1-26 spilled VGPRs in the fp16 fused forward, the fp64
    scalar forward, the 16-bit forward with 8-byte pieces and the 16-bit single-launch grad_value kernel.  (SGPR spills go
    to VGPR lanes, not to memory, and are not counted.)"""
    import glob
    import re
    import shutil
    import subprocess
    from msda_triton_amd import _lib
    llvm = "/opt/rocm/lib/llvm/bin"
    objdump, readelf = os.path.join(llvm, "llvm-objdump"), os.path.join(llvm, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf of the ROCm toolchain not found")
    lib = tmp_path / "lib.so"
    shutil.copy(_lib.LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", str(lib)], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)  # writes one file per bundle next to the copy
    objects = glob.glob(str(tmp_path / "lib.so.*gfx950"))
    assert len(objects) >= 8, objects  # one code object per dtype translation unit
    kernels, bad = 0, []
    pat = re.compile(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", re.S)
    for obj in objects:
        notes = subprocess.run([readelf, "--notes", obj], capture_output=True, text=True, check=True).stdout
        for m in pat.finditer(notes):
            kernels += 1
            if int(m.group(2)) or int(m.group(3)):
                bad.append((m.group(1), "scratch bytes", int(m.group(2)), "vgpr spills", int(m.group(3))))
    assert kernels >= 500, kernels
    assert not bad, bad


def test_header_is_plain_c_and_a_c_caller_links(tmp_path):
    """include/msda_hip.h is the boundary a C / cgo / JNI caller binds: it must compile as plain C (no C++, no HIP or torch
    types), and a C translation unit calling one entry point of every family with the documented argument lists must
    compile against it and link against libmsda_hip.so (a prototype that drifts from the library is a link-time or
    compile-time failure here; nothing is run: no GPU needed)."""
    import shutil
    import subprocess
    from msda_triton_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include <stddef.h>
#include "msda_hip.h"
int main(int argc, char **argv)
{
    if (msda_abi_version() != MSDA_ABI_VERSION) return 2;   /* the first thing every caller does */
    if (argc < 1000) return 0;                              /* never reached in the test: the calls below are link checks */
    void *p = argv; const int64_t *s = (const int64_t *)argv; void *st = NULL;
    int64_t ws = msda_bwd_workspace_bytes(1, 4, 1, 4, 1, 1, 1, 4, 4, 0, MSDA_WS_RECORDS_IN_GRADS | MSDA_WS_PASSES(2));
    ws += msda_bwd_fused_workspace_bytes(1, 4, 1, 4, 1, 1, 1, 4, 4, 0, 0);
    int rc = msda_fwd_f32(p, s, p, p, p, 1, 4, 1, 4, 1, 1, 1, MSDA_PADDING_ZEROS, 0, /*value_row_stride*/ 0, st);
    rc |= msda_bwd_bf16(p, p, s, p, p, p, p, p, 1, 4, 1, 4, 1, 1, 1, MSDA_PADDING_BORDER, 1, /*max_level_cells*/ 0,
                        /*value_row_stride*/ 0, p, ws, st);
    rc |= msda_fwd_fused_f16(p, s, p, p, p, 1, 4, 1, 4, 1, 1, 1, /*ref_dim*/ 2, 0, 0, 0, st);
    rc |= msda_bwd_fused_f32_sbf16(p, p, s, p, p, p, p, p, 1, 4, 1, 4, 1, 1, 1, 4, 0, 0, 0, 0, p, ws, st);
    rc |= msda_fwd_f32_vbf16(p, s, p, p, p, 1, 4, 1, 4, 1, 1, 1, 0, 0, 0, st);
    rc |= msda_bwd_supported(1, 4, 1, 4, 1, 1, 1, 4) | (int)msda_fused_lp_limit(32, 4) | msda_set_option("xcd_map", 1) |
          msda_get_option("xcd_map") | msda_last_launch_info("fwd_variant") | msda_profile_read((char *)p, 0);
    return rc + (msda_last_error()[0] != 0);
}
''')
    inc = os.path.join(ROOT, "include")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c", os.path.join(inc, "msda_hip.h")],
                   check=True)
    exe = tmp_path / "caller"
    libdir = os.path.dirname(_lib.LIB_PATH)
    res = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", f"-I{inc}", str(src), "-o", str(exe), f"-L{libdir}", "-lmsda_hip",
                          f"-Wl,-rpath,{libdir}", "-Wl,--unresolved-symbols=ignore-in-shared-libs"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    run = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert run.returncode == 0, (run.returncode, run.stdout)  # the ABI check passed; nothing touched a device


def test_padded_value_rows_layout_rule_on_the_host():
    """functional.padded_value_rows / _value_rows (no GPU needed): which layouts the launchers hand to the kernels in place —
    dense, or a [B, I, H, D] view whose pixels sit a constant, 16-byte-multiple number of bytes apart — and which they copy
    dense first (everything else: the reference's `.contiguous()`, kernels.py:367-370)."""
    from msda_triton_amd import functional as F
    assert F.value_row_pad(1024) == 128 and F.value_row_pad(512) == 128 and F.value_row_pad(1152) == 0 and F.value_row_pad(96) == 0
    dense = torch.randn(2, 5, 8, 32)
    t, row = F._value_rows(dense)
    assert t is dense and row == 0
    p = F.padded_value_rows(2, 5, 8, 32, torch.float32, "cpu")
    assert tuple(p.shape) == (2, 5, 8, 32) and p.stride() == (5 * 288, 288, 32, 1)
    t, row = F._value_rows(p)
    assert t is p and row == 1152
    p16 = F.padded_value_rows(1, 5, 8, 32, torch.bfloat16, "cpu")          # 512-byte rows: one more line
    assert F._value_rows(p16)[1] == 640
    odd = F.padded_value_rows(2, 5, 8, 32, torch.float32, "cpu", pad_bytes=20)  # not a multiple of 16 bytes: copied dense
    t, row = F._value_rows(odd)
    assert t.is_contiguous() and row == 0
    for other in (dense.transpose(1, 2).contiguous().transpose(1, 2), dense[:, ::2], dense.permute(0, 1, 3, 2)):
        t, row = F._value_rows(other)
        assert t.is_contiguous() and row == 0 and torch.equal(t, other)
    with pytest.raises(ValueError, match="multiple of the element size"):
        F.padded_value_rows(1, 2, 2, 4, torch.float32, "cpu", pad_bytes=6)
    # the host path takes a padded pyramid like any other strided tensor
    shapes = torch.tensor([[1, 5]])
    pts, att = torch.rand(2, 3, 8, 1, 2, 2), torch.rand(2, 3, 8, 1, 2)
    p.copy_(dense)
    torch.testing.assert_close(F.multiscale_deformable_attention(p, shapes, pts, att, "zeros", False),
                               F.multiscale_deformable_attention(dense, shapes, pts, att, "zeros", False))
