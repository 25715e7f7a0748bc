#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE implementation (build container only).

Imports /root/reference/src/msda_triton/frontend.py:native_multiscale_deformable_attention
(the reference's CPU path, frontend.py:15-68) and records, for a matrix of small cases
and every (padding_mode, align_corners) mode: the inputs, the forward output, a grad_out
and the three gradients PyTorch autograd produces through the reference function.
Also records digests (sum / abs-sum / 512 strided samples) of the reference's outputs on the
BASELINE.json-sized workloads (c1, c2 at Q = 1k / 5k / 10k, c4, and the c3 encoder shape in fp32)
whose inputs come from msda_triton_amd.synth.

The reference never travels to the GPU box; only the .npz files written here do.

    cd /root/repo && python tests/golden/make_golden.py [small] [digests]     (default: both)
"""
import importlib.metadata as _md
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference/src")
sys.dont_write_bytecode = True

# the reference package is not pip-installed: its __init__ asks importlib.metadata for a version
_orig_version = _md.version
_md.version = lambda n: "0.1.1" if n == "msda_triton" else _orig_version(n)
import msda_triton.frontend as _ref_frontend  # noqa: E402

assert _ref_frontend.__file__.startswith("/root/reference/"), _ref_frontend.__file__
reference_native = _ref_frontend.native_multiscale_deformable_attention
ReferenceModule = _ref_frontend.MultiscaleDeformableAttention

from msda_triton_amd import synth  # noqa: E402

MODES = [("zeros", False), ("zeros", True), ("border", False), ("border", True)]


def mode_key(pm, ac):
    return f"{pm}_{int(ac)}"


def run_reference(value, shapes, loc, attn, grad_out, pm, ac):
    v = value.clone().requires_grad_(True)
    s = loc.clone().requires_grad_(True)
    a = attn.clone().requires_grad_(True)
    out = reference_native(v, shapes, s, a, pm, ac)
    out.backward(grad_out)
    return out.detach(), v.grad, s.grad, a.grad


def small_cases():
    """name -> (levels, B, Q, H, D, P, point generator)"""
    rng = np.random.default_rng(20240531)

    def pts_uniform(lo, hi):
        return lambda shape: rng.uniform(lo, hi, size=shape)

    def pts_far(shape):
        p = rng.uniform(-3.0, 4.0, size=shape)
        flat = p.reshape(-1)
        flat[::7] = rng.choice([-1e6, 1e6, -17.25, 33.5], size=flat[::7].shape)
        return p

    def pts_exact(levels):
        def gen(shape):
            # B,Q,H,L,P,2 — coordinates that land exactly on pixel centres / image corners
            p = np.empty(shape)
            for l, (h, w) in enumerate(levels):
                kx = rng.integers(0, max(w, 1), size=shape[:3] + shape[4:5])
                ky = rng.integers(0, max(h, 1), size=shape[:3] + shape[4:5])
                p[:, :, :, l, :, 0] = kx / max(w - 1, 1)
                p[:, :, :, l, :, 1] = ky / max(h - 1, 1)
            flat = p.reshape(-1, 2)
            flat[::5] = rng.choice([0.0, 1.0, 0.5], size=flat[::5].shape)
            return p
        return gen

    lv_nonsq = [(6, 4), (3, 2), (2, 5)]
    lv_sq = [(8, 8), (4, 4)]
    lv_degen = [(1, 1), (1, 5), (7, 1), (2, 2)]
    lv_pyr = [(8, 8), (4, 4), (2, 2), (1, 1)]
    lv_c5 = [(8, 8), (4, 4), (2, 2), (1, 1), (3, 5)]
    return {
        "nonpow2_oob": (lv_nonsq, 2, 5, 2, 5, 3, pts_uniform(-0.5, 1.5)),
        "square_inrange": (lv_sq, 1, 8, 2, 8, 4, pts_uniform(0.0, 1.0)),
        "far_oob": (lv_nonsq, 1, 6, 3, 4, 2, pts_far),
        "exact_grid": (lv_nonsq, 1, 6, 2, 6, 3, pts_exact(lv_nonsq)),
        "degenerate_levels": (lv_degen, 2, 4, 1, 3, 2, pts_uniform(-0.25, 1.25)),
        "d32_pyramid": (lv_pyr, 1, 12, 2, 32, 4, pts_uniform(-0.1, 1.1)),
        "d64_l5_p8": (lv_c5, 1, 3, 2, 64, 8, pts_uniform(-0.1, 1.1)),
        "single_everything": ([(3, 3)], 1, 1, 1, 1, 1, pts_uniform(0.0, 1.0)),
    }, rng


def write_small_cases():
    cases, rng = small_cases()
    for name, (levels, B, Q, H, D, P, gen) in cases.items():
        L = len(levels)
        I = sum(h * w for h, w in levels)  # noqa: E741
        value64 = rng.standard_normal((B, I, H, D))
        loc64 = gen((B, Q, H, L, P, 2))
        attn64 = rng.uniform(0.0, 1.0, size=(B, Q, H, L, P))
        gout64 = rng.uniform(0.0, 1.0, size=(B, Q, H, D))
        shapes = torch.tensor(levels, dtype=torch.int64)
        for dt_name, dt in (("f32", torch.float32), ("f64", torch.float64)):
            value, loc, attn, gout = (torch.from_numpy(a).to(dt) for a in (value64, loc64, attn64, gout64))
            rec = {"value": value.numpy(), "shapes": shapes.numpy(), "loc": loc.numpy(),
                   "attn": attn.numpy(), "grad_out": gout.numpy()}
            for pm, ac in MODES:
                out, gv, gl, ga = run_reference(value, shapes, loc, attn, gout, pm, ac)
                k = mode_key(pm, ac)
                rec[f"out_{k}"] = out.numpy()
                rec[f"grad_value_{k}"] = gv.numpy()
                rec[f"grad_loc_{k}"] = gl.numpy()
                rec[f"grad_attn_{k}"] = ga.numpy()
            path = os.path.join(HERE, f"{name}_{dt_name}.npz")
            np.savez_compressed(path, **rec)
            print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# (workload, sampling-point range): c3 runs in fp32 here (the reference's CPU path is numerically broken in
# bf16 for levels > 2048 px); the bf16 GPU tests compare against these digests with bf16 tolerances.
DIGEST_CASES = (("c1_readme", (0.0, 1.0)), ("c2_q1k", (0.0, 1.0)), ("c1_readme", (-0.25, 1.25)),
                ("c2_q5k", (0.0, 1.0)), ("c2_q10k", (0.0, 1.0)), ("c4_gdino_dec", (0.0, 1.0)),
                ("c4_gdino_dec", (-0.25, 1.25)), ("c3_ddetr_enc", (0.0, 1.0)),
                # encoder-like locality: pixel-centre reference points + N(0, 2 px) offsets (synth loc_mode)
                ("c3_ddetr_enc_local", (0.0, 1.0)))


def write_digests(only=None):
    # full-size digests: inputs are regenerated from synth on the checking side
    for wl_name, pts_range in DIGEST_CASES:
        if only and wl_name not in only:
            continue
        wl = synth.WORKLOADS[wl_name]
        d = synth.make_inputs_torch(wl, "cpu", seed=0, dtype=torch.float32,
                                    loc_lo=pts_range[0], loc_hi=pts_range[1])
        rec = {"workload": wl_name, "seed": 0, "loc_lo": pts_range[0], "loc_hi": pts_range[1]}
        for pm, ac in MODES:
            out, gv, gl, ga = run_reference(d["value"], d["shapes"], d["loc"], d["attn"], d["grad_out"], pm, ac)
            k = mode_key(pm, ac)
            for nm, t in (("out", out), ("grad_value", gv), ("grad_loc", gl), ("grad_attn", ga)):
                dg = synth.digest(t.numpy())
                rec[f"{nm}_{k}_sum"] = dg["sum"]
                rec[f"{nm}_{k}_abs_sum"] = dg["abs_sum"]
                rec[f"{nm}_{k}_samples"] = dg["samples"].astype(np.float32)
                rec[f"{nm}_{k}_step"] = dg["step"]
        tag = "oob" if pts_range[0] < 0 else "in"
        path = os.path.join(HERE, f"digest_{wl_name}_{tag}_f32.npz")
        np.savez_compressed(path, **rec)
        print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# name -> (emb, hidden, levels, heads, points, B, Q, ref_dim, padding_mode, align_corners)
MODULE_CASES = {
    "ref2_nonsquare_zeros": (12, 16, [(5, 3), (2, 4), (1, 2)], 2, 3, 2, 7, 2, "zeros", False),
    "ref4_nonsquare_border": (12, 16, [(5, 3), (2, 4), (1, 2)], 2, 3, 2, 7, 4, "border", True),
    "ref2_square_border_false": (16, 32, [(8, 8), (4, 4)], 4, 2, 1, 9, 2, "border", False),
    "ref4_square_zeros_true": (16, 32, [(8, 8), (4, 4)], 4, 2, 1, 9, 4, "zeros", True),
    "ref2_d32_vector_path": (16, 64, [(6, 7), (3, 4), (2, 2)], 2, 4, 1, 5, 2, "zeros", False),
}
MODULE_PARAMS = ("img_input_proj.weight", "img_input_proj.bias", "query_input_proj.weight", "query_input_proj.bias",
                 "query_output_proj.weight", "query_output_proj.bias")


def write_module_cases():
    rng = np.random.default_rng(20250118)
    for name, (emb, hidden, levels, H, P, B, Q, ref_dim, pm, ac) in MODULE_CASES.items():
        L = len(levels)
        I = sum(h * w for h, w in levels)  # noqa: E741
        shapes = torch.tensor(levels, dtype=torch.int64)
        # parameters and inputs are drawn here (float64), so the fixture does not depend on torch's init RNG
        state64 = {
            "img_input_proj.weight": rng.standard_normal((hidden, emb)) / np.sqrt(emb),
            "img_input_proj.bias": 0.1 * rng.standard_normal(hidden),
            "query_input_proj.weight": 0.5 * rng.standard_normal((H * L * P * 3, emb)) / np.sqrt(emb),
            "query_input_proj.bias": 0.5 * rng.standard_normal(H * L * P * 3),
            "query_output_proj.weight": rng.standard_normal((emb, hidden)) / np.sqrt(hidden),
            "query_output_proj.bias": 0.1 * rng.standard_normal(emb),
        }
        img64 = rng.standard_normal((B, I, emb))
        q64 = rng.standard_normal((B, Q, emb))
        ref64 = rng.uniform(0.1, 0.9, size=(B, Q, ref_dim))
        if ref_dim == 4:
            ref64[..., 2:] = rng.uniform(0.05, 0.4, size=(B, Q, 2))  # box sizes
        gout64 = rng.uniform(0.0, 1.0, size=(B, Q, emb))
        for dt_name, dt in (("f32", torch.float32), ("f64", torch.float64)):
            m = ReferenceModule(emb, hidden, L, H, P, pm, ac).to(dt)
            m.load_state_dict({k: torch.from_numpy(v).to(dt) for k, v in state64.items()})
            img = torch.from_numpy(img64).to(dt).requires_grad_(True)
            q = torch.from_numpy(q64).to(dt).requires_grad_(True)
            ref = torch.from_numpy(ref64).to(dt).requires_grad_(True)
            gout = torch.from_numpy(gout64).to(dt)
            out = m(img, shapes, q, ref)  # CPU tensors: the reference's native path (frontend.py:170-172)
            out.backward(gout)
            rec = {"shapes": shapes.numpy(), "img": img.detach().numpy(), "queries": q.detach().numpy(),
                   "reference_points": ref.detach().numpy(), "grad_out": gout.numpy(), "out": out.detach().numpy(),
                   "grad_img": img.grad.numpy(), "grad_queries": q.grad.numpy(), "grad_reference_points": ref.grad.numpy(),
                   "meta": np.array([emb, hidden, L, H, P, B, Q, ref_dim, int(pm == "zeros"), int(ac)], dtype=np.int64)}
            params = dict(m.named_parameters())
            for k in MODULE_PARAMS:
                rec["param__" + k] = params[k].detach().numpy()
                rec["grad__" + k] = params[k].grad.numpy()
            path = os.path.join(HERE, f"module_{name}_{dt_name}.npz")
            np.savez_compressed(path, **rec)
            print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    what = set(sys.argv[1:]) or {"small", "digests", "module"}
    if "small" in what:
        write_small_cases()
    if "digests" in what:
        write_digests()
    only = {w.split(":", 1)[1] for w in what if w.startswith("digest:")}  # e.g. digest:c3_ddetr_enc_local
    if only:
        write_digests(only)
    if "module" in what:
        write_module_cases()


if __name__ == "__main__":
    main()
