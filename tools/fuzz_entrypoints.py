#!/usr/bin/env python3
"""Long differential fuzz of the OTHER entry points against the plain operator (which tools/fuzz_parity.py holds to the
oracle):
  fused   fused_module_core (softmax / sampling-point prologue inside the kernels, fwd + bwd incl. the reference
          points' gradient) vs module_sampling_inputs + multiscale_deformable_attention, fp64 at 1e-9 / fp32 at 1e-4
  mixed   a bf16 / fp16 value pyramid next to fp32 sampling inputs vs the fp32 operator on the rounded pyramid
  half    the whole operator in fp16 / bf16 vs the fp32 operator on the rounded inputs (loose: 2e-2 relative)
  storage the module kernels with 16-bit storage (value, projection, result and their gradients in bf16 / fp16, fp32
          reference points: msda_*_fused_f32_sbf16 / _sf16) vs the fp32 fused kernels on the rounded inputs
Usage: fuzz_entrypoints.py [seconds] [first_seed]; one line per failure, a summary, exit status 1 on failure."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from msda_triton_amd import _lib
from msda_triton_amd.functional import (fused_module_core, module_sampling_inputs, multiscale_deformable_attention,
                                        padded_value_rows)


def rows(t, pad_bytes):
    """`t` [B, I, H, D] handed over in padded rows (round 6: value_row_stride) — or as it is (pad_bytes 0)"""
    if not pad_bytes:
        return t
    p = padded_value_rows(*t.shape, t.dtype, t.device, pad_bytes)
    p.copy_(t)
    return p


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda")
MODES = [("border", True), ("border", False), ("zeros", True), ("zeros", False)]


def close(a, b, atol, rtol, what, frac_ok=1.0):
    d = (a.double() - b.double()).abs()
    ok = d <= atol + rtol * b.double().abs()
    nbad = int((~ok).sum().item())
    if nbad > (1.0 - frac_ok) * ok.numel():
        raise AssertionError(f"{what}: {nbad} of {ok.numel()} entries out of tolerance, max abs diff {d.max().item():.3e}, "
                             f"NaNs {int(torch.isnan(a).sum())} / {int(torch.isnan(b).sum())}, tolerance {atol:.2e} + {rtol:.1e} |ref|")


t0 = time.time()
n = fails = 0
seen = {}
seed = seed0
while time.time() - t0 < budget:
    rng = np.random.default_rng(91000 + seed)
    g = torch.Generator(device="cpu").manual_seed(91000 + seed)
    kind = ("fused", "mixed", "half", "storage")[seed % 4]
    B, H = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    D = int(rng.choice([8, 16, 24, 32, 40, 64]))
    L, P = int(rng.integers(1, 5)), int(rng.integers(1, 6))
    Q = int(rng.choice([1, 5, 33, 130, 400, 1200, 2500]))
    big = int(rng.choice([5, 12, 30]))
    levels = [(int(rng.integers(1, big + 1)), int(rng.integers(1, big + 1))) for _ in range(L)]
    I = sum(h * w for h, w in levels)
    pm, ac = MODES[int(rng.integers(0, 4))]
    shapes = torch.tensor(levels, dtype=torch.int64, device=dev)
    opts = {"value_path": int(rng.choice([0, 2, 3])), "small_ns": int(rng.choice([0, 0, 2, 3])),
            "lds_levels": int(rng.choice([1, 1, 2, 0])), "unit_fwd": int(rng.choice([1, 1, 0, 2])),
            "lds_planes": int(rng.choice([0, 2, 2, 1])), "linear_slots": int(rng.choice([320, 1, 6, 40])), "touch": int(rng.choice([1, 0, 2])), "unit_waves": int(rng.choice([1, 2])),
            "ws_passes": int(rng.choice([1, 1, 2, 3]))}
    pad = int(rng.choice([0, 0, 64, 128, 48]))  # bytes behind every pixel's rows of the TESTED route's value pyramid
    desc = dict(seed=seed, kind=kind, B=B, Q=Q, H=H, D=D, levels=levels, P=P, pm=pm, ac=ac, value_pad=pad, **opts)
    try:
        for k, v in opts.items():
            _lib.set_option(k, v)
        if kind == "fused":
            dt = torch.float64 if rng.integers(0, 2) else torch.float32
            coords = int(rng.choice([2, 4]))
            value = torch.randn(B, I, H, D, generator=g, dtype=dt)
            proj = torch.randn(B, Q, H, L, P, 3, generator=g, dtype=dt) * float(rng.choice([0.5, 1.5, 4.0]))
            ref = torch.rand(B, Q, coords, generator=g, dtype=dt)
            gout = torch.rand(B, Q, H, D, generator=g, dtype=dt)
            desc.update(dtype=str(dt), coords=coords)
            res = []
            for fused in (True, False):
                v, pr, rf = (t.clone().to(dev) for t in (value, proj, ref))
                v = rows(v, pad if fused else 0)
                v.requires_grad_(True), pr.requires_grad_(True), rf.requires_grad_(True)
                if fused:
                    out = fused_module_core(v, shapes, pr, rf, pm, ac)
                else:
                    pts, att = module_sampling_inputs(pr, shapes, rf)
                    out = multiscale_deformable_attention(v, shapes, pts, att, pm, ac)
                out.backward(gout.to(dev))
                res.append((out.detach(), v.grad, pr.grad, rf.grad))
            atol, rtol = (1e-9, 1e-8) if dt == torch.float64 else (2e-4, 2e-3)
            for name, a, b in zip(("out", "grad_value", "grad_proj", "grad_ref"), res[0], res[1]):
                # fp32: a sample whose pixel coordinate rounds across a grid line differently in the two routes has a
                # different grad_proj / grad_ref there (the kink the other fp32 tests mask): allow 1 % of entries
                close(a, b, atol, rtol, name, frac_ok=1.0 if dt == torch.float64 or name in ("out", "grad_value") else 0.99)
        elif kind == "storage":
            sdt = torch.bfloat16 if rng.integers(0, 2) else torch.float16
            coords = int(rng.choice([2, 4]))
            desc.update(storage=str(sdt), coords=coords)
            value = torch.randn(B, I, H, D, generator=g).to(sdt)
            proj = (torch.randn(B, Q, H, L, P, 3, generator=g) * float(rng.choice([0.5, 1.5, 4.0]))).to(sdt)
            ref = torch.rand(B, Q, coords, generator=g)
            gout = torch.rand(B, Q, H, D, generator=g).to(sdt)
            res = []
            for low in (True, False):
                v, pr, go = ((t if low else t.float()).clone().to(dev) for t in (value, proj, gout))
                v = rows(v, pad if low else 0)
                v.requires_grad_(True), pr.requires_grad_(True)
                rf = ref.clone().to(dev).requires_grad_(True)
                out = fused_module_core(v, shapes, pr, rf, pm, ac)
                assert out.dtype == (sdt if low else torch.float32), out.dtype
                out.backward(go)
                res.append((out.detach().float(), v.grad.float(), pr.grad.float(), rf.grad.float()))
            eps = 8e-3 if sdt == torch.bfloat16 else 1e-3
            for name, a, b in zip(("out", "grad_value", "grad_proj", "grad_ref"), res[0], res[1]):
                scale = float(b.abs().max()) + 1e-6
                if name == "grad_ref":  # fp32 sums of the same numbers
                    close(a, b, 2e-4 * scale, 2e-3, name, frac_ok=0.99)
                else:  # the fp32 result rounded to the storage type
                    # (+ half of fp16's subnormal step: a gradient that is zero up to rounding noise rounds onto that grid)
                    close(a, b, 2 * eps * scale * (2.0 ** -3) + 3.0e-8, 2 * eps, name, frac_ok=0.999)
        else:
            sdt = torch.bfloat16 if rng.integers(0, 2) else torch.float16
            desc.update(storage=str(sdt))
            value = torch.randn(B, I, H, D, generator=g).to(sdt)
            loc = torch.rand(B, Q, H, L, P, 2, generator=g) * 1.3 - 0.15
            attn = torch.softmax(torch.randn(B, Q, H, L * P, generator=g), -1).view(B, Q, H, L, P)
            gout = torch.rand(B, Q, H, D, generator=g)
            if kind == "half":
                loc, attn, gout = loc.to(sdt), attn.to(sdt), gout.to(sdt)
            res = []
            for low in (True, False):
                v = rows((value if low else value.float()).clone().to(dev), pad if low else 0).requires_grad_(True)
                lo = (loc if low else loc.float()).clone().to(dev).requires_grad_(True)
                at = (attn if low else attn.float()).clone().to(dev).requires_grad_(True)
                out = multiscale_deformable_attention(v, shapes, lo, at, pm, ac)
                out.backward((gout if low else gout.float()).to(dev))
                res.append((out.detach().float(), v.grad.float(), lo.grad.float(), at.grad.float()))
            eps = 8e-3 if sdt == torch.bfloat16 else 1e-3
            for name, a, b in zip(("out", "grad_value", "grad_loc", "grad_attn"), res[0], res[1]):
                scale = float(b.abs().max()) + 1e-6
                if kind == "mixed" and name != "grad_value":  # fp32 arithmetic on identical numbers
                    close(a, b, 2e-4 * scale, 2e-3, name, frac_ok=0.99 if name == "grad_loc" else 1.0)
                else:  # results rounded to the 16-bit type (and, for "half", 16-bit coordinates: kinks move)
                    close(a, b, 4 * eps * scale, 4 * eps, name, frac_ok=0.97 if name == "grad_loc" else 0.999)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("FAIL", json.dumps(desc), "::", str(e).strip().splitlines()[0][:300], flush=True)
    finally:
        for k in opts:
            _lib.set_option(k, {"lds_levels": 1, "unit_fwd": 1, "linear_slots": 320, "touch": 1, "unit_waves": 1, "ws_passes": 1}.get(k, 0))
    n += 1
    seen[kind] = seen.get(kind, 0) + 1
    seed += 1
    if n % 100 == 0:
        print(f"... {n} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_entrypoints: {n} cases (seeds {seed0}..{seed - 1}), {fails} failures, kinds {seen}, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
