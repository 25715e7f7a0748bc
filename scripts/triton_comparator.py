"""A same-GPU Triton comparator for the MSDA operator — MEASUREMENT ONLY, never on the product path.

SURVEY.md §8d: the reference's Triton kernels cannot travel to the GPU box, so "≥ 2x the reference Triton kernel on one
MI355X" has no direct measurement; it allows "a Triton kernel the build authors itself — say which".  This is that
kernel: written here from the operator's definition (out = sum over levels and points of attention weight x bilinear
sample), with the reference's PARALLELISATION so the comparison means something:

  * one Triton program per (query, batch element, head), grid [Q, B, H]       (as the reference, kernels.py:365)
  * all L * P samples of the program as one block axis, D channels as the other
  * backward: the forward is recomputed, grad_value is accumulated with tl.atomic_add into a zeroed buffer
    (the reference's choice, kernels.py:543-553) — the thing the HIP path replaces with a sort + segmented gather
  * num_warps autotuned over the reference's set {2, 4, 8, 16} (+1), keyed on the static shape

It is NOT the reference's code (block pointers, [L, P, C] tiles and helper split are the reference's; none of that is
here) and it is not tuned beyond num_warps — the same effort level as the reference.  What it proves: how a
straightforward Triton formulation of this operator runs on THIS GPU with THIS Triton (3.6, ROCm).  Numbers from it are
labelled "builder-authored Triton comparator" wherever they appear.

Semantics covered: padding_mode "border" / "zeros", align_corners True / False, fp32 (fp16 / bf16 inputs are computed
in fp32).  Finite sampling locations only — no NaN / inf handling, it is a timing comparator.  The parity test
tests/test_triton_comparator.py holds it to the HIP operator on the bench inputs.
"""
from __future__ import annotations

import torch

try:
    import triton
    import triton.language as tl
    HAVE_TRITON = True
except Exception:  # noqa: BLE001  (no Triton in this interpreter: the bench leg reports that and moves on)
    HAVE_TRITON = False


if HAVE_TRITON:
    _CONFIGS = [triton.Config({}, num_warps=w) for w in (1, 2, 4, 8, 16)]
    _KEY = ["H", "D", "L", "P", "ZEROS", "ALIGN"]
    _KEY_BWD = _KEY + ["SEM"]
    ATOMIC_SEM = "acq_rel"  # tl.atomic_add's default, what the reference's call gets (kernels.py:550-553); "relaxed"
                            # is the variant a maintainer could switch to: compare() times both

    @triton.jit
    def _sample_geometry(shapes_ptr, loc_ptr, sample0, s, live, L: tl.constexpr, P: tl.constexpr,
                         ZEROS: tl.constexpr, ALIGN: tl.constexpr):
        """Per sample of the block axis: level size, first pixel of the level, the four corner pixel offsets inside the
        level, their validity, the fractional position and d(pixel coordinate)/d(location)."""
        lvl = s // P
        hh = tl.zeros_like(s)
        ww = tl.zeros_like(s)
        first = tl.zeros_like(s)
        run = 0
        for l in tl.static_range(L):
            hl = tl.load(shapes_ptr + 2 * l).to(tl.int32)
            wl = tl.load(shapes_ptr + 2 * l + 1).to(tl.int32)
            here = lvl == l
            hh = tl.where(here, hl, hh)
            ww = tl.where(here, wl, ww)
            first = tl.where(here, run, first)
            run += hl * wl
        lx = tl.load(loc_ptr + (sample0 + s) * 2, mask=live, other=0.0).to(tl.float32)
        ly = tl.load(loc_ptr + (sample0 + s) * 2 + 1, mask=live, other=0.0).to(tl.float32)
        wf = ww.to(tl.float32)
        hf = hh.to(tl.float32)
        if ALIGN:
            sx = wf - 1.0
            sy = hf - 1.0
            px = lx * sx
            py = ly * sy
        else:
            sx = wf
            sy = hf
            px = lx * sx - 0.5
            py = ly * sy - 0.5
        if not ZEROS:  # border: the coordinate is clamped into the image, and a clamped coordinate has no gradient
            inx = (px > 0.0) & (px < wf - 1.0)
            iny = (py > 0.0) & (py < hf - 1.0)
            px = tl.minimum(tl.maximum(px, 0.0), wf - 1.0)
            py = tl.minimum(tl.maximum(py, 0.0), hf - 1.0)
            sx = tl.where(inx, sx, 0.0)
            sy = tl.where(iny, sy, 0.0)
        fx = tl.floor(px)
        fy = tl.floor(py)
        dx = px - fx
        dy = py - fy
        x0 = fx.to(tl.int32)
        y0 = fy.to(tl.int32)
        x1 = x0 + 1
        y1 = y0 + 1
        okx0 = (x0 >= 0) & (x0 < ww)
        okx1 = (x1 >= 0) & (x1 < ww)
        oky0 = (y0 >= 0) & (y0 < hh)
        oky1 = (y1 >= 0) & (y1 < hh)
        p00 = first + y0 * ww + x0
        p01 = first + y0 * ww + x1
        p10 = first + y1 * ww + x0
        p11 = first + y1 * ww + x1
        return (p00, p01, p10, p11, live & okx0 & oky0, live & okx1 & oky0, live & okx0 & oky1, live & okx1 & oky1,
                dx, dy, sx, sy)

    @triton.autotune(configs=_CONFIGS, key=_KEY)
    @triton.jit
    def _msda_fwd(value_ptr, shapes_ptr, loc_ptr, attn_ptr, out_ptr, I, Q,
                  H: tl.constexpr, D: tl.constexpr, L: tl.constexpr, P: tl.constexpr,
                  S2: tl.constexpr, D2: tl.constexpr, ZEROS: tl.constexpr, ALIGN: tl.constexpr):
        q = tl.program_id(0)
        b = tl.program_id(1)
        h = tl.program_id(2)
        s = tl.arange(0, S2)
        live = s < L * P
        c = tl.arange(0, D2)
        cm = c < D
        row = (b * Q + q) * H + h
        sample0 = row.to(tl.int64) * (L * P)
        p00, p01, p10, p11, m00, m01, m10, m11, dx, dy, sx, sy = _sample_geometry(
            shapes_ptr, loc_ptr, sample0, s, live, L, P, ZEROS, ALIGN)
        a = tl.load(attn_ptr + sample0 + s, mask=live, other=0.0).to(tl.float32)
        plane = value_ptr + (b.to(tl.int64) * I * H + h) * D
        step = H * D
        v00 = tl.load(plane + p00.to(tl.int64)[:, None] * step + c[None, :], mask=m00[:, None] & cm[None, :], other=0.0)
        v01 = tl.load(plane + p01.to(tl.int64)[:, None] * step + c[None, :], mask=m01[:, None] & cm[None, :], other=0.0)
        v10 = tl.load(plane + p10.to(tl.int64)[:, None] * step + c[None, :], mask=m10[:, None] & cm[None, :], other=0.0)
        v11 = tl.load(plane + p11.to(tl.int64)[:, None] * step + c[None, :], mask=m11[:, None] & cm[None, :], other=0.0)
        w00 = a * (1.0 - dx) * (1.0 - dy)
        w01 = a * dx * (1.0 - dy)
        w10 = a * (1.0 - dx) * dy
        w11 = a * dx * dy
        acc = (w00[:, None] * v00.to(tl.float32) + w01[:, None] * v01.to(tl.float32)
               + w10[:, None] * v10.to(tl.float32) + w11[:, None] * v11.to(tl.float32))
        res = tl.sum(acc, axis=0)
        tl.store(out_ptr + row.to(tl.int64) * D + c, res.to(out_ptr.dtype.element_ty), mask=cm)

    @triton.autotune(configs=_CONFIGS, key=_KEY_BWD, reset_to_zero=["gvalue_ptr"])
    @triton.jit
    def _msda_bwd(value_ptr, shapes_ptr, loc_ptr, attn_ptr, gout_ptr, gvalue_ptr, gloc_ptr, gattn_ptr, I, Q,
                  H: tl.constexpr, D: tl.constexpr, L: tl.constexpr, P: tl.constexpr,
                  S2: tl.constexpr, D2: tl.constexpr, ZEROS: tl.constexpr, ALIGN: tl.constexpr,
                  SEM: tl.constexpr):
        q = tl.program_id(0)
        b = tl.program_id(1)
        h = tl.program_id(2)
        s = tl.arange(0, S2)
        live = s < L * P
        c = tl.arange(0, D2)
        cm = c < D
        row = (b * Q + q) * H + h
        sample0 = row.to(tl.int64) * (L * P)
        p00, p01, p10, p11, m00, m01, m10, m11, dx, dy, sx, sy = _sample_geometry(
            shapes_ptr, loc_ptr, sample0, s, live, L, P, ZEROS, ALIGN)
        a = tl.load(attn_ptr + sample0 + s, mask=live, other=0.0).to(tl.float32)
        go = tl.load(gout_ptr + row.to(tl.int64) * D + c, mask=cm, other=0.0).to(tl.float32)
        plane_off = (b.to(tl.int64) * I * H + h) * D
        step = H * D
        o00 = plane_off + p00.to(tl.int64)[:, None] * step + c[None, :]
        o01 = plane_off + p01.to(tl.int64)[:, None] * step + c[None, :]
        o10 = plane_off + p10.to(tl.int64)[:, None] * step + c[None, :]
        o11 = plane_off + p11.to(tl.int64)[:, None] * step + c[None, :]
        k00 = m00[:, None] & cm[None, :]
        k01 = m01[:, None] & cm[None, :]
        k10 = m10[:, None] & cm[None, :]
        k11 = m11[:, None] & cm[None, :]
        # each corner's dot product with grad_out: everything the two small gradients need
        t00 = tl.sum(tl.load(value_ptr + o00, mask=k00, other=0.0).to(tl.float32) * go[None, :], axis=1)
        t01 = tl.sum(tl.load(value_ptr + o01, mask=k01, other=0.0).to(tl.float32) * go[None, :], axis=1)
        t10 = tl.sum(tl.load(value_ptr + o10, mask=k10, other=0.0).to(tl.float32) * go[None, :], axis=1)
        t11 = tl.sum(tl.load(value_ptr + o11, mask=k11, other=0.0).to(tl.float32) * go[None, :], axis=1)
        ex = 1.0 - dx
        ey = 1.0 - dy
        g_a = ex * ey * t00 + dx * ey * t01 + ex * dy * t10 + dx * dy * t11
        g_x = a * sx * (ey * (t01 - t00) + dy * (t11 - t10))
        g_y = a * sy * (ex * (t10 - t00) + dx * (t11 - t01))
        tl.store(gattn_ptr + sample0 + s, g_a.to(gattn_ptr.dtype.element_ty), mask=live)
        tl.store(gloc_ptr + (sample0 + s) * 2, g_x.to(gloc_ptr.dtype.element_ty), mask=live)
        tl.store(gloc_ptr + (sample0 + s) * 2 + 1, g_y.to(gloc_ptr.dtype.element_ty), mask=live)
        # grad_value: four corner tiles of atomics per program
        tl.atomic_add(gvalue_ptr + o00, (a * ex * ey)[:, None] * go[None, :], mask=k00, sem=SEM)
        tl.atomic_add(gvalue_ptr + o01, (a * dx * ey)[:, None] * go[None, :], mask=k01, sem=SEM)
        tl.atomic_add(gvalue_ptr + o10, (a * ex * dy)[:, None] * go[None, :], mask=k10, sem=SEM)
        tl.atomic_add(gvalue_ptr + o11, (a * dx * dy)[:, None] * go[None, :], mask=k11, sem=SEM)


def _dims(value, loc):
    B, I, H, D = value.shape
    _, Q, _, L, P, _ = loc.shape
    return B, I, H, D, Q, L, P


def triton_msda_fwd(value, shapes, loc, attn, padding_mode="border", align_corners=True):
    B, I, H, D, Q, L, P = _dims(value, loc)
    value, loc, attn, shapes = value.contiguous(), loc.contiguous(), attn.contiguous(), shapes.contiguous()
    out = torch.empty((B, Q, H, D), device=value.device, dtype=value.dtype)
    _msda_fwd[(Q, B, H)](value, shapes, loc, attn, out, I, Q, H=H, D=D, L=L, P=P,
                        S2=triton.next_power_of_2(L * P), D2=triton.next_power_of_2(D),
                        ZEROS=padding_mode == "zeros", ALIGN=bool(align_corners))
    return out


def triton_msda_bwd(grad_out, value, shapes, loc, attn, padding_mode="border", align_corners=True):
    B, I, H, D, Q, L, P = _dims(value, loc)
    value, loc, attn, shapes = value.contiguous(), loc.contiguous(), attn.contiguous(), shapes.contiguous()
    grad_out = grad_out.contiguous()
    g_value = torch.zeros(value.shape, device=value.device, dtype=torch.float32)  # atomics accumulate in fp32
    g_loc = torch.empty_like(loc)
    g_attn = torch.empty_like(attn)
    _msda_bwd[(Q, B, H)](value, shapes, loc, attn, grad_out, g_value, g_loc, g_attn, I, Q, H=H, D=D, L=L, P=P,
                        S2=triton.next_power_of_2(L * P), D2=triton.next_power_of_2(D),
                        ZEROS=padding_mode == "zeros", ALIGN=bool(align_corners), SEM=ATOMIC_SEM)
    return g_value.to(value.dtype), g_loc, g_attn


class _TritonComparatorFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, shapes, loc, attn, padding_mode, align_corners):
        ctx.save_for_backward(value, shapes, loc, attn)
        ctx.mode = (padding_mode, align_corners)
        return triton_msda_fwd(value, shapes, loc, attn, padding_mode, align_corners)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        value, shapes, loc, attn = ctx.saved_tensors
        g_value, g_loc, g_attn = triton_msda_bwd(grad_out, value, shapes, loc, attn, *ctx.mode)
        return g_value, None, g_loc, g_attn, None, None


def triton_comparator_msda(value, shapes, loc, attn, padding_mode="border", align_corners=True):
    """The comparator with autograd: same signature as ``msda_triton_amd.multiscale_deformable_attention``."""
    if not HAVE_TRITON:
        raise RuntimeError("Triton is not importable in this interpreter")
    return _TritonComparatorFunction.apply(value, shapes, loc, attn, padding_mode, align_corners)


def tuned_num_warps():
    """The autotuner's pick per kernel (after at least one call): {"fwd": n, "bwd": n}."""
    out = {}
    for name, k in (("fwd", _msda_fwd), ("bwd", _msda_bwd)):
        best = getattr(k, "best_config", None)
        out[name] = getattr(best, "num_warps", None)
    return out


def main():
    import argparse
    import json
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from msda_triton_amd import synth
    from msda_triton_amd.functional import multiscale_deformable_attention as hip_msda

    ap = argparse.ArgumentParser(description="time the Triton comparator next to the HIP operator")
    ap.add_argument("--workloads", default="c1_readme,c2_q1k,c2_q5k,c2_q10k,c4_gdino_dec")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda")
    rows = {}
    for name in args.workloads.split(","):
        rows[name] = compare(name, dev, args.steps)
        print(name, json.dumps(rows[name]), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"what": __doc__.split("\n\n")[0], "triton": triton.__version__, "torch": torch.__version__,
                       "device": torch.cuda.get_device_name(0), "rows": rows}, f, indent=1)


def compare(wl_name, dev, steps=20, warmup=5):
    """fwd and fwd+bwd ms of the comparator and of the HIP operator on one workload (fp32 inputs: the comparator's
    atomics are fp32; reduced-precision workloads are timed in fp32 for both)."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from msda_triton_amd import synth
    from msda_triton_amd.functional import multiscale_deformable_attention as hip_msda

    wl = synth.WORKLOADS[wl_name] if isinstance(wl_name, str) else wl_name  # (a synth.Workload works too)
    d = synth.make_inputs_torch(wl, dev, dtype=torch.float32)
    go = d.pop("grad_out")
    args = (d["shapes"], d["loc"], d["attn"], wl.padding_mode, wl.align_corners)
    leaves = [d["value"].requires_grad_(True), d["loc"].requires_grad_(True), d["attn"].requires_grad_(True)]

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / steps

    res = {}
    outs = {}
    for label, op in (("triton", triton_comparator_msda), ("hip", hip_msda)):
        def fwd():
            with torch.no_grad():
                return op(d["value"], *args)

        def step():
            for t in leaves:
                t.grad = None
            op(d["value"], *args).backward(go)

        res[label] = {"fwd_ms": timed(fwd), "fwd_bwd_ms": timed(step)}
        step()
        outs[label] = [fwd()] + [t.grad.clone() for t in leaves]
    # the same comparator with relaxed atomics (the default orders every atomic against the program's other accesses)
    global ATOMIC_SEM
    ATOMIC_SEM = "relaxed"
    try:
        def step_relaxed():
            for t in leaves:
                t.grad = None
            triton_comparator_msda(d["value"], *args).backward(go)
        res["triton_relaxed_atomics"] = {"fwd_bwd_ms": timed(step_relaxed)}
    finally:
        ATOMIC_SEM = "acq_rel"
    names = ("out", "grad_value", "grad_loc", "grad_attn")
    res["max_abs_diff"] = {n: float((a - b).abs().max()) for n, a, b in zip(names, outs["triton"], outs["hip"])}
    res["hip_speedup"] = {k: res["triton"][k] / res["hip"][k] for k in ("fwd_ms", "fwd_bwd_ms")}
    res["hip_speedup"]["fwd_bwd_ms_vs_relaxed_atomics"] = res["triton_relaxed_atomics"]["fwd_bwd_ms"] / res["hip"]["fwd_bwd_ms"]
    res["triton_num_warps"] = tuned_num_warps()
    return res


if __name__ == "__main__":
    main()
