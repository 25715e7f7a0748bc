"""MultiscaleDeformableAttention module at the c2 shape (B=4, Q=10 000, emb = hidden = 256, H=8, L=4, P=4) with and
without `value_dtype=torch.bfloat16`: forward and forward+backward ms, peak memory, and the attention
core alone (operator level: fp32 vs bf16 pyramid with fp32 sampling inputs).  Writes one JSON document.
Usage: python tools/module_core_bench.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msda_triton_amd import MultiscaleDeformableAttention, synth
from msda_triton_amd.functional import multiscale_deformable_attention

dev = torch.device("cuda", 0)
wl = synth.WORKLOADS["c2_q10k"]
shapes = torch.tensor(wl.levels, device=dev)
EMB = wl.H * wl.D


def timed(fn, steps=30, warmup=8):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


def module_case(value_dtype, autocast):
    torch.manual_seed(0)
    m = MultiscaleDeformableAttention(EMB, EMB, wl.L, wl.H, wl.P, wl.padding_mode, wl.align_corners,
                                      value_dtype=value_dtype).to(dev)
    img = torch.randn(wl.B, wl.I, EMB, device=dev, requires_grad=True)
    q = torch.randn(wl.B, wl.Q, EMB, device=dev, requires_grad=True)
    ref = torch.rand(wl.B, wl.Q, 2, device=dev)

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            m(img, shapes, q, ref)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            out = m(img, shapes, q, ref)
        out.float().sum().backward()
        m.zero_grad(set_to_none=True)
        img.grad = q.grad = None

    res = {"fwd_ms": timed(fwd), "fwd_bwd_ms": timed(step)}
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    step()
    torch.cuda.synchronize()
    res["peak_extra_MB"] = (torch.cuda.max_memory_allocated() - base) / 1e6
    return res


def operator_case(value_dtype):
    d = synth.make_inputs_torch(wl, dev, seed=0)
    v = d["value"].to(value_dtype).requires_grad_(True)
    pts, att = d["loc"].requires_grad_(True), d["attn"].requires_grad_(True)

    def fwd():
        with torch.no_grad():
            multiscale_deformable_attention(v, d["shapes"], pts, att, wl.padding_mode, wl.align_corners)

    go = torch.rand(wl.B, wl.Q, wl.H, wl.D, device=dev)

    def step():
        multiscale_deformable_attention(v, d["shapes"], pts, att, wl.padding_mode, wl.align_corners).backward(go)
        v.grad = pts.grad = att.grad = None

    return {"fwd_ms": timed(fwd, 50, 10), "fwd_bwd_ms": timed(step, 50, 10),
            "value_MB": v.numel() * v.element_size() / 1e6}


out = {"workload": f"c2_q10k: B={wl.B} Q={wl.Q} I={wl.I} H={wl.H} D={wl.D} L={wl.L} P={wl.P}, emb=hidden={EMB}",
       "operator": {"fp32": operator_case(torch.float32), "bf16_value_fp32_sampling": operator_case(torch.bfloat16),
                    "fp16_value_fp32_sampling": operator_case(torch.float16)},
       "module": {}}
for ac in (False, True):
    for vd in (None, torch.bfloat16):
        key = f"{'autocast_bf16' if ac else 'fp32_params'}/{'value_dtype=bf16' if vd else 'default'}"
        out["module"][key] = module_case(vd, ac)
text = json.dumps(out, indent=1)
print(text)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write(text + "\n")
