"""dev tool: one hipGraph holding forward + backward (the launcher API, static buffers) replayed back to back, next to
the eager autograd step and torch.cuda.make_graphed_callables, on the small workloads."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msda_triton_amd import synth, multiscale_deformable_attention
from msda_triton_amd.functional import msda_hip_fwd, msda_hip_bwd
dev = "cuda:0"
for name in sys.argv[1:] or ("c1_readme", "c2_q1k", "c4_gdino_dec"):
    wl = synth.WORKLOADS[name]
    d = synth.make_inputs_torch(wl, dev, seed=0)
    v, l, a, g, s = d["value"], d["loc"], d["attn"], d["grad_out"], d["shapes"]
    pm, ac = wl.padding_mode, wl.align_corners
    for _ in range(3):
        msda_hip_fwd(v, s, l, a, pm, ac); msda_hip_bwd(g, v, s, l, a, pm, ac)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        msda_hip_fwd(v, s, l, a, pm, ac); msda_hip_bwd(g, v, s, l, a, pm, ac)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = msda_hip_fwd(v, s, l, a, pm, ac)
        grads = msda_hip_bwd(g, v, s, l, a, pm, ac)
    def timed(fn, n=500):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    t_replay = timed(graph.replay)
    t_launchers = timed(lambda: (msda_hip_fwd(v, s, l, a, pm, ac), msda_hip_bwd(g, v, s, l, a, pm, ac)))
    vr, lr, ar = (t.clone().requires_grad_(True) for t in (v, l, a))
    def step():
        multiscale_deformable_attention(vr, s, lr, ar, pm, ac).backward(g)
        vr.grad = lr.grad = ar.grad = None
    t_eager = timed(step)
    print(f"{name}: eager autograd {t_eager:.4f} ms | launcher API fwd+bwd {t_launchers:.4f} ms | one hipGraph replay {t_replay:.4f} ms")
