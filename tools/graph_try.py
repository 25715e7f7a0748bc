import sys, time, torch
sys.path.insert(0, ".")
from msda_triton_amd import synth, multiscale_deformable_attention
dev = "cuda:0"
for name in ("c1_readme", "c4_gdino_dec", "c2_q1k"):
    wl = synth.WORKLOADS[name]
    d = synth.make_inputs_torch(wl, dev, seed=0)
    v, l, a = (d[k].requires_grad_(True) for k in ("value", "loc", "attn"))
    s = d["shapes"]
    g = d["grad_out"]
    def fn(v, l, a):
        return multiscale_deformable_attention(v, s, l, a, wl.padding_mode, wl.align_corners)
    def step(f):
        out = f(v, l, a)
        out.backward(g)
        v.grad = l.grad = a.grad = None
    def bench(f, n=200):
        for _ in range(20): step(f)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): step(f)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    t_eager = bench(fn)
    gfn = torch.cuda.make_graphed_callables(fn, (v, l, a))
    t_graph = bench(gfn)
    out_e = fn(v, l, a); out_g = gfn(v, l, a)
    print(name, f"eager {t_eager:.3f} ms  graphed {t_graph:.3f} ms  maxdiff {(out_e-out_g).abs().max().item():.1e}")
