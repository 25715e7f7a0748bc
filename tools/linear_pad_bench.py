"""dev tool (GPU box): what writing the value projection with padded rows costs the GEMM (round 6):
  dense        F.linear(x, W, b)                                  -> [N, 256]
  out_strided  torch.addmm(b, x, W.t(), out=buf[:, :256])         -> rows 288 elements apart (leading dimension 288)
  wide_weight  F.linear(x, cat(W, 0)[288, E], cat(b, 0))[:, :256] -> a dense [N, 288] product, 32 dead columns
  copy         F.linear dense, then buf[:, :256].copy_(y)
for N = B * I rows of the c2 pyramid (21 760) and a 4x larger one, fp32 and bf16.  HIP-event times, alternated."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

dev = torch.device("cuda", 0)
for dt in (torch.float32, torch.bfloat16):
    for N in (21760, 87040):
        E, O, PAD = 256, 256, 128 // torch.empty((), dtype=dt).element_size()
        x = torch.randn(N, E, device=dev, dtype=dt)
        W = torch.randn(O, E, device=dev, dtype=dt) * 0.05
        b = torch.randn(O, device=dev, dtype=dt)
        buf = torch.empty(N, O + PAD, device=dev, dtype=dt)
        Ww = torch.cat([W, torch.zeros(PAD, E, device=dev, dtype=dt)])
        bw = torch.cat([b, torch.zeros(PAD, device=dev, dtype=dt)])

        def dense():
            return F.linear(x, W, b)

        def out_strided():
            return torch.addmm(b, x, W.t(), out=buf[:, :O])

        def wide_weight():
            return F.linear(x, Ww, bw)[:, :O]

        def wide_weight_cat():
            return F.linear(x, torch.cat([W, torch.zeros(PAD, E, device=dev, dtype=dt)]),
                            torch.cat([b, torch.zeros(PAD, device=dev, dtype=dt)]))[:, :O]

        def copy():
            buf[:, :O].copy_(F.linear(x, W, b))
            return buf[:, :O]

        fns = dict(dense=dense, out_strided=out_strided, wide_weight=wide_weight, wide_weight_cat=wide_weight_cat, copy=copy)
        ref = dense()
        for name, fn in fns.items():
            torch.testing.assert_close(fn().float(), ref.float(), rtol=2e-2, atol=2e-2)
        res = {k: [] for k in fns}
        for _ in range(3):
            for name, fn in fns.items():
                for _ in range(20):
                    fn()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(50):
                    fn()
                e.record()
                torch.cuda.synchronize()
                res[name].append(round(s.elapsed_time(e) / 50 * 1e3, 1))
        print(str(dt).replace("torch.", ""), "N =", N, {k: v for k, v in res.items()}, flush=True)
