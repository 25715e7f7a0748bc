"""dev tool: the weight / bias gradients of the module's three Linear layers at the c2 module shape (tall-skinny K:
dW[out, in] = dY[N, out]^T X[N, in] with N = 21 760 / 40 000, out, in <= 384) — the GEMM the BLAS library picks
(16 workgroups of 64 x 64 tiles, no split-K) against a batched split over the rows.   python tools/linear_wgrad_bench.py"""
import sys

import torch

dev = torch.device("cuda", 0)
if "--tunable" in sys.argv:  # PyTorch's TunableOp: every rocBLAS / hipBLASLt solution is timed once per shape, the best is used
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_max_tuning_duration(300)


def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def splitk(gy, x, C, out_dtype=None):
    N = x.shape[0]
    S = N // C
    main = S * C
    part = torch.bmm(gy[:main].view(S, C, -1).transpose(1, 2), x[:main].view(S, C, -1)) if out_dtype is None else \
        torch.bmm(gy[:main].view(S, C, -1).transpose(1, 2), x[:main].view(S, C, -1), out_dtype=out_dtype)
    w = part.sum(0, dtype=torch.float32)
    if main < N:
        w = w + (gy[main:].t() @ x[main:]).float()
    return w


for dt in (torch.bfloat16, torch.float32):
    for N, i, o in ((21760, 256, 256), (40000, 256, 384), (40000, 256, 256)):
        x = torch.randn(N, i, device=dev, dtype=dt)
        gy = torch.randn(N, o, device=dev, dtype=dt)
        ref = (gy.double().t() @ x.double())
        line = ["%s N=%d in=%d out=%d" % (str(dt)[6:], N, i, o), "plain %.1f us" % timeit(lambda: gy.t() @ x)]
        for C in (256, 512, 1024, 2048):
            try:
                t = timeit(lambda: splitk(gy, x, C))
                err = ((splitk(gy, x, C).double() - ref).abs().max() / ref.abs().max()).item()
                line.append("C=%d %.1f us (err %.1e)" % (C, t, err))
            except Exception as e:  # noqa: BLE001
                line.append("C=%d failed %s" % (C, type(e).__name__))
        if dt != torch.float32:
            try:
                t = timeit(lambda: splitk(gy, x, 1024, torch.float32))
                line.append("C=1024 fp32-out %.1f us" % t)
            except Exception as e:  # noqa: BLE001
                line.append("fp32-out: %s" % type(e).__name__)
        perr = ((gy.t() @ x).double() - ref).abs().max() / ref.abs().max()
        line.append("plain err %.1e" % perr.item())
        print("  ".join(line), flush=True)
        # bias gradient
        b1 = timeit(lambda: gy.sum(0))
        b2 = timeit(lambda: gy.view(-1, 64, o).sum(1).sum(0)) if N % 64 == 0 else float("nan")
        b2b = timeit(lambda: gy.view(-1, 320 if N % 320 == 0 else 64, o).sum(1, dtype=torch.float32).sum(0))
        ones = torch.ones(1, N, device=dev, dtype=dt)
        b3 = timeit(lambda: ones @ gy)
        print("   bias: sum(0) %.1f us   two-stage %.1f / %.1f us   ones @ gy %.1f us" % (b1, b2, b2b, b3), flush=True)
