"""dev tool: the fused sort kernel next to the forked sample kernel, through the C ABI with a workspace the probe can
read back (control block = first words).   python tools/sort_probe.py [workload] [calls]"""
import sys, time
import torch
sys.path.insert(0, ".")
from msda_triton_amd import _lib, synth

wl = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2_q10k"]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
d = synth.make_inputs_torch(wl, "cuda", seed=0)
lib = _lib.load()
B, I, H, D, Q, L, P = wl.B, wl.I, wl.H, wl.D, wl.Q, wl.L, wl.P
suf = {"float32": "f32", "float16": "f16", "bfloat16": "bf16"}[wl.dtype]
fn = getattr(lib, f"msda_bwd_{suf}")
gv, gl, ga = torch.empty_like(d["value"]), torch.empty_like(d["loc"]), torch.empty_like(d["attn"])
pad = _lib.PADDING_MODES[wl.padding_mode]
for sort_path, overlap in ((0, 0), (0, 1)):
    _lib.set_option("sort_path", sort_path)
    _lib.set_option("overlap", overlap)
    nbytes = int(lib.msda_bwd_workspace_bytes(B, I, H, D, Q, L, P, d["loc"].element_size()))
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    slow = 0
    ts = []
    for i in range(calls):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = fn(d["grad_out"].data_ptr(), d["value"].data_ptr(), d["shapes"].data_ptr(), d["loc"].data_ptr(), d["attn"].data_ptr(),
                gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), B, I, H, D, Q, L, P, pad, int(wl.align_corners),
                ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        ts.append(dt)
        if dt > 20:
            slow += 1
            ctl = ws[:64].view(torch.int32).tolist()
            print(f"  call {i}: {dt:.1f} ms; ctl [ticket, errs, item, pair, l, k, trip, arrived, want, ticket_then, grid] = {ctl[:11]}", flush=True)
    ts.sort()
    print(f"sort_path={sort_path} overlap={overlap}: {calls} calls, median {ts[len(ts)//2]:.3f} ms, slow calls {slow}", flush=True)
