"""dev tool: phase clock of the tile-gather kernel (msda_value_binned.hpp, debug bit 256).
Calls msda_bwd_f32 directly with its own workspace and prints, per phase, the mean / max over workgroups of the
s_memtime cycles spent (100 MHz ticks -> us)."""
import sys
import numpy as np
import torch
from msda_triton_amd import _lib, synth

wl_name = sys.argv[1] if len(sys.argv) > 1 else "c2_q10k"
opts = [kv.split("=") for kv in sys.argv[2:]]
wl = synth.WORKLOADS[wl_name]
dev = torch.device("cuda:0")
d = synth.make_inputs_torch(wl, dev, seed=0)
lib = _lib.load()
_lib.set_option("value_path", 4)
for k, v in opts:
    _lib.set_option(k, int(v))
B, I, H, D = d["value"].shape
Q, L, P = d["loc"].shape[1], d["loc"].shape[3], d["loc"].shape[4]
es = d["value"].element_size()
ws_bytes = int(lib.msda_bwd_workspace_bytes(B, I, H, D, Q, L, P, es))
ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
go = torch.rand(B, Q, H, D, device=dev, dtype=d["value"].dtype)
gv = torch.empty_like(d["value"])
suf = {torch.float32: "f32", torch.float16: "f16", torch.bfloat16: "bf16", torch.float64: "f64"}[d["value"].dtype]
fn = getattr(lib, f"msda_bwd_{suf}")
pad = _lib.PADDING_MODES[wl.padding_mode]


def run():
    rc = fn(go.data_ptr(), d["value"].data_ptr(), d["shapes"].data_ptr(), d["loc"].data_ptr(), d["attn"].data_ptr(),
            gv.data_ptr(), None, None, B, I, H, D, Q, L, P, pad, int(wl.align_corners), ws.data_ptr(), ws_bytes,
            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "msda_bwd")


for _ in range(3):
    run()
torch.cuda.synchronize()
_lib.set_option("debug", _lib.get_option("debug") | 256)
run()
torch.cuda.synchronize()
st = ws[-65536:].view(torch.int64).cpu().numpy().reshape(-1, 8).astype(np.float64)
st = st[st.sum(1) > 0]
names = ["tables", "first item setup", "records+hist", "scan", "place+issue", "gather", "assembly", "rows out"]
print(f"{wl_name}: {len(st)} workgroups with stamps; s_memtime ticks at 100 MHz -> us")
for k, nme in enumerate(names):
    print(f"  {nme:18s} mean {st[:, k].mean() / 100:8.2f} us   max {st[:, k].max() / 100:8.2f} us")
print(f"  {'total':18s} mean {st.sum(1).mean() / 100:8.2f} us   max {st.sum(1).max() / 100:8.2f} us")
