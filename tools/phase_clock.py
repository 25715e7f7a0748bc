"""dev tool (needs a library built with -DMSDA_DEV, e.g. msda_triton_amd/libmsda_hip_dev.so copied over the shipped one):
per-phase cycles of the forward kernel's waves from the in-kernel s_memtime stamps (msda_set_option("debug", 2048)).
    python tools/phase_clock.py [workload] [k=v ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from msda_triton_amd import _lib, synth  # noqa: E402
from msda_triton_amd.functional import msda_hip_fwd  # noqa: E402

import dataclasses
wl = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 and "=" not in sys.argv[1] else "c2_q10k"]
if os.environ.get("Q"):
    wl = dataclasses.replace(wl, Q=int(os.environ["Q"]))  # (e.g. the decoder-sized calls of the query sweep)
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
d = synth.make_inputs_torch(wl, "cuda:0", seed=0) if hasattr(synth, "make_inputs_torch") else None
if d is None:
    h = synth.make_inputs_numpy(wl, seed=0)
    dt = getattr(torch, wl.dtype)
    d = {k: torch.from_numpy(v).to("cuda:0") if k == "shapes" else torch.from_numpy(v).to("cuda:0", dt) for k, v in h.items()}
shift = int(os.environ.get("SHIFT", "0"))  # floats: move `value` (and with SHIFT_ALL=1 every tensor) off its allocation's start
def shifted(t, by=None):
    by = shift if by is None else by
    if not by or t.dtype == torch.int64:
        return t
    buf = torch.empty(t.numel() + by, dtype=t.dtype, device=t.device)
    v = buf[by:].view(t.shape)
    v.copy_(t)
    return v
d["value"] = shifted(d["value"])
if os.environ.get("SHIFT_ALL"):
    d["loc"], d["attn"] = shifted(d["loc"]), shifted(d["attn"])
# LOC_SHIFT / ATTN_SHIFT / OUT_SHIFT (elements): move one of the STREAMED tensors instead — does the slow head follow it?
d["loc"] = shifted(d["loc"], int(os.environ.get("LOC_SHIFT", "0")))
d["attn"] = shifted(d["attn"], int(os.environ.get("ATTN_SHIFT", "0")))
out_buf = shifted(torch.empty(wl.B, wl.Q, wl.H, wl.D, dtype=d["value"].dtype, device="cuda:0"), int(os.environ.get("OUT_SHIFT", "0")))
print("ptr % 1024: value", d["value"].data_ptr() % 1024, " loc", d["loc"].data_ptr() % 1024, " attn", d["attn"].data_ptr() % 1024,
      " out", out_buf.data_ptr() % 1024)
_fwd = msda_hip_fwd
def msda_hip_fwd(*a, **k):  # noqa: E302  (every call below writes into the same, possibly shifted, result buffer)
    return _fwd(*a, out=out_buf, **k)
for _ in range(3):
    out = msda_hip_fwd(d["value"], d["shapes"], d["loc"], d["attn"], wl.padding_mode, wl.align_corners)
_lib.set_option("debug", 2048 | _lib.get_option("debug"))
if os.environ.get("COLD"):  # the reference benchmark's cold-cache recipe: L2 + Infinity Cache flushed before the launch
    for _ in range(3):
        out = msda_hip_fwd(d["value"], d["shapes"], d["loc"], d["attn"], wl.padding_mode, wl.align_corners)
    torch.empty(256 << 20, dtype=torch.int8, device="cuda").zero_()
out = msda_hip_fwd(d["value"], d["shapes"], d["loc"], d["attn"], wl.padding_mode, wl.align_corners)
torch.cuda.synchronize()
n8 = out.numel() // 8 * 8
r = out.float().reshape(-1)[: min(n8, 8 * 65536 * 4)].reshape(-1, 8).cpu().numpy()
r = r[r[:, 7] == 12345.0]
print("waves", len(r), "slices per wave", r[:, 4].mean())
names = ["phase1(+wait for points)", "memory gather", "LDS gather"]
tot = r[:, 5]
print("wave life cycles: mean %.0f  min %.0f  max %.0f" % (tot.mean(), tot.min(), tot.max()))
for i, n in enumerate(names):
    print("  %-26s mean cycles per wave %9.0f  (%.1f %% of life)  per slice %.0f" % (n, r[:, i].mean(), 100 * r[:, i].mean() / tot.mean(), (r[:, i] / np.maximum(r[:, 4], 1)).mean()))
wg = r[:, 6]
per_wg = np.array([tot[wg == g].max() for g in np.unique(wg)])
print("workgroup life (max over its waves): mean %.0f  min %.0f  max %.0f" % (per_wg.mean(), per_wg.min(), per_wg.max()))
ids = np.unique(wg).astype(int)
life = {g: tot[wg == g].max() for g in ids}
by_x, by_xcc = {}, {}
xcc = {g: int(r[wg == g][0, 3]) for g in ids}
for g in ids:
    by_x.setdefault(g % 8, []).append(life[g])
    by_xcc.setdefault(xcc[g], []).append(life[g])
print("by linear id % 8:", {int(k): int(np.mean(v)) for k, v in sorted(by_x.items())})
print("by XCC_ID       :", {int(k): (len(v), int(np.mean(v))) for k, v in sorted(by_xcc.items())})
order = sorted(ids, key=lambda g: life[g])
print("fastest:", [(int(g), int(life[g])) for g in order[:6]])
print("slowest:", [(int(g), int(life[g])) for g in order[-10:]])
ph = {g: r[wg == g][:, :3].mean(0) for g in order[:3] + order[-3:]}
for g, v in ph.items():
    print("  wg %4d phases per wave" % g, v.astype(int))
