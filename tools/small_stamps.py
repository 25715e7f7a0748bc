"""dev tool: where wave 0 of every msda_value_small_kernel workgroup spends its gather (temporary stamp build)."""
import ctypes, sys
import numpy as np, torch
from msda_triton_amd import _lib, synth
from msda_triton_amd.functional import multiscale_deformable_attention as msda
wl = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c4_gdino_dec"]
for o in sys.argv[2:]:
    k, v = o.split("="); _lib.set_option(k, int(v))
d = synth.make_inputs_torch(wl, "cuda")
go = d.pop("grad_out")
v = d["value"].requires_grad_(True)
for _ in range(5):
    v.grad = None
    out = msda(v, d["shapes"], d["loc"], d["attn"], wl.padding_mode, wl.align_corners)
    out.backward(go)
    torch.cuda.synchronize()
lib = _lib.load()
n = 8192 * 16
buf = np.zeros(n, dtype=np.uint64)
rc = lib.msda_dbg_stamps(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(n * 8))
assert rc == 0, rc
s = buf.reshape(-1, 16)
s = s[s[:, 6] == 1]
info = s[:, 5]
lvl = info >> np.uint64(32); rounds = info & np.uint64(0xffff); S = (info >> np.uint64(16)) & np.uint64(0xffff)
for l in sorted(set(lvl.tolist())):
    m = lvl == l
    x = s[m].astype(np.int64)
    print(f"lvl {l}: n={m.sum()} rounds={int(rounds[m][0])} S={int(S[m][0])} gather total med {np.median(x[:,0]):.0f} max {x[:,0].max()} | lists {np.median(x[:,1]):.0f} convert {np.median(x[:,2]):.0f} "
          f"loads+fma {np.median(x[:,3]):.0f} reduce+store {np.median(x[:,4]):.0f}  (max WG: lists {x[np.argmax(x[:,0]),1]} convert {x[np.argmax(x[:,0]),2]} loads+fma {x[np.argmax(x[:,0]),3]} store {x[np.argmax(x[:,0]),4]})")
idx = np.nonzero(buf.reshape(-1, 16)[:, 6] == 1)[0]
m = lvl == 0
order = np.argsort(s[m][:, 0])
print("level-0 WGs: (wgid, xcd = wgid % 8, gather cycles)")
print(" ".join(f"({int(idx[m][i])},{int(idx[m][i]) % 8},{int(s[m][i, 0])})" for i in order))
