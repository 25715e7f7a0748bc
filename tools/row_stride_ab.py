"""dev tool (GPU box): do PADDED value rows pay?  

The vector L1 picks one of its four tag RAMs from the low bits of the 128-byte line index; with H * D * sizeof = 1 024 the
rows of head 3 are the lines 8 p + 3 and use two of the four (HISTORY.md 4 item 5).  Rows 1 152 bytes apart (one extra line per
pixel) make every head's rows cycle through all residues mod 8.  The C ABI's `value_row_stride` argument (ABI 11) tells the
forward and sample-gradient kernels that the pixels' rows of `value` are that many bytes apart; this tool feeds them a
padded copy and alternates dense / padded inside one process, per-kernel device times from the library's own event pairs.
`--opt k=v` sets library options for the whole run (e.g. lds_planes=1: never two planes per workgroup).

  python tools/row_stride_ab.py [--workload c2_q10k] [--pads 0,128,256] [--rounds 3] [--fused]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2_q10k")
    ap.add_argument("--pads", default="0,128,256")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT")
    ap.add_argument("--no-spin", action="store_true", help="no spin-up, no profile option (runs under rocprofv3 --pmc)")
    args = ap.parse_args()

    import torch

    from msda_triton_amd import _lib, synth

    dev = torch.device("cuda", 0)
    wl = synth.WORKLOADS[args.workload]
    d = synth.make_inputs_torch(wl, dev, seed=0)
    value, shapes, loc, attn, go = d["value"], d["shapes"], d["loc"], d["attn"], d["grad_out"]
    B, I, H, D = value.shape
    Q, L, P = wl.Q, wl.L, wl.P
    es = value.element_size()
    suf = {torch.float32: "f32", torch.float16: "f16", torch.bfloat16: "bf16"}[value.dtype]
    lib = _lib.load()
    for kv in args.opt:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    fwd, bwd = getattr(lib, f"msda_fwd_{suf}"), getattr(lib, f"msda_bwd_{suf}")
    pm, ac = _lib.PADDING_MODES[wl.padding_mode], int(wl.align_corners)
    stream = torch.cuda.current_stream(dev).cuda_stream
    pads = [int(x) for x in args.pads.split(",")]

    def padded(pad):
        if pad == 0:
            return value
        assert pad % es == 0
        buf = torch.zeros(B, I, H * D + pad // es, dtype=value.dtype, device=dev)
        buf[:, :, :H * D] = value.reshape(B, I, H * D)
        return buf

    vals = {pad: padded(pad) for pad in pads}
    out = {pad: torch.empty(B, Q, H, D, dtype=loc.dtype, device=dev) for pad in pads}
    g_loc = {pad: torch.empty_like(loc) for pad in pads}
    g_att = {pad: torch.empty_like(attn) for pad in pads}

    def run(pad):
        vrow = 0 if pad == 0 else H * D * es + pad
        v = vals[pad]
        rc = fwd(v.data_ptr(), shapes.data_ptr(), loc.data_ptr(), attn.data_ptr(), out[pad].data_ptr(), B, I, H, D, Q, L, P,
                 pm, ac, vrow, stream)
        _lib.check(rc, "fwd")
        rc = bwd(go.data_ptr(), v.data_ptr(), shapes.data_ptr(), loc.data_ptr(), attn.data_ptr(), None,
                 g_loc[pad].data_ptr(), g_att[pad].data_ptr(), B, I, H, D, Q, L, P, pm, ac, 0, vrow, None, 0, stream)
        _lib.check(rc, "bwd")

    for pad in pads:
        run(pad)
    torch.cuda.synchronize()
    for pad in pads[1:]:
        same = all(torch.equal(a[pad], a[pads[0]]) for a in (out, g_loc, g_att))
        print(f"pad {pad}: bit-identical to pad {pads[0]}: {same}")
        assert same
    if args.no_spin:
        for _ in range(args.reps):
            for pad in pads:
                run(pad)
        torch.cuda.synchronize()
        return
    # spin-up
    for _ in range(300):
        run(pads[0])
    torch.cuda.synchronize()
    _lib.set_option("profile", 1)
    res = {pad: [] for pad in pads}
    for _ in range(args.rounds):
        for pad in pads:
            for _ in range(10):
                run(pad)
            torch.cuda.synchronize()
            _lib.profile_read()
            for _ in range(args.reps):
                run(pad)
            torch.cuda.synchronize()
            res[pad].append({k: round(v[1], 2) for k, v in _lib.profile_read().items()})
    _lib.set_option("profile", 0)
    for pad in pads:
        print(f"{args.workload} pad {pad:4d} (rows {H * D * es + pad} B apart):", json.dumps(res[pad]))


if __name__ == "__main__":
    main()
