"""Synthetic workloads for the BASELINE.json configs.

The generator is counter based (splitmix64 over the flat element index), so the
same (seed, shape) yields bit-identical inputs in the build container, on the
GPU box and in every rank of a sharded run without shipping any data.  The
distributions follow the reference benchmark (scripts/benchmark.py:33-36,93 in
/root/reference): value ~ N(0,1), sampling points ~ U[0,1), attention weights
= softmax(N(0,1)) over the last dim, grad_out ~ U[0,1).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Sequence

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n float64 values in [0,1), element i depends only on (seed, stream, offset+i)."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        bits = _splitmix64(idx ^ key)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n float64 N(0,1) values (Box-Muller over two uniform streams)."""
    u1 = uniform(seed, 2 * stream + 1000, n, offset)
    u2 = uniform(seed, 2 * stream + 1001, n, offset)
    return np.sqrt(-2.0 * np.log1p(-u1)) * np.cos(2.0 * np.pi * u2)


@dataclass(frozen=True)
class Workload:
    """One named configuration of the operator (shapes only, no data)."""

    name: str
    B: int
    Q: int
    H: int
    D: int
    levels: Sequence[tuple]  # (height, width) per level
    P: int
    dtype: str = "float32"
    padding_mode: str = "border"
    align_corners: bool = True
    # how sampling points are drawn: "uniform" = U[0, 1) (the reference benchmark, scripts/benchmark.py:34), "encoder"
    # = what a Deformable-DETR encoder layer produces (reference module, frontend.py:271-276: reference point + small
    # offset): query q sits on a pixel of the pyramid, its reference point is that pixel's centre, and every sample
    # is the reference point plus an N(0, 2 px) offset in the sampled level's pixels
    loc_mode: str = "uniform"
    L: int = field(init=False)
    I: int = field(init=False)  # noqa: E741

    def __post_init__(self):
        object.__setattr__(self, "L", len(self.levels))
        object.__setattr__(self, "I", int(sum(h * w for h, w in self.levels)))

    # ---- algorithmic (compulsory) bytes, BASELINE.md section 3 ----
    @property
    def elem_size(self) -> int:
        return {"float16": 2, "bfloat16": 2, "float32": 4, "float64": 8}[self.dtype]

    @property
    def alg_fwd_bytes(self) -> int:
        s = self.elem_size
        B, I, H, D, Q, L, P = self.B, self.I, self.H, self.D, self.Q, self.L, self.P
        return s * (B * I * H * D + 3 * B * Q * H * L * P + B * Q * H * D) + 16 * L

    @property
    def alg_bwd_bytes(self) -> int:
        s = self.elem_size
        B, I, H, D, Q, L, P = self.B, self.I, self.H, self.D, self.Q, self.L, self.P
        return s * (2 * B * I * H * D + 6 * B * Q * H * L * P + B * Q * H * D)

    @property
    def gather_fwd_bytes(self) -> int:
        return self.elem_size * 4 * self.D * self.B * self.Q * self.H * self.L * self.P


_PYR64 = ((64, 64), (32, 32), (16, 16), (8, 8))

WORKLOADS = {
    # BASELINE.json configs[0]: README synthetic (README.md:121-147 of the reference)
    "c1_readme": Workload("c1_readme", 2, 900, 8, 32, _PYR64, 4, "float32", "zeros", False),
    # configs[1]: scripts/benchmark.py:25-31 sweep
    "c2_q1k": Workload("c2_q1k", 4, 1000, 8, 32, _PYR64, 4, "float32", "border", True),
    "c2_q5k": Workload("c2_q5k", 4, 5000, 8, 32, _PYR64, 4, "float32", "border", True),
    "c2_q10k": Workload("c2_q10k", 4, 10000, 8, 32, _PYR64, 4, "float32", "border", True),
    # ... configs[1] also in the other mode SURVEY.md 8d lists for c2 (the reference benches border / True only)
    "c2_q10k_zeros": Workload("c2_q10k_zeros", 4, 10000, 8, 32, _PYR64, 4, "float32", "zeros", False),
    # configs[2]: Deformable-DETR encoder shape
    "c3_ddetr_enc": Workload("c3_ddetr_enc", 2, 17821, 8, 32,
                             ((100, 134), (50, 67), (25, 34), (13, 17)), 4, "bfloat16", "zeros", False),
    # ... the same shape with encoder-like locality of the sampling points ("pixel-centre reference +
    # N(0, 2 px) offsets"): neighbouring queries hit the same few cells
    "c3_ddetr_enc_local": Workload("c3_ddetr_enc_local", 2, 17821, 8, 32,
                                   ((100, 134), (50, 67), (25, 34), (13, 17)), 4, "bfloat16", "zeros", False, "encoder"),
    # configs[3]: Grounding-DINO decoder shape
    "c4_gdino_dec": Workload("c4_gdino_dec", 8, 900, 8, 32, _PYR64, 4, "float32", "zeros", False),
    # configs[4]: stress
    "c5_stress": Workload("c5_stress", 4, 100000, 8, 64,
                          ((128, 128), (64, 64), (32, 32), (16, 16), (8, 8)), 8, "float16", "zeros", False),
    # not a BASELINE config: the c4 decoder shape on the pyramid of an 800 x 1066 image (c3's levels) — what a
    # Grounding-DINO / Deformable-DETR decoder layer sees at COCO size (tools/, profiles/: "dec_coco")
    "dec_coco": Workload("dec_coco", 8, 900, 8, 32, ((100, 134), (50, 67), (25, 34), (13, 17)), 4, "float32", "zeros", False),
    # not BASELINE configs: tiny shapes for `bench.py --backend gloo --device cpu` (control-flow rehearsal of the
    # multi-GPU bench on a box without GPUs; the numbers mean nothing)
    "dryrun": Workload("dryrun", 2, 48, 2, 8, ((6, 5), (3, 3)), 2, "float32", "border", True),
    "dryrun_strong": Workload("dryrun_strong", 2, 96, 2, 8, ((6, 5), (3, 3), (2, 2)), 2, "float32", "zeros", False),
}


def encoder_reference_points(wl: Workload, q_begin: int = 0, q_end: int | None = None) -> np.ndarray:
    """[nq, 2] (x, y) in [0, 1]: query q is pixel floor(q * I / Q) of the level-packed pyramid (q itself when Q == I, the
    encoder's case) and its reference point is that pixel's centre, ((x + 0.5) / w, (y + 0.5) / h)."""
    q_end = wl.Q if q_end is None else q_end
    q = np.arange(q_begin, q_end, dtype=np.int64)
    pix = q if wl.Q == wl.I else (q * wl.I) // max(wl.Q, 1)
    ref = np.zeros((q.size, 2))
    start = 0
    for h, w in wl.levels:
        m = (pix >= start) & (pix < start + h * w)
        rel = pix[m] - start
        if w > 0:
            ref[m, 0] = ((rel % w) + 0.5) / w
            ref[m, 1] = ((rel // w) + 0.5) / h
        start += h * w
    return ref


def make_inputs_numpy(wl: Workload, seed: int = 0, q_begin: int = 0, q_end: int | None = None,
                      attn_mode: str = "softmax", loc_lo: float = 0.0, loc_hi: float = 1.0,
                      loc_mode: str | None = None, offset_px: float = 2.0):
    """float64 numpy inputs for queries [q_begin, q_end) of every batch element.

    Element values depend only on their *global* index, so a query shard equals the
    corresponding slice of the unsharded tensors bit for bit.
    Returns dict(value, shapes, loc, attn, grad_out).
    """
    q_end = wl.Q if q_end is None else q_end
    B, I, H, D, Q, L, P = wl.B, wl.I, wl.H, wl.D, wl.Q, wl.L, wl.P
    nq = q_end - q_begin
    value = normal(seed, 1, B * I * H * D).reshape(B, I, H, D)
    per_q_loc, per_q_att, per_q_out = H * L * P * 2, H * L * P, H * D
    loc = np.empty((B, nq, H, L, P, 2))
    att = np.empty((B, nq, H, L, P))
    gout = np.empty((B, nq, H, D))
    loc_mode = wl.loc_mode if loc_mode is None else loc_mode
    if loc_mode not in ("uniform", "encoder"):
        raise ValueError(f"unknown loc_mode {loc_mode!r}")
    if loc_mode == "encoder":
        ref = encoder_reference_points(wl, q_begin, q_end)                             # [nq, 2] (x, y)
        wh = np.asarray([(max(w, 1), max(h, 1)) for h, w in wl.levels], dtype=np.float64)  # [L, 2] (w, h)
    for b in range(B):
        g0 = b * Q + q_begin
        if loc_mode == "encoder":
            off = normal(seed, 5, nq * per_q_loc, g0 * per_q_loc).reshape(nq, H, L, P, 2) * offset_px
            loc[b] = ref[:, None, None, None, :] + off / wh[None, None, :, None, :]
        else:
            loc[b] = (loc_lo + (loc_hi - loc_lo) * uniform(seed, 2, nq * per_q_loc, g0 * per_q_loc)
                      ).reshape(nq, H, L, P, 2)
        if attn_mode == "softmax":
            a = normal(seed, 3, nq * per_q_att, g0 * per_q_att).reshape(nq, H, L, P)
            a = np.exp(a - a.max(-1, keepdims=True))
            att[b] = a / a.sum(-1, keepdims=True)
        else:  # README.md:140 uses torch.rand
            att[b] = uniform(seed, 3, nq * per_q_att, g0 * per_q_att).reshape(nq, H, L, P)
        gout[b] = uniform(seed, 4, nq * per_q_out, g0 * per_q_out).reshape(nq, H, D)
    shapes = np.asarray(wl.levels, dtype=np.int64)
    return {"value": value, "shapes": shapes, "loc": loc, "attn": att, "grad_out": gout}


def make_row_inputs_numpy(wl: Workload, seed: int = 0, r_begin: int = 0, r_end: int | None = None):
    """Rows [r_begin, r_end) of the flattened (b, q) row space (row = b * Q + q): the same numbers as the
    corresponding slices of :func:`make_inputs_numpy` (softmax attention, loc in [0, 1]).
    Returns dict(value [B,I,H,D], shapes, loc [n,H,L,P,2], attn [n,H,L,P], grad_out [n,H,D])."""
    B, I, H, D, Q, L, P = wl.B, wl.I, wl.H, wl.D, wl.Q, wl.L, wl.P
    r_end = B * Q if r_end is None else r_end
    n = r_end - r_begin
    per_q_loc, per_q_att, per_q_out = H * L * P * 2, H * L * P, H * D
    value = normal(seed, 1, B * I * H * D).reshape(B, I, H, D)
    if wl.loc_mode == "encoder":  # row r = b * Q + q: the batch elements' reference points repeat
        rows = np.arange(r_begin, r_end, dtype=np.int64)
        ref_all = encoder_reference_points(wl)
        ref = ref_all[rows % max(Q, 1)]
        wh = np.asarray([(max(w, 1), max(h, 1)) for h, w in wl.levels], dtype=np.float64)
        off = normal(seed, 5, n * per_q_loc, r_begin * per_q_loc).reshape(n, H, L, P, 2) * 2.0
        loc = ref[:, None, None, None, :] + off / wh[None, None, :, None, :]
    else:
        loc = uniform(seed, 2, n * per_q_loc, r_begin * per_q_loc).reshape(n, H, L, P, 2)
    a = normal(seed, 3, n * per_q_att, r_begin * per_q_att).reshape(n, H, L, P)
    a = np.exp(a - a.max(-1, keepdims=True))
    att = a / a.sum(-1, keepdims=True)
    gout = uniform(seed, 4, n * per_q_out, r_begin * per_q_out).reshape(n, H, D)
    return {"value": value, "shapes": np.asarray(wl.levels, dtype=np.int64), "loc": loc, "attn": att, "grad_out": gout}


def make_inputs_torch(wl: Workload, device="cpu", seed: int = 0, dtype=None, rows=None, **kw):
    """Same data as :func:`make_inputs_numpy` (or, with ``rows=(r0, r1)``, :func:`make_row_inputs_numpy`), as torch
    tensors of the workload dtype."""
    import torch

    dt = getattr(torch, wl.dtype) if dtype is None else dtype
    d = make_row_inputs_numpy(wl, seed, rows[0], rows[1]) if rows is not None else make_inputs_numpy(wl, seed, **kw)
    out = {}
    for k, v in d.items():
        t = torch.from_numpy(v)
        out[k] = t.to(device) if k == "shapes" else t.to(dt).to(device)
    return out


def digest(a: np.ndarray, n_samples: int = 512) -> dict:
    """Size-independent fingerprint of a tensor: sum, abs-sum and strided samples."""
    flat = np.asarray(a, dtype=np.float64).reshape(-1)
    step = max(1, flat.size // n_samples)
    return {"sum": float(flat.sum()), "abs_sum": float(np.abs(flat).sum()),
            "samples": flat[::step][:n_samples].copy(), "step": step, "size": flat.size}
