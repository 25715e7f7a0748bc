"""The module's projections (``nn.Linear`` parameters, frontend.py:214-222) on a GPU when the row count is large.

Forward and grad_input are the library GEMMs ``nn.Linear`` runs.  The WEIGHT gradient ``dW[out, in] = dY[N, out]^T
X[N, in]`` is the problem case: N is 20 000 - 160 000 rows (every pixel of the pyramid / every query of the batch)
while out and in are a few hundred, and the BLAS library serves it with one 64 x 64 tile per workgroup and no split
over N — 16 to 24 workgroups on a 256-CU device (c2 module shape, bf16: 122 us per layer, a third of the module's
whole training step; ``tools/linear_wgrad_bench.py``).  Here it is a batched product over row blocks, summed in fp32
(58 us), and the bias gradient a two-stage column sum (16 us against 28).  Same arithmetic up to summation order.
"""
from __future__ import annotations

import torch
from torch.nn import functional as F

ROW_SPLIT_MIN_ROWS = 8192   # below this the plain layer is used (decoder-sized calls: the library GEMM is fine)
_BLOCK_MAX, _BLOCK_MIN = 2048, 512


def _acc_dtype(dt: torch.dtype) -> torch.dtype:
    return torch.float64 if dt == torch.float64 else torch.float32


def _row_block(n: int) -> int:
    """Largest block length in [_BLOCK_MIN, _BLOCK_MAX] (a multiple of 8) that divides n, or 0."""
    for c in range(_BLOCK_MAX, _BLOCK_MIN - 1, -8):
        if n % c == 0:
            return c
    return 0


def row_split_weight_grad(gy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """``gy^T @ x`` for ``gy [N, out]``, ``x [N, in]`` as a batched product over row blocks; the blocks are summed in
    fp32 (fp64 operands: fp64)."""
    n = x.shape[0]
    acc = _acc_dtype(x.dtype)
    c = _row_block(n) or _BLOCK_MAX
    s = n // c
    main = s * c
    if s == 0:
        return (gy.t() @ x).to(acc)
    parts = torch.bmm(gy[:main].view(s, c, -1).transpose(1, 2), x[:main].view(s, c, -1))
    w = parts.sum(0, dtype=acc)
    if main < n:
        w = w + (gy[main:].t() @ x[main:]).to(acc)
    return w


def column_sum(gy: torch.Tensor) -> torch.Tensor:
    """``gy.sum(0)`` in two stages (the one-stage column reduction keeps few workgroups busy); summed in fp32 (fp64
    operands: fp64)."""
    n, o = gy.shape
    acc = _acc_dtype(gy.dtype)
    for c in (64, 40, 32, 16, 8):
        if n % c == 0:
            return gy.view(n // c, c, o).sum(1, dtype=acc).sum(0)
    return gy.sum(0, dtype=acc)


_ZEROS: dict = {}  # (shape, dtype, device) -> zeros (the dead rows of the widened weight / bias)


def _zeros(shape, like):
    key = (tuple(shape), like.dtype, like.device)
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros(shape, dtype=like.dtype, device=like.device)
    return z


def _linear_into_padded_rows(xc, wc, bc, pad_elems: int):
    """``F.linear(xc, wc, bc)`` with its output rows ``out_features + pad_elems`` elements apart, as the product with a
    WIDENED weight — ``pad_elems`` zero rows behind ``wc`` — whose dense ``[N, out + pad]`` result is that layout; returns
    the ``[..., out_features]`` view.  (Measured on MI355X, N = 21 760 rows, 256 -> 256, tools/linear_pad_bench.py: the
    dense GEMM 41 us in fp32 / 18 in bf16; the widened one 36 / 20; the same GEMM with ``out=`` a strided view 55 / 23 —
    the library takes a slower path for a leading dimension that is not the row length; dense + strided copy 51 / 25.)"""
    o = wc.shape[0]
    wp = torch.cat([wc, _zeros((pad_elems, wc.shape[1]), wc)])
    bp = None if bc is None else torch.cat([bc, _zeros((pad_elems,), bc)])
    return F.linear(xc, wp, bp)[..., :o]


class _RowSplitLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, pad_elems=0):
        # autocast: what F.linear would do to its arguments, done once so that the 16-bit copies can be saved
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            xc, wc, bc = x.to(dt), weight.to(dt), None if bias is None else bias.to(dt)
        else:
            xc, wc, bc = x, weight, bias
        ctx.save_for_backward(xc, wc)
        ctx.dtypes = (x.dtype, weight.dtype, None if bias is None else bias.dtype)
        with torch.autocast("cuda", enabled=False):
            if pad_elems:
                return _linear_into_padded_rows(xc, wc, bc, pad_elems)
            return F.linear(xc, wc, bc)

    @staticmethod
    def backward(ctx, gy):
        xc, wc = ctx.saved_tensors
        xdt, wdt, bdt = ctx.dtypes
        gx = gw = gb = None
        with torch.autocast("cuda", enabled=False):
            gy2 = gy.reshape(-1, gy.shape[-1]).to(xc.dtype)
            if ctx.needs_input_grad[0]:
                gx = (gy2 @ wc).view(xc.shape).to(xdt)
            if ctx.needs_input_grad[1]:
                gw = row_split_weight_grad(gy2, xc.reshape(-1, xc.shape[-1])).to(wdt)
            if bdt is not None and ctx.needs_input_grad[2]:
                gb = column_sum(gy2).to(bdt)
        return gx, gw, gb, None


def _plain_linear(layer) -> bool:
    """An ordinary ``nn.Linear`` and nothing else: not a subclass or a wrapper whose ``forward`` does more than
    ``F.linear(x, weight, bias)`` (LoRA / PEFT adapters keep the BASE weight in ``.weight``; quantised and observer
    layers), and no hooks of any kind that a call bypassing ``layer.__call__`` would skip."""
    if type(layer) is not torch.nn.Linear:
        return False
    if layer._forward_hooks or layer._forward_pre_hooks or layer._backward_hooks or layer._backward_pre_hooks:
        return False
    if getattr(layer, "_forward_hooks_with_kwargs", None) or getattr(layer, "_forward_pre_hooks_with_kwargs", None):
        return False
    g = torch.nn.modules.module
    return not (g._global_forward_hooks or g._global_forward_pre_hooks or g._global_backward_hooks
                or g._global_backward_pre_hooks)


def projection(layer: torch.nn.Linear, x: torch.Tensor, pad_rows: bool = False) -> torch.Tensor:
    """``layer(x)``; on a GPU with many rows through the row-split weight gradient above — only for a plain, hook-free
    ``nn.Linear`` (the reference module always calls the layer itself, frontend.py:253-267: whatever a user hung on it —
    forward hooks, a LoRA wrapper, a quantised replacement — has to run).

    ``pad_rows`` (the value projection): where this function runs the GEMM itself (same conditions, and plain no-grad
    inference calls), the output rows are written ``functional.value_row_pad`` bytes apart — a ``[..., out_features]``
    view of a wider buffer the attention kernels read in place."""
    gpu_plain = x.device.type == "cuda" and x.dim() >= 2 and x.is_floating_point() and \
        not torch.compiler.is_compiling() and _plain_linear(layer) and \
        (torch.is_autocast_enabled("cuda") or x.dtype == layer.weight.dtype)
    pad = 0
    if pad_rows and gpu_plain:
        from .functional import value_row_pad
        es = (torch.empty((), dtype=torch.get_autocast_dtype("cuda")) if torch.is_autocast_enabled("cuda") else x).element_size()
        # fp32 values only: measured at the c2 module shape (tools/module_pad_ab.py, profiles/r06_module_pad_ab.txt) the
        # fused kernels over fp32 rows gain (sample gradients 149 -> 127 us, forward 75.6 -> 73.7 at 10 000 queries; 42.7 ->
        # 39.7 / 32 -> 28.4 at 2 500) while the 16-bit storage kernels do not move (76 / 120 us either way)
        pad = value_row_pad(layer.out_features * es) // es if es == 4 else 0
    if gpu_plain and x.numel() // x.shape[-1] >= ROW_SPLIT_MIN_ROWS and torch.is_grad_enabled() and \
            (layer.weight.requires_grad or x.requires_grad):
        return _RowSplitLinear.apply(x, layer.weight, layer.bias, pad)
    if pad and gpu_plain and not (torch.is_grad_enabled() and (layer.weight.requires_grad or x.requires_grad or
                                                                (layer.bias is not None and layer.bias.requires_grad))):
        # nothing to differentiate (inference): the same GEMM straight into the padded rows
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            xc, wc, bc = x.to(dt), layer.weight.to(dt), None if layer.bias is None else layer.bias.to(dt)
        else:
            xc, wc, bc = x, layer.weight, layer.bias
        with torch.autocast("cuda", enabled=False):
            return _linear_into_padded_rows(xc.detach(), wc.detach(), None if bc is None else bc.detach(), pad)
    return layer(x)
