"""Functional API: the reference's operator interface over the MI355X HIP kernels.

Mirrors /root/reference/src/msda_triton/frontend.py (same names, argument meaning and errors):

  multiscale_deformable_attention          frontend.py:145-172   public entry
  hip_multiscale_deformable_attention      frontend.py:71-105    GPU path (there: triton_multiscale_…)
  native_multiscale_deformable_attention   frontend.py:15-68     host-tensor path
  msda_hip_fwd / msda_hip_bwd              kernels.py:351-379 / 556-592   launcher pair (the native seam)

Deliberate differences from the reference:
  * GPU tensors never fall back to a CPU/eager path.  The reference wraps its GPU path in
    ``try/except Exception`` (frontend.py:167-172); here a missing extension or a failed launch
    raises.  Host (CPU) tensors use the native formulation, as the reference documents
    (README.md:127 "cpu uses fallback native torch version").
  * bfloat16 runs natively on the GPU (the reference rejects it, frontend.py:84).
  * an unknown ``padding_mode`` raises ``ValueError`` instead of reaching ``grid_sample``.
"""
from __future__ import annotations

import os
from typing import Literal, Optional, Tuple

import torch
from torch.amp import custom_bwd, custom_fwd
from torch.autograd.function import Function, once_differentiable

from . import _ext, _lib

_SUFFIX = {
    torch.float32: "f32",
    torch.float16: "f16",
    torch.bfloat16: "bf16",
    torch.float64: "f64",
}
VALID_DTYPES = tuple(_SUFFIX)
# mixed storage: `img` (and its gradient) in 16 bits next to fp32 sampling points / attention weights / output
_MIXED_SUFFIX = {torch.bfloat16: "f32_vbf16", torch.float16: "f32_vf16"}


def dtypes_supported(img_dtype: torch.dtype, compute_dtype: torch.dtype) -> bool:
    """One dtype for every tensor, or a 16-bit `img` with fp32 everything else (the result is then fp32)."""
    if img_dtype == compute_dtype:
        return img_dtype in _SUFFIX
    return compute_dtype == torch.float32 and img_dtype in _MIXED_SUFFIX


# (img, sampling_points, attention_weights) dtypes the kernels take, for the fast path's single lookup
_DTYPE_TRIPLES = frozenset([(t, t, t) for t in _SUFFIX] + [(t, torch.float32, torch.float32) for t in _MIXED_SUFFIX])
_SHAPE_DTYPES = frozenset((torch.int64, torch.int32, torch.int16, torch.int8, torch.uint8))


def _suffix_for(img_dtype: torch.dtype, compute_dtype: torch.dtype) -> str:
    if not dtypes_supported(img_dtype, compute_dtype):
        raise ValueError(
            "`img`, `sampling_points` and `attention_weights` should share one dtype (or `img` be float16 / bfloat16 "
            f"next to float32 sampling inputs), but got {img_dtype} and {compute_dtype}.")
    return _SUFFIX[img_dtype] if img_dtype == compute_dtype else _MIXED_SUFFIX[img_dtype]


# the module's 16-bit storage (fused entry points only): value and projection in one 16-bit dtype, reference points fp32
_FUSED_STORAGE_SUFFIX = {torch.bfloat16: "f32_sbf16", torch.float16: "f32_sf16"}


def fused_storage_dtypes(img_dtype, proj_dtype, ref_dtype) -> bool:
    """16-bit value pyramid and projection next to fp32 reference points: what autocast's GEMMs hand the module core.
    The kernels compute in fp32 and return / differentiate in the 16-bit dtype (``msda_*_fused_f32_sbf16`` / ``_sf16``)."""
    return img_dtype == proj_dtype and proj_dtype in _FUSED_STORAGE_SUFFIX and ref_dtype == torch.float32


def _fused_suffix_for(img_dtype, proj_dtype, ref_dtype) -> str:
    if fused_storage_dtypes(img_dtype, proj_dtype, ref_dtype):
        return _FUSED_STORAGE_SUFFIX[proj_dtype]
    if ref_dtype != proj_dtype:
        raise ValueError(f"`proj` and `reference_points` should share one dtype (or `img` and `proj` be float16 / bfloat16 "
                         f"next to float32 `reference_points`), but got {proj_dtype} and {ref_dtype}.")
    return _suffix_for(img_dtype, proj_dtype)


def _padding_code(padding_mode: str) -> int:
    try:
        return _lib.PADDING_MODES[padding_mode]
    except KeyError:
        raise ValueError(f"`padding_mode` should be 'border' or 'zeros', but got {padding_mode!r}.") from None


def _dims(img: torch.Tensor, sampling_points: torch.Tensor, attention_weights: torch.Tensor, img_shapes: torch.Tensor):
    if img.dim() != 4 or sampling_points.dim() != 6 or attention_weights.dim() != 5:
        raise ValueError(
            "expected img [B,I,H,C], sampling_points [B,N,H,L,P,2], attention_weights [B,N,H,L,P]; got "
            f"{tuple(img.shape)}, {tuple(sampling_points.shape)}, {tuple(attention_weights.shape)}")
    B, I, H, D = img.shape
    B2, Q, H2, L, P, two = sampling_points.shape
    if (B2, H2, two) != (B, H, 2) or tuple(attention_weights.shape) != (B, Q, H, L, P):
        raise ValueError(
            f"inconsistent shapes: img {tuple(img.shape)}, sampling_points {tuple(sampling_points.shape)}, "
            f"attention_weights {tuple(attention_weights.shape)}")
    if tuple(img_shapes.shape) != (L, 2):
        raise ValueError(f"`img_shapes` should be [{L}, 2], but got {tuple(img_shapes.shape)}.")
    return B, I, H, D, Q, L, P


def _shapes_i64(img_shapes: torch.Tensor) -> torch.Tensor:
    if img_shapes.dtype not in (torch.int64, torch.int32, torch.int16, torch.int8, torch.uint8):
        raise ValueError(f"`img_shapes` should be an integer tensor, but got {img_shapes.dtype}.")
    if img_shapes.dtype == torch.int64 and img_shapes.is_contiguous():
        return img_shapes
    return img_shapes.to(torch.int64).contiguous()  # stays on the device: no host sync


_BWD_SUPPORTED: dict = {}  # (B, I, H, D, Q, L, P, element size) -> msda_bwd_supported


def check_backward_supported(img, sampling_points, num_queries: Optional[int] = None) -> None:
    """Raise at FORWARD time when the value pyramid needs a gradient the library cannot produce for these sizes (a plane
    of 2^22 pixels or more, or a head dimension beyond the sorted pipeline's 32-bit slot offsets, on a problem too
    large for the single-launch kernel) — not from the backward in the middle of a training step.
    ``num_queries``: ``sampling_points`` is a flat run of rows ``[rows, H, L, P, 2]`` of batch elements with that many
    queries each (the row-sharded operator)."""
    B, I, H, D = img.shape
    if num_queries is None:
        _, Q, _, L, P = sampling_points.shape[:5]
    else:
        Q, (_, _, L, P) = int(num_queries), sampling_points.shape[:4]
    key = (B, I, H, D, Q, L, P, sampling_points.element_size())
    ok = _BWD_SUPPORTED.get(key)
    if ok is None:
        ok = _BWD_SUPPORTED[key] = bool(_lib.load().msda_bwd_supported(*key))
    if not ok:
        raise ValueError(
            f"`img` requires a gradient, but grad_value is not available for this shape (I={I} pixels per plane, D={D}, "
            f"Q={Q}): the sorted pipeline addresses planes below 2^22 pixels with 4*I*D*sizeof(acc) < 2^31 bytes of "
            "partial rows.  Detach `img` or split the pyramid.")


def _check_devices(*tensors: torch.Tensor) -> torch.device:
    """All tensors must live on ONE gpu (reference: frontend.py:93-95).  Raised before any ``data_ptr()`` is handed
    to the library: a host or foreign-device pointer would otherwise fault inside the kernel."""
    devices = [t.device for t in tensors]
    if any(d.type != "cuda" for d in devices) or any(d != devices[0] for d in devices):
        raise ValueError(f"Expected all inputs to be on one gpu, but got {devices}.")
    return devices[0]


def padded_value_rows(B: int, I: int, H: int, D: int, dtype: torch.dtype, device, pad_bytes: Optional[int] = None):  # noqa: E741
    """A ``[B, I, H, D]`` tensor whose pixels' rows sit ``H * D * size + pad`` bytes apart (uninitialised): the view
    ``buf[:, :, :H * D].view(B, I, H, D)`` of a ``[B, I, H * D + pad / size]`` buffer.  The kernels read such a tensor
    in place (``value_row_stride`` of the C ABI, include/msda_hip.h).  ``pad_bytes=None``: :func:`value_row_pad`'s
    choice — one 128-byte line when the dense stride is a multiple of 256 bytes, else none."""
    es = torch.empty((), dtype=dtype).element_size()
    pad = value_row_pad(H * D * es) if pad_bytes is None else int(pad_bytes)
    if pad % es:
        raise ValueError(f"pad_bytes={pad} is not a multiple of the element size {es}")
    buf = torch.empty((B, I, H * D + pad // es), dtype=dtype, device=device)
    return buf[:, :, :H * D].view(B, I, H, D)


def value_row_pad(dense_row_bytes: int) -> int:
    """Bytes to put behind every pixel's rows of a value pyramid this package allocates itself.  The gather kernels are
    bound by the vector L1, which picks one of four tag RAMs from the low bits of a row's 128-byte line index: with the
    pixels' rows an EVEN number of lines apart (1 024 bytes: 8 heads x 32 channels x fp32; 512: the same in 16 bits) the
    rows of one head keep hitting the same tag RAMs, and that head's plane gathers up to 20 % slower (HISTORY.md 4 item 5).  One
    extra line makes the distance odd: every head cycles through all eight residues.  Measured (round 6,
    profiles/r06_row_stride_ab.txt): forward -4 ... -8 %, sample gradients -9 ... -10 % at 900 ... 5 000 queries per batch
    element; c3's bf16 forward -7.5 %; nothing at 10 000 queries, where two-plane workgroups already level it."""
    env = os.environ.get("MSDA_VALUE_ROW_PAD")  # (A/B runs and an escape hatch: bytes, 0 = never pad)
    if env is not None and env.strip().lstrip("-").isdigit():
        return max(0, int(env))
    return 128 if dense_row_bytes % 256 == 0 else 0


def _value_rows(img: torch.Tensor):
    """-> (tensor the kernels can address, its value_row_stride in bytes; 0: dense).  A ``[B, I, H, D]`` view whose
    pixels are a constant number of bytes apart with the H * D channels of a pixel contiguous (:func:`padded_value_rows`)
    is read in place; any other layout is copied dense, as the reference does (kernels.py:367-370)."""
    if img.is_contiguous():
        return img, 0
    B, I, H, D = img.shape  # noqa: E741
    st = img.stride()
    if D > 0 and H > 0 and I > 0 and st[3] == 1 and st[2] == D and st[1] >= H * D and (B == 1 or st[0] == I * st[1]) and \
            (st[1] * img.element_size()) % 16 == 0 and I * st[1] * img.element_size() < 2 ** 31:
        return img, st[1] * img.element_size()
    return img.contiguous(), 0


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_FUSED_LP_LIMIT: dict = {}  # (D, element size) -> msda_fused_lp_limit
_WS_BYTES: dict = {}  # (B, I, H, D, Q, L, P, elem, option epoch) -> msda_bwd_workspace_bytes


def _stream_ptr(device: torch.device) -> int:
    """hipStream_t of PyTorch's current stream on ``device`` (the raw getter is ~10x cheaper than building a
    torch.cuda.Stream object; same value)."""
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def _autocast_on() -> bool:
    try:
        return torch.is_autocast_enabled("cuda")
    except TypeError:  # older signature
        return torch.is_autocast_enabled()


class _OnDevice:
    """``with torch.cuda.device(d)`` only when ``d`` is not already the current device (the context manager costs
    several microseconds per call, which matters for Grounding-DINO-sized problems)."""

    __slots__ = ("ctx",)

    def __init__(self, device: torch.device):
        self.ctx = None if (device.index is None or device.index == torch.cuda.current_device()) \
            else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


class KernelTimer:
    """Per-kernel HIP-event timing for measurements (``bench.py``).

    While active (``with KernelTimer() as kt:``) every launcher brackets each kernel launch with
    events recorded on the stream the kernel is launched on, and the backward is issued as two C-ABI
    calls so its two kernels are timed separately.  ``kt.summary()`` (after a device sync) gives
    ``{kernel: (launches, mean_ms)}``.  Off by default: zero cost on the normal path.
    """

    active: Optional["KernelTimer"] = None

    def __init__(self):
        self.records = []  # (name, start_event, end_event)

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None
        return False

    def launch(self, name: str, device: torch.device, call):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.current_stream(device)
        start.record(stream)
        rc = call()
        end.record(stream)
        self.records.append((name, start, end))
        return rc

    def summary(self):
        acc = {}
        for name, start, end in self.records:
            n, total = acc.get(name, (0, 0.0))
            acc[name] = (n + 1, total + start.elapsed_time(end))
        return {k: (n, total / n) for k, (n, total) in acc.items()}


# ------------------------------------------------------------------------------------------
# launcher pair — the native seam (reference: kernels.py:351-379, 556-592)
# ------------------------------------------------------------------------------------------
def msda_hip_fwd(img, img_shapes, sampling_points, attention_weights, padding_mode, align_corners,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Allocate ``out`` (or write into the caller's contiguous ``[B, Q, H, D]`` buffer, e.g. a slice of a gather
    buffer) and enqueue the forward kernel on the current stream (no host sync)."""
    B, I, H, D, Q, L, P = _dims(img, sampling_points, attention_weights, img_shapes)
    _check_devices(img, img_shapes, sampling_points, attention_weights)
    pad = _padding_code(padding_mode)
    cdt = sampling_points.dtype  # dtype of everything but `img` (the same, or fp32 next to a 16-bit `img`)
    if attention_weights.dtype != cdt:
        raise ValueError(f"`sampling_points` and `attention_weights` should share one dtype, but got {cdt} and "
                         f"{attention_weights.dtype}.")
    suf = _suffix_for(img.dtype, cdt)
    (img, vrow), sampling_points, attention_weights = _value_rows(img), sampling_points.contiguous(), attention_weights.contiguous()
    shapes = _shapes_i64(img_shapes)
    if out is None:
        out = torch.empty((B, Q, H, D), dtype=cdt, device=img.device)
    elif tuple(out.shape) != (B, Q, H, D) or out.dtype != cdt or out.device != img.device or not out.is_contiguous():
        raise ValueError(f"`out` should be a contiguous {(B, Q, H, D)} {cdt} tensor on {img.device}, but got "
                         f"{tuple(out.shape)} {out.dtype} on {out.device} (contiguous: {out.is_contiguous()}).")
    lib = _lib.load()
    fn = getattr(lib, f"msda_fwd_{suf}")

    def call():
        return fn(img.data_ptr(), shapes.data_ptr(), sampling_points.data_ptr(), attention_weights.data_ptr(),
                  out.data_ptr(), B, I, H, D, Q, L, P, pad, int(bool(align_corners)), vrow, _stream_ptr(img.device))

    with _OnDevice(img.device):
        timer = KernelTimer.active
        rc = timer.launch("msda_fwd", img.device, call) if timer else call()
    _lib.check(rc, f"msda_fwd_{suf}")
    return out


def level_cells_of(level_shapes, num_levels: Optional[int] = None, num_pixels: Optional[int] = None) -> int:
    """``level_shapes``: the pyramid's (height, width) pairs AS HOST NUMBERS (e.g. Hugging Face's
    ``spatial_shapes_list``), or None.  Returns the bound ``msda_bwd_<dtype>`` takes as ``max_level_cells`` — the bilinear cells of the
    largest level, (h + 1) * (w + 1) — or 0 for "unknown".  With it the backward can use its single-launch grad_value
    kernel on decoder-sized calls over real-image pyramids (include/msda_hip.h); it must describe the same pyramid as
    the ``img_shapes`` tensor (a level larger than promised gets NaN gradients)."""
    if level_shapes is None:
        return 0
    if isinstance(level_shapes, torch.Tensor):
        if level_shapes.device.type != "cpu":
            raise ValueError("`level_shapes` should be host numbers (a device tensor would need a synchronisation); "
                             "pass e.g. `spatial_shapes_list`")
        level_shapes = level_shapes.tolist()
    # (plain loops: torch.compile traces this function with the list as a constant)
    levels, pixels, cells = 0, 0, 0
    for h, w in level_shapes:
        h, w = int(h), int(w)
        levels += 1
        pixels += h * w
        cells = max(cells, (h + 1) * (w + 1))
    # what can be checked without reading `img_shapes` back from the device: the level count and the pixel total of
    # the call it is supposed to describe (a list describing another pyramid would give NaN grad_value rows)
    if num_levels is not None and levels != num_levels:
        raise ValueError(f"`level_shapes` has {levels} levels, `img_shapes` {num_levels}")
    if num_pixels is not None and pixels != num_pixels:
        raise ValueError(f"`level_shapes` describes {pixels} pixels, `img` has {num_pixels}")
    return cells


def msda_hip_bwd(out_grad, img, img_shapes, sampling_points, attention_weights, padding_mode, align_corners,
                 needs: Tuple[bool, bool, bool] = (True, True, True),
                 out: Optional[Tuple[Optional[torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor]]] = None,
                 level_cells: int = 0,
                 ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Returns ``(img_grad, sampling_points_grad, attention_weights_grad)``; entries not in ``needs`` are None.
    ``level_cells``: see :func:`level_cells_of` (0: unknown).

    Gradients are allocated contiguous (the reference's ``zeros_like(...).contiguous()`` would write
    into a temporary for permuted inputs, kernels.py:570-578) and are fully written by the kernels,
    so they are not pre-zeroed.  ``out``: contiguous buffers of the gradients' shapes to write into instead (e.g.
    slices of a shard's gradient tensors); an entry that is None is allocated here.
    """
    B, I, H, D, Q, L, P = _dims(img, sampling_points, attention_weights, img_shapes)
    _check_devices(img, img_shapes, sampling_points, attention_weights, out_grad)
    pad = _padding_code(padding_mode)
    cdt = sampling_points.dtype
    if attention_weights.dtype != cdt:
        raise ValueError(f"`sampling_points` and `attention_weights` should share one dtype, but got {cdt} and "
                         f"{attention_weights.dtype}.")
    suf = _suffix_for(img.dtype, cdt)
    (img, vrow), sampling_points, attention_weights = _value_rows(img), sampling_points.contiguous(), attention_weights.contiguous()
    out_grad = out_grad.contiguous()
    if out_grad.dtype != cdt:
        out_grad = out_grad.to(cdt)
    shapes = _shapes_i64(img_shapes)
    want_value = bool(needs[0])
    want_sample = bool(needs[1] or needs[2])

    def buf(i, shape, wanted, dtype):
        if not wanted:
            return None
        t = out[i] if out is not None else None
        if t is None:
            return torch.empty(shape, dtype=dtype, device=img.device)
        if tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != img.device or not t.is_contiguous():
            raise ValueError(f"`out[{i}]` should be a contiguous {tuple(shape)} {dtype} tensor on {img.device}")
        return t

    g_img = buf(0, (B, I, H, D), want_value, img.dtype)
    g_pts = buf(1, (B, Q, H, L, P, 2), want_sample, cdt)
    g_att = buf(2, (B, Q, H, L, P), want_sample, cdt)
    if want_value or want_sample:
        lib = _lib.load()
        fn = getattr(lib, f"msda_bwd_{suf}")  # (the level-size bound travels as an argument: include/msda_hip.h)
        ws, ws_bytes = None, 0
        level_cells = int(level_cells)
        if want_value:  # scratch: the inverted index of grad_value (grad_loc / grad_attn never need any)
            # all three gradients in ONE call: the sorted records may use the gradient buffers themselves (smaller
            # workspace); the per-kernel timer issues the halves as two calls and needs the full size
            # (the library's own conditions: 16-byte aligned gradient buffers, no forced side-stream fork)
            flags = _lib.WS_RECORDS_IN_GRADS if (want_sample and KernelTimer.active is None
                                                 and g_pts.data_ptr() % 16 == 0 and g_att.data_ptr() % 16 == 0
                                                 and g_img.data_ptr() % 16 == 0
                                                 and _lib.get_option("overlap") != 1) else 0
            key = (B, I, H, D, Q, L, P, sampling_points.element_size(), img.element_size(), _lib.OPTION_EPOCH, level_cells,
                   flags)
            ws_bytes = _WS_BYTES.get(key)
            if ws_bytes is None:
                ws_bytes = _WS_BYTES[key] = int(lib.msda_bwd_workspace_bytes(*key[:9], level_cells, flags))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=img.device)

        def call(value_part: bool, sample_part: bool):
            return fn(out_grad.data_ptr(), img.data_ptr(), shapes.data_ptr(), sampling_points.data_ptr(),
                      attention_weights.data_ptr(),
                      g_img.data_ptr() if value_part else None,
                      g_pts.data_ptr() if sample_part else None,
                      g_att.data_ptr() if sample_part else None,
                      B, I, H, D, Q, L, P, pad, int(bool(align_corners)), level_cells, vrow,
                      ws.data_ptr() if ws is not None else None, ws_bytes,
                      _stream_ptr(img.device))

        with _OnDevice(img.device):
            timer = KernelTimer.active
            if timer is None:
                rc = call(want_value, want_sample)
            else:  # one C-ABI call per kernel so each gets its own event pair
                rc = 0
                if want_sample:
                    rc = timer.launch("msda_bwd_sample", img.device, lambda: call(False, True))
                if rc == 0 and want_value:
                    rc = timer.launch("msda_bwd_value", img.device, lambda: call(True, False))
        _lib.check(rc, f"msda_bwd_{suf}")
    return g_img, (g_pts if needs[1] else None), (g_att if needs[2] else None)


# ------------------------------------------------------------------------------------------
# autograd boundary (reference: _TritonMultiscaleDeformableAttentionFunction, frontend.py:108-142)
# ------------------------------------------------------------------------------------------
class _HipMultiscaleDeformableAttentionFunction(Function):

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)  # under autocast the op runs in fp32 (frontend.py:111)
    def forward(ctx, img, img_shapes, sampling_points, attention_weights, padding_mode, align_corners, level_cells=0):
        if ctx.needs_input_grad[0]:
            check_backward_supported(img, sampling_points)
        ctx.save_for_backward(img, img_shapes, sampling_points, attention_weights)
        ctx.padding_mode = padding_mode
        ctx.align_corners = align_corners
        ctx.level_cells = int(level_cells)
        return msda_hip_fwd(img, img_shapes, sampling_points, attention_weights, padding_mode, align_corners)

    @staticmethod
    @once_differentiable
    @custom_bwd(device_type="cuda")
    def backward(ctx, out_grad):
        img, img_shapes, sampling_points, attention_weights = ctx.saved_tensors
        needs = (ctx.needs_input_grad[0], ctx.needs_input_grad[2], ctx.needs_input_grad[3])
        g_img, g_pts, g_att = msda_hip_bwd(
            out_grad, img, img_shapes, sampling_points, attention_weights,
            ctx.padding_mode, ctx.align_corners, needs, level_cells=ctx.level_cells)
        return g_img, None, g_pts, g_att, None, None, None


def hip_multiscale_deformable_attention(
    img: torch.Tensor,
    img_shapes: torch.Tensor,
    sampling_points: torch.Tensor,
    attention_weights: torch.Tensor,
    padding_mode: Literal["border", "zeros"],
    align_corners: bool,
    level_shapes=None,
) -> torch.Tensor:
    """GPU path.  Same contract as the reference's ``triton_multiscale_deformable_attention``
    (frontend.py:71-105): ``ValueError`` on unsupported dtype or non-GPU inputs.  ``level_shapes`` (an addition): the
    pyramid's (h, w) pairs as host numbers, see :func:`level_cells_of`."""
    # Fast path (the small decoder / README shapes spend more host time than device time per call): every check below
    # as one conjunction; whatever fails it — or needs the Python Function — takes the checks that carry the messages.
    dev, dt, shp, ish = img.device, sampling_points.dtype, sampling_points.shape, img.shape
    if dev.type == "cuda" and not torch.compiler.is_compiling() and (ext := _ext.load()) is not None and \
            KernelTimer.active is None and \
            (img.dtype, dt, attention_weights.dtype) in _DTYPE_TRIPLES and padding_mode in _lib.PADDING_MODES and \
            img_shapes.device == dev and sampling_points.device == dev and attention_weights.device == dev and \
            len(ish) == 4 and len(shp) == 6 and shp[0] == ish[0] and shp[2] == ish[2] and shp[5] == 2 and \
            attention_weights.shape == shp[:5] and img_shapes.shape == (shp[3], 2) and \
            img_shapes.dtype in _SHAPE_DTYPES and not _autocast_on():
        if img.requires_grad and torch.is_grad_enabled():
            check_backward_supported(img, sampling_points)
        return ext.msda(img, img_shapes, sampling_points, attention_weights, _lib.PADDING_MODES[padding_mode],
                        bool(align_corners), level_cells_of(level_shapes, shp[3], ish[1]) if level_shapes is not None else 0)
    for name, t in (("img", img), ("sampling_points", sampling_points), ("attention_weights", attention_weights)):
        if t.dtype not in VALID_DTYPES:
            raise ValueError(f"Dtype of `{name}` should be in {list(VALID_DTYPES)}, but got {t.dtype}.")
    if sampling_points.dtype != attention_weights.dtype or not dtypes_supported(img.dtype, sampling_points.dtype):
        raise ValueError(
            "`img`, `sampling_points` and `attention_weights` should share one dtype (or `img` be float16 / bfloat16 "
            f"next to float32 sampling inputs), but got {img.dtype}, {sampling_points.dtype}, {attention_weights.dtype}.")
    _check_devices(img, img_shapes, sampling_points, attention_weights)
    _padding_code(padding_mode)
    if torch.compiler.is_compiling():  # traced by torch.compile / export: use the registered custom ops
        from .compile_op import compiled_multiscale_deformable_attention
        return compiled_multiscale_deformable_attention(
            img, img_shapes, sampling_points, attention_weights, padding_mode, align_corners,
            level_cells_of(level_shapes, img_shapes.shape[0], img.shape[1]))
    # Optional C++ autograd glue over the same C ABI (csrc/msda_torch_ext.cpp): same kernels, a fraction of the host
    # time per call.  The Python Function below serves autocast (fp32 casting), per-kernel timing and every
    # installation where the binding was not built.
    ext = _ext.load()
    if ext is not None and KernelTimer.active is None and not _autocast_on():
        _dims(img, sampling_points, attention_weights, img_shapes)
        _shapes_i64(img_shapes)
        if img.requires_grad and torch.is_grad_enabled():
            check_backward_supported(img, sampling_points)
        return ext.msda(img, img_shapes, sampling_points, attention_weights, _padding_code(padding_mode),
                        bool(align_corners), level_cells_of(level_shapes, img_shapes.shape[0], img.shape[1]))
    return _HipMultiscaleDeformableAttentionFunction.apply(
        img, img_shapes, sampling_points, attention_weights, padding_mode, bool(align_corners),
        level_cells_of(level_shapes, img_shapes.shape[0], img.shape[1]))


# ------------------------------------------------------------------------------------------
# module core with the prologue fused into the kernel (reference frontend.py:253-289)
# ------------------------------------------------------------------------------------------
def module_sampling_inputs(proj: torch.Tensor, img_shapes: torch.Tensor, reference_points: torch.Tensor):
    """The reference module's prologue in plain PyTorch: raw projection [B,N,H,L,P,3] ->
    (sampling_points [B,N,H,L,P,2], attention_weights [B,N,H,L,P])  (frontend.py:253-284)."""
    B, N, H, L, P, _ = proj.shape
    offsets, logits = proj[..., :2], proj[..., 2]
    attention_weights = logits.reshape(B, N, H, L * P).softmax(dim=-1).reshape(B, N, H, L, P)
    ref = reference_points[:, :, None, None, None, :]
    coords = reference_points.shape[-1]
    if coords == 2:
        # NB: the reference divides the (x, y) offsets by img_shapes in its stored (h, w) order
        # (frontend.py:275); reproduced for parity — it only matters for non-square levels.
        sampling_points = ref + offsets / img_shapes[:, None, :]
    elif coords == 4:
        sampling_points = ref[..., :2] + offsets * ref[..., 2:] / (2 * P)
    else:
        raise ValueError(f"`reference_points` should have the last dim either 2 or 4, but got {coords}.")
    return sampling_points, attention_weights


def msda_hip_fwd_fused(img, img_shapes, proj, reference_points, padding_mode, align_corners) -> Optional[torch.Tensor]:
    """Forward with softmax + sampling-point math done in the kernel prologue.  Returns None when the
    library declines (L*P too large for one pass): the caller then takes the unfused route."""
    B, I, H, D = img.shape
    B2, Q, H2, L, P, three = proj.shape
    if (B2, H2, three) != (B, H, 3) or tuple(reference_points.shape[:2]) != (B, Q):
        raise ValueError(f"inconsistent shapes: img {tuple(img.shape)}, proj {tuple(proj.shape)}, "
                         f"reference_points {tuple(reference_points.shape)}")
    ref_dim = reference_points.shape[-1]
    if ref_dim not in (2, 4):
        raise ValueError(f"`reference_points` should have the last dim either 2 or 4, but got {ref_dim}.")
    if tuple(img_shapes.shape) != (L, 2):
        raise ValueError(f"`img_shapes` should be [{L}, 2], but got {tuple(img_shapes.shape)}.")
    _check_devices(img, img_shapes, proj, reference_points)
    pad = _padding_code(padding_mode)
    cdt = proj.dtype
    suf = _fused_suffix_for(img.dtype, cdt, reference_points.dtype)
    (img, vrow), proj, reference_points = _value_rows(img), proj.contiguous(), reference_points.contiguous()
    shapes = _shapes_i64(img_shapes)
    out = torch.empty((B, Q, H, D), dtype=cdt, device=img.device)
    lib = _lib.load()
    fn = getattr(lib, f"msda_fwd_fused_{suf}")

    def call():
        return fn(img.data_ptr(), shapes.data_ptr(), proj.data_ptr(), reference_points.data_ptr(), out.data_ptr(),
                  B, I, H, D, Q, L, P, ref_dim, pad, int(bool(align_corners)), vrow, _stream_ptr(img.device))

    with _OnDevice(img.device):
        timer = KernelTimer.active
        rc = timer.launch("msda_fwd_fused", img.device, call) if timer else call()
    if rc == -5:  # MSDA_ERR_UNSUPPORTED
        return None
    _lib.check(rc, f"msda_fwd_fused_{suf}")
    return out


def msda_hip_bwd_fused(out_grad, img, img_shapes, proj, reference_points, padding_mode, align_corners,
                       need_img: bool = True, level_cells: int = 0, need_ref: bool = True):
    """Backward of the module core with the prologue's chain rule done in the kernel: returns
    ``(img_grad | None, proj_grad, reference_points_grad)``, or None when the library declines (L*P too large
    for one pass; nothing was launched)."""
    B, I, H, D = img.shape
    _, Q, _, L, P, _ = proj.shape
    ref_dim = reference_points.shape[-1]
    _check_devices(img, img_shapes, proj, reference_points, out_grad)
    pad = _padding_code(padding_mode)
    cdt = proj.dtype
    suf = _fused_suffix_for(img.dtype, cdt, reference_points.dtype)
    storage = fused_storage_dtypes(img.dtype, cdt, reference_points.dtype)  # (arithmetic and reference points fp32)
    (img, vrow), proj, reference_points = _value_rows(img), proj.contiguous(), reference_points.contiguous()
    out_grad = out_grad.contiguous()
    if out_grad.dtype != cdt:
        out_grad = out_grad.to(cdt)
    shapes = _shapes_i64(img_shapes)
    kw = dict(dtype=cdt, device=img.device)
    g_img = torch.empty((B, I, H, D), dtype=img.dtype, device=img.device) if need_img else None
    g_proj = torch.empty((B, Q, H, L, P, 3), **kw)
    g_ref_part = torch.empty((B, Q, H, ref_dim), dtype=reference_points.dtype, device=img.device)
    lib = _lib.load()
    fn = getattr(lib, f"msda_bwd_fused_{suf}")
    ws, ws_bytes = None, 0
    level_cells = int(level_cells)  # the level-size bound (level_cells_of), an argument of the size query and the launch
    if need_img:  # (a frozen value pyramid needs no workspace at all — ADVICE r04)
        ws_bytes = int(lib.msda_bwd_fused_workspace_bytes(B, I, H, D, Q, L, P, 4 if storage else proj.element_size(),
                                                          img.element_size(), level_cells, 0))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=img.device)

    def call():
        return fn(out_grad.data_ptr(), img.data_ptr(), shapes.data_ptr(), proj.data_ptr(), reference_points.data_ptr(),
                  g_img.data_ptr() if need_img else None, g_proj.data_ptr(), g_ref_part.data_ptr(),
                  B, I, H, D, Q, L, P, ref_dim, pad, int(bool(align_corners)), level_cells, vrow,
                  ws.data_ptr() if ws is not None else None, ws_bytes, _stream_ptr(img.device))

    with _OnDevice(img.device):
        timer = KernelTimer.active
        rc = timer.launch("msda_bwd_fused", img.device, call) if timer else call()
    if rc == -5:  # MSDA_ERR_UNSUPPORTED
        return None
    _lib.check(rc, f"msda_bwd_fused_{suf}")
    # (the kernel leaves per-head partials of grad_reference_points; their sum is a launch of its own — 16 us at the c2
    # shape — and reference points rarely require a gradient)
    return g_img, g_proj, (g_ref_part.sum(dim=2) if need_ref else None)


class _HipFusedModuleCoreFunction(Function):
    """value, raw projection, reference points -> attended values.  Forward and backward are one fused kernel
    (pipeline) each: the softmax / sampling-point prologue and its chain rule run inside the HIP kernels.  When the
    library declines (L*P too large for one LDS pass) the prologue is done in PyTorch around the plain operator."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, img, img_shapes, proj, reference_points, padding_mode, align_corners, level_cells=0):
        ctx.level_cells = int(level_cells)
        out = msda_hip_fwd_fused(img, img_shapes, proj, reference_points, padding_mode, align_corners)
        ctx.fused = out is not None  # the backward has the same L*P limit: do not ask twice
        if out is None:
            # (16-bit storage next to fp32 reference points: the prologue in fp32, the mixed-storage operator)
            pts, att = module_sampling_inputs(proj.to(reference_points.dtype), img_shapes, reference_points)
            out = msda_hip_fwd(img, img_shapes, pts, att, padding_mode, align_corners).to(proj.dtype)
        ctx.save_for_backward(img, img_shapes, proj, reference_points)
        ctx.padding_mode, ctx.align_corners = padding_mode, align_corners
        return out

    @staticmethod
    @once_differentiable
    @custom_bwd(device_type="cuda")
    def backward(ctx, out_grad):
        img, img_shapes, proj, reference_points = ctx.saved_tensors
        need_img, _, need_proj, need_ref = ctx.needs_input_grad[:4]
        if ctx.fused and (need_proj or need_ref):
            res = msda_hip_bwd_fused(out_grad, img, img_shapes, proj, reference_points, ctx.padding_mode,
                                     ctx.align_corners, need_img, level_cells=ctx.level_cells, need_ref=need_ref)
            if res is not None:
                g_img, g_proj, g_ref = res
                return g_img, None, (g_proj if need_proj else None), (g_ref if need_ref else None), None, None, None
        with torch.enable_grad():
            proj_ = proj.detach().to(reference_points.dtype).requires_grad_(need_proj)
            ref_ = reference_points.detach().requires_grad_(need_ref)
            pts, att = module_sampling_inputs(proj_, img_shapes, ref_)
        out_grad = out_grad.to(pts.dtype)
        need_sample = need_proj or need_ref
        g_img, g_pts, g_att = msda_hip_bwd(out_grad, img, img_shapes, pts.detach(), att.detach(), ctx.padding_mode,
                                           ctx.align_corners, (need_img, need_sample, need_sample),
                                           level_cells=ctx.level_cells)
        g_proj = g_ref = None
        if need_sample:
            wrt = [t for t, n in ((proj_, need_proj), (ref_, need_ref)) if n]
            grads = list(torch.autograd.grad([pts, att], wrt, [g_pts, g_att], allow_unused=True))
            if need_proj:
                g_proj = grads.pop(0).to(proj.dtype)
            if need_ref:
                g_ref = grads.pop(0)
        return g_img, None, g_proj, g_ref, None, None, None


def fused_module_core(img, img_shapes, proj, reference_points, padding_mode, align_corners, level_shapes=None) -> torch.Tensor:
    """``multiscale_deformable_attention(img, img_shapes, *module_sampling_inputs(proj, ...))`` — on GPU tensors
    with the prologue fused into the forward kernel; on host tensors exactly that composition.  ``level_shapes``: the
    level sizes as host numbers, optional (:func:`level_cells_of`)."""
    level_cells = level_cells_of(level_shapes, img_shapes.shape[0], img.shape[1])
    if img.device.type == "cuda" and img_shapes.device != img.device:
        # the level table is a handful of integers: follow `img` (the reference's module accepts a host-resident
        # img_shapes next to GPU tensors through its fallback, frontend.py:170-172)
        img_shapes = img_shapes.to(img.device)
    if img.device.type == "cuda":
        _check_devices(img, proj, reference_points)
    floating = img.is_floating_point() and proj.is_floating_point() and reference_points.is_floating_point()
    if img.device.type == "cuda" and floating and _autocast_on() and not torch.compiler.is_compiling():
        # under autocast the op computes in fp32 whatever the projections' dtypes are (custom_fwd casts every floating
        # input, frontend.py:111): mixed bf16 projections / fp32 reference points still take the fused kernels
        _padding_code(padding_mode)
        return _HipFusedModuleCoreFunction.apply(img, img_shapes, proj, reference_points, padding_mode,
                                                 bool(align_corners), level_cells)
    if img.device.type == "cuda" and floating and fused_storage_dtypes(img.dtype, proj.dtype, reference_points.dtype) and \
            not torch.compiler.is_compiling():
        # 16-bit value pyramid and projection (what autocast's GEMMs produce) next to fp32 reference points: fp32
        # arithmetic, 16-bit result and gradients — no fp32 copy of either tensor (msda_*_fused_f32_sbf16 / _sf16)
        _padding_code(padding_mode)
        return _HipFusedModuleCoreFunction.apply(img, img_shapes, proj, reference_points, padding_mode,
                                                 bool(align_corners), level_cells)
    if img.device.type == "cuda" and dtypes_supported(img.dtype, proj.dtype) and \
            reference_points.dtype == proj.dtype and not torch.compiler.is_compiling():
        pad = _padding_code(padding_mode)
        ext = _ext.load()
        if ext is not None and KernelTimer.active is None and not _autocast_on():
            # C++ autograd glue (see hip_multiscale_deformable_attention); only when the fused kernels take this L*P
            B, I, H, D = img.shape
            key = (D, proj.element_size())
            limit = _FUSED_LP_LIMIT.get(key)
            if limit is None:
                limit = _FUSED_LP_LIMIT[key] = int(ext.fused_lp_limit(*key))
            if proj.dim() == 6 and proj.shape[3] * proj.shape[4] <= limit and proj.shape[-1] == 3 \
                    and reference_points.dim() == 3 and reference_points.shape[-1] in (2, 4) \
                    and tuple(proj.shape[:3]) == (B, reference_points.shape[1], H) \
                    and reference_points.shape[0] == B and tuple(img_shapes.shape) == (proj.shape[3], 2):
                _shapes_i64(img_shapes)
                return ext.msda_fused(img, img_shapes, proj, reference_points, pad, bool(align_corners), level_cells)
        return _HipFusedModuleCoreFunction.apply(img, img_shapes, proj, reference_points, padding_mode,
                                                 bool(align_corners), level_cells)
    if img.device.type == "cuda" and torch.compiler.is_compiling() and \
            ((dtypes_supported(img.dtype, proj.dtype) and reference_points.dtype == proj.dtype) or
             fused_storage_dtypes(img.dtype, proj.dtype, reference_points.dtype)):
        from . import compile_op  # traced: keep the fused kernels as one custom op per direction
        if compile_op.fused_lp_ok(img, proj, reference_points):
            return compile_op.compiled_fused_module_core(img, img_shapes, proj, reference_points, padding_mode,
                                                         align_corners, level_cells)
    pts, att = module_sampling_inputs(proj, img_shapes, reference_points)
    return multiscale_deformable_attention(img, img_shapes, pts, att, padding_mode, align_corners, level_shapes=level_shapes)


# ------------------------------------------------------------------------------------------
# host-tensor path (reference: native_multiscale_deformable_attention, frontend.py:15-68)
# ------------------------------------------------------------------------------------------
def _unnormalise(coord: torch.Tensor, size: int, padding_mode: str, align_corners: bool) -> torch.Tensor:
    pix = coord * (size - 1) if align_corners else coord * size - 0.5
    if padding_mode == "border":
        # grid_sample clips the coordinate to [0, size-1] and kills its gradient at and beyond the border
        inside = (pix > 0) & (pix < size - 1)
        pix = torch.where(inside, pix, pix.detach().clamp(0, size - 1))
    return pix


def native_multiscale_deformable_attention(
    img: torch.Tensor,
    img_shapes: torch.Tensor,
    sampling_points: torch.Tensor,
    attention_weights: torch.Tensor,
    padding_mode: Literal["border", "zeros"],
    align_corners: bool,
) -> torch.Tensor:
    """Plain-PyTorch path for host tensors, differentiable through autograd (the reference's documented CPU
    behaviour, frontend.py:15-68).

    Per level, every (batch, head) plane is one ``[D, h, w]`` image for ``F.grid_sample`` — PyTorch's vectorised
    bilinear sampler, forward and backward — and the level's samples are folded into the result straight away by a
    contraction with the attention weights over the point axis, so the ``[B, Q, H, L, P, D]`` sample tensor is never
    built.  Like the reference it reads ``img_shapes`` on the host.  ``_gather_multiscale_deformable_attention`` below
    states the same operator without ``grid_sample`` (tests hold the two to each other and to the reference's golden
    vectors).
    """
    _padding_code(padding_mode)
    B, I, H, D, Q, L, P = _dims(img, sampling_points, attention_weights, img_shapes)
    dt = torch.result_type(img, sampling_points)
    planes = img.to(dt).permute(0, 2, 3, 1).reshape(B * H, D, I)          # [B*H, D, I]: a plane's channels as images
    grid = (2 * sampling_points.to(dt) - 1).permute(0, 2, 3, 1, 4, 5)      # [B, H, L, Q, P, 2] in grid_sample's [-1, 1]
    weights = attention_weights.to(dt).permute(0, 2, 3, 1, 4)              # [B, H, L, Q, P]
    out = None
    start = 0
    for lvl, (h, w) in enumerate(img_shapes.tolist()):
        level = planes[:, :, start:start + h * w].reshape(B * H, D, h, w)
        sampled = torch.nn.functional.grid_sample(level, grid[:, :, lvl].reshape(B * H, Q, P, 2), mode="bilinear",
                                                  padding_mode=padding_mode, align_corners=align_corners)  # [B*H, D, Q, P]
        part = torch.einsum("ndqp,nqp->nqd", sampled, weights[:, :, lvl].reshape(B * H, Q, P))
        out = part if out is None else out + part
        start += h * w
    if out is None:
        out = planes.new_zeros((B * H, Q, D))
    return out.reshape(B, H, Q, D).permute(0, 2, 1, 3).contiguous()


def _gather_multiscale_deformable_attention(img, img_shapes, sampling_points, attention_weights, padding_mode, align_corners):
    """The operator as index arithmetic + ``gather`` (no ``grid_sample``): an independent statement of SURVEY 9.1, kept
    as a cross-check of the host path (slow: its backward is four ``scatter_add`` per level)."""
    _padding_code(padding_mode)
    B, I, H, D, Q, L, P = _dims(img, sampling_points, attention_weights, img_shapes)
    planes = img.permute(0, 2, 1, 3)  # [B, H, I, D]
    out = img.new_zeros((B, H, Q, D), dtype=torch.result_type(img, sampling_points))
    start = 0
    for lvl, (h, w) in enumerate(img_shapes.tolist()):
        plane = planes[:, :, start:start + h * w]                      # [B, H, h*w, D]
        pts = sampling_points[:, :, :, lvl].permute(0, 2, 1, 3, 4)     # [B, H, Q, P, 2]
        att = attention_weights[:, :, :, lvl].permute(0, 2, 1, 3)      # [B, H, Q, P]
        px = _unnormalise(pts[..., 0], w, padding_mode, align_corners)
        py = _unnormalise(pts[..., 1], h, padding_mode, align_corners)
        x0, y0 = px.detach().floor(), py.detach().floor()
        dx, dy = px - x0, py - y0
        level_sample = 0
        for yo, xo, wgt in ((0, 0, (1 - dy) * (1 - dx)), (0, 1, (1 - dy) * dx), (1, 0, dy * (1 - dx)), (1, 1, dy * dx)):
            xi, yi = x0 + xo, y0 + yo
            valid = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1)
            idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).long()            # [B, H, Q, P]
            rows = plane.gather(2, idx.reshape(B, H, Q * P, 1).expand(-1, -1, -1, D)).reshape(B, H, Q, P, D)
            level_sample = level_sample + rows * (wgt * valid.to(wgt.dtype)).unsqueeze(-1)
        out = out + (level_sample * att.unsqueeze(-1)).sum(dim=3)
        start += h * w
    return out.permute(0, 2, 1, 3).contiguous()


# ------------------------------------------------------------------------------------------
# public entry (reference: multiscale_deformable_attention, frontend.py:145-172)
# ------------------------------------------------------------------------------------------
def multiscale_deformable_attention(
    img: torch.Tensor,
    img_shapes: torch.Tensor,
    sampling_points: torch.Tensor,
    attention_weights: torch.Tensor,
    padding_mode: Literal["border", "zeros"],
    align_corners: bool,
    level_shapes=None,
) -> torch.Tensor:
    """Differentiable multiscale deformable attention.

    Args:
        img: flattened image pyramid ``[batch, num_image, num_heads, num_channels]`` with
            ``num_image = sum(h*w)`` over the levels.
        img_shapes: ``[num_levels, 2]`` integer tensor of (height, width).
        sampling_points: ``[batch, num_queries, num_heads, num_levels, num_points, 2]`` in (x, y)
            order, normalised to [0, 1] with (0, 0) the top-left corner.
        attention_weights: ``[batch, num_queries, num_heads, num_levels, num_points]``.
        padding_mode: ``"border"`` (clamp to the nearest pixel) or ``"zeros"``.
        align_corners: grid alignment, as in ``torch.nn.functional.grid_sample``.
        level_shapes: optional, not in the reference — the same (height, width) pairs as HOST numbers (e.g. Hugging
            Face's ``spatial_shapes_list``).  ``img_shapes`` lives on the device and is never read back, so without
            this the backward must assume the largest level ``num_image`` pixels can form; with it, decoder-sized
            calls over real-image pyramids take the single-launch grad_value kernel (see :func:`level_cells_of`).

    Returns:
        ``[batch, num_queries, num_heads, num_channels]``.

    Tensors on an AMD GPU ("cuda" device type on ROCm) run the hand-written gfx950 kernels and
    never fall back; host tensors run the plain-PyTorch formulation.
    """
    if img.device.type == "cuda":
        return hip_multiscale_deformable_attention(
            img, img_shapes, sampling_points, attention_weights, padding_mode, align_corners, level_shapes)
    return native_multiscale_deformable_attention(
        img, img_shapes, sampling_points, attention_weights, padding_mode, align_corners)
