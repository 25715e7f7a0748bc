"""Query-sharded multi-GPU execution: one process per GPU, ``torch.distributed`` (RCCL over xGMI).

The reference has no multi-device code; every (b, q, h) of the operator is independent
(/root/reference/src/msda_triton/kernels.py:18-21), so the queries shard trivially:

  forward   each rank runs the HIP kernels on its rows against the (replicated) value pyramid and writes them
            straight into their place in the full ``[B, Q, H, D]`` result; the peers' rows arrive in place too —
            one in-place all-gather, or grouped point-to-point pieces that overlap the next piece's kernels;
  backward  grad_sampling_points / grad_attention_weights are shard-local (no communication);
            grad_value is a sum over the queries of a batch element: summed among the ranks that share that
            batch element (nothing to do while the ranks divide B), or all-reduced.

With 8 GPUs on one node a rank sends ``B*Q*H*D*s/8`` bytes to every peer over its own point-to-point xGMI link
(c4 0.9 MB, c5 51 MB per rank); the exchange is ONE in-place all-gather by default (pieces overlapped with compute only where
the forward is long enough to hide a piece's exchange: :func:`default_overlap_chunks`).  The launches of a rank's row range
go through the C++ binding (``csrc/msda_torch_ext.cpp``: ``rows_forward`` / ``rows_backward``; the whole operator is the
C++ autograd node ``msda_rows`` when there is nothing to exchange), the collectives stay in Python.  Every collective runs
over RCCL in ``tests/test_gpu_nccl.py`` (an in-process one-rank ``nccl`` group; ``loopback=True``).

Two partitions are offered:

  rows     (:func:`row_sharded_multiscale_deformable_attention`, what ``bench.py --gpus N`` runs) — contiguous
           ranges of the flattened ``[B*Q]`` row space.  The per-rank output slices are contiguous in the final
           ``[B, Q, H, D]`` tensor: the kernels write them in place and the exchange (one ``all_gather_into_tensor``,
           or grouped point-to-point pieces that overlap the next piece's kernels) lands the peers' rows in place
           too — no local tensor, padding or concatenation; a rank touches only the batch elements its rows fall
           into, so grad_value needs no communication at all when the ranks divide B, and otherwise only a sum
           among the ranks that share a batch element.
  queries  (:func:`sharded_multiscale_deformable_attention`) — every rank takes the same query range of every
           batch element (sequence-parallel style callers that already hold ``[B, Q/N, ...]`` slices).
"""
from __future__ import annotations

from typing import Literal, Optional, Tuple

import warnings
import weakref

import torch
import torch.distributed as dist
from torch.autograd.function import Function

from . import _lib
from .functional import check_backward_supported, multiscale_deformable_attention


def shard_bounds(num_queries: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [begin, end) of the query axis owned by ``rank`` (equal ceil-sized shards,
    the last ones may be short or empty)."""
    per = -(-num_queries // world_size)
    begin = min(num_queries, rank * per)
    return begin, min(num_queries, begin + per)


def _all_gather_into(buf: torch.Tensor, local: torch.Tensor, group) -> None:
    try:
        dist.all_gather_into_tensor(buf, local, group=group)
    except (RuntimeError, NotImplementedError):  # backend without the fused form
        dist.all_gather(list(buf.unbind(0)), local, group=group)


class _GatherQueryShards(Function):
    """local [B, per, H, D] on every rank  ->  full [B, Q, H, D] on every rank.

    Backward takes this rank's slice of the incoming gradient: correct when every rank evaluates
    the same downstream computation on the gathered output (replicated consumers, the
    Grounding-DINO decoder case); use ``grad_sync="reduce_scatter"`` when consumers differ per rank.
    """

    @staticmethod
    def forward(ctx, local: torch.Tensor, num_queries: int, group, grad_sync: str):
        world = dist.get_world_size(group)
        ctx.group, ctx.world, ctx.rank = group, world, dist.get_rank(group)
        ctx.num_queries, ctx.grad_sync = num_queries, grad_sync
        B, per, H, D = local.shape
        ctx.per = per
        local = local.contiguous()
        if B <= 8:
            # gather batch element by batch element straight into the final [B, world*per, H, D] layout:
            # B small collectives instead of one collective plus a transposing copy of the whole output
            full = local.new_empty((B, world * per, H, D))
            for b in range(B):
                _all_gather_into(full[b].view(world, per, H, D), local[b], group)
        else:
            buf = local.new_empty((world, B, per, H, D))
            _all_gather_into(buf, local, group)
            full = buf.permute(1, 0, 2, 3, 4).reshape(B, world * per, H, D)
        return full if world * per == num_queries else full[:, :num_queries].contiguous()

    @staticmethod
    def backward(ctx, grad_full: torch.Tensor):
        B, Q, H, D = grad_full.shape
        per, world = ctx.per, ctx.world
        if Q < world * per:
            grad_full = torch.nn.functional.pad(grad_full, (0, 0, 0, 0, 0, world * per - Q))
        if ctx.grad_sync == "reduce_scatter":
            shards = grad_full.reshape(B, world, per, H, D).permute(1, 0, 2, 3, 4).contiguous()
            mine = torch.empty_like(shards[0])
            dist.reduce_scatter_tensor(mine, shards.reshape(world * B, per, H, D), group=ctx.group)
        else:
            mine = grad_full[:, ctx.rank * per:(ctx.rank + 1) * per].contiguous()
        return mine, None, None, None


class _ReplicatedValue(Function):
    """Identity in forward; sums the gradient over the ranks in backward (value is shared by all
    query shards, so its gradient is the sum of the per-shard scatter results)."""

    @staticmethod
    def forward(ctx, value: torch.Tensor, group):
        ctx.group = group
        return value.view_as(value)

    @staticmethod
    def backward(ctx, grad: torch.Tensor):
        grad = grad.contiguous()
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=ctx.group)
        return grad, None


def sharded_multiscale_deformable_attention(
    img: torch.Tensor,
    img_shapes: torch.Tensor,
    sampling_points: torch.Tensor,
    attention_weights: torch.Tensor,
    padding_mode: Literal["border", "zeros"],
    align_corners: bool,
    group: Optional[dist.ProcessGroup] = None,
    inputs_are_sharded: bool = False,
    num_queries: Optional[int] = None,
    grad_sync: Literal["slice", "reduce_scatter"] = "slice",
) -> torch.Tensor:
    """Query-sharded operator; returns the full ``[B, Q, H, D]`` output on every rank.

    ``inputs_are_sharded=False``: ``sampling_points`` / ``attention_weights`` hold all Q queries
    (replicated) and each rank computes its ``shard_bounds`` slice.
    ``inputs_are_sharded=True``: they already hold only this rank's slice (``num_queries`` = global
    Q is then required, and every rank's slice must have the ceil-sized shard length except that
    trailing ranks may be shorter).
    ``img`` is the full value pyramid on every rank.
    """
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("torch.distributed is not initialised; call init_process_group first")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if inputs_are_sharded:
        if num_queries is None:
            raise ValueError("num_queries (global) is required when inputs_are_sharded=True")
        Q = int(num_queries)
        pts, att = sampling_points, attention_weights
    else:
        Q = sampling_points.shape[1]
        begin, end = shard_bounds(Q, world, rank)
        pts, att = sampling_points[:, begin:end], attention_weights[:, begin:end]
    per = -(-Q // world)
    if img.requires_grad:
        img = _ReplicatedValue.apply(img, group)
    local = multiscale_deformable_attention(img, img_shapes, pts, att, padding_mode, align_corners)
    if local.shape[1] < per:  # short / empty trailing shard: pad so the all-gather is regular
        local = torch.nn.functional.pad(local, (0, 0, 0, 0, 0, per - local.shape[1]))
    return _GatherQueryShards.apply(local, Q, group, grad_sync)


# ---------------------------------------------------------------------------------------------
# row partition: contiguous ranges of the flattened (b, q) row space
# ---------------------------------------------------------------------------------------------
def row_shard_bounds(num_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Rows [begin, end) of the flattened ``B*Q`` row space owned by ``rank`` (ceil-sized shards)."""
    return shard_bounds(num_rows, world_size, rank)


def row_segments(num_queries: int, r0: int, r1: int):
    """Split rows [r0, r1) into per-batch-element pieces ``(b, q0, q1)`` (row = b * Q + q)."""
    segs = []
    r = r0
    while r < r1:
        b, q0 = divmod(r, num_queries)
        q1 = min(num_queries, q0 + (r1 - r))
        segs.append((b, q0, q1))
        r += q1 - q0
    return segs


_OWNER_GROUPS: dict = {}
_WARNED_OWNERS_DOWNGRADE = False
_OWNER_GROUPS_OF = None  # weakref to the default process group the cached sub-groups were made under


def _owner_groups(B: int, Q: int, group):
    """For every batch element touched by more than one rank: (ranks, process group).

    ``dist.new_group`` is collective over the DEFAULT process group: every rank of the job has to enter it, in the
    same order.  So the sub-groups are only made when ``group`` is the default group (every rank of the job runs the
    operator); for a caller-supplied sub-group the function returns None and the caller falls back to an all-reduce on
    ``group`` (ranks outside it never call the operator, so they could not take part in the group creation).
    Groups made before a ``destroy_process_group`` / ``init_process_group`` cycle are never handed out again."""
    default = dist.group.WORLD
    if group is not None and group is not default:
        return None
    world = dist.get_world_size(group)
    # The cache lives and dies with the default group OBJECT (a weak reference, not its id(): after a
    # destroy_process_group / init_process_group cycle the new default group can be allocated at the freed one's address)
    global _OWNER_GROUPS_OF
    if _OWNER_GROUPS_OF is None or _OWNER_GROUPS_OF() is not default:
        _OWNER_GROUPS.clear()
        _OWNER_GROUPS_OF = weakref.ref(default)
    key = (B, Q, world)
    if key in _OWNER_GROUPS:
        return _OWNER_GROUPS[key]
    rows = B * Q
    owners = []
    for b in range(B):
        ranks = [r for r in range(world)
                 if max(row_shard_bounds(rows, world, r)[0], b * Q) < min(row_shard_bounds(rows, world, r)[1], (b + 1) * Q)]
        owners.append(ranks)
    made: dict = {}
    out = []
    for b, ranks in enumerate(owners):
        if len(ranks) <= 1:
            out.append((ranks, None))
            continue
        t = tuple(ranks)
        if t not in made:
            glob = [dist.get_global_rank(group, r) for r in ranks] if group is not None else list(ranks)
            made[t] = dist.new_group(ranks=glob)
        out.append((ranks, made[t]))
    _OWNER_GROUPS[key] = out
    return out


# What the piece rule assumes about the machine (one MI355X node, measured where it could be: the gather rate on one GPU;
# the link rate is the point-to-point xGMI figure a ring / direct exchange sustains in one direction)
_GATHER_BYTES_PER_S = 30e12   # logical bytes of bilinear rows the forward gathers per second (c2: 41 TB/s, c5: 27)
_LINK_BYTES_PER_S = 45e9      # one direction of one xGMI link


def default_overlap_chunks(rows: int, world: int, gather_bytes_per_row: int = 0, out_bytes_per_row: int = 0) -> int:
    """Pieces a rank's rows are computed and exchanged in (the same number on every rank: it follows the nominal shard
    size).  Piece k's exchange runs while piece k + 1 computes, so pieces can hide at most the FORWARD's own time — and
    they cost kernel time: measured on one GPU (bench.py ``shard_compute_bound``, round 5) c2's 40 000 rows per rank in
    4 pieces take 0.095 ms of forward against 0.068 ms in one piece, in 8 pieces 0.15 ms.

    So pieces are used only where the forward CAN hide a piece's exchange: its time per row (``gather_bytes_per_row`` =
    4 * L * P * H * D * element size, at the measured gather rate) is at least half of what the row's result takes over one
    xGMI link (``out_bytes_per_row`` = H * D * element size).  For this operator that is rare: a row of the result is
    produced in 2-6 ns and travels for 25 (c2: ratio 0.09, c5: 0.25 — one piece, one in-place all-gather); it takes
    L * P >= ~170 samples per unit.  Then: pieces of >= 8 192 rows, at most 8.  Without the two sizes (0): one piece."""
    if world <= 1 or gather_bytes_per_row <= 0 or out_bytes_per_row <= 0:
        return 1
    t_compute = gather_bytes_per_row / _GATHER_BYTES_PER_S
    t_exchange = out_bytes_per_row / _LINK_BYTES_PER_S
    if t_compute < 0.5 * t_exchange:
        return 1
    return max(1, min(8, -(-rows // world) // 8192))


def _chunk_bounds(n: int, chunks: int, k: int) -> Tuple[int, int]:
    """Rows [begin, end) of chunk ``k`` when ``n`` rows are cut into ``chunks`` ceil-sized pieces."""
    cs = -(-n // chunks) if chunks > 0 else n
    begin = min(n, k * cs)
    return begin, min(n, begin + cs)


def _run_pieces(fn, num_queries: int, row0: int, row1: int):
    """Call ``fn(b, nb, q0, q1, rows_before)`` for the pieces of the row range [row0, row1): runs of whole batch
    elements become ONE piece (nb > 1), a partial batch element its own."""
    segs = row_segments(num_queries, row0, row1)
    i, done = 0, 0
    while i < len(segs):
        b, q0, q1 = segs[i]
        nb = 1
        if q0 == 0 and q1 == num_queries:
            while i + nb < len(segs) and segs[i + nb][1] == 0 and segs[i + nb][2] == num_queries:
                nb += 1
        fn(b, nb, q0, q1, done)
        done += (q1 - q0) if nb == 1 else nb * num_queries
        i += nb


def _rows_ext(img: torch.Tensor):
    """The C++ binding's row-range launchers for GPU tensors (one call per chunk instead of a ctypes call per piece),
    or None: host tensors, no binding, or bench.py's per-kernel timer (which brackets the ctypes launchers)."""
    if img.device.type != "cuda":
        return None
    from . import _ext
    from .functional import KernelTimer
    return _ext.load() if KernelTimer.active is None else None


class _RowShardedMSDA(Function):
    """The sharded operator with its exchange as ONE autograd node (ranks > 1; one rank, or one rank played on one GPU,
    takes the C++ node ``msda_rows`` of csrc/msda_torch_ext.cpp instead — same launches, no exchange).

    forward: this rank's rows are computed in ``chunks`` pieces, each written by the kernel STRAIGHT into its place
    in the full ``[B*Q, H, D]`` result (no local tensor, no padding, no concatenation); as soon as a piece is
    enqueued its exchange starts — every rank sends the piece to every peer and receives the peers' pieces into
    their final rows (grouped point-to-point: on the xGMI full mesh each pair uses its own link) — while the next
    piece computes.  Equal shards and ``chunks == 1`` take one ``all_gather_into_tensor`` instead.
    backward: this rank's rows of the incoming gradient (replicated consumers) or a reduce-scatter; the local
    backward kernels; grad_value summed as ``grad_value_sync`` says.  Every rank runs every collective, whatever its
    shard holds (an empty shard computes nothing and still takes part).

    ``loopback`` (tests on ONE GPU): a one-rank group runs every collective of the N-rank code against itself — the
    in-place all-gather, or the pieces sent to and received from its own rank through ``batch_isend_irecv`` into a
    second buffer that becomes the result; the reduce-scatter and the grad_value sums over the one-rank group — so
    that the RCCL calls, their views and their stream ordering are exercised where only one GPU is at hand.
    """

    @staticmethod
    def forward(ctx, img, img_shapes, pts_rows, att_rows, padding_mode, align_corners, num_queries, group,
                grad_value_sync, grad_sync, owner_groups, chunks, compute_only_as=None, loopback=False):
        # compute_only_as = (world, rank): this process plays ONE rank of a larger job with every exchange left out
        # (row_sharded_multiscale_deformable_attention: the compute half of the scaling model on one GPU)
        emulated = compute_only_as is not None
        world, rank = compute_only_as if emulated else (dist.get_world_size(group), dist.get_rank(group))
        loopback = bool(loopback) and world == 1 and not emulated
        B, _, H, D = img.shape
        Q = int(num_queries)
        rows = B * Q
        bounds = [row_shard_bounds(rows, world, r) for r in range(world)]
        r0, r1 = bounds[rank]
        gpu = img.device.type == "cuda"
        # (the C++ launchers only for arguments the kernels take — one GPU, a supported dtype combination; anything else
        #  goes through the Python launchers, whose checks carry the error messages)
        ext = _rows_ext(img) if _rows_node_ok(img, img_shapes, pts_rows, att_rows, padding_mode) else None
        if ext is not None:  # (dense tensors, int64 level sizes)
            img, pts_rows, att_rows = img.contiguous(), pts_rows.contiguous(), att_rows.contiguous()
            img_shapes = img_shapes.to(torch.int64).contiguous()
            pad_code = _lib.PADDING_MODES[padding_mode]
        full = pts_rows.new_empty((rows, H, D))  # the sampling inputs' dtype (fp32 next to a 16-bit pyramid)
        echo = torch.empty_like(full) if loopback else None  # loopback: where the rows sent to myself arrive
        equal = all(e - b == bounds[0][1] - bounds[0][0] for b, e in bounds)
        chunks = max(1, int(chunks))
        pending = []
        for k in range(chunks):
            c0, c1 = _chunk_bounds(r1 - r0, chunks, k)

            def piece(b, nb, q0, q1, before, c0=c0):
                n = (q1 - q0) if nb == 1 else nb * Q
                at = c0 + before
                pts = pts_rows[at:at + n].reshape(nb, n // nb, *pts_rows.shape[1:])
                att = att_rows[at:at + n].reshape(nb, n // nb, *att_rows.shape[1:])
                dst = full[r0 + at:r0 + at + n].view(nb, n // nb, H, D)
                if gpu:
                    from .functional import msda_hip_fwd
                    msda_hip_fwd(img[b:b + nb], img_shapes, pts, att, padding_mode, align_corners, out=dst)
                else:
                    with torch.no_grad():
                        dst.copy_(multiscale_deformable_attention(img[b:b + nb], img_shapes, pts, att, padding_mode,
                                                                  align_corners))

            if ext is not None:  # every piece of the chunk in one call
                ext.rows_forward(img, img_shapes, pts_rows, att_rows, r0, full, r0 + c0, r0 + c1, Q, pad_code,
                                 bool(align_corners))
            else:
                _run_pieces(piece, Q, r0 + c0, r0 + c1)
            if emulated or (world == 1 and not loopback):
                continue
            if chunks == 1 and equal:
                per = r1 - r0
                _all_gather_into(full.view(world, per, H, D), full[r0:r1], group)  # in place: my rows are already there
                continue
            ops = []
            for peer in range(world):
                if peer == rank and not loopback:
                    continue
                gpeer = dist.get_global_rank(group, peer) if group is not None else peer
                if c1 > c0:
                    ops.append(dist.P2POp(dist.isend, full[r0 + c0:r0 + c1], gpeer, group))
                p0, p1 = bounds[peer]
                pc0, pc1 = _chunk_bounds(p1 - p0, chunks, k)
                if pc1 > pc0:
                    ops.append(dist.P2POp(dist.irecv, (echo if loopback else full)[p0 + pc0:p0 + pc1], gpeer, group))
            if ops:
                pending.extend(dist.batch_isend_irecv(ops))
        for req in pending:
            req.wait()
        if loopback and not (chunks == 1 and equal):
            full = echo  # what came back through the communicator
        ctx.save_for_backward(img, img_shapes, pts_rows, att_rows)
        ctx.meta = (padding_mode, align_corners, Q, group, grad_value_sync, grad_sync, owner_groups, world, rank, r0, r1)
        ctx.emulated, ctx.loopback = emulated, loopback
        return full.view(B, Q, H, D)

    @staticmethod
    def backward(ctx, grad_full):
        img, img_shapes, pts_rows, att_rows = ctx.saved_tensors
        padding_mode, align_corners, Q, group, grad_value_sync, grad_sync, owner_groups, world, rank, r0, r1 = ctx.meta
        B, _, H, D = img.shape
        rows = B * Q
        need_img, _, need_pts, need_att = ctx.needs_input_grad[:4]
        g_rows = grad_full.reshape(rows, H, D)
        if ctx.emulated:
            grad_value_sync, grad_sync = "none", "slice"
        comm = (world > 1 or ctx.loopback) and not ctx.emulated
        if grad_sync == "reduce_scatter" and comm:
            per = -(-rows // world)
            padded = g_rows if rows == world * per else torch.nn.functional.pad(g_rows, (0, 0, 0, 0, 0, world * per - rows))
            mine = torch.empty_like(padded[:per])
            dist.reduce_scatter_tensor(mine, padded.contiguous(), group=group)
            mine = mine[:r1 - r0]
        else:
            mine = g_rows[r0:r1]
        gpu = img.device.type == "cuda"
        ext = _rows_ext(img) if _rows_node_ok(img, img_shapes, pts_rows, att_rows, padding_mode) and \
            grad_full.device == img.device else None
        if ext is not None:
            g_img, g_pts, g_att = ext.rows_backward(mine, img, img_shapes, pts_rows, att_rows, r0, r1, Q,
                                                    _lib.PADDING_MODES[padding_mode], bool(align_corners),
                                                    bool(need_img), bool(need_pts), bool(need_att), 0)
        else:
            g_img, g_pts, g_att = _backward_pieces(mine, img, img_shapes, pts_rows, att_rows, padding_mode, align_corners,
                                                   Q, r0, r1, need_img, need_pts, need_att, gpu)
        if need_img and comm:
            if grad_value_sync == "all_reduce":
                dist.all_reduce(g_img, op=dist.ReduceOp.SUM, group=group)
            elif grad_value_sync == "owners":
                for b, (ranks, pg) in enumerate(owner_groups):  # ascending b on every rank: no cyclic waits
                    if pg is not None and rank in ranks:
                        dist.all_reduce(g_img[b], op=dist.ReduceOp.SUM, group=pg)
        return (g_img, None, g_pts, g_att) + (None,) * 10


def _backward_pieces(mine, img, img_shapes, pts_rows, att_rows, padding_mode, align_corners, Q, r0, r1, need_img, need_pts,
                     need_att, gpu):
    """The backward of rows [r0, r1) piece by piece through the Python launchers (host tensors, installations without
    the C++ binding, bench.py's per-kernel timer).  Gradient tensors of the shard; the kernels write every piece
    straight into its slice.  A contiguous row range meets a batch element in at most one piece, so grad_value needs
    no accumulation: the batch elements this rank does not touch are zeroed, the others are written whole."""
    B, _, H, D = img.shape
    g_img = torch.empty_like(img) if need_img else None
    g_pts = torch.empty_like(pts_rows) if need_pts else None
    g_att = torch.empty_like(att_rows) if need_att else None
    touched = [False] * B

    def piece(b, nb, q0, q1, before):
        n = (q1 - q0) if nb == 1 else nb * Q
        pts = pts_rows[before:before + n].reshape(nb, n // nb, *pts_rows.shape[1:])
        att = att_rows[before:before + n].reshape(nb, n // nb, *att_rows.shape[1:])
        go = mine[before:before + n].reshape(nb, n // nb, H, D)
        for k in range(b, b + nb):
            touched[k] = True
        if gpu:
            from .functional import msda_hip_bwd
            need_s = need_pts or need_att
            dst = (g_img[b:b + nb] if need_img else None,
                   g_pts[before:before + n].view(nb, n // nb, *pts_rows.shape[1:]) if need_pts else None,
                   g_att[before:before + n].view(nb, n // nb, *att_rows.shape[1:]) if need_att else None)
            msda_hip_bwd(go, img[b:b + nb], img_shapes, pts, att, padding_mode, align_corners,
                         (need_img, need_s, need_s), out=dst)
        else:
            with torch.enable_grad():
                v_ = img[b:b + nb].detach().requires_grad_(need_img)
                p_ = pts.detach().requires_grad_(need_pts)
                a_ = att.detach().requires_grad_(need_att)
                o_ = multiscale_deformable_attention(v_, img_shapes, p_, a_, padding_mode, align_corners)
            wrt = [t for t, nd in ((v_, need_img), (p_, need_pts), (a_, need_att)) if nd]
            got = list(torch.autograd.grad(o_, wrt, go)) if wrt else []
            if need_img:
                g_img[b:b + nb] = got.pop(0)
            if need_pts:
                g_pts[before:before + n] = got.pop(0).reshape(n, *pts_rows.shape[1:])
            if need_att:
                g_att[before:before + n] = got.pop(0).reshape(n, *att_rows.shape[1:])

    _run_pieces(piece, Q, r0, r1)
    if need_img:
        for b in range(B):
            if not touched[b]:
                g_img[b].zero_()
    return g_img, g_pts, g_att


def row_sharded_multiscale_deformable_attention(
    img: torch.Tensor,
    img_shapes: torch.Tensor,
    sampling_points: torch.Tensor,
    attention_weights: torch.Tensor,
    padding_mode: Literal["border", "zeros"],
    align_corners: bool,
    group: Optional[dist.ProcessGroup] = None,
    inputs_are_sharded: bool = False,
    num_queries: Optional[int] = None,
    grad_value_sync: Literal["all_reduce", "owners", "none"] = "all_reduce",
    grad_sync: Literal["slice", "reduce_scatter"] = "slice",
    overlap_chunks: Optional[int] = None,
    compute_only_as: Optional[Tuple[int, int]] = None,
    loopback: bool = False,
) -> torch.Tensor:
    """Row-sharded operator; returns the full ``[B, Q, H, D]`` output on every rank.

    ``compute_only_as=(world, rank)`` (measurement aid, no process group needed): run exactly what rank ``rank`` of a
    ``world``-rank job computes — its rows, in the same pieces, through the same kernels — and leave every exchange
    out: ``grad_value_sync`` and ``grad_sync`` are then overridden to ``"none"`` / ``"slice"`` whatever was passed.  Only
    that rank's rows of the result (and its share of the gradients) are meaningful.  ``bench.py``'s
    ``shard_compute_bound`` leg times it on one GPU: the speed-up ceiling before any byte crosses xGMI.

    ``overlap_chunks``: pieces the local rows are computed and exchanged in (piece k's exchange overlaps piece k+1's
    kernels); None: :func:`default_overlap_chunks` — 1 (one in-place all-gather) unless the forward is long enough to
    hide a piece's exchange, then up to 8 pieces of at least 8 192 rows.

    ``loopback`` (test aid, one-rank groups only): run the collectives of the N-rank code against the rank itself
    (see :class:`_RowShardedMSDA`).

    ``img`` is ``[B, I, H, D]`` on every rank (a rank only reads the batch elements its rows fall into).
    ``inputs_are_sharded=False``: ``sampling_points [B,Q,H,L,P,2]`` / ``attention_weights [B,Q,H,L,P]`` are
    replicated and each rank computes its :func:`row_shard_bounds` rows.
    ``inputs_are_sharded=True``: they hold only this rank's rows, flattened: ``[rows, H, L, P, 2]`` /
    ``[rows, H, L, P]`` (``num_queries`` = Q per batch element is then required).
    """
    if grad_value_sync not in ("all_reduce", "owners", "none"):
        raise ValueError(f"unknown grad_value_sync {grad_value_sync!r}")
    if grad_sync not in ("slice", "reduce_scatter"):
        raise ValueError(f"unknown grad_sync {grad_sync!r}")
    if compute_only_as is not None:
        world, rank = (int(v) for v in compute_only_as)
        if not 0 <= rank < world:
            raise ValueError(f"compute_only_as={compute_only_as!r}: need 0 <= rank < world")
        compute_only_as = (world, rank)
        grad_value_sync = "none"
    elif not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("torch.distributed is not initialised; call init_process_group first")
    if compute_only_as is None:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    if loopback and (world != 1 or compute_only_as is not None):
        raise ValueError("loopback runs a ONE-rank group's collectives against itself (a test aid)")
    B = img.shape[0]
    if inputs_are_sharded:
        if num_queries is None:
            raise ValueError("num_queries (per batch element) is required when inputs_are_sharded=True")
        Q = int(num_queries)
    else:
        Q = sampling_points.shape[1]
    rows = B * Q
    r0, r1 = row_shard_bounds(rows, world, rank)
    if inputs_are_sharded:
        if sampling_points.shape[0] != r1 - r0:
            raise ValueError(f"rank {rank} owns rows [{r0}, {r1}) but got {sampling_points.shape[0]} rows")
        pts_rows, att_rows = sampling_points, attention_weights
    else:
        pts_rows = sampling_points.reshape(rows, *sampling_points.shape[2:])[r0:r1]
        att_rows = attention_weights.reshape(rows, *attention_weights.shape[2:])[r0:r1]
    if img.device.type == "cuda" and img.requires_grad and torch.is_grad_enabled() and pts_rows.dim() == 5:
        check_backward_supported(img, pts_rows, Q)  # (at forward time, like the direct operator)
    if overlap_chunks is None:
        H, D = img.shape[2], img.shape[3]
        lp = pts_rows.shape[2] * pts_rows.shape[3] if pts_rows.dim() == 5 else 0
        overlap_chunks = default_overlap_chunks(rows, world, 4 * lp * H * D * img.element_size(),
                                                H * D * pts_rows.element_size())
    if (compute_only_as is not None or (world == 1 and not loopback)) and _rows_node_ok(img, img_shapes, pts_rows, att_rows,
                                                                                         padding_mode):
        # nothing to exchange: the whole operator is the C++ node (same launches as _RowShardedMSDA, no Python in the
        # step — at one rank the sharded operator costs what the direct one does)
        from . import _ext
        return _ext.load().msda_rows(img, img_shapes, pts_rows, att_rows, _lib.PADDING_MODES[padding_mode],
                                     bool(align_corners), Q, r0, r1, max(1, int(overlap_chunks)), 0)
    owners = _owner_groups(B, Q, group) if (grad_value_sync == "owners" and img.requires_grad) else None
    if grad_value_sync == "owners" and img.requires_grad and owners is None:
        grad_value_sync = "all_reduce"  # a caller-supplied sub-group: see _owner_groups
        global _WARNED_OWNERS_DOWNGRADE
        if not _WARNED_OWNERS_DOWNGRADE:
            _WARNED_OWNERS_DOWNGRADE = True
            warnings.warn("grad_value_sync='owners' needs sub-groups of the DEFAULT process group; on a caller-supplied "
                          "group grad_value is all-reduced over that group instead", stacklevel=2)
    if loopback and grad_value_sync == "owners" and owners is not None:
        # a one-rank job has no shared batch element: sum every batch element over the one-rank group instead
        owners = [([0], group if group is not None else dist.group.WORLD) for _ in range(B)]
    return _RowShardedMSDA.apply(img, img_shapes, pts_rows, att_rows, padding_mode, bool(align_corners), Q, group,
                                 grad_value_sync, grad_sync, owners, overlap_chunks, compute_only_as, bool(loopback))


def _rows_node_ok(img, img_shapes, pts_rows, att_rows, padding_mode) -> bool:
    """GPU tensors of a dtype combination the kernels take, the binding built, no autocast / tracing / per-kernel timer:
    what the C++ row node needs (everything else takes the Python node, whose launchers carry the error messages)."""
    from .functional import _DTYPE_TRIPLES, _SHAPE_DTYPES, _autocast_on
    if _rows_ext(img) is None or torch.compiler.is_compiling() or _autocast_on():
        return False
    dev = img.device
    return (img.dim() == 4 and pts_rows.dim() == 5 and att_rows.dim() == 4 and pts_rows.shape[-1] == 2
            and pts_rows.shape[1] == img.shape[2] and att_rows.shape == pts_rows.shape[:4]
            and (img.dtype, pts_rows.dtype, att_rows.dtype) in _DTYPE_TRIPLES and padding_mode in _lib.PADDING_MODES
            and img_shapes.device == dev and pts_rows.device == dev and att_rows.device == dev
            and img_shapes.dtype in _SHAPE_DTYPES and tuple(img_shapes.shape) == (pts_rows.shape[2], 2))


def owners_sum_bytes(B: int, Q: int, world: int, rank: int, plane_bytes: int) -> int:
    """Bytes of grad_value that ``rank`` of a ``world``-rank row-sharded job takes into a sum with other ranks
    (``grad_value_sync="owners"``): one ``[I, H, D]`` gradient per batch element whose rows it shares with a peer."""
    r0, r1 = row_shard_bounds(B * Q, world, rank)
    shared = 0
    for b, q0, q1 in row_segments(Q, r0, r1):
        if q1 - q0 < Q:
            shared += 1
    return shared * int(plane_bytes)
