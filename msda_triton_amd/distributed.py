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
(c4 0.9 MB, c5 51 MB per rank); the collectives are issued on whole tensors or on up to four pieces.

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

from .functional import multiscale_deformable_attention


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


def default_overlap_chunks(rows: int, world: int) -> int:
    """Pieces a rank's rows are computed and exchanged in (the same number on every rank: it follows the nominal shard
    size).  Piece k's exchange runs while piece k + 1 computes, so only the LAST piece's exchange is exposed: more
    pieces hide more of it — as long as a piece stays worth a kernel launch: measured on one GPU (bench.py
    ``shard_compute_bound``, round 5) c2's 40 000 rows per rank in 8 pieces of 5 000 rows take 0.15 ms of forward
    against 0.07 ms in one piece (small launches, and the gather kernels' LDS-served levels need a larger grid), so a
    piece has >= 8 192 rows: c2 weak scaling 4 pieces, c5 strong scaling over 8 ranks (50 000 rows per rank) 6."""
    if world <= 1:
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


class _RowShardedMSDA(Function):
    """The whole sharded operator as ONE autograd node.

    forward: this rank's rows are computed in ``chunks`` pieces, each written by the kernel STRAIGHT into its place
    in the full ``[B*Q, H, D]`` result (no local tensor, no padding, no concatenation); as soon as a piece is
    enqueued its exchange starts — every rank sends the piece to every peer and receives the peers' pieces into
    their final rows (grouped point-to-point: on the xGMI full mesh each pair uses its own link) — while the next
    piece computes.  Equal shards and ``chunks == 1`` take one ``all_gather_into_tensor`` instead.
    backward: this rank's rows of the incoming gradient (replicated consumers) or a reduce-scatter; the local
    backward kernels; grad_value summed as ``grad_value_sync`` says.  Every rank runs every collective, whatever its
    shard holds (an empty shard computes nothing and still takes part).
    """

    @staticmethod
    def forward(ctx, img, img_shapes, pts_rows, att_rows, padding_mode, align_corners, num_queries, group,
                grad_value_sync, grad_sync, owner_groups, chunks, compute_only_as=None):
        # compute_only_as = (world, rank): this process plays ONE rank of a larger job with every exchange left out
        # (row_sharded_multiscale_deformable_attention: the compute half of the scaling model on one GPU)
        emulated = compute_only_as is not None
        world, rank = compute_only_as if emulated else (dist.get_world_size(group), dist.get_rank(group))
        B, _, H, D = img.shape
        Q = int(num_queries)
        rows = B * Q
        bounds = [row_shard_bounds(rows, world, r) for r in range(world)]
        r0, r1 = bounds[rank]
        gpu = img.device.type == "cuda"
        full = pts_rows.new_empty((rows, H, D))  # the sampling inputs' dtype (fp32 next to a 16-bit pyramid)
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

            _run_pieces(piece, Q, r0 + c0, r0 + c1)
            if world == 1 or emulated:
                continue
            if chunks == 1 and equal:
                per = r1 - r0
                _all_gather_into(full.view(world, per, H, D), full[r0:r1], group)  # in place: my rows are already there
                continue
            ops = []
            for peer in range(world):
                if peer == rank:
                    continue
                gpeer = dist.get_global_rank(group, peer) if group is not None else peer
                if c1 > c0:
                    ops.append(dist.P2POp(dist.isend, full[r0 + c0:r0 + c1], gpeer, group))
                p0, p1 = bounds[peer]
                pc0, pc1 = _chunk_bounds(p1 - p0, chunks, k)
                if pc1 > pc0:
                    ops.append(dist.P2POp(dist.irecv, full[p0 + pc0:p0 + pc1], gpeer, group))
            if ops:
                pending.extend(dist.batch_isend_irecv(ops))
        for req in pending:
            req.wait()
        ctx.save_for_backward(img, img_shapes, pts_rows, att_rows)
        ctx.meta = (padding_mode, align_corners, Q, group, grad_value_sync, grad_sync, owner_groups, world, rank, r0, r1)
        ctx.emulated = emulated
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
        if grad_sync == "reduce_scatter" and world > 1:
            per = -(-rows // world)
            padded = g_rows if rows == world * per else torch.nn.functional.pad(g_rows, (0, 0, 0, 0, 0, world * per - rows))
            mine = torch.empty_like(padded[:per])
            dist.reduce_scatter_tensor(mine, padded.contiguous(), group=group)
            mine = mine[:r1 - r0]
        else:
            mine = g_rows[r0:r1]
        gpu = img.device.type == "cuda"
        # Gradient tensors of the shard; the kernels write every piece straight into its slice.  A contiguous row
        # range meets a batch element in at most one piece, so grad_value needs no accumulation: the batch elements
        # this rank does not touch are zeroed, the others are written whole.
        g_img = torch.empty_like(img) if need_img else None
        g_pts = torch.empty_like(pts_rows) if need_pts else None
        g_att = torch.empty_like(att_rows) if need_att else None
        touched = torch.zeros(B, dtype=torch.bool)

        def piece(b, nb, q0, q1, before):
            n = (q1 - q0) if nb == 1 else nb * Q
            pts = pts_rows[before:before + n].reshape(nb, n // nb, *pts_rows.shape[1:])
            att = att_rows[before:before + n].reshape(nb, n // nb, *att_rows.shape[1:])
            go = mine[before:before + n].reshape(nb, n // nb, H, D)
            touched[b:b + nb] = True
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
                if not bool(touched[b]):
                    g_img[b].zero_()
        if need_img and world > 1:
            if grad_value_sync == "all_reduce":
                dist.all_reduce(g_img, op=dist.ReduceOp.SUM, group=group)
            elif grad_value_sync == "owners":
                for b, (ranks, pg) in enumerate(owner_groups):  # ascending b on every rank: no cyclic waits
                    if pg is not None and rank in ranks:
                        dist.all_reduce(g_img[b], op=dist.ReduceOp.SUM, group=pg)
        return (g_img, None, g_pts, g_att) + (None,) * 9


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
) -> torch.Tensor:
    """Row-sharded operator; returns the full ``[B, Q, H, D]`` output on every rank.

    ``compute_only_as=(world, rank)`` (measurement aid, no process group needed): run exactly what rank ``rank`` of a
    ``world``-rank job computes — its rows, in the same pieces, through the same kernels — and leave every exchange
    out.  Only that rank's rows of the result (and its share of the gradients) are meaningful.  ``bench.py``'s
    ``shard_compute_bound`` leg times it on one GPU: the speed-up ceiling before any byte crosses xGMI.

    ``overlap_chunks``: pieces the local rows are computed and exchanged in (piece k's exchange overlaps piece k+1's
    kernels); None picks 1 .. 4 by shard size.

    ``img`` is ``[B, I, H, D]`` on every rank (a rank only reads the batch elements its rows fall into).
    ``inputs_are_sharded=False``: ``sampling_points [B,Q,H,L,P,2]`` / ``attention_weights [B,Q,H,L,P]`` are
    replicated and each rank computes its :func:`row_shard_bounds` rows.
    ``inputs_are_sharded=True``: they hold only this rank's rows, flattened: ``[rows, H, L, P, 2]`` /
    ``[rows, H, L, P]`` (``num_queries`` = Q per batch element is then required).
    """
    if compute_only_as is not None:
        world, rank = (int(v) for v in compute_only_as)
        if not 0 <= rank < world:
            raise ValueError(f"compute_only_as={compute_only_as!r}: need 0 <= rank < world")
        compute_only_as = (world, rank)
        grad_value_sync = "none"
    elif not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("torch.distributed is not initialised; call init_process_group first")
    if grad_value_sync not in ("all_reduce", "owners", "none"):
        raise ValueError(f"unknown grad_value_sync {grad_value_sync!r}")
    if compute_only_as is None:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
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
    owners = _owner_groups(B, Q, group) if (grad_value_sync == "owners" and img.requires_grad) else None
    if grad_value_sync == "owners" and img.requires_grad and owners is None:
        grad_value_sync = "all_reduce"  # a caller-supplied sub-group: see _owner_groups
        global _WARNED_OWNERS_DOWNGRADE
        if not _WARNED_OWNERS_DOWNGRADE:
            _WARNED_OWNERS_DOWNGRADE = True
            warnings.warn("grad_value_sync='owners' needs sub-groups of the DEFAULT process group; on a caller-supplied "
                          "group grad_value is all-reduced over that group instead", stacklevel=2)
    if overlap_chunks is None:
        overlap_chunks = default_overlap_chunks(rows, world)
    return _RowShardedMSDA.apply(img, img_shapes, pts_rows, att_rows, padding_mode, bool(align_corners), Q, group,
                                 grad_value_sync, grad_sync, owners, overlap_chunks, compute_only_as)


def owners_sum_bytes(B: int, Q: int, world: int, rank: int, plane_bytes: int) -> int:
    """Bytes of grad_value that ``rank`` of a ``world``-rank row-sharded job takes into a sum with other ranks
    (``grad_value_sync="owners"``): one ``[I, H, D]`` gradient per batch element whose rows it shares with a peer."""
    r0, r1 = row_shard_bounds(B * Q, world, rank)
    shared = 0
    for b, q0, q1 in row_segments(Q, r0, r1):
        if q1 - q0 < Q:
            shared += 1
    return shared * int(plane_bytes)
