"""Query-sharded multi-GPU execution: one process per GPU, ``torch.distributed`` (RCCL over xGMI).

The reference has no multi-device code; every (b, q, h) of the operator is independent
(/root/reference/src/msda_triton/kernels.py:18-21), so the queries shard trivially:

  forward   each rank runs the HIP kernels on its contiguous slice of the query axis against the
            (replicated) value pyramid, then ONE all-gather assembles ``[B, Q, H, D]`` on every rank;
  backward  grad_sampling_points / grad_attention_weights are shard-local (no communication);
            grad_value is a sum over all queries, so it is all-reduced across the ranks.

With 8 GPUs on one node the all-gather moves ``B*Q*H*D*s/8`` bytes per peer over point-to-point
xGMI links; the collectives are issued once per call on whole tensors (no bucketing needed at
these sizes: c4 0.9 MB, c5 51 MB per rank).

Two partitions are offered:

  rows     (:func:`row_sharded_multiscale_deformable_attention`, what ``bench.py --gpus N`` runs) — contiguous
           ranges of the flattened ``[B*Q]`` row space.  The per-rank output slices are contiguous in the final
           ``[B, Q, H, D]`` tensor, so ONE ``all_gather_into_tensor`` lands them in place; a rank touches only the
           batch elements its rows fall into, so grad_value needs no communication at all when the ranks divide
           B, and otherwise only a sum among the ranks that share a batch element.
  queries  (:func:`sharded_multiscale_deformable_attention`) — every rank takes the same query range of every
           batch element (sequence-parallel style callers that already hold ``[B, Q/N, ...]`` slices).
"""
from __future__ import annotations

from typing import Literal, Optional, Tuple

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


def _owner_groups(B: int, Q: int, group):
    """For every batch element touched by more than one rank: (ranks, process group).  Every rank of
    ``group`` must call this with the same (B, Q) (new_group is collective); results are cached."""
    world = dist.get_world_size(group)
    key = (id(group), B, Q, world)
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


class _GatherRows(Function):
    """local rows [per, H, D] on every rank -> [B, Q, H, D] on every rank with one all-gather (the slices are
    contiguous in the result).  Backward: this rank's rows of the incoming gradient (replicated consumers) or a
    reduce-scatter of it (``grad_sync="reduce_scatter"``)."""

    @staticmethod
    def forward(ctx, local: torch.Tensor, B: int, Q: int, group, grad_sync: str):
        world = dist.get_world_size(group)
        per, H, D = local.shape
        ctx.group, ctx.world, ctx.rank, ctx.per, ctx.grad_sync = group, world, dist.get_rank(group), per, grad_sync
        buf = local.new_empty((world * per, H, D))
        _all_gather_into(buf.view(world, per, H, D), local.contiguous(), group)
        return buf[:B * Q].view(B, Q, H, D)

    @staticmethod
    def backward(ctx, grad_full: torch.Tensor):
        B, Q, H, D = grad_full.shape
        per, world = ctx.per, ctx.world
        rows = grad_full.reshape(B * Q, H, D)
        if B * Q < world * per:
            rows = torch.nn.functional.pad(rows, (0, 0, 0, 0, 0, world * per - B * Q))
        if ctx.grad_sync == "reduce_scatter":
            mine = torch.empty_like(rows[:per])
            dist.reduce_scatter_tensor(mine, rows.contiguous(), group=ctx.group)
        else:
            mine = rows[ctx.rank * per:(ctx.rank + 1) * per]
        return mine, None, None, None, None


class _ValueGradSync(Function):
    """Identity in forward.  Backward: ``"all_reduce"`` — every rank gets the complete grad_value (value computed
    redundantly on every rank); ``"owners"`` — per batch element, summed among the ranks whose rows fall into
    it (other batch elements keep this rank's zeros; no communication when the ranks divide B); ``"none"``."""

    @staticmethod
    def forward(ctx, value: torch.Tensor, mode: str, owner_groups, group):
        ctx.mode, ctx.owner_groups, ctx.group = mode, owner_groups, group
        return value.view_as(value)

    @staticmethod
    def backward(ctx, grad: torch.Tensor):
        if ctx.mode == "all_reduce":
            grad = grad.contiguous()
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=ctx.group)
        elif ctx.mode == "owners":
            grad = grad.contiguous()
            me = dist.get_rank(ctx.group)
            for b, (ranks, pg) in enumerate(ctx.owner_groups):  # ascending b on every rank: no cyclic waits
                if pg is not None and me in ranks:
                    dist.all_reduce(grad[b], op=dist.ReduceOp.SUM, group=pg)
        return grad, None, None, None


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
) -> torch.Tensor:
    """Row-sharded operator; returns the full ``[B, Q, H, D]`` output on every rank.

    ``img`` is ``[B, I, H, D]`` on every rank (a rank only reads the batch elements its rows fall into).
    ``inputs_are_sharded=False``: ``sampling_points [B,Q,H,L,P,2]`` / ``attention_weights [B,Q,H,L,P]`` are
    replicated and each rank computes its :func:`row_shard_bounds` rows.
    ``inputs_are_sharded=True``: they hold only this rank's rows, flattened: ``[rows, H, L, P, 2]`` /
    ``[rows, H, L, P]`` (``num_queries`` = Q per batch element is then required).
    """
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("torch.distributed is not initialised; call init_process_group first")
    if grad_value_sync not in ("all_reduce", "owners", "none"):
        raise ValueError(f"unknown grad_value_sync {grad_value_sync!r}")
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
    per = -(-rows // world)
    if inputs_are_sharded:
        if sampling_points.shape[0] != r1 - r0:
            raise ValueError(f"rank {rank} owns rows [{r0}, {r1}) but got {sampling_points.shape[0]} rows")
        pts_rows, att_rows = sampling_points, attention_weights
    else:
        pts_rows = sampling_points.reshape(rows, *sampling_points.shape[2:])[r0:r1]
        att_rows = attention_weights.reshape(rows, *attention_weights.shape[2:])[r0:r1]
    if img.requires_grad and grad_value_sync != "none":
        owners = _owner_groups(B, Q, group) if grad_value_sync == "owners" else None
        img = _ValueGradSync.apply(img, grad_value_sync, owners, group)
    H, D = img.shape[2], img.shape[3]
    segs = row_segments(Q, r0, r1)
    pieces = []
    if segs and all(q0 == 0 and q1 == Q for _, q0, q1 in segs):  # whole batch elements: one launch
        b0, nb = segs[0][0], len(segs)
        out = multiscale_deformable_attention(img[b0:b0 + nb], img_shapes,
                                              pts_rows.reshape(nb, Q, *pts_rows.shape[1:]),
                                              att_rows.reshape(nb, Q, *att_rows.shape[1:]), padding_mode, align_corners)
        pieces.append(out.reshape(nb * Q, H, D))
    else:
        at = 0
        for b, q0, q1 in segs:
            n = q1 - q0
            out = multiscale_deformable_attention(img[b:b + 1], img_shapes, pts_rows[at:at + n].unsqueeze(0),
                                                  att_rows[at:at + n].unsqueeze(0), padding_mode, align_corners)
            pieces.append(out.reshape(n, H, D))
            at += n
    if not segs:
        # empty shard (more ranks than rows): still run the operator on zero rows, so that this rank's graph reaches
        # `img` and its _ValueGradSync backward joins the grad_value collective the other ranks are waiting in
        out = multiscale_deformable_attention(img[:1], img_shapes, pts_rows[:0].unsqueeze(0), att_rows[:0].unsqueeze(0),
                                              padding_mode, align_corners)
        pieces.append(out.reshape(0, H, D))
    if r1 - r0 < per:  # short / empty trailing shard: pad so the all-gather is regular
        pieces.append(img.new_zeros((per - (r1 - r0), H, D)))
    local = pieces[0] if len(pieces) == 1 else torch.cat(pieces, 0)
    return _GatherRows.apply(local, B, Q, group, grad_sync)
