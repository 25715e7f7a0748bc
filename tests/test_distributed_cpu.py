"""CPU-only, world_size 2 over gloo: the query-shard path (all-gather forward, all-reduce of
grad_value) equals the unsharded operator."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q_total, sharded_inputs, grad_sync, batch, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from msda_triton_amd import multiscale_deformable_attention, synth
        from msda_triton_amd.distributed import shard_bounds, sharded_multiscale_deformable_attention
        wl = synth.Workload("t", batch, q_total, 3, 8, ((6, 5), (3, 3)), 2, "float64", "zeros", False)
        d = synth.make_inputs_torch(wl, "cpu", seed=2, loc_lo=-0.2, loc_hi=1.2)
        v = d["value"].clone().requires_grad_(True)
        l = d["loc"].clone().requires_grad_(True)
        a = d["attn"].clone().requires_grad_(True)
        b, e = shard_bounds(q_total, world, rank)
        if sharded_inputs:
            l_in = d["loc"][:, b:e].clone().requires_grad_(True)
            a_in = d["attn"][:, b:e].clone().requires_grad_(True)
            out = sharded_multiscale_deformable_attention(v, d["shapes"], l_in, a_in, "zeros", False,
                                                          inputs_are_sharded=True, num_queries=q_total, grad_sync=grad_sync)
        else:
            l_in, a_in = l, a
            out = sharded_multiscale_deformable_attention(v, d["shapes"], l_in, a_in, "zeros", False, grad_sync=grad_sync)
        g = d["grad_out"] if grad_sync == "slice" else d["grad_out"] / world  # reduce_scatter sums the replicas
        out.backward(g)
        # unsharded reference on the same inputs
        v2, l2, a2 = (t.detach().clone().requires_grad_(True) for t in (d["value"], d["loc"], d["attn"]))
        ref = multiscale_deformable_attention(v2, d["shapes"], l2, a2, "zeros", False)
        ref.backward(d["grad_out"])
        ok = torch.allclose(out, ref, atol=1e-12)
        ok &= torch.allclose(v.grad, v2.grad, atol=1e-10)
        lg = l_in.grad if sharded_inputs else l_in.grad[:, b:e]
        ag = a_in.grad if sharded_inputs else a_in.grad[:, b:e]
        ok &= torch.allclose(lg, l2.grad[:, b:e], atol=1e-10) and torch.allclose(ag, a2.grad[:, b:e], atol=1e-10)
        if not sharded_inputs:  # gradients outside this rank's shard stay zero
            mask = torch.ones(q_total, dtype=torch.bool)
            mask[b:e] = False
            ok &= float(l_in.grad[:, mask].abs().sum()) == 0.0
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("q_total,sharded_inputs,grad_sync,batch",
                         [(10, False, "slice", 2), (9, True, "slice", 2), (8, False, "reduce_scatter", 2), (6, False, "slice", 9)])
def test_query_shard_gloo_world2(q_total, sharded_inputs, grad_sync, batch):
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), q_total, sharded_inputs, grad_sync, batch, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_shard_bounds_cover_and_partition():
    from msda_triton_amd.distributed import shard_bounds
    for q in (0, 1, 7, 8, 9, 900, 10000):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(q, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == q
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert len({e - b for b, e in spans if e - b == -(-q // world)}) <= 1


# ---------------------------------------------------------------------------------------------
# row partition (flattened (b, q) row space): what bench.py --gpus N runs
# ---------------------------------------------------------------------------------------------
def _row_worker(rank, world, port, batch, q_total, sharded_inputs, value_sync, grad_sync, chunks, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from msda_triton_amd import multiscale_deformable_attention, synth
        from msda_triton_amd.distributed import (row_segments, row_shard_bounds,
                                                 row_sharded_multiscale_deformable_attention)
        wl = synth.Workload("t", batch, q_total, 3, 8, ((6, 5), (3, 3)), 2, "float64", "zeros", False)
        d = synth.make_inputs_torch(wl, "cpu", seed=5)
        rows = batch * q_total
        r0, r1 = row_shard_bounds(rows, world, rank)
        v = d["value"].clone().requires_grad_(True)
        if sharded_inputs:
            dr = synth.make_inputs_torch(wl, "cpu", seed=5, rows=(r0, r1))
            l_in, a_in = dr["loc"].clone().requires_grad_(True), dr["attn"].clone().requires_grad_(True)
            out = row_sharded_multiscale_deformable_attention(v, d["shapes"], l_in, a_in, "zeros", False,
                                                              inputs_are_sharded=True, num_queries=q_total,
                                                              grad_value_sync=value_sync, grad_sync=grad_sync,
                                                              overlap_chunks=chunks)
        else:
            l_in, a_in = d["loc"].clone().requires_grad_(True), d["attn"].clone().requires_grad_(True)
            out = row_sharded_multiscale_deformable_attention(v, d["shapes"], l_in, a_in, "zeros", False,
                                                              grad_value_sync=value_sync, grad_sync=grad_sync,
                                                              overlap_chunks=chunks)
        g = d["grad_out"] if grad_sync == "slice" else d["grad_out"] / world
        out.backward(g)
        v2, l2, a2 = (t.detach().clone().requires_grad_(True) for t in (d["value"], d["loc"], d["attn"]))
        ref = multiscale_deformable_attention(v2, d["shapes"], l2, a2, "zeros", False)
        ref.backward(d["grad_out"])
        ok = out.shape == ref.shape and torch.allclose(out, ref, atol=1e-12)
        mine = sorted({b for b, _, _ in row_segments(q_total, r0, r1)})
        vg = v.grad if v.grad is not None else torch.zeros_like(v)
        if value_sync == "all_reduce":
            ok &= torch.allclose(vg, v2.grad, atol=1e-10)
        elif value_sync == "owners":
            for b in range(batch):
                want = v2.grad[b] if b in mine else torch.zeros_like(v2.grad[b])
                ok &= torch.allclose(vg[b], want, atol=1e-10)
        else:  # local partial sums: they add up to the full gradient
            tot = vg.clone()
            dist.all_reduce(tot)
            ok &= torch.allclose(tot, v2.grad, atol=1e-10)
        l2r = l2.grad.reshape(rows, *l2.grad.shape[2:])[r0:r1]
        a2r = a2.grad.reshape(rows, *a2.grad.shape[2:])[r0:r1]
        lg = l_in.grad if sharded_inputs else l_in.grad.reshape(rows, *l_in.grad.shape[2:])[r0:r1]
        ag = a_in.grad if sharded_inputs else a_in.grad.reshape(rows, *a_in.grad.shape[2:])[r0:r1]
        ok &= torch.allclose(lg, l2r, atol=1e-10) and torch.allclose(ag, a2r, atol=1e-10)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch,q_total,sharded_inputs,value_sync,grad_sync,chunks", [
    (2, 2, 7, True, "owners", "slice", 1),         # ranks divide B: whole batch elements, one in-place all-gather
    (2, 2, 7, True, "owners", "slice", 3),         # ... the same in three pieces (exchange overlapped with compute)
    (2, 1, 9, False, "owners", "slice", 2),        # more ranks than batch elements: both share b = 0; ragged shards
    (2, 3, 5, True, "all_reduce", "slice", 1),     # rank boundaries inside a batch element
    (2, 3, 4, False, "none", "reduce_scatter", 2),
    (3, 2, 5, True, "owners", "slice", 2),         # the middle rank belongs to two owner groups
    (3, 1, 2, False, "all_reduce", "slice", 1),    # more ranks than rows: rank 2's shard is EMPTY, it must still join
    (3, 1, 2, True, "owners", "slice", 2),         #   the exchange and the grad_value collective (hang)
    (2, 4, 6, True, "owners", "slice", 4),         # several whole batch elements per rank, cut into four pieces
    (4, 4, 6, True, "owners", "slice", 2),         # bench.py --gpus 4: one batch element per rank
    (8, 4, 6, True, "owners", "slice", 2),         # bench.py --gpus 8: two ranks share a batch element (pair-wise sums)
])
def test_row_shard_gloo(world, batch, q_total, sharded_inputs, value_sync, grad_sync, chunks):
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_row_worker, args=(world, _free_port(), batch, q_total, sharded_inputs, value_sync, grad_sync, chunks, ret),
             nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_row_segments_partition_the_row_space():
    from msda_triton_amd.distributed import row_segments, row_shard_bounds
    for B, Q, world in ((4, 10, 8), (3, 7, 2), (1, 5, 4), (8, 900, 8), (4, 10000, 8)):
        seen = []
        for r in range(world):
            r0, r1 = row_shard_bounds(B * Q, world, r)
            for b, q0, q1 in row_segments(Q, r0, r1):
                assert 0 <= b < B and 0 <= q0 < q1 <= Q
                seen.extend(range(b * Q + q0, b * Q + q1))
        assert seen == list(range(B * Q))


def _row_worker_mixed(rank, world, port, ret):
    """bf16 pyramid next to fp32 sampling inputs (the mixed-storage contract): the exchanged result is fp32, the
    pyramid's gradient stays bf16; ranks divide B, so grad_value needs no communication."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from msda_triton_amd import multiscale_deformable_attention, synth
        from msda_triton_amd.distributed import row_shard_bounds, row_sharded_multiscale_deformable_attention
        batch, q_total = 2, 6
        wl = synth.Workload("t", batch, q_total, 3, 8, ((6, 5), (3, 3)), 2, "float32", "border", True)
        d = synth.make_inputs_torch(wl, "cpu", seed=7)
        r0, r1 = row_shard_bounds(batch * q_total, world, rank)
        dr = synth.make_inputs_torch(wl, "cpu", seed=7, rows=(r0, r1))
        v = d["value"].bfloat16().requires_grad_(True)
        l_in, a_in = dr["loc"].clone().requires_grad_(True), dr["attn"].clone().requires_grad_(True)
        out = row_sharded_multiscale_deformable_attention(v, d["shapes"], l_in, a_in, "border", True,
                                                          inputs_are_sharded=True, num_queries=q_total,
                                                          grad_value_sync="owners", overlap_chunks=2)
        out.backward(d["grad_out"])
        v2 = d["value"].bfloat16().requires_grad_(True)
        l2, a2 = d["loc"].clone().requires_grad_(True), d["attn"].clone().requires_grad_(True)
        ref = multiscale_deformable_attention(v2, d["shapes"], l2, a2, "border", True)
        ref.backward(d["grad_out"])
        ok = out.dtype == torch.float32 and v.grad.dtype == torch.bfloat16 and torch.allclose(out, ref, atol=1e-6)
        ok &= torch.allclose(v.grad[rank].float(), v2.grad[rank].float(), atol=2e-2, rtol=2e-2)
        ok &= float(v.grad[1 - rank].float().abs().sum()) == 0.0
        rows = batch * q_total
        ok &= torch.allclose(l_in.grad, l2.grad.reshape(rows, *l2.grad.shape[2:])[r0:r1], atol=1e-5)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_row_shard_gloo_mixed_storage():
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_row_worker_mixed, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}


# ------------------------------------------------------------------------------------------
# bench.py's N-rank control flow, rehearsed on host tensors over gloo (no GPU): the JSON contract of the N > 1 line,
# both exchanges, the strong-scaling leg's plumbing, and that a failing optional leg costs neither the line nor the
# other legs (an exception in an optional leg must not lose the headline)
# ------------------------------------------------------------------------------------------
def _run_bench_dry(world, extra_env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world),
           "--steps", "2", "--warmup", "1", "--backend", "gloo", "--device", "cpu", "--workload", "dryrun"]
    res = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    return res, [json.loads(ln) for ln in lines]


def test_bench_dry_run_world2_prints_the_contract_line():
    res, lines = _run_bench_dry(2)
    assert res.returncode == 0, res.stderr[-2000:]
    assert len(lines) == 1, res.stdout[-2000:]  # ONE line, from rank 0
    r = lines[0]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in r, key
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert "WEAK" in r["metric"] and r["config"]["global_queries"] == 2 * 48
    assert abs(r["value"] - 2 * 2 * 48 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    ex = r["exchange_ms"]  # both exchanges were measured, the line carries the faster one
    assert ex["all_gather"]["fwd_bwd_ms"] > 0 and ex["pieces"]["fwd_bwd_ms"] > 0
    assert r["ms_per_step"] == min(ex["all_gather"]["fwd_bwd_ms"], ex["pieces"]["fwd_bwd_ms"])
    s = r["strong_scaling_c5"]
    assert s["scaling"] == "strong" and s["n_gpus"] == 2 and s["ms_per_step"] > 0
    # the unsharded leg timed on rank 0 of the same job (not only the cross-run record)
    assert s["n1_same_job_ms"] > 0 and abs(s["speedup_vs_n1_same_job"] - s["n1_same_job_ms"] / s["ms_per_step"]) < 1e-9
    assert "failed_legs" not in r


def test_bench_dry_run_failing_optional_leg_keeps_the_line():
    res, lines = _run_bench_dry(2, {"MSDA_BENCH_INJECT_FAIL": "strong_scaling_c5"})
    assert res.returncode != 0  # the failure is reported through the exit status ...
    assert len(lines) == 1, res.stdout[-2000:]  # ... after the line has been printed
    r = lines[0]
    assert r["ms_per_step"] > 0 and "injected failure" in r["strong_scaling_c5"]["error"]
    assert r["failed_legs"] == ["strong_scaling_c5"]


# ---------------------------------------------------------------------------------------------
# the piece schedule: piece k's exchange is issued BEFORE piece k + 1 is computed, all waits come last
# ---------------------------------------------------------------------------------------------
def _schedule_worker(rank, world, port, chunks, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        import msda_triton_amd.distributed as D
        from msda_triton_amd import synth
        log = []
        real_op, real_batch = D.multiscale_deformable_attention, dist.batch_isend_irecv

        def traced_op(*a, **k):
            log.append("compute")
            return real_op(*a, **k)

        class _Req:
            def __init__(self, r):
                self.r = r

            def wait(self):
                log.append("wait")
                return self.r.wait()

        def traced_batch(ops):
            log.append("exchange")
            return [_Req(r) for r in real_batch(ops)]

        D.multiscale_deformable_attention = traced_op
        dist.batch_isend_irecv = traced_batch
        try:
            wl = synth.Workload("t", 1, 40, 2, 8, ((6, 5), (3, 3)), 2, "float64", "zeros", False)  # one batch element: one piece per chunk
            d = synth.make_inputs_torch(wl, "cpu", seed=5)
            out = D.row_sharded_multiscale_deformable_attention(d["value"], d["shapes"], d["loc"], d["attn"], "zeros", False,
                                                                overlap_chunks=chunks)
        finally:
            D.multiscale_deformable_attention = real_op
            dist.batch_isend_irecv = real_batch
        ref = real_op(d["value"], d["shapes"], d["loc"], d["attn"], "zeros", False)
        ret[rank] = (log, bool(torch.allclose(out, ref, atol=1e-12)))
    finally:
        dist.destroy_process_group()


def test_piece_exchange_is_issued_before_the_next_piece_computes():
    """the schedule itself, not only the result.  With c pieces the forward must run
    compute, exchange, compute, exchange, ... and wait for the exchanges only after the last piece was issued — so that on
    RCCL piece k travels while piece k + 1's kernels run."""
    world, chunks = 2, 3
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_schedule_worker, args=(world, _free_port(), chunks, ret), nprocs=world, join=True)
    for rank in range(world):
        log, ok = ret[rank]
        assert ok
        assert log[:2 * chunks] == ["compute", "exchange"] * chunks, log
        assert log[2 * chunks:] and set(log[2 * chunks:]) == {"wait"}, log


def test_default_overlap_chunks():
    """Pieces only where the forward can hide a piece's exchange: never at one rank, never without
    the row sizes, not for the BASELINE shapes (a row of the result is computed in 2-6 ns and travels for ~25); pieces of
    at least 8 192 rows, at most 8, for sample counts large enough."""
    from msda_triton_amd.distributed import default_overlap_chunks

    def sizes(H, D, L, P, s):
        return 4 * L * P * H * D * s, H * D * s

    assert default_overlap_chunks(40000, 1, *sizes(8, 32, 4, 4, 4)) == 1
    assert default_overlap_chunks(4 * 10000 * 8, 8) == 1                              # sizes unknown: one piece
    assert default_overlap_chunks(4 * 10000 * 8, 8, *sizes(8, 32, 4, 4, 4)) == 1      # c2 weak scaling at 8 ranks
    assert default_overlap_chunks(400000, 8, *sizes(8, 64, 5, 8, 2)) == 1             # c5 strong scaling
    assert default_overlap_chunks(7200, 8, *sizes(8, 32, 4, 4, 4)) == 1               # c4
    assert default_overlap_chunks(400000, 8, *sizes(8, 32, 8, 32, 4)) == 6            # L * P = 256: 50 000 rows per rank
    assert default_overlap_chunks(400000, 2, *sizes(8, 32, 8, 32, 4)) == 8
    assert default_overlap_chunks(7200, 8, *sizes(8, 32, 8, 32, 4)) == 1              # ... but never pieces below 8 192 rows


# ------------------------------------------------------------------------------------------
# the compute half of the scaling model on ONE process: compute_only_as=(world, rank) runs what that
# rank of a world-rank job computes, no process group, no exchange
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world,rank", [(2, 0), (2, 1), (3, 1), (8, 0), (8, 7)])
def test_compute_only_as_equals_that_ranks_rows_of_the_unsharded_operator(world, rank):
    from msda_triton_amd import multiscale_deformable_attention, synth
    from msda_triton_amd.distributed import owners_sum_bytes, row_shard_bounds, row_sharded_multiscale_deformable_attention
    wl = synth.WORKLOADS["dryrun_strong"]
    d = synth.make_inputs_torch(wl, "cpu", seed=5, dtype=torch.float64)
    v = d["value"].clone().requires_grad_(True)
    ref = multiscale_deformable_attention(v, d["shapes"], d["loc"], d["attn"], wl.padding_mode, wl.align_corners)
    r0, r1 = row_shard_bounds(wl.B * wl.Q, world, rank)
    go = torch.zeros_like(ref).reshape(wl.B * wl.Q, wl.H, wl.D)
    go[r0:r1] = d["grad_out"].reshape(wl.B * wl.Q, wl.H, wl.D)[r0:r1]
    ref.backward(go.view_as(ref))
    v2 = d["value"].clone().requires_grad_(True)
    pts = d["loc"].reshape(wl.B * wl.Q, *d["loc"].shape[2:])[r0:r1].clone().requires_grad_(True)
    att = d["attn"].reshape(wl.B * wl.Q, *d["attn"].shape[2:])[r0:r1].clone().requires_grad_(True)
    out = row_sharded_multiscale_deformable_attention(v2, d["shapes"], pts, att, wl.padding_mode, wl.align_corners,
                                                      inputs_are_sharded=True, num_queries=wl.Q,
                                                      compute_only_as=(world, rank))
    rows = out.reshape(wl.B * wl.Q, wl.H, wl.D)
    torch.testing.assert_close(rows[r0:r1], ref.detach().reshape(wl.B * wl.Q, wl.H, wl.D)[r0:r1], atol=1e-12, rtol=1e-12)
    out.backward(d["grad_out"])  # (only this rank's rows of it are read)
    torch.testing.assert_close(v2.grad, v.grad, atol=1e-12, rtol=1e-12)
    assert pts.grad.shape == pts.shape and att.grad.shape == att.shape
    plane = wl.I * wl.H * wl.D * 8
    shared = owners_sum_bytes(wl.B, wl.Q, world, rank, plane)
    assert shared % plane == 0 and 0 <= shared <= wl.B * plane
    if world == 2:  # two ranks divide B = 2: nothing is shared
        assert shared == 0


def test_bench_dry_run_world1_carries_the_shard_compute_leg():
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--backend", "gloo", "--device", "cpu",
           "--workload", "dryrun"]
    res = subprocess.run(cmd, cwd=root, env=dict(os.environ, OMP_NUM_THREADS="2"), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    r = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    leg = r["shard_compute_bound"]
    for key in ("strong_dryrun_strong", "weak_dryrun"):
        assert set(leg[key]) == {"1", "2", "4", "8"}
        for n in ("2", "4", "8"):
            e = leg[key][n]
            assert e["fwd_ms"] > 0 and e["fwd_bwd_ms"] > 0 and e["speedup_ceiling"] > 0 and e["owners_sum_bytes"] >= 0
        assert leg[key]["2"]["ideal"] == (2 if key.startswith("strong") else 1)
    assert leg["strong_dryrun_strong"]["8"]["rows"] * 8 == leg["strong_dryrun_strong"]["1"]["rows"]
    assert leg["weak_dryrun"]["8"]["rows"] == leg["weak_dryrun"]["1"]["rows"]


def test_argument_errors_of_the_row_sharded_operator():
    """Rejected before anything is computed or exchanged: unknown sync modes, loopback outside a one-rank group (it is a
    test aid for ONE rank), a compute_only_as rank outside its world, missing num_queries for sharded inputs."""
    from msda_triton_amd import synth
    from msda_triton_amd.distributed import row_sharded_multiscale_deformable_attention as op
    wl = synth.WORKLOADS["dryrun"]
    d = synth.make_inputs_torch(wl, "cpu", seed=1)
    args = (d["value"], d["shapes"], d["loc"], d["attn"], wl.padding_mode, wl.align_corners)
    with pytest.raises(RuntimeError, match="not initialised"):
        op(*args)
    with pytest.raises(ValueError, match="grad_value_sync"):
        op(*args, compute_only_as=(2, 0), grad_value_sync="sometimes")
    with pytest.raises(ValueError, match="grad_sync"):
        op(*args, compute_only_as=(2, 0), grad_sync="gather")
    with pytest.raises(ValueError, match="0 <= rank < world"):
        op(*args, compute_only_as=(2, 2))
    with pytest.raises(ValueError, match="loopback"):
        op(*args, compute_only_as=(1, 0), loopback=True)
    with pytest.raises(ValueError, match="num_queries"):
        op(*args, compute_only_as=(2, 0), inputs_are_sharded=True)
