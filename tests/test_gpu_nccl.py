"""The multi-GPU code paths over RCCL (``backend="nccl"``) — IN-PROCESS at world size 1 on the one GPU of the test box.

The reference has no multi-device code (its kernels only state the independence of the (b, q, h) units,
/root/reference/src/msda_triton/kernels.py:18-21); SURVEY 8e's sharding lives in ``msda_triton_amd/distributed.py`` and until
round 6 had only ever run over gloo on host tensors.  Here the process group is created inside the pytest process
(``init_process_group("nccl", rank=0, world_size=1, device_id=cuda:0)`` — no subprocess, no re-exec) and every collective
the N-rank code issues runs on device tensors through RCCL: the in-place ``all_gather_into_tensor``, the grouped
point-to-point pieces (``batch_isend_irecv``, sent to and received from the rank itself: ``loopback=True``), the
``reduce_scatter_tensor`` of the incoming gradient, the grad_value sums (whole tensor and per batch element over a
sub-group), and the query-partitioned operator's gather / all-reduce.  Each result is held to the unsharded operator,
BIT-EXACT: the kernels are the same and every (b, q, h) unit is computed independently of how the rows are cut.
"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_world1():
    import torch.distributed as dist
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    created = False
    if not dist.is_initialized():
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(_free_port())
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    yield dist
    if created:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def _inputs(B, Q, H=4, D=32, levels=((12, 10), (6, 5), (3, 3)), P=3, dtype="float32", seed=3):
    from msda_triton_amd import synth
    wl = synth.Workload("t", B, Q, H, D, levels, P, dtype, "zeros", False)
    d = synth.make_inputs_torch(wl, "cuda", seed=seed, loc_lo=-0.1, loc_hi=1.1)
    return wl, d


def _reference(d, pm="zeros", ac=False):
    from msda_triton_amd import multiscale_deformable_attention
    v, l, a = (d[k].detach().clone().requires_grad_(True) for k in ("value", "loc", "attn"))
    out = multiscale_deformable_attention(v, d["shapes"], l, a, pm, ac)
    out.backward(d["grad_out"])
    return out.detach(), v.grad, l.grad, a.grad


def _same(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b)


@pytest.mark.parametrize("chunks", [1, 4])
@pytest.mark.parametrize("B,Q", [(2, 300), (3, 37), (1, 5)])
@pytest.mark.parametrize("sharded_inputs", [False, True])
def test_row_sharded_over_rccl_loopback_is_the_unsharded_operator(nccl_world1, chunks, B, Q, sharded_inputs):
    """chunks 1: one in-place all_gather_into_tensor; chunks 4: four rounds of batch_isend_irecv to / from the rank itself
    (3 x 37 = 111 rows in 4 pieces: pieces that straddle batch elements and a short last one; 1 x 5: pieces of 2, 2, 1, 0)."""
    from msda_triton_amd.distributed import row_sharded_multiscale_deformable_attention
    wl, d = _inputs(B, Q)
    ref_out, ref_gv, ref_gl, ref_ga = _reference(d)
    v = d["value"].clone().requires_grad_(True)
    l = d["loc"].clone()
    a = d["attn"].clone()
    if sharded_inputs:
        l, a = l.reshape(B * Q, *l.shape[2:]), a.reshape(B * Q, *a.shape[2:])
    l.requires_grad_(True)
    a.requires_grad_(True)
    out = row_sharded_multiscale_deformable_attention(v, d["shapes"], l, a, "zeros", False,
                                                      inputs_are_sharded=sharded_inputs, num_queries=Q if sharded_inputs else None,
                                                      overlap_chunks=chunks, loopback=True)
    out.backward(d["grad_out"])
    torch.cuda.synchronize()
    assert _same(out.detach(), ref_out)
    assert _same(v.grad, ref_gv)
    assert _same(l.grad.reshape(ref_gl.shape), ref_gl)
    assert _same(a.grad.reshape(ref_ga.shape), ref_ga)


@pytest.mark.parametrize("value_sync", ["all_reduce", "owners", "none"])
@pytest.mark.parametrize("grad_sync", ["slice", "reduce_scatter"])
def test_row_sharded_backward_collectives_over_rccl(nccl_world1, value_sync, grad_sync):
    """all_reduce of the whole grad_value, per-batch-element all_reduce over a (sub-)group, reduce_scatter_tensor of the
    incoming gradient: at one rank each is the identity, so the gradients must come back bit-exact."""
    from msda_triton_amd.distributed import row_sharded_multiscale_deformable_attention
    B, Q = 2, 111
    wl, d = _inputs(B, Q, seed=5)
    ref_out, ref_gv, ref_gl, ref_ga = _reference(d)
    v, l, a = (d[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
    out = row_sharded_multiscale_deformable_attention(v, d["shapes"], l, a, "zeros", False, grad_value_sync=value_sync,
                                                      grad_sync=grad_sync, overlap_chunks=2, loopback=True)
    out.backward(d["grad_out"])
    torch.cuda.synchronize()
    assert _same(out.detach(), ref_out) and _same(v.grad, ref_gv) and _same(l.grad, ref_gl) and _same(a.grad, ref_ga)


@pytest.mark.parametrize("dtype,pm,ac", [("bfloat16", "zeros", False), ("float16", "border", True), ("float32", "border", False)])
def test_row_sharded_dtypes_and_modes_over_rccl(nccl_world1, dtype, pm, ac):
    from msda_triton_amd import multiscale_deformable_attention
    from msda_triton_amd.distributed import row_sharded_multiscale_deformable_attention
    wl, d = _inputs(2, 130, dtype=dtype, seed=7)
    v2, l2, a2 = (d[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
    ref = multiscale_deformable_attention(v2, d["shapes"], l2, a2, pm, ac)
    ref.backward(d["grad_out"])
    for chunks in (1, 3):
        v, l, a = (d[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
        out = row_sharded_multiscale_deformable_attention(v, d["shapes"], l, a, pm, ac, overlap_chunks=chunks, loopback=True)
        out.backward(d["grad_out"])
        torch.cuda.synchronize()
        assert _same(out.detach(), ref.detach()) and _same(v.grad, v2.grad) and _same(l.grad, l2.grad) and _same(a.grad, a2.grad)


def test_empty_local_range_still_takes_part(nccl_world1):
    """Q = 0: the rank owns no rows; every collective is still entered and the (empty) result has the right shape."""
    from msda_triton_amd.distributed import row_sharded_multiscale_deformable_attention
    wl, d = _inputs(2, 0)
    v, l, a = (d[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
    for chunks in (1, 2):
        out = row_sharded_multiscale_deformable_attention(v, d["shapes"], l, a, "zeros", False, overlap_chunks=chunks,
                                                          loopback=True)
        assert tuple(out.shape) == (2, 0, wl.H, wl.D)
        out.sum().backward()
        torch.cuda.synchronize()
        assert float(v.grad.abs().sum()) == 0.0
        v.grad = None


@pytest.mark.parametrize("grad_sync", ["slice", "reduce_scatter"])
@pytest.mark.parametrize("sharded_inputs", [False, True])
@pytest.mark.parametrize("B", [2, 9])
def test_query_sharded_over_rccl_is_the_unsharded_operator(nccl_world1, grad_sync, sharded_inputs, B):
    """The query partition's _GatherQueryShards (per-batch-element all_gather_into_tensor for B <= 8, one gather plus a
    permute above) and _ReplicatedValue (all_reduce of grad_value) on RCCL."""
    from msda_triton_amd.distributed import sharded_multiscale_deformable_attention
    Q = 41
    wl, d = _inputs(B, Q, seed=9)
    ref_out, ref_gv, ref_gl, ref_ga = _reference(d)
    v, l, a = (d[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
    out = sharded_multiscale_deformable_attention(v, d["shapes"], l, a, "zeros", False, inputs_are_sharded=sharded_inputs,
                                                  num_queries=Q if sharded_inputs else None, grad_sync=grad_sync)
    out.backward(d["grad_out"])
    torch.cuda.synchronize()
    assert _same(out.detach(), ref_out) and _same(v.grad, ref_gv) and _same(l.grad, ref_gl) and _same(a.grad, ref_ga)


def test_one_rank_without_loopback_takes_the_cpp_row_node(nccl_world1):
    """At one rank there is nothing to exchange: the operator is the C++ autograd node (no Python in the step) and
    equals the direct operator; so does one rank of a larger job played on this GPU (compute_only_as)."""
    from msda_triton_amd import _ext
    from msda_triton_amd.distributed import row_shard_bounds, row_sharded_multiscale_deformable_attention
    assert _ext.load() is not None, "the C++ binding was not built"
    B, Q = 3, 50
    wl, d = _inputs(B, Q, seed=11)
    ref_out, ref_gv, ref_gl, ref_ga = _reference(d)
    v, l, a = (d[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
    out = row_sharded_multiscale_deformable_attention(v, d["shapes"], l, a, "zeros", False, overlap_chunks=3)
    assert "RowSharded" not in type(out.grad_fn).__name__, type(out.grad_fn).__name__  # (not the Python node)
    out.backward(d["grad_out"])
    torch.cuda.synchronize()
    assert _same(out.detach(), ref_out) and _same(v.grad, ref_gv) and _same(l.grad, ref_gl) and _same(a.grad, ref_ga)
    # rank 1 of 4: rows [38, 76) — a partial batch element, then another one
    world, rank = 4, 1
    r0, r1 = row_shard_bounds(B * Q, world, rank)
    v, l, a = (d[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
    out = row_sharded_multiscale_deformable_attention(v, d["shapes"], l, a, "zeros", False, compute_only_as=(world, rank),
                                                      overlap_chunks=2)
    assert "RowSharded" not in type(out.grad_fn).__name__
    g = torch.zeros_like(d["grad_out"]).reshape(B * Q, wl.H, wl.D)
    g[r0:r1] = d["grad_out"].reshape(B * Q, wl.H, wl.D)[r0:r1]
    out.backward(g.reshape(B, Q, wl.H, wl.D))
    torch.cuda.synchronize()
    rows = lambda t: t.reshape(B * Q, *t.shape[2:])[r0:r1]  # noqa: E731
    assert _same(rows(out.detach()), rows(ref_out))
    assert _same(rows(l.grad), rows(ref_gl)) and _same(rows(a.grad), rows(ref_ga))
    assert float(rows(l.grad).abs().sum()) > 0 and float(l.grad.abs().sum()) == float(rows(l.grad).abs().sum())
    # grad_value of the shard: the unsharded operator fed this rank's rows of grad_out only
    v3, l3, a3 = (d[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
    from msda_triton_amd import multiscale_deformable_attention
    multiscale_deformable_attention(v3, d["shapes"], l3, a3, "zeros", False).backward(g.reshape(B, Q, wl.H, wl.D))
    torch.cuda.synchronize()
    assert torch.allclose(v.grad, v3.grad, rtol=1e-5, atol=1e-6)  # (other summation order: the shard's pieces are their own launches)


def test_bench_force_dist_runs_the_nccl_group(nccl_world1):
    """What `bench.py --gpus 1 --force-dist` does at every step — the sharded operator on the default nccl group plus the
    barrier / all_reduce(MAX) bracket of the timed region — inside this process."""
    dist = nccl_world1
    from msda_triton_amd.distributed import row_sharded_multiscale_deformable_attention
    wl, d = _inputs(4, 500, H=8, seed=13)
    v = d["value"].clone().requires_grad_(True)
    l = d["loc"].reshape(4 * 500, *d["loc"].shape[2:]).clone().requires_grad_(True)
    a = d["attn"].reshape(4 * 500, *d["attn"].shape[2:]).clone().requires_grad_(True)
    for _ in range(3):
        out = row_sharded_multiscale_deformable_attention(v, d["shapes"], l, a, "zeros", False, inputs_are_sharded=True,
                                                          num_queries=500, grad_value_sync="owners")
        out.backward(torch.rand_like(out))
        v.grad = l.grad = a.grad = None
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.5
