"""The module's projections with the row-split weight gradient (msda_triton_amd/_linear.py) against ``nn.Linear``."""
import pytest
import torch

from msda_triton_amd import MultiscaleDeformableAttention, _linear


@pytest.mark.parametrize("n", [4096, 5000, 2048 * 3 + 17, 300])
def test_row_split_weight_grad_and_column_sum_match_the_plain_products(n):
    g = torch.Generator().manual_seed(n)
    gy = torch.randn(n, 24, generator=g, dtype=torch.float64)
    x = torch.randn(n, 40, generator=g, dtype=torch.float64)
    torch.testing.assert_close(_linear.row_split_weight_grad(gy, x).double(), gy.t() @ x, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(_linear.column_sum(gy).double(), gy.sum(0), rtol=1e-5, atol=1e-4)


def test_row_block_divides_or_is_zero():
    for n in (21760, 40000, 53176, 8192, 10007):
        c = _linear._row_block(n)
        assert c == 0 or (n % c == 0 and 512 <= c <= 2048)
    assert _linear._row_block(10007) == 0  # a prime: blocks of 2048 plus a tail product


def test_function_matches_linear_on_host_tensors():
    torch.manual_seed(0)
    layer = torch.nn.Linear(12, 20).double()
    x = torch.randn(3, 700, 12, dtype=torch.float64, requires_grad=True)
    y = _linear._RowSplitLinear.apply(x, layer.weight, layer.bias)
    go = torch.randn_like(y)
    got = torch.autograd.grad(y, (x, layer.weight, layer.bias), go)
    y2 = layer(x)
    want = torch.autograd.grad(y2, (x, layer.weight, layer.bias), go)
    torch.testing.assert_close(y, y2)
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=1e-9, atol=1e-9)


def test_projection_leaves_host_and_small_calls_to_the_layer():
    layer = torch.nn.Linear(8, 8)
    x = torch.randn(2, 16, 8)
    assert projection_is_plain(layer, x)


def projection_is_plain(layer, x):
    y = _linear.projection(layer, x.requires_grad_())
    return "RowSplit" not in type(y.grad_fn).__name__


@pytest.mark.gpu
@pytest.mark.parametrize("autocast", [False, True])
def test_module_step_with_row_split_projections_matches_plain_layers(autocast, monkeypatch):
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    levels = [(48, 48), (24, 24), (12, 12), (6, 6)]
    I = sum(h * w for h, w in levels)  # noqa: E741
    B, Q, E = 4, 2500, 64
    m = MultiscaleDeformableAttention(E, E, 4, 4, 4, "border", True).to(dev)
    img = torch.randn(B, I, E, device=dev, requires_grad=True)
    q = torch.randn(B, Q, E, device=dev, requires_grad=True)
    ref = torch.rand(B, Q, 2, device=dev)
    shapes = torch.tensor(levels, device=dev)
    go = torch.randn(B, Q, E, device=dev)
    assert B * Q >= _linear.ROW_SPLIT_MIN_ROWS and B * I >= _linear.ROW_SPLIT_MIN_ROWS

    def run():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            out = m(img, shapes, q, ref)
        grads = torch.autograd.grad(out, [img, q] + list(m.parameters()), go.to(out.dtype))
        return [out] + list(grads)

    got = run()
    assert "RowSplit" in type(_linear.projection(m.img_input_proj, img).grad_fn).__name__
    monkeypatch.setattr(_linear, "ROW_SPLIT_MIN_ROWS", 1 << 60)
    want = run()
    tol = dict(rtol=3e-2, atol=3e-2) if autocast else dict(rtol=1e-4, atol=1e-4)
    for a, b in zip(got, want):
        assert a.dtype == b.dtype and a.shape == b.shape
        scale = b.float().abs().max().clamp_min(1e-6)
        torch.testing.assert_close(a.float() / scale, b.float() / scale, **tol)


# ------------------------------------------------------------------------------------------
# the row-split route calls F.linear on layer.weight / layer.bias itself, so it may only replace a
# PLAIN nn.Linear without hooks — the reference module always calls the layer (frontend.py:253-267)
# ------------------------------------------------------------------------------------------
class _LoraLikeLinear(torch.nn.Linear):
    """``.weight`` is the base weight; the adapter path lives in ``forward`` (what peft's lora.Linear does)."""

    def __init__(self, i, o):
        super().__init__(i, o)
        self.a = torch.nn.Parameter(torch.randn(i, 2) * 0.1)
        self.b = torch.nn.Parameter(torch.randn(2, o) * 0.1)

    def forward(self, x):
        return super().forward(x) + (x @ self.a) @ self.b


def test_plain_linear_gate():
    layer = torch.nn.Linear(8, 8)
    assert _linear._plain_linear(layer)
    assert not _linear._plain_linear(_LoraLikeLinear(8, 8))
    for register in ("register_forward_hook", "register_forward_pre_hook", "register_full_backward_hook",
                     "register_full_backward_pre_hook"):
        layer = torch.nn.Linear(8, 8)
        handle = getattr(layer, register)(lambda *a: None)
        assert not _linear._plain_linear(layer), register
        handle.remove()
        assert _linear._plain_linear(layer), register
    layer = torch.nn.Linear(8, 8)
    handle = torch.nn.modules.module.register_module_forward_hook(lambda *a: None)
    try:
        assert not _linear._plain_linear(layer)
    finally:
        handle.remove()
    assert _linear._plain_linear(layer)


@pytest.mark.gpu
def test_hooks_and_wrapped_layers_run_at_row_split_sizes():
    """>= 8 192 rows on the GPU with grad enabled — where a plain layer takes the row-split route: a forward hook still
    fires, and a LoRA-style subclass keeps its adapter path (values and gradients equal the layer called directly)."""
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    x = torch.randn(4, 4096, 32, device=dev, requires_grad=True)
    assert x.numel() // 32 >= _linear.ROW_SPLIT_MIN_ROWS
    plain = torch.nn.Linear(32, 48).to(dev)
    assert "RowSplit" in type(_linear.projection(plain, x).grad_fn).__name__
    seen = []
    handle = plain.register_forward_hook(lambda mod, args, out: seen.append(out.shape) or out * 2)
    y = _linear.projection(plain, x)
    handle.remove()
    assert seen == [torch.Size([4, 4096, 48])] and "RowSplit" not in type(y.grad_fn).__name__
    torch.testing.assert_close(y, 2 * plain(x))
    lora = _LoraLikeLinear(32, 48).to(dev)
    y = _linear.projection(lora, x)
    want = lora(x)
    torch.testing.assert_close(y, want)
    g = torch.autograd.grad(y.sum(), [lora.a, lora.b])
    gw = torch.autograd.grad(want.sum(), [lora.a, lora.b])
    for p, q in zip(g, gw):
        torch.testing.assert_close(p, q)
    assert float(g[0].abs().sum()) > 0  # the adapter path took part
