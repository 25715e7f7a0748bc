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
