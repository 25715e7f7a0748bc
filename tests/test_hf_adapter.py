"""HF Deformable-DETR / Grounding-DINO adapter: same numbers as transformers' own pure-PyTorch module."""
import pytest
import torch

transformers = pytest.importorskip("transformers")


def _hf_attention_layer(d_model=64, heads=4, levels=3, points=2):
    from transformers.models.deformable_detr.configuration_deformable_detr import DeformableDetrConfig
    from transformers.models.deformable_detr.modeling_deformable_detr import DeformableDetrMultiscaleDeformableAttention
    cfg = DeformableDetrConfig(d_model=d_model, num_feature_levels=levels, encoder_attention_heads=heads,
                               decoder_attention_heads=heads)
    torch.manual_seed(0)
    layer = DeformableDetrMultiscaleDeformableAttention(cfg, num_heads=heads, n_points=points)
    with torch.no_grad():  # the HF init zeroes the offsets' weight; make the sampling pattern non-trivial
        layer.sampling_offsets.weight.normal_(0, 0.5)
    return layer


def _layer_inputs(device, d_model=64, levels=((6, 5), (3, 3), (2, 1)), B=2, Q=7):
    torch.manual_seed(1)
    I = sum(h * w for h, w in levels)  # noqa: E741
    hidden = torch.randn(B, Q, d_model, device=device)
    enc = torch.randn(B, I, d_model, device=device)
    ref = torch.rand(B, Q, len(levels), 2, device=device)
    shapes = torch.tensor(levels, device=device)
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    return dict(hidden_states=hidden, encoder_hidden_states=enc, reference_points=ref, spatial_shapes=shapes,
                spatial_shapes_list=[tuple(x) for x in levels], level_start_index=starts)


def _run(layer, kw):
    import inspect
    params = inspect.signature(layer.forward).parameters
    out = layer(**{k: v for k, v in kw.items() if k in params})
    return out[0] if isinstance(out, tuple) else out


def test_adapter_matches_hf_module_on_cpu():
    from msda_triton_amd.hf_adapter import replace_hf_msda
    layer = _hf_attention_layer()
    kw = _layer_inputs("cpu")
    ref = _run(layer, kw)
    assert replace_hf_msda(layer) == 1
    got = _run(layer, kw)
    torch.testing.assert_close(got, ref, atol=1e-5, rtol=1e-4)


@pytest.mark.gpu
def test_adapter_matches_hf_module_on_gpu_fwd_and_bwd():
    from msda_triton_amd.hf_adapter import replace_hf_msda
    dev = "cuda:0"
    layer = _hf_attention_layer().to(dev)
    kw = _layer_inputs(dev)
    kw["encoder_hidden_states"].requires_grad_(True)
    ref = _run(layer, kw)
    ref.sum().backward()
    g_ref = kw["encoder_hidden_states"].grad.clone()
    w_ref = layer.sampling_offsets.weight.grad.clone()
    kw["encoder_hidden_states"].grad = None
    layer.zero_grad()
    assert replace_hf_msda(layer) == 1
    got = _run(layer, kw)
    got.sum().backward()
    torch.testing.assert_close(got, ref, atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(kw["encoder_hidden_states"].grad, g_ref, atol=1e-3, rtol=1e-2)
    torch.testing.assert_close(layer.sampling_offsets.weight.grad, w_ref, atol=1e-3, rtol=1e-2)


@pytest.mark.gpu
def test_adapter_under_autocast_keeps_the_pyramid_in_16_bits():
    """Under bf16 autocast the HF layer hands the attention core a bf16 value pyramid and fp32 sampling locations:
    the adapter takes the mixed-storage kernels (no fp32 copy of the pyramid, fp32 coordinates) and stays within bf16
    error of transformers' own module."""
    from msda_triton_amd import functional
    from msda_triton_amd.hf_adapter import replace_hf_msda
    dev = "cuda:0"
    layer = _hf_attention_layer().to(dev)
    kw = _layer_inputs(dev, levels=((12, 10), (6, 5), (3, 3)), Q=40)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ref = _run(layer, kw)
    assert replace_hf_msda(layer) == 1
    seen = []
    orig = functional.msda_hip_fwd

    def spy(img, *a, **k):
        seen.append((img.dtype, a[1].dtype))
        return orig(img, *a, **k)

    functional.msda_hip_fwd = spy
    try:
        with functional.KernelTimer():  # (keeps the call on the Python launchers, where the spy sits)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                got = _run(layer, kw)
    finally:
        functional.msda_hip_fwd = orig
    assert seen == [(torch.bfloat16, torch.float32)]
    err = float((got.detach().float() - ref.detach().float()).norm() / ref.detach().float().norm())
    assert err < 3e-2, err
    got.float().sum().backward()
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in layer.parameters())
