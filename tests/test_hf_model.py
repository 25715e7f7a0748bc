"""Model-level parity with Hugging Face transformers (the reference README's demo, README.md:25-37: a detection model
gives the same results with either implementation): a tiny random-init DeformableDetrModel built from a config (no
download), every MultiScaleDeformableAttention swapped by replace_hf_msda, outputs and parameter gradients against
transformers' own pure-PyTorch path."""
import pytest
import torch

transformers = pytest.importorskip("transformers")


def tiny_deformable_detr(seed=0):
    from transformers import DeformableDetrConfig, DeformableDetrModel, ResNetConfig
    bb = ResNetConfig(num_channels=3, embedding_size=16, hidden_sizes=[16, 32, 64, 128], depths=[1, 1, 1, 1],
                      layer_type="basic", out_features=["stage2", "stage3", "stage4"])
    cfg = DeformableDetrConfig(use_timm_backbone=False, use_pretrained_backbone=False, backbone_config=bb, backbone=None,
                               d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                               decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, num_queries=30,
                               num_feature_levels=4, encoder_n_points=4, decoder_n_points=4, dropout=0.0,
                               attention_dropout=0.0, activation_dropout=0.0)
    torch.manual_seed(seed)
    model = DeformableDetrModel(cfg)
    with torch.no_grad():  # HF zero-initialises the sampling offsets' weights: make the sampling pattern data-dependent
        for name, prm in model.named_parameters():
            if name.endswith("sampling_offsets.weight"):
                prm.normal_(0, 0.05)
    return model.train()  # (dropout is 0: train mode only so that every parameter takes part in backward)


WATCHED = ("encoder.layers.0.self_attn.value_proj.weight", "encoder.layers.1.self_attn.sampling_offsets.weight",
           "encoder.layers.0.self_attn.attention_weights.bias", "decoder.layers.1.encoder_attn.value_proj.weight",
           "decoder.layers.0.encoder_attn.sampling_offsets.weight", "decoder.layers.1.encoder_attn.output_proj.weight",
           "input_proj.0.0.weight")


def run_model(model, x, mask, autocast_dtype=None):
    model.zero_grad(set_to_none=True)
    ctx = torch.autocast(x.device.type, dtype=autocast_dtype) if autocast_dtype is not None else torch.autocast(x.device.type, enabled=False)
    with ctx:
        out = model(pixel_values=x, pixel_mask=mask)
    hs = out.last_hidden_state
    # a fixed random linear functional of the decoder states (NOT mean(hs^2): behind the final LayerNorm that is
    # nearly constant, its gradient is pure cancellation noise in fp32)
    w = torch.randn(hs.shape, generator=torch.Generator().manual_seed(11)).to(hs.device)
    (hs.float() * w).sum().backward()
    named = dict(model.named_parameters())
    return hs.detach().float(), out.encoder_last_hidden_state.detach().float(), {k: named[k].grad.detach().float().clone() for k in WATCHED}


def _inputs(device):
    torch.manual_seed(3)
    x = torch.randn(2, 3, 96, 128, device=device)
    mask = torch.ones(2, 96, 128, dtype=torch.long, device=device)
    mask[1, :, 100:] = 0  # one padded image: the valid-ratio / padding-mask logic feeds the reference points
    return x, mask


def test_tiny_deformable_detr_matches_hf_on_cpu():
    from msda_triton_amd.hf_adapter import replace_hf_msda
    model = tiny_deformable_detr()
    x, mask = _inputs("cpu")
    hs0, enc0, g0 = run_model(model, x, mask)
    assert replace_hf_msda(model) == 4  # 2 encoder self-attentions + 2 decoder cross-attentions
    hs1, enc1, g1 = run_model(model, x, mask)
    torch.testing.assert_close(enc1, enc0, atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(hs1, hs0, atol=1e-5, rtol=1e-4)
    for k in WATCHED:  # relative L2 error per gradient tensor (fp32 round-off of two summation orders)
        err = float((g1[k] - g0[k]).norm() / g0[k].norm().clamp_min(1e-30))
        assert err < 1e-4, (k, err)


@pytest.mark.gpu
def test_tiny_deformable_detr_matches_hf_on_gpu_fp32():
    """fp32: the whole model's outputs (encoder memory, decoder states) and a spread of parameter gradients — value
    projections (grad_value path), sampling offsets (grad_loc path), attention weights (grad_attn path), the input
    projection below the first encoder layer — against transformers' own grid_sample implementation on the same GPU."""
    from msda_triton_amd.functional import KernelTimer
    from msda_triton_amd.hf_adapter import replace_hf_msda
    dev = "cuda:0"
    model = tiny_deformable_detr().to(dev)
    x, mask = _inputs(dev)
    hs0, enc0, g0 = run_model(model, x, mask)
    assert replace_hf_msda(model) == 4
    with KernelTimer() as kt:
        hs1, enc1, g1 = run_model(model, x, mask)
        torch.cuda.synchronize()
    s = kt.summary()
    assert s["msda_fwd"][0] == 4 and s["msda_bwd_sample"][0] == 4 and s["msda_bwd_value"][0] == 4, s  # the HIP kernels ran
    torch.testing.assert_close(enc1, enc0, atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(hs1, hs0, atol=1e-4, rtol=1e-3)
    for k in WATCHED:  # relative L2 error per gradient tensor (fp32 round-off of two different summation orders)
        err = float((g1[k] - g0[k]).norm() / g0[k].norm().clamp_min(1e-30))
        assert err < 2e-3, (k, err)


@pytest.mark.gpu
def test_tiny_deformable_detr_matches_hf_on_gpu_bf16_autocast():
    """Under bf16 autocast both implementations carry bf16 round-off; they agree to bf16 accuracy (relative L2 error of
    the decoder states and of every watched gradient), and the adapter keeps the value pyramid in 16 bits (mixed-storage
    kernels)."""
    from msda_triton_amd.hf_adapter import replace_hf_msda
    dev = "cuda:0"
    model = tiny_deformable_detr().to(dev)
    x, mask = _inputs(dev)
    hs0, enc0, g0 = run_model(model, x, mask, torch.bfloat16)
    hs_fp32, _, _ = run_model(model, x, mask)  # the yardstick: how far bf16 autocast itself is from fp32
    assert replace_hf_msda(model) == 4
    hs1, enc1, g1 = run_model(model, x, mask, torch.bfloat16)

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-12))

    noise = rel(hs0, hs_fp32)
    assert rel(hs1, hs0) < max(3 * noise, 3e-2), (rel(hs1, hs0), noise)
    assert rel(enc1, enc0) < 3e-2
    for k in WATCHED:
        assert torch.isfinite(g1[k]).all()
        assert rel(g1[k], g0[k]) < 0.15, (k, rel(g1[k], g0[k]))


# ------------------------------------------------------------------------------------------
# Grounding-DINO (the reference README's parity demo is run on it, README.md:25-37; BASELINE north_star: "drops into
# Grounding-DINO-style models unchanged"): a tiny random-init GroundingDinoModel from configs — Swin backbone, BERT text
# encoder, text-fused encoder layers, two-stage query selection, 2 + 2 deformable layers.  Its decoder calls the core with
# text-fused queries and its own spatial_shapes_list plumbing.
# ------------------------------------------------------------------------------------------
def tiny_grounding_dino(seed=0):
    from transformers import BertConfig, GroundingDinoConfig, GroundingDinoModel, SwinConfig
    bb = SwinConfig(image_size=64, patch_size=4, num_channels=3, embed_dim=16, depths=[1, 1, 1, 1], num_heads=[1, 2, 4, 8],
                    window_size=4, out_features=["stage2", "stage3", "stage4"], drop_path_rate=0.0, hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0)
    txt = BertConfig(vocab_size=200, hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                     max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cfg = GroundingDinoConfig(backbone_config=bb, use_timm_backbone=False, use_pretrained_backbone=False, backbone=None,
                              text_config=txt, d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                              decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, num_queries=20,
                              num_feature_levels=4, encoder_n_points=4, decoder_n_points=4, dropout=0.0,
                              attention_dropout=0.0, activation_dropout=0.0, max_text_len=16, fusion_dropout=0.0,
                              fusion_droppath=0.0)
    torch.manual_seed(seed)
    model = GroundingDinoModel(cfg)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if name.endswith("sampling_offsets.weight"):
                prm.normal_(0, 0.05)
    return model.train()


GDINO_WATCHED = ("encoder.layers.0.deformable_layer.self_attn.value_proj.weight",
                 "encoder.layers.1.deformable_layer.self_attn.sampling_offsets.weight",
                 "encoder.layers.0.deformable_layer.self_attn.attention_weights.bias",
                 "decoder.layers.1.encoder_attn.value_proj.weight",
                 "decoder.layers.0.encoder_attn.sampling_offsets.weight",
                 "decoder.layers.1.encoder_attn.output_proj.weight",
                 "input_proj_vision.0.0.weight", "text_projection.weight")


def _gdino_inputs(device):
    x, mask = _inputs(device)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(1, 200, (2, 9), generator=g).to(device)
    return dict(pixel_values=x, pixel_mask=mask, input_ids=ids, attention_mask=torch.ones_like(ids),
                token_type_ids=torch.zeros_like(ids))


def run_gdino(model, inputs, autocast_dtype=None):
    model.zero_grad(set_to_none=True)
    dev = inputs["pixel_values"].device.type
    ctx = torch.autocast(dev, dtype=autocast_dtype) if autocast_dtype is not None else torch.autocast(dev, enabled=False)
    with ctx:
        out = model(**inputs)
    hs, enc = out.last_hidden_state, out.encoder_last_hidden_state_vision
    w = torch.randn(hs.shape, generator=torch.Generator().manual_seed(11)).to(hs.device)
    w2 = torch.randn(enc.shape, generator=torch.Generator().manual_seed(12)).to(hs.device)
    ((hs.float() * w).sum() + 0.1 * (enc.float() * w2).sum()).backward()
    named = dict(model.named_parameters())
    return (hs.detach().float(), enc.detach().float(), out.init_reference_points.detach().float(),
            {k: named[k].grad.detach().float().clone() for k in GDINO_WATCHED})


def test_tiny_grounding_dino_matches_hf_on_cpu():
    from msda_triton_amd.hf_adapter import replace_hf_msda
    model = tiny_grounding_dino()
    inputs = _gdino_inputs("cpu")
    hs0, enc0, ref0, g0 = run_gdino(model, inputs)
    assert replace_hf_msda(model) == 4  # 2 text-fused encoder layers + 2 decoder cross-attentions
    hs1, enc1, ref1, g1 = run_gdino(model, inputs)
    torch.testing.assert_close(ref1, ref0, atol=1e-5, rtol=1e-4)  # (the two-stage query selection picked the same proposals)
    torch.testing.assert_close(enc1, enc0, atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(hs1, hs0, atol=1e-5, rtol=1e-4)
    for k in GDINO_WATCHED:
        err = float((g1[k] - g0[k]).norm() / g0[k].norm().clamp_min(1e-30))
        assert err < 1e-4, (k, err)


@pytest.mark.gpu
def test_tiny_grounding_dino_matches_hf_on_gpu_fp32():
    """fp32 on the GPU: encoder memory, the selected reference points, decoder states and eight parameter gradients (both
    deformable stacks, the vision input projection, the text projection) against transformers' own core; the HIP kernels
    are asserted to have run (2 encoder + 2 decoder calls)."""
    from msda_triton_amd.functional import KernelTimer
    from msda_triton_amd.hf_adapter import replace_hf_msda
    dev = "cuda:0"
    model = tiny_grounding_dino().to(dev)
    inputs = _gdino_inputs(dev)
    hs0, enc0, ref0, g0 = run_gdino(model, inputs)
    assert replace_hf_msda(model) == 4
    with KernelTimer() as kt:
        hs1, enc1, ref1, g1 = run_gdino(model, inputs)
        torch.cuda.synchronize()
    s = kt.summary()
    assert s["msda_fwd"][0] == 4 and s["msda_bwd_sample"][0] == 4 and s["msda_bwd_value"][0] == 4, s
    torch.testing.assert_close(ref1, ref0, atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(enc1, enc0, atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(hs1, hs0, atol=1e-4, rtol=1e-3)
    for k in GDINO_WATCHED:
        err = float((g1[k] - g0[k]).norm() / g0[k].norm().clamp_min(1e-30))
        assert err < 2e-3, (k, err)


class _PinnedTopk:
    """The two-stage query selection of GroundingDinoModel is ONE ``torch.topk`` over the encoder's proposal logits
    (modeling_grounding_dino.py: ``topk_proposals = torch.topk(topk_logits, topk, dim=1)[1]``).  Under bf16 autocast two
    runs whose encoder memories differ in the last bits can order near-tied logits differently, and decoder states of
    different proposals are not comparable.  record(): run with the real top-k and keep its indices; replay(): hand the
    recorded indices back (with the values found at them), so that both runs decode the SAME proposals and every
    comparison below is made on every run."""

    def __init__(self, monkeypatch):
        self.mp, self.real, self.indices, self.calls = monkeypatch, torch.topk, [], 0

    def record(self):
        def topk(x, k, dim=-1, **kw):
            out = self.real(x, k, dim=dim, **kw)
            self.indices.append(out[1].clone())
            return out
        self.mp.setattr(torch, "topk", topk)

    def replay(self):
        pending = list(self.indices)

        def topk(x, k, dim=-1, **kw):
            idx = pending.pop(0)
            self.calls += 1
            assert idx.shape[dim] == k
            return torch.return_types.topk((torch.gather(x, dim, idx), idx))
        self.mp.setattr(torch, "topk", topk)

    def restore(self):
        self.mp.setattr(torch, "topk", self.real)


@pytest.mark.gpu
def test_tiny_grounding_dino_matches_hf_on_gpu_bf16_autocast(monkeypatch):
    """bf16 autocast: encoder memory, reference points, decoder states and the eight watched gradients against
    transformers' own core — with the proposal selection pinned to the baseline run's (``_PinnedTopk``), so nothing is
    skipped whatever bf16 round-off does to near-tied proposal logits."""
    from msda_triton_amd.hf_adapter import replace_hf_msda
    dev = "cuda:0"
    model = tiny_grounding_dino().to(dev)
    inputs = _gdino_inputs(dev)
    pin = _PinnedTopk(monkeypatch)
    pin.record()
    hs0, enc0, ref0, g0 = run_gdino(model, inputs, torch.bfloat16)
    assert len(pin.indices) == 1  # (one selection per forward: the recipe above still describes the model)
    pin.restore()
    hs_fp32, _, _, _ = run_gdino(model, inputs)
    assert replace_hf_msda(model) == 4
    pin.replay()
    hs1, enc1, ref1, g1 = run_gdino(model, inputs, torch.bfloat16)
    pin.restore()
    assert pin.calls == 1

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-12))

    assert rel(enc1, enc0) < 3e-2
    assert rel(ref1, ref0) < 3e-2, rel(ref1, ref0)  # the same proposals: their boxes differ by the encoder's round-off only
    noise = rel(hs0, hs_fp32)  # how far bf16 autocast itself is from fp32 (its own top-k: an upper-ish yardstick)
    assert rel(hs1, hs0) < max(3 * noise, 3e-2), (rel(hs1, hs0), noise)
    for k in GDINO_WATCHED:
        assert torch.isfinite(g1[k]).all()
        assert rel(g1[k], g0[k]) < 0.2, (k, rel(g1[k], g0[k]))
