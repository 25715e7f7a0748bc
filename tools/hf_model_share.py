"""dev tool: share of a Deformable-DETR / Grounding-DINO training step spent inside multi-scale deformable attention,
with transformers' own pure-PyTorch module and with this package's kernels (replace_hf_msda).  Random-init model from a
config (no download), COCO-like 800 x 1066 input, fp32 and bf16 autocast.

    python tools/hf_model_share.py [--model deformable_detr|grounding_dino] [out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from transformers import DeformableDetrConfig, DeformableDetrModel, ResNetConfig  # noqa: E402

from msda_triton_amd.hf_adapter import replace_hf_msda  # noqa: E402

dev = "cuda:0"


def build_gdino():
    """Grounding-DINO-T shaped: Swin-T backbone, BERT-base-sized text encoder cut to 2 layers (random init, the text side
    is not what is measured), d_model 256, 6 + 6 layers, 900 queries"""
    from transformers import BertConfig, GroundingDinoConfig, GroundingDinoModel, SwinConfig
    bb = SwinConfig(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7,
                    out_features=["stage2", "stage3", "stage4"], drop_path_rate=0.0, hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0)
    txt = BertConfig(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cfg = GroundingDinoConfig(backbone_config=bb, use_timm_backbone=False, use_pretrained_backbone=False, backbone=None,
                              text_config=txt, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                              fusion_dropout=0.0, fusion_droppath=0.0)
    torch.manual_seed(0)
    m = GroundingDinoModel(cfg).to(dev).train()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("sampling_offsets.weight"):
                p.normal_(0, 0.02)
    return m


def build():
    bb = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                      layer_type="bottleneck", out_features=["stage2", "stage3", "stage4"])
    cfg = DeformableDetrConfig(use_timm_backbone=False, use_pretrained_backbone=False, backbone_config=bb, backbone=None,
                               dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)  # d_model 256, 6 + 6 layers, 300 queries
    torch.manual_seed(0)
    m = DeformableDetrModel(cfg).to(dev).train()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("sampling_offsets.weight"):
                p.normal_(0, 0.02)
    return m


class Timed(torch.nn.Module):
    """wraps an attention-core module: device time of its forward (events) is accumulated; the backward of the core is
    timed through autograd hooks on its output / inputs"""
    acc = []

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = self.inner(*a, **k)
        e1.record()
        Timed.acc.append((e0, e1))
        if out.requires_grad:
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            out.register_hook(lambda g: (b0.record(), g)[1])           # backward enters the core
            value = a[0] if a else k["value"]
            if value.requires_grad:
                value.register_hook(lambda g: (b1.record(), Timed.acc.append((b0, b1)), g)[2])  # ... and has left it
        return out


def wrap(model):
    targets = [(parent, name, child) for parent in model.modules() for name, child in parent.named_children()
               if type(child).__name__ == "MultiScaleDeformableAttention" and not isinstance(parent, Timed)]
    for parent, name, child in targets:
        setattr(parent, name, Timed(child))
    return len(targets)


def measure(model, x, mask, autocast, steps=5, extra=None):
    def step():
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            out = model(pixel_values=x, pixel_mask=mask, **(extra or {}))
        (out.last_hidden_state.float() ** 2).mean().backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    Timed.acc.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) * 1e3 / steps
    msda_ms = sum(a.elapsed_time(b) for a, b in Timed.acc) / steps
    return {"step_ms": round(step_ms, 2), "msda_ms": round(msda_ms, 2), "msda_share": round(msda_ms / step_ms, 3)}


def main():
    args = sys.argv[1:]
    which = "deformable_detr"
    if "--model" in args:
        i = args.index("--model")
        which = args[i + 1]
        del args[i:i + 2]
    out_path = args[0] if args else os.path.join(ROOT, "profiles", "hf_model_msda_share_%s.json" % which)
    torch.manual_seed(1)
    x = torch.randn(2, 3, 800, 1066, device=dev)
    mask = torch.ones(2, 800, 1066, dtype=torch.long, device=dev)
    extra = None
    if which == "grounding_dino":
        ids = torch.randint(1000, 20000, (2, 12), device=dev)
        extra = dict(input_ids=ids, attention_mask=torch.ones_like(ids), token_type_ids=torch.zeros_like(ids))
        what = ("GroundingDinoModel (transformers %s), random init, Swin-T-shaped backbone, 2-layer BERT text encoder, d_model 256, "
                "6 + 6 layers, 900 queries, 4 levels x 4 points, input 2 x 3 x 800 x 1066 + 12 text tokens, fwd + bwd")
    else:
        what = ("DeformableDetrModel (transformers %s), random init, ResNet-50-shaped backbone, d_model 256, 6 + 6 layers, "
                "300 queries, 4 levels x 4 points, input 2 x 3 x 800 x 1066, fwd + bwd")
    res = {"model": what % __import__("transformers").__version__,
           "timing": "msda_ms: device time between events around the attention core's forward and (autograd hooks) its backward, "
                     "summed over the 12 layers; step_ms: wall time of a training step (no optimizer)"}
    for impl in ("transformers", "msda_triton_amd"):
        model = build_gdino() if which == "grounding_dino" else build()
        if impl == "msda_triton_amd":
            assert replace_hf_msda(model) == 12
        wrap(model)
        res[impl] = {"fp32": measure(model, x, mask, False, extra=extra), "bf16_autocast": measure(model, x, mask, True, extra=extra)}
        del model
        torch.cuda.empty_cache()
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
