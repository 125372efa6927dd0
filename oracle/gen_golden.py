"""Generate tests/golden/*.npz -- the vectors that pin oracle/meme_oracle.py.

TEST INFRASTRUCTURE.  Run in the build container only (needs `transformers`):

    python -m oracle.gen_golden

The reference scripts cannot be imported here (torchvision / timm missing, hub
downloads), so the pin is the third-party library the reference itself calls:
transformers' BertModel / ViTModel built from explicit local configs
(attn_implementation="eager", every dropout 0), wired exactly as
example_scripts/Multimodal_example_task2C.txt:152-197 wires its towers, with
nn.CrossEntropyLoss + torch.optim.Adam(lr=2e-5) as in ...task2C.txt:248-249.
Parameters come from oracle.meme_oracle.init_params(seed) (same state-dict
names), so a fixture holds only inputs, checksums and expected outputs.
"""
from __future__ import annotations

import os
import re
import sys

import numpy as np
import torch
import torch.nn as nn

from . import meme_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class HFReferenceModel(nn.Module):
    """Organizers' wiring (...task2C.txt:152-197) over transformers towers."""

    def __init__(self, cfg: O.OracleConfig):
        super().__init__()
        from transformers import BertConfig, BertModel, ViTConfig, ViTModel
        t, v = cfg.text, cfg.image
        bc = BertConfig(vocab_size=t.vocab_size, hidden_size=t.hidden, num_hidden_layers=t.layers,
                        num_attention_heads=t.heads, intermediate_size=t.intermediate,
                        max_position_embeddings=t.max_position, type_vocab_size=max(t.type_vocab, 1),
                        layer_norm_eps=t.ln_eps, hidden_act="gelu", hidden_dropout_prob=0.0,
                        attention_probs_dropout_prob=0.0, pad_token_id=0)
        bc._attn_implementation = "eager"
        self.bert = BertModel(bc, add_pooling_layer=False)
        vc = ViTConfig(hidden_size=v.hidden, num_hidden_layers=v.layers, num_attention_heads=v.heads,
                       intermediate_size=v.intermediate, image_size=v.image_size, patch_size=v.patch,
                       num_channels=v.channels, layer_norm_eps=v.ln_eps, hidden_act="gelu",
                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, qkv_bias=True)
        vc._attn_implementation = "eager"
        self.image_model = ViTModel(vc, add_pooling_layer=False)
        self.bert_drop = nn.Dropout(0.0)
        self.bert_fc = nn.Linear(t.hidden, cfg.proj)
        self.image_fc = nn.Linear(v.hidden, cfg.proj)
        self.fusion_fc = nn.Linear(2 * cfg.proj, cfg.proj)
        self.output_fc = nn.Linear(cfg.proj, cfg.num_classes)
        self.pool = cfg.pool

    def forward(self, text, image, mask):
        h = self.bert(text, attention_mask=mask, return_dict=False)[0]
        h = h[:, -1, :] if self.pool == "last" else h[:, 0]
        t = self.bert_fc(self.bert_drop(h))
        v = self.image_fc(self.image_model(pixel_values=image, return_dict=False)[0][:, 0])
        return self.output_fc(self.fusion_fc(torch.cat((t, v), dim=1)))


_VIT_RENAMES = (  # transformers 4.39.2 names (the reference's pin) -> names in the installed 5.x
    (r"encoder\.layer\.(\d+)\.attention\.attention\.query", r"layers.\1.attention.q_proj"),
    (r"encoder\.layer\.(\d+)\.attention\.attention\.key", r"layers.\1.attention.k_proj"),
    (r"encoder\.layer\.(\d+)\.attention\.attention\.value", r"layers.\1.attention.v_proj"),
    (r"encoder\.layer\.(\d+)\.attention\.output\.dense", r"layers.\1.attention.o_proj"),
    (r"encoder\.layer\.(\d+)\.intermediate\.dense", r"layers.\1.mlp.fc1"),
    (r"encoder\.layer\.(\d+)\.output\.dense", r"layers.\1.mlp.fc2"),
    (r"encoder\.layer\.(\d+)\.layernorm_", r"layers.\1.layernorm_"),
)


def hf_name(k: str, sd) -> str:
    """Oracle (4.39.2-style) parameter name -> name used by the installed transformers."""
    if k in sd or not k.startswith("image_model."):
        return k
    for pat, rep in _VIT_RENAMES:
        k2 = re.sub(pat, rep, k)
        if k2 != k:
            return k2
    return k


def load_oracle_params(model: nn.Module, p: O.Params):
    sd = model.state_dict()
    p = {hf_name(k, sd): v for k, v in p.items()}
    missing = [k for k in p if k not in sd]
    assert not missing, f"oracle names absent from the transformers modules: {missing[:5]}"
    extra = [k for k in sd if k not in p and not k.endswith("position_ids") and "token_type_ids" not in k]
    assert not extra, f"transformers parameters the oracle does not model: {extra[:5]}"
    model.load_state_dict({k: v.clone() for k, v in p.items()}, strict=False)


def checksum(p: O.Params) -> np.ndarray:
    return np.array([float(sum(v.double().sum() for v in p.values())),
                     float(sum(v.double().abs().sum() for v in p.values()))])


SAMPLE = 6


def sample_index(n: int) -> torch.Tensor:
    """SAMPLE evenly spaced flat indices of an n-element tensor (float32 linspace, as the first fixtures were made;
    exact integer arithmetic once float32 can no longer represent n)."""
    if n <= (1 << 24):
        return torch.linspace(0, n - 1, SAMPLE).long()
    return (torch.arange(SAMPLE, dtype=torch.int64) * (n - 1)) // (SAMPLE - 1)


def sample_of(t: torch.Tensor) -> np.ndarray:
    f = t.detach().reshape(-1)
    return f[sample_index(f.numel())].numpy().astype(np.float32)


def gen_case(name: str, cfg: O.OracleConfig, batch: int, seq: int, seed: int, steps: int,
             with_grads: bool, all_ones_mask: bool = False, store_inputs: bool = True):
    """store_inputs=False: the batch is NOT stored (a 32-image batch is 19 MB); the test regenerates it with
    O.synthetic_batch(cfg, batch, seq, seed=1234 + seed) (torch's CPU generator is deterministic) and checks it against
    the stored checksums."""
    torch.manual_seed(0)
    p = O.init_params(cfg, seed)
    text, image, mask, labels = O.synthetic_batch(cfg, batch, seq, seed=1234 + seed,
                                                  all_ones_mask=all_ones_mask)
    model = HFReferenceModel(cfg)
    load_oracle_params(model, p)
    model.train()
    out = {"param_checksum": checksum(p), "seed": np.array(seed), "pool": np.array(cfg.pool),
           "batch": np.array(batch), "seq": np.array(seq)}
    if store_inputs:
        out.update({"text": text.numpy(), "image": image.numpy(), "mask": mask.numpy(), "labels": labels.numpy()})
    else:
        out["input_checksum"] = np.array([float(image.double().sum()), float(image.double().abs().sum()),
                                          float(text.sum()), float(mask.sum()), float(labels.sum())])
    if not with_grads:
        with torch.no_grad():
            out["logits"] = model(text, image, mask).numpy()
        th = model.bert(text, attention_mask=mask, return_dict=False)[0].detach()
        ih = model.image_model(pixel_values=image, return_dict=False)[0].detach()
        out["text_hidden_sample"] = th[:, :, :4].numpy()
        out["image_cls"] = ih[:, 0].numpy()
    else:
        crit = nn.CrossEntropyLoss()
        opt = torch.optim.Adam(model.parameters(), lr=2e-5)
        sd = model.state_dict()
        back = {hf_name(k, sd): k for k in p}                  # installed-transformers name -> oracle name
        names = [back[n] for n, _ in model.named_parameters()]
        for s in range(steps):
            opt.zero_grad()
            logits = model(text, image, mask)
            loss = crit(logits, labels)
            loss.backward()
            if s == 0:
                out["logits"] = logits.detach().numpy()
                out["loss"] = np.array(loss.item(), dtype=np.float64)
                out["grad_names"] = np.array(names)
                out["grad_norms"] = np.array([float(q.grad.double().norm()) for _, q in model.named_parameters()])
                out["grad_samples"] = np.stack([sample_of(q.grad) for _, q in model.named_parameters()])
            opt.step()
            if s in (0, steps - 1):
                out[f"param_samples_step{s + 1}"] = np.stack([sample_of(q) for _, q in model.named_parameters()])
                out[f"param_delta_norm_step{s + 1}"] = np.array(
                    [float((q.detach() - p[back[n]]).double().norm()) for n, q in model.named_parameters()])
        with torch.no_grad():
            out["logits_after"] = model(text, image, mask).numpy()
        out["steps"] = np.array(steps)
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


_CLIP_RENAMES = (   # oracle (ViT-style) name -> transformers CLIPVisionModel name
    (r"^image_model\.embeddings\.cls_token$", "vision_model.embeddings.class_embedding"),
    (r"^image_model\.embeddings\.position_embeddings$", "vision_model.embeddings.position_embedding.weight"),
    (r"^image_model\.embeddings\.patch_embeddings\.projection\.weight$", "vision_model.embeddings.patch_embedding.weight"),
    (r"^image_model\.pre_layernorm\.", "vision_model.pre_layrnorm."),
    (r"^image_model\.layernorm\.", "vision_model.post_layernorm."),
    (r"^image_model\.encoder\.layer\.(\d+)\.attention\.attention\.query\.", r"vision_model.encoder.layers.\1.self_attn.q_proj."),
    (r"^image_model\.encoder\.layer\.(\d+)\.attention\.attention\.key\.", r"vision_model.encoder.layers.\1.self_attn.k_proj."),
    (r"^image_model\.encoder\.layer\.(\d+)\.attention\.attention\.value\.", r"vision_model.encoder.layers.\1.self_attn.v_proj."),
    (r"^image_model\.encoder\.layer\.(\d+)\.attention\.output\.dense\.", r"vision_model.encoder.layers.\1.self_attn.out_proj."),
    (r"^image_model\.encoder\.layer\.(\d+)\.layernorm_before\.", r"vision_model.encoder.layers.\1.layer_norm1."),
    (r"^image_model\.encoder\.layer\.(\d+)\.layernorm_after\.", r"vision_model.encoder.layers.\1.layer_norm2."),
    (r"^image_model\.encoder\.layer\.(\d+)\.intermediate\.dense\.", r"vision_model.encoder.layers.\1.mlp.fc1."),
    (r"^image_model\.encoder\.layer\.(\d+)\.output\.dense\.", r"vision_model.encoder.layers.\1.mlp.fc2."),
)


def clip_name(k: str) -> str:
    for pat, rep in _CLIP_RENAMES:
        k2 = re.sub(pat, rep, k)
        if k2 != k:
            return k2
    raise KeyError(k)


def gen_clip_case(name: str = "clip_l14_336_2layer", layers: int = 2, batch: int = 2, seed: int = 9):
    """Pins the oracle's CLIP branch (quick-GELU, pre_layrnorm, bias-free 14x14 patch conv, eps 1e-5; BASELINE config 5's
    image tower at its true widths, `layers` blocks) against transformers' CLIPVisionModel: pooler_output and the
    gradient of sum(pooler_output * r) w.r.t. every parameter.  Inputs are regenerated from the seed by the test."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg = O.config5("cls", layers=layers)
    v = cfg.image
    p_all = O.init_params(cfg, seed)
    p = {k: t for k, t in p_all.items() if k.startswith("image_model.")}
    cc = CLIPVisionConfig(hidden_size=v.hidden, intermediate_size=v.intermediate, num_hidden_layers=v.layers,
                          num_attention_heads=v.heads, image_size=v.image_size, patch_size=v.patch, num_channels=v.channels,
                          hidden_act="quick_gelu", layer_norm_eps=v.ln_eps, attention_dropout=0.0)
    cc._attn_implementation = "eager"
    model = CLIPVisionModel(cc)
    sd = model.state_dict()
    mapped = {}
    strip = "vision_model.embeddings.class_embedding" not in sd      # transformers 5.x dropped the 4.39.2 `vision_model.` prefix
    fix = (lambda n: n[len("vision_model."):]) if strip else (lambda n: n)
    for k, t in p.items():
        hk = fix(clip_name(k))
        t2 = t.reshape(sd[hk].shape)          # cls_token (1,1,D) -> (D,), position_embeddings (1,N,D) -> (N,D)
        mapped[hk] = t2.clone()
    extra = [k for k in sd if k not in mapped and not k.endswith("position_ids")]
    assert not extra, extra
    model.load_state_dict(mapped, strict=False)
    g = torch.Generator().manual_seed(1000 + seed)
    image = torch.randn((batch, v.channels, v.image_size, v.image_size), generator=g)
    r = torch.randn((batch, v.hidden), generator=g)
    out = model(pixel_values=image).pooler_output
    (out * r).sum().backward()
    back = {fix(clip_name(k)): k for k in p}
    names = [back[n] for n, _ in model.named_parameters()]
    res = {"seed": np.array(seed), "layers": np.array(layers), "batch": np.array(batch),
           "input_checksum": np.array([float(image.double().sum()), float(image.double().abs().sum()), float(r.double().sum())]),
           "param_checksum": checksum(p), "pooler_output": out.detach().numpy(),
           "grad_names": np.array(names),
           "grad_norms": np.array([float(q.grad.double().norm()) for _, q in model.named_parameters()]),
           "grad_samples": np.stack([sample_of(q.grad) for _, q in model.named_parameters()])}
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **res)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def gen_resnet_case(name: str = "resnet_1111_w64", layers=(1, 1, 1, 1), width: int = 64, batch: int = 2, size: int = 64, seed: int = 13):
    """Pins oracle/resnet_oracle.py against transformers' ResNetModel (bottleneck, stride on the 3x3 conv = torchvision
    v1.5): pooled features, the gradient of sum(pooled * r) w.r.t. every parameter, running statistics after the step."""
    from transformers import ResNetConfig, ResNetModel
    from . import resnet_oracle as R
    p = {k: v for k, v in R.resnet_init(layers, width, 10, seed).items() if not k.startswith("fc.")}
    cfg = ResNetConfig(num_channels=3, embedding_size=width, hidden_sizes=[width * 4, width * 8, width * 16, width * 32],
                       depths=list(layers), layer_type="bottleneck", hidden_act="relu", downsample_in_bottleneck=False)
    model = ResNetModel(cfg)
    sd = model.state_dict()

    def hf(k):
        if k.startswith("conv1."):
            return "embedder.embedder.convolution." + k[6:]
        if k.startswith("bn1."):
            return "embedder.embedder.normalization." + k[4:]
        m_ = re.match(r"layer(\d)\.(\d+)\.(conv|bn)(\d)\.(.*)", k)
        if m_:
            li, bi, kind, idx, rest = m_.groups()
            return f"encoder.stages.{int(li) - 1}.layers.{bi}.layer.{int(idx) - 1}.{'convolution' if kind == 'conv' else 'normalization'}.{rest}"
        m_ = re.match(r"layer(\d)\.(\d+)\.downsample\.(\d)\.(.*)", k)
        li, bi, idx, rest = m_.groups()
        return f"encoder.stages.{int(li) - 1}.layers.{bi}.shortcut.{'convolution' if idx == '0' else 'normalization'}.{rest}"

    mapped = {hf(k): v.clone() for k, v in p.items()}
    missing = [k for k in mapped if k not in sd]
    assert not missing, missing[:4]
    extra = [k for k in sd if k not in mapped and "running" not in k and "num_batches" not in k]
    assert not extra, extra[:4]
    model.load_state_dict(mapped, strict=False)
    model.train()
    g = torch.Generator().manual_seed(2000 + seed)
    image = torch.randn((batch, 3, size, size), generator=g)
    r = torch.randn((batch, width * 32), generator=g)
    out = model(pixel_values=image).pooler_output.flatten(1)
    (out * r).sum().backward()
    back = {hf(k): k for k in p}
    names = [back[n] for n, _ in model.named_parameters()]
    after = model.state_dict()
    res = {"seed": np.array(seed), "layers": np.array(layers), "width": np.array(width), "batch": np.array(batch), "size": np.array(size),
           "input_checksum": np.array([float(image.double().sum()), float(image.double().abs().sum()), float(r.double().sum())]),
           "param_checksum": checksum(p), "pooled": out.detach().numpy(), "grad_names": np.array(names),
           "grad_norms": np.array([float(q.grad.double().norm()) for _, q in model.named_parameters()]),
           "grad_samples": np.stack([sample_of(q.grad) for _, q in model.named_parameters()]),
           "bn1_running_mean": after["embedder.embedder.normalization.running_mean"].numpy(),
           "bn1_running_var": after["embedder.embedder.normalization.running_var"].numpy(),
           "last_running_var": after[hf(f"layer4.{layers[3] - 1}.bn3.weight").replace("weight", "running_var")].numpy()}
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **res)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def gen_index_fixtures():
    """(iii) bit-exact index fixtures: patch order of a counting image, token gather rows."""
    img = torch.arange(2 * 3 * 32 * 32, dtype=torch.float32).view(2, 3, 32, 32)
    # pin patchify() against conv2d with one-hot kernels (what the patch-embed conv computes)
    K = 3 * 16 * 16
    w = torch.eye(K).view(K, 3, 16, 16)
    conv = torch.nn.functional.conv2d(img, w, stride=16)          # [2,K,2,2]
    ref = conv.flatten(2).transpose(1, 2)                          # HF ViTPatchEmbeddings order
    assert torch.equal(ref, O.patchify(img, 16))
    ids = torch.tensor([[2, 7, 7, 0], [2, 300, 7, 5]])
    table = torch.arange(512 * 4, dtype=torch.float32).view(512, 4)
    np.savez_compressed(os.path.join(GOLDEN, "index_fixtures.npz"),
                        counting_image=img.numpy(), patches=ref.numpy(),
                        ids=ids.numpy(), table=table.numpy(),
                        gathered=torch.nn.functional.embedding(ids, table).numpy())
    print("wrote index_fixtures.npz")


def gen_focal_case():
    """sigmoid focal loss (Kevin's criterion, Multimodal_example_task2C.py:167,711 = torchvision.ops.sigmoid_focal_loss, not
    installed here) pinned to the other implementation of the same detectron formula that IS importable: transformers' DETR
    loss helper (inputs [B, 1], num_boxes = B: loss.mean(1).sum() / B = the mean over the batch)."""
    from transformers.loss.loss_for_object_detection import sigmoid_focal_loss as hf_focal
    g = torch.Generator().manual_seed(21)
    out = {}
    for i, (alpha, gamma) in enumerate(((0.25, 2.0), (0.5, 1.0), (-1.0, 2.0))):
        x = (torch.randn(48, generator=g) * 3).requires_grad_(True)
        t = (torch.rand(48, generator=g) < 0.28).float()
        loss = hf_focal(x[:, None], t[:, None], num_boxes=x.numel(), alpha=alpha, gamma=gamma)
        loss.backward()
        out.update({f"x{i}": x.detach().numpy(), f"t{i}": t.numpy(), f"loss{i}": loss.detach().numpy(), f"dx{i}": x.grad.numpy(),
                    f"alpha{i}": np.float32(alpha), f"gamma{i}": np.float32(gamma)})
    path = os.path.join(GOLDEN, "focal_hf.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


def main():
    torch.set_num_threads(8)
    if "--only-focal" in sys.argv:
        gen_focal_case()
        return
    if "--only-resnet" in sys.argv:
        gen_resnet_case()
        return
    if "--only-clip" in sys.argv:
        gen_clip_case()
        return
    if "--only-b32" in sys.argv:
        gen_case("config3_b32", O.config3("cls"), batch=32, seq=128, seed=4, steps=1, with_grads=True, store_inputs=False)
        return
    gen_index_fixtures()
    gen_case("tiny_cls", O.tiny_config("cls"), batch=4, seq=16, seed=1, steps=3, with_grads=True)
    gen_case("tiny_last", O.tiny_config("last"), batch=4, seq=16, seed=2, steps=3, with_grads=True)
    if "--no-full" not in sys.argv:
        gen_case("config3_b2", O.config3("cls"), batch=2, seq=128, seed=3, steps=0, with_grads=False)
    if "--b32" in sys.argv:      # the benchmarked configuration itself (BASELINE.json configs[2]): one full step at batch 32
        gen_case("config3_b32", O.config3("cls"), batch=32, seq=128, seed=4, steps=1, with_grads=True, store_inputs=False)


if __name__ == "__main__":
    main()
