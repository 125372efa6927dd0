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


def sample_of(t: torch.Tensor) -> np.ndarray:
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, SAMPLE).long()
    return f[idx].numpy().astype(np.float32)


def gen_case(name: str, cfg: O.OracleConfig, batch: int, seq: int, seed: int, steps: int,
             with_grads: bool, all_ones_mask: bool = False):
    torch.manual_seed(0)
    p = O.init_params(cfg, seed)
    text, image, mask, labels = O.synthetic_batch(cfg, batch, seq, seed=1234 + seed,
                                                  all_ones_mask=all_ones_mask)
    model = HFReferenceModel(cfg)
    load_oracle_params(model, p)
    model.train()
    out = {"text": text.numpy(), "image": image.numpy(), "mask": mask.numpy(), "labels": labels.numpy(),
           "param_checksum": checksum(p), "seed": np.array(seed), "pool": np.array(cfg.pool)}
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


def main():
    torch.set_num_threads(8)
    gen_index_fixtures()
    gen_case("tiny_cls", O.tiny_config("cls"), batch=4, seq=16, seed=1, steps=3, with_grads=True)
    gen_case("tiny_last", O.tiny_config("last"), batch=4, seq=16, seed=2, steps=3, with_grads=True)
    if "--no-full" not in sys.argv:
        gen_case("config3_b2", O.config3("cls"), batch=2, seq=128, seed=3, steps=0, with_grads=False)


if __name__ == "__main__":
    main()
