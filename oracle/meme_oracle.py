"""Pure-PyTorch fp32 CPU restatement of the Subtask-2C fine-tune step.

TEST INFRASTRUCTURE (see oracle/__init__.py) - never imported by the product.

What it restates, and where the reference does it (paths under /root/reference):

* model wiring ........ example_scripts/Multimodal_example_task2C.txt:152-197
  (text tower -> dropout -> Linear(768,512); image tower -> Linear(.,512);
  torch.cat -> Linear(1024,512) -> Linear(512,num_classes), no non-linearity)
* pooling ............. ...task2C.txt:178 takes ``[:, -1, :]`` (pool="last");
  example_scripts/Multimodal_example_task2C.py:359-360 takes ``[:, 0]``
  (pool="cls").
* loss / optimizer .... ...task2C.txt:248-249 (nn.CrossEntropyLoss, mean;
  optim.Adam(lr=2e-5)); Trainer variant AdamW b=(.9,.999) eps=1e-8 wd=0 with
  max_grad_norm=1.0 (example_scripts/DistilBERT_example_task2A.ipynb:3211-3213,
  3280).
* step ................ ...task2C.txt:200-223 (zero_grad, forward, loss,
  backward, optimizer.step).

The encoder arithmetic lives in third-party wheels that are NOT vendored in
the reference: transformers==4.39.2 (BertModel / DistilBertModel, call sites
...task2C.txt:158,175 and ...task2C.py:317,337), timm==0.9.16
(vit_base_patch16_224, ...task2C.py:82,569-570), torch==2.2.2 (poetry.lock).
This file restates their published algorithms:

* BERT (post-LN): emb = word + position + token_type(0) -> LayerNorm(1e-12);
  per layer: q,k,v = Linear(x); scores = q k^T / sqrt(d_h) + (1-mask)*min;
  softmax; ctx = P v; x = LN(x + Linear(ctx)); x = LN(x + W2 gelu_erf(W1 x)).
* ViT (pre-LN, timm vit_base_patch16_224 == HF ViTModel w/ eps 1e-6):
  conv16/s16 patch-embed (+bias) -> [cls; patches] + pos;
  per layer: x = x + Wo attn(LN1 x); x = x + W2 gelu_erf(W1 LN2 x); final LN;
  pooled = token 0.

Pinning: the reference has no tests and cannot be imported here (torchvision
and timm are not installed; weights are hub downloads), so parity is pinned
against transformers' own BertModel / ViTModel classes instantiated from
explicit local configs -- see oracle/gen_golden.py, tests/golden/*.npz and
tests/test_oracle_golden.py.  Dropout is p=0 everywhere.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, asdict
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


@dataclass
class TextConfig:
    vocab_size: int = 64000          # AraBERTv2 (model card)
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    max_position: int = 512
    type_vocab: int = 2              # 0 => DistilBERT-style (no token_type table)
    pad_token_id: int = 0            # nn.Embedding(padding_idx=...): that row receives no gradient
    ln_eps: float = 1e-12


@dataclass
class ImageConfig:
    image_size: int = 224
    patch: int = 16
    channels: int = 3
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    ln_eps: float = 1e-6             # timm ViT eps
    # CLIP vision tower (transformers CLIPVisionModel; BASELINE.json configs[4], reference evidence
    # example_scripts/mm_model_mm_example_task2C.py:49): quick-GELU, a LayerNorm on the embeddings in front
    # of the first block (pre_layrnorm), no bias on the patch conv
    act: str = "gelu"
    pre_ln: bool = False
    patch_bias: bool = True

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch) ** 2

    @property
    def n_tokens(self) -> int:
        return self.n_patches + 1


@dataclass
class OracleConfig:
    text: TextConfig = field(default_factory=TextConfig)
    image: ImageConfig = field(default_factory=ImageConfig)
    proj: int = 512                  # ...task2C.txt:161,165
    num_classes: int = 2
    pool: str = "cls"                # "cls" (...task2C.py:359) | "last" (...task2C.txt:178)

    def to_dict(self):
        return asdict(self)


def tiny_config(pool: str = "cls") -> OracleConfig:
    """Smallest shape the HIP kernels accept (hidden % 128 == 0, head dim 64)."""
    return OracleConfig(
        text=TextConfig(vocab_size=512, hidden=128, layers=2, heads=2,
                        intermediate=256, max_position=64, type_vocab=2),
        image=ImageConfig(image_size=32, patch=16, hidden=128, layers=2, heads=2,
                          intermediate=256),
        proj=128, num_classes=2, pool=pool)


def config5(pool: str = "cls", layers: int = 24) -> OracleConfig:
    """BASELINE.json configs[4]: CLIP ViT-L/14 @336 image tower (577 tokens, D 1024, quick-GELU, pre-LN, bias-free
    14x14 patch conv, LayerNorm eps 1e-5) + BERT-large (V = 30522), S = 256.  ``layers`` < 24 keeps every width and
    sequence length and only shortens the stacks (parity tests)."""
    return OracleConfig(
        text=TextConfig(vocab_size=30522, hidden=1024, layers=layers, heads=16, intermediate=4096, max_position=512),
        image=ImageConfig(image_size=336, patch=14, hidden=1024, layers=layers, heads=16, intermediate=4096,
                          ln_eps=1e-5, act="quick_gelu", pre_ln=True, patch_bias=False),
        proj=512, num_classes=2, pool=pool)


def config3(pool: str = "cls") -> OracleConfig:
    """BASELINE.json configs[2]: ViT-B/16 + BERT-base(V=64000), 224x224 + S=128."""
    return OracleConfig(pool=pool)


# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------

def _text_shapes(c: TextConfig, pfx: str = "bert.") -> Dict[str, Tuple[int, ...]]:
    D, I = c.hidden, c.intermediate
    s = {
        pfx + "embeddings.word_embeddings.weight": (c.vocab_size, D),
        pfx + "embeddings.position_embeddings.weight": (c.max_position, D),
    }
    if c.type_vocab > 0:
        s[pfx + "embeddings.token_type_embeddings.weight"] = (c.type_vocab, D)
    s[pfx + "embeddings.LayerNorm.weight"] = (D,)
    s[pfx + "embeddings.LayerNorm.bias"] = (D,)
    for i in range(c.layers):
        L = f"{pfx}encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[L + f"attention.self.{n}.weight"] = (D, D)
            s[L + f"attention.self.{n}.bias"] = (D,)
        s[L + "attention.output.dense.weight"] = (D, D)
        s[L + "attention.output.dense.bias"] = (D,)
        s[L + "attention.output.LayerNorm.weight"] = (D,)
        s[L + "attention.output.LayerNorm.bias"] = (D,)
        s[L + "intermediate.dense.weight"] = (I, D)
        s[L + "intermediate.dense.bias"] = (I,)
        s[L + "output.dense.weight"] = (D, I)
        s[L + "output.dense.bias"] = (D,)
        s[L + "output.LayerNorm.weight"] = (D,)
        s[L + "output.LayerNorm.bias"] = (D,)
    return s


def _image_shapes(c: ImageConfig, pfx: str = "image_model.") -> Dict[str, Tuple[int, ...]]:
    D, I = c.hidden, c.intermediate
    s = {
        pfx + "embeddings.cls_token": (1, 1, D),
        pfx + "embeddings.position_embeddings": (1, c.n_tokens, D),
        pfx + "embeddings.patch_embeddings.projection.weight": (D, c.channels, c.patch, c.patch),
    }
    if c.patch_bias:
        s[pfx + "embeddings.patch_embeddings.projection.bias"] = (D,)
    if c.pre_ln:
        s[pfx + "pre_layernorm.weight"] = (D,)
        s[pfx + "pre_layernorm.bias"] = (D,)
    for i in range(c.layers):
        L = f"{pfx}encoder.layer.{i}."
        s[L + "layernorm_before.weight"] = (D,)
        s[L + "layernorm_before.bias"] = (D,)
        for n in ("query", "key", "value"):
            s[L + f"attention.attention.{n}.weight"] = (D, D)
            s[L + f"attention.attention.{n}.bias"] = (D,)
        s[L + "attention.output.dense.weight"] = (D, D)
        s[L + "attention.output.dense.bias"] = (D,)
        s[L + "layernorm_after.weight"] = (D,)
        s[L + "layernorm_after.bias"] = (D,)
        s[L + "intermediate.dense.weight"] = (I, D)
        s[L + "intermediate.dense.bias"] = (I,)
        s[L + "output.dense.weight"] = (D, I)
        s[L + "output.dense.bias"] = (D,)
    s[pfx + "layernorm.weight"] = (D,)
    s[pfx + "layernorm.bias"] = (D,)
    return s


def _head_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    P = cfg.proj
    return {
        "bert_fc.weight": (P, cfg.text.hidden), "bert_fc.bias": (P,),
        "image_fc.weight": (P, cfg.image.hidden), "image_fc.bias": (P,),
        "fusion_fc.weight": (P, 2 * P), "fusion_fc.bias": (P,),
        "output_fc.weight": (cfg.num_classes, P), "output_fc.bias": (cfg.num_classes,),
    }


def param_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    s.update(_text_shapes(cfg.text))
    s.update(_image_shapes(cfg.image))
    s.update(_head_shapes(cfg))
    return s


def init_params(cfg: OracleConfig, seed: int = 0) -> Params:
    """Deterministic "pretrained-like" random init (no checkpoints offline).

    Encoder matrices / embeddings ~ N(0, 0.02) (the BERT / ViT initializer_range);
    biases ~ N(0, 0.02) and LayerNorm gamma = 1 + N(0, 0.02), beta ~ N(0, 0.02)
    so every bias / affine path is exercised by the parity tests; head Linear
    layers use nn.Linear's default U(-1/sqrt(fan_in), 1/sqrt(fan_in)).
    One torch CPU generator, tensors drawn in ``param_shapes`` order.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    out: Params = {}
    for name, shape in param_shapes(cfg).items():
        head = name.split(".")[0] in ("bert_fc", "image_fc", "fusion_fc", "output_fc")
        if head:
            fan_in = shape[1] if len(shape) == 2 else _head_shapes(cfg)[name.replace("bias", "weight")][1]
            b = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * b
        else:
            t = torch.randn(shape, generator=g, dtype=torch.float32) * 0.02
            if "LayerNorm.weight" in name or ("layernorm" in name and name.endswith("weight")):
                t = t + 1.0
        out[name] = t
    return out


def n_params(cfg: OracleConfig) -> int:
    return sum(int(torch.tensor(s).prod()) for s in param_shapes(cfg).values())


# --------------------------------------------------------------------------
# towers
# --------------------------------------------------------------------------

def _mha(x_q: torch.Tensor, wq, bq, wk, bk, wv, bv, heads: int,
         add_mask: Optional[torch.Tensor], p_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, S, D = x_q.shape
    dh = D // heads
    q = F.linear(x_q, wq, bq).view(B, S, heads, dh).transpose(1, 2)
    k = F.linear(x_q, wk, bk).view(B, S, heads, dh).transpose(1, 2)
    v = F.linear(x_q, wv, bv).view(B, S, heads, dh).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
    if add_mask is not None:
        scores = scores + add_mask
    p = torch.softmax(scores, dim=-1)
    if p_mask is not None:            # nn.Dropout on the probabilities, mask given as 0 or 1/(1-p)
        p = p * p_mask
    ctx = torch.matmul(p, v).transpose(1, 2).reshape(B, S, D)
    return ctx


def text_tower(p: Params, ids: torch.Tensor, mask: torch.Tensor, c: TextConfig,
               pfx: str = "bert.", masks: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """BERT / DistilBERT encoder -> last_hidden_state [B,S,D].
    ``masks`` injects dropout masks (values 0 or 1/(1-p)) at BERT's four dropout sites -- "emb" [B,S,D],
    "attn{l}" [B,H,S,S], "so{l}" (BertSelfOutput) and "ffn{l}" (BertOutput) [B,S,D] -- so that a
    training-mode run can be compared element for element; None = p 0 everywhere."""
    masks = masks or {}
    B, S = ids.shape
    pos = torch.arange(S, device=ids.device)
    # transformers builds word_embeddings with padding_idx=pad_token_id: forward is a plain
    # row gather, backward leaves the PAD row's gradient at zero.
    x = F.embedding(ids, p[pfx + "embeddings.word_embeddings.weight"], padding_idx=c.pad_token_id) \
        + p[pfx + "embeddings.position_embeddings.weight"][pos][None]
    if c.type_vocab > 0:
        x = x + p[pfx + "embeddings.token_type_embeddings.weight"][0][None, None]
    x = F.layer_norm(x, (c.hidden,), p[pfx + "embeddings.LayerNorm.weight"],
                     p[pfx + "embeddings.LayerNorm.bias"], c.ln_eps)
    if "emb" in masks:
        x = x * masks["emb"]
    # transformers' extended attention mask: (1 - mask) * finfo(dtype).min
    add_mask = (1.0 - mask.to(x.dtype))[:, None, None, :] * torch.finfo(x.dtype).min
    for i in range(c.layers):
        L = f"{pfx}encoder.layer.{i}."
        ctx = _mha(x,
                   p[L + "attention.self.query.weight"], p[L + "attention.self.query.bias"],
                   p[L + "attention.self.key.weight"], p[L + "attention.self.key.bias"],
                   p[L + "attention.self.value.weight"], p[L + "attention.self.value.bias"],
                   c.heads, add_mask, masks.get(f"attn{i}"))
        a = F.linear(ctx, p[L + "attention.output.dense.weight"], p[L + "attention.output.dense.bias"])
        if f"so{i}" in masks:
            a = a * masks[f"so{i}"]
        x = F.layer_norm(x + a, (c.hidden,), p[L + "attention.output.LayerNorm.weight"],
                         p[L + "attention.output.LayerNorm.bias"], c.ln_eps)
        h = F.gelu(F.linear(x, p[L + "intermediate.dense.weight"], p[L + "intermediate.dense.bias"]))
        o = F.linear(h, p[L + "output.dense.weight"], p[L + "output.dense.bias"])
        if f"ffn{i}" in masks:
            o = o * masks[f"ffn{i}"]
        x = F.layer_norm(x + o, (c.hidden,), p[L + "output.LayerNorm.weight"],
                         p[L + "output.LayerNorm.bias"], c.ln_eps)
    return x


def patchify(image: torch.Tensor, patch: int) -> torch.Tensor:
    """[B,C,H,W] -> [B, (H/p)*(W/p), C*p*p]; patch order row-major over (h,w),
    feature order (c, i, j) == the flattened conv weight [D, C, p, p]."""
    B, C, H, W = image.shape
    gh, gw = H // patch, W // patch
    x = image.view(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, gh * gw, C * patch * patch)


def image_tower(p: Params, image: torch.Tensor, c: ImageConfig,
                pfx: str = "image_model.") -> torch.Tensor:
    """ViT encoder -> final-LayerNorm'd sequence [B,N+1,D]."""
    B = image.shape[0]
    D = c.hidden
    w = p[pfx + "embeddings.patch_embeddings.projection.weight"].reshape(D, -1)
    x = F.linear(patchify(image, c.patch), w, p.get(pfx + "embeddings.patch_embeddings.projection.bias"))
    x = torch.cat([p[pfx + "embeddings.cls_token"].expand(B, -1, -1), x], dim=1)
    x = x + p[pfx + "embeddings.position_embeddings"]
    if c.pre_ln:      # CLIPVisionTransformer.pre_layrnorm
        x = F.layer_norm(x, (D,), p[pfx + "pre_layernorm.weight"], p[pfx + "pre_layernorm.bias"], c.ln_eps)
    act = (lambda z: z * torch.sigmoid(1.702 * z)) if c.act == "quick_gelu" else F.gelu
    for i in range(c.layers):
        L = f"{pfx}encoder.layer.{i}."
        h = F.layer_norm(x, (D,), p[L + "layernorm_before.weight"], p[L + "layernorm_before.bias"], c.ln_eps)
        ctx = _mha(h,
                   p[L + "attention.attention.query.weight"], p[L + "attention.attention.query.bias"],
                   p[L + "attention.attention.key.weight"], p[L + "attention.attention.key.bias"],
                   p[L + "attention.attention.value.weight"], p[L + "attention.attention.value.bias"],
                   c.heads, None)
        x = x + F.linear(ctx, p[L + "attention.output.dense.weight"], p[L + "attention.output.dense.bias"])
        h = F.layer_norm(x, (D,), p[L + "layernorm_after.weight"], p[L + "layernorm_after.bias"], c.ln_eps)
        h = act(F.linear(h, p[L + "intermediate.dense.weight"], p[L + "intermediate.dense.bias"]))
        x = x + F.linear(h, p[L + "output.dense.weight"], p[L + "output.dense.bias"])
    return F.layer_norm(x, (D,), p[pfx + "layernorm.weight"], p[pfx + "layernorm.bias"], c.ln_eps)


def pool_text(hidden: torch.Tensor, pool: str) -> torch.Tensor:
    if pool == "cls":
        return hidden[:, 0]
    if pool == "last":
        return hidden[:, -1]
    raise ValueError(f"Unsupported pooling type: {pool}")


def forward(p: Params, text: torch.Tensor, image: torch.Tensor, mask: torch.Tensor,
            cfg: OracleConfig, masks: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """MultimodalClassifier.forward(text, image, mask) -> logits [B,num_classes]
    (...task2C.txt:172-197; argument order text, image, mask).  ``masks``: see text_tower, plus "head" [B,D]
    for ``bert_drop`` (...task2C.txt:160,178)."""
    t = pool_text(text_tower(p, text, mask, cfg.text, masks=masks), cfg.pool)
    if masks and "head" in masks:
        t = t * masks["head"]
    t = F.linear(t, p["bert_fc.weight"], p["bert_fc.bias"])
    v = image_tower(p, image, cfg.image)[:, 0]
    v = F.linear(v, p["image_fc.weight"], p["image_fc.bias"])
    f = F.linear(torch.cat((t, v), dim=1), p["fusion_fc.weight"], p["fusion_fc.bias"])
    return F.linear(f, p["output_fc.weight"], p["output_fc.bias"])


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """nn.CrossEntropyLoss() default: mean over the batch (...task2C.txt:248)."""
    lse = torch.logsumexp(logits, dim=1)
    return (lse - logits.gather(1, labels[:, None]).squeeze(1)).mean()


def sigmoid_focal_loss(inputs: torch.Tensor, targets: torch.Tensor, alpha: float = 0.25, gamma: float = 2.0) -> torch.Tensor:
    """torchvision.ops.sigmoid_focal_loss(..., reduction="mean") restated (torchvision 0.17.2 is not installed;
    call site Multimodal_example_task2C.py:167,711):  ce * (1 - p_t)^gamma * alpha_t."""
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean()


def loss_and_grads(p: Params, text, image, mask, labels, cfg: OracleConfig, masks=None):
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    logits = forward(leaves, text, image, mask, cfg, masks=masks)
    loss = cross_entropy(logits, labels)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return logits.detach(), loss.detach(), grads


# --------------------------------------------------------------------------
# optimizer
# --------------------------------------------------------------------------

@dataclass
class AdamState:
    step: int = 0
    m: Params = field(default_factory=dict)
    v: Params = field(default_factory=dict)


def global_grad_norm(grads: Params) -> torch.Tensor:
    return torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()


def adam_step(p: Params, grads: Params, st: AdamState, lr: float = 2e-5,
              betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
              decoupled: bool = False, max_grad_norm: Optional[float] = None, lr_of=None) -> Params:
    """torch.optim.Adam / AdamW single-tensor algorithm, dense over every parameter
    (...task2C.txt:249,217).  ``max_grad_norm`` applies clip_grad_norm_ first
    (coef = max_norm / (norm + 1e-6), clamped to 1)."""
    b1, b2 = betas
    scale = 1.0
    if max_grad_norm is not None:
        scale = min(1.0, float(max_grad_norm / (global_grad_norm(grads) + 1e-6)))
    st.step += 1
    t = st.step
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    out: Params = {}
    for k, w in p.items():
        g = grads[k] * scale
        w = w.clone()
        if weight_decay != 0.0:
            if decoupled:
                w = w * (1.0 - lr * weight_decay)
            else:
                g = g + weight_decay * w
        m = st.m.get(k)
        v = st.v.get(k)
        if m is None:
            m = torch.zeros_like(w)
            v = torch.zeros_like(w)
        m = m * b1 + (1.0 - b1) * g
        v = v * b2 + (1.0 - b2) * g * g
        st.m[k], st.v[k] = m, v
        denom = v.sqrt() / math.sqrt(bc2) + eps
        lr_k = lr_of(k) if lr_of is not None else lr      # parameter groups (Multimodal_example_task2C.py:645-664)
        out[k] = w - (lr_k / bc1) * (m / denom)
    return out


def train_step(p: Params, st: AdamState, text, image, mask, labels, cfg: OracleConfig, **adam_kw):
    logits, loss, grads = loss_and_grads(p, text, image, mask, labels, cfg)
    return adam_step(p, grads, st, **adam_kw), logits, loss, grads


# --------------------------------------------------------------------------
# synthetic batch
# --------------------------------------------------------------------------

def synthetic_batch(cfg: OracleConfig, batch: int, seq_len: int, seed: int = 1234,
                    all_ones_mask: bool = False):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    ic = cfg.image
    image = torch.randn((batch, ic.channels, ic.image_size, ic.image_size), generator=g)
    V = cfg.text.vocab_size
    text = torch.randint(5, V, (batch, seq_len), generator=g, dtype=torch.int64)
    lo = min(8, seq_len)
    lens = torch.randint(lo, seq_len + 1, (batch,), generator=g)
    if all_ones_mask:
        lens = torch.full((batch,), seq_len)
    ar = torch.arange(seq_len)[None]
    mask = (ar < lens[:, None]).to(torch.int64)
    text = text * mask                      # PAD id 0 on padded positions
    text[:, 0] = 2                          # fixed [CLS]-like id
    labels = (torch.rand((batch,), generator=g) < 0.28).to(torch.int64)
    return text, image, mask, labels
