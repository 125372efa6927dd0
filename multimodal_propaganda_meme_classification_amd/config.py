"""Model configuration + flat parameter layout for the dual-encoder.

Parameter names follow transformers 4.39.2 (the reference's pin: BertModel under ``bert.``,
ViTModel under ``image_model.``) plus the organizers' head names ``bert_fc`` / ``fusion_fc`` /
``output_fc`` (example_scripts/Multimodal_example_task2C.txt:161-170; the image projection is
``image_fc`` because the image tower here is a ViT, ``resnet_fc`` is accepted as an alias).

All parameters live in ONE flat fp32 buffer (gradients and Adam moments mirror it), ordered so
that (a) q/k/v weights and biases of a layer are adjacent -> the fused [3D, D] QKV GEMM operand is
a plain slice, (b) the big GEMM matrices come first, in backward-completion order (last layer
first, both towers interleaved) -> gradient buckets for the RCCL all-reduce close early and the
bf16 shadow copy of the GEMM operands is one prefix ``[0, n_shadow)`` of the buffer.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass, field
from typing import Dict, List, Tuple


@dataclass
class TextConfig:
    vocab_size: int = 64000
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    max_position: int = 512
    type_vocab: int = 2
    pad_token_id: int = 0
    ln_eps: float = 1e-12
    hidden_dropout: float = 0.0        # BertConfig.hidden_dropout_prob (0.1 in the reference's checkpoints)
    attention_dropout: float = 0.0     # BertConfig.attention_probs_dropout_prob (0.1)


@dataclass
class ImageConfig:
    image_size: int = 224
    patch: int = 16
    channels: int = 3
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    ln_eps: float = 1e-6
    # CLIP vision tower (BASELINE.json configs[4]: openai/clip-vit-large-patch14-336; transformers CLIPVisionModel):
    act: str = "gelu"            # "gelu" (erf; timm / HF ViT) | "quick_gelu" (x * sigmoid(1.702 x))
    pre_ln: bool = False         # CLIP's pre_layrnorm: a LayerNorm on the embeddings, in front of the first block
    patch_bias: bool = True      # CLIP's patch conv has no bias

    @property
    def patch_dim(self) -> int:
        return self.channels * self.patch * self.patch

    @property
    def patch_dim_padded(self) -> int:
        """Contraction length of the patch-projection GEMM: C*p*p rounded up to the GEMM's K tile (588 -> 640 at p = 14)."""
        return (self.patch_dim + 63) // 64 * 64

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch) ** 2

    @property
    def n_tokens(self) -> int:
        return self.n_patches + 1


@dataclass
class ModelConfig:
    text: TextConfig = field(default_factory=TextConfig)
    image: ImageConfig = field(default_factory=ImageConfig)
    proj: int = 512
    num_classes: int = 2
    pool: str = "cls"   # "cls" (Multimodal_example_task2C.py:359) | "last" (...task2C.txt:178)
    # 16-bit storage / MFMA operand type of the towers: "bf16" (libmemehip.so) or "fp16"
    # (libmemehip_f16.so: 11-bit significand, needs the power-of-two gradient-stream scale below,
    # the static counterpart of the GradScaler in Multimodal_example_task2C.py:60-64,712-717)
    compute_dtype: str = "bf16"
    grad_stream_scale: float = 0.0      # 0 = automatic: 1 for bf16, 8192 for fp16
    pack_text: bool = True              # padding-free text tower: positions with attention_mask == 0 are never computed
                                        # (they influence neither logits nor gradients); False = dense [B,S] rows
    image_fc_name: str = "image_fc"     # state_dict name of the image projection: "resnet_fc" round-trips checkpoints of the
                                        # organizers' module (Multimodal_example_task2C.txt:165); both are accepted on load
    head_dropout: float = 0.0           # nn.Dropout(0.3) on the pooled text features (...task2C.txt:160); the
                                        # parity / measurement plan runs every dropout at p = 0

    def with_reference_dropout(self) -> "ModelConfig":
        """The reference's training-mode dropout: 0.1 hidden / 0.1 attention (BERT), 0.3 head; timm ViT has none."""
        self.text.hidden_dropout, self.text.attention_dropout, self.head_dropout = 0.1, 0.1, 0.3
        return self

    @staticmethod
    def from_dict(d: dict) -> "ModelConfig":
        return ModelConfig(text=TextConfig(**d["text"]), image=ImageConfig(**d["image"]), proj=d["proj"],
                           num_classes=d["num_classes"], pool=d["pool"], compute_dtype=d.get("compute_dtype", "bf16"),
                           grad_stream_scale=d.get("grad_stream_scale", 0.0), head_dropout=d.get("head_dropout", 0.0),
                           pack_text=d.get("pack_text", True), image_fc_name=d.get("image_fc_name", "image_fc"))

    @property
    def stream_scale(self) -> float:
        if self.grad_stream_scale:
            return float(self.grad_stream_scale)
        return 8192.0 if self.compute_dtype == "fp16" else 1.0

    def to_dict(self) -> dict:
        return asdict(self)

    def validate(self):
        if self.pool not in ("cls", "last"):
            raise ValueError(f"Unsupported pooling type: {self.pool}")
        if self.image_fc_name not in ("image_fc", "resnet_fc"):
            raise ValueError(f"image_fc_name must be 'image_fc' or 'resnet_fc', got {self.image_fc_name!r}")
        if self.compute_dtype not in ("bf16", "fp16"):
            raise ValueError(f"compute_dtype must be 'bf16' or 'fp16', got {self.compute_dtype!r}")
        for pr in (self.text.hidden_dropout, self.text.attention_dropout, self.head_dropout):
            if not 0.0 <= pr < 1.0:
                raise ValueError(f"dropout probability has to be in [0, 1), got {pr}")
        for nm, c in (("text", self.text), ("image", self.image)):
            if c.hidden % 128 or c.intermediate % 128:
                raise ValueError(f"{nm}: hidden and intermediate sizes must be multiples of 128 (GEMM tile)")
            if c.hidden != c.heads * 64:
                raise ValueError(f"{nm}: head dim must be 64 (hidden == heads * 64)")
        if self.image.image_size % self.image.patch:
            raise ValueError("image: patch must divide image_size")
        if self.image.patch_dim % 2:
            raise ValueError("image: C*patch*patch must be even (two 16-bit elements per 32-bit word)")
        if self.image.act not in ("gelu", "quick_gelu"):
            raise ValueError(f"image: unsupported activation {self.image.act!r}")


@dataclass
class ParamSpec:
    name: str
    shape: Tuple[int, ...]
    offset: int
    numel: int
    init: str           # "normal" | "ones" | "zeros" | "linear_w:<fan_in>" | "linear_b:<fan_in>"


def _numel(shape) -> int:
    n = 1
    for s in shape:
        n *= s
    return n


class Layout:
    """Flat order of every parameter.  ``spec[name]`` -> ParamSpec."""

    def __init__(self, cfg: ModelConfig):
        cfg.validate()
        self.cfg = cfg
        self.specs: List[ParamSpec] = []
        self.spec: Dict[str, ParamSpec] = {}
        self._off = 0
        self.layer_ranges: List[Tuple[int, int, int]] = []   # (layer index, start, end) of region A, in order
        t, v = cfg.text, cfg.image
        nl = max(t.layers, v.layers)
        # ---- region A: GEMM matrices (bf16-shadowed), last layer first -------------------------------
        for l in range(nl - 1, -1, -1):
            start = self._off
            if l < t.layers:
                L = f"bert.encoder.layer.{l}."
                for n in ("query", "key", "value"):
                    self._add(L + f"attention.self.{n}.weight", (t.hidden, t.hidden), "normal")
                self._add(L + "attention.output.dense.weight", (t.hidden, t.hidden), "normal")
                self._add(L + "intermediate.dense.weight", (t.intermediate, t.hidden), "normal")
                self._add(L + "output.dense.weight", (t.hidden, t.intermediate), "normal")
            if l < v.layers:
                L = f"image_model.encoder.layer.{l}."
                for n in ("query", "key", "value"):
                    self._add(L + f"attention.attention.{n}.weight", (v.hidden, v.hidden), "normal")
                self._add(L + "attention.output.dense.weight", (v.hidden, v.hidden), "normal")
                self._add(L + "intermediate.dense.weight", (v.intermediate, v.hidden), "normal")
                self._add(L + "output.dense.weight", (v.hidden, v.intermediate), "normal")
            self.layer_ranges.append((l, start, self._off))
        self._add("image_model.embeddings.patch_embeddings.projection.weight",
                  (v.hidden, v.channels, v.patch, v.patch), "normal")
        self.n_shadow = self._off
        # ---- region B: fp32-only ------------------------------------------------------------------------
        P = cfg.proj
        self._add("bert_fc.weight", (P, t.hidden), f"linear_w:{t.hidden}")
        self._add("bert_fc.bias", (P,), f"linear_b:{t.hidden}")
        self._add("image_fc.weight", (P, v.hidden), f"linear_w:{v.hidden}")
        self._add("image_fc.bias", (P,), f"linear_b:{v.hidden}")
        self._add("fusion_fc.weight", (P, 2 * P), f"linear_w:{2 * P}")
        self._add("fusion_fc.bias", (P,), f"linear_b:{2 * P}")
        self._add("output_fc.weight", (cfg.num_classes, P), f"linear_w:{P}")
        self._add("output_fc.bias", (cfg.num_classes,), f"linear_b:{P}")
        for l in range(t.layers - 1, -1, -1):
            L = f"bert.encoder.layer.{l}."
            for n in ("query", "key", "value"):
                self._add(L + f"attention.self.{n}.bias", (t.hidden,), "zeros")
            self._add(L + "attention.output.dense.bias", (t.hidden,), "zeros")
            self._add(L + "attention.output.LayerNorm.weight", (t.hidden,), "ones")
            self._add(L + "attention.output.LayerNorm.bias", (t.hidden,), "zeros")
            self._add(L + "intermediate.dense.bias", (t.intermediate,), "zeros")
            self._add(L + "output.dense.bias", (t.hidden,), "zeros")
            self._add(L + "output.LayerNorm.weight", (t.hidden,), "ones")
            self._add(L + "output.LayerNorm.bias", (t.hidden,), "zeros")
        for l in range(v.layers - 1, -1, -1):
            L = f"image_model.encoder.layer.{l}."
            self._add(L + "layernorm_before.weight", (v.hidden,), "ones")
            self._add(L + "layernorm_before.bias", (v.hidden,), "zeros")
            for n in ("query", "key", "value"):
                self._add(L + f"attention.attention.{n}.bias", (v.hidden,), "zeros")
            self._add(L + "attention.output.dense.bias", (v.hidden,), "zeros")
            self._add(L + "layernorm_after.weight", (v.hidden,), "ones")
            self._add(L + "layernorm_after.bias", (v.hidden,), "zeros")
            self._add(L + "intermediate.dense.bias", (v.intermediate,), "zeros")
            self._add(L + "output.dense.bias", (v.hidden,), "zeros")
        self._add("image_model.layernorm.weight", (v.hidden,), "ones")
        self._add("image_model.layernorm.bias", (v.hidden,), "zeros")
        if v.pre_ln:
            self._add("image_model.pre_layernorm.weight", (v.hidden,), "ones")
            self._add("image_model.pre_layernorm.bias", (v.hidden,), "zeros")
        if v.patch_bias:
            self._add("image_model.embeddings.patch_embeddings.projection.bias", (v.hidden,), "zeros")
        self._add("image_model.embeddings.cls_token", (1, 1, v.hidden), "normal")
        self._add("image_model.embeddings.position_embeddings", (1, v.n_tokens, v.hidden), "normal")
        self._add("bert.embeddings.LayerNorm.weight", (t.hidden,), "ones")
        self._add("bert.embeddings.LayerNorm.bias", (t.hidden,), "zeros")
        if t.type_vocab > 0:
            self._add("bert.embeddings.token_type_embeddings.weight", (t.type_vocab, t.hidden), "normal")
        self._add("bert.embeddings.position_embeddings.weight", (t.max_position, t.hidden), "normal")
        self._add("bert.embeddings.word_embeddings.weight", (t.vocab_size, t.hidden), "normal")
        self.n_total = self._off

    def _add(self, name: str, shape, init: str):
        n = _numel(shape)
        assert n % 4 == 0 or name.startswith("output_fc"), name
        s = ParamSpec(name, tuple(shape), self._off, n, init)
        self.specs.append(s)
        self.spec[name] = s
        self._off += (n + 3) // 4 * 4          # keep every tensor 16-byte aligned

    # canonical (state_dict) order = the reference modules' registration order
    def state_dict_order(self) -> List[str]:
        t, v = self.cfg.text, self.cfg.image
        names = ["bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight"]
        if t.type_vocab > 0:
            names.append("bert.embeddings.token_type_embeddings.weight")
        names += ["bert.embeddings.LayerNorm.weight", "bert.embeddings.LayerNorm.bias"]
        for l in range(t.layers):
            L = f"bert.encoder.layer.{l}."
            for n in ("query", "key", "value"):
                names += [L + f"attention.self.{n}.weight", L + f"attention.self.{n}.bias"]
            names += [L + "attention.output.dense.weight", L + "attention.output.dense.bias",
                      L + "attention.output.LayerNorm.weight", L + "attention.output.LayerNorm.bias",
                      L + "intermediate.dense.weight", L + "intermediate.dense.bias",
                      L + "output.dense.weight", L + "output.dense.bias",
                      L + "output.LayerNorm.weight", L + "output.LayerNorm.bias"]
        names += ["bert_fc.weight", "bert_fc.bias"]
        names += ["image_model.embeddings.cls_token", "image_model.embeddings.position_embeddings",
                  "image_model.embeddings.patch_embeddings.projection.weight"]
        if v.patch_bias:
            names.append("image_model.embeddings.patch_embeddings.projection.bias")
        if v.pre_ln:
            names += ["image_model.pre_layernorm.weight", "image_model.pre_layernorm.bias"]
        for l in range(v.layers):
            L = f"image_model.encoder.layer.{l}."
            names += [L + "layernorm_before.weight", L + "layernorm_before.bias"]
            for n in ("query", "key", "value"):
                names += [L + f"attention.attention.{n}.weight", L + f"attention.attention.{n}.bias"]
            names += [L + "attention.output.dense.weight", L + "attention.output.dense.bias",
                      L + "layernorm_after.weight", L + "layernorm_after.bias",
                      L + "intermediate.dense.weight", L + "intermediate.dense.bias",
                      L + "output.dense.weight", L + "output.dense.bias"]
        names += ["image_model.layernorm.weight", "image_model.layernorm.bias",
                  "image_fc.weight", "image_fc.bias", "fusion_fc.weight", "fusion_fc.bias",
                  "output_fc.weight", "output_fc.bias"]
        assert sorted(names) == sorted(self.spec), "layout / state_dict order drift"
        return names
