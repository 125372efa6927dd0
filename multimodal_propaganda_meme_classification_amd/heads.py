"""The reference's other module surfaces over the HIP towers.

* ``TextClassifier`` -- ``LLMWithClassificationHead(model_name, pooling_type, num_classes)`` of
  example_scripts/DistilBERT_example_task2A.py:140-210: the HF-Trainer protocol (``model(**batch)`` with ``labels``
  returns ``(loss, logits)``, without them ``logits``), poolings cls / max / mean / attention / cnn, ``ValueError`` for
  anything else (:173).  ``head="distilbert"`` is the stock ``DistilBertForSequenceClassification`` head of the
  notebook variant (pre_classifier + ReLU + classifier; 135 326 210 parameters with DistilBERT-multilingual,
  DistilBERT_example_task2A.ipynb:4301).
* ``TrainerModel`` -- the two-tower ``MultimodalClassifier`` behind the same protocol with the batch keys of
  ResNet_example_task2B.py:206-210 (``pixel_values``, ``labels``) plus ``input_ids`` / ``attention_mask``.
* ``KevinMultimodalClassifier`` -- ``MultimodalClassifier(fusion_method)`` of Multimodal_example_task2C.py:587-685: the
  five-argument ``forward(text, image, mask, caption_text, caption_text_mask) -> [B]``, ``get_params(lr)`` with the
  reference's grouping (:645-664), ``ConcatAttention3`` (:476-499), ``Linear + BatchNorm1d + ReLU`` projections
  (:599-612), ``Linear(512, 1) + BatchNorm1d(1)`` (:641-643).

The encoders run in libmemehip (model.MultimodalClassifier / TextEncoder); poolings, projections and the fusion run in
the HIP kernels of ops.py (pool_* / linear_bn_act_* / softmax_gate_*).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import _lib, fused
from .config import ImageConfig, ModelConfig, TextConfig
from .model import CrossEntropyLoss, MultimodalClassifier, TextEncoder, flatten_parameters

POOLINGS = ("cls", "max", "mean", "attention", "cnn")


class _Composite(nn.Module):
    """A module whose children include HIP towers (TextEncoder / MultimodalClassifier keep their parameters in a flat buffer
    behind reference-style key names): load_state_dict hands every child ITS slice of the checkpoint, so the tower's own
    loader (key translation, 16-bit shadow invalidation) runs instead of nn.Module's recursion into the tower's internals."""

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        missing, unexpected = [], []
        seen = set()
        for name, child in self._modules.items():
            if child is None:
                continue
            pfx = name + "."
            sub = {k[len(pfx):]: v for k, v in state_dict.items() if k.startswith(pfx)}
            seen.update(pfx + k for k in sub)
            if not sub and not list(child.state_dict().keys()):
                continue
            res = child.load_state_dict(sub, strict=False)
            missing += [pfx + k for k in res.missing_keys]
            unexpected += [pfx + k for k in res.unexpected_keys]
        own = {n for n, _ in self.named_parameters(recurse=False)} | {n for n, _ in self.named_buffers(recurse=False)}
        with torch.no_grad():
            for n in own:
                if n in state_dict:
                    getattr(self, n).copy_(state_dict[n])
                    seen.add(n)
                else:
                    missing.append(n)
        unexpected += [k for k in state_dict if k not in seen]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:4]} unexpected {unexpected[:4]}")
        return nn.modules.module._IncompatibleKeys(missing, unexpected)


class SequencePooling(nn.Module):
    """The pooling branches of ``LLMWithClassificationHead`` (Multimodal_example_task2C.py:339-392,
    DistilBERT_example_task2A.py:162-210) over a last hidden state [B, S, D] (f32, device) and the attention mask."""

    def __init__(self, pooling_type: str, hidden_size: int = 768, attention_hidden_size: int = 512, cnn_kernel_size: int = 3):
        super().__init__()
        self.pooling_type = pooling_type
        if pooling_type == "attention":
            self.attention = nn.Sequential(nn.Linear(hidden_size, attention_hidden_size), nn.Tanh(),
                                           nn.Linear(attention_hidden_size, 1))
        elif pooling_type == "cnn":
            self.conv1d = nn.Conv1d(hidden_size, hidden_size, kernel_size=cnn_kernel_size, padding=cnn_kernel_size // 2)

    def forward(self, hidden: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        kind = self.pooling_type
        if kind == "cls":
            return hidden[:, 0]
        if kind == "max":
            return fused.max_pool(hidden)
        if kind == "mean":
            return fused.masked_mean_pool(hidden, attention_mask)
        if kind == "attention":
            a = self.attention
            return fused.attention_pool(hidden, attention_mask, a[0].weight, a[0].bias, a[2].weight, a[2].bias)
        if kind == "cnn":
            return fused.conv1d_relu_max_pool(hidden, self.conv1d.weight, self.conv1d.bias)
        raise ValueError(f"Unsupported pooling type: {kind}")


class TextClassifier(_Composite):
    """``LLMWithClassificationHead`` with the HF-Trainer protocol (DistilBERT_example_task2A.py:140-183).

    ``text`` is the encoder's shape (DistilBERT-multilingual: ``TextConfig(vocab_size=119547, layers=6, type_vocab=0)``).
    ``forward(input_ids, attention_mask, labels=None)``: ``(loss, logits)`` when ``labels`` is given, else ``logits``."""

    def __init__(self, text: TextConfig, pooling_type: str = "attention", num_classes: int = 2, hidden_size: Optional[int] = None,
                 attention_hidden_size: int = 512, cnn_kernel_size: int = 3, head: str = "linear", compute_dtype: str = "bf16",
                 seed: int = 0):
        super().__init__()
        if pooling_type not in POOLINGS:
            raise ValueError(f"Unsupported pooling type: {pooling_type}")
        if head not in ("linear", "distilbert"):
            raise ValueError(f"head must be 'linear' or 'distilbert', got {head!r}")
        D = hidden_size or text.hidden
        self.pooling_type, self.hidden_size, self.num_classes, self.head = pooling_type, D, num_classes, head
        self.model = TextEncoder(text, pool="cls", compute_dtype=compute_dtype, seed=seed)
        self.pool = SequencePooling(pooling_type, D, attention_hidden_size, cnn_kernel_size)
        if head == "distilbert":        # DistilBertForSequenceClassification: pooled -> pre_classifier -> ReLU -> (dropout) -> classifier
            self.pre_classifier = nn.Linear(D, D)
            self.classifier = nn.Linear(D, num_classes)
        else:
            self.output_layer = nn.Linear(D, num_classes)
        self.loss_fct = CrossEntropyLoss()

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        return self

    def forward(self, input_ids=None, attention_mask=None, labels=None, **unused):
        if input_ids is None or attention_mask is None:
            raise ValueError("TextClassifier.forward needs input_ids and attention_mask")
        if self.pooling_type == "cls":
            pooled = self.model(input_ids, attention_mask)
        else:
            pooled = self.pool(self.model.hidden_states(input_ids, attention_mask), attention_mask)
        if self.head == "distilbert":
            hid = fused.linear(pooled, self.pre_classifier.weight, self.pre_classifier.bias, act="relu")
            logits = fused.linear(hid, self.classifier.weight, self.classifier.bias)
        else:
            logits = fused.linear(pooled, self.output_layer.weight, self.output_layer.bias)
        if labels is not None:
            loss = self.loss_fct(logits.view(-1, self.num_classes), labels.view(-1))
            return loss, logits
        return logits

    def n_parameters(self) -> int:
        """Trainable parameters of the classifier the reference counts (DistilBERT_example_task2A.ipynb:4301): the text
        encoder + head, without the inert stub image side of the lockstep launch plan."""
        enc = sum(v.numel() for v in self.model.state_dict().values())
        own = sum(p.numel() for n, p in self.named_parameters() if not n.startswith("model."))
        return enc + own


class TrainerModel(_Composite):
    """The Subtask-2C two-tower classifier behind the HF-Trainer protocol: ``model(**batch)`` with the collator's keys
    ``input_ids, attention_mask, pixel_values[, labels]`` (ResNet_example_task2B.py:206-210 for the image side,
    DistilBERT_example_task2A.py:159 for the text side) -> ``(loss, logits)`` / ``logits``."""

    def __init__(self, model: MultimodalClassifier):
        super().__init__()
        self.model = model
        self.loss_fct = CrossEntropyLoss()

    def forward(self, input_ids=None, attention_mask=None, pixel_values=None, labels=None, **unused):
        if input_ids is None or attention_mask is None or pixel_values is None:
            raise ValueError("TrainerModel.forward needs input_ids, attention_mask and pixel_values")
        logits = self.model(input_ids, pixel_values, attention_mask)
        if labels is not None:
            return self.loss_fct(logits, labels.view(-1)), logits
        return logits


class OrganizersMultimodalClassifier(_Composite):
    """The organizers' Subtask-2C model exactly as written (example_scripts/Multimodal_example_task2C.txt:152-197):
    ``bert`` (DistilBERT-multilingual via AutoModel) -> ``[:, -1, :]`` -> ``bert_drop`` -> ``bert_fc(768, 512)``;
    ``resnet`` (torchvision resnet50, its 1000 logits) -> ``resnet_fc(1000, 512)``; ``cat`` -> ``fusion_fc(1024, 512)`` ->
    ``output_fc(512, num_classes)``; ``forward(text, image, mask)``.  Same attribute names and state_dict keys
    (``bert.transformer.layer.0.attention.q_lin.weight``, ``resnet.layer1.0.conv1.weight``, ``resnet_fc.bias`` ...), so a
    checkpoint of the reference module loads.  Towers: TextEncoder (DistilBERT naming) and ResNet50 on the HIP kernels; the
    four Linear layers run in the exact-f32 MFMA GEMM."""

    def __init__(self, num_classes: int = 2, text: Optional[TextConfig] = None, compute_dtype: str = "fp16", resnet_layers=(3, 4, 6, 3),
                 seed: int = 0):
        super().__init__()
        from .resnet import ResNet50
        # default = distilbert-base-multilingual-cased as its checkpoint configures it (dropout 0.1, attention_dropout 0.1)
        tc = text or TextConfig(vocab_size=119547, hidden=768, layers=6, heads=12, intermediate=3072, max_position=512, type_vocab=0,
                                hidden_dropout=0.1, attention_dropout=0.1)
        self.bert = TextEncoder(tc, pool="last", compute_dtype=compute_dtype, seed=seed, naming="distilbert")
        self.bert_drop = nn.Dropout(0.3)
        self.bert_fc = nn.Linear(tc.hidden, 512)
        self.resnet = ResNet50(num_classes=1000, compute_dtype=compute_dtype, layers=resnet_layers, seed=seed + 1)
        self.resnet_fc = nn.Linear(1000, 512)
        self.fusion_fc = nn.Linear(1024, 512)
        self.output_fc = nn.Linear(512, num_classes)

    def forward(self, text, image, mask):
        for t_ in (text, image, mask):
            if not t_.is_cuda:
                raise _lib.MemehipError("memehip runs on the HIP device only (no CPU fallback): move the batch with .to(device)")
        bert_output = self.bert_drop(self.bert(text, mask))             # the LAST position, as the reference takes it
        bert_output = fused.linear(bert_output, self.bert_fc.weight, self.bert_fc.bias)
        resnet_output = self.resnet(image)
        resnet_output = fused.linear(resnet_output, self.resnet_fc.weight, self.resnet_fc.bias)
        features = torch.cat((bert_output, resnet_output), dim=1)
        features = fused.linear(features, self.fusion_fc.weight, self.fusion_fc.bias)
        return fused.linear(features, self.output_fc.weight, self.output_fc.bias)


class LinearBNReLU(nn.Sequential):
    """``nn.Sequential(nn.Linear(i, o), nn.BatchNorm1d(o), nn.ReLU())`` (Multimodal_example_task2C.py:599-601) with the
    reference's state_dict keys (``0.weight``, ``1.running_mean`` ...), run as ONE fused HIP op."""

    def __init__(self, in_features: int, out_features: int, relu: bool = True):
        mods = [nn.Linear(in_features, out_features), nn.BatchNorm1d(out_features)]
        if relu:
            mods.append(nn.ReLU())
        super().__init__(*mods)
        self.relu = relu

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return fused.linear_bn_act(x, self[0], self[1], self.relu)


class FineTuneMLP(nn.Sequential):
    """``CustomDenseNet161.fine_tune`` (Multimodal_example_task2C.py:571-574): Linear -> ReLU -> Dropout(0.35) -> Linear, same
    state_dict keys (``0.*``, ``3.*``); the two Linear layers (the first with its ReLU fused) run in mh_gemm_f32."""

    def __init__(self, in_features: int, width: int):
        super().__init__(nn.Linear(in_features, width), nn.ReLU(inplace=True), nn.Dropout(p=0.35), nn.Linear(width, width))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = fused.linear(x, self[0].weight, self[0].bias, act="relu")
        return fused.linear(self[2](x), self[3].weight, self[3].bias)


class ConcatAttention3(nn.Module):
    """Multimodal_example_task2C.py:476-499: softmax-gated concat of the three towers' features + reduce."""

    def __init__(self, input_dim: int, attention_dim: int):
        super().__init__()
        self.attention_layer = nn.Sequential(nn.Linear(input_dim, input_dim), nn.BatchNorm1d(input_dim), nn.ReLU(),
                                             nn.Softmax(dim=1))
        self.reduce = LinearBNReLU(input_dim, attention_dim)

    def forward(self, text_features, image_features, caption_features):
        concatenated = torch.cat((text_features, image_features, caption_features), dim=1)
        gate_in = fused.linear_bn_act(concatenated, self.attention_layer[0], self.attention_layer[1], True)
        attended = fused.softmax_gate(gate_in, concatenated)        # softmax(gate_in, dim=1) * concatenated
        return self.reduce(attended)


class MCA3(nn.Module):
    """Multimodal_example_task2C.py:423-448 (``fusion_method="mca"``): additive attention of every image row over the text rows
    of the batch (the reference un-squeezes only the image features, so its tanh score broadcasts to [B, B, units]), softmax
    over dim 1, the attended text and caption features concatenated and reduced.  Same submodule names as the reference."""

    def __init__(self, units: int):
        super().__init__()
        self.W1, self.W2, self.W3 = nn.Linear(units, units), nn.Linear(units, units), nn.Linear(units, units)
        self.V = nn.Linear(units, 1)
        self.reduce = nn.Linear(2 * units, units)

    def forward(self, text_features, image_features, caption_features):
        if text_features.dim() != 2:
            raise ValueError("MCA3 on the HIP path takes [B, units] features (what the reference's forward feeds it)")
        pa = fused.linear(text_features, self.W1.weight, self.W1.bias)
        pi = fused.linear(image_features, self.W2.weight, self.W2.bias)
        pc = fused.linear(caption_features, self.W3.weight, self.W3.bias)
        ctx = fused.mca3_attention(pa, pc, pi, self.V.weight, self.V.bias, text_features, caption_features)
        return fused.linear(ctx, self.reduce.weight, self.reduce.bias)


class KevinMultimodalClassifier(_Composite):
    """``MultimodalClassifier(fusion_method)`` of Multimodal_example_task2C.py:587-685 on the HIP towers.

    text tower + image tower = one lockstep ``MultimodalClassifier`` (``towers``; the reference's ``text_model`` and
    ``image_model.image_model``), caption tower = ``TextEncoder`` (``caption_text_model``).  The image tower's
    ``fine_tune`` MLP of ``CustomDenseNet161`` (:571-574) takes the tower's feature width instead of the hard-coded 512
    (the reference shape-errors for anything but ResNet-18/34)."""

    def __init__(self, fusion_method: str = "concatenation", text: Optional[TextConfig] = None,
                 image: Optional[ImageConfig] = None, caption: Optional[TextConfig] = None, proj: int = 512,
                 compute_dtype: str = "bf16", seed: int = 0, grad_stream_scale: float = 0.0):
        super().__init__()
        # the reference lists four methods (:86); "cross_modal" and "self_attention" build two-input modules that its own
        # three-input forward (:677) cannot call (TypeError on the first batch), so only the two that run are offered
        if fusion_method not in ("concatenation", "mca"):
            raise ValueError(f"Unsupported fusion method: {fusion_method}")
        # defaults = the checkpoints' own dropout (0.1 / 0.1); explicit configs say what they want
        tc, ic = text or TextConfig(hidden_dropout=0.1, attention_dropout=0.1), image or ImageConfig()
        cc = caption or TextConfig(vocab_size=30522, hidden_dropout=0.1, attention_dropout=0.1)
        self.fusion_method = fusion_method
        self.towers = MultimodalClassifier.from_config(ModelConfig(text=tc, image=ic, compute_dtype=compute_dtype,
                                                                   grad_stream_scale=grad_stream_scale), seed=seed)
        self.caption_text_model = TextEncoder(cc, pool="cls", compute_dtype=compute_dtype, seed=seed + 1, grad_stream_scale=grad_stream_scale)
        self.text_dropout, self.caption_text_dropout = nn.Dropout(0.3), nn.Dropout(0.3)
        self.text_fc = LinearBNReLU(tc.hidden, proj)
        self.caption_text_fc = LinearBNReLU(cc.hidden, proj)
        self.image_fine_tune = FineTuneMLP(ic.hidden, proj)
        self.fusion_layer = ConcatAttention3(3 * proj, proj) if fusion_method == "concatenation" else MCA3(proj)
        self.output_fc = LinearBNReLU(proj, 1, relu=False)
        self._head_flat = False

    def head_modules(self):
        return [self.text_fc, self.caption_text_fc, self.image_fine_tune, self.fusion_layer, self.output_fc]

    def _flatten_head(self):
        """The head's parameters live in one flat buffer (flatten_parameters), so the ONE fused ``memehip.Adam`` over
        ``get_params(lr)`` updates them in a single launch and counts them in the global clip norm."""
        if not self._head_flat and next(self.output_fc.parameters()).is_cuda:
            holder = nn.ModuleList(self.head_modules())
            flatten_parameters(holder)
            self._head_holder_flat = holder._memehip_flat
            self._head_flat = True

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._head_flat = False
        self._flatten_head()
        return self

    def get_params(self, lr: float):
        """Multimodal_example_task2C.py:645-664: fusion layer and everything that is neither text nor image model at
        ``lr``; ``text_model`` parameters (the caption tower's name ``caption_text_model`` contains it too) and
        ``image_model`` parameters (the whole CustomDenseNet161, its fine_tune MLP included) at ``0.8 * lr``."""
        self._flatten_head()
        attention, text, image = [], [], []
        for name, p in self.towers.named_parameters():
            (text if name.startswith("bert.") else image if name.startswith("image_model.") else attention).append(p)
        text += list(self.caption_text_model.parameters())
        image += list(self.image_fine_tune.parameters())
        for m in (self.text_fc, self.caption_text_fc, self.fusion_layer, self.output_fc):
            attention += list(m.parameters())
        return [{"params": attention, "lr": lr}, {"params": text, "lr": lr * 0.8}, {"params": image, "lr": lr * 0.8}]

    # ---- checkpoints of the reference module ------------------------------------------------------------------------------
    _REF_PREFIXES = (("text_model.model.", "towers.bert."), ("caption_text_model.model.", "caption_text_model."),
                     ("image_model.fine_tune.", "image_fine_tune."))

    @staticmethod
    def _timm_vit_to_hf(sd: dict) -> dict:
        """timm ``VisionTransformer`` keys (``blocks.N.attn.qkv`` fused, ``patch_embed.proj``, ``cls_token``, ``pos_embed``, ``norm``;
        what ``timm.create_model(...)`` + ``reset_classifier(0)`` holds, Multimodal_example_task2C.py:569-570) -> the transformers-4.39.2
        ViTModel keys the towers use."""
        out = {}
        top = {"cls_token": "embeddings.cls_token", "pos_embed": "embeddings.position_embeddings",
               "patch_embed.proj.weight": "embeddings.patch_embeddings.projection.weight",
               "patch_embed.proj.bias": "embeddings.patch_embeddings.projection.bias", "norm.weight": "layernorm.weight",
               "norm.bias": "layernorm.bias"}
        blk = {"norm1": "layernorm_before", "norm2": "layernorm_after", "attn.proj": "attention.output.dense", "mlp.fc1": "intermediate.dense",
               "mlp.fc2": "output.dense"}
        for k, v in sd.items():
            if k in top:
                out[top[k]] = v
                continue
            parts = k.split(".")
            if parts[0] != "blocks":
                continue                      # head / fc_norm: Identity after reset_classifier(0)
            i, rest, wb = parts[1], ".".join(parts[2:-1]), parts[-1]
            L = f"encoder.layer.{i}."
            if rest == "attn.qkv":
                D = v.shape[0] // 3
                for j, q in enumerate(("query", "key", "value")):
                    out[L + f"attention.attention.{q}.{wb}"] = v[j * D:(j + 1) * D]
            elif rest in blk:
                out[L + blk[rest] + "." + wb] = v
        return out

    def load_reference_state_dict(self, state_dict: dict, strict: bool = True):
        """Loads a ``state_dict()`` of the reference's ``MultimodalClassifier`` (Multimodal_example_task2C.py:587-643): ``text_model.model.*``
        / ``caption_text_model.model.*`` (transformers BertModel keys), ``image_model.image_model.*`` (timm ViT keys, or the
        transformers ViTModel keys), ``image_model.fine_tune.*``, and the head modules under their own names."""
        mapped, vit = {}, {}
        for k, v in state_dict.items():
            if k.startswith("image_model.image_model."):
                vit[k[len("image_model.image_model."):]] = v
                continue
            for a, b in self._REF_PREFIXES:
                if k.startswith(a):
                    k = b + k[len(a):]
                    break
            mapped[k] = v
        if any(k.startswith("blocks.") for k in vit):
            vit = self._timm_vit_to_hf(vit)
        mapped.update({"towers.image_model." + k: v for k, v in vit.items()})
        own = set(self.state_dict().keys())
        # the towers module carries the organizers' four-Linear head as well (unused here): not part of the reference checkpoint
        res = self.load_state_dict(mapped, strict=False)
        missing = [k for k in res.missing_keys if not (k.startswith("towers.") and "_fc." in k)]
        unexpected = [k for k in res.unexpected_keys if k in mapped and k not in own and not k.endswith("position_ids")
                      and "token_type_ids" not in k and "pooler." not in k]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_reference_state_dict: missing {missing[:4]} unexpected {unexpected[:4]}")
        return nn.modules.module._IncompatibleKeys(missing, unexpected)

    def forward(self, text, image, mask, caption_text, caption_text_mask):
        for t_ in (text, image, mask, caption_text, caption_text_mask):
            if not t_.is_cuda:
                raise _lib.MemehipError("memehip runs on the HIP device only (no CPU fallback): move the batch with .to(device)")
        t, v = self.towers.encode(text, image, mask)
        c = self.caption_text_model(caption_text, caption_text_mask)
        text_output = self.text_fc(self.text_dropout(t))
        caption_output = self.caption_text_fc(self.caption_text_dropout(c))
        image_output = self.image_fine_tune(v)
        fused_output = self.fusion_layer(text_output, image_output, caption_output)
        return self.output_fc(fused_output).squeeze(1)
