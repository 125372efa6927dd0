"""MultimodalClassifier / Adam / CrossEntropyLoss: the reference's Python surface over the HIP path.

Mirrors example_scripts/Multimodal_example_task2C.txt:
  * ``MultimodalClassifier(num_classes)`` with ``forward(text, image, mask) -> logits`` (:152-197,
    positional order text, image, mask), ``.to() / .train() / .eval() / .parameters() /
    .state_dict()``;
  * ``criterion = CrossEntropyLoss()`` (:248) and ``optimizer = Adam(model.parameters(), lr=2e-5)``
    (:249) with ``zero_grad() / step()``, used exactly as the reference loop does (:205-217):
    ``output = model(text, image, mask); loss = criterion(output, labels); loss.backward();
    optimizer.step()``.
Pooling: ``pool="last"`` is the organizers' ``[:, -1, :]`` (:178); ``pool="cls"`` is
Multimodal_example_task2C.py:359-360.  Unknown pooling raises ValueError like ...task2C.py:352.

Everything numeric runs in libmemehip.so on the current HIP stream.  ``torch.autograd`` only sees
two opaque nodes (the model, the loss); parameters are fp32 views of one flat buffer, gradients
fp32 views of a second one, GEMM operands a bf16 shadow of the matrix prefix.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import weakref
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib, ops
from .config import Layout, ModelConfig, TextConfig, ImageConfig  # noqa: F401
from .engine import Engine, Plan

BF16, F32 = torch.bfloat16, torch.float32

# flat-parameter storage address -> owning model (lets Adam(model.parameters()) find the bf16 shadow)
_REGISTRY: "weakref.WeakValueDictionary[int, MultimodalClassifier]" = weakref.WeakValueDictionary()
# flat-parameter storage address of a flatten_parameters() head -> its flat gradient buffer (zeroed by Adam.zero_grad)
_LOOSE: "weakref.WeakValueDictionary[int, torch.Tensor]" = weakref.WeakValueDictionary()


class _Node(nn.Module):
    """Name-space container so state_dict() keys equal the transformers / reference names."""


def _register(root: nn.Module, dotted: str, param: nn.Parameter):
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    node.register_parameter(parts[-1], param)


def _run_forward(plan: Plan) -> int:
    """Run the plan's forward launches; returns the plan's forward generation.  The activations a backward needs live
    in the plan's static buffers, so a second forward of the same shape before that backward would silently replace
    them: every forward bumps the generation and the autograd nodes refuse a backward whose generation is stale."""
    plan.generation = getattr(plan, "generation", 0) + 1
    plan.fwd.run(torch.cuda.current_stream().cuda_stream)
    return plan.generation


def _check_generation(ctx):
    if ctx.generation != ctx.plan.generation:
        raise RuntimeError(
            "memehip: backward() of a forward whose activations were overwritten by a later forward of the same shape "
            "(the launch plan keeps ONE set of activations per (batch, seq_len)); run backward before the next forward. "
            "Gradient accumulation over micro-batches is not supported: every backward overwrites .grad.")


class _ModelFn(torch.autograd.Function):
    """Opaque autograd node: forward = Plan.fwd, backward = the backward segments."""

    @staticmethod
    def forward(ctx, anchor, model, plan):
        ctx.model, ctx.plan = model, plan
        ctx.generation = _run_forward(plan)
        return plan.buf["logits"].clone()

    @staticmethod
    def backward(ctx, dlogits):
        model, plan = ctx.model, ctx.plan
        _check_generation(ctx)
        plan.buf["dlogits"].copy_(dlogits.to(F32))
        model._run_backward(plan)
        return None, None, None


class _EncodeFn(torch.autograd.Function):
    """Opaque autograd node of the two towers alone: forward = a features plan (stops at the pooled features),
    backward = its segments, started from the gradient of the pooled features."""

    @staticmethod
    def forward(ctx, anchor, model, plan):
        ctx.model, ctx.plan = model, plan
        ctx.generation = _run_forward(plan)
        return plan.buf["h.pooled"].clone()

    @staticmethod
    def backward(ctx, d_pooled):
        model, plan = ctx.model, ctx.plan
        _check_generation(ctx)
        plan.buf["h.d_pooled"].copy_(d_pooled.to(F32))
        model._run_backward(plan)
        return None, None, None


class _SequenceFn(torch.autograd.Function):
    """The towers with their whole last hidden states as the boundary (a "sequence" features plan)."""

    @staticmethod
    def forward(ctx, anchor, model, plan):
        ctx.model, ctx.plan = model, plan
        ctx.generation = _run_forward(plan)
        B, S = plan.B, plan.S
        return plan.buf["t.xlast32"].view(B, S, -1).clone(), plan.buf["i.xf32"].view(B, -1, model.config.image.hidden).clone()

    @staticmethod
    def backward(ctx, d_text, d_image):
        model, plan = ctx.model, ctx.plan
        _check_generation(ctx)
        scale = model.config.stream_scale          # the 16-bit gradient streams carry this power of two (fp16 build)
        plan.buf["h.d_seq_t"].copy_((d_text.to(F32) * scale).reshape(plan.buf["h.d_seq_t"].shape))
        plan.buf["h.d_seq_i"].copy_((d_image.to(F32) * scale).reshape(plan.buf["h.d_seq_i"].shape))
        model._run_backward(plan)
        return None, None, None


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, crit):
        B, Cn = logits.shape
        w = crit._workspace(logits.device, B, Cn)
        ops.ce_fwd_bwd(logits.contiguous(), labels, w["loss"], w["dlogits"], w["ncorrect"])
        ctx.dl = w["dlogits"]
        return w["loss"][0].clone()

    @staticmethod
    def backward(ctx, gout):
        return ctx.dl * gout, None, None


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() stand-in (mean reduction) running the fused HIP loss kernel; also
    exposes ``last_correct`` = #argmax==label of the last call (the reference computes it with
    torch.max, ...task2C.txt:219-220)."""

    def __init__(self):
        super().__init__()
        self._ws: Dict = {}

    def _workspace(self, device, B, Cn):
        key = (str(device), B, Cn)
        if key not in self._ws:
            self._ws[key] = dict(loss=torch.zeros(1, device=device), dlogits=torch.zeros((B, Cn), device=device),
                                 ncorrect=torch.zeros(1, dtype=torch.int32, device=device))
        self._last = self._ws[key]
        return self._ws[key]

    @property
    def last_correct(self) -> torch.Tensor:
        return self._last["ncorrect"]

    def forward(self, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        if not logits.is_cuda:
            raise _lib.MemehipError("CrossEntropyLoss: logits must be on the HIP device (no CPU fallback)")
        return _LossFn.apply(logits, labels, self)


class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, crit):
        flat = logits.reshape(-1).contiguous()
        w = crit._workspace(logits.device, flat.numel())
        ops.focal_fwd_bwd(flat, targets.to(F32).reshape(-1).contiguous(), w["loss"], w["dlogits"], w["ncorrect"], crit.alpha,
                          crit.gamma)
        ctx.dl, ctx.shape = w["dlogits"], logits.shape
        return w["loss"][0].clone()

    @staticmethod
    def backward(ctx, gout):
        return (ctx.dl * gout).view(ctx.shape), None, None


class _BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, mod, relu):
        x = x.contiguous().to(F32)
        training = mod.training or mod.running_mean is None
        y, sm, sr = ops.bn1d_fwd(x, gamma, beta, mod.running_mean, mod.running_var, mod.eps, mod.momentum, training, relu)
        if training and mod.num_batches_tracked is not None:
            mod.num_batches_tracked += 1
        ctx.save_for_backward(x, y, gamma, sm, sr)
        ctx.relu, ctx.training = relu, training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, sm, sr = ctx.saved_tensors
        dx, dg, db = ops.bn1d_bwd(dy.contiguous().to(F32), x, y, gamma, sm, sr, ctx.relu, frozen_stats=not ctx.training)
        return dx, dg, db, None, None


class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d over [B, F] running the HIP kernels (mh_bn1d_fwd / mh_bn1d_bwd); ``relu=True`` fuses the ReLU that
    follows it in Kevin's ``Linear + BatchNorm1d + ReLU`` projections (Multimodal_example_task2C.py:603-605).  Same
    parameters, buffers and state_dict keys as the torch module (affine, running statistics, momentum 0.1)."""

    def __init__(self, num_features: int, eps: float = 1e-5, momentum: float = 0.1, relu: bool = False):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=True, track_running_stats=True)
        self.relu = relu

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 2 or not x.is_cuda:
            raise _lib.MemehipError("memehip.BatchNorm1d: expected a [B, F] tensor on the HIP device (no CPU fallback)")
        return _BatchNormFn.apply(x, self.weight, self.bias, self, self.relu)


class SigmoidFocalLoss(nn.Module):
    """``criterion = sigmoid_focal_loss`` of Multimodal_example_task2C.py:167 called as
    ``criterion(output, labels, alpha=0.25, gamma=2.0, reduction='mean')`` (:711), on one logit per sample
    (a model built with ``num_classes=1``)."""

    def __init__(self, alpha: float = 0.25, gamma: float = 2.0):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma
        self._ws: Dict = {}

    def _workspace(self, device, B):
        key = (str(device), B)
        if key not in self._ws:
            self._ws[key] = dict(loss=torch.zeros(1, device=device), dlogits=torch.zeros(B, device=device),
                                 ncorrect=torch.zeros(1, dtype=torch.int32, device=device))
        self._last = self._ws[key]
        return self._ws[key]

    @property
    def last_correct(self) -> torch.Tensor:
        return self._last["ncorrect"]

    def forward(self, logits, targets, alpha=None, gamma=None, reduction: str = "mean"):
        if reduction != "mean":
            raise ValueError("SigmoidFocalLoss supports reduction='mean' (the reference's call)")
        if alpha is not None:
            self.alpha = alpha
        if gamma is not None:
            self.gamma = gamma
        if not logits.is_cuda:
            raise _lib.MemehipError("SigmoidFocalLoss: logits must be on the HIP device (no CPU fallback)")
        return _FocalFn.apply(logits, targets, self)


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int, last_epoch: int = -1):
    """transformers.get_linear_schedule_with_warmup (Multimodal_example_task2C.py:172-174): lr multiplier rises
    linearly 0 -> 1 over the warm-up steps, then decays linearly to 0 at num_training_steps."""
    def lr_lambda(step: int):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda, last_epoch)


class MultimodalClassifier(nn.Module):
    """Dual-encoder late-fusion classifier (BERT-family text tower + ViT image tower)."""

    def __init__(self, num_classes: int = 2, config: Optional[ModelConfig] = None, device="cpu", seed: int = 0,
                 init: bool = True):
        super().__init__()
        # no config given = the reference's module as a user constructs it (Multimodal_example_task2C.txt:152-170): Dropout(0.3) on the
        # text features, the BERT checkpoint's 0.1 / 0.1; an explicit ModelConfig says what it wants (the dataclass defaults are
        # p = 0: the parity / measurement setting of BASELINE.md section 3)
        cfg = config if config is not None else ModelConfig(num_classes=num_classes).with_reference_dropout()
        if config is None:
            cfg.num_classes = num_classes
        cfg.validate()
        self.config = cfg
        self.layout = Layout(cfg)
        n = self.layout.n_total
        self._P = torch.zeros(n, dtype=F32, device=device)
        self._G = torch.zeros(n, dtype=F32, device=device)
        self._T16 = torch.float16 if cfg.compute_dtype == "fp16" else BF16
        self._SH = torch.zeros(self.layout.n_shadow, dtype=self._T16, device=device)
        self._names = self.layout.state_dict_order()
        self._params: Dict[str, nn.Parameter] = {}      # keyed by the layout's (internal) names
        for name in self._names:
            s = self.layout.spec[name]
            prm = nn.Parameter(self._P[s.offset:s.offset + s.numel].view(s.shape), requires_grad=True)
            self._params[name] = prm
            _register(self, self._external_name(name), prm)
        self._engine: Optional[Engine] = None
        self._shadow_stale = True
        self._shadow_version = -1       # self._P._version the 16-bit shadow was cast from (see weights_changed)
        self._rng_seed, self._rng_step = 0x5EED1234, 0
        self._rng_ring = None
        if init:
            self.reset_parameters(seed)
        self._attach_grads()
        _REGISTRY[self._P.untyped_storage().data_ptr()] = self

    def _external_name(self, name: str) -> str:
        """state_dict / named_parameters name of a layout entry: the image projection is ``image_fc`` or, with
        ``config.image_fc_name = "resnet_fc"``, the organizers' ``resnet_fc`` (Multimodal_example_task2C.txt:165)."""
        if name.startswith("image_fc."):
            return self.config.image_fc_name + name[len("image_fc"):]
        return name

    # ---- construction helpers -------------------------------------------------------------------------
    @classmethod
    def from_config(cls, config: ModelConfig, device="cpu", seed: int = 0, init: bool = True):
        return cls(config.num_classes, config=config, device=device, seed=seed, init=init)

    def reset_parameters(self, seed: int = 0):
        """Random init with the towers' initializer_range 0.02 (no checkpoints offline) and
        nn.Linear defaults for the head."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.no_grad():
            for s in self.layout.specs:
                kind = s.init
                if kind == "normal":
                    val = torch.randn(s.numel, generator=g) * 0.02
                elif kind == "ones":
                    val = torch.ones(s.numel)
                elif kind == "zeros":
                    val = torch.zeros(s.numel)
                else:
                    bound = 1.0 / math.sqrt(int(kind.split(":")[1]))
                    val = (torch.rand(s.numel, generator=g) * 2 - 1) * bound
                self._P[s.offset:s.offset + s.numel].copy_(val)
        self._shadow_stale = True

    def _attach_grads(self):
        for name, prm in self._params.items():
            s = self.layout.spec[name]
            prm.grad = self._G[s.offset:s.offset + s.numel].view(s.shape)

    # ---- nn.Module protocol ---------------------------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        new_p = fn(self._P)
        if new_p.dtype != F32:
            raise TypeError("MultimodalClassifier keeps fp32 master parameters; bf16 compute is internal")
        if new_p.device != self._P.device or new_p.data_ptr() != self._P.data_ptr():
            self._P = new_p
            self._G = fn(self._G)
            self._SH = torch.zeros(self.layout.n_shadow, dtype=self._T16, device=new_p.device)
            for name, prm in self._params.items():
                s = self.layout.spec[name]
                prm.data = self._P[s.offset:s.offset + s.numel].view(s.shape)
            self._attach_grads()
            self._engine = None
            self._shadow_stale = True
            _REGISTRY[self._P.untyped_storage().data_ptr()] = self
        return self

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = {("image_fc" + k[len("resnet_fc"):] if k.startswith("resnet_fc.") else k): v for k, v in state_dict.items()}
        missing = [self._external_name(k) for k in self._names if k not in sd]
        unexpected = [k for k in sd if k not in self._params]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:4]} unexpected {unexpected[:4]}")
        with torch.no_grad():
            for k, val in sd.items():
                if k in self._params:
                    if tuple(val.shape) != tuple(self._params[k].shape):
                        raise RuntimeError(f"size mismatch for {k}: {tuple(val.shape)} vs {tuple(self._params[k].shape)}")
                    self._params[k].data.copy_(val.to(F32))
        self._shadow_stale = True
        return nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ---- engine ------------------------------------------------------------------------------------------------
    def _get_engine(self) -> Engine:
        if not self._P.is_cuda:
            raise _lib.MemehipError("MultimodalClassifier runs only on a HIP device: call .to('cuda') first "
                                    "(libmemehip has no CPU fallback)")
        if self._engine is None:
            self._engine = Engine(self.config, self.layout, self._P, self._G, self._SH)
        return self._engine

    def refresh_shadow(self):
        """bf16 copy of the GEMM matrices (kept in sync by Adam.step(); call after editing weights by hand)."""
        self._get_engine()
        ops.cast_f32_bf16(self._P[:self.layout.n_shadow], self._SH)
        self._shadow_synced()

    def mark_weights_changed(self):
        self._shadow_stale = True

    def _weights_version(self) -> int:
        # every Parameter was created as a view of the first flat buffer and keeps sharing that buffer's autograd
        # version counter for life (Tensor.data = ... / .to(device) swap the storage, not the counter): one counter
        # for all parameters, bumped by any in-place torch op on any of them
        return self._params[self._names[0]]._version

    def _shadow_synced(self):
        self._shadow_stale = False
        self._shadow_version = self._weights_version()

    def weights_changed(self) -> bool:
        """True when the fp32 master weights were edited since the 16-bit shadow (what every tower GEMM reads) was last
        cast from them.  All parameters share one autograd version counter, so ANY in-place update through torch --
        ``torch.optim.Adam(model.parameters()).step()``, HF Trainer's AdamW, SGD, ``with torch.no_grad(): p.mul_(..)`` --
        is seen here without a hook; the fused ``memehip.Adam`` rewrites the shadow itself (mh_adam_step) and re-syncs.
        Edits that bypass the counter by construction (``p.data.mul_()``, raw pointers) need ``mark_weights_changed()``."""
        return self._shadow_stale or self._weights_version() != self._shadow_version

    def manual_seed(self, seed: int):
        """Seed of the dropout masks (stateless counter RNG: mask = f(seed, step, site, element))."""
        self._rng_seed, self._rng_step = int(seed) & 0x7FFFFFFFFFFFFFFF, 0

    def _advance_rng(self, plan: Plan):
        """New dropout masks for the next step: writes {seed_lo, seed_hi, step, 0} to the plan's device words
        (pinned ring + async copy, so a captured graph picks the new values up on replay)."""
        if not getattr(plan, "dropout_on", False):
            return
        if self._rng_ring is None:
            self._rng_ring = torch.zeros((32, 4), dtype=torch.int32).pin_memory()
            self._rng_events = [None] * 32
        self._rng_step += 1
        slot = self._rng_step % 32
        if self._rng_events[slot] is not None:        # the copy issued from this slot 32 steps ago must have run before it is rewritten
            self._rng_events[slot].synchronize()
        host = self._rng_ring[slot]
        lo, hi = self._rng_seed & 0x7FFFFFFF, (self._rng_seed >> 31) & 0x7FFFFFFF
        host.copy_(torch.tensor([lo, hi, self._rng_step & 0x7FFFFFFF, 0], dtype=torch.int32))
        plan.buf["rng"].copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._rng_events[slot] = ev

    def _prepare(self, text, image, mask, labels=None, features=False) -> Plan:
        eng = self._get_engine()
        if text.dim() != 2 or mask.shape != text.shape:
            raise ValueError("text and mask must be [B, S] int64 tensors of the same shape")
        B, S = text.shape
        v = self.config.image
        if tuple(image.shape) != (B, v.channels, v.image_size, v.image_size):
            raise ValueError(f"image must be [B,{v.channels},{v.image_size},{v.image_size}], got {tuple(image.shape)}")
        plan = eng.plan(B, S, self.training, features=features)
        if self.weights_changed():
            self.refresh_shadow()
        if self.training:
            self._advance_rng(plan)
        plan.buf["ids"].copy_(text.to(torch.int64), non_blocking=True)
        plan.buf["mask"].copy_(mask.to(torch.int64), non_blocking=True)
        plan.buf["image"].copy_(image.to(F32), non_blocking=True)
        if labels is not None:
            plan.buf["labels"].copy_(labels.to(torch.int64), non_blocking=True)
        return plan

    def _run_backward(self, plan: Plan, hook=None):
        self._get_engine().before_backward(plan)
        stream = torch.cuda.current_stream().cuda_stream
        for seg in plan.bwd:
            seg.run(stream)
            if hook is not None:
                hook(seg.name, plan.bucket_after.get(seg.name))
        self._attach_grads()

    def forward(self, text: torch.Tensor, image: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        plan = self._prepare(text, image, mask)
        if torch.is_grad_enabled():          # eval mode only switches dropout off; the reference back-propagates through eval-mode modules too
            anchor = self._params[self._names[0]]
            return _ModelFn.apply(anchor, self, plan)
        _run_forward(plan)
        return plan.buf["logits"].clone()

    # ---- fused step (no autograd): forward + loss + backward ---------------------------------------------------
    def encode(self, text, image, mask):
        """The two towers without the built-in head: returns ``(text_features [B, Dt], image_features [B, Di])`` (f32) --
        the pooled text token (``pool``) and the ViT class token after the final LayerNorm -- differentiable, so any
        PyTorch head can sit on top: Kevin's ``Linear + BatchNorm1d + ReLU`` projections, ``ConcatAttention3``, the
        1-logit ``Linear(512, 1) + BatchNorm1d(1)`` with the focal loss (Multimodal_example_task2C.py:590-685).
        The built-in head's parameters receive zero gradients on this path."""
        for t_ in (text, image, mask):
            if not t_.is_cuda:
                raise _lib.MemehipError("memehip runs on the HIP device only (no CPU fallback): move the batch with .to(device)")
        plan = self._prepare(text, image, mask, features=True)
        Dt = self.config.text.hidden
        if torch.is_grad_enabled():          # eval mode only switches dropout off; the reference back-propagates through eval-mode modules too
            pooled = _EncodeFn.apply(self._params[self._names[0]], self, plan)
        else:
            _run_forward(plan)
            pooled = plan.buf["h.pooled"].clone()
        return pooled[:, :Dt], pooled[:, Dt:]

    def encode_sequence(self, text, image, mask):
        """Like ``encode`` but returns the towers' whole last hidden states, ``(text [B,S,Dt], image [B,Nt,Di])`` f32 and
        differentiable, for the poolings of ``LLMWithClassificationHead`` that read every position -- max, masked mean,
        tanh-attention, conv1d-max (Multimodal_example_task2C.py:356-392) -- written in PyTorch on top.  Every position is
        computed, padded ones included (the reference's max / conv poolings read them)."""
        for t_ in (text, image, mask):
            if not t_.is_cuda:
                raise _lib.MemehipError("memehip runs on the HIP device only (no CPU fallback): move the batch with .to(device)")
        plan = self._prepare(text, image, mask, features="sequence")
        if torch.is_grad_enabled():          # eval mode only switches dropout off; the reference back-propagates through eval-mode modules too
            return _SequenceFn.apply(self._params[self._names[0]], self, plan)
        _run_forward(plan)
        B, S = plan.B, plan.S
        return (plan.buf["t.xlast32"].view(B, S, -1).clone(),
                plan.buf["i.xf32"].view(B, -1, self.config.image.hidden).clone())

    @torch.no_grad()
    def get_features(self, text, image, mask, pooler=None):
        """Forward-only feature dump for the SVM baseline (baselines/extract_feat.py:52-67): dict of f32 tensors.
        ``pooler = (weight [D, D], bias [D])`` of the checkpoint's ``bert.pooler.dense`` additionally yields BertModel's
        ``pooler_output`` = tanh(W h_cls + b), which is what the baseline script stores for the text side (the pooler
        is not part of the fine-tune path, so its two tensors are not model parameters here)."""
        was = self.training
        self.eval()
        try:
            t_, i_ = self.encode(text, image, mask)
        finally:
            self.train(was)
        out = {"text": t_.clone(), "image": i_.clone()}
        if pooler is not None:      # BertPooler: tanh(dense(h_cls)), one exact-f32 HIP GEMM with the tanh in its epilogue
            from . import fused
            w, b = (x.to(t_.device, F32).contiguous() for x in pooler)
            out["pooler_output"] = fused.linear(out["text"], w, b, act="tanh")
        return out

    def forward_backward(self, text, image, mask, labels, grad_hook=None):
        """Returns (loss[1], n_correct[1], logits[B,C]) device tensors; gradients land in .grad."""
        plan = self._prepare(text, image, mask, labels)
        stream = torch.cuda.current_stream().cuda_stream
        _run_forward(plan)
        plan.loss.run(stream)
        self._run_backward(plan, grad_hook)
        return plan.buf["loss"], plan.buf["ncorrect"], plan.buf["logits"]

    def get_params(self, lr: float):
        """Kevin's parameter groups (Multimodal_example_task2C.py:645-664): fusion / head parameters at ``lr``,
        text-encoder and image-encoder parameters at ``0.8 * lr``."""
        head, text, image = [], [], []
        for name, p in self.named_parameters():
            (text if name.startswith("bert.") else image if name.startswith("image_model.") else head).append(p)
        return [{"params": head, "lr": lr}, {"params": text, "lr": lr * 0.8}, {"params": image, "lr": lr * 0.8}]

    @property
    def flat_params(self):
        return self._P

    @property
    def flat_grads(self):
        return self._G

    @property
    def flat_shadow(self):
        return self._SH


class TextEncoder(nn.Module):
    """A text tower on its own -- e.g. the caption tower of Kevin's three-tower model
    (``LLMWithClassificationHead(english_text_model, "cls")``, Multimodal_example_task2C.py:608-611, forward :668-670).

    The launch plan is built for two lockstep towers, so this wraps a ``MultimodalClassifier`` whose image side is a
    stub ViT (one 128-wide layer over a single 16x16 patch: a few microseconds of kernels inside the grouped
    launches) fed with a constant image, and exposes only the text half:
    ``forward(input_ids, attention_mask) -> pooled [B, D]`` (f32, differentiable), ``hidden_states(...)`` for the
    non-cls poolings.  ``state_dict()`` / ``load_state_dict()`` use the BertModel key names without the ``bert.``
    prefix of the two-tower module; the stub's parameters get zero gradients (the head on top never reads them), so one
    fused ``Adam(encoder.parameters())`` leaves them unchanged."""

    def __init__(self, text: "TextConfig", pool: str = "cls", compute_dtype: str = "bf16", seed: int = 0, naming: str = "bert",
                 grad_stream_scale: float = 0.0):
        super().__init__()
        if naming not in ("bert", "distilbert"):
            raise ValueError(f"naming must be 'bert' or 'distilbert', got {naming!r}")
        self.naming = naming
        from .config import ImageConfig
        stub = ImageConfig(image_size=16, patch=16, hidden=128, layers=1, heads=2, intermediate=128)
        cfg = ModelConfig(text=text, image=stub, proj=128, num_classes=2, pool=pool, compute_dtype=compute_dtype,
                          grad_stream_scale=grad_stream_scale)
        self.inner = MultimodalClassifier.from_config(cfg, seed=seed)
        self._image = None

    def _stub_image(self, B, device):
        if self._image is None or self._image.shape[0] != B or self._image.device != device:
            self._image = torch.zeros((B, 3, 16, 16), dtype=F32, device=device)
        return self._image

    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        t, _ = self.inner.encode(input_ids, self._stub_image(input_ids.shape[0], input_ids.device), attention_mask)
        return t

    def hidden_states(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        h, _ = self.inner.encode_sequence(input_ids, self._stub_image(input_ids.shape[0], input_ids.device), attention_mask)
        return h

    # DistilBertModel (transformers 4.39.2) names <-> the BertModel names the layout uses: same post-LN arithmetic
    _DISTIL = (("transformer.layer.", "encoder.layer."), (".attention.q_lin.", ".attention.self.query."),
               (".attention.k_lin.", ".attention.self.key."), (".attention.v_lin.", ".attention.self.value."),
               (".attention.out_lin.", ".attention.output.dense."), (".sa_layer_norm.", ".attention.output.LayerNorm."),
               (".ffn.lin1.", ".intermediate.dense."), (".ffn.lin2.", ".output.dense."), (".output_layer_norm.", ".output.LayerNorm."))

    @classmethod
    def _to_bert_name(cls, k: str) -> str:
        for a, b in cls._DISTIL:
            k = k.replace(a, b)
        return k

    @classmethod
    def _to_distil_name(cls, k: str) -> str:
        for a, b in cls._DISTIL:
            k = k.replace(b, a)
        return k

    def state_dict(self, *args, destination=None, prefix: str = "", keep_vars: bool = False, **kwargs):
        """BertModel key names without the two-tower module's ``bert.`` prefix (``naming="distilbert"``: DistilBertModel's
        ``transformer.layer.N.attention.q_lin.weight`` ...), so a parent module's state_dict() carries e.g. ``bert.embeddings...``
        exactly as the reference's ``self.bert = AutoModel.from_pretrained(...)`` does."""
        inner = self.inner.state_dict(keep_vars=keep_vars)
        out = destination if destination is not None else {}
        for k, v in inner.items():
            if k.startswith("bert."):
                name = k[len("bert."):]
                out[prefix + (self._to_distil_name(name) if self.naming == "distilbert" else name)] = v
        return out

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Called when a PARENT module loads a checkpoint: take this encoder's keys (either naming) out of it."""
        mine = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        res = self.load_state_dict(mine, strict=False)
        missing_keys.extend(prefix + k for k in res.missing_keys)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = {("bert." + self._to_bert_name(k)): v for k, v in state_dict.items()}
        res = self.inner.load_state_dict(sd, strict=False)
        missing = [k[len("bert."):] for k in res.missing_keys if k.startswith("bert.")]
        if strict and (missing or res.unexpected_keys):
            raise RuntimeError(f"load_state_dict: missing {missing[:4]} unexpected {list(res.unexpected_keys)[:4]}")
        return nn.modules.module._IncompatibleKeys(missing, list(res.unexpected_keys))


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam / AdamW semantics over the model's flat buffers in fused HIP launches (dense over all
    parameters, as the reference's optim.Adam is): one launch per contiguous run of a parameter group, so
    ``Adam(model.parameters())`` is ONE launch and ``Adam(model.get_params(lr))`` (Kevin's 0.8x encoder learning
    rate, Multimodal_example_task2C.py:645-664,168) is three.  ``max_grad_norm`` adds the clip of
    DistilBERT_example_task2A.ipynb:3280 / ...task2C.py:713-715 fused into the same launches; learning-rate
    schedulers (torch.optim.lr_scheduler.*) work unchanged because the rate is re-read every step."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, decoupled_weight_decay: bool = False, max_grad_norm: Optional[float] = None,
                 model: Optional[MultimodalClassifier] = None, skip_untouched_embedding_rows: bool = True,
                 skip_nonfinite=None, clip_scaled_gradients: bool = False):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.decoupled = decoupled_weight_decay
        self.max_grad_norm = max_grad_norm
        # clip_scaled_gradients: the clip coefficient min(1, max_grad_norm / norm) is computed on the norm of the LOSS-SCALED gradients
        # (true norm x GradScaler scale x the fp16 build's static stream scale) -- what the reference's default branch does: USE_FP16 =
        # True (Multimodal_example_task2C.py:60), clip_grad_norm_(model.parameters(), 1.0) on the gradients scaler.scale(loss).backward()
        # left, no unscale_ first, then scaler.step (:712-717).  With torch's initial scale of 65536 that clips nearly every step to a
        # true-gradient norm of 1.5e-5 -- elements ~1e-9, below Adam's eps -- so the update is a small fraction of lr.  False (default):
        # clip the true gradients (what the fp32 branch does and what the fp16 branch presumably meant).  kevin.train() turns it on when
        # a scaler is passed, as the reference behaves.
        self.clip_scaled_gradients = bool(clip_scaled_gradients)
        # skip_nonfinite: never let a non-finite gradient (an overflowed fp16 gradient stream) into the master weights -- the
        # reference's fp16 branch gets this from GradScaler (Multimodal_example_task2C.py:60-64,712-717).
        #   None (default): on when a model this optimizer updates stores 16-bit values as fp16, or a GradScaler is attached
        #   True: ``step()`` computes the global norm and skips the whole step when it is inf / nan (all or nothing, exactly
        #         GradScaler.step); inside GraphedStep's optimizer-in-backward schedule, where a layer's slice is updated before
        #         the global norm can exist, the GUARDED kernels are used instead: the first slice that meets a non-finite gradient
        #         stops the step from there on (an overflow at the loss skips all of it, as GradScaler does; one that appears further
        #         down the gradient stream leaves the layers above it updated with the finite gradients they had), and the step is
        #         counted as an overflow (loss scale backed off)
        #   "strict": all or nothing everywhere (turns the optimizer-in-backward overlap off, like clipping does)
        # Clipping implies the all-or-nothing check.
        if skip_nonfinite not in (None, True, False, "strict"):
            raise ValueError(f"skip_nonfinite must be None, True, False or 'strict', got {skip_nonfinite!r}")
        self.skip_nonfinite = skip_nonfinite
        self._scaler = None
        self._hyper_init = False
        self._model = model
        self._flat = None
        self._step = 0
        self._grad_scale = 1.0         # DDP sets 1/world_size (all-reduce sums)
        # word-embedding rows that never received a gradient have g = m = v = 0: the dense Adam update is the identity
        # on them (weight_decay == 0), so they are skipped -- same numbers as torch.optim.Adam, ~1.4 GB less HBM traffic
        self.skip_untouched_rows = bool(skip_untouched_embedding_rows)

    @property
    def grad_scale(self) -> float:
        return self._grad_scale

    @grad_scale.setter
    def grad_scale(self, value: float):
        """hyper[7] lives on the device and, once the step's accounting kernel owns it, is not rewritten from the host: a change here
        (GraphedStep sets 1 / world from the reducer) must re-initialise it, or the clip threshold and the non-finite check keep the
        stale factor."""
        if float(value) != self._grad_scale:
            self._grad_scale = float(value)
            self._hyper_init = False

    def _clip_mult(self) -> float:
        """mh_adam_step's clip_norm_mult: 0 = clip the true gradients; else |grad_scale| x the static stream scale, so that the
        buffer's norm (true gradient x dynamic loss scale) times it is the norm of the gradients as the reference's scaler left them."""
        if not self.clip_scaled_gradients or self.max_grad_norm is None:
            return 0.0
        model = self._flat["model"] if self._flat is not None else self._model
        static = float(model.config.stream_scale) if (model is not None and hasattr(model, "config")) else 1.0
        return abs(self._grad_scale) * static

    def _bind(self):
        """Group the parameters by the flat buffer they live in.  Usually that is ONE MultimodalClassifier; Kevin's
        three-tower model (Multimodal_example_task2C.py:645-664: ONE optim.Adam over fusion head + text + caption + image
        parameters) spans several: the two-tower buffer, the caption TextEncoder's, and the head's
        (``flatten_parameters(head)``).  Each buffer is a *bucket* with its own moments; the gradient norm used for
        clipping is the GLOBAL one over all buckets, as clip_grad_norm_(model.parameters(), .) is (...task2C.py:713-715)."""
        if self._flat is not None:
            return
        by_storage: Dict[int, list] = {}
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                by_storage.setdefault(p.untyped_storage().data_ptr(), []).append((gi, p))
        buckets = []
        for sptr, items in by_storage.items():
            base = min((p for _, p in items), key=lambda p: p.data_ptr())
            storage = base.untyped_storage()
            n = storage.nbytes() // 4
            if not base.is_cuda:
                raise _lib.MemehipError("Adam: parameters must live on the HIP device (no CPU fallback)")
            if n % 4:
                raise ValueError(f"memehip.Adam: a parameter buffer of {n} elements is not a multiple of 4 (16-byte kernel "
                                 "accesses): re-home loose parameters with memehip.flatten_parameters(module) first")
            P = torch.empty(0, dtype=F32, device=base.device).set_(storage, 0, (n,), (1,))
            if base.grad is None:
                raise RuntimeError("Adam.step() before any backward")
            gs = base.grad.untyped_storage()
            for _, p in items:
                if p.grad is None or p.grad.untyped_storage().data_ptr() != gs.data_ptr() or \
                        p.grad.storage_offset() != p.storage_offset():
                    raise ValueError("memehip.Adam: the gradients must mirror the parameters' flat buffer (the parameters of a "
                                     "MultimodalClassifier / TextEncoder, or of a module passed through flatten_parameters)")
            G = torch.empty(0, dtype=F32, device=base.device).set_(gs, 0, (n,), (1,))
            dev = base.device
            # contiguous runs (in elements, 4-aligned) of every parameter group
            runs = []
            for gi in sorted({gi for gi, _ in items}):
                spans = sorted((p.storage_offset(), p.storage_offset() + (p.numel() + 3) // 4 * 4) for g2, p in items if g2 == gi)
                cur = None
                for a, b in spans:
                    if cur is not None and a <= cur[1]:
                        cur[1] = max(cur[1], b)
                    else:
                        if cur is not None:
                            runs.append((gi, cur[0], cur[1]))
                        cur = [a, b]
                if cur is not None:
                    runs.append((gi, cur[0], cur[1]))
            model = _REGISTRY.get(storage.data_ptr())
            buckets.append(dict(P=P, G=G, M=torch.zeros(n, device=dev), V=torch.zeros(n, device=dev),
                                ws=torch.zeros(1024, device=dev), nrm=torch.zeros(1, device=dev), runs=runs, model=model))
        # the bucket of the (first) two-tower model is the primary one: GraphedStep and the optimizer-in-backward slices
        # address it by flat offsets
        buckets.sort(key=lambda bk: 0 if (self._model is not None and bk["model"] is self._model) else
                     (1 if isinstance(bk["model"], MultimodalClassifier) else 2))
        if self._model is None:
            self._model = buckets[0]["model"]
        dev = buckets[0]["P"].device
        shared = dict(hyper=[torch.zeros(8, device=dev) for _ in self.param_groups],
                      # ONE pinned allocation per ring (tidiness only: round 3's experiments, profiles/r03_segfault_experiments.md,
                      # showed that the number of pinned blocks has nothing to do with the round-2 hipGraphLaunch segfault)
                      ring=torch.zeros((32, len(self.param_groups), 8), dtype=F32).pin_memory(),
                      nrm_total=torch.zeros(1, device=dev),
                      # GradScaler bookkeeping (mh_adam_skip_account): [skipped steps, last step skipped]; the host's step count
                      skip_state=torch.zeros(2, dtype=torch.int32, device=dev), step_dev=torch.zeros(1, dtype=torch.int32, device=dev),
                      overflow=torch.zeros(1, dtype=torch.int32, device=dev),
                      step_ring=torch.zeros((32, 1), dtype=torch.int32).pin_memory())
        if self.skip_nonfinite is None:
            self.skip_nonfinite = self._scaler is not None or any(
                bk["model"] is not None and bk["model"].config.compute_dtype == "fp16" for bk in buckets)
        if len(self.param_groups) > _lib.MH_ADAM_MAX_GROUPS and (self.max_grad_norm is not None or self.skip_nonfinite):
            raise ValueError(f"memehip.Adam: at most {_lib.MH_ADAM_MAX_GROUPS} parameter groups with clipping / skip_nonfinite")
        for bk in buckets:
            bk.update(shared)
        self._buckets = buckets
        self._flat = buckets[0]

    def zero_grad(self, set_to_none: bool = False):
        # every tower gradient is overwritten by the next backward (the embedding table re-zeroes the rows
        # it touched), so there is nothing to clear; the views stay attached.  Heads re-homed by
        # flatten_parameters() receive their gradients from autograd, which ACCUMULATES: zero those buffers.
        if self._flat is not None:
            for f in self._buckets:
                if f["model"] is None:
                    f["G"].zero_()
        else:
            seen = set()
            for g in self.param_groups:
                for p in g["params"]:
                    sp = p.untyped_storage().data_ptr()
                    if sp not in seen and sp in _LOOSE:
                        seen.add(sp)
                        _LOOSE[sp].zero_()
        return None

    @property
    def _accounted(self) -> bool:
        """mh_adam_skip_account runs every step (after the updates) and owns hyper[5..7] from the second step on."""
        return self.max_grad_norm is not None or bool(self.skip_nonfinite)

    def _resolve_skip(self, model=None):
        if self.skip_nonfinite is None and model is not None:
            self.skip_nonfinite = self._scaler is not None or model.config.compute_dtype == "fp16"

    def _attach_scaler(self, scaler):
        if self._scaler is not None and self._scaler is not scaler:
            raise ValueError("memehip.Adam: another GradScaler is already attached to this optimizer")
        if self._scaler is None:
            self._scaler = scaler
            self._hyper_init = False
            scaler._optimizers.append(self)        # a host-side change of the scale must reach hyper[7] (GradScaler._scale_written)
            if self.skip_nonfinite in (None, False):
                self.skip_nonfinite = True

    def _write_hyper(self):
        t = self._step
        slot = t % 32
        evs = self._flat.setdefault("ring_events", [None] * 32)
        if evs[slot] is not None:         # a host that runs > 32 steps ahead of the device would rewrite a slot whose copy is still pending
            evs[slot].synchronize()
        host = self._flat["ring"][slot]
        first = not (self._accounted and self._hyper_init)
        if first and self._accounted and t > 1:          # re-initialisation mid-run (a scaler attached, a checkpoint loaded)
            self._host_skipped = int(self._flat["skip_state"][0])
        for gi, g in enumerate(self.param_groups):
            b1, b2 = g["betas"]
            t_eff = max(t - (self._host_skipped if first else 0), 1)
            host[gi].copy_(torch.tensor([g["lr"], b1, b2, g["eps"], g["weight_decay"], 1.0 / (1.0 - b1 ** t_eff),
                                         1.0 / math.sqrt(1.0 - b2 ** t_eff), self.grad_scale], dtype=F32))
            if first:
                self._flat["hyper"][gi].copy_(host[gi], non_blocking=True)
                if self._scaler is not None and self._scaler.is_enabled():
                    self._flat["hyper"][gi][7:8].div_(self._scaler._tensor(self._flat["P"].device))
            else:           # bias corrections and grad_scale / loss scale were written on the device by the last step's accounting
                self._flat["hyper"][gi][:5].copy_(host[gi][:5], non_blocking=True)
        self._hyper_init = True
        self._flat["step_ring"][slot][0] = t
        self._flat["step_dev"].copy_(self._flat["step_ring"][slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        evs[slot] = ev

    _host_skipped = 0          # skipped steps known to the host when the device scalars are (re)initialised (load_state_dict)

    @torch.no_grad()
    def step(self, closure=None):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("memehip.Adam.step() inside a hipGraph capture would freeze the step count, the bias corrections and "
                               "the learning rate in the graph: write the per-step scalars outside (optimizer._step += 1; "
                               "optimizer._write_hyper()) and capture optimizer.launch() -- GraphedStep does exactly that")
        self._bind()
        self._step += 1
        self._write_hyper()
        self.launch()

    def launch(self, only: Optional[tuple] = None, skip: Optional[list] = None, guarded: bool = False):
        """Enqueue grad-norm (if clipping / all-or-nothing) + the fused update(s) + the per-step accounting; hyper-parameters are
        read from device memory.  ``only=(a, b)`` updates just that slice of the primary flat buffer (optimizer-in-backward: a
        layer's matrices are updated on the side stream as soon as their gradients are complete); ``skip`` = slices already done;
        ``guarded``: this launch belongs to a schedule that updates slices before the global norm exists (see skip_nonfinite)."""
        nrm, flag = None, None
        if self._accounted:
            if guarded and self.max_grad_norm is None and self.skip_nonfinite != "strict":
                flag = self._flat["overflow"]
            else:
                assert only is None, "clipping / the all-or-nothing check need the global gradient norm before any update"
                nrm = self._global_norm_sq()
        for bi, f in enumerate(self._buckets):
            if bi > 0 and only is not None:
                break
            self._launch_bucket(f, nrm, only if bi == 0 else None, skip if bi == 0 else None, flag)
        if self._accounted and only is None:
            self._skip_account(nrm, flag)

    def _skip_account(self, nrm, flag=None):
        """After the update launches of a step: count it if its gradients were not finite (the norm the exact kernels saw / the flag
        the guarded kernels raised), move the dynamic loss scale (GradScaler.update's rule) and write the bias corrections and
        grad_scale / loss scale the NEXT step's kernels read -- for the steps actually taken: GradScaler.step never calls
        optimizer.step() after an overflow, so torch's Adam does not advance t either.  One 64-thread launch, graph-replayable
        (the host's step count is read from device memory)."""
        f = self._flat
        self._guard_ordinal = 0
        grp = _lib.MhAdamSkipGroups()
        grp.n = len(self.param_groups)
        for gi, g in enumerate(self.param_groups):
            grp.hyper[gi] = f["hyper"][gi].data_ptr()
            grp.beta1[gi], grp.beta2[gi] = float(g["betas"][0]), float(g["betas"][1])
        ls = None
        sc = self._scaler if (self._scaler is not None and self._scaler.is_enabled()) else None
        if sc is not None or flag is not None:
            ls = _lib.MhLossScale()
            dev = f["P"].device
            ls.scale = sc._tensor(dev).data_ptr() if sc is not None else None
            ls.growth = sc._growth.data_ptr() if sc is not None else None
            ls.overflow = flag.data_ptr() if flag is not None else None
            ls.growth_factor, ls.backoff_factor = (sc.growth_factor, sc.backoff_factor) if sc is not None else (1.0, 1.0)
            ls.min_scale, ls.max_scale = (sc.min_scale, sc.max_scale) if sc is not None else (1.0, 1.0)
            ls.growth_interval = sc.growth_interval if sc is not None else 1
            ls.base_grad_scale = float(self.grad_scale)
        _lib.check(_lib.load().mh_adam_skip_account(C.byref(grp), None if nrm is None else nrm.data_ptr(), f["skip_state"].data_ptr(),
                                                    f["step_dev"].data_ptr(), None if ls is None else C.byref(ls),
                                                    torch.cuda.current_stream().cuda_stream), "mh_adam_skip_account")

    @property
    def skipped_steps(self) -> int:
        """Steps skipped so far because the gradient norm was inf / nan (synchronises; read it lazily, e.g. once per epoch, to
        lower a static fp16 gradient-stream scale or to warn about a run that keeps overflowing)."""
        self._bind()
        return int(self._flat["skip_state"][0])

    @property
    def last_step_skipped(self) -> bool:
        self._bind()
        return bool(int(self._flat["skip_state"][1]))

    def _global_norm_sq(self) -> torch.Tensor:
        """Sum of squares of every gradient this optimizer owns (device f32[1]): one mh_sumsq_f32 per flat buffer."""
        for f in self._buckets:
            ops.sumsq(f["G"], f["ws"], f["nrm"])
        if len(self._buckets) == 1:
            return self._buckets[0]["nrm"]
        total = self._flat["nrm_total"]
        torch.sum(torch.stack([f["nrm"][0] for f in self._buckets]).view(1, -1), dim=1, out=total)
        return total

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the gradients times the data-parallel scale, what ``clip_grad_norm_(model.parameters(),
        float("inf"))`` returns in the reference's loop (Multimodal_example_task2C.py:713): a device scalar, no sync."""
        self._bind()
        n = self._global_norm_sq().sqrt()[0] * abs(self.grad_scale)
        if self._scaler is not None and self._scaler.is_enabled():
            n = n / self._scaler._tensor(self._flat["P"].device)[0]
        return n

    _guard_ordinal = 0         # guarded update launches issued so far in the current step (mh_adam_step: guard_ordinal)

    def _next_ordinal(self, flag) -> int:
        """Position of the next guarded launch within the step, in issue (= stream) order; the step's accounting launch resets it.
        Inside a captured step the ordinals are constants of the graph, which is what they should be."""
        if flag is None:
            return 0
        self._guard_ordinal += 1
        return self._guard_ordinal

    def _launch_bucket(self, f, nrm, only, skip, flag=None):
        model = f["model"]
        n_shadow = model.layout.n_shadow if model is not None else 0
        runs = []
        for gi, a, b in f["runs"]:
            pieces = [(a, b)]
            if only is not None:
                pieces = [(max(a, only[0]), min(b, only[1]))]
            for sa, sb in (skip or []):
                nxt = []
                for x, y in pieces:
                    if sb <= x or sa >= y:
                        nxt.append((x, y))
                    else:
                        if x < sa:
                            nxt.append((x, sa))
                        if sb < y:
                            nxt.append((sb, y))
                pieces = nxt
            runs += [(gi, x, y) for x, y in pieces if y > x]
        table = self._word_table(f)
        if table is not None:       # carve the word-embedding table out of its run: it goes through the row-skipping kernel
            ta, tb, V, D, touched = table
            nxt = []
            for gi, a, b in runs:
                if a <= ta and tb <= b and float(self.param_groups[gi]["weight_decay"]) == 0.0:
                    if a < ta:
                        nxt.append((gi, a, ta))
                    ops.adam_step_rows(f["P"][ta:tb], f["M"][ta:tb], f["V"][ta:tb], f["G"][ta:tb], f["row_live"], touched, V, D,
                                       f["hyper"][gi], self.decoupled, nrm, float(self.max_grad_norm or 0.0), flag, self._next_ordinal(flag),
                                       self._clip_mult())
                    if tb < b:
                        nxt.append((gi, tb, b))
                else:
                    nxt.append((gi, a, b))
            runs = nxt
        for gi, a, b in runs:
            sh_n = max(0, min(b, n_shadow) - a)            # part of this run that has a 16-bit shadow
            shadow = model.flat_shadow[a:a + sh_n] if (model is not None and sh_n > 0) else None
            ops.adam_step(f["P"][a:b], f["M"][a:b], f["V"][a:b], f["G"][a:b], shadow, sh_n, f["hyper"][gi], self.decoupled,
                          nrm, float(self.max_grad_norm or 0.0), flag, self._next_ordinal(flag), self._clip_mult())
        if model is not None and only is None:
            model._shadow_synced()

    def _word_table(self, f=None):
        """(start, end, rows, D, touched-row bytes) of the word-embedding table in a bucket's flat buffer, or None."""
        f = self._flat if f is None else f
        model = f["model"]
        if not self.skip_untouched_rows or model is None:
            return None
        if "row_live" not in f:
            sp = model.layout.spec["bert.embeddings.word_embeddings.weight"]
            V, D = sp.shape
            if sp.offset + V * D > model.layout.n_shadow and sp.numel == V * D:
                f["table"] = (sp.offset, sp.offset + V * D, V, D)
                f["row_live"] = torch.zeros(V, dtype=torch.uint8, device=f["P"].device)
            else:
                f["table"], f["row_live"] = None, None
        if f["table"] is None:
            return None
        return f["table"] + (model._get_engine().word_row_live,)

    def state_dict(self):
        """Checkpoint of the optimizer: step count, the moment buffers (copies, flat over each parameter buffer; plain
        tensors for the usual single-model case, lists when the optimizer spans several buffers) and the per-group
        hyper-parameters."""
        self._bind()
        ms = [f["M"].clone() for f in self._buckets]
        vs = [f["V"].clone() for f in self._buckets]
        one = len(self._buckets) == 1
        return dict(step=self._step, skipped=int(self._flat["skip_state"][0]), exp_avg=ms[0] if one else ms, exp_avg_sq=vs[0] if one else vs,
                    param_groups=[{k: v for k, v in g.items() if k != "params"} for g in self.param_groups])

    def load_state_dict(self, state_dict):
        """Restore ``state_dict()``'s output: moments, step count (the bias corrections follow from it), group
        hyper-parameters; the word-embedding rows with optimizer history are rebuilt from the moments."""
        self._bind()
        for key in ("step", "exp_avg", "exp_avg_sq"):
            if key not in state_dict:
                raise KeyError(f"memehip.Adam.load_state_dict: missing {key!r} (expected the dict Adam.state_dict() returns)")
        ms, vs = state_dict["exp_avg"], state_dict["exp_avg_sq"]
        ms = ms if isinstance(ms, (list, tuple)) else [ms]
        vs = vs if isinstance(vs, (list, tuple)) else [vs]
        if len(ms) != len(self._buckets) or len(vs) != len(self._buckets):
            raise ValueError("optimizer state covers a different number of parameter buffers")
        for f, m, v in zip(self._buckets, ms, vs):
            if m.numel() != f["M"].numel() or v.numel() != f["V"].numel():
                raise ValueError(f"optimizer state has {m.numel()} elements, the flat buffer {f['M'].numel()}")
            f["M"].copy_(m.to(f["M"].device, F32).view(-1))
            f["V"].copy_(v.to(f["V"].device, F32).view(-1))
            table = self._word_table(f)
            if table is not None:       # rows whose moments are non-zero have history: they must keep being updated
                ta, tb, V, D, _ = table
                live = ((f["M"][ta:tb].view(V, D) != 0) | (f["V"][ta:tb].view(V, D) != 0)).any(dim=1)
                f["row_live"].copy_(live.to(torch.uint8))
        self._step = int(state_dict["step"])
        self._host_skipped = int(state_dict.get("skipped", 0))
        self._hyper_init = False          # the next step re-initialises the device-side bias corrections for step - skipped
        self._flat["skip_state"].copy_(torch.tensor([self._host_skipped, 0], dtype=torch.int32))
        groups = state_dict.get("param_groups")
        if groups is not None:
            if len(groups) != len(self.param_groups):
                raise ValueError("loaded state dict has a different number of parameter groups")
            for g, sg in zip(self.param_groups, groups):
                g.update({k: val for k, val in sg.items() if k != "params"})


class GradScaler:
    """``torch.cuda.amp.GradScaler``'s interface (the reference's fp16 branch: ``scaler.scale(loss).backward()``,
    ``scaler.step(optimizer)``, ``scaler.update()``, Multimodal_example_task2C.py:60-64,712-717) over a DEVICE-RESIDENT scale, so a
    step never synchronises with the host and a captured hipGraph replays it.

    The fp16 build carries a static power-of-two factor (``ModelConfig.grad_stream_scale``, 8192) on its 16-bit gradient streams
    inside the kernels; this object holds the DYNAMIC factor on top of it: ``scale(loss)`` multiplies the loss by it (the fused step
    reads it in the loss kernel), ``memehip.Adam`` divides it out again inside its update (``hyper[7]``), a step whose gradients were
    not finite is skipped / guarded and halves the factor, ``growth_interval`` clean steps double it -- the rule of
    ``GradScaler.update``.  ``init_scale`` and ``max_scale`` are relative to the static factor: the defaults 1 and 8 mean 8192 and
    65536 (torch's initial scale) in total.  With a torch optimizer ``step`` falls back to torch's own unscale-and-check (one host
    synchronisation per step, as torch.cuda.amp.GradScaler has)."""

    def __init__(self, init_scale: float = 1.0, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000,
                 enabled: bool = True, max_scale: float = 8.0, min_scale: float = 2.0 ** -14):
        if not (growth_factor >= 1.0 and 0.0 < backoff_factor <= 1.0 and growth_interval >= 1 and 0.0 < min_scale <= init_scale <= max_scale):
            raise ValueError("GradScaler: need growth_factor >= 1, 0 < backoff_factor <= 1, growth_interval >= 1, "
                             "0 < min_scale <= init_scale <= max_scale")
        self.init_scale, self.growth_factor, self.backoff_factor = float(init_scale), float(growth_factor), float(backoff_factor)
        self.growth_interval, self.max_scale, self.min_scale = int(growth_interval), float(max_scale), float(min_scale)
        self._enabled = bool(enabled)
        self._scale = None            # device f32[1]
        self._growth = None           # device i32[1]
        self._found = None            # torch-optimizer path: device f32[1]
        self._unscaled = set()        # torch-optimizer path: ids of optimizers whose gradients unscale_() has already divided this step
        self._optimizers = []         # memehip.Adam objects this scaler is attached to

    def _scale_written(self):
        """The scale word was written from the HOST (update(new_scale), load_state_dict): the attached optimizers' grad_scale /
        loss-scale word hyper[7] is otherwise only rewritten by the device-side accounting of a step, so Adam would divide by the old
        scale for one step -- make their next step re-initialise it."""
        for opt in self._optimizers:
            opt._hyper_init = False

    def is_enabled(self) -> bool:
        return self._enabled

    def _tensor(self, device) -> torch.Tensor:
        if self._scale is None:
            self._scale = torch.full((1,), self.init_scale, dtype=F32, device=device)
            self._growth = torch.zeros(1, dtype=torch.int32, device=device)
        return self._scale

    def attach_model(self, model: "MultimodalClassifier"):
        """The fused step (``forward_backward`` / ``GraphedStep``) computes the loss inside the model's launch plan: its loss kernel
        reads the model's loss-scale word, which becomes this scaler's tensor."""
        eng = model._get_engine()
        if self._scale is None:
            self._scale = eng.loss_scale
            self._scale.fill_(self.init_scale)
            self._growth = torch.zeros(1, dtype=torch.int32, device=self._scale.device)
        elif self._scale.data_ptr() != eng.loss_scale.data_ptr():
            raise ValueError("GradScaler.attach_model: this scaler already drives another model / optimizer")
        return self

    def scale(self, outputs):
        if not self._enabled:
            return outputs
        return outputs * self._tensor(outputs.device)[0]

    def unscale_(self, optimizer):
        """``torch.cuda.amp.GradScaler.unscale_``: divide the optimizer's gradients by the scale now (so that a ``clip_grad_norm_``
        between backward and step sees the true gradients) and remember it -- ``step`` must not unscale a second time.  A no-op for
        ``memehip.Adam``: its fused update divides the scale out itself and clips on the true norm (or, with
        ``clip_scaled_gradients``, on the scaled one)."""
        if not self._enabled or isinstance(optimizer, Adam):
            return
        if id(optimizer) in self._unscaled:
            raise RuntimeError("unscale_() has already been called on this optimizer since the last update().")
        grads = [p.grad for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
        if not grads:
            return
        sc = self._tensor(grads[0].device)
        if self._found is None:
            self._found = torch.zeros(1, dtype=F32, device=sc.device)
        torch._amp_foreach_non_finite_check_and_unscale_(grads, self._found, 1.0 / sc)
        self._unscaled.add(id(optimizer))

    def step(self, optimizer, *args, **kwargs):
        if not self._enabled:
            return optimizer.step(*args, **kwargs)
        if isinstance(optimizer, Adam):
            optimizer._attach_scaler(self)
            return optimizer.step(*args, **kwargs)
        if id(optimizer) not in self._unscaled:
            self.unscale_(optimizer)
        if self._found is None:
            return None
        if float(self._found) == 0.0:
            return optimizer.step(*args, **kwargs)
        return None

    def update(self, new_scale: Optional[float] = None):
        if not self._enabled:
            return
        self._unscaled.clear()
        if new_scale is not None:
            self._tensor(self._scale.device if self._scale is not None else "cuda").fill_(float(new_scale))
            self._growth.zero_()
            self._found = None
            self._scale_written()
        elif self._found is not None:          # torch-optimizer path; with memehip.Adam the step's accounting kernel has done it
            torch._amp_update_scale_(self._scale, self._growth, self._found, self.growth_factor, self.backoff_factor, self.growth_interval)
            self._scale.clamp_(self.min_scale, self.max_scale)
            self._found = None

    def get_scale(self) -> float:
        """The dynamic factor (synchronises).  Times the model's static ``stream_scale`` it is what torch's get_scale() reports."""
        return self.init_scale if self._scale is None else float(self._scale)

    def state_dict(self):
        return dict(scale=self.get_scale(), growth_tracker=0 if self._growth is None else int(self._growth), growth_factor=self.growth_factor,
                    backoff_factor=self.backoff_factor, growth_interval=self.growth_interval)

    def load_state_dict(self, sd):
        self.growth_factor, self.backoff_factor, self.growth_interval = float(sd["growth_factor"]), float(sd["backoff_factor"]), int(sd["growth_interval"])
        if self._scale is None:
            self.init_scale = float(sd["scale"])
        else:
            self._scale.fill_(float(sd["scale"]))
            self._growth.fill_(int(sd["growth_tracker"]))
            self._scale_written()


def flatten_parameters(module: nn.Module) -> nn.Module:
    """Re-home every parameter of ``module`` (a PyTorch head: Linear / BatchNorm layers ...) into ONE flat fp32 buffer
    with a mirrored gradient buffer, in place: ``p.data`` and ``p.grad`` become views, 16-byte aligned.  The fused
    ``memehip.Adam`` then updates the whole head in one launch and includes it in its global gradient norm.  Call it
    after ``module.to(device)``; gradients are accumulated into the views by autograd (``zero_grad()`` of memehip.Adam
    zeroes the buffer instead of dropping it)."""
    ps = [p for p in module.parameters() if p.requires_grad]
    if not ps:
        return module
    dev = ps[0].device
    offs, n = [], 0
    for p in ps:
        offs.append(n)
        n += (p.numel() + 3) // 4 * 4
    P = torch.zeros(n, dtype=F32, device=dev)
    G = torch.zeros(n, dtype=F32, device=dev)
    with torch.no_grad():
        for p, o in zip(ps, offs):
            P[o:o + p.numel()].copy_(p.detach().reshape(-1).to(F32))
            p.data = P[o:o + p.numel()].view(p.shape)
            p.grad = G[o:o + p.numel()].view(p.shape)
    module._memehip_flat = (P, G)
    _LOOSE[P.untyped_storage().data_ptr()] = G
    return module


# Every capture is THREAD-LOCAL: other threads may touch the runtime while this one captures.  The process group's watchdog thread
# queries the events of earlier collectives (parameter broadcast, the warm-up step's all-reduces) and a DataLoader's pin-memory
# thread allocates -- under the default "global" mode either one invalidates the capture (`bench.py --force-ddp` died in
# hipErrorStreamCaptureInvalidated at the first segment graph, round 3: the path every N > 1 run takes).
CAPTURE_MODE = "thread_local"


class GraphedStep:
    """Whole fine-tune step (forward, loss, backward, clip, Adam) captured into hipGraphs and
    replayed per batch, so the ~600 kernel launches of a step cost a handful of host calls.

    Single GPU: ONE graph.  Data parallel (``reducer`` given), ``ddp_mode``:

    * ``"segments"`` (default): one graph per backward segment (pairs of layers), the all-reduce issued between graph
      launches.  Every graph boundary joins all streams; measured 10.55 ms against 9.91 for the single graph at one rank.
    * ``"stream"``: the forward + loss is one graph; the backward is launched eagerly with exactly the single-GPU two-stream
      schedule (weight-gradient GEMMs on a side stream, no joins between layers), and as soon as a bucket's gradient slice
      is complete -- its segment's main-stream work and its weight-gradient event -- the all-reduce of that slice is issued
      behind a fence stream and, behind the all-reduce, that slice's Adam update.  The host issues a step in 5.1 ms, so it
      stays ahead of the GPU, but eagerly launched cross-stream edges resolve more slowly than a graph's (10-35 us gaps in
      the kernel trace): 10.82 ms at one rank.  Kept for multi-rank A/B.
    """

    def __init__(self, model: MultimodalClassifier, optimizer: Adam, batch: int, seq_len: int, use_graph: bool = True,
                 reducer=None, overlap_wgrad: bool = True, overlap_optimizer: bool = True, ddp_mode: Optional[str] = None,
                 scaler: Optional["GradScaler"] = None):
        self.model, self.opt = model, optimizer
        optimizer._model = model
        optimizer._resolve_skip(model)
        # fp16 storage: the dynamic loss scale is on by default (GradScaler's rule on the device, no host synchronisation);
        # ``scaler=GradScaler(enabled=False)`` keeps the static 8192 alone
        if scaler is None and model.config.compute_dtype == "fp16" and optimizer.skip_nonfinite:
            scaler = GradScaler()
        if scaler is not None and scaler.is_enabled():
            scaler.attach_model(model)
            optimizer._attach_scaler(scaler)
        self.scaler = scaler
        eng = model._get_engine()
        self.plan = eng.plan(batch, seq_len, True, gather_world=(reducer.world if reducer is not None else 0))
        if model.weights_changed():
            model.refresh_shadow()
        from . import ddp as _ddp
        _ddp.note_process_group()
        if use_graph and _ddp.communicator_was_destroyed() and os.environ.get("MEMEHIP_GRAPH_AFTER_PG_DESTROY") != "1":
            # the known crash configuration (ddp._DESTROYED): run the same plan with eager launches instead of replaying a hipGraph
            import warnings
            warnings.warn("memehip.GraphedStep: an RCCL process group has been destroyed in this process; hipGraph replay after a "
                          "communicator create / destroy cycle can fault inside hipGraphLaunch on this ROCm stack "
                          "(profiles/r04_segfault_record.md) -- falling back to eager launches of the same step "
                          "(MEMEHIP_GRAPH_AFTER_PG_DESTROY=1 overrides).  Create ONE communicator per process.", RuntimeWarning, stacklevel=2)
            use_graph = False
        self.use_graph = use_graph
        self.reducer = reducer
        if reducer is not None:
            optimizer.grad_scale = reducer.grad_scale
        self.graphs = None
        ddp_mode = ddp_mode or os.environ.get("MEMEHIP_DDP_MODE", "segments")
        if ddp_mode not in ("stream", "segments", "graph"):
            raise ValueError(f"ddp_mode must be 'stream', 'segments' or 'graph', got {ddp_mode!r}")
        # "graph": the "stream" schedule -- two-stream backward, every completed gradient slice all-reduced behind a fence stream, its
        # Adam slice behind the all-reduce -- captured WHOLE, collectives included, into ONE hipGraph (RCCL's launches are capturable;
        # torch's process group records them as nodes on its own stream, forked from and joined to the capture stream by events)
        self.ddp_graph = reducer is not None and ddp_mode == "graph"
        self.ddp_stream = reducer is not None and ddp_mode in ("stream", "graph")
        # weight-gradient GEMMs run on a second stream beside the LayerNorm / attention / dgrad chain
        # (a high-priority side stream was measured: 16.1 vs 9.79 ms for config 3, 92.0 vs 81.2 ms for config 5 -- as with the
        #  high-priority main stream of round 1, any non-default stream priority inside the graph loses badly on this stack)
        self.side = torch.cuda.Stream() if ((overlap_wgrad or overlap_optimizer) and (reducer is None or self.ddp_stream)) else None
        self.ddp_fence = torch.cuda.Stream() if self.ddp_stream else None
        self.wgrad_side = bool(overlap_wgrad)
        # optimizer-in-backward: the (HBM-bound) Adam update of a layer pair's matrices follows their weight-gradient
        # GEMMs on the side stream, under the (MFMA-bound) backward chain of the layers below; only the tail
        # (embeddings, biases, head) is updated after the backward.  Needs no global clip and a single GPU.
        # (with skip_nonfinite=True the slices go through the GUARDED update kernels; clipping and "strict" need the global norm first)
        self.opt_in_bwd = bool(overlap_optimizer and self.side is not None and reducer is None and optimizer.max_grad_norm is None
                               and optimizer.skip_nonfinite != "strict")
        # data parallel: the same idea behind the all-reduce -- as soon as a layer pair's gradient slice has been summed
        # over the ranks, its Adam update runs on a side stream under the backward of the layers below
        self.ddp_opt_in_bwd = bool(overlap_optimizer and reducer is not None and optimizer.max_grad_norm is None
                                   and optimizer.skip_nonfinite != "strict")
        self.ddp_side = torch.cuda.Stream() if self.ddp_opt_in_bwd else None
        self.ddp_wside = torch.cuda.Stream() if (reducer is not None and overlap_wgrad and not self.ddp_stream) else None
        self.closed = False
        from . import ddp as _ddp
        _ddp._LIVE_STEPS.add(self)

    def close(self):
        """Explicit teardown, in dependency order: wait for the device, reset the hipGraphs (last captured first; the data-parallel
        ones share ONE private memory pool, which is released with the last of them), drop the closures that keep this object in a
        reference cycle, then the side / fence streams.  With a reducer this has to happen while the process group is alive
        (``ddp.shutdown()`` closes steps, then reducers, then destroys the group).  Idempotent; ``step()`` raises afterwards."""
        if self.closed:
            return
        self.closed = True
        if self.reducer is not None and not self.reducer.closed:
            self.reducer.wait()
        torch.cuda.synchronize()
        for g, _, _ in reversed(self.graphs or []):
            if g is not None:
                g.reset()
        self.graphs = None
        self._tail = None
        self._opt_done = None
        self.side = self.ddp_fence = self.ddp_side = self.ddp_wside = None
        self.plan = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- pieces ------------------------------------------------------------------------------------------
    def _pieces(self):
        p = self.plan
        def fwd(stream):
            if self.side is None:
                p.fwd.run(stream)
            else:
                p.fwd.run2(torch.cuda.current_stream(), None, {})
            p.loss.run(stream)
        pieces = [("fwd", fwd, None)]
        if self.side is None and not self.ddp_stream:
            # data parallel: `ddp_group` consecutive layer segments share one hipGraph and one all-reduce bucket (their
            # gradient ranges are adjacent in the flat buffer): fewer graph boundaries, larger RCCL messages
            # (measured on a 1-rank RCCL group: 11.21 ms/step with one graph per layer, 10.99 in pairs, 10.97 in threes).
            # The last two layers keep their own buckets, so the all-reduce left exposed after the backward stays small.
            group = max(1, int(os.environ.get("MEMEHIP_DDP_GROUP", "2")))
            n_layer_segs = sum(1 for sg in p.bwd if sg.name.startswith("bwd_layer_"))
            seen = 0
            pend = []

            def flush():
                if not pend:
                    return
                segs = list(pend)
                pend.clear()
                rngs = [p.bucket_after[sg.name] for sg in segs if p.bucket_after.get(sg.name) is not None]
                rng = (min(r[0] for r in rngs), max(r[1] for r in rngs)) if rngs else None

                def run(stream, segs=segs):
                    if self.ddp_wside is None or len(segs) < 2:
                        for sg in segs:
                            sg.run(stream)
                        return
                    # inside a pair the first layer's weight-gradient GEMMs run on a side stream beside the second layer's
                    # chain (the buffers they read belong to the other layer parity); everything is joined at the end of
                    # the graph, before the pair's gradient slice goes to the all-reduce
                    main, events = torch.cuda.current_stream(), {}
                    for sg in segs:
                        sg.run2(main, self.ddp_wside, events)
                    for ev in events.values():
                        main.wait_event(ev)
                pieces.append((segs[0].name, run, rng))

            for seg in p.bwd:
                if seg.name.startswith("bwd_layer_"):
                    seen += 1
                if seg.name.startswith("bwd_layer_") and group > 1 and seen <= n_layer_segs - 2:
                    pend.append(seg)
                    if len(pend) == group:
                        flush()
                    continue
                flush()
                if seg.name == "bwd_embed_tables":      # needs every rank's token ids + embedding-gradient rows first
                    pieces.append(("gather", None, None))
                pieces.append((seg.name, seg.run, p.bucket_after.get(seg.name)))
            flush()
        else:
            def bwd(stream):
                main = torch.cuda.current_stream()
                events = {}
                done = []
                red = self.reducer if self.ddp_stream else None
                group = max(1, int(os.environ.get("MEMEHIP_DDP_GROUP", "2")))
                n_layer_segs = sum(1 for sg in p.bwd if sg.name.startswith("bwd_layer_"))
                pend, pend_ev, seen = [], [], 0

                def flush():
                    """all-reduce the union of the pending (adjacent) gradient ranges once everything that writes them has
                    been issued: the fence stream waits for the main stream's position and the segments' weight-gradient
                    events, the collective is enqueued behind the fence, the slice's Adam update behind the collective."""
                    if not pend:
                        return
                    rng = (min(r[0] for r in pend), max(r[1] for r in pend))
                    layer_rngs = list(pend)
                    self.ddp_fence.wait_stream(main)
                    for ev in pend_ev:
                        self.ddp_fence.wait_event(ev)
                    with torch.cuda.stream(self.ddp_fence):
                        works = red.reduce_range(rng)
                    if self.ddp_opt_in_bwd and pend_layer[0]:
                        with torch.cuda.stream(self.ddp_side):
                            for w in works:
                                w.wait()
                            for r in layer_rngs:
                                self.opt.launch(only=r, guarded=True)
                        done.extend(layer_rngs)
                    pend.clear(); pend_ev.clear(); pend_layer[0] = True

                pend_layer = [True]       # every pending range belongs to a layer segment (its Adam slice may run early)
                for seg in p.bwd:
                    if red is not None and seg.name == "bwd_embed_tables":      # needs every rank's ids + gradient rows first
                        red.gather(p.gather)
                    seg.run2(main, self.side if self.wgrad_side else None, events)
                    rng = p.bucket_after.get(seg.name)
                    is_layer = seg.name.startswith("bwd_layer_")
                    if red is not None:
                        if rng is not None:
                            if is_layer:
                                seen += 1
                            pend.append(rng)
                            pend_layer[0] = pend_layer[0] and is_layer
                            if seg.name in events:
                                pend_ev.append(events[seg.name])
                            # the last two layers (and everything that is not a layer) go out on their own, so the
                            # all-reduce left exposed after the backward stays small
                            if (not is_layer) or len(pend) >= group or seen > n_layer_segs - 2:
                                flush()
                        continue
                    if self.opt_in_bwd and is_layer and rng is not None:
                        if seg.name not in events:      # weight gradients ran on the main stream: fork behind them
                            e = torch.cuda.Event()
                            e.record(main)
                            self.side.wait_event(e)
                        with torch.cuda.stream(self.side):
                            self.opt.launch(only=rng, guarded=True)
                        ev = torch.cuda.Event()
                        ev.record(self.side)
                        events[seg.name] = ev
                        done.append(rng)
                if red is not None:
                    flush()
                for ev in events.values():      # join before the optimizer reads the remaining gradients
                    main.wait_event(ev)
                self._opt_done = done
            pieces.append(("bwd", bwd, None))
        if self.ddp_opt_in_bwd and not self.ddp_stream:
            self._opt_done = [p.bucket_after[seg.name] for seg in p.bwd
                              if seg.name.startswith("bwd_layer_") and p.bucket_after.get(seg.name) is not None]
        pieces.append(("opt", lambda stream: self.opt.launch(skip=getattr(self, "_opt_done", None),
                                                             guarded=self.opt_in_bwd or self.ddp_opt_in_bwd), None))
        return pieces

    def _after_segment(self, name, rng):
        """Data parallel: start the all-reduce of the gradient slice a segment completed and, behind it on the side
        stream, that slice's Adam update."""
        works = self.reducer.reduce_range(rng)
        if self.ddp_opt_in_bwd and rng is not None and name.startswith("bwd_layer_"):
            self.ddp_side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.ddp_side):
                for w in works:
                    w.wait()
                self.opt.launch(only=rng, guarded=True)

    def _run_eager(self):
        stream = torch.cuda.current_stream().cuda_stream
        for name, fn, rng in self._pieces():
            if name == "gather":
                self.reducer.gather(self.plan.gather)
                continue
            if name == "opt" and self.reducer is not None:
                self.reducer.wait()
            fn(stream)
            if self.reducer is not None and not self.ddp_stream:
                self._after_segment(name, rng)
        if self.ddp_side is not None:
            torch.cuda.current_stream().wait_stream(self.ddp_side)

    def load_batch(self, text, image, mask, labels):
        b = self.plan.buf
        b["ids"].copy_(text, non_blocking=True)
        b["mask"].copy_(mask, non_blocking=True)
        b["image"].copy_(image, non_blocking=True)
        b["labels"].copy_(labels, non_blocking=True)

    def step(self):
        """One step on the batch currently in the static input buffers. Returns (loss, n_correct) device tensors."""
        if self.closed:
            raise RuntimeError("GraphedStep is closed")
        opt = self.opt
        if opt._flat is None:
            self.model._attach_grads()
            opt._bind()
        if self.model.weights_changed():      # the master weights were edited outside the fused optimizer
            self.model.refresh_shadow()
        opt._step += 1
        opt._write_hyper()
        self.model._advance_rng(self.plan)
        self.model._get_engine().before_backward(self.plan)
        if not self.use_graph:
            self._run_eager()
        else:
            if self.graphs is None:
                self._capture()
            if self.ddp_graph:        # everything, the collectives included, is one graph
                self.graphs[0][0].replay()
                self.reducer.reduced_elems += self._per_replay[0]
                self.reducer.wire_bytes += self._per_replay[1]
                return self.plan.buf["loss"], self.plan.buf["ncorrect"]
            if self.ddp_stream:       # forward graph, then the backward + optimizer as stream-ordered eager launches
                self.graphs[0][0].replay()
                stream = torch.cuda.current_stream().cuda_stream
                for name, fn, rng in self._tail:
                    if name == "opt":
                        self.reducer.wait()
                    fn(stream)
                if self.ddp_side is not None:
                    torch.cuda.current_stream().wait_stream(self.ddp_side)
                return self.plan.buf["loss"], self.plan.buf["ncorrect"]
            for g, name, rng in self.graphs:
                if name == "gather":
                    self.reducer.gather(self.plan.gather)
                    continue
                if name == "opt" and self.reducer is not None:
                    self.reducer.wait()
                g.replay()
                if self.reducer is not None:
                    self._after_segment(name, rng)
            if self.ddp_side is not None:
                torch.cuda.current_stream().wait_stream(self.ddp_side)
        return self.plan.buf["loss"], self.plan.buf["ncorrect"]

    def _capture(self):
        # one warm-up run on a side stream (lazy module loads, workspace allocation), undone afterwards so
        # it does not count as a training step; then capture the same launches
        f = self.opt._flat
        snap = (f["P"].clone(), f["M"].clone(), f["V"].clone())
        acct = [f["skip_state"], f["overflow"]] + list(f["hyper"])        # the warm-up's accounting launch must not count as a step
        if self.opt._scaler is not None and self.opt._scaler._scale is not None:
            acct += [self.opt._scaler._scale, self.opt._scaler._growth]
        acct_snap = [t.clone() for t in acct]
        counters = (self.reducer.reduced_elems, self.reducer.wire_bytes) if self.reducer is not None else None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            stream = s.cuda_stream
            for name, fn, rng in self._pieces():
                if name == "gather":
                    self.reducer.gather(self.plan.gather)
                else:
                    if name == "opt" and self.ddp_stream:
                        self.reducer.wait()
                    fn(stream)
            if self.ddp_side is not None and self.ddp_stream:
                s.wait_stream(self.ddp_side)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        f["P"].copy_(snap[0]); f["M"].copy_(snap[1]); f["V"].copy_(snap[2])
        for t, c in zip(acct, acct_snap):
            t.copy_(c)
        if counters is not None:      # (the stream schedule's warm-up ran the collectives for real)
            self.reducer.reduced_elems, self.reducer.wire_bytes = counters
        self.model.refresh_shadow()
        # (the warm-up left prev_ids == ids, so the first real step re-zeroes exactly the embedding-gradient
        #  rows the warm-up wrote)
        pieces = self._pieces()
        graphs = []
        if self.ddp_graph:
            e0, w0 = self.reducer.reduced_elems, self.reducer.wire_bytes
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                stream = torch.cuda.current_stream().cuda_stream
                for name, fn, rng in pieces:
                    if name == "opt":
                        self.reducer.wait()
                    fn(stream)
                if self.ddp_side is not None:
                    torch.cuda.current_stream().wait_stream(self.ddp_side)
            graphs.append((g, "step", None))
            self._per_replay = (self.reducer.reduced_elems - e0, self.reducer.wire_bytes - w0)      # the host-side counters of one step
            self.reducer.reduced_elems, self.reducer.wire_bytes = e0, w0
        elif self.ddp_stream:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                pieces[0][1](torch.cuda.current_stream().cuda_stream)
            graphs.append((g, "fwd", None))
            self._tail = pieces[1:]
        elif self.reducer is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                stream = torch.cuda.current_stream().cuda_stream
                for name, fn, rng in pieces:
                    fn(stream)
            graphs.append((g, "step", None))
        else:
            pool = None
            for name, fn, rng in pieces:
                if name == "gather":
                    graphs.append((None, "gather", None))
                    continue
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode=CAPTURE_MODE):
                    fn(torch.cuda.current_stream().cuda_stream)
                pool = g.pool()
                graphs.append((g, name, rng))
        self.graphs = graphs
