"""Kevin's 2C path RUN from the reference's own source (see oracle/gen_ref_hotpath.py for the contract; build container only).

TEST INFRASTRUCTURE.  ``gen_kevin()`` executes, out of the AST of /root/reference/example_scripts/Multimodal_example_task2C.py,
``MultimodalDataset`` (:208-304), ``LLMWithClassificationHead`` (:307-392), ``MCA3`` / ``ConcatAttention3`` (:423-499),
``CustomDenseNet161`` (:562-585), ``MultimodalClassifier`` (:587-685) and ``train`` / ``test`` / ``evaluate`` (:688-879) with the
script's globals set as ``setup()`` sets them (:59-174; ``USE_FP16 = False``: the fp32 branch, clip at 10.0) and writes
tests/golden/ref_kevin_2c.npz.

One epoch over 24 memes in 4 batches of 6: ``check_interval = total_batches // 2 = 2`` (:696), so the reference's own mid-epoch
``test()`` calls (:755-769) happen after batches 2 and 4 -- and, because ``test()`` leaves the model in eval mode (:780) and
``train()`` never switches back, batches 3 and 4 are TRAINED IN EVAL MODE (BatchNorm on running statistics, dropout off).  The
fixture records that: ``train_mode_flags`` = [1, 1, 0, 0].
"""
from __future__ import annotations

import contextlib
import json
import os
import random
import tempfile

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ref_env as E
from .gen_ref_hotpath import REF_PY, GOLDEN, _identity, set_dropout, calibrate_bn

TEXT_NAME, ENG_NAME, IMAGE_NAME = "aubmindlab/bert-base-arabertv2", "roberta-base", "vit_small_patch16_224"
NAMES = ("MultimodalDataset", "LLMWithClassificationHead", "MCA3", "ConcatAttention3", "CustomDenseNet161", "MultimodalClassifier",
         "train", "test", "evaluate")


def kevin_namespace(workdir: str, cfg=E.KEVIN, fp16: bool = False):
    import pandas as pd
    from PIL import Image
    from sklearn.metrics import auc, f1_score, roc_curve
    from torch.utils.data import DataLoader, Dataset
    from transformers import get_linear_schedule_with_warmup
    recs, caps = E.records24(), E.captions24()
    tok_ar = E.EncodePlusTokenizer([r["text"] for r in recs], workdir, "vocab_ar")
    tok_en = E.EncodePlusTokenizer(caps, workdir, "vocab_en")
    state = E.kevin_state(tok_ar.vocab_size, tok_en.vocab_size, cfg)

    def sub(prefix):
        return {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}

    class AutoTokenizer:
        @staticmethod
        def from_pretrained(name, *a, **k):
            return {TEXT_NAME: tok_ar, ENG_NAME: tok_en}[name]

    class AutoModel:
        @staticmethod
        def from_pretrained(name, *a, **k):
            if name == TEXT_NAME:
                m = E.local_bert(tok_ar.vocab_size, cfg["text_layers"], dropout=0.1)
                E.load_text_state(m, sub("text_model.model."))
            else:
                m = E.local_bert(tok_en.vocab_size, cfg["caption_layers"], dropout=0.1)
                E.load_text_state(m, sub("caption_text_model.model."))
            return m

    class timm:
        @staticmethod
        def create_model(name, pretrained=False, **k):
            m = E.TimmViT(**cfg["vit"])
            m.load_oracle_state(sub("image_model.image_model."))
            return m

    class ImageCaptioning:          # the BLIP captioner (:195-206) is outside the hot path: its output is an input here
        def __init__(self):
            self.pos = 0

        def generate_caption(self, images, texts):
            out = caps[self.pos:self.pos + len(images)]
            self.pos += len(images)
            return out

    ns = dict(torch=torch, nn=nn, optim=torch.optim, F=F, np=np, pd=pd, json=json, random=random, tqdm=_identity, Image=Image,
              f1_score=f1_score, roc_curve=roc_curve, auc=auc, DataLoader=DataLoader, Dataset=Dataset, timm=timm, transforms=E.transforms,
              sigmoid_focal_loss=E.focal_standin, get_linear_schedule_with_warmup=get_linear_schedule_with_warmup, AutoModel=AutoModel,
              AutoTokenizer=AutoTokenizer, ImageCaptioning=ImageCaptioning,
              # setup()'s globals (:59-86).  fp16=True: the script's DEFAULT (USE_FP16 = True, :60); on this CPU `autocast()` is a null
              # context (the arithmetic stays fp32: what is pinned is the order clip -> unscale of :712-717, not half precision)
              USE_FP16=bool(fp16), autocast=contextlib.nullcontext, scaler=None, fold=0, text_model=TEXT_NAME, english_text_model=ENG_NAME, image_model=IMAGE_NAME,
              fusion_method="concatenation", train_max_seq_len=cfg["seq_len"], best_macro_f1=0.0, device=torch.device("cpu"))
    E.extract(REF_PY, NAMES, ns)
    return ns, tok_ar, tok_en, state


def collate(items):
    out = {}
    for k in items[0]:
        vals = [it[k] for it in items]
        out[k] = torch.stack(vals) if isinstance(vals[0], torch.Tensor) else vals
    return out


def ref_key(n: str, vit) -> str:
    """stand-in parameter path -> the key used in ref_env.kevin_state (ViT in oracle naming)."""
    pfx = "image_model.image_model.vit."
    if n.startswith(pfx):
        return None
    return n


def gen_kevin(cfg=E.KEVIN, fp16: bool = False):
    """fp16=False: the fp32 branch (clip at 10.0 on the true gradients) -> ref_kevin_2c.npz.  fp16=True: the reference's DEFAULT branch
    (:701-717: scaler.scale(loss).backward(), clip_grad_norm_ at 1.0 on the SCALED gradients, scaler.step, scaler.update) with
    torch.amp.GradScaler("cpu", init_scale=65536) -> ref_kevin_2c_fp16.npz (+ the scale after every step, both clip calls' norms)."""
    import pandas as pd
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as work:
        E.write_dataset(work)
        os.chdir(work)
        try:
            ns, tok_ar, tok_en, state = kevin_namespace(work, cfg, fp16)
            recs = E.records24()
            l2id = {"not_propaganda": 0, "propaganda": 1}
            df = pd.DataFrame({"id": [r["id"] for r in recs], "text": [r["text"] for r in recs], "image": [r["img_path"] for r in recs],
                               "label": [l2id[r["class_label"]] for r in recs]})
            E.AUG_GEN.manual_seed(cfg["aug_seed"])
            ds = ns["MultimodalDataset"](df["id"], df["text"], df["image"], df["label"])              # :141-143
            assert list(ds.precalculated_captions) == E.captions24()
            E.AUG_GEN.manual_seed(cfg["aug_seed"])
            items = [ds[i] for i in range(len(ds))]
            out["ds_keys"] = np.array(sorted(items[0]))
            for k in ("text", "text_mask", "caption_text", "caption_text_mask", "label"):
                out["ds_" + k] = torch.stack([it[k] for it in items]).numpy()
            imgs = torch.stack([it["image"] for it in items])
            again = E.kevin_images(work, cfg["aug_seed"])
            assert torch.equal(imgs, again), "ref_env.kevin_images must reproduce the reference Dataset's image tensors"
            out["ds_image_checksum"] = np.stack([[float(im.double().sum()), float(im.double().abs().sum())] for im in imgs])
            out["vocab_sizes"] = np.array([tok_ar.vocab_size, tok_en.vocab_size])
            B = cfg["batch"]
            batches = [collate(items[i:i + B]) for i in range(0, len(items), B)]
            loader = E.ListLoader(batches)
            # ---- :164-174
            device = torch.device("cpu")
            model = ns["MultimodalClassifier"](fusion_method="concatenation")
            heads = {k: v for k, v in state.items() if not k.startswith(("text_model.model.", "caption_text_model.model.", "image_model.image_model."))}
            res = model.load_state_dict(heads, strict=False)
            assert not res.unexpected_keys, res.unexpected_keys[:4]
            left = [k for k in res.missing_keys if not k.startswith(("text_model.model.", "caption_text_model.model.", "image_model.image_model."))
                    and "running" not in k and "num_batches" not in k]
            assert not left, left[:4]
            model.to(device)
            out["dropout_as_constructed"] = np.array(sorted({f"{n.split('.')[-1] if 'model.' in n else n}={m.p}" for n, m in model.named_modules()
                                                             if isinstance(m, nn.Dropout)}))
            # BatchNorm running statistics of a "partly trained" checkpoint: part of the initial state of both sides
            stats = calibrate_bn(model, lambda: [model(b["text"], b["image"], b["text_mask"], b["caption_text"], b["caption_text_mask"])
                                                 for b in batches])
            out["init_bn_names"] = np.array(sorted(stats))
            out["init_bn_sizes"] = np.array([stats[n][0].numel() for n in sorted(stats)])
            out["init_bn_running_mean"] = np.concatenate([stats[n][0].numpy() for n in sorted(stats)])
            out["init_bn_running_var"] = np.concatenate([stats[n][1].numpy() for n in sorted(stats)])
            criterion = ns["sigmoid_focal_loss"]
            lr = cfg["lr"]
            optimizer = torch.optim.Adam(model.get_params(lr))
            groups = optimizer.param_groups
            out["group_sizes"] = np.array([sum(p.numel() for p in g["params"]) for g in groups])
            out["group_lrs"] = np.array([g["lr"] for g in groups])
            num_epochs = 2
            total_steps = len(loader) * num_epochs
            warmup_steps = 2          # the script: int(0.1 * total_steps) (:171); 2 of 8 here so that warm-up AND decay are both stepped through
            scheduler = ns["get_linear_schedule_with_warmup"](optimizer, num_warmup_steps=warmup_steps, num_training_steps=total_steps)
            ns.update(model=model, criterion=criterion, optimizer=optimizer, scheduler=scheduler, test_df=loader, val_df=loader,
                      train_df=loader, num_epochs=num_epochs, total_steps=total_steps, warmup_steps=warmup_steps)
            # ---- instrumentation: forward outputs of the train loop, what the nested test() / evaluate() calls return
            phase = {"name": "train"}
            fwd = []
            model.register_forward_hook(lambda m, a, o: fwd.append((phase["name"], bool(m.training), o.detach().clone())))
            mid_tests, n_eval = [], [0]
            ref_test, ref_evaluate = ns["test"], ns["evaluate"]

            def test_wrapper(*a, **k):
                prev, phase["name"] = phase["name"], "test"
                r = ref_test(*a, **k)
                phase["name"] = prev
                mid_tests.append([float(x) for x in r])
                return r

            def evaluate_wrapper(*a, **k):
                prev, phase["name"] = phase["name"], "evaluate"
                r = ref_evaluate(*a, **k)
                phase["name"] = prev
                n_eval[0] += 1
                return r
            ns["test"], ns["evaluate"] = test_wrapper, evaluate_wrapper
            # ---- test() + evaluate() on the initial state (eval mode, BatchNorm on the checkpoint's running statistics)
            r0 = ref_test(model, loader, criterion, device, 0)
            out["initial_test"] = np.array([float(x) for x in r0])
            out["initial_test_outputs"] = torch.stack([o for ph, t, o in fwd]).numpy()
            fwd.clear()
            ref_evaluate(model, loader, r0[3], device)
            out["initial_evaluate_outputs"] = torch.stack([o for ph, t, o in fwd]).numpy()
            fwd.clear()
            out["initial_evaluate_tsv"] = np.array(open("task2C_kevinmathew.tsv", encoding="utf-8").read().split("\n"))
            out["initial_evaluate_probs_tsv"] = np.array(open("task2C_kevinmathew_probs_fold_0.tsv", encoding="utf-8").read().split("\n"))
            first, lrs, clip_norms = {}, [], []
            real_clip = torch.nn.utils.clip_grad_norm_

            def clip_spy(params, max_norm, *a, **k):          # what the reference's two clip calls return (:728-730): the total norm
                r = real_clip(params, max_norm, *a, **k)
                clip_norms.append((float(max_norm), float(r)))
                return r
            torch.nn.utils.clip_grad_norm_ = clip_spy

            def grab(opt, args, kwargs):
                lrs.append([g["lr"] for g in opt.param_groups])
                if not first:
                    first.update({n: p.grad.detach().clone() for n, p in model.named_parameters()})
            optimizer.register_step_pre_hook(grab)
            saved = set_dropout(model, 0.0)
            ref_scaler = torch.amp.GradScaler("cpu", init_scale=65536.0) if fp16 else None          # :61-62 GradScaler(): torch's default initial scale
            scales = []
            if fp16:
                real_update = ref_scaler.update

                def update_spy(*a, **k):
                    r = real_update(*a, **k)
                    scales.append(float(ref_scaler.get_scale()))
                    return r
                ref_scaler.update = update_spy
            train_loss, acc = ns["train"](model, loader, criterion, optimizer, scheduler, device, 0, ref_scaler)        # :178-180
            set_dropout(model, saved=saved)
            torch.nn.utils.clip_grad_norm_ = real_clip
            out["grad_norm_before_clip"] = np.array([n for mx, n in clip_norms if mx == float("inf")])
            out["clip_max_norms"] = np.array(sorted({mx for mx, n in clip_norms if mx != float("inf")}))
            out["scaler_scale_after_step"] = np.array(scales)
            tr = [(t, o) for ph, t, o in fwd if ph == "train"]
            out["mid_epoch_test_outputs"] = torch.stack([o for ph, t, o in fwd if ph == "test"]).numpy()      # 4 test() calls x 4 batches
            out["train_outputs"] = torch.stack([o for _, o in tr]).numpy()
            out["train_mode_flags"] = np.array([int(t) for t, _ in tr])
            out["train_loss"], out["train_acc"] = np.array(train_loss), np.array(acc)
            out["mid_epoch_tests"] = np.array(mid_tests)                 # rows: (loss, accuracy, macro_f1, threshold), test_df then val_df
            out["mid_epoch_evaluate_calls"] = np.array(n_eval[0])
            out["step_lrs"] = np.array(lrs)
            out["best_macro_f1_after"] = np.array(ns["best_macro_f1"])
            fwd.clear()
            # ---- parameters: step-1 gradients (after the clip call, as the optimizer saw them), values after the epoch
            vit = model.image_model.image_model
            vit_names = [k[len("image_model.image_model."):] for k in state if k.startswith("image_model.image_model.")]
            inv = {"image_model.image_model.vit." + vit._map(k): "image_model.image_model." + k for k in vit_names}
            names, gn, gs, fin, dl = [], [], [], [], []
            for n, p in model.named_parameters():
                key = inv.get(n, n)
                assert key in state, (n, key)
                names.append(key)
                g0 = first[n]
                gn.append(float(g0.double().norm())); gs.append(E.sample_of(g0)); fin.append(E.sample_of(p))
                dl.append(float((p.detach() - state[key].reshape(p.shape)).double().norm()))
            out["param_names"] = np.array(names)
            out["grad_norms_step1"], out["grad_samples_step1"] = np.array(gn), np.stack(gs)
            out["global_grad_norm_step1"] = np.array(float(torch.sqrt(sum(g.double().pow(2).sum() for g in first.values()))))
            out["param_samples_after"], out["param_delta_norm_after"] = np.stack(fin), np.array(dl)
            sd = model.state_dict()
            for k in ("text_fc.1", "caption_text_fc.1", "fusion_layer.attention_layer.1", "fusion_layer.reduce.1", "output_fc.1"):
                out[f"bn_{k}_running_mean_after"] = sd[k + ".running_mean"].numpy()
                out[f"bn_{k}_running_var_after"] = sd[k + ".running_var"].numpy()
                out[f"bn_{k}_num_batches_tracked_after"] = sd[k + ".num_batches_tracked"].numpy()
            # ---- a last test() + evaluate() as setup() runs them after the epoch (:181) and on a new best F1 (:769)
            r = ns["test"](model, loader, criterion, device, 0)
            out["final_test"] = np.array([float(x) for x in r])
            out["final_test_outputs"] = torch.stack([o for ph, t, o in fwd if ph == "test"]).numpy()
            fwd.clear()
            ns["evaluate"](model, loader, r[3], device)
            out["final_evaluate_outputs"] = torch.stack([o for ph, t, o in fwd if ph == "evaluate"]).numpy()
            out["evaluate_tsv"] = np.array(open("task2C_kevinmathew.tsv", encoding="utf-8").read().split("\n"))
            out["evaluate_probs_tsv"] = np.array(open("task2C_kevinmathew_probs_fold_0.tsv", encoding="utf-8").read().split("\n"))
        finally:
            os.chdir(cwd)
    out["notes"] = np.array(("DEFAULT branch (USE_FP16=True: clip 1.0 on the SCALED gradients, then scaler.step; torch.amp.GradScaler('cpu', 65536); "
                             "autocast = null context, fp32 arithmetic; grad_*_step1 are the gradients the optimizer saw: scaled, clipped, unscaled)" if fp16 else
                             "fp32 branch (USE_FP16=False, clip 10.0)") + "; seq_len=128 (script: 512); batch 6 x 4; ListLoader over items read once "
                            "from the reference Dataset (AUG_GEN seeded); train pass with every nn.Dropout p set to 0; warmup_steps=2 of 8; "
                            "BERT 4 layers + caption BERT 2 layers + ViT(512 wide, 8 heads, 4 layers); cpu fp32")
    for k, v in cfg.items():
        if not isinstance(v, dict):
            out["cfg_" + k] = np.array(v)
    path = os.path.join(GOLDEN, "ref_kevin_2c_fp16.npz" if fp16 else "ref_kevin_2c.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays); train loss {train_loss:.6f} acc {acc:.4f}")
    print("train outputs:", out["train_outputs"], "modes", out["train_mode_flags"], "\nmid tests", out["mid_epoch_tests"], "\nfinal", out["final_test"])


if __name__ == "__main__":
    import sys
    gen_kevin(fp16="--fp16" in sys.argv)
