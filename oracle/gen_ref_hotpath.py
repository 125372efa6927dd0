"""Fixtures made by RUNNING the reference's own hot path (build container only): ``python -m oracle.gen_ref_hotpath``.

TEST INFRASTRUCTURE.  Reads /root/reference/example_scripts/Multimodal_example_task2C.txt and ...task2C.py as text, executes the
definitions named below out of their ASTs (oracle/ref_env.py: ``extract`` and the table of what every third-party name is bound
to) and stores what they computed -- tensors, numbers and the TSV lines they wrote -- in tests/golden/ref_organizers_2c.npz and
tests/golden/ref_kevin_2c.npz.  No reference source text is stored.

Organizers (``Multimodal_example_task2C.txt``): ``read_data`` (:88-104), ``MultimodalDataset`` (:28-71), ``MultimodalClassifier``
(:152-197), ``train`` / ``test`` (:200-242), ``evaluate`` (:259-280) are run as the script's own module-level code runs them
(:109-115, :245-256, :282): DistilBERT (6 x 768, as ``distilbert-base-multilingual-cased``) + ResNet-50 (3, 4, 6, 3) at 224 x 224,
batch 8, ``nn.CrossEntropyLoss`` + ``optim.Adam(lr=2e-5)``, one epoch over 24 memes, then ``test`` and ``evaluate``.

Kevin (``Multimodal_example_task2C.py``): ``MultimodalDataset`` (:208-304), ``LLMWithClassificationHead`` (:307-392),
``ConcatAttention3`` (:476-499), ``CustomDenseNet161`` (:562-585), ``MultimodalClassifier`` (:587-685), ``train`` / ``test`` /
``evaluate`` (:688-879) with the fp32 branch (``USE_FP16 = False``): BERT text tower + ViT image tower (timm stand-in) + caption
BERT, sigmoid focal loss, ``optim.Adam(model.get_params(lr))``, ``get_linear_schedule_with_warmup``, clip at 10.

Deviations from what a user of the reference would run, all forced and all stated in the fixture (``notes``): sequence length 128
(the scripts set 512; a module-level hyper-parameter), ``shuffle=False``, and the dropout probabilities of the TRAINING passes set
to 0 on the constructed model (the random masks of two implementations cannot agree; the HIP path's dropout is pinned by
tests/test_dropout_gpu.py) -- evaluation passes run with the reference's dropout modules as constructed.
"""
from __future__ import annotations

import csv
import json
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ref_env as E
from . import resnet_oracle as R

REF_TXT = "/root/reference/example_scripts/Multimodal_example_task2C.txt"
REF_PY = "/root/reference/example_scripts/Multimodal_example_task2C.py"
GOLDEN = E.GOLDEN

ORG = dict(seq_len=128, batch=8, text_layers=6, resnet_layers=(3, 4, 6, 3), seed=31, lr=2e-5)


def _identity(x, *a, **k):
    return x


def organizers_namespace(workdir: str, cfg=ORG):
    """The names the organizers' definitions look up at run time, bound as oracle/ref_env.py's table says."""
    import pandas as pd
    from PIL import Image
    from torch.utils.data import DataLoader, Dataset
    recs = E.records24()
    tok = E.EncodePlusTokenizer([r["text"] for r in recs], workdir, "vocab_ar")
    state = E.organizers_state(tok.vocab_size, cfg["text_layers"], cfg["resnet_layers"], cfg["seed"])

    class AutoTokenizer:
        @staticmethod
        def from_pretrained(name, *a, **k):
            return tok

    class AutoModel:
        @staticmethod
        def from_pretrained(name, *a, **k):
            m = E.local_distilbert(tok.vocab_size, cfg["text_layers"], dropout=0.1)          # the checkpoint's config: 0.1 / 0.1
            E.load_text_state(m, {k_[len("bert."):]: v for k_, v in state.items() if k_.startswith("bert.")})
            return m

    class models:
        @staticmethod
        def resnet50(pretrained=True, *a, **k):
            m = E.TorchvisionResNet50(cfg["resnet_layers"])
            m.load_tv_state({k_[len("resnet."):]: v for k_, v in state.items() if k_.startswith("resnet.")})
            return m

    ns = dict(torch=torch, nn=nn, optim=torch.optim, np=np, json=json, pd=pd, csv=csv, Image=Image, transforms=E.transforms,
              Dataset=Dataset, DataLoader=DataLoader, AutoTokenizer=AutoTokenizer, AutoModel=AutoModel, models=models, tqdm=_identity,
              train_max_seq_len=cfg["seq_len"], text_model_name="distilbert-base-multilingual-cased")
    E.extract(REF_TXT, ("MultimodalDataset", "read_data", "MultimodalClassifier", "train", "test", "evaluate"), ns)
    return ns, tok, state


def set_dropout(model: nn.Module, p=None, saved=None):
    """p given: remember every dropout probability and set it to p; ``saved`` given: restore."""
    if saved is not None:
        for m, q in saved:
            m.p = q
        return None
    mem = []
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            mem.append((m, m.p))
            m.p = p
    return mem


def calibrate_bn(model: nn.Module, run) -> dict:
    """Gives the BatchNorm layers the running statistics a pretrained checkpoint would carry: the cumulative average of the batch
    statistics over ``run()`` (train mode, no gradient, dropout off); momentum and the batch counter are put back afterwards.
    Returns {module path: (running_mean, running_var)} -- stored in the fixture, part of the initial state of both sides."""
    bns = {n: m for n, m in model.named_modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)}
    mom = {n: m.momentum for n, m in bns.items()}
    for m in bns.values():
        m.reset_running_stats()
        m.momentum = None
    was = model.training
    model.train()
    saved = set_dropout(model, 0.0)
    with torch.no_grad():
        run()
    set_dropout(model, saved=saved)
    model.train(was)
    for n, m in bns.items():
        m.momentum = mom[n]
        m.num_batches_tracked.zero_()
    return {n: (m.running_mean.clone(), m.running_var.clone()) for n, m in bns.items()}


def gen_organizers(cfg=ORG):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as work:
        json_path = E.write_dataset(work)
        os.chdir(work)                                  # the reference opens obj['img_path'] as given (relative paths)
        try:
            ns, tok, state = organizers_namespace(work, cfg)
            l2id = {"not_propaganda": 0, "propaganda": 1}                    # ...task2C.txt:107
            # ---- ...task2C.txt:109-115: read_data -> label map -> MultimodalDataset
            df = ns["read_data"](json_path)
            df["label"] = df["label"].map(l2id)
            ds = ns["MultimodalDataset"](df["id"], df["text"], df["image"], df["label"])
            items = [ds[i] for i in range(len(ds))]
            assert sorted(items[0]) == ["id", "image", "label", "text", "text_mask"]
            out["ds_keys"] = np.array(sorted(items[0]))
            out["ds_ids"] = np.array([it["id"] for it in items])
            out["ds_text"] = torch.stack([it["text"] for it in items]).numpy()
            out["ds_text_mask"] = torch.stack([it["text_mask"] for it in items]).numpy()
            out["ds_label"] = torch.stack([it["label"] for it in items]).numpy()
            imgs = torch.stack([it["image"] for it in items])
            out["ds_image_checksum"] = np.stack([[float(im.double().sum()), float(im.double().abs().sum())] for im in imgs])
            out["ds_image_patch"] = imgs[:, :, 100:108, 100:108].numpy()
            ds_t = ns["MultimodalDataset"](df["id"], df["text"], df["image"], df["label"], is_test=True)
            out["ds_test_keys"] = np.array(sorted(ds_t[0]))
            df_t = ns["read_data"](json_path, is_test=True)
            out["read_data_columns"] = np.array(list(df.columns))
            out["read_data_test_columns"] = np.array(list(df_t.columns))
            out["vocab_size"] = np.array(tok.vocab_size)
            # ---- ...task2C.txt:142 (shuffle off), :245-249
            loader = ns["DataLoader"](ds, batch_size=cfg["batch"], shuffle=False, drop_last=True)
            device = torch.device("cpu")
            model = ns["MultimodalClassifier"](num_classes=2)
            missing = model.load_state_dict({k: v for k, v in state.items() if k.split(".")[0].endswith("_fc")}, strict=False)
            assert not [k for k in missing.missing_keys if k.split(".")[0].endswith("_fc")]
            model.to(device)
            drops = {n: m.p for n, m in model.named_modules() if isinstance(m, nn.Dropout)}
            out["dropout_as_constructed"] = np.array([f"{n}={p}" for n, p in sorted(drops.items())])
            criterion = nn.CrossEntropyLoss()
            optimizer = torch.optim.Adam(model.parameters(), lr=cfg["lr"])
            # the running statistics of a "pretrained" ResNet-50: part of the initial state (torchvision key names)
            stats = calibrate_bn(model, lambda: [model(b["text"], b["image"], b["text_mask"]) for b in loader])
            tv_bn = [k[: -len(".weight")] for k, v in R.resnet_param_shapes(cfg["resnet_layers"], 64, 1000).items()
                     if len(v) == 1 and k.endswith(".weight") and not k.startswith("fc.")]
            hf_of = {"resnet.body." + E._resnet_hf_name(k + ".weight")[: -len(".weight")]: k for k in tv_bn}
            assert sorted(hf_of) == sorted(stats), (sorted(hf_of)[:3], sorted(stats)[:3])
            out["init_bn_names"] = np.array([hf_of[n] for n in sorted(stats)])
            out["init_bn_sizes"] = np.array([stats[n][0].numel() for n in sorted(stats)])
            out["init_bn_running_mean"] = np.concatenate([stats[n][0].numpy() for n in sorted(stats)])
            out["init_bn_running_var"] = np.concatenate([stats[n][1].numpy() for n in sorted(stats)])
            seen = []
            hook = model.register_forward_hook(lambda m, a, o: seen.append(o.detach().clone()))
            first = {}

            def grab(opt, args, kwargs):
                if not first:
                    first.update({n: p.grad.detach().clone() for n, p in model.named_parameters()})
            optimizer.register_step_pre_hook(grab)
            # ---- ...task2C.txt:252-256: one epoch of the reference's train()
            saved = set_dropout(model, 0.0)
            train_loss, acc = ns["train"](model, loader, criterion, optimizer, device)
            set_dropout(model, saved=saved)
            out["train_logits"] = torch.stack(seen).numpy()
            out["train_loss"], out["train_acc"] = np.array(train_loss), np.array(acc)
            seen.clear()

            def ref_name(n):            # stand-in module name -> the reference model's state_dict key
                if n.startswith("resnet.body."):
                    return None
                return n
            named = {n: p for n, p in model.named_parameters() if ref_name(n)}
            tv_names = list(R.resnet_param_shapes(cfg["resnet_layers"], 64, 1000))
            tv_now = model.resnet.tv_state_dict(tv_names)
            inv = {model.resnet._map(k): k for k in tv_names}
            names, grads, finals, deltas = [], [], [], []
            for n, p in model.named_parameters():
                key = ("resnet." + inv[n[len("resnet."):]]) if n.startswith("resnet.") else n
                names.append(key)
                g0 = first[n]
                grads.append((float(g0.double().norm()), E.sample_of(g0)))
                finals.append(E.sample_of(p))
                deltas.append(float((p.detach() - state[key]).double().norm()))
            out["param_names"] = np.array(names)
            out["grad_norms_step1"] = np.array([g[0] for g in grads])
            out["grad_samples_step1"] = np.stack([g[1] for g in grads])
            out["param_samples_after"] = np.stack(finals)
            out["param_delta_norm_after"] = np.array(deltas)
            sd = model.resnet.body.state_dict()
            out["bn1_running_mean_after"] = sd["embedder.embedder.normalization.running_mean"].numpy()
            out["bn1_running_var_after"] = sd["embedder.embedder.normalization.running_var"].numpy()
            last = E._resnet_hf_name(f"layer4.{cfg['resnet_layers'][3] - 1}.bn3.weight").replace("weight", "")
            out["last_bn_running_mean_after"] = sd[last + "running_mean"].numpy()
            out["last_bn_running_var_after"] = sd[last + "running_var"].numpy()
            out["bn1_num_batches_tracked_after"] = sd["embedder.embedder.normalization.num_batches_tracked"].numpy()
            # ---- test() and evaluate() on the same 24 memes (eval mode: dropout modules as constructed, running statistics)
            val_loader = ns["DataLoader"](ds, batch_size=cfg["batch"], shuffle=False, drop_last=True)
            test_loss, test_acc = ns["test"](model, val_loader, criterion, device)
            out["test_logits"] = torch.stack(seen).numpy()
            out["test_loss"], out["test_acc"] = np.array(test_loss), np.array(test_acc)
            seen.clear()
            ns["evaluate"](model, val_loader, device)                      # writes task2C_TeamName.tsv into the cwd (:274)
            out["evaluate_tsv"] = np.array(open("task2C_TeamName.tsv", encoding="utf-8").read().split("\n"))
            hook.remove()
        finally:
            os.chdir(cwd)
    out["notes"] = np.array("seq_len=128 (script: 512); shuffle=False; train pass with every nn.Dropout p set to 0 on the constructed "
                            "model; test/evaluate passes with the modules as constructed; lr=2e-5; batch 8; 24 memes; cpu fp32")
    for k, v in cfg.items():
        out["cfg_" + k] = np.array(v)
    path = os.path.join(GOLDEN, "ref_organizers_2c.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays); train loss {train_loss:.6f} acc {acc:.4f}; "
          f"test loss {test_loss:.6f} acc {test_acc:.4f}")
    print("train logits[0]:", out["train_logits"][0][:3], "test logits[0]:", out["test_logits"][0][:3])


def main():
    if "--kevin" not in sys.argv:
        gen_organizers()
    if "--organizers" not in sys.argv:
        from .gen_ref_kevin import gen_kevin
        gen_kevin()


if __name__ == "__main__":
    main()
