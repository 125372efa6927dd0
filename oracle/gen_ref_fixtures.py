"""Fixtures made by RUNNING the reference's own pure-torch classes (build container only).

TEST INFRASTRUCTURE.  ``python -m oracle.gen_ref_fixtures`` reads
/root/reference/example_scripts/Multimodal_example_task2C.py as text, takes the class definitions it needs out of its
AST -- ``LLMWithClassificationHead`` (:307-392, the pooling branches), ``ConcatAttention3`` (:476-499), ``MCA3`` (:423-448) and
``MultimodalClassifier`` (:587-685, for ``get_params`` only; the class is never constructed: its __init__ downloads
checkpoints) -- executes exactly those definitions with ``torch / nn / F`` in scope, and records inputs and outputs in
tests/golden/ref_kevin_heads.npz.  The script as a whole cannot be imported (torchvision / timm / network); nothing of the
reference's text is stored: the fixture holds tensors only.

``AutoModel.from_pretrained`` inside ``LLMWithClassificationHead.__init__`` is bound to a holder module that returns a
given last_hidden_state (the encoder is not what these vectors pin -- the poolings on top of it are).
"""
from __future__ import annotations

import ast
import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference/example_scripts/Multimodal_example_task2C.py"
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
WANT = ("LLMWithClassificationHead", "ConcatAttention3", "MCA3", "MultimodalClassifier")


class _HiddenHolder(nn.Module):
    """Stands where the pretrained encoder would be: forward returns the hidden states it was given."""

    def __init__(self, hidden: torch.Tensor):
        super().__init__()
        self.hidden = nn.Parameter(hidden.clone())

    def forward(self, input_ids=None, attention_mask=None):
        return SimpleNamespace(last_hidden_state=self.hidden)


def extract_reference_classes():
    tree = ast.parse(open(REF).read(), filename=REF)
    nodes = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in WANT]
    assert sorted(n.name for n in nodes) == sorted(WANT), [n.name for n in nodes]
    holder = {}

    class AutoModel:
        @staticmethod
        def from_pretrained(name):
            return _HiddenHolder(holder["hidden"])

    ns = {"torch": torch, "nn": nn, "F": F, "AutoModel": AutoModel, "__name__": "reference_extract"}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), REF, "exec"), ns)
    return ns, holder


def main():
    torch.manual_seed(0)
    ns, holder = extract_reference_classes()
    out = {}
    g = torch.Generator().manual_seed(42)
    # ---- pooling branches -------------------------------------------------------------------------------------
    B, S, D, A = 3, 10, 128, 64
    hidden = torch.randn((B, S, D), generator=g)
    lens = torch.tensor([10, 4, 7])
    mask = (torch.arange(S)[None] < lens[:, None]).to(torch.int64)
    r = torch.randn((B, D), generator=g)
    out.update(pool_hidden=hidden.numpy(), pool_mask=mask.numpy(), pool_r=r.numpy())
    for kind in ("cls", "max", "mean", "attention", "cnn"):
        holder["hidden"] = hidden
        torch.manual_seed(7)
        m = ns["LLMWithClassificationHead"]("stub", kind, hidden_size=D, attention_hidden_size=A, cnn_kernel_size=3)
        for n_, p_ in m.named_parameters():
            if not n_.startswith("model."):
                out[f"pool_{kind}_param_{n_}"] = p_.detach().numpy().copy()
        y = m(torch.zeros((B, S), dtype=torch.long), mask.float() if kind == "attention" else mask)
        (y * r).sum().backward()
        out[f"pool_{kind}_out"] = y.detach().numpy()
        out[f"pool_{kind}_dh"] = m.model.hidden.grad.numpy().copy()
        for n_, p_ in m.named_parameters():
            if not n_.startswith("model."):
                out[f"pool_{kind}_grad_{n_}"] = p_.grad.numpy().copy()
    # unsupported pooling raises ValueError (Multimodal_example_task2C.py:351-352)
    holder["hidden"] = hidden
    bad = ns["LLMWithClassificationHead"]("stub", "median", hidden_size=D)
    try:
        bad(torch.zeros((B, S), dtype=torch.long), mask)
        raise AssertionError("expected ValueError")
    except ValueError as e:
        out["pool_bad_message"] = np.array(str(e))
    # ---- ConcatAttention3 ---------------------------------------------------------------------------------------
    P, Bc = 64, 8
    torch.manual_seed(11)
    ca = ns["ConcatAttention3"](3 * P, P)
    ca.train()
    for n_, t_ in ca.state_dict().items():
        out[f"ca_init_{n_}"] = t_.numpy().copy()
    feats = [torch.randn((Bc, P), generator=g).requires_grad_(True) for _ in range(3)]
    rc = torch.randn((Bc, P), generator=g)
    y = ca(*feats)
    (y * rc).sum().backward()
    out.update(ca_text=feats[0].detach().numpy(), ca_image=feats[1].detach().numpy(), ca_caption=feats[2].detach().numpy(),
               ca_r=rc.numpy(), ca_out=y.detach().numpy())
    for i, nm in enumerate(("text", "image", "caption")):
        out[f"ca_d{nm}"] = feats[i].grad.numpy().copy()
    for n_, p_ in ca.named_parameters():
        out[f"ca_grad_{n_}"] = p_.grad.numpy().copy()
    for n_, t_ in ca.state_dict().items():
        if "running" in n_ or "num_batches" in n_:
            out[f"ca_after_{n_}"] = t_.numpy().copy()
    ca.eval()
    with torch.no_grad():
        out["ca_out_eval"] = ca(*[f_.detach() for f_ in feats]).numpy()
    # ---- MCA3 (fusion_method = "mca") on 2-D features, as MultimodalClassifier.forward feeds it -------------------------------
    Um, Bm = 64, 6
    torch.manual_seed(13)
    mca = ns["MCA3"](Um)
    for n_, t_ in mca.state_dict().items():
        out[f"mca_init_{n_}"] = t_.numpy().copy()
    mf = [torch.randn((Bm, Um), generator=g).requires_grad_(True) for _ in range(3)]
    rm = torch.randn((Bm, Um), generator=g)
    ym = mca(*mf)
    (ym * rm).sum().backward()
    out.update(mca_text=mf[0].detach().numpy(), mca_image=mf[1].detach().numpy(), mca_caption=mf[2].detach().numpy(),
               mca_r=rm.numpy(), mca_out=ym.detach().numpy())
    for i, nm in enumerate(("text", "image", "caption")):
        out[f"mca_d{nm}"] = mf[i].grad.numpy().copy()
    for n_, p_ in mca.named_parameters():
        out[f"mca_grad_{n_}"] = p_.grad.numpy().copy()
    # ---- get_params grouping (Multimodal_example_task2C.py:645-664) --------------------------------------------
    dummy = nn.Module()
    dummy.text_model = nn.Module()
    dummy.text_model.model = nn.Linear(2, 2)
    dummy.text_fc = nn.Sequential(nn.Linear(2, 2), nn.BatchNorm1d(2))
    dummy.caption_text_model = nn.Module()
    dummy.caption_text_model.model = nn.Linear(2, 2)
    dummy.caption_text_fc = nn.Sequential(nn.Linear(2, 2), nn.BatchNorm1d(2))
    dummy.image_model = nn.Module()
    dummy.image_model.image_model = nn.Linear(2, 2)
    dummy.image_model.fine_tune = nn.Sequential(nn.Linear(2, 2))
    dummy.fusion_layer = nn.Linear(2, 2)
    dummy.output_fc = nn.Sequential(nn.Linear(2, 1), nn.BatchNorm1d(1))
    groups = ns["MultimodalClassifier"].get_params(dummy, 1.0)
    by_id = {}
    for gi, grp in enumerate(groups):
        for p_ in grp["params"]:
            by_id[id(p_)] = (gi, grp["lr"])
    names = [n_ for n_, _ in dummy.named_parameters()]
    out["gp_names"] = np.array(names)
    out["gp_group"] = np.array([by_id[id(p_)][0] for _, p_ in dummy.named_parameters()])
    out["gp_lr"] = np.array([by_id[id(p_)][1] for _, p_ in dummy.named_parameters()])
    path = os.path.join(GOLDEN, "ref_kevin_heads.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
