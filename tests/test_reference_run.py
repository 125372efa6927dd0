"""CPU side of the reference-run fixtures (tests/golden/ref_organizers_2c.npz, ref_kevin_2c.npz; made by executing the reference's
own definitions, oracle/gen_ref_hotpath.py + oracle/gen_ref_kevin.py):

* the ORACLE (oracle/meme_oracle.py + oracle/resnet_oracle.py) reproduces what the reference's ``MultimodalClassifier.forward`` /
  ``train`` computed -- this is what pins the oracle to the reference itself, not only to transformers;
* the product's host code (``read_data``, ``MultimodalDataset``, ``KevinMultimodalDataset``) yields the reference Dataset's items bit
  for bit (token ids, masks, labels, dict keys) and its image tensors to float rounding.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import meme_oracle as O
from oracle import ref_env as E
from oracle import resnet_oracle as R


def _z(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def test_fixture_inputs_rebuild_from_seeds(golden_dir, tmp_path):
    """The GPU tests rebuild files / tokenizer / images from seeds: they must be the ones the reference run saw."""
    z = _z(golden_dir, "ref_organizers_2c")
    E.write_dataset(str(tmp_path))
    tok = E.EncodePlusTokenizer([r["text"] for r in E.records24()], str(tmp_path), "vocab_ar")
    assert tok.vocab_size == int(z["vocab_size"])
    assert list(z["ds_ids"]) == [r["id"] for r in E.records24()]
    zk = _z(golden_dir, "ref_kevin_2c")
    imgs = E.kevin_images(str(tmp_path), E.KEVIN["aug_seed"])
    chk = np.stack([[float(im.double().sum()), float(im.double().abs().sum())] for im in imgs])
    np.testing.assert_allclose(chk, zk["ds_image_checksum"], rtol=1e-9)


def test_product_dataset_yields_the_reference_datasets_items(golden_dir, tmp_path):
    """read_data -> label map -> MultimodalDataset (Multimodal_example_task2C.txt:88-115) through the product's data.py against the
    items the reference's own classes produced from the same files and tokenizer."""
    import multimodal_propaganda_meme_classification_amd as pkg
    z = _z(golden_dir, "ref_organizers_2c")
    json_path = E.write_dataset(str(tmp_path))
    tok = E.EncodePlusTokenizer([r["text"] for r in E.records24()], str(tmp_path), "vocab_ar")
    df = pkg.read_data(json_path)
    assert list(df.columns) == [str(c) for c in z["read_data_columns"]]
    assert list(pkg.read_data(json_path, is_test=True).columns) == [str(c) for c in z["read_data_test_columns"]]
    df["label"] = df["label"].map(pkg.l2id)
    ds = pkg.MultimodalDataset(df["id"], df["text"], df["image"], df["label"], tokenizer=tok, max_seq_len=int(z["cfg_seq_len"]),
                               image_root=str(tmp_path))
    assert len(ds) == 24
    items = [ds[i] for i in range(24)]
    assert sorted(items[0]) == [str(k) for k in z["ds_keys"]]
    assert [it["id"] for it in items] == [str(i) for i in z["ds_ids"]]
    assert np.array_equal(torch.stack([it["text"] for it in items]).numpy(), z["ds_text"])
    assert np.array_equal(torch.stack([it["text_mask"] for it in items]).numpy(), z["ds_text_mask"])
    lab = torch.stack([it["label"] for it in items])
    assert lab.dtype == torch.int64 and np.array_equal(lab.numpy(), z["ds_label"])
    imgs = torch.stack([it["image"] for it in items])
    assert imgs.shape == (24, 3, 224, 224) and imgs.dtype == torch.float32
    chk = np.stack([[float(im.double().sum()), float(im.double().abs().sum())] for im in imgs])
    np.testing.assert_allclose(chk, z["ds_image_checksum"], rtol=1e-6)
    assert float(np.abs(imgs[:, :, 100:108, 100:108].numpy() - z["ds_image_patch"]).max()) < 1e-6
    ds_t = pkg.MultimodalDataset(df["id"], df["text"], df["image"], df["label"], is_test=True, tokenizer=tok, image_root=str(tmp_path))
    assert sorted(ds_t[0]) == [str(k) for k in z["ds_test_keys"]]
    # uint8 variant (ToTensor + Normalize on the device): the same crop, bytes exact
    ds_u8 = pkg.MultimodalDataset(df["id"], df["text"], df["image"], df["label"], tokenizer=tok, image_root=str(tmp_path), device_normalize=True)
    u8 = ds_u8[5]["image"]
    mean, std = torch.tensor(pkg.data.IMAGENET_MEAN), torch.tensor(pkg.data.IMAGENET_STD)
    back = ((u8.float() / 255.0 - mean) / std).permute(2, 0, 1)
    assert float((back - imgs[5]).abs().max()) < 1e-6


def test_kevin_dataset_tokens_match_the_reference_dataset(golden_dir, tmp_path):
    import multimodal_propaganda_meme_classification_amd as pkg
    z = _z(golden_dir, "ref_kevin_2c")
    E.write_dataset(str(tmp_path))
    recs, caps = E.records24(), E.captions24()
    tok_ar = E.EncodePlusTokenizer([r["text"] for r in recs], str(tmp_path), "vocab_ar")
    tok_en = E.EncodePlusTokenizer(caps, str(tmp_path), "vocab_en")
    ds = pkg.kevin.KevinMultimodalDataset([r["id"] for r in recs], [r["text"] for r in recs], [r["img_path"] for r in recs],
                                          [pkg.l2id[r["class_label"]] for r in recs], captions=caps, tokenizer=tok_ar, english_tokenizer=tok_en,
                                          max_seq_len=E.KEVIN["seq_len"], image_root=str(tmp_path))
    items = [ds[i] for i in range(24)]
    assert sorted(items[0]) == [str(k) for k in z["ds_keys"]]
    for k in ("text", "text_mask", "caption_text", "caption_text_mask", "label"):
        assert np.array_equal(torch.stack([it[k] for it in items]).numpy(), z["ds_" + k]), k
    assert items[3]["image"].dtype == np.uint8 and items[3]["image"].shape == E.synthetic_meme(3).shape       # decoded, the device resizes


def _organizers_oracle(z):
    cfg = {k[4:]: z[k] for k in z.files if k.startswith("cfg_")}
    layers = tuple(int(x) for x in cfg["resnet_layers"])
    V = int(z["vocab_size"])
    state = E.organizers_state(V, int(cfg["text_layers"]), layers, int(cfg["seed"]))
    tcfg = O.TextConfig(vocab_size=V, hidden=768, layers=int(cfg["text_layers"]), heads=12, intermediate=3072, max_position=512, type_vocab=0)
    back = {E.bert_to_distil_name(k): k for k in O._text_shapes(tcfg, pfx="")}
    p_text = {"bert." + back[k[len("bert."):]]: v for k, v in state.items() if k.startswith("bert.")}
    p_res = {k[len("resnet."):]: v for k, v in state.items() if k.startswith("resnet.")}
    heads = {k: v for k, v in state.items() if k.split(".")[0].endswith("_fc")}
    return cfg, layers, tcfg, p_text, p_res, heads


def test_oracle_reproduces_the_reference_organizers_forward_and_gradients(golden_dir, tmp_path):
    """oracle text tower (DistilBERT = BERT without token types, last position) + oracle ResNet-50 + the four Linear layers against
    batch 1 of the reference's train(): logits, loss, and the step-1 gradients its optimizer saw."""
    z = _z(golden_dir, "ref_organizers_2c")
    cfg, layers, tcfg, p_text, p_res, heads = _organizers_oracle(z)
    B = int(cfg["batch"])
    E.write_dataset(str(tmp_path))
    tf = E.organizers_transform()
    from PIL import Image
    image = torch.stack([tf(Image.open(os.path.join(tmp_path, r["img_path"])).convert("RGB")) for r in E.records24()[:B]])
    text, mask, labels = torch.from_numpy(z["ds_text"][:B]), torch.from_numpy(z["ds_text_mask"][:B]), torch.from_numpy(z["ds_label"][:B])
    torch.set_num_threads(8)
    leaves = {k: v.clone().requires_grad_(True) for k, v in {**p_text, **{"resnet." + k: v for k, v in p_res.items()}, **heads}.items()}
    st = R.new_bn_state(p_res)
    t = O.text_tower({k: v for k, v in leaves.items() if k.startswith("bert.")}, text, mask, tcfg)[:, -1]
    t = F.linear(t, leaves["bert_fc.weight"], leaves["bert_fc.bias"])
    r = R.resnet_forward({k[len("resnet."):]: v for k, v in leaves.items() if k.startswith("resnet.")}, st, image, layers, training=True)
    r = F.linear(r, leaves["resnet_fc.weight"], leaves["resnet_fc.bias"])
    f = F.linear(torch.cat((t, r), 1), leaves["fusion_fc.weight"], leaves["fusion_fc.bias"])
    logits = F.linear(f, leaves["output_fc.weight"], leaves["output_fc.bias"])
    np.testing.assert_allclose(logits.detach().numpy(), z["train_logits"][0], atol=3e-5)
    F.cross_entropy(logits, labels).backward()
    names = [str(n) for n in z["param_names"]]
    back = {E.bert_to_distil_name(k): k for k in O._text_shapes(tcfg, pfx="")}
    worst = 0.0
    for i, n in enumerate(names):
        key = ("bert." + back[n[len("bert."):]]) if n.startswith("bert.") else n
        g = leaves[key].grad
        ref = float(z["grad_norms_step1"][i])
        if ref > 1e-7:
            worst = max(worst, abs(float(g.double().norm()) - ref) / ref)
        fl = g.reshape(-1)
        np.testing.assert_allclose(fl[E.sample_index(fl.numel())].numpy(), z["grad_samples_step1"][i], rtol=3e-2,
                                   atol=3e-4 * float(np.abs(z["grad_samples_step1"][i]).max()) + 1e-7, err_msg=n)
    assert worst < 5e-3, worst


def test_oracle_reproduces_the_reference_kevin_forward(golden_dir, tmp_path):
    """oracle BERT (cls) x 2 + oracle ViT (cls token) + Kevin's head written with torch.nn against batch 1 of the reference's
    train(): the [B] outputs of MultimodalClassifier.forward (Multimodal_example_task2C.py:666-685) and the focal loss."""
    z = _z(golden_dir, "ref_kevin_2c")
    cfg = E.KEVIN
    E.write_dataset(str(tmp_path))
    Vt, Vc = (int(v) for v in z["vocab_sizes"])
    state = E.kevin_state(Vt, Vc, cfg)
    B, P, v = cfg["batch"], cfg["proj"], cfg["vit"]
    images = E.kevin_images(str(tmp_path), cfg["aug_seed"])[:B]
    tcfg = O.TextConfig(vocab_size=Vt, hidden=768, layers=cfg["text_layers"], heads=12, intermediate=3072, max_position=512)
    ccfg = O.TextConfig(vocab_size=Vc, hidden=768, layers=cfg["caption_layers"], heads=12, intermediate=3072, max_position=512)
    icfg = O.ImageConfig(image_size=v["image_size"], patch=v["patch"], hidden=v["hidden"], layers=v["layers"], heads=v["heads"], intermediate=v["intermediate"])
    sub = lambda pfx, new: {new + k[len(pfx):]: t for k, t in state.items() if k.startswith(pfx)}
    text, mask = torch.from_numpy(z["ds_text"][:B]), torch.from_numpy(z["ds_text_mask"][:B])
    cap, cmask = torch.from_numpy(z["ds_caption_text"][:B]), torch.from_numpy(z["ds_caption_text_mask"][:B])
    labels = torch.from_numpy(z["ds_label"][:B]).float()
    torch.set_num_threads(8)
    with torch.no_grad():
        t = O.text_tower(sub("text_model.model.", "bert."), text, mask, tcfg)[:, 0]
        c = O.text_tower(sub("caption_text_model.model.", "bert."), cap, cmask, ccfg)[:, 0]
        vi = O.image_tower(sub("image_model.image_model.", "image_model."), images, icfg)[:, 0]

        out = E.kevin_head_cpu(state, t, vi, c)
    np.testing.assert_allclose(out.numpy(), z["train_outputs"][0], atol=2e-4)
    assert abs(float(O.sigmoid_focal_loss(out, labels, 0.25, 2.0)) - float(E.focal_standin(torch.from_numpy(z["train_outputs"][0]), labels, 0.25, 2.0, "mean"))) < 1e-5
