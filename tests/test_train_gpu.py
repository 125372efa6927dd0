"""End-to-end on the GPU with the reference's own calling sequence: read_data -> MultimodalDataset ->
DataLoader -> train()/test()/evaluate() -> TSV accepted by the task's format (GPU box only)."""
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import multimodal_propaganda_meme_classification_amd as m
    return m

LINE = re.compile(r'^([\w:]+\/.*?\.[\w:]+)\t(propaganda|not_propaganda)\t[\w-]+')     # format_checker/task2.py:20


def test_reference_style_training_loop(golden_dir, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import multimodal_propaganda_meme_classification_amd as pkg
    cfg = pkg.ModelConfig(text=pkg.TextConfig(vocab_size=512, hidden=128, layers=2, heads=2, intermediate=256, max_position=64),
                          image=pkg.ImageConfig(image_size=32, hidden=128, layers=2, heads=2, intermediate=256), proj=128)
    df = pkg.read_data(os.path.join(golden_dir, "dev_sample12.json"))
    df["label"] = df["label"].map(pkg.l2id)
    ds = pkg.MultimodalDataset(df["id"], df["text"], df["image"], df["label"], max_seq_len=32, image_size=32,
                               synthetic_images=True, vocab_size=512)
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, drop_last=True)
    device = torch.device("cuda")
    model = pkg.MultimodalClassifier(num_classes=2, config=cfg, seed=1)
    model.to(device)
    criterion = pkg.CrossEntropyLoss()
    optimizer = pkg.Adam(model.parameters(), lr=1e-3)
    losses = []
    for epoch in range(6):
        loss, acc = pkg.train(model, loader, criterion, optimizer, device)
        assert loss == loss and 0.0 <= acc <= 1.0
        losses.append(loss)
    assert losses[-1] < losses[0], losses                 # it learns the 12 memes
    tl, ta = pkg.test(model, loader, criterion, device)
    assert tl == tl and 0.0 <= ta <= 1.0
    # torch's own criterion / optimizer are drop-in too (the boundary is the nn.Module protocol)
    opt2 = torch.optim.Adam(model.parameters(), lr=1e-4)
    loss2, _ = pkg.train(model, loader, torch.nn.CrossEntropyLoss(), opt2, device)
    model.mark_weights_changed()                           # weights edited outside the fused optimizer
    assert loss2 == loss2
    out = pkg.evaluate(model, loader, device, out_path=str(tmp_path / "task2C_memehip.tsv"))
    lines = open(out).read().strip().split("\n")
    assert lines[0] == "id\tlabel\trun_id" and len(lines) == 13
    ids = set(df["id"])
    for ln in lines[1:]:
        assert LINE.match(ln), ln
        assert ln.split("\t")[0] in ids


def test_feature_dump_feeds_the_svm_baseline(pkg, golden_dir, tmp_path):
    """SURVEY 8 f rank 3 end to end: get_features over a DataLoader (baselines/extract_feat.py:52-67) -> the
    {"imgfeats", "textfeats"} JSON (:110) -> the linear SVM over concat(img, text) features of
    baselines/subtask_2c.py:74-95 -> a TSV the task's format accepts."""
    import json
    from sklearn.svm import SVC
    df = pkg.read_data(os.path.join(golden_dir, "dev_sample12.json"))
    labels = [pkg.l2id[l] for l in df["label"]]
    ds = pkg.MultimodalDataset(df["id"], df["text"], df["image"], labels, max_seq_len=32, image_size=32, synthetic_images=True,
                               vocab_size=512)
    cfg = pkg.ModelConfig(text=pkg.TextConfig(vocab_size=512, hidden=128, layers=2, heads=2, intermediate=256, max_position=64),
                          image=pkg.ImageConfig(image_size=32, hidden=128, layers=2, heads=2, intermediate=256), proj=128)
    model = pkg.MultimodalClassifier.from_config(cfg, seed=4).to("cuda")
    g = torch.Generator().manual_seed(0)
    pooler = (torch.randn((128, 128), generator=g) * 0.05, torch.randn((128,), generator=g) * 0.05)
    loader = torch.utils.data.DataLoader(ds, batch_size=5)
    img_feats, text_feats = pkg.get_features(loader, model, "cuda", pooler=pooler)
    assert sorted(img_feats) == sorted(str(i) for i in df["id"]) and len(img_feats[str(df["id"][0])]) == 128
    # pooler_output = tanh(W h_cls + b) of the pooled text row
    b0 = ds[0]
    f0 = model.get_features(b0["text"][None].cuda(), b0["image"][None].cuda(), b0["text_mask"][None].cuda())
    want = torch.tanh(f0["text"].cpu() @ pooler[0].t() + pooler[1])[0]
    assert float((torch.tensor(text_feats[str(df["id"][0])]) - want).abs().max()) < 1e-5
    path = pkg.dump_features(str(tmp_path / "features" / "dev_feats.json"), img_feats, text_feats)
    feats = json.load(open(path))
    assert set(feats) == {"imgfeats", "textfeats"}
    # run_imgbert_baseline's ten lines
    id_lab = [[str(i), l] for i, l in zip(df["id"], df["label"])]
    x = np.array([feats["imgfeats"][i] + feats["textfeats"][i] for i, _ in id_lab])
    clf = SVC(C=1, kernel="linear", random_state=0)
    clf.fit(x, [l for _, l in id_lab])
    out = tmp_path / "task2C_imgbert.tsv"
    with open(out, "w") as f:
        f.write("id\tclass_label\trun_id\n")
        for (i, _), lab in zip(id_lab, clf.predict(x)):
            f.write(f"{i}\t{lab}\timgbert\n")
    lines = open(out).read().splitlines()[1:]
    assert len(lines) == len(id_lab) and all(LINE.match(l) for l in lines)
