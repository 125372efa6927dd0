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


def test_kevin_dataset_and_loops_end_to_end(pkg, tmp_path):
    """Kevin's callers (Multimodal_example_task2C.py:208-304, 688-871) on the HIP path: the dataset's dict keys, one epoch of
    train() with focal loss + fused Adam over get_params + warm-up schedule + the device image pipeline (augmentations on),
    test() -> (loss, accuracy, macro F1, ROC-optimal threshold), evaluate() -> the two TSVs in the reference's formats."""
    import re
    from PIL import Image
    kv = pkg.kevin
    rng = np.random.default_rng(0)
    n = 24
    names = []
    for i in range(n):                                   # real image files of different sizes: the device resizes them
        h, w = int(rng.integers(40, 90)), int(rng.integers(40, 90))
        arr = rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
        arr[:, :, 0] = (arr[:, :, 0] // 4 + (190 if i % 2 else 10)).astype(np.uint8)      # a learnable cue: red level = label
        p = tmp_path / f"img{i}.png"
        Image.fromarray(arr).save(p)
        names.append(p.name)
    ids = [f"id{i}" for i in range(n)]
    texts = [("propaganda words here " if i % 2 else "plain words there ") + str(i) for i in range(n)]
    labels = [i % 2 for i in range(n)]
    caps = [f"a meme of thing {i % 3}" for i in range(n)]
    with pytest.raises(ValueError, match="captions"):
        kv.KevinMultimodalDataset(ids, texts, names, labels)
    ds = kv.KevinMultimodalDataset(ids, texts, names, labels, captions=caps, max_seq_len=16, image_size=32, image_root=str(tmp_path),
                                   vocab_size=600, english_vocab_size=500)
    item = ds[3]
    assert sorted(item) == sorted(["id", "text", "text_mask", "caption_text", "caption_text_mask", "image", "label"])      # :290-303
    assert item["text"].shape == (16,) and item["caption_text_mask"].dtype == torch.int64 and item["image"].dtype == np.uint8
    assert sorted(kv.KevinMultimodalDataset(ids, texts, names, labels, is_test=True, captions=caps, image_root=str(tmp_path))[0]) == \
        sorted(["id", "text", "text_mask", "caption_text", "caption_text_mask", "image"])
    loader = torch.utils.data.DataLoader(ds, batch_size=8, shuffle=False, collate_fn=kv.kevin_collate)
    tc = pkg.TextConfig(vocab_size=600, hidden=128, layers=2, heads=2, intermediate=256, max_position=64)
    ic = pkg.ImageConfig(image_size=32, hidden=128, layers=2, heads=2, intermediate=256)
    cc = pkg.TextConfig(vocab_size=500, hidden=128, layers=1, heads=2, intermediate=256, max_position=64)
    model = pkg.KevinMultimodalClassifier("concatenation", text=tc, image=ic, caption=cc, proj=64, compute_dtype="fp16", seed=1).cuda()
    opt = pkg.Adam(model.get_params(2e-3), max_grad_norm=1.0)
    sched = pkg.get_linear_schedule_with_warmup(opt, num_warmup_steps=1, num_training_steps=12)
    crit = pkg.SigmoidFocalLoss()
    dev = torch.device("cuda")
    pipe_train = pkg.data.DeviceImagePipeline(image_size=32, mode="stretch", augment=True, device=dev)
    pipe_eval = pkg.data.DeviceImagePipeline(image_size=32, mode="stretch", augment=False, device=dev)
    calls = []
    hist = [kv.train(model, loader, crit, opt, sched, dev, epoch, image_pipeline=pipe_train, eval_fn=calls.append, log_every=0)
            for epoch in range(4)]
    assert all(np.isfinite(h[0]) for h in hist) and hist[-1][0] < hist[0][0], hist
    assert calls[:3] == [1, 2, 3]                               # check_interval = 3 // 2 = 1 -> every batch, as the reference computes it
    loss, acc, f1, thr = kv.test(model, loader, crit, dev, 0, image_pipeline=pipe_eval)
    assert np.isfinite(loss) and 0.0 <= acc <= 1.0 and 0.0 <= f1 <= 1.0 and np.isfinite(thr)
    f_lab, f_prob = kv.evaluate(model, loader, thr, dev, team_name="t", fold=2, out_dir=str(tmp_path), image_pipeline=pipe_eval)
    lines = open(f_lab).read().splitlines()
    assert lines[0] == "id\tlabel\trun_id" and len(lines) == n + 1
    assert all(re.fullmatch(r"id\d+\t(not_propaganda|propaganda)\t\S+", ln) for ln in lines[1:])
    plines = open(f_prob).read().splitlines()
    assert os.path.basename(f_prob) == "task2C_t_probs_fold_2.tsv" and plines[0] == "id\tlabel\tprob\trun_id"
    for ln in plines[1:]:
        _id, lab, prob, _ = ln.split("\t")
        assert (float(prob) > thr) == (lab == "propaganda")
