"""End-to-end on the GPU with the reference's own calling sequence: read_data -> MultimodalDataset ->
DataLoader -> train()/test()/evaluate() -> TSV accepted by the task's format (GPU box only)."""
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu

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
