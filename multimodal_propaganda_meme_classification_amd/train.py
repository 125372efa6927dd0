"""train / test / evaluate with the reference's signatures (example_scripts/Multimodal_example_task2C.txt:200-282).

The loops are the reference's in behaviour: zero_grad -> forward -> criterion -> backward -> step,
running loss weighted by batch size, accuracy from argmax, both divided by ``len(loader.dataset)``;
``evaluate`` writes the 3-column TSV ``id<TAB>label<TAB>run_id`` with the header the reference
writes (:275), which the task's format checker and scorer accept.
The only addition: loss / accuracy are accumulated on the GPU and read back once per epoch instead
of two ``.item()`` syncs per step (:218-220), unless ``sync_every_step=True``.
"""
from __future__ import annotations

from typing import Tuple

import torch

from .data import id2l, normalize_images


def train(model, train_loader, criterion, optimizer, device, sync_every_step: bool = False) -> Tuple[float, float]:
    model.train()
    loss_sum = torch.zeros((), device=device)
    correct = torch.zeros((), device=device, dtype=torch.long)
    for data in train_loader:
        optimizer.zero_grad()
        text = data["text"].to(device)
        image = normalize_images(data["image"].to(device, non_blocking=True))
        mask = data["text_mask"].to(device)
        labels = data["label"].to(device)
        output = model(text, image, mask)
        loss = criterion(output, labels)
        loss.backward()
        optimizer.step()
        loss_sum += loss.detach() * labels.size(0)
        _, predicted = torch.max(output.detach(), 1)
        correct += (predicted == labels).sum()
        if sync_every_step:
            loss_sum.item()
    n = len(train_loader.dataset)
    return float(loss_sum) / n, int(correct) / n


def test(model, test_loader, criterion, device) -> Tuple[float, float]:
    model.eval()
    loss_sum = torch.zeros((), device=device)
    correct = torch.zeros((), device=device, dtype=torch.long)
    with torch.no_grad():
        for data in test_loader:
            text = data["text"].to(device)
            image = normalize_images(data["image"].to(device, non_blocking=True))
            mask = data["text_mask"].to(device)
            labels = data["label"].to(device)
            output = model(text, image, mask)
            loss = criterion(output, labels)
            loss_sum += loss * labels.size(0)
            _, predicted = torch.max(output, 1)
            correct += (predicted == labels).sum()
    n = len(test_loader.dataset)
    return float(loss_sum) / n, int(correct) / n


def evaluate(model, test_loader, device, out_path: str = "task2C_TeamName.tsv", run_id: str = "ViT-BERT-memehip") -> str:
    model.eval()
    predictions, ids = [], []
    with torch.no_grad():
        for data in test_loader:
            text = data["text"].to(device)
            image = normalize_images(data["image"].to(device, non_blocking=True))
            mask = data["text_mask"].to(device)
            output = model(text, image, mask)
            _, predicted = torch.max(output, 1)
            predictions.append(predicted.cpu())
            ids.append(data["id"])
    with open(out_path, "w") as f:
        f.write("id\tlabel\trun_id\n")
        for i, line in enumerate(predictions):
            for indx, l in enumerate(line.tolist()):
                f.write(f"{ids[i][indx]}\t{id2l[l]}\t{run_id}\n")
    return out_path
