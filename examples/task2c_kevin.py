#!/usr/bin/env python
"""Kevin's three-tower Subtask-2C step (example_scripts/Multimodal_example_task2C.py:587-776) on the MI355X path, written
the way the reference writes it:

    model = MultimodalClassifier("concatenation")                       -> memehip.KevinMultimodalClassifier
    criterion = sigmoid_focal_loss                                      -> memehip.SigmoidFocalLoss
    optimizer = optim.Adam(model.get_params(lr), lr=lr)                 -> memehip.Adam (ONE optimizer, three groups, three
                                                                           flat buffers, ONE global clip norm)
    scheduler = get_linear_schedule_with_warmup(...)                    -> memehip.get_linear_schedule_with_warmup
    output = model(text, image, mask, caption_text, caption_text_mask); loss = criterion(output, labels, alpha=0.25,
    gamma=2.0, reduction="mean"); loss.backward(); clip to 1.0; optimizer.step(); scheduler.step()

Towers, poolings, Linear+BatchNorm1d+ReLU projections, ConcatAttention3, the focal loss and Adam all run in libmemehip.
Synthetic inputs (no dataset or checkpoints offline).   python examples/task2c_kevin.py [--full]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_propaganda_meme_classification_amd as memehip


def main():
    small = "--full" not in sys.argv
    device = torch.device("cuda")
    torch.manual_seed(0)
    if small:
        tc = memehip.TextConfig(vocab_size=2000, hidden=128, layers=2, heads=2, intermediate=256, max_position=64)
        ic = memehip.ImageConfig(image_size=64, hidden=128, layers=2, heads=2, intermediate=256)
        cc = memehip.TextConfig(vocab_size=2000, hidden=128, layers=2, heads=2, intermediate=256, max_position=64)
        model = memehip.KevinMultimodalClassifier("concatenation", text=tc, image=ic, caption=cc, proj=128, compute_dtype="fp16")
        B, S = 8, 32
    else:          # AraBERT-base + ViT-B/16 + bert-base-uncased captions
        model = memehip.KevinMultimodalClassifier("concatenation", compute_dtype="fp16")
        tc, ic, cc = model.towers.config.text, model.towers.config.image, memehip.TextConfig(vocab_size=30522)
        B, S = 32, 128
    model.to(device).train()
    g = torch.Generator().manual_seed(1)
    text = torch.randint(5, tc.vocab_size, (B, S), generator=g)
    cap = torch.randint(5, cc.vocab_size, (B, S), generator=g)
    lens = torch.randint(4, S + 1, (B,), generator=g)
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    image = torch.randn((B, 3, ic.image_size, ic.image_size), generator=g)
    labels = (torch.rand(B, generator=g) < 0.28).float().to(device)
    batch = [x.to(device) for x in (text * mask, image, mask, cap * mask, mask)]

    lr = 2e-5
    optimizer = memehip.Adam(model.get_params(lr), lr=lr, max_grad_norm=1.0)      # clip_grad_norm_(model.parameters(), 1.0), fused
    scheduler = memehip.get_linear_schedule_with_warmup(optimizer, num_warmup_steps=2, num_training_steps=10)
    criterion = memehip.SigmoidFocalLoss()
    for step in range(5):
        optimizer.zero_grad()
        output = model(*batch)
        loss = criterion(output, labels, alpha=0.25, gamma=2.0, reduction="mean")
        loss.backward()
        grad_norm = optimizer.grad_norm()          # clip_grad_norm_(model.parameters(), float("inf")) of the reference's log line
        optimizer.step()
        scheduler.step()
        print(f"step {step}: focal loss {float(loss.detach()):.5f} | LR {scheduler.get_last_lr()[0]:.2e} | Grad Norm {float(grad_norm):.4f}",
              flush=True)
    print("ok")


if __name__ == "__main__":
    main()
