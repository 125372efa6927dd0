#!/usr/bin/env python
"""Kevin's three-tower Subtask-2C model (example_scripts/Multimodal_example_task2C.py:590-717) on the MI355X path:

  text tower + image tower   -> memehip.MultimodalClassifier.encode()            (hand-written HIP, lockstep launches)
  caption tower              -> memehip.TextEncoder                              (same kernels)
  Linear+BatchNorm1d+ReLU projections, ConcatAttention3, Linear(512,1)+BatchNorm1d(1)
                             -> PyTorch modules with memehip.BatchNorm1d          (HIP BatchNorm, fused ReLU)
  sigmoid focal loss, parameter groups (0.8x encoder learning rate), linear warm-up, clipping with step skipping
                             -> memehip.SigmoidFocalLoss / Adam / get_linear_schedule_with_warmup

Synthetic inputs (no dataset or checkpoints offline); prints the loss of a few steps.   python examples/task2c_kevin.py
"""
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_propaganda_meme_classification_amd as memehip


class ConcatAttention3(nn.Module):                                   # ...task2C.py:476-499
    def __init__(self, input_dim, attention_dim):
        super().__init__()
        self.attention_fc, self.attention_bn = nn.Linear(input_dim, input_dim), memehip.BatchNorm1d(input_dim, relu=True)
        self.reduce_fc, self.reduce_bn = nn.Linear(input_dim, attention_dim), memehip.BatchNorm1d(attention_dim, relu=True)

    def forward(self, text_features, image_features, caption_features):
        cat = torch.cat((text_features, image_features, caption_features), dim=1)
        weights = torch.softmax(self.attention_bn(self.attention_fc(cat)), dim=1)
        return self.reduce_bn(self.reduce_fc(weights * cat))


class KevinClassifier(nn.Module):                                    # ...task2C.py:590-685
    def __init__(self, small: bool):
        super().__init__()
        if small:      # shapes for a quick run
            tc = memehip.TextConfig(vocab_size=2000, hidden=128, layers=2, heads=2, intermediate=256, max_position=64)
            ic = memehip.ImageConfig(image_size=64, hidden=128, layers=2, heads=2, intermediate=256)
            cc = memehip.TextConfig(vocab_size=2000, hidden=128, layers=2, heads=2, intermediate=256, max_position=64)
        else:          # AraBERT-base + ViT-B/16 + bert-base-uncased captions
            tc, ic, cc = memehip.TextConfig(), memehip.ImageConfig(), memehip.TextConfig(vocab_size=30522)
        self.towers = memehip.MultimodalClassifier.from_config(memehip.ModelConfig(text=tc, image=ic))
        self.caption_text_model = memehip.TextEncoder(cc)
        P = 512 if not small else 128
        self.text_dropout, self.caption_text_dropout = nn.Dropout(0.3), nn.Dropout(0.3)
        self.text_fc = nn.Sequential(nn.Linear(tc.hidden, P), memehip.BatchNorm1d(P, relu=True))
        self.caption_text_fc = nn.Sequential(nn.Linear(cc.hidden, P), memehip.BatchNorm1d(P, relu=True))
        self.image_fine_tune = nn.Sequential(nn.Linear(ic.hidden, P), nn.ReLU(inplace=True), nn.Dropout(0.35), nn.Linear(P, P))
        self.fusion_layer = ConcatAttention3(3 * P, P)
        self.output_fc = nn.Sequential(nn.Linear(P, 1), memehip.BatchNorm1d(1))

    def get_params(self, lr):                                        # ...task2C.py:645-664
        enc = list(self.towers.parameters()) + list(self.caption_text_model.parameters())
        enc_ids = {id(p) for p in enc}
        head = [p for p in self.parameters() if id(p) not in enc_ids]
        return head, enc

    def forward(self, text, image, mask, caption_text, caption_text_mask):
        t, v = self.towers.encode(text, image, mask)
        c = self.caption_text_model(caption_text, caption_text_mask)
        t = self.text_fc(self.text_dropout(t))
        c = self.caption_text_fc(self.caption_text_dropout(c))
        fused = self.fusion_layer(t, self.image_fine_tune(v), c)
        return self.output_fc(fused).squeeze(1)


def main():
    small = "--full" not in sys.argv
    device = torch.device("cuda")
    torch.manual_seed(0)
    model = KevinClassifier(small).to(device)
    model.train()
    B, S = (8, 32) if small else (32, 128)
    tc, ic = model.towers.config.text, model.towers.config.image
    g = torch.Generator().manual_seed(1)
    text = torch.randint(5, tc.vocab_size, (B, S), generator=g)
    cap = torch.randint(5, tc.vocab_size if small else 30522, (B, S), generator=g)
    lens = torch.randint(4, S + 1, (B,), generator=g)
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    image = torch.randn((B, 3, ic.image_size, ic.image_size), generator=g)
    labels = (torch.rand(B, generator=g) < 0.28).float()
    batch = [x.to(device) for x in (text * mask, image, mask, cap * mask, mask)]

    lr = 2e-5
    head, enc = model.get_params(lr)
    # the fused Adam updates one flat buffer per tower module; the small head uses torch.optim.Adam
    opt_towers = memehip.Adam(model.towers.parameters(), lr=lr * 0.8, max_grad_norm=1.0)
    opt_caption = memehip.Adam(model.caption_text_model.parameters(), lr=lr * 0.8, max_grad_norm=1.0)
    opt_head = torch.optim.Adam(head, lr=lr)
    sched = [memehip.get_linear_schedule_with_warmup(o, num_warmup_steps=2, num_training_steps=10)
             for o in (opt_towers, opt_caption, opt_head)]
    criterion = memehip.SigmoidFocalLoss(alpha=0.25, gamma=2.0)
    for step in range(5):
        for o in (opt_towers, opt_caption, opt_head):
            o.zero_grad()
        output = model(*batch)
        loss = criterion(output, labels.to(device))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(head, 1.0)
        for o in (opt_towers, opt_caption, opt_head):
            o.step()
        for s in sched:
            s.step()
        print(f"step {step}: focal loss {float(loss.detach()):.5f}", flush=True)
    print("ok")


if __name__ == "__main__":
    main()
