"""Name the `__amd_rocclr_copyBuffer` launches of a config-2 (ResNet-50) step: one EAGER step under torch.profiler with Python
stacks; every aten::copy_ / clone / contiguous / _to_copy that ran on the device is printed with its shapes, dtypes and the first
package frame of its stack.   python tools/find_copies.py [--config 2|3]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_propaganda_meme_classification_amd as pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = pkg.ResNetClassifier(num_labels=2, compute_dtype="fp16").to(dev)
    pkg.flatten_parameters(model)
    model.train()
    opt = pkg.Adam(model.parameters(), lr=2e-5, skip_nonfinite=True)
    image = torch.randn((a.batch, 3, 224, 224), device=dev)
    labels = (torch.rand((a.batch,), device=dev) < 0.28).long()

    def step():
        opt.zero_grad()
        loss, _ = model(pixel_values=image, labels=labels)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    names = collections.Counter()
    sites = collections.Counter()
    for ev in prof.events():
        n = ev.name
        if "memcpy" in n.lower() or "copyBuffer" in n:
            names[n] += 1
        if n in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::detach", "aten::add_", "aten::add", "aten::fill_",
                 "aten::zero_"):
            frame = "?"
            for fr in (ev.stack or []):
                if "multimodal_propaganda" in fr or "find_copies" in fr or "autograd" in fr:
                    frame = fr
                    break
            sites[(n, str(ev.input_shapes)[:80], frame[-110:])] += 1
    print("device-side copy events:", dict(names))
    for (n, shp, fr), c in sorted(sites.items(), key=lambda kv: -kv[1])[:40]:
        print(f"{c:4d} x {n:18s} {shp:80s} {fr}")


if __name__ == "__main__":
    main()
