import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_propaganda_meme_classification_amd import ops
dev = torch.device("cuda"); BF16 = torch.bfloat16
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3
D = 768
# rotate over several buffers so the data is not L2/MALL-hot (as in the real step)
for rows in (4096, 6304):
    bufs = [(torch.randn(rows, D, device=dev).to(BF16), torch.randn(rows, D, device=dev).to(BF16), torch.randn(rows, D, device=dev).to(BF16)) for _ in range(24)]
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    mean, rstd = torch.zeros(rows, device=dev), torch.ones(rows, device=dev)
    y = torch.empty(rows, D, device=dev, dtype=BF16)
    i = [0]
    def fwd():
        x, _, _ = bufs[i[0] % 24]; i[0] += 1
        ops.layernorm_fwd(x, g, b, 1e-6, y=y, mean=mean, rstd=rstd)
    print(f"rows {rows}: fwd {t(fwd):6.2f} us  ({rows*D*4/1e6:.1f} MB)")
    for n_part in (128, 256, 512, 1024, 2048):
        part = torch.empty((2, n_part, D), device=dev)
        def bwd():
            x, dy, add = bufs[i[0] % 24]; i[0] += 1
            ops.layernorm_bwd(dy, x, g, mean, rstd, part, dx=y, dx_add=add)
        print(f"   bwd n_part {n_part:5d}: {t(bwd):6.2f} us  ({rows*D*8/1e6:.1f} MB + partials {2*n_part*D*4/1e6:.1f} MB)")
# adam
n = 221_714_692
p, m, v, gr = (torch.zeros(n, device=dev) for _ in range(4))
sh = torch.empty(171_000_000, device=dev, dtype=BF16)
hy = torch.tensor([1e-3, .9, .999, 1e-8, 0., 10., 31.6, 1.], device=dev)
print(f"adam: {t(lambda: ops.adam_step(p, m, v, gr, sh, 171_000_000, hy), 5):8.1f} us")
