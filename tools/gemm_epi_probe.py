"""Does the epilogue's extra traffic explain the in-step vs stand-alone gap of the FFN launches?  FFN-up forward with
(bias) / (bias + GELU + pre-activation copy), FFN-down dgrad with / without the gelu' operand; back to back (hot caches) and
with a 600 MB buffer touched between launches (cold L2 / Infinity Cache, closer to the step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_propaganda_meme_classification_amd import ops

dev = torch.device("cuda")
H16 = torch.bfloat16
Tt, Ti, D, I = 2093, 6304, 768, 3072
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(H16)


def case(kind):
    ps = []
    for T in (Tt, Ti):
        if kind.startswith("ffn1"):
            x, w, y = rnd(T, D), rnd(I, D), torch.empty((T, I), dtype=H16, device=dev)
            kw = dict(bias=torch.zeros(I, device=dev))
            if kind == "ffn1+gelu+aux":
                kw.update(gelu=True, aux=torch.empty((T, I), dtype=H16, device=dev))
            ps.append(ops.Gemm(x, w, y, T, I, D, D, D, I, **kw))
            lay = (False, False)
        elif kind.startswith("ffn2d"):      # dx[T, I] = dy[T, D] @ W2[D, I]  (* gelu'(pre))
            dy, w, dx = rnd(T, D), rnd(D, I), torch.empty((T, I), dtype=H16, device=dev)
            kw = dict(mul=rnd(T, I)) if kind == "ffn2d+mul" else {}
            ps.append(ops.Gemm(dy, w, dx, T, I, D, D, I, I, **kw))
            lay = (False, True)
        elif kind.startswith("out"):
            x, w, y = rnd(T, D), rnd(D, D), torch.empty((T, D), dtype=H16, device=dev)
            kw = dict(bias=torch.zeros(D, device=dev))
            if kind == "out+res":
                kw.update(residual=rnd(T, D))
            ps.append(ops.Gemm(x, w, y, T, D, D, D, D, D, **kw))
            lay = (False, False)
    fl = 2.0 * (Tt + Ti) * (I * D if not kind.startswith("out") else D * D)
    return ps, lay, fl


trash = torch.empty(300 << 20, dtype=torch.int16, device=dev)
for kind in ("ffn1", "ffn1+gelu+aux", "ffn2d", "ffn2d+mul", "out", "out+res"):
    ps, (ak, bk), fl = case(kind)
    for cold in (False, True):
        ts = []
        for _ in range(12):
            if cold:
                trash.add_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm_grouped(ps, ak, bk)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        us = ts[len(ts) // 2]
        print(f"{kind:16s} {'cold' if cold else 'hot ':4s} {us:7.1f} us  {fl / us / 1e6:7.1f} TF/s")
