"""What cache state costs a grouped GEMM launch of config 3: everything hot (back-to-back repeats, what tools/gemm_shapes.py times), everything
evicted (a 1-GiB fill before the launch), only the weights cold (evict, then read the activations), only the activations cold.  In the step
the activations were just written by the previous kernel and the weights were last read a whole pass ago (342 MB of 16-bit weights per pass
against a 256-MB Infinity Cache).  One launch per measurement, HIP events, median of 9."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from multimodal_propaganda_meme_classification_amd import ops  # noqa: E402

dev = torch.device("cuda")
T16 = torch.float16
Tt, Ti, D, I = 2096, 6304, 768, 3072
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GiB


def rnd(*s, scale=0.5):
    return (torch.randn(*s, device=dev) * scale).to(T16)


def case(N, K, bk):
    ps, xs, ws = [], [], []
    for T in (Tt, Ti):
        x = rnd(T, K)
        w = rnd(K, N, scale=0.05) if bk else rnd(N, K, scale=0.05)
        y = torch.empty((T, N), dtype=T16, device=dev)
        ps.append(ops.Gemm(x, w, y, T, N, K, K, N if bk else K, N, bias=None if bk else torch.randn(N, device=dev)))
        xs.append(x)
        ws.append(w)
    return ps, bk, xs, ws, 2.0 * (Tt + Ti) * N * K


cases = {"fwd_qkv N2304 K768": case(3 * D, D, False), "fwd_out N768 K768": case(D, D, False), "fwd_ffn1 N3072 K768": case(I, D, False),
         "fwd_ffn2 N768 K3072": case(D, I, False), "dgrad_ffn2 N3072 K768": case(I, D, True), "dgrad_ffn1 N768 K3072": case(D, I, True)}


def once(ps, bk, prep):
    prep()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm_grouped(ps, False, bk)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for name, (ps, bk, xs, ws, fl) in cases.items():
    def hot():
        ops.gemm_grouped(ps, False, bk)

    def cold():
        flush.fill_(1.0)

    def w_cold():
        flush.fill_(1.0)
        for x in xs:
            x.float().sum()

    def x_cold():
        flush.fill_(1.0)
        for w in ws:
            w.float().sum()
    res = {}
    for tag, prep in (("hot", hot), ("all cold", cold), ("weights cold", w_cold), ("activations cold", x_cold)):
        ts = sorted(once(ps, bk, prep) for _ in range(9))
        res[tag] = ts[len(ts) // 2]
    print(f"{name:24s} " + " | ".join(f"{k}: {v:6.1f} us" for k, v in res.items()), flush=True)
