"""Stream-K GEMM against the tile-per-workgroup kernel in ONE process: bit-identity and time on the path's grouped shapes
(config 3 row counts) and on ragged / packed cases.   usage: gemm_sk_check.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_propaganda_meme_classification_amd import ops, _lib

dev = torch.device("cuda")
T16 = torch.float16
lib = _lib.load("fp16")
MODE = int(os.environ.get("SK_MODE", "2"))
ops.ensure_streamk(lib, "fp16", dev, mode=MODE)
ws = ops._STREAMK_WS[("fp16", torch.cuda.current_device())]
g = torch.Generator(device="cuda").manual_seed(0)


def run(probs, bk, sk):
    _lib.check(lib.mh_gemm_set_streamk(ws.data_ptr() if sk else None, MODE if sk else 0), "set")
    ops.gemm_grouped(probs, False, bk)


def case(name, shapes, bk, live=None, epi=False):
    """shapes: list of (M, N, K); bk: B is [K][N]"""
    tensors = []
    for (M, N, K) in shapes:
        A = (torch.randn((M, K), device=dev, generator=g) * 0.5).to(T16)
        Bm = (torch.randn((K, N) if bk else (N, K), device=dev, generator=g) * 0.05).to(T16)
        bias = torch.randn(N, device=dev, generator=g) if epi else None
        res = (torch.randn((M, N), device=dev, generator=g)).to(T16) if epi else None
        tensors.append((A, Bm, bias, res))
    rows = torch.tensor([live], dtype=torch.int32, device=dev) if live is not None else None
    outs = []
    for sk in (False, True):
        Cs = [torch.full((M, N), 3.0, dtype=T16, device=dev) for (M, N, K) in shapes]
        probs = [ops.Gemm(A, Bm, C, M, N, K, K, (N if bk else K), N, bias=bias, residual=res, rows_dev=(rows if i == 0 else None))
                 for i, ((M, N, K), (A, Bm, bias, res), C) in enumerate(zip(shapes, tensors, Cs))]
        for _ in range(3):
            run(probs, bk, sk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for rep in range(5):
            e0.record()
            for _ in range(10):
                run(probs, bk, sk)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        outs.append((Cs, sorted(ts)[2]))
    diff = max(float((a[: (live if (live is not None and i == 0) else a.shape[0])].float() - b[: (live if (live is not None and i == 0) else b.shape[0])].float()).abs().max())
               for i, (a, b) in enumerate(zip(outs[0][0], outs[1][0])))
    same = all(torch.equal(a[: (live if (live is not None and i == 0) else a.shape[0])], b[: (live if (live is not None and i == 0) else b.shape[0])])
               for i, (a, b) in enumerate(zip(outs[0][0], outs[1][0])))
    flags = ws[512 * 512 * 32 * 4:].view(torch.int32)
    print(f"{name:28s} tiles-per-wg {outs[0][1]:7.1f} us  stream-K {outs[1][1]:7.1f} us ({(outs[1][1] / outs[0][1] - 1) * 100:+.1f} %)  "
          f"bit-identical: {same} (max diff {diff:.2e})  flags clean: {int(flags[:512].abs().sum()) == 0}  timeout: {int(flags[512])}", flush=True)
    assert diff < 2e-2


Ti, Tt = 6304, 2096
case("fwd qkv (1206 tiles)", [(Ti, 2304, 768), (Tt, 2304, 768)], False, epi=True)
case("fwd out (402)", [(Ti, 768, 768), (Tt, 768, 768)], False, epi=True)
case("fwd ffn1 (1608)", [(Ti, 3072, 768), (Tt, 3072, 768)], False, epi=True)
case("fwd ffn2 (402, K 3072)", [(Ti, 768, 3072), (Tt, 768, 3072)], False, epi=True)
case("dgrad ffn2 (1608)", [(Ti, 3072, 768), (Tt, 3072, 768)], True)
case("dgrad ffn1 (402, K 3072)", [(Ti, 768, 3072), (Tt, 768, 3072)], True)
case("dgrad qkv (402, K 2304)", [(Ti, 768, 2304), (Tt, 768, 2304)], True)
case("packed text rows", [(4096, 768, 3072), (Ti, 768, 3072)], False, live=2093, epi=True)
case("one live row", [(4096, 768, 3072)], False, live=1)
case("ragged single", [(1000, 1024, 4096)], False, epi=True)
case("config 5 ffn1", [(18464, 4096, 1024), (8192, 4096, 1024)], False, epi=True)
