import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_propaganda_meme_classification_amd import ops, _lib
dev = torch.device("cuda"); BF16 = torch.bfloat16
Tt, Ti, D, I = 4096, 6304, 768, 3072
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF16)
def wg(T, Nout, Kin):
    return ops.Gemm(rnd(T, Nout), rnd(T, Kin), torch.empty((Nout, Kin), dtype=torch.float32, device=dev), Nout, Kin, T, Nout, Kin, Kin,
                    rowsum=torch.empty(Nout, device=dev))
def fw(T, N, K):
    return ops.Gemm(rnd(T, K), rnd(N, K), torch.empty((T, N), dtype=BF16, device=dev), T, N, K, K, K, N, bias=torch.zeros(N, device=dev))
def dg(T, Nout, Kin):
    return ops.Gemm(rnd(T, Nout), rnd(Nout, Kin), torch.empty((T, Kin), dtype=BF16, device=dev), T, Kin, Nout, Nout, Kin, Kin)
def timeit(launches):
    def run():
        for ps, ak, bk in launches: ops.gemm_grouped(ps, ak, bk)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    return best * 1e3
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
_lib.load().mh_gemm_set_variant(variant)
print("variant", variant)
for name, mk, ak, bk in (("fwd qkv", lambda T: fw(T, 3 * D, D), False, False), ("fwd out", lambda T: fw(T, D, D), False, False),
                         ("fwd ffn1", lambda T: fw(T, I, D), False, False), ("fwd ffn2", lambda T: fw(T, D, I), False, False),
                         ("dgrad ffn2", lambda T: dg(T, D, I), False, True), ("dgrad ffn1", lambda T: dg(T, I, D), False, True),
                         ("dgrad out", lambda T: dg(T, D, D), False, True), ("dgrad qkv", lambda T: dg(T, 3 * D, D), False, True)):
    t, i = mk(Tt), mk(Ti)
    print(f"{name:12s} grouped {timeit([([t, i], ak, bk)]):7.1f} us   separate {timeit([([t], ak, bk), ([i], ak, bk)]):7.1f} us "
          f"(text {timeit([([t], ak, bk)]):6.1f} image {timeit([([i], ak, bk)]):6.1f})")
shapes = ((D, I), (I, D), (D, D), (3 * D, D))
wt = [wg(Tt, a, b) for a, b in shapes]; wi = [wg(Ti, a, b) for a, b in shapes]
print(f"wgrad 8 grouped {timeit([(wt + wi, True, True)]):7.1f}  text|image {timeit([(wt, True, True), (wi, True, True)]):7.1f}  "
      f"image|text {timeit([(wi, True, True), (wt, True, True)]):7.1f}  image-first grouped {timeit([(wi + wt, True, True)]):7.1f}")
# other splits: big ones (ffn) together, small ones together
print(f"wgrad split by size: ffn(4)|attn(4) {timeit([([wt[0], wt[1], wi[0], wi[1]], True, True), ([wt[2], wt[3], wi[2], wi[3]], True, True)]):7.1f}")
