"""Run three grouped GEMM cases a few times with one variant (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_propaganda_meme_classification_amd import ops, _lib

dev = torch.device("cuda")
BF16 = torch.bfloat16
Tt, Ti, D, I = 4096, 6304, 768, 3072
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF16)
v = int(sys.argv[1]) if len(sys.argv) > 1 else 1
_lib.load().mh_gemm_set_variant(v)
# fwd ffn2 (K=3072, plain epilogue), dgrad ffn1 (tr reads on B), wgrad
ps = [ops.Gemm(rnd(T, I), rnd(D, I), torch.empty((T, D), dtype=BF16, device=dev), T, D, I, I, I, D) for T in (Tt, Ti)]
pd = [ops.Gemm(rnd(T, I), rnd(I, D), torch.empty((T, D), dtype=BF16, device=dev), T, D, I, I, D, D) for T in (Tt, Ti)]
pw = [ops.Gemm(rnd(T, I), rnd(T, D), torch.empty((I, D), dtype=torch.float32, device=dev), I, D, T, I, D, D) for T in (Tt, Ti)]
for _ in range(3):
    ops.gemm_grouped(ps, False, False)
    ops.gemm_grouped(pd, False, True)
    ops.gemm_grouped(pw, True, True)
torch.cuda.synchronize()
