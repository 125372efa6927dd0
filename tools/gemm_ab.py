"""A/B the GEMM staging variants on the real grouped shapes of config 3 (one process, interleaved)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_propaganda_meme_classification_amd import ops, _lib

dev = torch.device("cuda")
BF16 = torch.bfloat16
Tt, Ti, D, I = int(os.environ.get("TT", "4096")), 6304, 768, 3072


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(BF16)


def fwd_group(N, K):
    ps = []
    for T in (Tt, Ti):
        ps.append(ops.Gemm(rnd(T, K), rnd(N, K), torch.empty((T, N), dtype=BF16, device=dev), T, N, K, K, K, N,
                           bias=torch.zeros(N, device=dev)))
    return ps, False, False, 2.0 * (Tt + Ti) * N * K


def dgrad_group(Nout, Kin):
    ps = []
    for T in (Tt, Ti):
        ps.append(ops.Gemm(rnd(T, Nout), rnd(Nout, Kin), torch.empty((T, Kin), dtype=BF16, device=dev), T, Kin, Nout, Nout, Kin, Kin))
    return ps, False, True, 2.0 * (Tt + Ti) * Nout * Kin


def wgrad_group():
    ps = []
    fl = 0.0
    for T in (Tt, Ti):
        for (Nout, Kin) in ((D, I), (I, D), (D, D), (3 * D, D)):
            ps.append(ops.Gemm(rnd(T, Nout), rnd(T, Kin), torch.empty((Nout, Kin), dtype=torch.float32, device=dev), Nout, Kin, T,
                               Nout, Kin, Kin, rowsum=torch.empty(Nout, device=dev)))
            fl += 2.0 * T * Nout * Kin
    return ps, True, True, fl


cases = {"fwd qkv": fwd_group(3 * D, D), "fwd out": fwd_group(D, D), "fwd ffn1": fwd_group(I, D), "fwd ffn2": fwd_group(D, I),
         "dgrad ffn2": dgrad_group(D, I), "dgrad ffn1": dgrad_group(I, D), "dgrad out": dgrad_group(D, D), "dgrad qkv": dgrad_group(3 * D, D),
         "wgrad layer": wgrad_group()}


def run():
    variants = [int(v) for v in (sys.argv[1:] or ["0", "1"])]
    lib = _lib.load()
    res = {}
    for rnd_i in range(5):
        for name, (ps, ak, bk, fl) in cases.items():
            for v in variants:
                lib.mh_gemm_set_variant(v)
                for _ in range(2):
                    ops.gemm_grouped(ps, ak, bk)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.gemm_grouped(ps, ak, bk)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault((name, v), []).append(e0.elapsed_time(e1) / 10)
    tot = {v: 0.0 for v in variants}
    for name, (ps, ak, bk, fl) in cases.items():
        line = f"{name:12s}"
        for v in variants:
            ms = sorted(res[(name, v)])[len(res[(name, v)]) // 2]
            tot[v] += ms * (1 if name.startswith("wgrad") else 1)
            line += f"  v{v}: {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF"
        print(line)
    print("sum per layer (ms):", {v: round(t, 3) for v, t in tot.items()}, " x12 layers =", {v: round(12 * t, 2) for v, t in tot.items()})


if __name__ == "__main__":
    run()
