"""Probe: one layer's forward GEMM chain for both towers -- grouped launches on ONE stream (what the engine does)
against the two towers as independent chains on TWO streams (the hardware interleaves their workgroups)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_propaganda_meme_classification_amd import ops

BF = torch.bfloat16
def mk(T):
    d = {}
    d["x"] = torch.randn((T, 768), device="cuda").to(BF)
    d["qkv"] = torch.empty((T, 2304), device="cuda", dtype=BF)
    d["a"] = torch.empty((T, 768), device="cuda", dtype=BF)
    d["h"] = torch.empty((T, 3072), device="cuda", dtype=BF)
    d["g"] = torch.empty((T, 3072), device="cuda", dtype=BF)
    d["f"] = torch.empty((T, 768), device="cuda", dtype=BF)
    d["Wqkv"] = (torch.randn((2304, 768), device="cuda") * 0.02).to(BF)
    d["Wo"] = (torch.randn((768, 768), device="cuda") * 0.02).to(BF)
    d["W1"] = (torch.randn((3072, 768), device="cuda") * 0.02).to(BF)
    d["W2"] = (torch.randn((768, 3072), device="cuda") * 0.02).to(BF)
    d["T"] = T
    return d
def chain(d):
    T = d["T"]
    return [ops.Gemm(d["x"], d["Wqkv"], d["qkv"], T, 2304, 768, 768, 768, 2304),
            ops.Gemm(d["qkv"], d["Wo"], d["a"], T, 768, 768, 2304, 768, 768),
            ops.Gemm(d["a"], d["W1"], d["g"], T, 3072, 768, 768, 768, 3072, aux=d["h"], gelu=True),
            ops.Gemm(d["g"], d["W2"], d["f"], T, 768, 3072, 3072, 3072, 768, residual=d["a"])]
img, txt = mk(6304), mk(2093)
ci, ct = chain(img), chain(txt)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def grouped():
    for a, b in zip(ct, ci):
        ops.gemm_grouped([a, b], False, False)
def two_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        for g in ci: ops.gemm_grouped([g], False, False)
    with torch.cuda.stream(s2):
        for g in ct: ops.gemm_grouped([g], False, False)
    cur.wait_stream(s1); cur.wait_stream(s2)
def sequential():
    for g in ci: ops.gemm_grouped([g], False, False)
    for g in ct: ops.gemm_grouped([g], False, False)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, fn in (("grouped, one stream", grouped), ("two streams", two_streams), ("separate, one stream", sequential)):
    # also as graphs
    g = torch.cuda.CUDAGraph()
    fn(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    print(f"{name:24s} eager {timeit(fn):8.1f} us   hipGraph {timeit(g.replay):8.1f} us", flush=True)
