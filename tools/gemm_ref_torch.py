"""Reference point: what torch.matmul (hipBLASLt / rocBLAS) reaches on the path's GEMM shapes, next to the grouped kernel."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_propaganda_meme_classification_amd import ops

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

T = 6304
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    x = torch.randn((T, K), device="cuda").to(torch.bfloat16)
    w = torch.randn((N, K), device="cuda").to(torch.bfloat16)
    dy = torch.randn((T, N), device="cuda").to(torch.bfloat16)
    y = torch.empty((T, N), device="cuda", dtype=torch.bfloat16)
    dx = torch.empty((T, K), device="cuda", dtype=torch.bfloat16)
    dw = torch.empty((N, K), device="cuda", dtype=torch.float32)
    fl = 2.0 * T * N * K
    t_f = timeit(lambda: torch.matmul(x, w.t(), out=y))
    t_d = timeit(lambda: torch.matmul(dy, w, out=dx))
    t_w = timeit(lambda: torch.matmul(dy.t(), x))
    m_f = timeit(lambda: ops.gemm_grouped([ops.Gemm(x, w, y, T, N, K, K, K, N)], False, False))
    m_d = timeit(lambda: ops.gemm_grouped([ops.Gemm(dy, w, dx, T, K, N, N, K, K)], False, True))
    m_w = timeit(lambda: ops.gemm_grouped([ops.Gemm(dy, x, dw, N, K, T, N, K, K)], True, True))
    print(f"T={T} N={N} K={K}: torch fwd {fl/t_f/1e6:7.0f} dgrad {fl/t_d/1e6:7.0f} wgrad {fl/t_w/1e6:7.0f} TF | "
          f"memehip fwd {fl/m_f/1e6:7.0f} dgrad {fl/m_d/1e6:7.0f} wgrad(f32 out) {fl/m_w/1e6:7.0f} TF", flush=True)
