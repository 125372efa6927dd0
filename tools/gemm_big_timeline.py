"""Per-workgroup stamps {entry, first tiles landed, main loop done, stores issued} of the lab library's 256x256 kernel (variant 13; run with
MEMEHIP_LIB_F16=.../libmemehip_lab_f16.so) or of the product's 128x128 kernel (GEMM_VARIANT=-2 or the product library) on square and
config-5 shapes; rate against torch.matmul."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multimodal_propaganda_meme_classification_amd import ops, _lib

T16 = torch.float16
lib = _lib.load("fp16")
if hasattr(lib, "mh_gemm_set_variant"):      # the lab library: variant 13 = the 256x256 kernel (GEMM_VARIANT=-2 for the product's)
    lib.mh_gemm_set_variant(int(os.environ.get("GEMM_VARIANT", "13")))
buf = torch.zeros(4 * 8192, dtype=torch.int64, device="cuda")
shapes = [(8192, 8192, 8192, 0), (18464, 3072, 1024, 0), (18464, 4096, 1024, 0), (18464, 4096, 1024, 1), (22688, 3072, 1024, 1)]
for (M, N, K, RES) in shapes:
    A = (torch.rand((M, K), device="cuda") * 2 - 1).to(T16)
    B = (torch.rand((N, K), device="cuda") * 2 - 1).to(T16)
    C = torch.empty((M, N), dtype=T16, device="cuda")
    R = (torch.rand((M, N), device="cuda") * 2 - 1).to(T16) if RES else None
    bias = torch.rand((N,), device="cuda") if RES else None
    ps = [ops.Gemm(A, B, C, M, N, K, K, K, N, residual=R, bias=bias)]

    def run():
        ops.gemm_grouped(ps, False, False)

    def lib_run():
        torch.matmul(A, B.t(), out=C)
    res = {}
    for nm, fn in (("memehip", run), ("torch", lib_run)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 5)
        res[nm] = float(np.median(ts))
    fl = 2.0 * M * N * K
    _lib.check(lib.mh_gemm_set_trace(buf.data_ptr()), "trace")
    buf.zero_()
    run()
    torch.cuda.synchronize()
    _lib.check(lib.mh_gemm_set_trace(None), "trace off")
    t = buf.view(-1, 4).cpu().numpy().astype(np.float64)
    t = t[t[:, 0] > 0]
    us = (t - t[:, 0].min()) / 100.0
    fill, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    print(f"M{M} N{N} K{K}{' +bias+residual' if RES else ''}: memehip {res['memehip']:7.1f} us ({fl / res['memehip'] / 1e6:6.0f} TFLOP/s)  torch {res['torch']:7.1f} us ({fl / res['torch'] / 1e6:6.0f}) | "
          f"{len(t)} workgroups: fill med {np.median(fill):5.2f}  main loop med {np.median(loop):6.2f} (p10 {np.percentile(loop, 10):6.2f} p90 {np.percentile(loop, 90):6.2f}) "
          f"= {np.median(loop) * 2400 / (K / 64):5.0f} clk per K tile  epilogue med {np.median(epi):5.2f} (p90 {np.percentile(epi, 90):5.2f})  last store {us[:, 3].max():6.1f} us", flush=True)
