"""Where a grouped GEMM launch of config 3 spends its time, per workgroup: the default kernel's 100-MHz stamps (mh_gemm_set_trace)
{entry, first K stage landed, main loop done, epilogue stores issued} for the four forward shapes and two dgrad shapes at the
benchmark's row counts (2096 live text rows + 6304 image rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
os.environ.setdefault("TT", "2096")
import tools.gemm_ab as ab          # noqa: E402  (defines the cases; exits before timing when imported)
from multimodal_propaganda_meme_classification_amd import ops, _lib  # noqa: E402

lib = _lib.load()
if hasattr(lib, "mh_gemm_set_variant"):
    lib.mh_gemm_set_variant(int(os.environ.get("GEMM_VARIANT", "-2")))
buf = torch.zeros(4 * 4096, dtype=torch.int64, device="cuda")
for name in ("fwd qkv", "fwd out", "fwd ffn1", "fwd ffn2", "dgrad ffn2", "dgrad qkv"):
    ps, ak, bk, fl = ab.cases[name]
    for _ in range(3):
        ops.gemm_grouped(ps, ak, bk)
    torch.cuda.synchronize()
    _lib.check(lib.mh_gemm_set_trace(buf.data_ptr()), "trace")
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm_grouped(ps, ak, bk)
    e1.record()
    torch.cuda.synchronize()
    _lib.check(lib.mh_gemm_set_trace(None), "trace off")
    t = buf.view(-1, 4).cpu().numpy().astype(np.float64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0                              # 100 MHz -> microseconds
    fill, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    order = np.argsort(us[:, 0])
    starts = us[order, 0]
    span = us[:, 3].max()
    first_wave = starts[:512].max() if len(starts) >= 512 else starts.max()
    print(f"{name:11s} {len(t):5d} workgroups, event time {e0.elapsed_time(e1) * 1e3:6.1f} us, first entry -> last store {span:6.1f} us | "
          f"fill med {np.median(fill):5.2f} (p90 {np.percentile(fill, 90):5.2f})  main loop med {np.median(loop):5.2f} (p10 {np.percentile(loop, 10):5.2f} p90 {np.percentile(loop, 90):5.2f})  "
          f"epilogue med {np.median(epi):5.2f} (p90 {np.percentile(epi, 90):5.2f}) | first 512 entered by {first_wave:5.2f} us; "
          f"entries at {[round(float(x), 1) for x in np.percentile(starts, [10, 25, 50, 75, 90, 100])]}")
    # busy slots over time: how many workgroups are inside [entry, end] at each microsecond
    grid = np.arange(0, span, 1.0)
    busy = [(int(((us[:, 0] <= g_) & (us[:, 3] > g_)).sum()), int(((us[:, 1] <= g_) & (us[:, 2] > g_)).sum())) for g_ in grid]
    print("            resident / in-main-loop workgroups per us:", " ".join(f"{a}/{b}" for a, b in busy[::3]))
