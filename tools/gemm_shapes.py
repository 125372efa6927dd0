"""Per-shape GEMM table of config 3: the 4 forward, 4 dgrad and the weight-gradient launches of a layer pair at
the benchmark's row counts (text: 2 096 live rows of a 4 096-row packed operand, read from rows_dev as in the step; image: 6 304),
with the step's epilogues, timed the same way for every candidate: N interleaved rounds in ONE process, HIP events around 10
back-to-back launches, median over rounds.

    python tools/gemm_shapes.py [--cands product,v10,v12,v3,torch] [--csv profiles/r04_gemm_shapes.csv]

candidates: `product` = the kernel the product library dispatches; `vN` = lab variant N (csrc/lab/memehip_lab.h; needs the lab build
loaded: MEMEHIP_LIB_F16=.../libmemehip_lab_f16.so after `make -C csrc LAB=1`): v10 = epilogue straight from transposed accumulators,
v12 = v10 + the 128x256x32 tile wherever legal ("wide"), v3 = 256x128 ping-pong ring ("pp"), v9 = persistent;

`torch` = torch.matmul (hipBLASLt / rocBLAS) on the same operands: the two towers' problems as two calls (a library has no grouped
launch) WITHOUT the fused epilogue (bias / GELU / residual would be further launches) -- an upper bound for what the library gives.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from multimodal_propaganda_meme_classification_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cands", default="product,torch")
ap.add_argument("--csv", default="")
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--live", type=int, default=0)
ap.add_argument("--config", type=int, default=3, help="3: ViT-B/16 + BERT-base rows / widths; 5: CLIP ViT-L/14@336 + BERT-large (S = 256)")
args = ap.parse_args()

dev = torch.device("cuda")
T16 = torch.float16 if args.dtype == "fp16" else torch.bfloat16
if args.config == 5:
    Tt_alloc, Ti, D, I = 8192, 18464, 1024, 4096
else:
    Tt_alloc, Ti, D, I = 4096, 6304, 768, 3072
live = args.live or (4224 if args.config == 5 else 2096)
rows_dev = torch.tensor([live], dtype=torch.int32, device=dev)


def rnd(*s, scale=0.5):
    return (torch.randn(*s, device=dev) * scale).to(T16)


def f32(*s):
    return torch.randn(*s, device=dev)


def fwd(N, K, **epi):
    """y[T, N] = x[T, K] W[N, K]^T  (+ the step's epilogue for this projection)"""
    ps, mm = [], []
    for T, packed in ((Tt_alloc, True), (Ti, False)):
        x, w, y = rnd(T, K), rnd(N, K, scale=0.05), torch.empty((T, N), dtype=T16, device=dev)
        kw = dict(bias=f32(N))
        if epi.get("residual"):
            kw["residual"] = rnd(T, N)
        if epi.get("gelu"):
            kw.update(aux=torch.empty((T, N), dtype=T16, device=dev), gelu=True, deriv_aux=True)
        if packed:
            kw["rows_dev"] = rows_dev
        ps.append(ops.Gemm(x, w, y, T, N, K, K, K, N, **kw))
        Tl = live if packed else T
        mm.append((x[:Tl], w, y[:Tl]))
    flops = 2.0 * (live + Ti) * N * K
    return ps, False, False, flops, [lambda a=a, b=b, c=c: torch.matmul(a, b.t(), out=c) for a, b, c in mm]


def dgrad(Nout, Kin, **epi):
    """dx[T, Kin] = dy[T, Nout] W[Nout, Kin]"""
    ps, mm = [], []
    for T, packed in ((Tt_alloc, True), (Ti, False)):
        dy, w, dx = rnd(T, Nout), rnd(Nout, Kin, scale=0.05), torch.empty((T, Kin), dtype=T16, device=dev)
        kw = {}
        if epi.get("mul"):
            kw.update(mul=rnd(T, Kin), deriv_aux=True)
        if epi.get("residual") and packed:        # the text tower adds the residual-branch gradient
            kw["residual"] = rnd(T, Kin)
        if packed:
            kw["rows_dev"] = rows_dev
        ps.append(ops.Gemm(dy, w, dx, T, Kin, Nout, Nout, Kin, Kin, **kw))
        Tl = live if packed else T
        mm.append((dy[:Tl], w, dx[:Tl]))
    flops = 2.0 * (live + Ti) * Nout * Kin
    return ps, False, True, flops, [lambda a=a, b=b, c=c: torch.matmul(a, b, out=c) for a, b, c in mm]


def wgrad(T, packed):
    """the four weight gradients of one tower in one launch (as in the step): dW[Nout, Kin] = dy[T, Nout]^T x[T, Kin], f32, + bias gradients"""
    ps, mm, fl = [], [], 0.0
    Tl = live if packed else T
    for (Nout, Kin) in ((D, I), (I, D), (D, D), (3 * D, D)):
        dy, x = rnd(T, Nout), rnd(T, Kin)
        dw, db = torch.empty((Nout, Kin), dtype=torch.float32, device=dev), torch.empty(Nout, device=dev)
        kw = dict(rows_dev=rows_dev) if packed else {}
        ps.append(ops.Gemm(dy, x, dw, Nout, Kin, T, Nout, Kin, Kin, rowsum=db, alpha=1.0 / 1024, **kw))
        fl += 2.0 * Tl * Nout * Kin
        mm.append((dy[:Tl], x[:Tl]))
    return ps, True, True, fl, [lambda a=a, b=b: torch.matmul(a.t(), b) for a, b in mm]


cases = {
    f"fwd_qkv   N{3 * D} K{D}": fwd(3 * D, D),
    f"fwd_out   N{D} K{D}": fwd(D, D, residual=True),
    f"fwd_ffn1  N{I} K{D}": fwd(I, D, gelu=True),
    f"fwd_ffn2  N{D} K{I}": fwd(D, I, residual=True),
    f"dgrad_ffn2 N{I} K{D}": dgrad(D, I, mul=True),
    f"dgrad_ffn1 N{D} K{I}": dgrad(I, D, residual=True),
    f"dgrad_out  N{D} K{D}": dgrad(D, D),
    f"dgrad_qkv  N{D} K{3 * D}": dgrad(3 * D, D, residual=True),
    "wgrad_text  4 problems": wgrad(Tt_alloc, True),
    "wgrad_image 4 problems": wgrad(Ti, False),
}

kind = "fp16" if T16 == torch.float16 else "bf16"
lib = _lib.load(kind)
cands = args.cands.split(",")


def run(name, cand):
    ps, ak, bk, fl, mms = cases[name]
    if cand == "torch":
        fn = lambda: [m() for m in mms]
    else:
        if hasattr(lib, "mh_gemm_set_variant"):
            lib.mh_gemm_set_variant(int(cand[1:]) if cand.startswith("v") else -2)
        elif cand.startswith("v"):
            raise SystemExit("lab variants need the lab build: MEMEHIP_LIB_F16=<repo>/multimodal_propaganda_meme_classification_amd/libmemehip_lab_f16.so")
        fn = lambda: ops.gemm_grouped(ps, ak, bk)
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100.0      # us per launch (ms / 10 launches * 1000)


res = {}
for r in range(args.rounds):
    for name in cases:
        for c in cands:
            res.setdefault((name, c), []).append(run(name, c))
if hasattr(lib, "mh_gemm_set_variant"):
    lib.mh_gemm_set_variant(-2)

rows = ["shape,candidate,us_median,us_min,TFLOPs_median"]
tot = {c: 0.0 for c in cands}
for name in cases:
    fl = cases[name][3]
    line = f"{name:24s}"
    for c in cands:
        xs = sorted(res[(name, c)])
        med, mn = xs[len(xs) // 2], xs[0]
        tot[c] += med
        line += f" | {c}: {med:7.1f} us {fl / med / 1e6:6.0f} TF"
        rows.append(f"{name.strip().replace('  ', ' ')},{c},{med:.2f},{mn:.2f},{fl / med / 1e6:.1f}")
    print(line, flush=True)
print("sum per layer pair (us):", {c: round(t, 1) for c, t in tot.items()}, " x12 =", {c: round(12 * t / 1e3, 3) for c, t in tot.items()}, "ms")
if args.csv:
    os.makedirs(os.path.dirname(args.csv), exist_ok=True)
    with open(args.csv, "w") as f:
        f.write(f"# tools/gemm_shapes.py --cands {args.cands} --dtype {args.dtype} (live text rows {live}, image rows {Ti}); "
                f"{torch.cuda.get_device_name(0)}\n")
        f.write("\n".join(rows) + "\n")
