"""Launch plan of one fine-tune step: which HIP kernels run, on which buffers, in which order.

A ``Plan`` is built once per (batch, seq_len): every activation / gradient workspace is allocated
up front (288 GB of HBM: nothing is recomputed, nothing is freed) and every kernel launch is
recorded as a prepared C-ABI call.  Running a step is then a replay of prepared calls -- eagerly,
or captured once into a hipGraph (model.GraphedStep) so the ~220 launches cost one host call.

The two towers run in LOCKSTEP: layer l of the text encoder and layer l of the image encoder
issue the same kernel sequence, so every launch is a *grouped* launch carrying both towers'
problems: GEMMs (forward, dgrad; the weight gradients as one launch per tower), LayerNorms and
attention.  That is what fills 256 CUs at these sizes (768-wide projections give only 100-300
128x128 tiles per tower).  The text tower is padding-free: its launches are sized for B*S rows and
clamp themselves to the live row count on the device (see the "pk.*" buffers in _build).

Reference for the math: BertModel / timm ViT as called at
example_scripts/Multimodal_example_task2C.txt:175,183 and the step at :200-223.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import (MH_GEMM_ACCUM, MH_GEMM_DERIV_AUX, MH_GEMM_GELU, MH_GEMM_OUT_F32, MH_GEMM_QUICK_GELU, MhAttnProblem, MhColsumJob, MhGemmProblem, MhHeadGrads,
                   MhHeadParams, MhLnBwdJob, MhLnFwdJob)
from .config import Layout, ModelConfig

BF16, F32, I64 = torch.bfloat16, torch.float32, torch.int64
import os as _os
LN_PARTS = int(_os.environ.get("MEMEHIP_LN_PARTS", "512"))     # workgroups (= sets of dgamma/dbeta partials) per LayerNorm backward
DERIV_AUX = _os.environ.get("MEMEHIP_GEMM_DERIV_AUX", "1") != "0"     # FFN-up epilogue stores gelu'(x) for the backward (A/B: 0 = store x)


class Segment:
    """An ordered list of prepared launches."""

    def __init__(self, name: str, lib=None):
        self.name = name
        self.lib = lib if lib is not None else _lib.load()
        self.calls: List[Tuple] = []

    def c(self, fn_name: str, *args, tag: Optional[str] = None, work: float = 0.0, lane: int = 0):
        self.calls.append((getattr(self.lib, fn_name), args, fn_name, tag or fn_name, work, lane))

    def py(self, fn: Callable[[], None]):
        self.calls.append((None, fn, "py", "py", 0.0, 0))

    def wait(self, key: str):
        """On a multi-stream run: the main stream waits for the lane-1 work of segment `key`."""
        self.calls.append((None, key, "wait", "wait", 0.0, 0))

    def run(self, stream: int):
        for fn, args, name, _, _, _ in self.calls:
            if fn is None:
                if name == "py":
                    args()
            else:
                st = fn(*args, stream)
                if st != 0:
                    _lib.check(st, name)

    def run2(self, main: "torch.cuda.Stream", side: Optional["torch.cuda.Stream"], events: Dict[str, "torch.cuda.Event"]):
        """Two-stream replay (single GPU).  lane 1 = the weight-gradient GEMMs, which nothing later in the backward
        chain reads: they go to `side`, forked behind everything issued so far on `main`, and their completion event
        is stored under this segment's name for a later wait().  (A third stream for the image tower's LayerNorm /
        attention beside the text tower's was measured slower -- the cross-stream edges cost more than the overlap of
        those short kernels returned -- and removed; both towers now share grouped launches instead.)"""
        mp = main.cuda_stream
        sp = side.cuda_stream if side is not None else mp      # side=None: lane-1 work stays on the main stream
        forked1 = False
        for fn, args, name, _, _, lane in self.calls:
            if fn is None:
                if name == "py":
                    args()
                elif name == "wait":
                    if args in events:
                        main.wait_event(events.pop(args))
                continue
            if lane == 1 and side is not None:
                if not forked1:
                    e = torch.cuda.Event()
                    e.record(main)
                    side.wait_event(e)
                    forked1 = True
                st = fn(*args, sp)
            else:
                st = fn(*args, mp)
            if st != 0:
                _lib.check(st, name)
        if forked1:
            d = torch.cuda.Event()
            d.record(side)
            events[self.name] = d

    def run_timed(self, stream: int, tag: str, out: list, dyn_scale: float = 1.0):
        """Eager replay that brackets every launch tagged `tag` with events on the launch stream;
        appends (start_event, end_event, work) to `out`.  `work` of a launch is (fixed, per-live-row part at the
        maximum row count): dyn_scale = live rows / maximum rows of the packed text tower."""
        for fn, args, name, t, work, _ in self.calls:
            if isinstance(work, tuple):
                work = work[0] + work[1] * dyn_scale
            if fn is None:
                if name == "py":
                    args()
                continue
            if t == tag:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                st = fn(*args, stream)
                e1.record()
                out.append((e0, e1, work))
            else:
                st = fn(*args, stream)
            if st != 0:
                _lib.check(st, name)

    def __len__(self):
        return len(self.calls)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class Plan:
    def __init__(self, B: int, S: int, lib=None):
        self.B, self.S = B, S
        self.lib = lib
        self.buf: Dict[str, torch.Tensor] = {}
        self.keep: list = []                  # ctypes arrays referenced by prepared calls
        self.fwd = Segment("fwd", lib)
        self.loss = Segment("loss", lib)
        self.bwd: List[Segment] = []          # backward segments in execution order
        self.bucket_after: Dict[str, Tuple[int, int]] = {}   # segment name -> flat grad range complete after it
        self.n_launches = 0
        self.gemm_flops = 0.0                 # algorithmic GEMM FLOPs of one step (fwd + dgrad + wgrad), no row packed away
        self.gemm_flops_dyn = 0.0             # the part of it that scales with the live rows of the packed text tower
        self.packed = False


class Engine:
    def __init__(self, cfg: ModelConfig, layout: Layout, P: torch.Tensor, G: torch.Tensor, SH: torch.Tensor):
        self.cfg, self.layout = cfg, layout
        self.P, self.G, self.SH = P, G, SH
        self.dev = P.device
        assert P.is_cuda and G.is_cuda and SH.is_cuda, "memehip needs HIP device buffers (no CPU fallback)"
        self.kind = cfg.compute_dtype
        self.T16 = torch.float16 if self.kind == "fp16" else torch.bfloat16
        assert SH.dtype == self.T16
        self.lib = _lib.load(self.kind)
        self.gscale = cfg.stream_scale          # scale carried by the 16-bit gradient streams
        # padding-free text tower (rows with attention_mask == 0 are never computed); MEMEHIP_PACK_TEXT=0 for A/B runs
        import os
        self.pack_text = bool(getattr(cfg, "pack_text", True)) and os.environ.get("MEMEHIP_PACK_TEXT", "1") != "0"
        self.plans: Dict[Tuple[int, int], Plan] = {}
        # one byte per word-embedding row: has it ever received a gradient?  (rows that have not keep g = m = v = 0,
        # for which Adam is the identity: model.Adam skips them, mh_adam_step_rows)
        self.word_row_live = torch.zeros(cfg.text.vocab_size, dtype=torch.uint8, device=self.dev)
        # id index of the embedding-gradient kernel (first position / multiplicity per vocabulary row; self-restoring)
        self.word_first = torch.full((cfg.text.vocab_size,), 0x7fffffff, dtype=torch.int32, device=self.dev)
        self.word_count = torch.zeros(cfg.text.vocab_size, dtype=torch.int32, device=self.dev)
        # dynamic loss scale read by the fused loss launch (1 = none; a memehip.GradScaler attached to the model owns it)
        self.loss_scale = torch.ones(1, dtype=F32, device=self.dev)

    # ---- parameter / gradient views -----------------------------------------------------------------
    def _slice(self, flat: torch.Tensor, name: str, count: int = 1) -> torch.Tensor:
        s = self.layout.spec[name]
        return flat[s.offset:s.offset + s.numel * count]

    def p(self, name, count=1):
        return self._slice(self.P, name, count)

    def g(self, name, count=1):
        return self._slice(self.G, name, count)

    def w(self, name, count=1):
        s = self.layout.spec[name]
        assert s.offset + s.numel * count <= self.layout.n_shadow, name
        return self.SH[s.offset:s.offset + s.numel * count]

    # ---- call builders ---------------------------------------------------------------------------------
    def _gemm(self, plan: Plan, seg: Segment, probs: List[dict], a_k: bool, b_k: bool, lane: int = 0):
        if not probs:
            return
        n = len(probs)
        assert n <= _lib.MH_GEMM_MAX_GROUP
        arr = (MhGemmProblem * n)()
        for i, d in enumerate(probs):
            A, Bm, Cm = d["A"], d["B"], d["C"]
            M, N, K = d["M"], d["N"], d["K"]
            lda, ldb, ldc = d["lda"], d["ldb"], d["ldc"]
            assert A.dtype == self.T16 and Bm.dtype == self.T16 and Cm.dtype in (self.T16, F32)
            a_need = (K - 1) * lda + M if a_k else (M - 1) * lda + K
            b_need = (K - 1) * ldb + N if b_k else (N - 1) * ldb + K
            assert A.numel() >= a_need and Bm.numel() >= b_need and Cm.numel() >= (M - 1) * ldc + N, (M, N, K)
            for key, dt, need in (("bias", F32, N), ("residual", self.T16, (M - 1) * ldc + N), ("aux", self.T16, (M - 1) * ldc + N),
                                  ("mul", self.T16, (M - 1) * ldc + N), ("rowsum", F32, M)):
                t = d.get(key)
                assert t is None or (t.dtype == dt and t.numel() >= need), key
            e = arr[i]
            e.A, e.B, e.C = _ptr(A), _ptr(Bm), _ptr(Cm)
            e.bias, e.residual, e.aux = _ptr(d.get("bias")), _ptr(d.get("residual")), _ptr(d.get("aux"))
            e.mul, e.rowsum = _ptr(d.get("mul")), _ptr(d.get("rowsum"))
            e.M, e.N, e.K, e.lda, e.ldb, e.ldc = M, N, K, lda, ldb, ldc
            e.flags = (MH_GEMM_GELU if d.get("gelu") else 0) | (MH_GEMM_OUT_F32 if Cm.dtype == F32 else 0) | \
                      (MH_GEMM_ACCUM if d.get("accum") else 0) | (MH_GEMM_QUICK_GELU if d.get("quick") else 0) | \
                      (MH_GEMM_DERIV_AUX if d.get("deriv_aux") else 0)
            e.alpha = float(d.get("alpha", 1.0))
            if d.get("drop") is not None:
                e.drop_rng, e.drop_p, e.drop_stream = d["drop"]
                e.drop_rows = _ptr(d.get("drop_rows"))
            e.rows_dev = _ptr(d.get("rows_dev"))
        plan.keep.append(arr)
        fl = [2.0 * d["M"] * d["N"] * d["K"] for d in probs]
        dyn = float(sum(x for x, d in zip(fl, probs) if d.get("rows_dev") is not None))    # scales with the live rows
        fixed = float(sum(fl)) - dyn
        plan.gemm_flops += fixed + dyn
        plan.gemm_flops_dyn += dyn
        seg.c("mh_gemm_bf16_grouped", arr, n, int(a_k), int(b_k), tag=f"gemm<{int(a_k)},{int(b_k)}>", work=(fixed, dyn),
              lane=lane)

    @staticmethod
    def _fwd_prob(x, w, out, T, N, K, **kw):
        return dict(A=x, B=w, C=out, M=T, N=N, K=K, lda=K, ldb=K, ldc=N, **kw)

    @staticmethod
    def _dgrad_prob(dy, w, out, T, N_out, K_in, **kw):
        # out[T, K_in] = dy[T, N_out] @ w[N_out, K_in]
        return dict(A=dy, B=w, C=out, M=T, N=K_in, K=N_out, lda=N_out, ldb=K_in, ldc=K_in, **kw)

    def _wgrad_prob(self, dy, x, dw, db, T, N_out, K_in, **kw):
        # dw[N_out, K_in] = dy[T, N_out]^T @ x[T, K_in]   (dy carries the gradient-stream scale: alpha removes it)
        return dict(A=dy, B=x, C=dw, M=N_out, N=K_in, K=T, lda=N_out, ldb=K_in, ldc=K_in, rowsum=db,
                    alpha=1.0 / self.gscale, **kw)

    @staticmethod
    def _ln_groups(by_d):
        import os
        if os.environ.get("MEMEHIP_LN_GROUP", "1") == "0":      # A/B switch: one launch per LayerNorm
            return [(D, [j]) for D, js in by_d.items() for j in js]
        return list(by_d.items())

    def _ln_fwd_job(self, x, gname, bname, y, mean, rstd, rows, D, eps, y32=None, rows_dev=None):
        return dict(x=x, gamma=self.p(gname), beta=self.p(bname), y=y, y32=y32, mean=mean, rstd=rstd, rows=rows, D=D,
                    eps=float(eps), rows_dev=rows_dev)

    def _ln_fwd(self, plan: Plan, seg: Segment, jobs: List[Optional[dict]]):
        """LayerNorms that sit at the same point of the lockstep schedule go out as ONE grouped launch per width."""
        by_d: Dict[int, List[dict]] = {}
        for j in jobs:
            if j is not None:
                by_d.setdefault(j["D"], []).append(j)
        for D, js in self._ln_groups(by_d):
            assert len(js) <= _lib.MH_LN_MAX_JOBS
            arr = (MhLnFwdJob * len(js))()
            for e, j in zip(arr, js):
                assert j["x"].numel() >= j["rows"] * D and j["y"].numel() >= j["rows"] * D
                e.x, e.gamma, e.beta, e.y, e.y_f32 = _ptr(j["x"]), _ptr(j["gamma"]), _ptr(j["beta"]), _ptr(j["y"]), _ptr(j["y32"])
                e.mean, e.rstd, e.rows, e.eps = _ptr(j["mean"]), _ptr(j["rstd"]), j["rows"], j["eps"]
                e.rows_dev = _ptr(j["rows_dev"])
            plan.keep.append(arr)
            seg.c("mh_layernorm_fwd_grouped", arr, len(js), D)

    def _ln_bwd_job(self, plan, dy, x, gname, bname, mean, rstd, dx, rows, D, dx_add=None, dx_drop=None, drop=None,
                    rows_dev=None, drop_rows=None):
        part = torch.empty((2, LN_PARTS, D), dtype=F32, device=self.dev)
        plan.buf[f"lnpart.{gname}"] = part
        plan._ln_jobs.setdefault(D, []).append((part, self.g(gname), self.g(bname)))
        rng, p_, sid = drop if (drop is not None and dx_drop is not None) else (None, 0.0, 0)
        return dict(dy=dy, x=x, gamma=self.p(gname), mean=mean, rstd=rstd, dx_add=dx_add, dx=dx, part=part,
                    dx_drop=dx_drop if rng is not None else None, rng=rng, p=float(p_), sid=int(sid), rows=rows, D=D,
                    rows_dev=rows_dev, drop_rows=drop_rows if rng is not None else None)

    def _ln_bwd(self, plan: Plan, seg: Segment, jobs: List[Optional[dict]]):
        by_d: Dict[int, List[dict]] = {}
        for j in jobs:
            if j is not None:
                by_d.setdefault(j["D"], []).append(j)
        for D, js in self._ln_groups(by_d):
            assert len(js) <= _lib.MH_LN_MAX_JOBS
            arr = (MhLnBwdJob * len(js))()
            for e, j in zip(arr, js):
                n = j["rows"] * D
                assert j["dy"].numel() >= n and j["x"].numel() >= n and j["dx"].numel() >= n
                assert j["dx_add"] is None or j["dx_add"].numel() >= n
                e.dy, e.x, e.gamma, e.mean, e.rstd = _ptr(j["dy"]), _ptr(j["x"]), _ptr(j["gamma"]), _ptr(j["mean"]), _ptr(j["rstd"])
                e.dx_add, e.dx, e.part, e.dx_drop, e.rng = _ptr(j["dx_add"]), _ptr(j["dx"]), _ptr(j["part"]), _ptr(j["dx_drop"]), j["rng"]
                e.n_part, e.rows, e.drop_p, e.drop_stream = LN_PARTS, j["rows"], j["p"], j["sid"]
                e.rows_dev, e.drop_rows = _ptr(j["rows_dev"]), _ptr(j["drop_rows"])
            plan.keep.append(arr)
            seg.c("mh_layernorm_bwd_grouped", arr, len(js), D)

    def _attn(self, plan: Plan, seg: Segment, probs: List[Optional[dict]], backward: bool):
        """Both towers' attention of a layer pair as one grouped call (one launch per kernel when the pair fits the
        dual kernels, see include/memehip.h: mh_attn_fwd_grouped)."""
        probs = [d for d in probs if d is not None]
        if not probs:
            return
        arr = (MhAttnProblem * len(probs))()
        for e, d in zip(arr, probs):
            for k in ("qkv", "key_mask", "out", "lse", "dout", "delta", "dqkv", "cu", "row_map"):
                setattr(e, k, _ptr(d.get(k)))
            rng, p_, sid = d.get("drop") or (None, 0.0, 0)
            e.rng, e.drop_p, e.drop_stream = rng, float(p_), int(sid)
            e.B, e.S, e.H = d["B"], d["S"], d["H"]
            n = d["B"] * d["S"] * d["H"] * 64
            assert d["qkv"].numel() >= 3 * n and d["out"].numel() >= n and d["lse"].numel() >= d["B"] * d["H"] * d["S"]
            if backward:
                assert d["dout"].numel() >= n and d["dqkv"].numel() >= 3 * n and d["delta"].numel() >= d["B"] * d["H"] * d["S"]
        plan.keep.append(arr)
        seg.c("mh_attn_bwd_grouped" if backward else "mh_attn_fwd_grouped", arr, len(probs))

    def before_backward(self, plan: Plan):
        """Each plan re-zeroes the word-embedding gradient rows IT touched last time (the table gradient is dense,
        only touched rows are ever non-zero).  When the previous backward ran under a different plan (another
        batch size / sequence length), clear that plan's rows first so no stale row survives the switch."""
        last = getattr(self, "_last_bwd_plan", None)
        if last is not None and last is not plan:
            gword = self.g("bert.embeddings.word_embeddings.weight")
            for key in ("prev_ids_all", "prev_ids"):
                prev = last.buf.get(key)
                if prev is not None:
                    st = self.lib.mh_zero_rows_f32(prev.data_ptr(), gword.data_ptr(), prev.numel(), self.cfg.text.hidden,
                                                   self.cfg.text.vocab_size, torch.cuda.current_stream().cuda_stream)
                    _lib.check(st, "mh_zero_rows_f32")
                    prev.zero_()
        self._last_bwd_plan = plan

    # ---- plan ----------------------------------------------------------------------------------------------
    def plan(self, B: int, S: int, training: bool = True, gather_world: int = 0, features=False) -> Plan:
        """Plans differ between train and eval only when some dropout probability is non-zero.
        features = True: the plan stops at the pooled tower features (text pooled row | ViT class token, f32) and its
        backward starts from their gradient: the late-fusion head is then whatever the caller builds on top.
        features = "sequence": the same with the towers' whole last hidden states ([B,S,Dt] and [B,Nt,Di], f32) as the
        boundary -- for poolings that read every position (max / mean / attention / conv1d); the reference pools over the
        padded positions too, so this plan computes them (no row packing).
        gather_world = W > 0 (data parallel): the embedding-table gradients are built from the all-gathered token ids
        and embedding-gradient rows of all W ranks (6 MB per rank) instead of all-reducing the dense 196-MB table."""
        cfg = self.cfg
        has_drop = (cfg.text.hidden_dropout > 0 or cfg.text.attention_dropout > 0 or cfg.head_dropout > 0)
        features = features if features == "sequence" else bool(features)
        pack = self.pack_text and features != "sequence"
        key = (B, S, bool(training and has_drop), int(gather_world), pack, features)
        if key not in self.plans:
            self.plans[key] = self._build(B, S, key[2], key[3], key[4], key[5])
        return self.plans[key]

    def _build(self, B: int, S: int, dropout_on: bool = False, gather_world: int = 0, pack: bool = False,
               features=False) -> Plan:
        cfg, t, v = self.cfg, self.cfg.text, self.cfg.image
        if S > t.max_position:
            raise ValueError(f"sequence length {S} > max_position {t.max_position}")
        if B > 1024:
            raise ValueError("batch > 1024 not supported by the loss kernel")
        pl = Plan(B, S, self.lib)
        pl._ln_jobs = {}
        BF16 = self.T16          # every 16-bit buffer below uses the configured storage type
        # dropout sites: (rng words pointer, p, site id); None when off.  rng = int32[4] {seed_lo, seed_hi, step, -}
        rng_t = torch.zeros(4, dtype=torch.int32, device=self.dev)
        pl.buf["rng"] = rng_t
        pl.dropout_on = dropout_on
        p_h = t.hidden_dropout if dropout_on else 0.0
        p_a = t.attention_dropout if dropout_on else 0.0
        p_head = cfg.head_dropout if dropout_on else 0.0

        def site(p_, sid):
            return (rng_t.data_ptr(), float(p_), int(sid)) if p_ > 0 else None

        def site_args(p_, sid):
            return (rng_t.data_ptr(), float(p_), int(sid)) if p_ > 0 else (None, 0.0, 0)
        dev = self.dev
        Dt, It, Ht, Lt = t.hidden, t.intermediate, t.heads, t.layers
        Di, Ii, Hi, Li = v.hidden, v.intermediate, v.heads, v.layers
        Np, Nt = v.n_patches, v.n_tokens
        Kp = v.channels * v.patch * v.patch
        Tt, Ti = B * S, B * Nt
        P_, Cn = cfg.proj, cfg.num_classes

        def alloc(name, shape, dtype=BF16, zero=False):
            tns = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=dev)
            pl.buf[name] = tns
            return tns

        # static inputs
        ids = alloc("ids", (B, S), I64, zero=True)
        mask = alloc("mask", (B, S), I64, zero=True)
        image = alloc("image", (B, v.channels, v.image_size, v.image_size), F32, zero=True)
        labels = alloc("labels", (B,), I64, zero=True)
        prev_ids = alloc("prev_ids", (B, S), I64, zero=True)

        # ------------------------------------------------------------------ forward ------------------
        f = pl.fwd
        TXT, IMG = "bert.", "image_model."
        # text embeddings (dense [B*S] rows: the tables are indexed by position)
        pool_index = 0 if cfg.pool == "cls" else S - 1
        pre0 = alloc("t.pre0", (Tt, Dt))
        x0d = alloc("t.x0", (Tt, Dt))
        m0, r0 = alloc("t.m0", (Tt,), F32), alloc("t.r0", (Tt,), F32)
        type0 = self.p(TXT + "embeddings.token_type_embeddings.weight")[:Dt] if t.type_vocab > 0 else None
        f.c("mh_bert_embed_fwd", _ptr(ids), _ptr(self.p(TXT + "embeddings.word_embeddings.weight")),
            _ptr(self.p(TXT + "embeddings.position_embeddings.weight")), _ptr(type0),
            _ptr(self.p(TXT + "embeddings.LayerNorm.weight")), _ptr(self.p(TXT + "embeddings.LayerNorm.bias")),
            _ptr(pre0), _ptr(x0d), _ptr(m0), _ptr(r0), B, S, Dt, t.vocab_size, float(t.ln_eps), *site_args(p_h, 1))
        # padding-free text tower: the encoder layers see only the rows that matter (attention_mask != 0, plus the
        # pooled position), packed back to back.  The row count lives on the device (n_rows): every text launch below
        # is sized for B*S rows and clamps itself at run time, so one plan / one hipGraph serves every batch.
        pl.packed = pack
        if pack:
            I32 = torch.int32
            cu, row_map = alloc("pk.cu", (B + 1,), I32, zero=True), alloc("pk.row_map", (Tt,), I32, zero=True)
            inv_map, pmask = alloc("pk.inv_map", (Tt,), I32, zero=True), alloc("pk.pmask", (Tt,), I64, zero=True)
            pool_rows, n_rows = alloc("pk.pool_rows", (B,), I32, zero=True), alloc("pk.n_rows", (1,), I32, zero=True)
            f.c("mh_pack_plan", _ptr(mask), B, S, pool_index, _ptr(cu), _ptr(row_map), _ptr(inv_map), _ptr(pmask),
                _ptr(pool_rows), _ptr(n_rows))
            xt = [alloc("t.x0p", (Tt, Dt))]
            f.c("mh_pack_rows", _ptr(x0d), _ptr(row_map), _ptr(n_rows), _ptr(xt[0]), Tt, Dt)
            tp = dict(rows_dev=n_rows)                  # every text GEMM problem / LayerNorm job
            tpd = dict(rows_dev=n_rows, drop_rows=row_map)
        else:
            cu = row_map = inv_map = pmask = pool_rows = n_rows = None
            xt = [x0d]
            tp, tpd = {}, {}

        def text_attn(a_, sid, **bw):
            base = dict(qkv=a_["qkv"], out=a_["ctx"], lse=a_["lse"], B=B, S=S, H=Ht, drop=site(p_a, sid), **bw)
            if pack:
                return dict(base, key_mask=pmask, cu=cu, row_map=row_map)
            return dict(base, key_mask=mask)

        # image embeddings
        Kpp = v.patch_dim_padded                      # contraction of the patch projection (588 -> 640 for CLIP's 14x14 patches)
        patches = alloc("i.patches", (B * Np, Kpp))
        proj = alloc("i.proj", (B * Np, Di))
        xi = [alloc("i.x0", (Ti, Di))]
        wp = self.w(IMG + "embeddings.patch_embeddings.projection.weight")
        if Kpp == Kp and v.patch % 8 == 0:
            f.c("mh_patchify", _ptr(image), _ptr(patches), B, v.channels, v.image_size, v.image_size, v.patch)
        else:
            f.c("mh_patchify_ld", _ptr(image), _ptr(patches), B, v.channels, v.image_size, v.image_size, v.patch, Kpp)
        if Kpp != Kp:       # zero-padded copy of the [Di][Kp] weight shadow (two 16-bit elements per word)
            wpad = alloc("i.wp_pad", (Di, Kpp))
            f.c("mh_copy2d_u32", _ptr(wp), Kp // 2, _ptr(wpad), Kpp // 2, Di, Kp // 2, Kpp // 2)
            wp = wpad
        pbias = self.p(IMG + "embeddings.patch_embeddings.projection.bias") if v.patch_bias else None
        self._gemm(pl, f, [self._fwd_prob(patches, wp, proj, B * Np, Di, Kpp, bias=pbias)], False, False)
        if v.pre_ln:        # CLIP: embeddings -> pre_layrnorm -> blocks
            x0pre = alloc("i.x0pre", (Ti, Di))
            m_pre, r_pre = alloc("i.mpre", (Ti,), F32), alloc("i.rpre", (Ti,), F32)
            f.c("mh_vit_assemble_fwd", _ptr(proj), _ptr(self.p(IMG + "embeddings.cls_token")),
                _ptr(self.p(IMG + "embeddings.position_embeddings")), _ptr(x0pre), B, Np, Di)
            self._ln_fwd(pl, f, [self._ln_fwd_job(x0pre, IMG + "pre_layernorm.weight", IMG + "pre_layernorm.bias", xi[0],
                                                  m_pre, r_pre, Ti, Di, v.ln_eps)])
        else:
            f.c("mh_vit_assemble_fwd", _ptr(proj), _ptr(self.p(IMG + "embeddings.cls_token")),
                _ptr(self.p(IMG + "embeddings.position_embeddings")), _ptr(xi[0]), B, Np, Di)
        quick = dict(quick=True) if v.act == "quick_gelu" else {}

        xt_last32 = alloc("t.xlast32", (Tt, Dt), F32)   # unrounded tower outputs for the fp32 head
        xf32 = alloc("i.xf32", (Ti, Di), F32)
        tl: List[dict] = []   # per-layer saved activations, text
        il: List[dict] = []
        for l in range(Lt):
            tl.append(dict(qkv=alloc(f"t{l}.qkv", (Tt, 3 * Dt)), ctx=alloc(f"t{l}.ctx", (Tt, Dt)),
                           lse=alloc(f"t{l}.lse", (B, Ht, S), F32), a=alloc(f"t{l}.a", (Tt, Dt)),
                           y=alloc(f"t{l}.y", (Tt, Dt)), m1=alloc(f"t{l}.m1", (Tt,), F32), r1=alloc(f"t{l}.r1", (Tt,), F32),
                           h=alloc(f"t{l}.h", (Tt, It)), g=alloc(f"t{l}.g", (Tt, It)), f=alloc(f"t{l}.f", (Tt, Dt)),
                           m2=alloc(f"t{l}.m2", (Tt,), F32), r2=alloc(f"t{l}.r2", (Tt,), F32)))
            xt.append(alloc(f"t.x{l + 1}", (Tt, Dt)))
        for l in range(Li):
            il.append(dict(u=alloc(f"i{l}.u", (Ti, Di)), m1=alloc(f"i{l}.m1", (Ti,), F32), r1=alloc(f"i{l}.r1", (Ti,), F32),
                           qkv=alloc(f"i{l}.qkv", (Ti, 3 * Di)), ctx=alloc(f"i{l}.ctx", (Ti, Di)),
                           lse=alloc(f"i{l}.lse", (B, Hi, Nt), F32), xp=alloc(f"i{l}.xp", (Ti, Di)),
                           w=alloc(f"i{l}.w", (Ti, Di)), m2=alloc(f"i{l}.m2", (Ti,), F32), r2=alloc(f"i{l}.r2", (Ti,), F32),
                           h=alloc(f"i{l}.h", (Ti, Ii)), g=alloc(f"i{l}.g", (Ti, Ii))))
            xi.append(alloc(f"i.x{l + 1}", (Ti, Di)))
        xf = alloc("i.xf", (Ti, Di))
        mf, rf = alloc("i.mf", (Ti,), F32), alloc("i.rf", (Ti,), F32)

        def img_ln1(l):
            """The ViT block's pre-attention LayerNorm of layer l; past the last layer, the final ViT LayerNorm.
            It is issued together with the text tower's LayerNorm that closes layer l-1 (one grouped launch)."""
            if l < Li:
                LI = f"{IMG}encoder.layer.{l}."
                return self._ln_fwd_job(xi[l], LI + "layernorm_before.weight", LI + "layernorm_before.bias", il[l]["u"],
                                        il[l]["m1"], il[l]["r1"], Ti, Di, v.ln_eps)
            if l == Li:
                return self._ln_fwd_job(xi[Li], IMG + "layernorm.weight", IMG + "layernorm.bias", xf, mf, rf, Ti, Di,
                                        v.ln_eps, y32=xf32)
            return None

        self._ln_fwd(pl, f, [img_ln1(0)])
        for l in range(max(Lt, Li)):
            has_t, has_i = l < Lt, l < Li
            LT, LI = f"{TXT}encoder.layer.{l}.", f"{IMG}encoder.layer.{l}."
            a = tl[l] if has_t else None
            b_ = il[l] if has_i else None
            # QKV
            pr = []
            if has_t:
                pr.append(self._fwd_prob(xt[l], self.w(LT + "attention.self.query.weight", 3), a["qkv"], Tt, 3 * Dt, Dt,
                                         bias=self.p(LT + "attention.self.query.bias", 3), **tp))
            if has_i:
                pr.append(self._fwd_prob(b_["u"], self.w(LI + "attention.attention.query.weight", 3), b_["qkv"], Ti,
                                         3 * Di, Di, bias=self.p(LI + "attention.attention.query.bias", 3)))
            self._gemm(pl, f, pr, False, False)
            self._attn(pl, f, [
                dict(qkv=b_["qkv"], out=b_["ctx"], lse=b_["lse"], B=B, S=Nt, H=Hi) if has_i else None,
                text_attn(a, 16 * (l + 1) + 1) if has_t else None], backward=False)
            # attention output projection + residual
            pr = []
            if has_t:
                pr.append(self._fwd_prob(a["ctx"], self.w(LT + "attention.output.dense.weight"), a["a"], Tt, Dt, Dt,
                                         bias=self.p(LT + "attention.output.dense.bias"), residual=xt[l],
                                         drop=site(p_h, 16 * (l + 1) + 2), **tpd))
            if has_i:
                pr.append(self._fwd_prob(b_["ctx"], self.w(LI + "attention.output.dense.weight"), b_["xp"], Ti, Di, Di,
                                         bias=self.p(LI + "attention.output.dense.bias"), residual=xi[l]))
            self._gemm(pl, f, pr, False, False)
            self._ln_fwd(pl, f, [
                self._ln_fwd_job(a["a"], LT + "attention.output.LayerNorm.weight", LT + "attention.output.LayerNorm.bias",
                                 a["y"], a["m1"], a["r1"], Tt, Dt, t.ln_eps, **tp) if has_t else None,
                self._ln_fwd_job(b_["xp"], LI + "layernorm_after.weight", LI + "layernorm_after.bias", b_["w"], b_["m2"],
                                 b_["r2"], Ti, Di, v.ln_eps) if has_i else None])
            # FFN up + GELU; "h" keeps gelu'(pre-activation) for the backward (computed here from the activation's own exponential:
            # MH_GEMM_DERIV_AUX; MEMEHIP_GEMM_DERIV_AUX=0 keeps the pre-activation and evaluates gelu' in the backward epilogue)
            pr = []
            if has_t:
                pr.append(self._fwd_prob(a["y"], self.w(LT + "intermediate.dense.weight"), a["g"], Tt, It, Dt,
                                         bias=self.p(LT + "intermediate.dense.bias"), aux=a["h"], gelu=True, deriv_aux=DERIV_AUX, **tp))
            if has_i:
                pr.append(self._fwd_prob(b_["w"], self.w(LI + "intermediate.dense.weight"), b_["g"], Ti, Ii, Di,
                                         bias=self.p(LI + "intermediate.dense.bias"), aux=b_["h"], gelu=True, deriv_aux=DERIV_AUX, **quick))
            self._gemm(pl, f, pr, False, False)
            # FFN down + residual
            pr = []
            if has_t:
                pr.append(self._fwd_prob(a["g"], self.w(LT + "output.dense.weight"), a["f"], Tt, Dt, It,
                                         bias=self.p(LT + "output.dense.bias"), residual=a["y"],
                                         drop=site(p_h, 16 * (l + 1) + 3), **tpd))
            if has_i:
                pr.append(self._fwd_prob(b_["g"], self.w(LI + "output.dense.weight"), xi[l + 1], Ti, Di, Ii,
                                         bias=self.p(LI + "output.dense.bias"), residual=b_["xp"]))
            self._gemm(pl, f, pr, False, False)
            # the text LayerNorm closing this layer + the image LayerNorm opening the next (or the final ViT one)
            self._ln_fwd(pl, f, [
                self._ln_fwd_job(a["f"], LT + "output.LayerNorm.weight", LT + "output.LayerNorm.bias", xt[l + 1], a["m2"],
                                 a["r2"], Tt, Dt, t.ln_eps, y32=xt_last32 if l == Lt - 1 else None, **tp) if has_t else None,
                img_ln1(l + 1) if has_i else None])

        # head
        hp, hg = MhHeadParams(), MhHeadGrads()
        for fld, nm in (("Wt", "bert_fc.weight"), ("bt", "bert_fc.bias"), ("Wi", "image_fc.weight"),
                        ("bi", "image_fc.bias"), ("Wf", "fusion_fc.weight"), ("bf_", "fusion_fc.bias"),
                        ("Wo", "output_fc.weight"), ("bo", "output_fc.bias")):
            setattr(hp, fld, _ptr(self.p(nm)))
            setattr(hg, fld, _ptr(self.g(nm)))
        pl.keep += [hp, hg]
        pooled = alloc("h.pooled", (B, Dt + Di), F32)
        feat, fused = alloc("h.feat", (B, 2 * P_), F32), alloc("h.fused", (B, P_), F32)
        logits = alloc("logits", (B, Cn), F32)
        pl.features = features
        if features == "sequence":
            assert not pack
            d_seq_t, d_seq_i = alloc("h.d_seq_t", (Tt, Dt), F32, zero=True), alloc("h.d_seq_i", (Ti, Di), F32, zero=True)
        elif features:
            d_pooled = alloc("h.d_pooled", (B, Dt + Di), F32, zero=True)
            f.c("mh_pool_fwd", _ptr(xt_last32), _ptr(xf32), pool_index, _ptr(pooled), B, S, Nt, Dt, Di, _ptr(pool_rows))
        else:
            f.c("mh_head_fwd", C.byref(hp), _ptr(xt_last32), _ptr(xf32), pool_index, _ptr(pooled), _ptr(feat), _ptr(fused),
                _ptr(logits), B, S, Nt, Dt, Di, P_, Cn, *site_args(p_head, 7), _ptr(pool_rows))

        # loss
        loss = alloc("loss", (1,), F32, zero=True)
        dlogits = alloc("dlogits", (B, Cn), F32, zero=True)
        ncorrect = alloc("ncorrect", (1,), torch.int32, zero=True)
        grad_scale = 1.0
        pl.loss.c("mh_ce_fwd_bwd", _ptr(logits), _ptr(labels), _ptr(loss), _ptr(dlogits), _ptr(ncorrect), B, Cn,
                  float(grad_scale), _ptr(self.loss_scale))

        # ------------------------------------------------------------------ backward -----------------
        def seg(name):
            s = Segment(name, self.lib)
            pl.bwd.append(s)
            return s

        s = seg("bwd_head")
        # the weight-gradient GEMMs of layer l may run on a side stream while the chain of layers l-1 and
        # l-2 proceeds: everything they read is double-buffered by layer parity (dX: rotation of three),
        # and layer l-2 waits for them before it overwrites those buffers
        dXt = [alloc(f"t.dX{i}", (Tt, Dt)) for i in range(3)]
        dXi = [alloc(f"i.dX{i}", (Ti, Di)) for i in range(3)]
        dXf = alloc("i.dXf", (Ti, Di))
        dfeat, dfused = alloc("h.dfeat", (B, 2 * P_), F32), alloc("h.dfused", (B, P_), F32)
        s.py(dXt[0].zero_)
        s.py(dXf.zero_)
        if features:
            # the built-in head is not part of this graph: its gradient slots must not keep stale values
            h0 = self.layout.spec["bert_fc.weight"].offset
            h1 = self.layout.spec["output_fc.bias"].offset + self.layout.spec["output_fc.bias"].numel
            head_grads = self.G[h0:h1]
            s.py(head_grads.zero_)
            if features == "sequence":      # the caller's gradient w.r.t. every hidden state (already times the stream scale)
                s.c("mh_cast_f32_bf16", _ptr(d_seq_t), _ptr(dXt[0]), Tt * Dt)
                s.c("mh_cast_f32_bf16", _ptr(d_seq_i), _ptr(dXf), Ti * Di)
            else:
                s.c("mh_pool_bwd", _ptr(d_pooled), _ptr(dXt[0]), _ptr(dXf), pool_index, B, S, Nt, Dt, Di, float(self.gscale),
                    _ptr(pool_rows))
        else:
            s.c("mh_head_bwd", C.byref(hp), C.byref(hg), _ptr(dlogits), _ptr(pooled), _ptr(feat), _ptr(fused), _ptr(dfeat),
                _ptr(dfused), _ptr(dXt[0]), _ptr(dXf), pool_index, B, S, Nt, Dt, Di, P_, Cn, float(self.gscale),
                *site_args(p_head, 7), _ptr(pool_rows))
        # backward temporaries, one set per layer parity
        T_ = [dict(dh=alloc(f"t.dh{i}", (Tt, It)), da=alloc(f"t.da{i}", (Tt, Dt)),
                   dqkv=alloc(f"t.dqkv{i}", (Tt, 3 * Dt)),
                   dam=alloc(f"t.dam{i}", (Tt, Dt)) if p_h > 0 else None)     # da times the dropout mask of the Linear
              for i in range(2)]                                              # output that fed the LayerNorm
        # df (gradient at the FFN output) is written one segment EARLY, by the LayerNorm backward grouped into the
        # end of the layer above: rotation of three, so the side-stream GEMMs of layer l+1 never see it overwritten
        DF_ = [dict(df=alloc(f"t.df{i}", (Tt, Dt)), dfm=alloc(f"t.dfm{i}", (Tt, Dt)) if p_h > 0 else None)
               for i in range(3)]
        I_ = [dict(dh=alloc(f"i.dh{i}", (Ti, Ii)), dxp=alloc(f"i.dxp{i}", (Ti, Di)),
                   dqkv=alloc(f"i.dqkv{i}", (Ti, 3 * Di))) for i in range(2)]
        t_dy, t_dctx = alloc("t.dy", (Tt, Dt)), alloc("t.dctx", (Tt, Dt))
        t_delta = alloc("t.delta", (B, Ht, S), F32)
        i_dw, i_dctx, i_du = alloc("i.dw", (Ti, Di)), alloc("i.dctx", (Ti, Di)), alloc("i.du", (Ti, Di))
        i_delta = alloc("i.delta", (B, Hi, Nt), F32)

        def text_ln2_bwd(l, ct_):
            """Backward of the LayerNorm that closes text layer l (incoming gradient dXt[ct_]).  Issued at the END of
            the segment above, grouped with the image tower's LayerNorm there."""
            if not (0 <= l < Lt):
                return None
            LT_ = f"{TXT}encoder.layer.{l}."
            return self._ln_bwd_job(pl, dXt[ct_], tl[l]["f"], LT_ + "output.LayerNorm.weight", LT_ + "output.LayerNorm.bias",
                                    tl[l]["m2"], tl[l]["r2"], DF_[l % 3]["df"], Tt, Dt,
                                    dx_drop=DF_[l % 3]["dfm"] if p_h > 0 else None, drop=site(p_h, 16 * (l + 1) + 3), **tpd)

        nl = max(Lt, Li)
        self._ln_bwd(pl, s, [
            self._ln_bwd_job(pl, dXf, xi[Li], IMG + "layernorm.weight", IMG + "layernorm.bias", mf, rf, dXi[0], Ti, Di),
            text_ln2_bwd(nl - 1, 0) if Lt == nl else None])

        t_dpre = alloc("t.dpre", (Tt, Dt))
        ct, ci = 0, 0   # current ping-pong index of the incoming gradient
        for l in range(nl - 1, -1, -1):
            has_t, has_i = l < Lt, l < Li
            LT, LI = f"{TXT}encoder.layer.{l}.", f"{IMG}encoder.layer.{l}."
            s = seg(f"bwd_layer_{l}")
            s.wait(f"bwd_layer_{l + 2}")
            a = tl[l] if has_t else None
            b_ = il[l] if has_i else None
            t_dh, t_da, t_dqkv = (T_[l & 1][k] for k in ("dh", "da", "dqkv"))
            t_df = DF_[l % 3]["df"]
            t_dfm = DF_[l % 3]["dfm"] if p_h > 0 else t_df     # gradient w.r.t. the (dropped) FFN output dense
            t_dam = T_[l & 1]["dam"] if p_h > 0 else t_da      # gradient w.r.t. the (dropped) attention output dense
            i_dh, i_dxp, i_dqkv = (I_[l & 1][k] for k in ("dh", "dxp", "dqkv"))
            # d gelu_in = (d_out @ W2) * gelu'(h)
            pr = []
            if has_t:
                pr.append(self._dgrad_prob(t_dfm, self.w(LT + "output.dense.weight"), t_dh, Tt, Dt, It, mul=a["h"], deriv_aux=DERIV_AUX, **tp))
            if has_i:
                pr.append(self._dgrad_prob(dXi[ci], self.w(LI + "output.dense.weight"), i_dh, Ti, Di, Ii, mul=b_["h"], deriv_aux=DERIV_AUX,
                                           **quick))
            self._gemm(pl, s, pr, False, True)
            # through W1 (text adds the residual branch df)
            pr = []
            if has_t:
                pr.append(self._dgrad_prob(t_dh, self.w(LT + "intermediate.dense.weight"), t_dy, Tt, It, Dt, residual=t_df, **tp))
            if has_i:
                pr.append(self._dgrad_prob(i_dh, self.w(LI + "intermediate.dense.weight"), i_dw, Ti, Ii, Di))
            self._gemm(pl, s, pr, False, True)
            self._ln_bwd(pl, s, [
                self._ln_bwd_job(pl, t_dy, a["a"], LT + "attention.output.LayerNorm.weight",
                                 LT + "attention.output.LayerNorm.bias", a["m1"], a["r1"], t_da, Tt, Dt,
                                 dx_drop=t_dam if p_h > 0 else None, drop=site(p_h, 16 * (l + 1) + 2), **tpd) if has_t else None,
                self._ln_bwd_job(pl, i_dw, b_["xp"], LI + "layernorm_after.weight", LI + "layernorm_after.bias", b_["m2"],
                                 b_["r2"], i_dxp, Ti, Di, dx_add=dXi[ci]) if has_i else None])
            # through the attention output projection
            pr = []
            if has_t:
                pr.append(self._dgrad_prob(t_dam, self.w(LT + "attention.output.dense.weight"), t_dctx, Tt, Dt, Dt, **tp))
            if has_i:
                pr.append(self._dgrad_prob(i_dxp, self.w(LI + "attention.output.dense.weight"), i_dctx, Ti, Di, Di))
            self._gemm(pl, s, pr, False, True)
            self._attn(pl, s, [
                dict(qkv=b_["qkv"], out=b_["ctx"], lse=b_["lse"], dout=i_dctx, delta=i_delta, dqkv=i_dqkv, B=B, S=Nt,
                     H=Hi) if has_i else None,
                text_attn(a, 16 * (l + 1) + 1, dout=t_dctx, delta=t_delta, dqkv=t_dqkv) if has_t else None], backward=True)
            # through the QKV projection
            pr = []
            if has_t:
                pr.append(self._dgrad_prob(t_dqkv, self.w(LT + "attention.self.query.weight", 3), dXt[(ct + 1) % 3], Tt, 3 * Dt, Dt,
                                           residual=t_da, **tp))
            if has_i:
                pr.append(self._dgrad_prob(i_dqkv, self.w(LI + "attention.attention.query.weight", 3), i_du, Ti, 3 * Di, Di))
            self._gemm(pl, s, pr, False, True)
            # the image LayerNorm that opens this layer + the text LayerNorm that closes the layer below (at layer 0:
            # the text embedding LayerNorm), one grouped launch
            ct_next = (ct + 1) % 3 if has_t else ct
            if l >= 1:
                t_job = text_ln2_bwd(l - 1, ct_next)
            else:
                d_emb = dXt[ct_next]
                if pack:        # back to dense [B*S] rows (zeros at the dropped positions) for the embedding tables
                    d_emb = alloc("t.dX_dense", (Tt, Dt))
                    s.c("mh_unpack_rows", _ptr(dXt[ct_next]), _ptr(inv_map), _ptr(d_emb), Tt, Dt)
                if p_h > 0:     # gradient w.r.t. the dropped embedding output -> w.r.t. the LayerNorm output
                    s.c("mh_dropout_apply", _ptr(d_emb), Tt * Dt, rng_t.data_ptr(), float(p_h), 1)
                t_job = self._ln_bwd_job(pl, d_emb, pre0, TXT + "embeddings.LayerNorm.weight",
                                         TXT + "embeddings.LayerNorm.bias", m0, r0, t_dpre, Tt, Dt)
            self._ln_bwd(pl, s, [
                t_job,
                self._ln_bwd_job(pl, i_du, xi[l], LI + "layernorm_before.weight", LI + "layernorm_before.bias", b_["m1"],
                                 b_["r1"], dXi[(ci + 1) % 3], Ti, Di, dx_add=i_dxp) if has_i else None])
            # weight gradients (+ bias gradients as row sums): one grouped launch per tower.  The 4+4 problems in ONE launch (864 tiles)
            # measured 254 us against 215 us for the two 432-tile launches back to back (round 1), and in the step 11.2 ms against 10.2 ms
            # (round 3, with or without a tile order giving every XCD its share of every problem, tools/lab/RESULTS.md (r3k)): the side stream must not hold every workgroup
            # slot -- two 432-tile launches leave 80 of the 512 slots to the main chain.
            wg_t = [
                    self._wgrad_prob(t_dfm, a["g"], self.g(LT + "output.dense.weight"), self.g(LT + "output.dense.bias"), Tt, Dt, It, **tp),
                    self._wgrad_prob(t_dh, a["y"], self.g(LT + "intermediate.dense.weight"), self.g(LT + "intermediate.dense.bias"), Tt, It, Dt, **tp),
                    self._wgrad_prob(t_dam, a["ctx"], self.g(LT + "attention.output.dense.weight"), self.g(LT + "attention.output.dense.bias"), Tt, Dt, Dt, **tp),
                    self._wgrad_prob(t_dqkv, xt[l], self.g(LT + "attention.self.query.weight", 3), self.g(LT + "attention.self.query.bias", 3), Tt, 3 * Dt, Dt, **tp)] if has_t else []
            wg_i = [
                    self._wgrad_prob(dXi[ci], b_["g"], self.g(LI + "output.dense.weight"), self.g(LI + "output.dense.bias"), Ti, Di, Ii),
                    self._wgrad_prob(i_dh, b_["w"], self.g(LI + "intermediate.dense.weight"), self.g(LI + "intermediate.dense.bias"), Ti, Ii, Di),
                    self._wgrad_prob(i_dxp, b_["ctx"], self.g(LI + "attention.output.dense.weight"), self.g(LI + "attention.output.dense.bias"), Ti, Di, Di),
                    self._wgrad_prob(i_dqkv, b_["u"], self.g(LI + "attention.attention.query.weight", 3), self.g(LI + "attention.attention.query.bias", 3), Ti, 3 * Di, Di)] if has_i else []
            self._gemm(pl, s, wg_t, True, True, lane=1)
            self._gemm(pl, s, wg_i, True, True, lane=1)
            if has_t:
                ct = (ct + 1) % 3
            if has_i:
                ci = (ci + 1) % 3
            rng = [r for r in self.layout.layer_ranges if r[0] == l][0]
            pl.bucket_after[s.name] = (rng[1], rng[2])

        # embeddings
        s = seg("bwd_embed")
        gword = self.g(TXT + "embeddings.word_embeddings.weight")
        gtype0 = self.g(TXT + "embeddings.token_type_embeddings.weight")[:Dt] if t.type_vocab > 0 else None
        gpos = self.g(TXT + "embeddings.position_embeddings.weight")

        def table_grads(sg, ids_, dpre_, prev_, nb):
            # dense-Adam semantics: the table gradient is dense; only rows touched last step need re-zeroing
            sg.c("mh_zero_rows_f32", _ptr(prev_), _ptr(gword), nb * S, Dt, t.vocab_size)
            sg.c("mh_bert_embed_bwd", _ptr(ids_), _ptr(dpre_), _ptr(gword), _ptr(gpos), _ptr(gtype0), nb, S, Dt, t.vocab_size,
                 int(t.pad_token_id), 1.0 / self.gscale, self.word_row_live.data_ptr(), self.word_first.data_ptr(),
                 self.word_count.data_ptr())
            sg.py(lambda: prev_.copy_(ids_))

        i_dproj = alloc("i.dproj", (B * Np, Di))
        d_x0 = dXi[ci]
        if v.pre_ln:        # through CLIP's pre_layrnorm
            d_x0 = alloc("i.dx0pre", (Ti, Di))
            self._ln_bwd(pl, s, [self._ln_bwd_job(pl, dXi[ci], pl.buf["i.x0pre"], IMG + "pre_layernorm.weight",
                                                  IMG + "pre_layernorm.bias", pl.buf["i.mpre"], pl.buf["i.rpre"], d_x0, Ti, Di)])
        s.c("mh_vit_assemble_bwd", _ptr(d_x0), _ptr(i_dproj), _ptr(self.g(IMG + "embeddings.cls_token")),
            _ptr(self.g(IMG + "embeddings.position_embeddings")), B, Np, Di, 1.0 / self.gscale)
        # (the patch-projection weight gradient is a 36-tile GEMM with a 6272-deep contraction, 115 us on 14 % of the CUs;
        #  moving it to the side stream beside the table gradients measured 0.12 ms SLOWER per step: it delays the join
        #  in front of the optimizer tail)
        gw = self.g(IMG + "embeddings.patch_embeddings.projection.weight")
        gb = self.g(IMG + "embeddings.patch_embeddings.projection.bias") if v.patch_bias else None
        gw_out = gw
        if Kpp != Kp:       # the GEMM writes a [Di][Kpp] gradient; its first Kp columns are copied into the flat buffer
            gw = alloc("i.dwp_pad", (Di * Kpp,), F32)
        Tp = B * Np
        nsplit = min(_lib.MH_GEMM_MAX_GROUP, max(1, Tp // 512))
        if nsplit > 1:
            # split-K: the contraction over B*Np patches is cut into `nsplit` grouped problems with their own f32
            # outputs (288 tiles instead of 36), summed in a fixed order by the partial-sum kernel
            chunk = -(-Tp // nsplit)
            chunk = -(-chunk // 8) * 8
            part_w = alloc("i.dwp_part", (2, nsplit, Di * Kpp), F32)      # [which][split][Di*Kpp]; which = 1 unused
            part_b = alloc("i.dbp_part", (2, nsplit, Di), F32)
            probs = []
            for i in range(nsplit):
                k0 = i * chunk
                kk = min(chunk, Tp - k0)
                if kk <= 0:
                    part_w[0, i].zero_(), part_b[0, i].zero_()
                    continue
                probs.append(self._wgrad_prob(i_dproj[k0:], patches[k0:], part_w[0, i], part_b[0, i], kk, Di, Kpp))
            self._gemm(pl, s, probs, True, True)
            for part, out, D_ in ((part_w, gw, Di * Kpp), (part_b, gb, Di)):
                if out is None:
                    continue
                arr = (MhColsumJob * 1)()
                arr[0].part, arr[0].out0, arr[0].out1 = _ptr(part), _ptr(out), None
                pl.keep.append(arr)
                s.c("mh_colsum_partials_f32", arr, 1, nsplit, D_, 1.0)
        else:
            self._gemm(pl, s, [self._wgrad_prob(i_dproj, patches, gw, gb, Tp, Di, Kpp)], True, True)
        if Kpp != Kp:
            s.c("mh_copy2d_u32", _ptr(gw), Kpp, _ptr(gw_out), Kp, Di, Kp, Kp)
        if gather_world <= 0:
            table_grads(s, ids, t_dpre, prev_ids, B)
        # finish every LayerNorm's dgamma / dbeta from the partials
        for D, jobs in pl._ln_jobs.items():
            for i0 in range(0, len(jobs), _lib.MH_COLSUM_MAX_JOBS):
                chunk = jobs[i0:i0 + _lib.MH_COLSUM_MAX_JOBS]
                arr = (MhColsumJob * len(chunk))()
                for j, (part, o0, o1) in enumerate(chunk):
                    arr[j].part, arr[j].out0, arr[j].out1 = _ptr(part), _ptr(o0), _ptr(o1)
                pl.keep.append(arr)
                s.c("mh_colsum_partials_f32", arr, len(chunk), LN_PARTS, D, 1.0 / self.gscale)
        tail0 = self.layout.layer_ranges[-1][2]
        if gather_world <= 0:
            pl.bucket_after[s.name] = (tail0, self.layout.n_total)
        else:
            # data parallel: word / position / token-type tables sit at the end of the flat buffer; their gradients
            # come from the gathered rows of every rank (already the global sum), so the all-reduce stops before them
            first_tbl = TXT + ("embeddings.token_type_embeddings.weight" if t.type_vocab > 0
                               else "embeddings.position_embeddings.weight")
            tbl0 = self.layout.spec[first_tbl].offset
            assert self.layout.spec[TXT + "embeddings.word_embeddings.weight"].offset + t.vocab_size * Dt + 3 >= self.layout.n_total - 3
            pl.bucket_after[s.name] = (tail0, tbl0)
            W = gather_world
            ids_all = alloc("ids_all", (W * B, S), I64, zero=True)
            prev_all = alloc("prev_ids_all", (W * B, S), I64, zero=True)
            dpre_all = alloc("t.dpre_all", (W * Tt, Dt))
            pl.gather = [(ids, ids_all), (t_dpre, dpre_all)]      # (local, gathered) pairs, exchanged between segments
            s2 = seg("bwd_embed_tables")
            table_grads(s2, ids_all, dpre_all, prev_all, W * B)
        pl.n_launches = sum(1 for sg in [pl.fwd, pl.loss] + pl.bwd for c in sg.calls if c[0] is not None)
        return pl
