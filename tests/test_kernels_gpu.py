"""Per-kernel parity through the C ABI (libmemehip.so) against plain PyTorch fp32 references.
Runs on the MI355X box only (-m gpu)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodal_propaganda_meme_classification_amd import ops as _ops
    return _ops


def dev():
    return torch.device("cuda:0")


def rnd(*shape, scale=1.0, dtype=BF16, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + int(np.prod(shape)) % 1000)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


def ints(*shape, lo=-2, hi=3, seed=0, dtype=BF16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).to(dtype).to(dev())


BOTH = pytest.mark.parametrize("T16", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])


def close(a, b, rtol, atol, what=""):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4g} (ref max {float(b.abs().max()):.4g})"


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------

@BOTH
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (394, 256, 192), (64, 128, 128), (1000, 384, 768)])
def test_gemm_exact_integer_all_layouts(ops, M, N, K, T16):
    """Small-integer operands: every product and sum is exact, so the three layouts must be
    bit-exact against fp32 matmul (catches any fragment / transpose / swizzle mix-up)."""
    A = ints(M, K, seed=1, dtype=T16)            # asymmetric random integers
    Bw = ints(N, K, seed=2, dtype=T16)
    ref = A.float() @ Bw.float().t()
    out = torch.empty((M, N), dtype=F32, device=dev())
    ops.gemm_grouped([ops.Gemm(A, Bw, out, M, N, K, K, K, N)], False, False)
    assert torch.equal(out, ref), "forward layout (0,0)"
    # dgrad layout: C[M,N] = A[M,K] @ B[K,N] with B stored [K][N]
    Bk = Bw.t().contiguous()
    out.zero_()
    ops.gemm_grouped([ops.Gemm(A, Bk, out, M, N, K, K, N, N)], False, True)
    assert torch.equal(out, ref), "dgrad layout (0,1)"
    # wgrad layout: both stored K-major; M must be a multiple of 128 there
    M2 = ((M + 127) // 128) * 128
    A2 = ints(K, M2, seed=3, dtype=T16)
    ref2 = A2.float().t() @ Bk.float()
    out2 = torch.empty((M2, N), dtype=F32, device=dev())
    rows = torch.empty(M2, dtype=F32, device=dev())
    ops.gemm_grouped([ops.Gemm(A2, Bk, out2, M2, N, K, M2, N, N, rowsum=rows, alpha=0.25)], True, True)
    assert torch.equal(out2, 0.25 * ref2), "wgrad layout (1,1) with alpha"
    assert torch.equal(rows, 0.25 * A2.float().sum(0)), "wgrad rowsum (bias gradient)"


def test_gemm_wgrad_ragged_contraction(ops):
    """wgrad contracts over tokens: T = 2*197 is not a multiple of the 64-deep K tile."""
    T, N, K = 394, 256, 128
    dy, x = ints(T, N, seed=4), ints(T, K, seed=5)
    dw = torch.empty((N, K), dtype=F32, device=dev())
    db = torch.empty(N, dtype=F32, device=dev())
    ops.linear_wgrad(dy, x, dw, db)
    assert torch.equal(dw, dy.float().t() @ x.float())
    assert torch.equal(db, dy.float().sum(0))
    ops.linear_wgrad(dy, x, dw, None, accum=True)
    assert torch.equal(dw, 2 * (dy.float().t() @ x.float()))


def test_gemm_epilogues_and_grouping(ops):
    T1, T2, N, K = 256, 394, 384, 256
    x1, x2 = rnd(T1, K, seed=1), rnd(T2, K, seed=2)
    w1, w2 = rnd(N, K, scale=0.05, seed=3), rnd(N, K, scale=0.05, seed=4)
    b1, b2 = rnd(N, dtype=F32, seed=5), rnd(N, dtype=F32, seed=6)
    r1, r2 = rnd(T1, N, seed=7), rnd(T2, N, seed=8)
    o1, o2 = torch.empty((T1, N), dtype=BF16, device=dev()), torch.empty((T2, N), dtype=BF16, device=dev())
    a1, a2 = torch.empty_like(o1), torch.empty_like(o2)
    # bias + gelu with pre-activation aux, two problems in one launch
    ops.gemm_grouped([ops.Gemm(x1, w1, o1, T1, N, K, K, K, N, bias=b1, aux=a1, gelu=True),
                      ops.Gemm(x2, w2, o2, T2, N, K, K, K, N, bias=b2, aux=a2, gelu=True)], False, False)
    for x, w, b, o, a in ((x1, w1, b1, o1, a1), (x2, w2, b2, o2, a2)):
        pre = x.float() @ w.float().t() + b
        close(a, pre, 8e-3, 2e-3, "aux pre-activation")
        close(o, torch.nn.functional.gelu(pre), 8e-3, 2e-3, "bias+gelu")
    # bias + residual
    ops.gemm_grouped([ops.Gemm(x1, w1, o1, T1, N, K, K, K, N, bias=b1, residual=r1),
                      ops.Gemm(x2, w2, o2, T2, N, K, K, K, N, bias=b2, residual=r2)], False, False)
    close(o1, x1.float() @ w1.float().t() + b1 + r1.float(), 8e-3, 2e-3, "bias+residual p0")
    close(o2, x2.float() @ w2.float().t() + b2 + r2.float(), 8e-3, 2e-3, "bias+residual p1")
    # dgrad * gelu'(pre)
    dy = rnd(T2, N, seed=9)
    pre = rnd(T2, K, seed=10)
    dx = ops.linear_dgrad(dy, w2, mul=pre)
    pf = pre.float().requires_grad_(True)
    torch.nn.functional.gelu(pf).backward(dy.float() @ w2.float())
    close(dx, pf.grad, 1e-2, 3e-3, "dgrad*gelu'")
    # MH_GEMM_DERIV_AUX: the forward epilogue stores gelu'(pre) (erf and quick form), the dgrad multiplies by it as it is
    for quick in (False, True):
        aD = torch.empty_like(o2)
        oD = torch.empty_like(o2)
        ops.gemm_grouped([ops.Gemm(x2, w2, oD, T2, N, K, K, K, N, bias=b2, aux=aD, gelu=True, quick=quick, deriv_aux=True)], False, False)
        pre2 = (x2.float() @ w2.float().t() + b2).requires_grad_(True)
        act = pre2 * torch.sigmoid(1.702 * pre2) if quick else torch.nn.functional.gelu(pre2)
        act.sum().backward()
        close(oD, act.detach(), 8e-3, 2e-3, f"activation (quick={quick})")
        close(aD, pre2.grad, 6e-3, 4e-3, f"stored derivative (quick={quick})")
        dyD = rnd(T2, K, seed=11)       # a gradient of the [T2, K]-shaped FFN output going back through w [K <- N]
        wD = rnd(K, N, scale=0.05, seed=12)
        dxD = ops.linear_dgrad(dyD, wD, mul=aD, quick=quick, deriv_aux=True)
        close(dxD, (dyD.float() @ wD.float()) * aD.float(), 1e-2, 3e-3, f"dgrad * stored derivative (quick={quick})")


def test_gemm_rejects_bad_shapes(ops):
    from multimodal_propaganda_meme_classification_amd._lib import MemehipError
    A, Bw = rnd(64, 64), rnd(100, 64)
    out = torch.empty((64, 100), dtype=BF16, device=dev())
    with pytest.raises((MemehipError, ValueError)):
        ops.gemm_grouped([ops.Gemm(A, Bw, out, 64, 100, 64, 64, 64, 100)], False, False)  # N % 128
    with pytest.raises(MemehipError):
        ops.linear_fwd(A.cpu(), Bw.cpu())  # no CPU fallback


# ------------------------------------------------------------------------------------------------
# LayerNorm
# ------------------------------------------------------------------------------------------------

@BOTH
@pytest.mark.parametrize("rows,D,eps", [(37, 128, 1e-12), (394, 768, 1e-6), (130, 1024, 1e-12)])
def test_layernorm_fwd_bwd(ops, rows, D, eps, T16):
    x = rnd(rows, D, seed=1, dtype=T16)
    g = (1 + 0.1 * torch.randn(D)).to(dev())
    b = (0.1 * torch.randn(D)).to(dev())
    y32 = torch.empty((rows, D), device=dev())
    y, mean, rstd = ops.layernorm_fwd(x, g, b, eps, y_f32=y32)
    xf = x.float().requires_grad_(True)
    gf, bf = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xf, (D,), gf, bf, eps)
    close(y, ref, 8e-3, 8e-3, "ln fwd")
    close(y32, ref, 1e-5, 1e-5, "ln fwd (unrounded copy)")
    close(mean, xf.mean(1), 1e-5, 1e-5, "ln mean")
    dy = rnd(rows, D, seed=2, dtype=T16)
    add = rnd(rows, D, seed=3, dtype=T16)
    ref.backward(dy.float())
    n_part = 16
    part = torch.zeros((2, n_part, D), dtype=F32, device=dev())
    dx = ops.layernorm_bwd(dy, x, g, mean, rstd, part)
    close(dx, xf.grad, 1e-2, 1e-2, "ln dx")
    dx2 = ops.layernorm_bwd(dy, x, g, mean, rstd, part, dx_add=add)
    close(dx2, xf.grad + add.float(), 1e-2, 1.5e-2, "ln dx + add")
    dg, db = torch.empty(D, device=dev()), torch.empty(D, device=dev())
    ops.colsum_partials([(part, dg, db)], n_part, D, scale=0.5)
    close(dg, 0.5 * gf.grad, 1e-3, 1e-3 * math.sqrt(rows), "ln dgamma")
    close(db, 0.5 * bf.grad, 1e-3, 1e-3 * math.sqrt(rows), "ln dbeta")


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------

def _attn_ref(qkv, mask, B, S, H):
    q, k, v = qkv.float().view(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)     # [B,H,S,64]
    sc = q @ k.transpose(-1, -2) / 8.0
    if mask is not None:
        sc = sc + (1.0 - mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    p = torch.softmax(sc, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * S, H * 64)
    lse = torch.logsumexp(sc, -1)
    return o, lse


@BOTH
@pytest.mark.parametrize("B,S,H,masked", [(2, 128, 2, True), (2, 197, 3, False), (3, 16, 2, True), (2, 5, 2, False),
                                           (1, 256, 1, True), (1, 577, 1, False)])
def test_attention_fwd_bwd(ops, B, S, H, masked, T16):
    qkv = rnd(B * S, 3 * H * 64, seed=S, dtype=T16)
    mask = None
    if masked:
        lens = torch.randint(max(1, S // 8), S + 1, (B,), generator=torch.Generator().manual_seed(S))
        mask = (torch.arange(S)[None] < lens[:, None]).to(torch.int64).to(dev())
    out, lse = ops.attn_fwd(qkv, mask, B, S, H)
    qf = qkv.float().requires_grad_(True)
    ref, lse_ref = _attn_ref(qf, mask, B, S, H)
    close(out, ref, 1e-2, 6e-3, "attention out")
    close(lse, lse_ref, 1e-3, 2e-3, "attention lse")
    dout = rnd(B * S, H * 64, seed=S + 1, dtype=T16)
    ref.backward(dout.float())
    dqkv = ops.attn_bwd(qkv, mask, out, dout, lse, B, S, H)
    scale = float(qf.grad.abs().max())
    close(dqkv, qf.grad, 2e-2, 1.5e-2 * scale, "attention dqkv")


def test_attention_exact_integer(ops):
    """One-hot value rows + constant scores: O must equal the exact mean of V rows (layout check
    with asymmetric data, independent of exp precision)."""
    B, S, H = 1, 64, 1
    qkv = torch.zeros((B * S, 3 * 64), dtype=BF16, device=dev())
    v = torch.arange(S * 64, dtype=torch.float32).view(S, 64) % 7 - 3           # small exact integers
    qkv[:, 128:192] = v.to(BF16)
    out, lse = ops.attn_fwd(qkv, None, B, S, H)
    ref = v.mean(0, keepdim=True).expand(S, 64)
    close(out, ref, 4e-3, 1e-6, "uniform attention = mean of V")
    close(lse, torch.full((1, 1, S), math.log(S)), 1e-6, 1e-5, "lse of zeros")


# ------------------------------------------------------------------------------------------------
# embeddings, patches
# ------------------------------------------------------------------------------------------------

def test_bert_embed_fwd_bwd(ops):
    B, S, D, V, P = 3, 16, 128, 50, 32
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(1, V, (B, S), generator=g)
    ids[:, -3:] = 0                       # PAD tail
    ids[0, 1] = ids[1, 2] = ids[2, 5] = 7  # duplicates across samples
    ids = ids.to(dev())
    word = torch.randn(V, D, generator=g).to(dev())
    pos = torch.randn(P, D, generator=g).to(dev())
    typ = torch.randn(2, D, generator=g).to(dev())
    gam = (1 + 0.1 * torch.randn(D, generator=g)).to(dev())
    bet = (0.1 * torch.randn(D, generator=g)).to(dev())
    pre = torch.empty((B * S, D), dtype=BF16, device=dev())
    y = torch.empty_like(pre)
    mean, rstd = torch.empty(B * S, device=dev()), torch.empty(B * S, device=dev())
    ops.bert_embed_fwd(ids, word, pos, typ[0], gam, bet, 1e-12, pre, y, mean, rstd)
    wf = word.clone().requires_grad_(True)
    pf = pos.clone().requires_grad_(True)
    tf = typ.clone().requires_grad_(True)
    e = torch.nn.functional.embedding(ids, wf, padding_idx=0) + pf[:S][None] + tf[0][None, None]
    close(pre, e.reshape(B * S, D), 4e-3, 4e-3, "pre-LN sum")
    close(y, torch.nn.functional.layer_norm(e, (D,), gam, bet, 1e-12).reshape(B * S, D), 8e-3, 8e-3, "embed LN")
    d_pre = rnd(B * S, D, seed=5)
    e.backward(d_pre.float().view(B, S, D))
    dword = torch.zeros((V, D), device=dev())
    dpos = torch.zeros((P, D), device=dev())
    dtyp = torch.zeros(D, device=dev())
    ops.bert_embed_bwd(ids, d_pre, dword, dpos, dtyp, 0, scale=0.5)
    close(2 * dword, wf.grad, 1e-6, 1e-5, "dword with scale")
    ops.bert_embed_bwd(ids, d_pre, dword, dpos, dtyp, 0)
    close(dword, wf.grad, 1e-6, 1e-5, "dword (dups summed, PAD row zero)")
    assert float(dword[0].abs().max()) == 0.0
    close(dpos, pf.grad, 1e-6, 1e-5, "dpos")
    close(dtyp, tf.grad[0], 1e-5, 1e-4, "dtype0")
    # run-to-run bitwise reproducibility (no atomics)
    dword2 = torch.zeros_like(dword)
    ops.bert_embed_bwd(ids, d_pre, dword2, dpos, dtyp, 0)
    assert torch.equal(dword, dword2)
    ops.zero_rows(ids.view(-1), dword)
    assert float(dword.abs().max()) == 0.0


def test_embed_bwd_over_gathered_ranks_equals_sum_of_ranks(ops):
    """Data parallel: the table gradients are computed once from the all-gathered (ids, gradient rows) of every rank;
    that must equal the sum of the per-rank gradients (what an all-reduce of the dense tables would give)."""
    W, B, S, D, V, P = 3, 2, 16, 128, 40, 32
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, V, (W * B, S), generator=g).to(dev())          # many duplicates across "ranks", some PAD
    d_pre = rnd(W * B * S, D, seed=2)
    tot_w, tot_p, tot_t = (torch.zeros((V, D), device=dev()), torch.zeros((P, D), device=dev()), torch.zeros(D, device=dev()))
    for r in range(W):
        dw, dp, dt = torch.zeros((V, D), device=dev()), torch.zeros((P, D), device=dev()), torch.zeros(D, device=dev())
        ops.bert_embed_bwd(ids[r * B:(r + 1) * B].contiguous(), d_pre[r * B * S:(r + 1) * B * S].contiguous(), dw, dp, dt, 0)
        tot_w += dw; tot_p += dp; tot_t += dt
    aw, ap, at = torch.zeros((V, D), device=dev()), torch.zeros((P, D), device=dev()), torch.zeros(D, device=dev())
    ops.bert_embed_bwd(ids, d_pre, aw, ap, at, 0)
    close(aw, tot_w, 1e-5, 1e-5, "gathered dword")
    close(ap, tot_p, 1e-5, 1e-5, "gathered dpos")
    close(at, tot_t, 1e-5, 1e-4, "gathered dtype0")


def test_embed_bwd_indexed_kernel_is_bit_identical_and_restores_its_index(ops):
    """The linear-time embedding gradient (first-position / multiplicity index per vocabulary row) sums duplicates in
    the same position order as the owner-scan kernel: bit-identical tables; the index is INT32_MAX / 0 again after."""
    for (B, S, D, V) in ((3, 16, 128, 40), (64, 128, 256, 500), (8, 512, 768, 64000)):
        g = torch.Generator().manual_seed(B + S)
        ids = torch.randint(0, V, (B, S), generator=g).to(dev())
        ids[:, S // 2:] = torch.where(torch.rand((B, S - S // 2), generator=g).to(dev()) < 0.5, 0, ids[:, S // 2:])   # PAD
        ids[:, 0] = 2                                                                                               # CLS everywhere
        d_pre = rnd(B * S, D, seed=3)
        first = torch.full((V,), 0x7fffffff, dtype=torch.int32, device=dev())
        count = torch.zeros(V, dtype=torch.int32, device=dev())
        live = torch.zeros(V, dtype=torch.uint8, device=dev())
        a, b = torch.zeros((V, D), device=dev()), torch.zeros((V, D), device=dev())
        pa, pb = torch.zeros((S, D), device=dev()), torch.zeros((S, D), device=dev())
        ops.bert_embed_bwd(ids, d_pre, a, pa, None, 0)
        for _ in range(2):          # twice: the second call runs on the restored index
            b.zero_()
            ops.bert_embed_bwd(ids, d_pre, b, pb, None, 0, row_live=live, index=(first, count))
            torch.cuda.synchronize()
            assert torch.equal(a, b) and torch.equal(pa, pb)
            assert int((first != 0x7fffffff).sum()) == 0 and int(count.abs().sum()) == 0
        used = torch.unique(ids[ids != 0])
        assert int(live.sum()) == used.numel() and bool(live[used.long()].all())


def test_patchify_bit_exact_and_assemble(ops, golden_dir):
    z = np.load(os.path.join(golden_dir, "index_fixtures.npz"))
    img = torch.from_numpy(z["counting_image"]).to(dev())
    out = ops.patchify(img, 16)
    want = torch.from_numpy(z["patches"]).reshape(-1, 768).to(BF16)     # RNE cast of the exact gather
    assert torch.equal(out.cpu(), want), "patch order / feature order"
    B, Np, D = 2, 4, 128
    proj = rnd(B * Np, D, seed=1)
    cls, pos = torch.randn(D).to(dev()), torch.randn(Np + 1, D).to(dev())
    x = torch.empty((B * (Np + 1), D), dtype=BF16, device=dev())
    ops.vit_assemble_fwd(proj, cls, pos, x, B, Np, D)
    ref = torch.cat([cls.expand(B, 1, D), proj.float().view(B, Np, D)], 1) + pos[None]
    close(x, ref.reshape(-1, D), 4e-3, 4e-3, "assemble fwd")
    dx = rnd(B * (Np + 1), D, seed=2)
    dproj = torch.empty_like(proj)
    dcls, dpos = torch.empty(D, device=dev()), torch.empty((Np + 1, D), device=dev())
    ops.vit_assemble_bwd(dx, dproj, dcls, dpos, B, Np, D)
    dxf = dx.float().view(B, Np + 1, D)
    assert torch.equal(dproj.view(B, Np, D), dx.view(B, Np + 1, D)[:, 1:])
    close(dpos, dxf.sum(0), 1e-6, 1e-5, "dpos")
    close(dcls, dxf[:, 0].sum(0), 1e-6, 1e-5, "dcls")


# ------------------------------------------------------------------------------------------------
# head, loss, optimizer
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("B,pool", [(4, 0), (32, 15), (40, 3)])
def test_head_ce_fwd_bwd(ops, B, pool):
    S, Nt, Dt, Di, P, Cn = 16, 5, 128, 256, 64, 2
    g = torch.Generator().manual_seed(B)
    th, ih = rnd(B * S, Dt, seed=1), rnd(B * Nt, Di, seed=2)
    shapes = [(P, Dt), (P,), (P, Di), (P,), (P, 2 * P), (P,), (Cn, P), (Cn,)]
    params = [(torch.randn(s, generator=g) * 0.1).to(dev()) for s in shapes]
    grads = [torch.empty_like(p) for p in params]
    pooled = torch.empty((B, Dt + Di), device=dev())
    feat, fused = torch.empty((B, 2 * P), device=dev()), torch.empty((B, P), device=dev())
    logits = torch.empty((B, Cn), device=dev())
    ops.head_fwd(params, th.float(), ih.float(), pool, pooled, feat, fused, logits, B, S, Nt, Dt, Di, P, Cn)
    labels = torch.randint(0, Cn, (B,), generator=g).to(dev())
    loss = torch.empty(1, device=dev())
    dlogits = torch.empty_like(logits)
    ncorr = torch.empty(1, dtype=torch.int32, device=dev())
    ops.ce_fwd_bwd(logits, labels, loss, dlogits, ncorr)
    # reference
    pr = [p.clone().requires_grad_(True) for p in params]
    thf = th.float().view(B, S, Dt).requires_grad_(True)
    ihf = ih.float().view(B, Nt, Di).requires_grad_(True)
    t = torch.nn.functional.linear(thf[:, pool], pr[0], pr[1])
    v = torch.nn.functional.linear(ihf[:, 0], pr[2], pr[3])
    f = torch.nn.functional.linear(torch.cat((t, v), 1), pr[4], pr[5])
    zl = torch.nn.functional.linear(f, pr[6], pr[7])
    rl = torch.nn.functional.cross_entropy(zl, labels)
    rl.backward()
    close(logits, zl, 1e-4, 1e-5, "logits")
    close(loss, rl.reshape(1), 1e-5, 1e-6, "loss")
    assert int(ncorr) == int((zl.argmax(1) == labels).sum())
    dth = torch.zeros((B * S, Dt), dtype=BF16, device=dev())
    dih = torch.zeros((B * Nt, Di), dtype=BF16, device=dev())
    dfeat, dfused = torch.empty_like(feat), torch.empty_like(fused)
    ops.head_bwd(params, grads, dlogits, pooled, feat, fused, dfeat, dfused, dth, dih, pool, B, S, Nt, Dt, Di, P, Cn,
                 out_scale=4.0)
    dth, dih = (dth.float() / 4).to(BF16), (dih.float() / 4).to(BF16)
    for gr, p, nm in zip(grads, pr, ("Wt", "bt", "Wi", "bi", "Wf", "bf", "Wo", "bo")):
        close(gr, p.grad, 1e-4, 1e-6, "grad " + nm)
    close(dth.view(B, S, Dt), thf.grad, 8e-3, 1e-6, "d text hidden")
    close(dih.view(B, Nt, Di), ihf.grad, 8e-3, 1e-6, "d image hidden")


def test_adam_sumsq_cast(ops):
    n = 4 * 1237
    g0 = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g0)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3)
    p, m, v = p0.clone().to(dev()), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    shadow = torch.empty(n, dtype=BF16, device=dev())
    ws, nrm = torch.empty(1024, device=dev()), torch.empty(1, device=dev())
    for step in range(1, 4):
        g = torch.randn(n, generator=g0) * (0.0 if step == 2 else 1.0)
        ref.grad = g.clone()
        opt.step()
        gd = g.to(dev())
        ops.sumsq(gd, ws, nrm)
        close(nrm, (g.double() ** 2).sum().float().reshape(1), 1e-5, 1e-6, "sumsq")
        hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 0.0, 1 / (1 - 0.9 ** step), 1 / math.sqrt(1 - 0.999 ** step), 1.0]).to(dev())
        ops.adam_step(p, m, v, gd, shadow, n, hyper)
        close(p, ref.detach(), 1e-6, 1e-7, f"adam step {step}")
        assert torch.equal(shadow, p.to(BF16))
    # clipping path == clip_grad_norm_ + Adam
    ref2 = torch.nn.Parameter(p0.clone())
    opt2 = torch.optim.AdamW([ref2], lr=1e-3, weight_decay=0.01)
    g = torch.randn(n, generator=g0) * 3
    ref2.grad = g.clone()
    torch.nn.utils.clip_grad_norm_([ref2], 1.0)
    opt2.step()
    p, m, v = p0.clone().to(dev()), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    gd = g.to(dev())
    ops.sumsq(gd, ws, nrm)
    hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 0.01, 1 / (1 - 0.9), 1 / math.sqrt(1 - 0.999), 1.0]).to(dev())
    ops.adam_step(p, m, v, gd, None, 0, hyper, decoupled=True, gnorm_sq=nrm, max_norm=1.0)
    close(p, ref2.detach(), 1e-6, 1e-7, "adamw + clip")


def test_gemm_split_k_weight_gradient(ops):
    """Split-K inside the grouped GEMM (conv weight gradients: few output tiles, contraction = B*H*W): exact on small integers,
    equal to the unsplit launch, ragged contraction lengths."""
    for (M, N, K) in ((64, 64, 5000), (64, 576, 12544), (256, 64, 777)):
        dy, x = ints(K, M, seed=K), ints(K, N, seed=K + 1)
        want = dy.float().t() @ x.float()
        got = ops.wgrad_splitk(dy, x, M, N, K, M, N)
        assert torch.equal(got, want), (M, N, K)
        got2 = ops.wgrad_splitk(dy, x, M, N, K, M, N, alpha=0.5, target_tiles=8)
        assert torch.equal(got2, 0.5 * want)
