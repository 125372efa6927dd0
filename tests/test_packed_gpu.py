"""Padding-free text tower (mh_pack_plan / packed attention / device-side live row counts).

BertModel computes every padded position and then ignores it; the HIP path never computes rows with
attention_mask == 0 (reference call: self.bert(text, attention_mask=mask), Multimodal_example_task2C.txt:175).
Checked here:
  * the row bookkeeping is bit-exact against a numpy restatement (prefix masks, masks with holes, all ones,
    one-token sequences, both pooling positions);
  * pack -> unpack round trips;
  * packed attention == dense masked attention on the kept rows;
  * a GEMM launch clamped by the device-side row count touches exactly the live rows (fwd) / contracts over them (wgrad);
  * the whole step, packed vs dense plan of the same model: bit-identical logits, matching gradients, with and
    without dropout, for masks with holes and for the organizers' last-position pooling;
  * packed step vs the CPU oracle on a mask with holes (the oracle applies the additive mask the way the
    reference's BertModel does).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import multimodal_propaganda_meme_classification_amd as m
    return m


def _oracle():
    from oracle import meme_oracle as O
    return O


def _masks(B, S, kind, seed):
    g = np.random.default_rng(seed)
    if kind == "prefix":
        lens = g.integers(1, S + 1, size=B)
        m = (np.arange(S)[None] < lens[:, None])
    elif kind == "holes":
        m = g.random((B, S)) < 0.6
        m[:, 0] = True
    elif kind == "ones":
        m = np.ones((B, S), bool)
    elif kind == "single":
        m = np.zeros((B, S), bool)
        m[:, 0] = True
    else:
        raise ValueError(kind)
    return m.astype(np.int64)


def _plan_numpy(mask, pool):
    B, S = mask.shape
    keep = (mask != 0) | (np.arange(S)[None] == pool)
    cu = np.zeros(B + 1, np.int32)
    cu[1:] = np.cumsum(keep.sum(1))
    n = int(cu[-1])
    row_map = np.full(B * S, -1, np.int32)
    inv_map = np.full(B * S, -1, np.int32)
    pmask = np.zeros(B * S, np.int64)
    dense = np.flatnonzero(keep.reshape(-1))
    row_map[:n] = dense
    inv_map[dense] = np.arange(n, dtype=np.int32)
    pmask[:n] = (mask.reshape(-1)[dense] != 0)
    pool_rows = inv_map[np.arange(B) * S + pool]
    return dict(cu=cu, row_map=row_map, inv_map=inv_map, pmask=pmask, pool_rows=pool_rows, n_rows=np.array([n], np.int32))


@pytest.mark.parametrize("kind", ["prefix", "holes", "ones", "single"])
@pytest.mark.parametrize("B,S", [(1, 8), (5, 16), (32, 128), (70, 200), (1024, 64)])
def test_pack_plan_bit_exact(pkg, kind, B, S):
    from multimodal_propaganda_meme_classification_amd import ops
    for pool in (0, S - 1):
        mask = _masks(B, S, kind, seed=B * 1000 + S + pool)
        got = ops.pack_plan(torch.from_numpy(mask).cuda(), pool)
        want = _plan_numpy(mask, pool)
        for k, v in want.items():
            assert np.array_equal(got[k].cpu().numpy(), v), (kind, B, S, pool, k)


def test_pack_unpack_round_trip(pkg):
    from multimodal_propaganda_meme_classification_amd import ops
    B, S, D = 7, 40, 256
    mask = _masks(B, S, "holes", 3)
    plan = ops.pack_plan(torch.from_numpy(mask).cuda(), 0)
    x = torch.randn((B * S, D), device="cuda").to(torch.bfloat16)
    packed = ops.pack_rows(x, plan, D)
    n = int(plan["n_rows"])
    rm = plan["row_map"][:n].long()
    assert torch.equal(packed[:n], x[rm])
    back = ops.unpack_rows(packed, plan, D)
    keep = torch.from_numpy(mask.reshape(-1) != 0).cuda()
    assert torch.equal(back[keep], x[keep])
    assert float(back[~keep].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kind,B,S,H", [("prefix", 6, 128, 4), ("holes", 4, 64, 2), ("prefix", 3, 200, 2), ("prefix", 2, 300, 2)])
def test_packed_attention_equals_dense_on_kept_rows(pkg, dtype, kind, B, S, H):
    from multimodal_propaganda_meme_classification_amd import ops
    mask_np = _masks(B, S, kind, 11 + S)
    mask = torch.from_numpy(mask_np).cuda()
    plan = ops.pack_plan(mask, 0)
    n = int(plan["n_rows"])
    rm = plan["row_map"][:n].long()
    qkv = (torch.randn((B * S, 3 * H * 64), device="cuda") * 0.7).to(dtype)
    dout = (torch.randn((B * S, H * 64), device="cuda") * 0.1).to(dtype)
    keep = (mask.reshape(-1) != 0)
    dout = dout * keep[:, None].to(dtype)          # nothing downstream reads the padded rows: their gradient is zero
    out_d, lse_d = ops.attn_fwd(qkv, mask, B, S, H)
    dqkv_d = ops.attn_bwd(qkv, mask, out_d, dout, lse_d, B, S, H)
    qkv_p, dout_p = ops.pack_rows(qkv, plan, 3 * H * 64), ops.pack_rows(dout, plan, H * 64)
    out_p, lse_p = ops.attn_fwd_packed(qkv_p, plan, B, S, H)
    dqkv_p = ops.attn_bwd_packed(qkv_p, plan, out_p, dout_p, lse_p, B, S, H)
    torch.cuda.synchronize()
    if kind == "prefix":      # same keys in the same 32-key sub-tiles: the same arithmetic
        assert torch.equal(out_p[:n], out_d[rm])
    else:                     # holes: the kept keys are compacted, the online softmax visits them in other sub-tiles
        assert float((out_p[:n].float() - out_d[rm].float()).abs().max()) <= 8e-3 * float(out_d.float().abs().max())
    got, want = dqkv_p[:n].float(), dqkv_d[rm].float()
    # dQ rows are computed identically; dK / dV sum the same non-zero terms (padded queries contribute exact zeros)
    rel = 2e-3 if kind == "prefix" else 1e-2      # holes: other sub-tile order, a 16-bit ulp of the stored gradient
    assert float((got - want).abs().max()) <= rel * float(want.abs().max()) + 1e-6


def test_gemm_clamps_to_live_rows(pkg):
    from multimodal_propaganda_meme_classification_amd import ops
    T, K, N, live = 512, 256, 384, 137
    n_dev = torch.tensor([live], dtype=torch.int32, device="cuda")
    x = torch.randn((T, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.full((T, N), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.gemm_grouped([ops.Gemm(x, w, y, T, N, K, K, K, N, rows_dev=n_dev)], False, False)
    ref = (x.float() @ w.float().t())
    assert float((y[:live].float() - ref[:live]).abs().max()) < 0.05
    assert float((y[live:].float() - 7.0).abs().max()) == 0.0            # rows past the live count are never written
    # wgrad: the contraction stops at the live rows
    dy = (torch.randn((T, N), device="cuda") * 0.1).to(torch.bfloat16)
    dw = torch.zeros((N, K), device="cuda", dtype=torch.float32)
    db = torch.zeros((N,), device="cuda", dtype=torch.float32)
    ops.gemm_grouped([ops.Gemm(dy, x, dw, N, K, T, N, K, K, rowsum=db, rows_dev=n_dev)], True, True)
    ref_dw = dy[:live].float().t() @ x[:live].float()
    assert float((dw - ref_dw).abs().max()) <= 2e-2 * float(ref_dw.abs().max())
    assert float((db - dy[:live].float().sum(0)).abs().max()) <= 2e-2 * float(db.abs().max())


def _tiny(pkg, O, pool, seed=21, dropout=False):
    cfg = O.tiny_config(pool)
    params = O.init_params(cfg, seed)
    mc = pkg.ModelConfig.from_dict(cfg.to_dict())
    if dropout:
        mc.with_reference_dropout()
    model = pkg.MultimodalClassifier.from_config(mc, init=False)
    model.load_state_dict(params)
    model.to("cuda")
    return model, params, cfg


@pytest.mark.parametrize("pool", ["cls", "last"])
@pytest.mark.parametrize("kind", ["prefix", "holes"])
@pytest.mark.parametrize("dropout", [False, True])
def test_packed_step_equals_dense_step(pkg, pool, kind, dropout):
    O = _oracle()
    B, S = 6, 32
    text, image, _, labels = O.synthetic_batch(O.tiny_config(pool), B, S, seed=77)
    mask = torch.from_numpy(_masks(B, S, kind, 5))
    if pool == "cls":
        mask[:, 0] = 1
    text = text * mask
    dev = [t.cuda() for t in (text, image, mask, labels)]
    outs = []
    for pack in (True, False):
        model, _, _ = _tiny(pkg, O, pool, dropout=dropout)
        model.manual_seed(1234)
        eng = model._get_engine()
        eng.pack_text = pack
        model.train()
        loss, _, logits = model.forward_backward(*dev)
        torch.cuda.synchronize()
        assert model._get_engine().plan(B, S, True).packed == pack
        outs.append((float(loss), logits.detach().float().cpu().clone(), model.flat_grads.detach().float().cpu().clone()))
    (l1, z1, g1), (l2, z2, g2) = outs
    if kind == "prefix":
        assert torch.equal(z1, z2), (z1, z2)      # every kept row goes through the same arithmetic
        assert l1 == l2
    else:                                         # compacted keys: another online-softmax order (a few 16-bit ulps)
        assert float((z1 - z2).abs().max()) <= 2e-3, (z1, z2)
        assert abs(l1 - l2) <= 2e-3
    # weight gradients sum the same non-zero terms in a different K-tile partition
    rel = 2e-3 if kind == "prefix" else 2e-2
    assert float((g1 - g2).norm()) <= rel * float(g2.norm())
    assert float((g1 - g2).abs().max()) <= 2.5 * rel * float(g2.abs().max())


def test_packed_step_with_mask_holes_matches_oracle(pkg):
    if os.environ.get("MEMEHIP_PACK_TEXT", "1") == "0":
        pytest.skip("row packing switched off by MEMEHIP_PACK_TEXT=0")
    O = _oracle()
    B, S = 5, 24
    cfg = O.tiny_config("cls")
    text, image, _, labels = O.synthetic_batch(cfg, B, S, seed=31)
    mask = torch.from_numpy(_masks(B, S, "holes", 9))
    text = text * mask
    model, params, _ = _tiny(pkg, O, "cls", seed=13)
    assert model._get_engine().pack_text
    model.train()
    loss, _, logits = model.forward_backward(text.cuda(), image.cuda(), mask.cuda(), labels.cuda())
    ref_logits, ref_loss, ref_grads = O.loss_and_grads(params, text, image, mask, labels, cfg)
    tol = 3e-3
    assert float((logits.detach().float().cpu() - ref_logits).abs().max()) <= tol
    assert abs(float(loss) - float(ref_loss)) <= tol
    for name, p in model.named_parameters():
        ref = ref_grads[name]
        got = p.grad.detach().float().cpu()
        if ".key.bias" in name:
            continue
        assert float((got - ref).norm()) <= 3e-2 * float(ref.norm()) + 2e-6, name


def test_adam_skipping_untouched_embedding_rows_is_bit_identical(pkg):
    """torch.optim.Adam (dense, Multimodal_example_task2C.txt:249) leaves a parameter with g = m = v = 0 unchanged;
    the fused Adam skips word-embedding rows that never received a gradient.  Same numbers, bit for bit, over steps
    whose batches touch different rows (rows touched once keep being updated while their moments decay)."""
    O = _oracle()
    cfg = O.tiny_config("cls")
    models, opts = [], []
    for skip in (True, False):
        model, _, _ = _tiny(pkg, O, "cls", seed=3)
        models.append(model)
        opts.append(pkg.Adam(model.parameters(), lr=1e-3, model=model, skip_untouched_embedding_rows=skip))
    for step in range(4):
        text, image, mask, labels = O.synthetic_batch(cfg, 4, 16, seed=100 + step)
        dev = [t.cuda() for t in (text, image, mask, labels)]
        for model, opt in zip(models, opts):
            model.train()
            model.forward_backward(*dev)
            opt.step()
        torch.cuda.synchronize()
        assert torch.equal(models[0].flat_params, models[1].flat_params), step
        assert torch.equal(opts[0]._flat["M"], opts[1]._flat["M"]) and torch.equal(opts[0]._flat["V"], opts[1]._flat["V"])
    live = opts[0]._flat["row_live"]
    assert 0 < int(live.sum()) < live.numel()          # some rows updated, most of the table skipped
    assert opts[1]._flat.get("row_live") is None or int(opts[1]._flat["row_live"].sum()) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("packed", [True, False])
def test_grouped_attention_equals_separate_launches(pkg, dtype, packed):
    """The dual launch (ViT heads + text heads in one grid) runs the same per-head code as the two separate launches."""
    from multimodal_propaganda_meme_classification_amd import ops
    B, Hn, Si, St = 5, 3, 197, 128
    mask = torch.from_numpy(_masks(B, St, "prefix", 41)).cuda()
    plan = ops.pack_plan(mask, 0)
    qi = (torch.randn((B * Si, 3 * Hn * 64), device="cuda") * 0.7).to(dtype)
    qt = (torch.randn((B * St, 3 * Hn * 64), device="cuda") * 0.7).to(dtype)
    doi = (torch.randn((B * Si, Hn * 64), device="cuda") * 0.1).to(dtype)
    dot = (torch.randn((B * St, Hn * 64), device="cuda") * 0.1).to(dtype)
    rng = torch.tensor([123, 0, 7, 0], dtype=torch.int32, device="cuda")
    for drop in (None, (rng, 0.1, 17)):
        # separate launches
        oi, li = ops.attn_fwd(qi, None, B, Si, Hn)
        di = ops.attn_bwd(qi, None, oi, doi, li, B, Si, Hn)
        if packed:
            ot, lt = ops.attn_fwd_packed(qt, plan, B, St, Hn, drop=drop)
            dt = ops.attn_bwd_packed(qt, plan, ot, dot, lt, B, St, Hn, drop=drop)
            extra = dict(key_mask=plan["pmask"], cu=plan["cu"], row_map=plan["row_map"])
        else:
            ot, lt = ops.attn_fwd(qt, mask, B, St, Hn, drop=drop)
            dt = ops.attn_bwd(qt, mask, ot, dot, lt, B, St, Hn, drop=drop)
            extra = dict(key_mask=mask)
        # grouped
        z = lambda t: torch.zeros_like(t)
        pi = dict(qkv=qi, out=z(oi), lse=z(li), dout=doi, delta=z(li), dqkv=z(di), B=B, S=Si, H=Hn)
        pt = dict(qkv=qt, out=z(ot), lse=z(lt), dout=dot, delta=z(lt), dqkv=z(dt), B=B, S=St, H=Hn, drop=drop, **extra)
        for order in ([pi, pt], [pt, pi]):
            for d in order:
                d["out"].zero_(), d["dqkv"].zero_()
            ops.attn_grouped(order, backward=False)
            ops.attn_grouped(order, backward=True)
            torch.cuda.synchronize()
            n = int(plan["n_rows"]) if packed else B * St
            assert torch.equal(pi["out"], oi) and torch.equal(pi["dqkv"], di)
            assert torch.equal(pt["out"][:n], ot[:n]) and torch.equal(pt["dqkv"][:n], dt[:n])


def test_one_graph_serves_batches_with_different_masks(pkg):
    """The packed plan is captured ONCE into a hipGraph; the live row count is read on the device at replay time, so
    batches with other attention masks (other numbers of live rows) replay the same graph.  Must equal the eager
    autograd-style loop step for step (same kernels, same order: bit-identical parameters)."""
    if os.environ.get("MEMEHIP_PACK_TEXT", "1") == "0":
        pytest.skip("row packing switched off by MEMEHIP_PACK_TEXT=0")
    O = _oracle()
    cfg = O.tiny_config("cls")
    B, S = 6, 32
    m_graph, _, _ = _tiny(pkg, O, "cls", seed=5)
    m_eager, _, _ = _tiny(pkg, O, "cls", seed=5)
    o_graph = pkg.Adam(m_graph.parameters(), lr=1e-3, model=m_graph)
    o_eager = pkg.Adam(m_eager.parameters(), lr=1e-3, model=m_eager)
    gs = pkg.GraphedStep(m_graph, o_graph, B, S, use_graph=True)
    live = []
    for step, kind in enumerate(["prefix", "ones", "holes", "single", "prefix"]):
        text, image, _, labels = O.synthetic_batch(cfg, B, S, seed=50 + step)
        mask = torch.from_numpy(_masks(B, S, kind, 70 + step))
        mask[:, 0] = 1
        text = text * mask
        dev = [t.cuda() for t in (text, image, mask, labels)]
        gs.load_batch(*dev)
        loss_g, _ = gs.step()
        m_eager.train()
        loss_e, _, _ = m_eager.forward_backward(*dev)
        o_eager.step()
        torch.cuda.synchronize()
        live.append(int(gs.plan.buf["pk.n_rows"]))
        assert live[-1] == int(mask.sum())
        assert float(loss_g) == float(loss_e), (step, kind)
        assert torch.equal(m_graph.flat_params, m_eager.flat_params), (step, kind)
    assert len(set(live)) >= 4 and gs.graphs is not None and len(gs.graphs) == 1       # ONE captured graph for all of them
