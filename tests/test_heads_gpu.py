"""Heads on the HIP towers (SURVEY section 8 f ranks 1-2, boundary 8 b) on the MI355X.

The poolings of LLMWithClassificationHead, ConcatAttention3 and get_params are checked against
tests/golden/ref_kevin_heads.npz -- vectors produced by RUNNING the reference's own class definitions
(oracle/gen_ref_fixtures.py extracts them from example_scripts/Multimodal_example_task2C.py).  The fp32 GEMM under them
is checked for exactness, the three-tower model of ...task2C.py:587-685 end to end against the CPU oracle towers + the
same head in PyTorch on the CPU, and the HF-Trainer protocol of DistilBERT_example_task2A.py:159-183 on real dev texts.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
F32 = torch.float32


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import multimodal_propaganda_meme_classification_amd as m
    return m


@pytest.fixture(scope="module")
def ref(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_kevin_heads.npz"))


def _oracle():
    from oracle import meme_oracle as O
    return O


def close(a, b, atol, rtol=1e-4, what=""):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    err = (a - b).abs()
    assert bool((err <= atol + rtol * b.abs()).all()), f"{what}: max err {float(err.max()):.3e} (|ref| max {float(b.abs().max()):.3e})"


# ---------------------------------------------------------------------------------------------------
# mh_gemm_f32
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(32, 512, 768), (5, 1, 512), (70, 130, 33), (4096, 64, 128), (64, 64, 32)])
def test_gemm_f32_matches_fp64_in_every_layout(pkg, M, N, K):
    fused = pkg.fused
    g = torch.Generator().manual_seed(M + N + K)
    a, b = torch.randn((M, K), generator=g), torch.randn((N, K), generator=g)
    bias = torch.randn((N,), generator=g)
    want = (a.double() @ b.double().t() + bias.double())
    scale = float(want.abs().max())
    for ak, bk in ((False, False), (False, True), (True, True), (True, False)):
        A = (a.t().contiguous() if ak else a).cuda()
        Bm = (b.t().contiguous() if bk else b).cuda()
        out = torch.full((M, N), 7.0, device="cuda")
        fused.gemm_f32(A, Bm, out, M, N, K, M if ak else K, N if bk else K, N, a_kmajor=ak, b_kmajor=bk, bias=bias.cuda())
        assert float((out.double().cpu() - want).abs().max()) < 2e-6 * scale * max(1.0, K ** 0.5 / 8), (ak, bk)
    # exactness: small integers are summed without any rounding, whatever the order
    ai, bi = torch.randint(-3, 4, (M, K), generator=g).float(), torch.randint(-3, 4, (N, K), generator=g).float()
    out = torch.empty((M, N), device="cuda")
    fused.gemm_f32(ai.cuda(), bi.cuda(), out, M, N, K, K, K, N)
    assert torch.equal(out.cpu(), ai @ bi.t())
    # accumulate + relu / tanh epilogues
    base = torch.randn((M, N), generator=g)
    out = base.clone().cuda()
    fused.gemm_f32(a.cuda(), b.cuda(), out, M, N, K, K, K, N, flags=pkg._lib.MH_F32_ACCUM)
    close(out, base.double() + a.double() @ b.double().t(), 3e-6 * scale * max(1.0, K ** 0.5 / 8), what="accumulate")
    out = torch.empty((M, N), device="cuda")
    a_s = a * (1.0 / K ** 0.5)             # pre-activations of order 1: tanh is not saturated and the f32 sum is tight
    fused.gemm_f32(a_s.cuda(), b.cuda(), out, M, N, K, K, K, N, bias=bias.cuda(), flags=pkg._lib.MH_F32_TANH)
    close(out, torch.tanh(a_s.double() @ b.double().t() + bias.double()), 3e-6, what="tanh")


@pytest.mark.parametrize("B,relu", [(32, True), (8, False), (64, True), (100, True)])
def test_linear_batchnorm_relu_fused_matches_torch(pkg, B, relu):
    fused = pkg.fused
    torch.manual_seed(B)
    K, N = 96, 200
    lin, bn = nn.Linear(K, N), nn.BatchNorm1d(N)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.normal_(0, 0.2)
    lin_g, bn_g = nn.Linear(K, N).cuda(), nn.BatchNorm1d(N).cuda()
    lin_g.load_state_dict(lin.state_dict()), bn_g.load_state_dict(bn.state_dict())
    x = torch.randn(B, K)
    r = torch.randn(B, N)
    for step in range(2):          # second step: running statistics carried over
        xc = x.clone().requires_grad_(True)
        y = bn(lin(xc))
        y = F.relu(y) if relu else y
        (y * r).sum().backward()
        xg = x.cuda().requires_grad_(True)
        yg = fused.linear_bn_act(xg, lin_g, bn_g, relu)
        (yg * r.cuda()).sum().backward()
        close(yg, y, 2e-5, what="y")
        close(xg.grad, xc.grad, 5e-5, 2e-4, what="dx")
        for (n, p), (_, q) in zip(list(lin.named_parameters()) + list(bn.named_parameters()),
                                  list(lin_g.named_parameters()) + list(bn_g.named_parameters())):
            close(q.grad, p.grad, 1e-4, 3e-4, what=f"d{n}")
            p.grad = None
            q.grad = None
        close(bn_g.running_mean, bn.running_mean, 1e-6, what="running_mean")
        close(bn_g.running_var, bn.running_var, 1e-6, what="running_var")
        assert int(bn_g.num_batches_tracked) == int(bn.num_batches_tracked) == step + 1
    bn.eval(), bn_g.eval()
    with torch.no_grad():
        y = bn(lin(x))
        y = F.relu(y) if relu else y
        close(fused.linear_bn_act(x.cuda(), lin_g, bn_g, relu), y, 2e-5, what="eval y")


# ---------------------------------------------------------------------------------------------------
# against the reference's own classes (fixture)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["cls", "max", "mean", "attention", "cnn"])
def test_poolings_match_the_reference_class(pkg, ref, kind):
    D, A = 128, 64
    pool = pkg.SequencePooling(kind, D, A, 3)
    sd = {k[len(f"pool_{kind}_param_"):]: torch.from_numpy(ref[k]) for k in ref.files if k.startswith(f"pool_{kind}_param_")}
    pool.load_state_dict(sd)
    pool.cuda()
    h = torch.from_numpy(ref["pool_hidden"]).cuda().requires_grad_(True)
    mask = torch.from_numpy(ref["pool_mask"]).cuda()
    out = pool(h, mask)
    (out * torch.from_numpy(ref["pool_r"]).cuda()).sum().backward()
    close(out, ref[f"pool_{kind}_out"], 2e-5, what=f"{kind} out")
    close(h.grad, ref[f"pool_{kind}_dh"], 2e-5, 2e-4, what=f"{kind} dh")
    for n, p in pool.named_parameters():
        close(p.grad, ref[f"pool_{kind}_grad_{n}"], 5e-5, 3e-4, what=f"{kind} d{n}")


def test_unsupported_pooling_raises_like_the_reference(pkg, ref):
    with pytest.raises(ValueError) as e:
        pkg.SequencePooling("median", 128)(torch.zeros(1, 2, 128, device="cuda"), torch.ones(1, 2, dtype=torch.long, device="cuda"))
    assert str(e.value) == str(ref["pool_bad_message"])
    with pytest.raises(ValueError):
        pkg.TextClassifier(pkg.TextConfig(vocab_size=512, hidden=128, layers=1, heads=2, intermediate=256, max_position=64), "median")


def test_concat_attention3_matches_the_reference_class(pkg, ref):
    P = 64
    ca = pkg.ConcatAttention3(3 * P, P)
    sd = {k[len("ca_init_"):]: torch.from_numpy(ref[k]) for k in ref.files if k.startswith("ca_init_")}
    assert sorted(sd) == sorted(ca.state_dict()), "state_dict keys differ from the reference module's"
    ca.load_state_dict(sd)
    ca.cuda().train()
    feats = [torch.from_numpy(ref[f"ca_{n}"]).cuda().requires_grad_(True) for n in ("text", "image", "caption")]
    y = ca(*feats)
    (y * torch.from_numpy(ref["ca_r"]).cuda()).sum().backward()
    close(y, ref["ca_out"], 3e-5, what="out")
    for f_, n in zip(feats, ("text", "image", "caption")):
        close(f_.grad, ref[f"ca_d{n}"], 3e-5, 3e-4, what=f"d{n}")
    for n, p in ca.named_parameters():
        if n.endswith(".0.bias"):      # a Linear bias in front of BatchNorm: its gradient is analytically 0 (the batch mean
            assert float(p.grad.abs().max()) < 2e-3 and float(np.abs(ref[f"ca_grad_{n}"]).max()) < 2e-3      # is removed) --
            continue                   # both sides hold rounding noise only
        close(p.grad, ref[f"ca_grad_{n}"], 5e-5, 5e-4, what=f"d{n}")
    after = ca.state_dict()
    for k in ref.files:
        if k.startswith("ca_after_"):
            close(after[k[len("ca_after_"):]], ref[k], 1e-6, what=k)
    ca.eval()
    with torch.no_grad():
        close(ca(*[f_.detach() for f_ in feats]), ref["ca_out_eval"], 3e-5, what="eval out")


def test_mca3_matches_the_reference_class(pkg, ref):
    """fusion_method = "mca": MCA3 (Multimodal_example_task2C.py:423-448) run by oracle/gen_ref_fixtures.py on 2-D features --
    output, input gradients and every parameter gradient of the HIP path (three mh_gemm_f32 Linears, mh_mca3_fwd / _bwd,
    the reduce Linear) against the reference class's own; its [B,1,U] image broadcast makes this an attention across the
    rows of the batch, which is what is reproduced."""
    U = 64
    m = pkg.MCA3(U)
    sd = {k[len("mca_init_"):]: torch.from_numpy(ref[k]) for k in ref.files if k.startswith("mca_init_")}
    assert sorted(sd) == sorted(m.state_dict()), "state_dict keys differ from the reference module's"
    m.load_state_dict(sd)
    m.cuda().train()
    feats = [torch.from_numpy(ref[f"mca_{n}"]).cuda().requires_grad_(True) for n in ("text", "image", "caption")]
    y = m(*feats)
    (y * torch.from_numpy(ref["mca_r"]).cuda()).sum().backward()
    close(y, ref["mca_out"], 2e-5, what="out")
    for f_, n in zip(feats, ("text", "image", "caption")):
        close(f_.grad, ref[f"mca_d{n}"], 2e-5, 2e-4, what=f"d{n}")
    for n, p in m.named_parameters():
        if n == "V.bias":              # softmax is shift-invariant: analytically 0, rounding noise on both sides
            assert float(p.grad.abs().max()) < 1e-4
            continue
        close(p.grad, ref[f"mca_grad_{n}"], 3e-5, 3e-4, what=f"d{n}")


def test_kevin_model_with_mca_fusion_trains(pkg):
    model, tc, ic, cc = _small_kevin(pkg, fusion="mca")
    model.cuda().train()
    assert isinstance(model.fusion_layer, pkg.MCA3)
    g = torch.Generator().manual_seed(5)
    B = 8
    text = torch.randint(5, tc.vocab_size, (B, 16), generator=g).cuda()
    cap = torch.randint(5, cc.vocab_size, (B, 12), generator=g).cuda()
    mask, cmask = torch.ones((B, 16), dtype=torch.long).cuda(), torch.ones((B, 12), dtype=torch.long).cuda()
    image = torch.randn((B, 3, 32, 32), generator=g).cuda()
    labels = torch.tensor([0, 1, 1, 0, 1, 0, 0, 1.0]).cuda()
    opt = pkg.Adam(model.get_params(2e-3), max_grad_norm=1.0)
    crit = pkg.SigmoidFocalLoss()
    losses = []
    for _ in range(8):
        opt.zero_grad()
        out = model(text, image, mask, cap, cmask)
        loss = crit(out, labels)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert out.shape == (B,) and np.isfinite(losses).all() and losses[-1] < losses[0], losses
    with pytest.raises(ValueError, match="Unsupported fusion method"):
        pkg.KevinMultimodalClassifier("cross_modal", text=tc, image=ic, caption=cc, proj=64)


def _small_kevin(pkg, dtype="fp16", fusion="concatenation"):
    tc = pkg.TextConfig(vocab_size=600, hidden=128, layers=2, heads=2, intermediate=256, max_position=64)
    ic = pkg.ImageConfig(image_size=32, hidden=128, layers=2, heads=2, intermediate=256)
    cc = pkg.TextConfig(vocab_size=500, hidden=128, layers=1, heads=2, intermediate=256, max_position=64)
    return pkg.KevinMultimodalClassifier(fusion, text=tc, image=ic, caption=cc, proj=64, compute_dtype=dtype, seed=3), tc, ic, cc


def test_get_params_groups_like_the_reference(pkg, ref):
    model, *_ = _small_kevin(pkg)
    model.cuda()
    groups = model.get_params(1.0)
    assert [g["lr"] for g in groups] == [1.0, 0.8, 0.8]
    gid = {id(p): gi for gi, g in enumerate(groups) for p in g["params"]}
    assert len(gid) == sum(1 for _ in model.parameters()), "every parameter in exactly one group"
    # the reference's rule, from the fixture: reference module (name prefix) -> group
    rule = {}
    for n, g_ in zip(ref["gp_names"], ref["gp_group"]):
        parts = str(n).split(".")
        key = ".".join(parts[:2]) if parts[0] == "image_model" else parts[0]
        assert rule.setdefault(key, int(g_)) == int(g_)
    here_to_ref = {"towers.bert.": "text_model", "towers.image_model.": "image_model.image_model", "image_fine_tune.": "image_model.fine_tune",
                   "caption_text_model.": "caption_text_model", "text_fc.": "text_fc", "caption_text_fc.": "caption_text_fc",
                   "fusion_layer.": "fusion_layer", "output_fc.": "output_fc"}
    checked = 0
    for name, p in model.named_parameters():
        hits = [r for h, r in here_to_ref.items() if name.startswith(h)]
        if not hits:
            assert name.startswith("towers.") and ("_fc." in name), name      # the built-in head of the towers module: unused, group 0
            continue
        assert gid[id(p)] == rule[hits[0]], (name, hits[0])
        checked += 1
    assert checked > 50
    with pytest.raises(ValueError, match="Unsupported fusion method"):
        pkg.KevinMultimodalClassifier("median_fusion")


def test_kevin_three_tower_step_matches_the_cpu_composition(pkg):
    """forward(text, image, mask, caption_text, caption_text_mask) -> [B] logits, sigmoid focal loss, ONE fused Adam over
    get_params(lr) with ONE global clip norm (Multimodal_example_task2C.py:666-685,711-716) against: CPU oracle towers +
    the same head written with torch.nn on the CPU + torch clip_grad_norm_ + torch.optim.Adam with the same groups."""
    O = _oracle()
    model, tc, ic, cc = _small_kevin(pkg)
    model.text_dropout.p = model.caption_text_dropout.p = 0.0
    model.image_fine_tune[2].p = 0.0
    model.cuda().train()
    B, S = 8, 16
    g = torch.Generator().manual_seed(5)
    text = torch.randint(5, tc.vocab_size, (B, S), generator=g)
    cap = torch.randint(5, cc.vocab_size, (B, S), generator=g)
    lens, clens = torch.randint(3, S + 1, (B,), generator=g), torch.randint(3, S + 1, (B,), generator=g)
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    cmask = (torch.arange(S)[None] < clens[:, None]).long()
    text, cap = text * mask, cap * cmask
    image = torch.randn((B, 3, 32, 32), generator=g)
    labels = (torch.rand(B, generator=g) < 0.4).float()

    # ---- CPU composition with the same parameters
    ocfg = O.OracleConfig(text=O.TextConfig(**{k: getattr(tc, k) for k in ("vocab_size", "hidden", "layers", "heads", "intermediate", "max_position", "type_vocab")}),
                          image=O.ImageConfig(image_size=32, hidden=128, layers=2, heads=2, intermediate=256), proj=128)
    ccfg = O.TextConfig(**{k: getattr(cc, k) for k in ("vocab_size", "hidden", "layers", "heads", "intermediate", "max_position", "type_vocab")})
    tp = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in model.towers.state_dict().items()}
    cp = {"bert." + k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in model.caption_text_model.state_dict().items()}
    P = 64
    head = nn.ModuleDict(dict(
        text_fc=nn.Sequential(nn.Linear(128, P), nn.BatchNorm1d(P), nn.ReLU()),
        caption_text_fc=nn.Sequential(nn.Linear(128, P), nn.BatchNorm1d(P), nn.ReLU()),
        image_fine_tune=nn.Sequential(nn.Linear(128, P), nn.ReLU(), nn.Dropout(0.0), nn.Linear(P, P)),
        att=nn.Sequential(nn.Linear(3 * P, 3 * P), nn.BatchNorm1d(3 * P), nn.ReLU(), nn.Softmax(dim=1)),
        red=nn.Sequential(nn.Linear(3 * P, P), nn.BatchNorm1d(P), nn.ReLU()),
        output_fc=nn.Sequential(nn.Linear(P, 1), nn.BatchNorm1d(1))))
    head.text_fc.load_state_dict({k: v.cpu() for k, v in model.text_fc.state_dict().items()})
    head.caption_text_fc.load_state_dict({k: v.cpu() for k, v in model.caption_text_fc.state_dict().items()})
    head.image_fine_tune.load_state_dict({k: v.cpu() for k, v in model.image_fine_tune.state_dict().items()})
    head.att.load_state_dict({k: v.cpu() for k, v in model.fusion_layer.attention_layer.state_dict().items()})
    head.red.load_state_dict({k: v.cpu() for k, v in model.fusion_layer.reduce.state_dict().items()})
    head.output_fc.load_state_dict({k: v.cpu() for k, v in model.output_fc.state_dict().items()})
    head.train()

    def cpu_forward():
        t = O.text_tower(tp, text, mask, ocfg.text)[:, 0]
        v = O.image_tower(tp, image, ocfg.image)[:, 0]
        c = O.text_tower(cp, cap, cmask, ccfg)[:, 0]
        cat = torch.cat((head.text_fc(t), head.image_fine_tune(v), head.caption_text_fc(c)), dim=1)
        return head.output_fc(head.red(head.att(cat) * cat)).squeeze(1)

    ref_out = cpu_forward()
    ref_loss = O.sigmoid_focal_loss(ref_out, labels, 0.25, 2.0)
    ref_loss.backward()
    lr = 1e-4
    enc_names = [k for k in tp if not k.split(".")[0].endswith("_fc")]
    g_head = list(head.text_fc.parameters()) + list(head.caption_text_fc.parameters()) + list(head.att.parameters()) + \
        list(head.red.parameters()) + list(head.output_fc.parameters())
    g_enc = [tp[k] for k in enc_names] + list(cp.values()) + list(head.image_fine_tune.parameters())
    everything = g_head + g_enc
    ref_norm = torch.nn.utils.clip_grad_norm_(everything, 1.0)
    topt = torch.optim.Adam([{"params": g_head, "lr": lr}, {"params": g_enc, "lr": lr * 0.8}])
    topt.step()

    # ---- the HIP path
    opt = pkg.Adam(model.get_params(lr), lr=lr, max_grad_norm=1.0)
    crit = pkg.SigmoidFocalLoss(alpha=0.25, gamma=2.0)
    opt.zero_grad()
    out = model(text.cuda(), image.cuda(), mask.cuda(), cap.cuda(), cmask.cuda())
    assert out.shape == (B,)
    loss = crit(out, labels.cuda(), alpha=0.25, gamma=2.0, reduction="mean")
    loss.backward()
    # (three BatchNorm layers over a batch of 8 divide by batch standard deviations of ~0.05-0.3: the fp16 towers' 1e-4
    #  feature error is amplified accordingly)
    close(out, ref_out.detach(), 1e-2, what="logits")
    assert abs(float(loss) - float(ref_loss)) < 2e-3
    norm = float(opt.grad_norm())
    assert abs(norm - float(ref_norm)) < 0.03 * float(ref_norm) + 1e-6, (norm, float(ref_norm))
    opt.step()
    torch.cuda.synchronize()
    # parameters after the clipped step: encoders moved by 0.8 lr-sized steps, head by lr-sized steps, both like torch's
    for k in enc_names:
        if ".key.bias" in k:
            continue
        assert float((model.towers.state_dict()[k].float().cpu() - tp[k].detach()).abs().max()) < 2.05 * lr, k
    got_head = dict(model.text_fc.state_dict())
    for k, v in head.text_fc.state_dict().items():
        if "num_batches" not in k:
            assert float((got_head[k].float().cpu() - v).abs().max()) < 2.05 * lr, k
    d_enc = float((model.towers.state_dict()["bert.encoder.layer.0.output.dense.weight"].float().cpu()
                   - tp["bert.encoder.layer.0.output.dense.weight"].detach()).abs().mean())
    assert d_enc < 0.3 * lr          # agreement far below the step size on average


# ---------------------------------------------------------------------------------------------------
# HF-Trainer protocol (BASELINE config 1: DistilBERT text-only, seq 64, batch 8, dev JSON texts)
# ---------------------------------------------------------------------------------------------------
def test_text_classifier_trainer_protocol_on_dev_texts(pkg, golden_dir):
    df = pkg.read_data(os.path.join(golden_dir, "dev_sample12.json"))
    tok = pkg.HashTokenizer(2000)
    enc = [tok.encode_plus(t, add_special_tokens=True, max_length=64, padding="max_length", truncation=True) for t in list(df["text"])[:8]]
    batch = {"input_ids": torch.cat([e["input_ids"] for e in enc]).cuda(),
             "attention_mask": torch.cat([e["attention_mask"] for e in enc]).cuda(),
             "labels": torch.tensor([pkg.l2id[l] for l in list(df["label"])[:8]]).cuda()}
    tc = pkg.TextConfig(vocab_size=2000, hidden=128, layers=2, heads=2, intermediate=256, max_position=64, type_vocab=0)   # DistilBERT-shaped
    for kind in ("attention", "cls", "mean", "max", "cnn"):
        model = pkg.TextClassifier(tc, pooling_type=kind, num_classes=2, attention_hidden_size=64, compute_dtype="fp16").cuda()
        model.train()
        loss, logits = model(**batch)                                   # labels given -> (loss, logits)
        assert loss.dim() == 0 and logits.shape == (8, 2)
        only = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"])
        assert only.shape == (8, 2)                                     # no labels -> logits
        opt = torch.optim.AdamW(model.parameters(), lr=5e-4)            # HF Trainer's optimizer: plain torch AdamW
        losses = []
        for _ in range(6):
            opt.zero_grad()
            loss, _ = model(**batch)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)     # Trainer's max_grad_norm = 1.0
            opt.step()
            losses.append(float(loss))
        assert losses[-1] < losses[0], (kind, losses)


def test_distilbert_known_parameter_count(pkg):
    tc = pkg.TextConfig(vocab_size=119547, hidden=768, layers=6, heads=12, intermediate=3072, max_position=512, type_vocab=0)
    model = pkg.TextClassifier(tc, pooling_type="cls", num_classes=2, head="distilbert")
    assert model.n_parameters() == 135_326_210          # DistilBERT_example_task2A.ipynb:4301


@pytest.mark.parametrize("size", ["small", "distilbert-base-multilingual-cased"])
def test_hf_trainer_runs_the_text_classifier(pkg, golden_dir, tmp_path, size):
    """BASELINE config 1 through transformers.Trainer itself (the reference's caller, DistilBERT_example_task2A.py:324-334):
    seq 64, batch 8, the dev JSON sample -- with a small DistilBERT-shaped encoder, and at the true shape of the reference's
    checkpoint (6 x 768, vocabulary 119 547, no token-type table: the 135 326 210-parameter model of the notebook), random-init,
    `head="distilbert"` (pre_classifier + classifier, as DistilBertForSequenceClassification)."""
    transformers = pytest.importorskip("transformers")
    from transformers import Trainer, TrainingArguments, default_data_collator
    df = pkg.read_data(os.path.join(golden_dir, "dev_sample12.json"))
    tok = pkg.HashTokenizer(2000)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(df["text"])

        def __getitem__(self, i):
            e = tok.encode_plus(df["text"].iloc[i], add_special_tokens=True, max_length=64, padding="max_length", truncation=True)
            return {"input_ids": e["input_ids"][0], "attention_mask": e["attention_mask"][0], "labels": int(pkg.l2id[df["label"].iloc[i]])}

    if size == "small":
        tc = pkg.TextConfig(vocab_size=2000, hidden=128, layers=2, heads=2, intermediate=256, max_position=64, type_vocab=0)
        model = pkg.TextClassifier(tc, pooling_type="attention", num_classes=2, attention_hidden_size=64, compute_dtype="fp16")
    else:
        tc = pkg.TextConfig(vocab_size=119547, hidden=768, layers=6, heads=12, intermediate=3072, max_position=512, type_vocab=0)
        model = pkg.TextClassifier(tc, pooling_type="cls", num_classes=2, head="distilbert", compute_dtype="fp16")
        assert model.n_parameters() == 135_326_210
    args = TrainingArguments(output_dir=str(tmp_path), learning_rate=2e-4 if size == "small" else 5e-5, num_train_epochs=3, per_device_train_batch_size=8,
                             per_device_eval_batch_size=8, save_strategy="no", report_to=[], logging_steps=1, seed=42,
                             remove_unused_columns=False)
    trainer = Trainer(model=model, args=args, train_dataset=DS(), eval_dataset=DS(), data_collator=default_data_collator)
    res = trainer.train()
    assert np.isfinite(res.training_loss)
    pred = trainer.predict(DS())
    assert pred.predictions.shape == (len(df["text"]), 2)
    logs = [h["loss"] for h in trainer.state.log_history if "loss" in h]
    assert len(logs) >= 2 and logs[-1] < logs[0] + 0.05
