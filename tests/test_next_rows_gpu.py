"""SURVEY section 8(f) rank 1 ("Kevin's training-step extras"), the parts built so far: sigmoid focal loss on a
1-logit head, parameter-group learning rates (0.8x encoders), linear warm-up schedule, gradient clipping.
GPU box only."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
LR = 2e-5


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import multimodal_propaganda_meme_classification_amd as m
    return m


def test_sigmoid_focal_loss_kernel(pkg):
    from oracle import meme_oracle as O
    g = torch.Generator().manual_seed(0)
    for B in (1, 8, 37, 200):
        x = torch.randn(B, generator=g) * 3
        t = (torch.rand(B, generator=g) < 0.3).float()
        for alpha, gamma in ((0.25, 2.0), (-1.0, 0.0), (0.5, 1.5)):
            xr = x.clone().requires_grad_(True)
            ref = O.sigmoid_focal_loss(xr, t, alpha, gamma)
            ref.backward()
            crit = pkg.SigmoidFocalLoss()
            xd = x.cuda().requires_grad_(True)
            loss = crit(xd, t.cuda(), alpha=alpha, gamma=gamma, reduction="mean")
            loss.backward()
            np.testing.assert_allclose(float(loss), float(ref), rtol=2e-5, atol=1e-7)
            np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=2e-4, atol=1e-7)
            assert int(crit.last_correct) == int(((x > 0) == (t > 0.5)).sum())
    # and against the fixture made with transformers' implementation of the same loss (tests/golden/focal_hf.npz)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "focal_hf.npz"))
    for i in range(3):
        xd = torch.from_numpy(z[f"x{i}"]).cuda().requires_grad_(True)
        loss = pkg.SigmoidFocalLoss()(xd, torch.from_numpy(z[f"t{i}"]).cuda(), alpha=float(z[f"alpha{i}"]), gamma=float(z[f"gamma{i}"]),
                                      reduction="mean")
        loss.backward()
        np.testing.assert_allclose(float(loss), float(z[f"loss{i}"]), rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(xd.grad.cpu().numpy(), z[f"dx{i}"], rtol=2e-4, atol=1e-7)


def test_kevin_style_step_focal_param_groups_warmup_clip(pkg):
    """One-logit head + focal loss + Adam over get_params(lr) (encoders at 0.8 lr) + linear warm-up + clip 1.0,
    three steps against the oracle (fp16 build: logits within 1e-3)."""
    from oracle import meme_oracle as O
    cfg = O.tiny_config("cls")
    cfg.num_classes = 1
    params = O.init_params(cfg, 31)
    d = cfg.to_dict()
    d["compute_dtype"] = "fp16"
    model = pkg.MultimodalClassifier.from_config(pkg.ModelConfig.from_dict(d), init=False)
    model.load_state_dict(params)
    model.to("cuda")
    text, image, mask, labels = O.synthetic_batch(cfg, 8, 16, seed=9)
    tgt = labels.float()
    lr0 = 1e-3
    opt = pkg.Adam(model.get_params(lr0), max_grad_norm=1.0)
    assert [g["lr"] for g in opt.param_groups] == [lr0, 0.8 * lr0, 0.8 * lr0]
    sched = pkg.get_linear_schedule_with_warmup(opt, num_warmup_steps=2, num_training_steps=10)
    crit = pkg.SigmoidFocalLoss()
    st = O.AdamState()
    p_ref = params
    model.train()
    for step in range(3):
        mult = sched.get_last_lr()[0] / lr0
        opt.zero_grad()
        out = model(text.cuda(), image.cuda(), mask.cuda())
        assert out.shape == (8, 1)
        loss = crit(out.squeeze(1), tgt.cuda(), alpha=0.25, gamma=2.0, reduction="mean")
        loss.backward()
        # oracle
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p_ref.items()}
        ref_out = O.forward(leaves, text, image, mask, cfg).squeeze(1)
        ref_loss = O.sigmoid_focal_loss(ref_out, tgt, 0.25, 2.0)
        ref_loss.backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
        assert float((out.detach().float().cpu().squeeze(1) - ref_out.detach()).abs().max()) <= 1e-3
        assert abs(float(loss.detach()) - float(ref_loss)) <= 1e-3
        lr_now = lr0 * mult
        p_ref = O.adam_step(p_ref, grads, st, lr=lr_now, max_grad_norm=1.0,
                            lr_of=lambda k: lr_now * (0.8 if k.startswith(("bert.", "image_model.")) else 1.0))
        opt.step()
        sched.step()
        sd = model.state_dict()
        for k, ref in p_ref.items():
            dmax = float((sd[k].detach().float().cpu() - ref).abs().max())
            assert dmax <= 2.05 * lr0 * (step + 1), (k, dmax)
        if step == 0:
            assert mult == 0.0          # warm-up starts at 0: the first update must not move any parameter
            for k, v in params.items():
                assert torch.equal(sd[k].cpu(), v), k
    # mean |diff| far below the step size
    tot = sum(float((sd[k].float().cpu() - p_ref[k]).abs().sum()) for k in p_ref if ".key.bias" not in k)
    cnt = sum(p_ref[k].numel() for k in p_ref if ".key.bias" not in k)
    assert tot / cnt < 0.05 * lr0


def test_linear_schedule_matches_transformers(pkg):
    import transformers
    w = torch.nn.Parameter(torch.zeros(3))
    o1, o2 = torch.optim.SGD([w], lr=0.5), torch.optim.SGD([w], lr=0.5)
    s1 = pkg.get_linear_schedule_with_warmup(o1, 3, 11)
    s2 = transformers.get_linear_schedule_with_warmup(o2, 3, 11)
    for _ in range(13):
        assert s1.get_last_lr() == s2.get_last_lr()
        o1.step(); o2.step(); s1.step(); s2.step()


# ---------------------------------------------------------------------------------------------------------------
# heads built outside the library on the towers' pooled features
# ---------------------------------------------------------------------------------------------------------------
class _KevinHead(torch.nn.Module):
    """The head of Multimodal_example_task2C.py:590-685 for two towers: Dropout-free text projection
    Linear+BatchNorm1d+ReLU (:603-605), the image fine-tune MLP (:573-576), ConcatAttention3's gate and
    reduce stages (:476-499, two inputs), and the 1-logit Linear(512,1)+BatchNorm1d(1) (:641-643)."""

    def __init__(self, dt, di, p):
        super().__init__()
        nn = torch.nn
        self.text_fc = nn.Sequential(nn.Linear(dt, p), nn.BatchNorm1d(p), nn.ReLU())
        self.image_ft = nn.Sequential(nn.Linear(di, p), nn.ReLU(), nn.Linear(p, p))
        self.attention_layer = nn.Sequential(nn.Linear(2 * p, 2 * p), nn.BatchNorm1d(2 * p), nn.ReLU(), nn.Softmax(dim=1))
        self.reduce = nn.Sequential(nn.Linear(2 * p, p), nn.BatchNorm1d(p), nn.ReLU())
        self.output_fc = nn.Sequential(nn.Linear(p, 1), nn.BatchNorm1d(1))

    def forward(self, t, v):
        cat = torch.cat((self.text_fc(t), self.image_ft(v)), dim=1)
        fused = self.reduce(self.attention_layer(cat) * cat)
        return self.output_fc(fused).squeeze(1)


class _MlpHead(torch.nn.Module):
    """A well-conditioned external head (no batch statistics) for the strict gradient comparison."""

    def __init__(self, dt, di, p):
        super().__init__()
        nn = torch.nn
        self.t, self.v = nn.Linear(dt, p), nn.Linear(di, p)
        self.o = nn.Sequential(nn.ReLU(), nn.Linear(2 * p, 1))

    def forward(self, t, v):
        return self.o(torch.cat((self.t(t), self.v(v)), dim=1)).squeeze(1)


@pytest.mark.parametrize("pool", ["cls", "last"])
@pytest.mark.parametrize("head_kind", ["mlp", "kevin_bn"])
def test_encode_with_external_head_matches_oracle(pool, head_kind):
    """model.encode() exposes the pooled tower features with autograd; a torch head + the sigmoid focal loss on top
    of it (on the GPU) must give the oracle's loss and tower gradients (same head in fp32 on the CPU).  Kevin's
    BatchNorm head divides by the between-sample spread of the features (tiny for a random-init model), so its
    gradients are only checked loosely; the MLP head is the strict comparison."""
    import copy
    import multimodal_propaganda_meme_classification_amd as pkg
    from oracle import meme_oracle as O
    cfg = O.tiny_config(pool)
    params = O.init_params(cfg, 17)
    mc = pkg.ModelConfig.from_dict(cfg.to_dict())
    # strict comparison on the 11-bit-significand build; the BatchNorm head's 1/std amplification would overflow
    # fp16's static gradient-stream scale on a random-init model, so it runs on the bf16 build
    mc.compute_dtype = "fp16" if head_kind == "mlp" else "bf16"
    model = pkg.MultimodalClassifier.from_config(mc, init=False)
    model.load_state_dict(params)
    model.to("cuda").train()
    text, image, mask, labels = O.synthetic_batch(cfg, 16, 16, seed=3)
    torch.manual_seed(0)
    Head = _MlpHead if head_kind == "mlp" else _KevinHead
    head_cpu = Head(cfg.text.hidden, cfg.image.hidden, 64)
    head_gpu = copy.deepcopy(head_cpu).cuda()
    crit = pkg.SigmoidFocalLoss(alpha=0.25, gamma=2.0)

    # HIP towers + torch head on the GPU
    t_feat, i_feat = model.encode(text.cuda(), image.cuda(), mask.cuda())
    out = head_gpu(t_feat, i_feat)
    loss = crit(out, labels.cuda().float())
    loss.backward()

    # oracle towers + the same head on the CPU
    p_ref = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    th = O.pool_text(O.text_tower(p_ref, text, mask, cfg.text), cfg.pool)
    ih = O.image_tower(p_ref, image, cfg.image)[:, 0]
    out_ref = head_cpu(th, ih)
    loss_ref = O.sigmoid_focal_loss(out_ref, labels.float(), alpha=0.25, gamma=2.0)
    loss_ref.backward()

    strict = head_kind == "mlp"
    ftol = 4e-3 if strict else 3e-2
    assert float((t_feat.detach().cpu() - th.detach()).abs().max()) <= ftol * float(th.abs().max())
    assert float((i_feat.detach().cpu() - ih.detach()).abs().max()) <= ftol * float(ih.abs().max())
    assert abs(float(loss) - float(loss_ref)) <= (1e-2 if strict else 1e-1) * max(1.0, abs(float(loss_ref)))
    head_tol, tower_tol = (0.02, 0.03) if strict else (None, None)    # BatchNorm head: gradients finite, not compared
    gmax = max(float(b.grad.norm()) for b in head_cpu.parameters())
    for (n, a), (_, b) in zip(head_gpu.named_parameters(), head_cpu.named_parameters()):
        assert torch.isfinite(a.grad).all()
        if head_tol is None or float(b.grad.norm()) < 1e-3 * gmax:
            continue
        assert float((a.grad.cpu() - b.grad).norm()) <= head_tol * float(b.grad.norm()) + 1e-6, n
    num = den = 0.0
    for name, p in model.named_parameters():
        ref = p_ref[name].grad
        if name.split(".")[0] in ("bert_fc", "image_fc", "fusion_fc", "output_fc"):
            assert float(p.grad.abs().max()) == 0.0, name          # the built-in head is not on this path
            continue
        if ref is None or ".key.bias" in name:
            continue
        assert torch.isfinite(p.grad).all(), name
        num += float((p.grad.detach().float().cpu() - ref).pow(2).sum())
        den += float(ref.pow(2).sum())
    if tower_tol is not None:
        assert (num / den) ** 0.5 <= tower_tol, (num / den) ** 0.5

    g = torch.Generator().manual_seed(1)
    pw, pb = torch.randn((cfg.text.hidden, cfg.text.hidden), generator=g) * 0.05, torch.randn(cfg.text.hidden, generator=g) * 0.05
    feats = model.get_features(text.cuda(), image.cuda(), mask.cuda(), pooler=(pw, pb))
    assert feats["text"].shape == (16, cfg.text.hidden) and feats["image"].shape == (16, cfg.image.hidden)
    if strict:      # BertModel.pooler_output = tanh(dense(h[:, 0])) on the pooled token
        want = torch.tanh(th.detach() @ pw.t() + pb)
        assert float((feats["pooler_output"].cpu() - want).abs().max()) <= 5e-3
    assert model.training


def test_device_side_totensor_normalize_is_bit_exact(tmp_path):
    """ToTensor + Normalize on the device from the resized / cropped uint8 pixels must give the
    host transform's float32 values bit for bit (Multimodal_example_task2C.txt:37-41), also through the Dataset."""
    import multimodal_propaganda_meme_classification_amd as pkg
    from multimodal_propaganda_meme_classification_amd import data as D, ops
    g = torch.Generator().manual_seed(4)
    u8 = torch.randint(0, 256, (5, 224, 224, 3), generator=g, dtype=torch.uint8)
    u8[0] = 0
    u8[1] = 255
    mean = torch.tensor(D.IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(D.IMAGENET_STD).view(1, 3, 1, 1)
    ref = (torch.from_numpy(u8.numpy().astype(np.float32) / 255.0).permute(0, 3, 1, 2) - mean) / std
    got = pkg.normalize_images(u8.cuda())
    assert got.dtype == torch.float32 and tuple(got.shape) == (5, 3, 224, 224)
    assert torch.equal(got.cpu(), ref)
    assert pkg.normalize_images(ref.cuda()).data_ptr() != 0          # float batches pass through
    with pytest.raises(pkg.MemehipError):
        ops.image_normalize_u8(u8, D.IMAGENET_MEAN, D.IMAGENET_STD)  # CPU tensor: no fallback
    # through the Dataset, from an image file
    from PIL import Image
    arr = torch.randint(0, 256, (300, 400, 3), generator=g, dtype=torch.uint8).numpy()
    Image.fromarray(arr).save(tmp_path / "m.png")
    host = D.load_image(str(tmp_path / "m.png"))
    ds = pkg.MultimodalDataset(["a"], ["some text"], ["m.png"], [0], image_root=str(tmp_path), device_normalize=True)
    item = ds[0]
    assert item["image"].dtype == torch.uint8 and tuple(item["image"].shape) == (224, 224, 3)
    dev = pkg.normalize_images(item["image"][None].cuda())[0].cpu()
    assert torch.equal(dev, host)


class _PoolingZooHead(torch.nn.Module):
    """Every pooling of LLMWithClassificationHead (Multimodal_example_task2C.py:356-392) at once -- max, masked mean,
    tanh-attention, conv1d-max over the text states; mean over the ViT patch tokens -- into one logit."""

    def __init__(self, dt, di, p, with_max=True):
        super().__init__()
        nn = torch.nn
        self.with_max = with_max
        self.attention = nn.Sequential(nn.Linear(dt, p), nn.Tanh(), nn.Linear(p, 1))
        self.conv1d = nn.Conv1d(dt, dt, kernel_size=3, padding=1)
        self.out = nn.Linear(4 * dt + di, 1)

    def forward(self, h, img, mask):
        m = mask.unsqueeze(-1).float()
        mx = torch.max(h, dim=1)[0]                                           # :363 (padded positions included)
        mean = (h * m).sum(1) / torch.clamp(m.expand_as(h).sum(1), min=1e-9)  # :366-377
        sc = self.attention(h).squeeze(-1) + (1.0 - mask.float()) * -1e9      # :379-387
        att = (h * torch.softmax(sc, dim=1).unsqueeze(-1)).sum(1)
        cnn = torch.max(torch.relu(self.conv1d(h.permute(0, 2, 1))), dim=-1)[0]   # :389-393
        if not self.with_max:      # the smooth poolings only (strict gradient comparison)
            mx, cnn = torch.zeros_like(mx), torch.zeros_like(cnn)
        return self.out(torch.cat((mx, mean, att, cnn, img[:, 1:].mean(1)), dim=1)).squeeze(1)


@pytest.mark.parametrize("with_max", [False, True])
def test_encode_sequence_with_the_pooling_zoo_matches_oracle(with_max):
    """model.encode_sequence() hands every hidden state of both towers to PyTorch with autograd: the reference's
    max / masked-mean / attention / conv1d poolings on top must give the oracle's loss and tower gradients."""
    import copy
    import multimodal_propaganda_meme_classification_amd as pkg
    from oracle import meme_oracle as O
    cfg = O.tiny_config("cls")
    params = O.init_params(cfg, 23)
    mc = pkg.ModelConfig.from_dict(cfg.to_dict())
    mc.compute_dtype = "fp16"
    model = pkg.MultimodalClassifier.from_config(mc, init=False)
    model.load_state_dict(params)
    model.to("cuda").train()
    text, image, mask, labels = O.synthetic_batch(cfg, 6, 16, seed=8)
    torch.manual_seed(1)
    head_cpu = _PoolingZooHead(cfg.text.hidden, cfg.image.hidden, 32, with_max=with_max)
    head_gpu = copy.deepcopy(head_cpu).cuda()
    crit = pkg.SigmoidFocalLoss()

    h, im = model.encode_sequence(text.cuda(), image.cuda(), mask.cuda())
    assert tuple(h.shape) == (6, 16, cfg.text.hidden) and im.shape[0] == 6 and im.shape[2] == cfg.image.hidden
    loss = crit(head_gpu(h, im, mask.cuda()), labels.cuda().float())
    loss.backward()

    p_ref = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    h_ref = O.text_tower(p_ref, text, mask, cfg.text)
    im_ref = O.image_tower(p_ref, image, cfg.image)
    loss_ref = O.sigmoid_focal_loss(head_cpu(h_ref, im_ref, mask), labels.float(), alpha=0.25, gamma=2.0)
    loss_ref.backward()

    assert float((h.detach().cpu() - h_ref.detach()).abs().max()) <= 4e-3 * float(h_ref.abs().max())   # padded rows too
    assert float((im.detach().cpu() - im_ref.detach()).abs().max()) <= 4e-3 * float(im_ref.abs().max())
    assert abs(float(loss) - float(loss_ref)) <= 1e-2
    for (n, a), (_, b) in zip(head_gpu.named_parameters(), head_cpu.named_parameters()):
        # the max over positions routes each gradient to ONE position: a near-tie flips with the towers' rounding,
        # so the parameters in front of a max (the conv) are compared loosely
        if b.grad is None or float(b.grad.norm()) == 0.0:
            continue
        tol = 0.2 if n.startswith("conv1d") else (0.1 if with_max else 0.03)
        assert float((a.grad.cpu() - b.grad).norm()) <= tol * float(b.grad.norm()) + 1e-6, n
    num = den = 0.0
    for name, p in model.named_parameters():
        ref = p_ref[name].grad
        if ref is None or ".key.bias" in name or name.split(".")[0] in ("bert_fc", "image_fc", "fusion_fc", "output_fc"):
            continue
        num += float((p.grad.detach().float().cpu() - ref).pow(2).sum())
        den += float(ref.pow(2).sum())
    assert (num / den) ** 0.5 <= (0.15 if with_max else 0.03), (num / den) ** 0.5
    with torch.no_grad():
        model.eval()
        h2, _ = model.encode_sequence(text.cuda(), image.cuda(), mask.cuda())
        assert float((h2 - h.detach()).abs().max()) == 0.0


@pytest.mark.parametrize("B,Fn,relu", [(32, 512, True), (32, 1, False), (7, 96, True), (200, 33, False)])
def test_batchnorm1d_kernel_matches_torch(B, Fn, relu):
    """memehip.BatchNorm1d (mh_bn1d_fwd / mh_bn1d_bwd, optional fused ReLU) against torch.nn.BatchNorm1d (+ ReLU) in
    fp32 on the CPU: outputs, input / affine gradients, running statistics over several steps, eval mode."""
    import multimodal_propaganda_meme_classification_amd as pkg
    g = torch.Generator().manual_seed(B + Fn)
    ref = torch.nn.BatchNorm1d(Fn)
    with torch.no_grad():
        ref.weight.copy_(1 + 0.2 * torch.randn(Fn, generator=g))
        ref.bias.copy_(0.1 * torch.randn(Fn, generator=g))
    mine = pkg.BatchNorm1d(Fn, relu=relu)
    mine.load_state_dict(ref.state_dict())
    mine.cuda()
    for step in range(3):
        x = torch.randn((B, Fn), generator=g) * 2 + 0.5
        dy = torch.randn((B, Fn), generator=g)
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)
        if relu:
            yr = torch.relu(yr)
        yr.backward(dy)
        xm = x.cuda().requires_grad_(True)
        ym = mine(xm)
        ym.backward(dy.cuda())
        np.testing.assert_allclose(ym.detach().cpu().numpy(), yr.detach().numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(xm.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(mine.weight.grad.cpu().numpy(), ref.weight.grad.numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(mine.bias.grad.cpu().numpy(), ref.bias.grad.numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(mine.running_mean.cpu().numpy(), ref.running_mean.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(mine.running_var.cpu().numpy(), ref.running_var.numpy(), rtol=1e-5, atol=1e-6)
        assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == step + 1
        for m in (mine, ref):
            m.zero_grad()
    ref.eval(), mine.eval()
    x = torch.randn((B, Fn), generator=g)
    ye = ref(x)
    if relu:
        ye = torch.relu(ye)
    np.testing.assert_allclose(mine(x.cuda()).detach().cpu().numpy(), ye.detach().numpy(), rtol=2e-5, atol=2e-5)
    assert sorted(mine.state_dict().keys()) == sorted(ref.state_dict().keys())


@pytest.mark.parametrize("clip", [None, 1.0])
def test_adam_skips_the_step_on_nonfinite_gradients(clip):
    """GradScaler.step of the reference's AMP path (Multimodal_example_task2C.py:712-717) skips the optimizer step
    when a gradient overflowed; the fused Adam does the same when the gradient norm is inf / nan (skip_nonfinite, or
    any clipping): parameters, moments and the 16-bit shadow stay bit-identical, the next clean step proceeds."""
    import multimodal_propaganda_meme_classification_amd as pkg
    from oracle import meme_oracle as O
    cfg = O.tiny_config("cls")
    model = pkg.MultimodalClassifier.from_config(pkg.ModelConfig.from_dict(cfg.to_dict()), init=False)
    model.load_state_dict(O.init_params(cfg, 2))
    model.to("cuda").train()
    opt = pkg.Adam(model.parameters(), lr=1e-3, model=model, max_grad_norm=clip, skip_nonfinite=True)
    text, image, mask, labels = (t.cuda() for t in O.synthetic_batch(cfg, 4, 16, seed=1))
    model.forward_backward(text, image, mask, labels)
    opt.step()                                             # a clean step first (moments become non-zero)
    torch.cuda.synchronize()
    p0, m0, v0 = model.flat_params.clone(), opt._flat["M"].clone(), opt._flat["V"].clone()
    sh0 = model.flat_shadow.clone()
    model.forward_backward(text, image, mask, labels)
    for bad in (float("inf"), float("nan")):
        model.flat_grads[12345] = bad                      # an overflowed gradient element
        opt.step()
        torch.cuda.synchronize()
        assert torch.equal(model.flat_params, p0) and torch.equal(opt._flat["M"], m0) and torch.equal(opt._flat["V"], v0)
        assert torch.equal(model.flat_shadow, sh0)
    model.forward_backward(text, image, mask, labels)      # clean gradients again
    opt.step()
    torch.cuda.synchronize()
    assert not torch.equal(model.flat_params, p0) and bool(torch.isfinite(model.flat_params).all())


def test_skipped_steps_do_not_advance_adams_bias_correction():
    """torch's GradScaler.step does not call optimizer.step() after an overflow, so torch.optim.Adam's step count -- and with it
    the bias corrections 1 - beta^t -- only advances on finite steps.  The fused Adam counts skipped steps on the device
    (mh_adam_skip_account) and recomputes the corrections for the steps actually taken: a gradient sequence [g1, inf, g2, g3]
    must give exactly what torch.optim.Adam gives for [g1, g2, g3]; the skipped step is reported, survives a checkpoint,
    and the same holds when the step is a replayed hipGraph."""
    import multimodal_propaganda_meme_classification_amd as pkg
    torch.manual_seed(0)
    holder = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(64, 32)), torch.nn.Parameter(torch.randn(100))]).cuda()
    pkg.flatten_parameters(holder)
    ref = [p.detach().clone().cpu().requires_grad_(True) for p in holder]
    topt = torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.95))
    opt = pkg.Adam(holder.parameters(), lr=1e-2, betas=(0.9, 0.95), skip_nonfinite=True)
    g = torch.Generator().manual_seed(1)
    grads = [[torch.randn(p.shape, generator=g) for p in ref] for _ in range(3)]

    def mine(gs, poison=False):
        for p, gg in zip(holder, gs):
            p.grad.copy_(gg.cuda())
        if poison:
            holder[1].grad[7] = float("inf")
        opt.step()

    def theirs(gs):
        for p, gg in zip(ref, gs):
            p.grad = gg.clone()
        topt.step()

    opt.zero_grad()
    mine(grads[0]); theirs(grads[0])
    mine(grads[1], poison=True)                 # skipped: no torch step
    assert opt.last_step_skipped and opt.skipped_steps == 1
    sd = opt.state_dict()
    assert sd["step"] == 2 and sd["skipped"] == 1
    mine(grads[1]); theirs(grads[1])
    assert not opt.last_step_skipped and opt.skipped_steps == 1
    opt.load_state_dict(opt.state_dict())       # round trip keeps the effective step
    mine(grads[2]); theirs(grads[2])
    torch.cuda.synchronize()
    for p, r in zip(holder, ref):
        np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().numpy(), rtol=2e-6, atol=2e-7)
    # without the accounting the third real update would use t = 4 instead of 3: |delta| ~ 1e-3 here, far above the tolerance


def test_three_tower_model_with_caption_encoder_matches_oracle():
    """Kevin's forward(text, image, mask, caption_text, caption_text_mask) (Multimodal_example_task2C.py:666-685):
    text + image towers (MultimodalClassifier.encode) and a caption tower (TextEncoder), a three-input fusion head in
    PyTorch on top, focal loss; against the oracle towers with the same head on the CPU."""
    import copy
    import multimodal_propaganda_meme_classification_amd as pkg
    from oracle import meme_oracle as O
    cfg = O.tiny_config("cls")
    params = O.init_params(cfg, 41)
    cap_params = {k[len("bert."):]: v for k, v in O.init_params(cfg, 43).items() if k.startswith("bert.")}
    mc = pkg.ModelConfig.from_dict(cfg.to_dict())
    mc.compute_dtype = "fp16"
    model = pkg.MultimodalClassifier.from_config(mc, init=False)
    model.load_state_dict(params)
    model.to("cuda").train()
    cap = pkg.TextEncoder(mc.text, pool="cls", compute_dtype="fp16")
    cap.load_state_dict(cap_params)
    cap.to("cuda").train()
    assert sorted(cap.state_dict().keys()) == sorted(cap_params.keys())

    B, S = 8, 16
    text, image, mask, labels = O.synthetic_batch(cfg, B, S, seed=5)
    ctext, _, cmask, _ = O.synthetic_batch(cfg, B, S, seed=6)
    D = cfg.text.hidden

    class Fusion(torch.nn.Module):                      # three-input gate + reduce (ConcatAttention3 without BatchNorm)
        def __init__(self):
            super().__init__()
            self.gate = torch.nn.Linear(2 * D + cfg.image.hidden, 2 * D + cfg.image.hidden)
            self.out = torch.nn.Linear(2 * D + cfg.image.hidden, 1)

        def forward(self, t, v, c):
            cat = torch.cat((t, v, c), dim=1)
            return self.out(torch.softmax(torch.relu(self.gate(cat)), dim=1) * cat).squeeze(1)

    torch.manual_seed(3)
    head_cpu = Fusion()
    head_gpu = copy.deepcopy(head_cpu).cuda()
    crit = pkg.SigmoidFocalLoss()
    t_feat, i_feat = model.encode(text.cuda(), image.cuda(), mask.cuda())
    c_feat = cap(ctext.cuda(), cmask.cuda())
    loss = crit(head_gpu(t_feat, i_feat, c_feat), labels.cuda().float())
    loss.backward()

    p_ref = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    c_ref = {"bert." + k: v.clone().requires_grad_(True) for k, v in cap_params.items()}
    th = O.pool_text(O.text_tower(p_ref, text, mask, cfg.text), "cls")
    ih = O.image_tower(p_ref, image, cfg.image)[:, 0]
    ch = O.pool_text(O.text_tower(c_ref, ctext, cmask, cfg.text), "cls")
    loss_ref = O.sigmoid_focal_loss(head_cpu(th, ih, ch), labels.float(), alpha=0.25, gamma=2.0)
    loss_ref.backward()

    assert float((c_feat.detach().cpu() - ch.detach()).abs().max()) <= 4e-3 * float(ch.abs().max())
    assert abs(float(loss) - float(loss_ref)) <= 1e-2
    for mod, ref, prefix in ((model, p_ref, ""), (cap.inner, c_ref, "bert.")):
        num = den = 0.0
        for name, p in mod.named_parameters():
            if not name.startswith("bert.") and mod is cap.inner:
                assert float(p.grad.abs().max()) == 0.0, name          # the stub image tower and the unused head
                continue
            r = ref.get(name)
            if r is None or r.grad is None or ".key.bias" in name or name.split(".")[0] in ("bert_fc", "image_fc", "fusion_fc", "output_fc"):
                continue
            num += float((p.grad.detach().float().cpu() - r.grad).pow(2).sum())
            den += float(r.grad.pow(2).sum())
        assert (num / den) ** 0.5 <= 0.03, (num / den) ** 0.5
    # one fused Adam per tower module leaves the stub untouched and updates the caption BERT
    before = cap.inner.flat_params.clone()
    opt = pkg.Adam(cap.parameters(), lr=1e-3)
    opt.step()
    torch.cuda.synchronize()
    lay = cap.inner.layout
    changed = (cap.inner.flat_params != before)
    w0 = lay.spec["bert.encoder.layer.0.attention.self.query.weight"]
    s0 = lay.spec["image_model.encoder.layer.0.attention.attention.query.weight"]
    assert bool(changed[w0.offset:w0.offset + w0.numel].any()) and not bool(changed[s0.offset:s0.offset + s0.numel].any())


def test_device_image_pipeline_equals_the_pil_path(pkg, tmp_path):
    """SURVEY 8 f rank 4: decoded uint8 images of different sizes -> one pinned async H2D copy -> PIL's antialiased bilinear
    Resize(256) + CenterCrop(224) in its own fixed point + ToTensor + Normalize, all on the device: bit-identical to the host
    transform (data.load_image), i.e. to the reference's torchvision pipeline on PIL images."""
    from PIL import Image, ImageEnhance
    from multimodal_propaganda_meme_classification_amd.data import DeviceImagePipeline, load_image, load_image_u8
    rng = np.random.default_rng(3)
    paths, imgs = [], []
    for i, (h, w) in enumerate(((300, 400), (427, 640), (640, 427), (97, 131), (224, 224), (512, 512))):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = str(tmp_path / f"im{i}.png")
        Image.fromarray(a).save(p)
        paths.append(p)
        imgs.append(Image.open(p))
    pipe = DeviceImagePipeline(224, 256, "center_crop", device="cuda")
    got = pipe(imgs)
    torch.cuda.synchronize()
    want_u8 = torch.stack([load_image_u8(p) for p in paths])
    assert torch.equal(pipe.last_u8.cpu(), want_u8)
    want = torch.stack([load_image(p) for p in paths])
    assert torch.equal(got.cpu(), want)
    # Kevin's Resize((224, 224)) + flip + ColorJitter + rotation, with factors that make each op checkable against PIL
    pipe2 = DeviceImagePipeline(224, mode="stretch", augment=True, device="cuda", generator=torch.Generator().manual_seed(5),
                                jitter=(0.1, 0.1, 0.1, 0.0), degrees=0.0)
    out = pipe2(imgs)
    torch.cuda.synchronize()
    assert out.shape == (6, 3, 224, 224) and bool(torch.isfinite(out).all())
    # replay the same random draws on the host with PIL's own ImageEnhance ops (hue 0, angle 0)
    g = torch.Generator().manual_seed(5)
    flips = (torch.rand(6, generator=g) < 0.5).numpy()
    u = torch.rand((6, 5), generator=g).numpy()
    for b, im in enumerate(imgs):
        perm = torch.randperm(4, generator=g).tolist()
        ref = im.convert("RGB").resize((224, 224), Image.BILINEAR)
        if flips[b]:
            ref = ref.transpose(Image.FLIP_LEFT_RIGHT)
        f = [float(np.float32(0.9 + 0.2 * u[b, k])) for k in range(3)]
        for op in perm:
            if op == 0:
                ref = ImageEnhance.Brightness(ref).enhance(f[0])
            elif op == 1:
                ref = ImageEnhance.Contrast(ref).enhance(f[1])
            elif op == 2:
                ref = ImageEnhance.Color(ref).enhance(f[2])
        d = np.abs(np.asarray(ref).astype(int) - pipe2.last_u8[b].cpu().numpy().astype(int))
        assert d.max() == 0, (b, perm, int(d.max()), float((d > 0).mean()))     # bit-exact since round 3 (the blend's product and sum are rounded separately, as in ImagingBlend)
    # rotation: 90 degrees is an exact permutation of the pixels
    pipe3 = DeviceImagePipeline(224, mode="stretch", augment=True, device="cuda", generator=torch.Generator().manual_seed(7),
                                jitter=(0.0, 0.0, 0.0, 0.0), degrees=0.0)
    base = pipe3(imgs[:2])
    base_u8 = pipe3.last_u8.clone()
    assert bool(torch.isfinite(base).all())
    ref0 = np.asarray(imgs[0].convert("RGB").resize((224, 224), Image.BILINEAR))
    g7 = torch.Generator().manual_seed(7)
    fl = (torch.rand(2, generator=g7) < 0.5).numpy()
    assert np.array_equal(base_u8[0].cpu().numpy(), ref0[:, ::-1] if fl[0] else ref0)          # identity jitter, angle 0


def test_device_jitter_hue_and_rotation_equal_torchvisions_pil_path(pkg, tmp_path):
    """ColorJitter(0.1, 0.1, 0.1, 0.1) + RandomRotation(15) as Kevin's Dataset applies them (Multimodal_example_task2C.py:222-235), with
    NON-ZERO hue shifts and angles: the device pipeline, given the factors / op order / angle of a draw, must reproduce torchvision's
    PIL code path (oracle/ref_env.py: ImageEnhance blends, the HSV hue shift, Image.rotate NEAREST) bit for bit."""
    from PIL import Image
    from oracle import ref_env as E
    from multimodal_propaganda_meme_classification_amd.data import DeviceImagePipeline
    rng = np.random.default_rng(9)
    imgs = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in ((300, 400), (427, 640), (97, 131), (224, 224))]
    imgs.append(Image.fromarray(E.synthetic_meme(3)))
    B = len(imgs)
    params = dict(flip=[1, 0, 1, 0, 1],
                  factors=[[1.07, 0.93, 1.04, 0.0713], [0.91, 1.1, 0.9, -0.0831], [1.0, 1.0, 1.0, 0.1], [1.1, 0.9, 1.1, -0.1], [0.95, 1.05, 0.97, 0.0039]],
                  order=[[0, 1, 2, 3], [3, 2, 1, 0], [1, 3, 0, 2], [2, 0, 3, 1], [3, 1, 0, 2]],
                  angle=[3.7, -14.99, 11.25, -7.0, 0.0])
    pipe = DeviceImagePipeline(224, mode="stretch", augment=True, device="cuda")
    out = pipe(imgs, params=params)
    torch.cuda.synchronize()
    assert out.shape == (B, 3, 224, 224)
    T = E.transforms
    for b, im in enumerate(imgs):
        ref = T.Resize((224, 224))(im.convert("RGB"))
        if params["flip"][b]:
            ref = ref.transpose(Image.FLIP_LEFT_RIGHT)
        f = {k: float(np.float32(params["factors"][b][k])) for k in range(3)}      # the float32 the kernel receives (PIL rounds to C float too)
        f[3] = params["factors"][b][3]
        ref = E.apply_color_jitter(ref, params["order"][b], f)
        ref = E.rotate_pil(ref, params["angle"][b])
        got = pipe.last_u8[b].cpu().numpy()
        d = np.abs(np.asarray(ref).astype(int) - got.astype(int))
        assert d.max() == 0, (b, int(d.max()), float((d > 0).mean()))
    # and the normalised float batch is ToTensor + Normalize of exactly those bytes
    want = torch.stack([T.Normalize((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))(T.ToTensor()(Image.fromarray(pipe.last_u8[b].cpu().numpy()))) for b in range(B)])
    assert float((out.cpu() - want).abs().max()) < 1e-6
