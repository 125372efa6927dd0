"""Round-2 parity and protocol checks on the MI355X (through the C ABI):

* the benchmarked configuration itself -- config 3 at batch 32 -- against the committed fixture
  (tests/golden/config3_b32.npz, transformers BertModel / ViTModel + nn.CrossEntropyLoss on the CPU), in BOTH storage
  types, with the tolerance each one is held to written here;
* torch.optim.* on the model's parameters (the reference's own ``optim.Adam(model.parameters())``) keeps the 16-bit
  GEMM operands in sync without any hook;
* a backward whose activations were overwritten by a later forward raises instead of pairing the wrong tensors;
* optimizer checkpoint / resume.
"""
import os

import numpy as np
import pytest
import torch
from conftest import parity_log

pytestmark = pytest.mark.gpu

# north_star: logits within 1e-3 of the reference CPU path.  fp16 storage (11-bit significand) is held to exactly that.
# bf16 storage (8-bit significand, unit roundoff 2^-9 on every GEMM operand) is held to the bound it meets at 12 layers
# and batch 32; see DESIGN.md section 2 for the measurements behind the number.
LOGIT_TOL = {"fp16": 1e-3, "bf16": 8e-3}
GRAD_REL_TOL = {"fp16": 2e-2, "bf16": 6e-2}        # per-tensor ||hip - ref|| / ||ref|| from the stored norms is not available
                                                   # (norm-of-difference needs the tensors); norms and samples are compared


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import multimodal_propaganda_meme_classification_amd as m
    return m


def _oracle():
    from oracle import meme_oracle as O
    return O


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_config3_batch32_matches_the_fixture(pkg, golden_dir, dtype):
    O = _oracle()
    from oracle.gen_golden import sample_index
    z = np.load(os.path.join(golden_dir, "config3_b32.npz"))
    cfg = O.config3("cls")
    seed, B, S = int(z["seed"]), int(z["batch"]), int(z["seq"])
    params = O.init_params(cfg, seed)
    text, image, mask, labels = O.synthetic_batch(cfg, B, S, seed=1234 + seed)
    chk = np.array([float(image.double().sum()), float(image.double().abs().sum()), float(text.sum()), float(mask.sum()),
                    float(labels.sum())])
    np.testing.assert_allclose(chk, z["input_checksum"], rtol=1e-9)          # the regenerated batch IS the fixture's batch
    d = cfg.to_dict()
    d["compute_dtype"] = dtype
    model = pkg.MultimodalClassifier.from_config(pkg.ModelConfig.from_dict(d), init=False)
    model.load_state_dict(params)
    model.to("cuda").train()
    crit = pkg.CrossEntropyLoss()
    logits = model(text.cuda(), image.cuda(), mask.cuda())
    loss = crit(logits, labels.cuda())
    loss.backward()
    torch.cuda.synchronize()
    err = float(np.abs(logits.detach().float().cpu().numpy() - z["logits"]).max())
    parity_log(f"[{dtype}] config 3, batch 32: max |logit - fixture| = {err:.3e}  (tolerance {LOGIT_TOL[dtype]:.0e})")
    assert err < LOGIT_TOL[dtype]
    assert abs(float(loss.detach()) - float(z["loss"])) < LOGIT_TOL[dtype]
    # gradients: every tensor's norm and the stored samples of it
    names = [str(n) for n in z["grad_names"]]
    got = dict(model.named_parameters())
    worst = ("", 0.0)
    for name, ref_norm, ref_s in zip(names, z["grad_norms"], z["grad_samples"]):
        g = got[name].grad.detach().float().cpu()
        if ".key.bias" in name:           # analytically zero
            assert float(g.norm()) < 1e-4
            continue
        n = float(g.double().norm())
        rel = abs(n - ref_norm) / (ref_norm + 1e-12)
        if rel > worst[1]:
            worst = (name, rel)
        assert rel < GRAD_REL_TOL[dtype], f"{name}: ||g|| {n:.4e} vs {ref_norm:.4e}"
        f = g.reshape(-1)
        idx = sample_index(f.numel())
        s_err = float((f[idx] - torch.from_numpy(ref_s)).abs().max())
        assert s_err <= GRAD_REL_TOL[dtype] * 4 * ref_norm / np.sqrt(f.numel()) + GRAD_REL_TOL[dtype] * float(np.abs(ref_s).max()) + 1e-7, \
            (name, s_err)
    parity_log(f"[{dtype}] worst gradient-norm deviation {worst[1]:.3%} ({worst[0]})")


def test_config3_batch32_full_gradient_tensors_and_signal_relative_error(pkg, golden_dir):
    """The benchmarked configuration, one step at batch 32 (fp16 build), against the oracle RUN HERE on the host cores (the oracle is
    pinned to the reference-run and transformers fixtures by the CPU suite): (a) every weight-gradient MATRIX compared element for
    element -- ||hip - ref||_F / ||ref||_F per tensor, all 148 GEMM weights + the embedding tables, not norms and samples; (b) the
    pooled tower features and the logits relative to their SPREAD OVER THE BATCH, i.e. relative to the signal the classifier uses
    (random-init towers give nearly input-independent outputs: the batch spread of the logits is ~0.06)."""
    O = _oracle()
    z = np.load(os.path.join(golden_dir, "config3_b32.npz"))
    cfg = O.config3("cls")
    seed, B, S = int(z["seed"]), int(z["batch"]), int(z["seq"])
    params = O.init_params(cfg, seed)
    text, image, mask, labels = O.synthetic_batch(cfg, B, S, seed=1234 + seed)
    d = cfg.to_dict()
    d["compute_dtype"] = "fp16"
    model = pkg.MultimodalClassifier.from_config(pkg.ModelConfig.from_dict(d), init=False)
    model.load_state_dict(params)
    model.to("cuda").train()
    logits = model(text.cuda(), image.cuda(), mask.cuda())
    pkg.CrossEntropyLoss()(logits, labels.cuda()).backward()
    with torch.no_grad():
        t_hip, v_hip = model.encode(text.cuda(), image.cuda(), mask.cuda())
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    t_ref = O.text_tower(leaves, text, mask, cfg.text)[:, 0]
    v_ref = O.image_tower(leaves, image, cfg.image)[:, 0]
    ref_logits = O.forward(leaves, text, image, mask, cfg)
    O.cross_entropy(ref_logits, labels).backward()
    np.testing.assert_allclose(ref_logits.detach().numpy(), z["logits"], atol=3e-5)       # the oracle reproduces the transformers fixture
    # (b) errors relative to the batch spread
    for nm, got, ref in (("text cls features", t_hip, t_ref), ("image cls features", v_hip, v_ref), ("logits", logits, ref_logits)):
        got, ref = got.detach().float().cpu(), ref.detach()
        spread = float(ref.std(dim=0).mean())            # per-feature standard deviation over the 32 memes, averaged
        err = float((got - ref).abs().max())
        rms = float((got - ref).pow(2).mean().sqrt())
        parity_log(f"[fp16, batch 32] {nm}: max error {err:.3e}, rms error {rms:.3e}, batch spread {spread:.3e} -> rms error / spread = {rms / spread:.3e}")
        assert rms < 0.02 * spread, nm
    # (a) full gradient tensors
    got = dict(model.named_parameters())
    worst, n_mat = ("", 0.0), 0
    for name, leaf in leaves.items():
        if leaf.dim() < 2 or ".key.bias" in name:
            continue
        g, r = got[name].grad.detach().float().cpu(), leaf.grad
        rel = float((g - r).norm() / (r.norm() + 1e-20))
        n_mat += 1
        if rel > worst[1]:
            worst = (name, rel)
        assert rel < 1e-2, f"{name}: ||hip - ref|| / ||ref|| = {rel:.3e}"
    parity_log(f"[fp16, batch 32] {n_mat} gradient matrices compared element for element; worst ||hip - ref||_F / ||ref||_F = {worst[1]:.3e} ({worst[0]})")
    assert n_mat >= 148


def _tiny(pkg, O, dtype="bf16", seed=11):
    cfg = O.tiny_config("cls")
    d = cfg.to_dict()
    d["compute_dtype"] = dtype
    params = O.init_params(cfg, seed)
    model = pkg.MultimodalClassifier.from_config(pkg.ModelConfig.from_dict(d), init=False)
    model.load_state_dict(params)
    model.to("cuda").train()
    return cfg, params, model


def test_torch_optimizers_keep_the_gemm_operands_in_sync(pkg):
    """The reference's own ``optim.Adam(model.parameters(), lr=2e-5)`` (Multimodal_example_task2C.txt:249) on the HIP
    module: three steps must land where the fused memehip.Adam lands, i.e. the 16-bit shadow the GEMMs read follows the
    fp32 master weights although nobody calls mark_weights_changed()."""
    O = _oracle()
    cfg, params, m_torch = _tiny(pkg, O)
    _, _, m_fused = _tiny(pkg, O)
    text, image, mask, labels = (t.cuda() for t in O.synthetic_batch(cfg, 4, 16, seed=5))
    crit = pkg.CrossEntropyLoss()
    o_torch = torch.optim.Adam(m_torch.parameters(), lr=1e-3)          # large steps: a stale shadow would show at once
    o_fused = pkg.Adam(m_fused.parameters(), lr=1e-3)
    for step in range(3):
        outs = []
        for model, opt in ((m_torch, o_torch), (m_fused, o_fused)):
            opt.zero_grad()
            out = model(text, image, mask)
            crit(out, labels).backward()
            opt.step()
            outs.append(out.detach().float())
        assert float((outs[0] - outs[1]).abs().max()) < 2e-3, f"step {step}: logits diverge -> the shadow went stale"
    # logits moved (lr 1e-3 over 3 steps), and both paths moved together
    with torch.no_grad():
        m_torch.eval(), m_fused.eval()
        a, b = m_torch(text, image, mask), m_fused(text, image, mask)
    assert float((a - b).abs().max()) < 2e-3
    p_t, p_f = m_torch.flat_params, m_fused.flat_params
    assert float((p_t - p_f).abs().max()) < 2.05 * 3 * 1e-3
    assert float((p_t - p_f).abs().mean()) < 2e-5
    # any in-place torch op on any parameter is seen (one shared version counter)
    w = dict(m_torch.named_parameters())["bert.encoder.layer.0.intermediate.dense.weight"]
    assert not m_torch.weights_changed()
    with torch.no_grad():
        w.mul_(0.5)
    assert m_torch.weights_changed()


def test_backward_after_a_second_forward_raises(pkg):
    O = _oracle()
    cfg, _, model = _tiny(pkg, O)
    text, image, mask, labels = (t.cuda() for t in O.synthetic_batch(cfg, 4, 16, seed=6))
    crit = pkg.CrossEntropyLoss()
    out1 = model(text, image, mask)
    out2 = model(text, image, mask)               # same shape: replaces the activations out1's backward would need
    with pytest.raises(RuntimeError, match="overwritten"):
        crit(out1, labels).backward()
    crit(out2, labels).backward()                 # the latest forward is fine


def test_optimizer_checkpoint_and_resume(pkg):
    O = _oracle()
    cfg, _, model = _tiny(pkg, O)
    batch = [t.cuda() for t in O.synthetic_batch(cfg, 4, 16, seed=7)]
    crit = pkg.CrossEntropyLoss()

    def step(m, o):
        o.zero_grad()
        crit(m(*batch[:3]), batch[3]).backward()
        o.step()

    opt = pkg.Adam(model.parameters(), lr=1e-4)
    step(model, opt), step(model, opt)
    sd_m = {k: v.clone() for k, v in model.state_dict().items()}
    sd_o = opt.state_dict()
    assert sd_o["exp_avg"].data_ptr() != opt._flat["M"].data_ptr()           # copies, not live aliases
    step(model, opt)
    want = model.flat_params.clone()
    # resume in fresh objects
    _, _, model2 = _tiny(pkg, O)
    model2.load_state_dict(sd_m)
    opt2 = pkg.Adam(model2.parameters(), lr=1e-4)
    crit(model2(*batch[:3]), batch[3]).backward()      # binds the gradient views
    opt2.load_state_dict(sd_o)
    step(model2, opt2)
    assert opt2._step == 3
    assert torch.equal(model2.flat_params, want), float((model2.flat_params - want).abs().max())
