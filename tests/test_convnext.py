"""ConvNeXt-tiny, the image encoder of the SVM baseline's feature dump (baselines/extract_feat.py:52-60,82-85).
CPU: the oracle (oracle/convnext_oracle.py) against an independent implementation of the published network (transformers'
ConvNextModel; torchvision itself is not in this image: "parity pinned to a second implementation"), and the product module's
state_dict surface (torchvision's names, strict load).  GPU: the depthwise kernel exact on integers, the HIP tower against the oracle."""
import pytest
import torch
import torch.nn.functional as F

from oracle import convnext_oracle as co

F16, BF16, F32 = torch.float16, torch.bfloat16, torch.float32


def test_oracle_matches_an_independent_implementation():
    from transformers import ConvNextConfig, ConvNextModel
    p = co.init_params(seed=3)
    hf = ConvNextModel(ConvNextConfig(depths=list(co.DEPTHS), hidden_sizes=list(co.DIMS), layer_norm_eps=1e-6)).eval()
    missing, unexpected = hf.load_state_dict(co.to_hf_state_dict(p), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref = hf(x).last_hidden_state
    got = co.convnext_features(p, x)
    assert got.shape == ref.shape == (2, 768, 2, 3)
    assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    pooled = co.convnext_pooled_features(p, x)
    assert torch.allclose(pooled, ref.mean(dim=(2, 3)), atol=2e-5, rtol=1e-5)


def test_module_has_torchvisions_state_dict():
    from multimodal_propaganda_meme_classification_amd.convnext import ConvNeXtTiny
    m = ConvNeXtTiny()
    shapes = co.convnext_param_shapes()
    sd = m.state_dict()
    assert set(sd) == set(shapes)
    assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    assert sum(v.numel() for v in sd.values()) == 28_589_128          # torchvision's convnext_tiny parameter count
    m.load_state_dict(co.init_params(seed=1), strict=True)
    with pytest.raises(Exception):          # no CPU path
        m.pooled_features(torch.zeros(1, 3, 32, 32))


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import multimodal_propaganda_meme_classification_amd as m
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("T16", [F16, BF16])
@pytest.mark.parametrize("shape", [(2, 96, 56, 56), (3, 192, 28, 28), (2, 384, 14, 14), (2, 768, 7, 7), (1, 16, 9, 13), (2, 200, 5, 8)])
def test_depthwise_kernel_is_exact_on_integers(pkg, T16, shape):
    """small integers: every product and partial sum is exact in f32 and the result fits the 16-bit type, so any tap / halo / channel
    indexing error shows as a mismatch; shapes cover both tile sizes, ragged tiles and a channel count that is not a chunk multiple"""
    lib = pkg._lib.load("fp16" if T16 == F16 else "bf16")
    st = torch.cuda.current_stream().cuda_stream
    B, C, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + C)
    x = torch.randint(-2, 3, (B, C, H, W), generator=g).float()
    w = torch.randint(-1, 2, (C, 1, 7, 7), generator=g).float()
    b = torch.randint(-3, 4, (C,), generator=g).float()
    ref = F.conv2d(x, w, b, padding=3, groups=C)
    xd = x.permute(0, 2, 3, 1).contiguous().to(T16).cuda()
    wd, bd = w.cuda(), b.cuda()
    wt = torch.empty((49, C), dtype=F32, device="cuda")
    assert lib.mh_dwconv_weight_pack(wd.data_ptr(), wt.data_ptr(), C, 7, st) == 0
    y = torch.empty_like(xd)
    assert lib.mh_dwconv_nhwc(xd.data_ptr(), wt.data_ptr(), bd.data_ptr(), y.data_ptr(), B, H, W, C, 7, st) == 0
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert torch.equal(got, ref)
    assert lib.mh_dwconv_nhwc(xd.data_ptr(), wt.data_ptr(), None, y.data_ptr(), B, H, W, C, 7, st) == 0      # no bias
    assert torch.equal(y.float().cpu().permute(0, 3, 1, 2), F.conv2d(x, w, None, padding=3, groups=C))
    assert lib.mh_dwconv_nhwc(xd.data_ptr(), wt.data_ptr(), None, y.data_ptr(), B, H, W, C, 3, st) != 0      # only 7 x 7
    assert lib.mh_dwconv_nhwc(xd.data_ptr(), wt.data_ptr(), None, y.data_ptr(), B, H, W, 12, 7, st) != 0     # C % 8


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [("fp16", 4e-3), ("bf16", 3e-2)])
def test_pooled_features_match_the_oracle(pkg, dtype, tol):
    """avgpool(features(x)) of the full-depth network at 224 x 224 (the reference's ConvNeXt_Tiny_Weights.DEFAULT.transforms()
    crop) against the f32 oracle; 18 blocks of 16-bit activations: error relative to the feature spread, tolerance = 2x the
    measured value (fp16 1.9e-3, bf16 1.4e-2 on MI355X)."""
    from conftest import parity_log
    p = co.init_params(seed=5)
    m = pkg.ConvNeXtTiny(compute_dtype=dtype)
    m.load_state_dict(p, strict=True)
    m = m.cuda().eval()
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    ref = co.convnext_pooled_features(p, x)
    got = m.pooled_features(x.cuda()).cpu()
    assert got.shape == ref.shape == (2, 768) and got.dtype == F32
    err = float((got - ref).abs().max()) / float(ref.std())
    parity_log(f"convnext_tiny pooled features {dtype}", err)
    assert err < tol, err
    again = m.pooled_features(x.cuda()).cpu()
    assert torch.equal(got, again)                       # deterministic, weight images cached
    with torch.no_grad():
        m.features[1][0].layer_scale.mul_(2.0)           # a changed parameter rebuilds the packed weights
    assert not torch.equal(m.pooled_features(x.cuda()).cpu(), got)


@pytest.mark.gpu
def test_feature_dump_uses_the_convnext_encoder(pkg, tmp_path):
    """extract_feat.py's loop with image_model=ConvNeXtTiny: imgfeats are the 768-d pooled ConvNeXt features, textfeats the text tower's."""
    import json
    p = co.init_params(seed=7)
    cnx = pkg.ConvNeXtTiny(compute_dtype="fp16")
    cnx.load_state_dict(p, strict=True)
    cnx = cnx.cuda().eval()
    model = pkg.TextEncoder(pkg.TextConfig(vocab_size=128, hidden=128, layers=1, heads=2, intermediate=256, max_position=32),
                            compute_dtype="fp16").cuda().eval()
    g = torch.Generator().manual_seed(0)
    images = torch.randn(3, 3, 64, 64, generator=g)
    toks = torch.randint(1, 128, (3, 16), generator=g)
    loader = [(["a", "b", "c"], images, toks)]
    img_feats, text_feats = pkg.get_features(loader, model, torch.device("cuda"), image_model=cnx)
    ref = co.convnext_pooled_features(p, images)
    got = torch.tensor([img_feats[k] for k in ("a", "b", "c")])
    assert got.shape == (3, 768)
    assert float((got - ref).abs().max()) / float(ref.std()) < 4e-3
    assert len(text_feats["a"]) == 128
    path = pkg.dump_features(str(tmp_path / "features" / "f.json"), img_feats, text_feats)
    d = json.load(open(path))
    assert set(d) == {"imgfeats", "textfeats"} and len(d["imgfeats"]["b"]) == 768
