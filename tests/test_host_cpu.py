"""Host-side logic that needs no GPU: JSON loader / Dataset surface, TSV format, flat parameter
layout, gradient-bucket cover, and the world_size-2 gloo all-reduce path of the DDP reducer."""
import os
import re

import numpy as np
import pytest
import torch

import multimodal_propaganda_meme_classification_amd as pkg
from multimodal_propaganda_meme_classification_amd import ddp
from multimodal_propaganda_meme_classification_amd.config import Layout

# the task's format checker regex, restated (format_checker/task2.py:20)
LINE = re.compile(r'^([\w:]+\/.*?\.[\w:]+)\t(propaganda|not_propaganda)\t[\w-]+')


def _tiny_cfg():
    return pkg.ModelConfig(text=pkg.TextConfig(vocab_size=512, hidden=128, layers=2, heads=2, intermediate=256, max_position=64),
                           image=pkg.ImageConfig(image_size=32, hidden=128, layers=2, heads=2, intermediate=256), proj=128)


def test_read_data_and_dataset_surface(golden_dir):
    df = pkg.read_data(os.path.join(golden_dir, "dev_sample12.json"))
    assert list(df.columns) == ["id", "text", "image", "label"] and len(df) == 12
    assert list(pkg.read_data(os.path.join(golden_dir, "dev_sample12.json"), is_test=True).columns) == ["id", "text", "image"]
    labels = df["label"].map(pkg.l2id)
    ds = pkg.MultimodalDataset(df["id"], df["text"], df["image"], labels, max_seq_len=32, image_size=32, synthetic_images=True,
                               vocab_size=512)
    assert len(ds) == 12
    item = ds[3]
    assert set(item) == {"id", "text", "text_mask", "image", "label"}
    assert item["text"].dtype == torch.int64 and item["text"].shape == (32,)
    assert item["text_mask"].dtype == torch.int64 and item["image"].shape == (3, 32, 32) and item["image"].dtype == torch.float32
    n = int(item["text_mask"].sum())
    assert item["text"][0] == 2 and item["text"][n - 1] == 3 and (item["text"][n:] == 0).all()      # [CLS] .. [SEP] PAD..
    assert int(item["text"].max()) < 512
    assert torch.equal(ds[3]["image"], item["image"])                                              # deterministic
    assert "label" not in pkg.MultimodalDataset(df["id"], df["text"], df["image"], None, is_test=True, synthetic_images=True)[0]
    with pytest.raises(FileNotFoundError):
        pkg.MultimodalDataset(df["id"], df["text"], df["image"], labels)[0]
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=4)))
    assert batch["text"].shape == (4, 32) and batch["image"].shape == (4, 3, 32, 32) and len(batch["id"]) == 4


def test_tsv_lines_pass_the_task_format(tmp_path, golden_dir):
    df = pkg.read_data(os.path.join(golden_dir, "dev_sample12.json"))
    for id_ in df["id"]:
        for lab in pkg.id2l.values():
            assert LINE.match(f"{id_}\t{lab}\tViT-BERT-memehip")


def test_image_transform_matches_definition(tmp_path):
    from PIL import Image
    from multimodal_propaganda_meme_classification_amd.data import load_image
    arr = (np.arange(300 * 400 * 3) % 251).astype(np.uint8).reshape(300, 400, 3)
    path = str(tmp_path / "x.png")
    Image.fromarray(arr).save(path)
    x = load_image(path)
    assert x.shape == (3, 224, 224)
    ref = Image.open(path).convert("RGB").resize((341, 256), Image.BILINEAR)            # shorter side -> 256
    ref = np.asarray(ref.crop((58, 16, 58 + 224, 16 + 224)), dtype=np.float32) / 255.0
    ref = (ref - np.array([0.485, 0.456, 0.406], dtype=np.float32)) / np.array([0.229, 0.224, 0.225], dtype=np.float32)
    np.testing.assert_allclose(x.numpy(), ref.transpose(2, 0, 1), atol=1e-6)


def test_layout_config3():
    lay = Layout(pkg.ModelConfig())
    real = sum(s.numel for s in lay.specs)
    assert abs(real - 221.7e6) < 0.1e6
    assert lay.n_total - real < 16                         # only alignment padding
    for s in lay.specs:
        assert s.offset % 4 == 0
    q = lay.spec["bert.encoder.layer.3.attention.self.query.weight"]
    k = lay.spec["bert.encoder.layer.3.attention.self.key.weight"]
    v = lay.spec["bert.encoder.layer.3.attention.self.value.weight"]
    assert k.offset == q.offset + q.numel and v.offset == k.offset + k.numel       # fused QKV operand is a slice
    qb = lay.spec["image_model.encoder.layer.0.attention.attention.query.bias"]
    kb = lay.spec["image_model.encoder.layer.0.attention.attention.key.bias"]
    assert kb.offset == qb.offset + qb.numel
    assert lay.n_shadow == 2 * 12 * (4 * 768 * 768 + 2 * 768 * 3072) + 768 * 768
    # gradient buckets (one per layer pair, then everything else) tile the flat buffer exactly once
    buckets = {f"bwd_layer_{l}": (a, b) for l, a, b in lay.layer_ranges}
    buckets["bwd_embed"] = (lay.layer_ranges[-1][2], lay.n_total)
    assert ddp.check_bucket_cover(buckets, lay.n_total) == lay.n_total
    with pytest.raises(AssertionError):
        ddp.check_bucket_cover({"a": (0, 10), "b": (12, 20)}, 20)
    names = lay.state_dict_order()
    assert names[0] == "bert.embeddings.word_embeddings.weight" and names[-1] == "output_fc.bias"


def test_module_protocol_on_cpu():
    m = pkg.MultimodalClassifier.from_config(_tiny_cfg(), seed=3)
    sd = m.state_dict()
    assert list(sd) == m.layout.state_dict_order()
    assert len(list(m.parameters())) == len(sd)
    assert all(p.grad is not None and p.grad.shape == p.shape for p in m.parameters())
    m2 = pkg.MultimodalClassifier.from_config(_tiny_cfg(), init=False)
    m2.load_state_dict({("resnet_fc" + k[8:] if k.startswith("image_fc") else k): v for k, v in sd.items()})   # alias
    assert torch.equal(m2.flat_params, m.flat_params)
    with pytest.raises(RuntimeError):
        m2.load_state_dict({"bert_fc.weight": sd["bert_fc.weight"]})
    with pytest.raises(ValueError):
        pkg.ModelConfig(pool="max").validate()                 # "Unsupported pooling type" like the reference
    with pytest.raises(TypeError):
        m.half()
    with pytest.raises(pkg.MemehipError):
        m(torch.zeros((1, 8), dtype=torch.long), torch.zeros((1, 3, 32, 32)), torch.ones((1, 8), dtype=torch.long))


def _ddp_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1000
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)
        red = ddp.GradientReducer(g, bucket_cap_elems=128)
        for rng in ((0, 300), (300, 301), (301, 1000)):         # ragged buckets, issued in order
            red.hook("seg", rng)
        red.hook("none", None)
        red.wait()
        want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok = torch.equal(g, want) and red.grad_scale == 1.0 / world and red.reduced_elems == n
        p = torch.full((16,), float(rank))
        ddp.broadcast_parameters(p, src=0)
        ok = ok and bool((p == 0).all())
        out[rank] = ok
    finally:
        dist.destroy_process_group()


def _ddp_bf16_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 5000
        gen = torch.Generator().manual_seed(100 + rank)
        mine = torch.randn(n, generator=gen)
        g32, g16 = mine.clone(), mine.clone()
        ddp.GradientReducer(g32, bucket_cap_elems=2048).hook("a", (0, n))
        red = ddp.GradientReducer(g16, bucket_cap_elems=2048, compress="bf16")
        for rng in ((0, 1000), (1000, 1008), (1008, n)):
            for w in red.reduce_range(rng):
                w.wait()
        red.wait()
        # exact model of the compressed path: bf16(sum_r bf16(g_r)) with the sum in fp32
        parts = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)).to(torch.bfloat16).float() for r in range(world)]
        want = torch.stack(parts).sum(0).to(torch.bfloat16).float()
        ok = torch.equal(g16, want)
        ok = ok and float((g16 - g32).abs().max()) <= 2.0 ** -7 * float(g32.abs().max())     # within bf16 rounding of the fp32 sum
        ok = ok and red.reduced_elems == n and red.wire_bytes < 0.6 * 4 * 2 * (world - 1) / world * n + 4096   # about half the fp32 bytes
        gathered = [torch.empty_like(g16) for _ in range(world)]
        dist.all_gather(gathered, g16)
        ok = ok and all(torch.equal(gathered[0], t) for t in gathered)                        # identical on every rank
        out[rank] = ok
    finally:
        dist.destroy_process_group()


def test_gradient_reducer_bf16_compression_gloo_world2():
    """bf16 on the wire, fp32 accumulation on receipt (all-to-all + all-gather): against the fp32 all-reduce and an exact
    model of the compressed arithmetic."""
    import torch.multiprocessing as mp
    world = 2
    from conftest import free_port
    port = free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ddp_bf16_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
    with pytest.raises(ValueError):
        ddp.GradientReducer(torch.zeros(8), compress="int8")


def test_gradient_reducer_gloo_world2():
    import torch.multiprocessing as mp
    world = 2
    from conftest import free_port
    port = free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ddp_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_resize_and_crop_follow_torchvision_formulas():
    """torchvision 0.17.2: Resize(int) truncates the long side, CenterCrop rounds (half to even) the offset --
    known answers worked by hand from transforms/functional.py (_compute_resized_output_size, center_crop)."""
    from multimodal_propaganda_meme_classification_amd.data import center_crop_box, resized_size
    assert resized_size(640, 427) == (383, 256)            # 256 * 640 / 427 = 383.7 -> 383 (round() would give 384)
    assert resized_size(427, 640) == (256, 383)
    assert resized_size(400, 300) == (341, 256)
    assert resized_size(500, 500) == (256, 256)
    assert center_crop_box(383, 256, 224) == (80, 16, 304, 240)     # (383-224)/2 = 79.5 -> 80 (half to even), not 79
    assert center_crop_box(341, 256, 224) == (58, 16, 282, 240)     # 58.5 -> 58
    assert center_crop_box(256, 256, 224) == (16, 16, 240, 240)


def test_image_projection_name_is_configurable():
    """state_dict() can carry the organizers' `resnet_fc.*` keys (Multimodal_example_task2C.txt:165); either name loads."""
    cfg = _tiny_cfg()
    cfg.image_fc_name = "resnet_fc"
    m = pkg.MultimodalClassifier.from_config(cfg, seed=1)
    keys = list(m.state_dict())
    assert "resnet_fc.weight" in keys and "resnet_fc.bias" in keys and not any(k.startswith("image_fc") for k in keys)
    assert [n for n, _ in m.named_parameters()] == keys
    m2 = pkg.MultimodalClassifier.from_config(_tiny_cfg(), init=False)             # default name: image_fc
    m2.load_state_dict(m.state_dict())
    assert torch.equal(m2.flat_params, m.flat_params)
    m3 = pkg.MultimodalClassifier.from_config(cfg, init=False)
    m3.load_state_dict(m2.state_dict())
    assert torch.equal(m3.flat_params, m.flat_params)
    bad = _tiny_cfg()
    bad.image_fc_name = "vit_fc"
    with pytest.raises(ValueError):
        bad.validate()


def test_pil_resample_restatement_is_bit_exact():
    """data.pil_resample_coeffs (PIL's precompute_coeffs / normalize_coeffs_8bpc restated) + the two-pass uint8 arithmetic the
    HIP kernels run (data.resample_u8_reference) against PIL's own Image.resize(BILINEAR): every pixel equal, for down- and
    up-scaling, for the centre-crop window of Resize(256) and for Resize((224, 224))."""
    from PIL import Image
    from multimodal_propaganda_meme_classification_amd.data import (center_crop_box, pil_resample_coeffs, resample_u8_reference,
                                                                    resized_size)
    rng = np.random.default_rng(0)
    for (h, w) in ((300, 400), (427, 640), (97, 131), (1000, 333)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        nw, nh = resized_size(w, h, 256)
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        left, top, _, _ = center_crop_box(nw, nh, 224)
        got = resample_u8_reference(img, *pil_resample_coeffs(w, nw, left, 224), *pil_resample_coeffs(h, nh, top, 224))
        assert np.array_equal(got, ref[top:top + 224, left:left + 224]), (h, w)
        ref2 = np.asarray(Image.fromarray(img).resize((224, 224), Image.BILINEAR))
        got2 = resample_u8_reference(img, *pil_resample_coeffs(w, 224), *pil_resample_coeffs(h, 224))
        assert np.array_equal(got2, ref2), (h, w)


def test_kevin_dataset_keys_and_collate_on_the_host():
    """Kevin's dataset (Multimodal_example_task2C.py:208-304): constructor order, dict keys with and without labels, captions
    required (the BLIP captioner is not part of this package) or produced by a caller's function, kevin_collate."""
    import multimodal_propaganda_meme_classification_amd as pkg
    kv = pkg.kevin
    ids, texts, imgs, labels = ["a", "b", "c"], ["one two", "three", "four five six"], ["x.png", "y.png", "z.png"], [0, 1, 0]
    with pytest.raises(ValueError, match="captions"):
        kv.KevinMultimodalDataset(ids, texts, imgs, labels)
    with pytest.raises(ValueError, match="2 captions for 3"):
        kv.KevinMultimodalDataset(ids, texts, imgs, labels, captions=["p", "q"])
    seen = []
    ds = kv.KevinMultimodalDataset(ids, texts, imgs, labels, caption_fn=lambda paths: seen.extend(paths) or [f"a meme of {i}" for i in range(3)],
                                   max_seq_len=12, image_size=16, synthetic_images=True, image_root="root")
    assert seen == [os.path.join("root", p) for p in imgs] and len(ds) == 3
    it = ds[1]
    assert list(it) == ["id", "text", "text_mask", "caption_text", "caption_text_mask", "image", "label"]      # the reference's order (:290-300)
    assert it["text"].shape == (12,) and it["caption_text"].shape == (12,) and int(it["label"]) == 1
    assert it["image"].dtype == np.uint8 and it["image"].shape == (16, 16, 3)
    ds_t = kv.KevinMultimodalDataset(ids, texts, imgs, None, is_test=True, captions=["p", "q", "r"], synthetic_images=True, image_size=16)
    assert "label" not in ds_t[0]
    with pytest.raises(FileNotFoundError):
        kv.KevinMultimodalDataset(ids, texts, imgs, labels, captions=["p", "q", "r"])[0]
    batch = kv.kevin_collate([ds[0], ds[2]])
    assert batch["text"].shape == (2, 12) and batch["label"].tolist() == [0, 0] and batch["id"] == ["a", "c"]
    assert isinstance(batch["image"], list) and len(batch["image"]) == 2


def test_hue_and_rotation_restatements_are_bit_exact_against_pil():
    """The two augmentations round 2 left pinned to nothing: torchvision's adjust_hue on a PIL image (Pillow's 8-bit HSV round trip
    with a wrapped H shift) and F.rotate = Image.rotate(angle, NEAREST, expand=False, fillcolor=0) (16.16 fixed-point affine walk).
    data.hue_shift_u8_reference is the numpy statement of the kernel's hue arithmetic, data.pil_rotate_fixed_coeffs what the host
    hands the rotation kernel; both must reproduce PIL bit for bit (the GPU test then pins the kernels to the same PIL calls)."""
    from PIL import Image
    from oracle import ref_env as E
    from multimodal_propaganda_meme_classification_amd.data import hue_shift_u8_reference, pil_rotate_fixed_coeffs
    rng = np.random.default_rng(4)
    # a 1024 x 1024 image: random colours + every grey + saturated primaries
    rgb = rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
    rgb[0, :256] = np.arange(256, dtype=np.uint8)[:, None]
    rgb[1, :6] = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255], [255, 0, 255]], dtype=np.uint8)
    img = Image.fromarray(rgb)
    for hf in (0.0, 0.0713, -0.0831, 0.1, -0.1, 0.5, -0.5, 0.0039):
        assert np.array_equal(hue_shift_u8_reference(rgb, hf), np.asarray(E.adjust_hue_pil(img, hf))), hf

    def rotate(a, co):
        h, w = a.shape[:2]
        ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
        xin, yin = (co[2] + ys * co[1] + xs * co[0]) >> 16, (co[5] + ys * co[4] + xs * co[3]) >> 16
        ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
        out = np.zeros_like(a)
        out[ok] = a[yin[ok], xin[ok]]
        return out

    for (h, w) in ((224, 224), (64, 96), (97, 131)):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        im = Image.fromarray(a)
        for ang in (3.7, -14.99, 11.25, 0.31, -7.0, 15.0, -15.0, 45.0, 123.4):
            assert np.array_equal(rotate(a, pil_rotate_fixed_coeffs(ang, w, h)), np.asarray(E.rotate_pil(im, ang))), (h, w, ang)
    assert pil_rotate_fixed_coeffs(0.0, 10, 10) is None and pil_rotate_fixed_coeffs(360.0, 10, 10) is None
