"""Implicit-GEMM convolution (csrc/convgemm.hip) on the MI355X through the C ABI: forward, input gradient and weight gradient
of the ResNet-50 convolutions (Multimodal_example_task2C.txt:164,183: models.resnet50) with no im2col panel.

Oracle: torch's fp32 conv2d on the CPU.  On small integers every product and partial sum is exactly representable in the 16-bit
operands and the f32 accumulators, so the comparison is BIT-EXACT up to the one rounding of the 16-bit store (the bar for index work:
which pixel, which tap, which channel goes where); the BatchNorm partial sums of the forward epilogue are compared the same way."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
F16, BF16, F32 = torch.float16, torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import multimodal_propaganda_meme_classification_amd as m
    return m


def _nhwc(x, T16):
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).to(T16).cuda().contiguous()


def _nchw(m, B, H, W):
    return m.float().cpu().view(B, H, W, -1).permute(0, 3, 1, 2)


def _geom(pkg, B, H, W, C, k, s, p, Cout, ldk):
    g = pkg._lib.MhConvGeom()
    g.B, g.H, g.W, g.C, g.KH, g.KW, g.stride, g.pad, g.Cout, g.ldk = B, H, W, C, k, k, s, p, Cout, ldk
    return g


def _pack(pkg, lib, w, Cp, ldk, T16):
    Cout, Cin, k, _ = w.shape
    wd = w.cuda().contiguous()
    wk = torch.empty((Cout, ldk), dtype=T16, device="cuda")
    pkg._lib.check(lib.mh_conv_weight_pack(wd.data_ptr(), wk.data_ptr(), Cout, Cin, k, k, Cp, ldk, torch.cuda.current_stream().cuda_stream), "pack")
    return wk


# (B, Cin, H, W, Cout, k, stride, pad): uniform-tap walk (Cin % 64 == 0) and the generic one (the stem's 8 padded channels,
# 24 channels), ragged M (rows past the last full tile), Cout below / above one column tile, 1x1 strided (downsample)
CASES = [
    (2, 64, 9, 7, 64, 3, 1, 1),
    (3, 128, 10, 10, 192, 3, 1, 1),
    (2, 64, 12, 11, 128, 3, 2, 1),
    (2, 256, 8, 8, 64, 1, 2, 0),
    (2, 64, 6, 6, 256, 1, 1, 0),
    (2, 8, 20, 18, 64, 7, 2, 3),
    (1, 24, 7, 9, 16, 3, 1, 1),
    (5, 64, 14, 14, 64, 3, 1, 1),
    (1, 16, 6, 5, 24, 5, 1, 2),       # 5x5, padding 2, one image
    (3, 64, 4, 4, 64, 3, 1, 1),       # 4x4 images: every window touches the padding
    (2, 72, 9, 9, 40, 3, 3, 0),       # stride 3, no padding, channel counts that are multiples of 8 only
    (1, 128, 3, 3, 128, 3, 1, 0),     # one output pixel per image
    (2, 64, 12, 10, 128, 3, 2, 1),    # stride 2 on an even image: the input gradient runs as four parity-class problems
    (3, 128, 8, 8, 64, 3, 2, 1),
    (2, 192, 14, 14, 256, 3, 2, 1),   # more than one row tile per class, two column tiles
]


def _dgrad_is_implicit(k, s, p, H, W, Cout):
    """what mh_conv_dgrad serves without a panel: stride 1, or 3x3-style stride 2 on even images (parity classes)"""
    if Cout % 64:
        return False
    return s == 1 or (s == 2 and k > 1 and H % 2 == 0 and W % 2 == 0)


@pytest.mark.parametrize("T16", [F16, BF16])
def test_implicit_conv_forward_and_bn_partials_are_exact_on_integers(pkg, T16):
    lib = pkg._lib.load("fp16" if T16 == F16 else "bf16")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(11)
    for (B, C, H, W, Cout, k, s, p) in CASES:
        x = torch.randint(-1, 2, (B, C, H, W), generator=g).float()
        w = torch.randint(-2, 3, (Cout, C, k, k), generator=g).float()
        ref = F.conv2d(x, w, stride=s, padding=p).to(T16).float()       # integer sums are exact in f32; the store rounds once
        Ho, Wo = ref.shape[2], ref.shape[3]
        M = B * Ho * Wo
        ldk = (k * k * C + 63) // 64 * 64
        xd, wk = _nhwc(x, T16), _pack(pkg, lib, w, C, ldk, T16)
        y = torch.full((M, Cout), 7.0, dtype=T16, device="cuda")
        nblk = (M + 127) // 128
        part = torch.full((2, Cout, nblk), -1.0, dtype=F32, device="cuda")
        geom = _geom(pkg, B, H, W, C, k, s, p, Cout, ldk)
        pkg._lib.check(lib.mh_conv_fwd(xd.data_ptr(), wk.data_ptr(), y.data_ptr(), part.data_ptr(), None, geom, st), "mh_conv_fwd")
        assert torch.equal(_nchw(y, B, Ho, Wo), ref), (B, C, H, W, Cout, k, s, p)
        rows = ref.permute(0, 2, 3, 1).reshape(M, Cout)
        pad_rows = torch.zeros((nblk * 128, Cout))
        pad_rows[:M] = rows
        blocks = pad_rows.view(nblk, 128, Cout)
        assert torch.equal(part[0].cpu(), blocks.sum(1).t().contiguous())
        assert torch.equal(part[1].cpu(), (blocks * blocks).sum(1).t().contiguous())
        # without the statistics request the output is the same
        y2 = torch.empty_like(y)
        pkg._lib.check(lib.mh_conv_fwd(xd.data_ptr(), wk.data_ptr(), y2.data_ptr(), None, None, geom, st), "mh_conv_fwd")
        assert torch.equal(y, y2)


@pytest.mark.parametrize("T16", [F16, BF16])
def test_implicit_conv_input_gradient_is_exact_on_integers(pkg, T16):
    lib = pkg._lib.load("fp16" if T16 == F16 else "bf16")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(12)
    for (B, C, H, W, Cout, k, s, p) in CASES:
        if not _dgrad_is_implicit(k, s, p, H, W, Cout):
            geom = _geom(pkg, B, H, W, C, k, s, p, Cout, (k * k * C + 63) // 64 * 64)
            d = torch.zeros(16, dtype=T16, device="cuda")
            assert lib.mh_conv_dgrad(d.data_ptr(), d.data_ptr(), d.data_ptr(), None, geom, None, st) == 2       # MH_ESHAPE: the explicit path serves it
            continue
        x = torch.zeros((B, C, H, W), requires_grad=True)
        w = torch.randint(-2, 3, (Cout, C, k, k), generator=g).float()
        yr = F.conv2d(x, w, stride=s, padding=p)
        Ho, Wo = yr.shape[2], yr.shape[3]
        dy = torch.randint(-2, 3, yr.shape, generator=g).float()
        yr.backward(dy)
        ldk = (k * k * C + 63) // 64 * 64
        wk = _pack(pkg, lib, w, C, ldk, T16)
        dyd = _nhwc(dy, T16)
        dx = torch.full((B * H * W, C), 7.0, dtype=T16, device="cuda")
        geom = _geom(pkg, B, H, W, C, k, s, p, Cout, ldk)
        pkg._lib.check(lib.mh_conv_dgrad(dyd.data_ptr(), wk.data_ptr(), dx.data_ptr(), None, geom, None, st), "mh_conv_dgrad")
        assert torch.equal(_nchw(dx, B, H, W), x.grad.to(T16).float()), (B, C, H, W, Cout, k, s, p)


@pytest.mark.parametrize("T16", [F16, BF16])
def test_implicit_conv_weight_gradient_is_exact_on_integers(pkg, T16):
    lib = pkg._lib.load("fp16" if T16 == F16 else "bf16")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(13)
    for (B, C, H, W, Cout, k, s, p) in CASES:
        x = torch.randint(-2, 3, (B, C, H, W), generator=g).float()
        w = torch.zeros((Cout, C, k, k), requires_grad=True)
        yr = F.conv2d(x, w, stride=s, padding=p)
        Ho, Wo = yr.shape[2], yr.shape[3]
        dy = torch.randint(-1, 2, yr.shape, generator=g).float()
        yr.backward(dy)
        M = B * Ho * Wo
        ldk = (k * k * C + 63) // 64 * 64
        xd, dyd = _nhwc(x, T16), _nhwc(dy, T16)
        geom = _geom(pkg, B, H, W, C, k, s, p, Cout, ldk)
        for want in (1, 3):
            sp = max(1, lib.mh_gemm_ksplit_for(M, want))
            slabs = torch.full((sp, Cout, ldk), 5.0, dtype=F32, device="cuda")
            pkg._lib.check(lib.mh_conv_wgrad(dyd.data_ptr(), xd.data_ptr(), slabs.data_ptr(), sp, 0.5, geom, st), "mh_conv_wgrad")
            gk = slabs.sum(0)[:, :k * k * C].cpu().view(Cout, k * k, C)          # [co][tap][ci]
            got = gk.permute(0, 2, 1).reshape(Cout, C, k, k)
            assert torch.equal(got, 0.5 * w.grad), (B, C, H, W, Cout, k, s, p, sp)
        # an impossible split is refused
        assert lib.mh_conv_wgrad(dyd.data_ptr(), xd.data_ptr(), slabs.data_ptr(), 10 ** 6, 1.0, geom, st) == 2


def test_implicit_conv_equals_the_explicit_im2col_path_bit_for_bit(pkg):
    """Same tile, same K order (tap-major, then channel): on random 16-bit data the implicit kernel and the im2col + grouped-GEMM
    path must agree in every bit, forward and weight gradient."""
    lib = pkg._lib.load("fp16")
    st = torch.cuda.current_stream().cuda_stream
    ops = pkg.ops
    g = torch.Generator().manual_seed(14)
    B, C, H, W, Cout, k, s, p = 4, 128, 28, 28, 128, 3, 1, 1
    x = torch.randn((B, C, H, W), generator=g)
    w = torch.randn((Cout, C, k, k), generator=g) * 0.05
    Ho, Wo = H, W
    M, ldk = B * Ho * Wo, k * k * C
    xd, wk = _nhwc(x, F16), _pack(pkg, lib, w, C, ldk, F16)
    col = torch.empty((M, ldk), dtype=F16, device="cuda")
    pkg._lib.check(lib.mh_im2col_nhwc(xd.data_ptr(), col.data_ptr(), B, H, W, C, k, k, s, p, ldk, st), "im2col")
    y_ref = torch.empty((M, Cout), dtype=F16, device="cuda")
    ops.gemm_grouped([ops.Gemm(col, wk, y_ref, M, Cout, ldk, ldk, ldk, Cout)], False, False)
    y = torch.empty_like(y_ref)
    geom = _geom(pkg, B, H, W, C, k, s, p, Cout, ldk)
    pkg._lib.check(lib.mh_conv_fwd(xd.data_ptr(), wk.data_ptr(), y.data_ptr(), None, None, geom, st), "mh_conv_fwd")
    assert torch.equal(y, y_ref)
    # against the fp32 oracle the error is the 16-bit rounding of the operands and of the result
    ref = F.conv2d(x.to(F16).float(), w.to(F16).float(), stride=s, padding=p)
    assert float((_nchw(y, B, Ho, Wo) - ref).abs().max()) < 2e-3 * float(ref.abs().max())
    dy = torch.randn((M, Cout), generator=g).to(F16).cuda()
    slabs_ref, sp = ops.wgrad_slabs(dy, col, Cout, ldk, M, Cout, ldk, alpha=1.0)
    slabs = torch.empty_like(slabs_ref)
    pkg._lib.check(lib.mh_conv_wgrad(dy.data_ptr(), xd.data_ptr(), slabs.data_ptr(), sp, 1.0, geom, st), "mh_conv_wgrad")
    assert torch.equal(slabs, slabs_ref)
    # dgrad: implicit vs GEMM + col2im.  The explicit path rounds each tap's product to 16 bits before col2im sums the taps; the
    # implicit one accumulates all taps in f32 -- it is the more accurate of the two, compare both with the fp32 oracle
    dcol = torch.empty((M, ldk), dtype=F16, device="cuda")
    ops.gemm_grouped([ops.Gemm(dy, wk, dcol, M, ldk, Cout, Cout, ldk, ldk)], False, True)
    dx_ref = torch.empty((B * H * W, C), dtype=F16, device="cuda")
    pkg._lib.check(lib.mh_col2im_nhwc(dcol.data_ptr(), dx_ref.data_ptr(), B, H, W, C, k, k, s, p, ldk, st), "col2im")
    dx = torch.empty_like(dx_ref)
    pkg._lib.check(lib.mh_conv_dgrad(dy.data_ptr(), wk.data_ptr(), dx.data_ptr(), None, geom, None, st), "mh_conv_dgrad")
    xr = x.to(F16).float().requires_grad_(True)
    F.conv2d(xr, w.to(F16).float(), stride=s, padding=p).backward(_nchw(dy, B, Ho, Wo))
    scale = float(xr.grad.abs().max())
    e_imp = float((_nchw(dx, B, H, W) - xr.grad).abs().max())
    e_exp = float((_nchw(dx_ref, B, H, W) - xr.grad).abs().max())
    assert e_imp < 1.5e-3 * scale and e_imp <= e_exp * 1.05


@pytest.mark.parametrize("T16", [F16, BF16])
def test_split_k_forward_and_dgrad_equal_the_unsplit_kernels_on_integers(pkg, T16):
    """Deep, few-tile layers (ResNet-50 layer3 / layer4 at batch 32) are cut into K chunks: on integers the slab sum is exact, so the
    split results -- output, BatchNorm partials, input gradient -- must equal the unsplit ones bit for bit (and torch's conv2d)."""
    lib = pkg._lib.load("fp16" if T16 == F16 else "bf16")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(15)
    for (B, C, H, W, Cout, k, s, p) in ((2, 512, 7, 7, 512, 3, 1, 1), (3, 256, 14, 14, 256, 3, 1, 1), (2, 2048, 7, 7, 512, 1, 1, 0), (4, 256, 14, 14, 256, 3, 2, 1)):
        x = torch.randint(-1, 2, (B, C, H, W), generator=g).float()
        w = torch.randint(-1, 2, (Cout, C, k, k), generator=g).float()
        xr = x.clone().requires_grad_(True)
        yr = F.conv2d(xr, w, stride=s, padding=p)
        Ho, Wo = yr.shape[2], yr.shape[3]
        M, ldk = B * Ho * Wo, k * k * C
        xd, wk = _nhwc(x, T16), _pack(pkg, lib, w, C, ldk, T16)
        geom = _geom(pkg, B, H, W, C, k, s, p, Cout, ldk)
        sp = int(lib.mh_conv_splitk(geom, 0))
        assert sp > 1, (B, C, H, W, Cout, k, s, p)
        nblk = (M + 127) // 128
        outs = []
        for ws in (None, torch.empty((sp, M, Cout), dtype=F32, device="cuda")):
            y = torch.full((M, Cout), 7.0, dtype=T16, device="cuda")
            part = torch.full((2, Cout, nblk), -1.0, dtype=F32, device="cuda")
            pkg._lib.check(lib.mh_conv_fwd(xd.data_ptr(), wk.data_ptr(), y.data_ptr(), part.data_ptr(), None if ws is None else ws.data_ptr(), geom, st), "fwd")
            outs.append((y, part))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert torch.equal(_nchw(outs[1][0], B, Ho, Wo), yr.detach().to(T16).float())
        if s != 1:
            continue
        dy = torch.randint(-1, 2, yr.shape, generator=g).float()
        yr.backward(dy)
        dyd = _nhwc(dy, T16)
        spd = int(lib.mh_conv_splitk(geom, 1))
        assert spd > 1 or k == 1          # (the 1x1 layer's input gradient contracts over 512 filters only: too short to cut)
        res = []
        for ws in (None, torch.empty((max(spd, 1), B * H * W, C), dtype=F32, device="cuda")):
            dx = torch.full((B * H * W, C), 7.0, dtype=T16, device="cuda")
            pkg._lib.check(lib.mh_conv_dgrad(dyd.data_ptr(), wk.data_ptr(), dx.data_ptr(), None if ws is None else ws.data_ptr(), geom, None, st), "dgrad")
            res.append(dx)
        assert torch.equal(res[0], res[1]) and torch.equal(_nchw(res[1], B, H, W), xr.grad.to(T16).float())


def test_strided_dgrad_runs_as_parity_classes_with_the_batchnorm_epilogue(pkg):
    """3x3 / stride 2 (ResNet-50's layer2-4 first blocks): mh_conv_dgrad launches the four parity classes of input pixels as one grouped
    implicit-GEMM launch -- against the explicit path (GEMM into the tap panel + col2im) on integers, and, with `bn`, the masked
    gradient and the TOTALS of the per-tile BatchNorm sums (tiles are per class, mh_bn2d_bwd_parts adds them all)."""
    lib = pkg._lib.load("fp16")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(21)
    for (B, C, H, W, Cout) in ((4, 128, 28, 28, 128), (2, 256, 14, 14, 256), (3, 64, 10, 6, 64)):
        k, s_, p = 3, 2, 1
        Ho, Wo = H // 2, W // 2
        Mi = B * H * W
        w = torch.randint(-1, 2, (Cout, C, k, k), generator=g).float()
        dy = torch.randint(-2, 3, (B, Cout, Ho, Wo), generator=g).float()
        x = torch.zeros((B, C, H, W), requires_grad=True)
        F.conv2d(x, w, stride=s_, padding=p).backward(dy)
        ldk = k * k * C
        wk, dyd = _pack(pkg, lib, w, C, ldk, F16), _nhwc(dy, F16)
        geom = _geom(pkg, B, H, W, C, k, s_, p, Cout, ldk)
        assert int(lib.mh_conv_splitk(geom, 1)) == 1
        dx_ref = torch.full((Mi, C), 7.0, dtype=F16, device="cuda")
        pkg._lib.check(lib.mh_conv_dgrad(dyd.data_ptr(), wk.data_ptr(), dx_ref.data_ptr(), None, geom, None, st), "strided dgrad")
        assert torch.equal(_nchw(dx_ref, B, H, W), x.grad.to(F16).float()), (B, C, H, W)
        z = (torch.randn((Mi, C), generator=g) * 1.3 + 0.2).to(F16).cuda()
        mean, rstd = (torch.randn(C, generator=g) * 0.3).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
        gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
        nblk = 4 * ((B * Ho * Wo + 127) // 128)
        f = pkg._lib.MhConvBnBwd()
        part = torch.full((2, C, nblk), -7.0, dtype=F32, device="cuda")
        f.z, f.mean, f.rstd, f.gamma, f.beta, f.part, f.relu = z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), part.data_ptr(), 1
        dx = torch.empty_like(dx_ref)
        pkg._lib.check(lib.mh_conv_dgrad(dyd.data_ptr(), wk.data_ptr(), dx.data_ptr(), None, geom, f, st), "strided dgrad + bn")
        xh = (z.float() - mean) * rstd
        v = xh * gamma + beta
        keep = v.to(F16).float() > 0
        want = torch.where(keep, dx_ref.float(), torch.zeros_like(v))
        sure = v.abs() > 1e-3
        assert torch.equal(dx.float()[sure], want[sure]), (B, C, H, W)
        gm = dx.float()
        assert torch.allclose(part[0].sum(1), gm.sum(0), rtol=1e-4, atol=5e-2) and torch.allclose(part[1].sum(1), (gm * xh).sum(0), rtol=1e-4, atol=5e-2)


def test_dgrad_epilogue_does_the_producing_batchnorms_mask_and_statistics(pkg):
    """mh_conv_dgrad with `bn`: the input gradient of a convolution IS the dy of the BatchNorm (+ReLU) that produced its input; the
    epilogue (or the split-K finishing launch) masks it and leaves sum g', sum g' xhat per 128-row tile.  Checked against the unfused
    dgrad + the definition, for an unsplit 3x3 layer, a split-K one and a 1x1 layer."""
    lib = pkg._lib.load("fp16")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(16)
    for (B, C, H, W, Cout, k, p) in ((4, 64, 14, 14, 64, 3, 1), (2, 512, 7, 7, 512, 3, 1), (3, 128, 10, 10, 256, 1, 0)):
        Mi = B * H * W
        w = torch.randint(-1, 2, (Cout, C, k, k), generator=g).float()
        dy = torch.randint(-2, 3, (B, Cout, H, W), generator=g).float()
        ldk = k * k * C
        wk, dyd = _pack(pkg, lib, w, C, ldk, F16), _nhwc(dy, F16)
        geom = _geom(pkg, B, H, W, C, k, 1, p, Cout, ldk)
        sp = int(lib.mh_conv_splitk(geom, 1))
        ws = torch.empty((sp, Mi, C), dtype=F32, device="cuda") if sp > 1 else None
        dx_ref = torch.empty((Mi, C), dtype=F16, device="cuda")
        pkg._lib.check(lib.mh_conv_dgrad(dyd.data_ptr(), wk.data_ptr(), dx_ref.data_ptr(), None if ws is None else ws.data_ptr(), geom, None, st), "dgrad")
        z = (torch.randn((Mi, C), generator=g) * 1.3 + 0.2).to(F16).cuda()
        mean, rstd = (torch.randn(C, generator=g) * 0.3).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
        gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
        nblk = (Mi + 127) // 128
        for relu in (1, 0):
            f = pkg._lib.MhConvBnBwd()
            part = torch.full((2, C, nblk), -7.0, dtype=F32, device="cuda")
            f.z, f.mean, f.rstd, f.gamma, f.beta, f.part, f.relu = z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), part.data_ptr(), relu
            dx = torch.empty_like(dx_ref)
            pkg._lib.check(lib.mh_conv_dgrad(dyd.data_ptr(), wk.data_ptr(), dx.data_ptr(), None if ws is None else ws.data_ptr(), geom, f, st), "dgrad+bn")
            xh = (z.float() - mean) * rstd
            v = xh * gamma + beta
            keep = (v.to(F16).float() > 0) if relu else torch.ones_like(v, dtype=torch.bool)
            want = torch.where(keep, dx_ref.float(), torch.zeros_like(v))
            sure = (v.abs() > 1e-3) | (relu == 0)          # (a value a rounding away from zero may fall either way)
            assert torch.equal(dx.float()[sure], want[sure]), (B, C, H, W, Cout, k, relu)
            gm = dx.float()                                # the statistics are of what was stored
            pad = torch.zeros((nblk * 128, C), device="cuda")
            pad[:Mi] = gm
            pad2 = torch.zeros((nblk * 128, C), device="cuda")
            pad2[:Mi] = gm * xh
            s0, s1 = pad.view(nblk, 128, C).sum(1).t(), pad2.view(nblk, 128, C).sum(1).t()
            assert torch.allclose(part[0], s0, rtol=1e-4, atol=1e-2) and torch.allclose(part[1], s1, rtol=1e-4, atol=1e-2), (B, C, k, relu)
