"""Exact (integer) checks of the grouped GEMM's EPILOGUE paths through the C ABI: 16-bit and f32 output in the three layouts, every
epilogue operand (bias, residual, stored-derivative multiply, aux copy) alone and combined, ragged M, partial last column tile,
packed rows (rows_dev), split-K slabs.  (Written for the direct-from-accumulator epilogue of round 4 -- a lab kernel now -- and kept
because the older tests checked the 16-bit output path only approximately.)  Small-integer operands make every product, sum and 16-bit rounding exact,
so the result must equal the fp32 torch expression bit for bit -- a swapped lane, column block or operand piece cannot hide.
(reference call sites: every nn.Linear behind Multimodal_example_task2C.txt:175,183 and loss.backward() :216)"""
import pytest
import torch

pytestmark = pytest.mark.gpu

F32 = torch.float32
BOTH = pytest.mark.parametrize("T16", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodal_propaganda_meme_classification_amd import ops as _ops
    return _ops


def dev():
    return torch.device("cuda:0")


def ints(*shape, lo=-1, hi=2, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).to(dtype).to(dev())


def _operands(layout, M, N, K, T16):
    """A, B as the layout stores them + the fp32 product A.B^T [M, N]"""
    if layout == (0, 0):
        A, Bw = ints(M, K, seed=1, dtype=T16), ints(N, K, seed=2, dtype=T16)
        return A, Bw, A.float() @ Bw.float().t(), (K, K)
    if layout == (0, 1):
        A, Bk = ints(M, K, seed=1, dtype=T16), ints(K, N, seed=2, dtype=T16)
        return A, Bk, A.float() @ Bk.float(), (K, N)
    A, Bk = ints(K, M, seed=1, dtype=T16), ints(K, N, seed=2, dtype=T16)
    return A, Bk, A.float().t() @ Bk.float(), (M, N)


@BOTH
@pytest.mark.parametrize("layout", [(0, 0), (0, 1), (1, 1)], ids=["fwd", "dgrad", "wgrad"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (394, 192, 128), (1000, 384, 192), (264, 64, 64)])
def test_16bit_output_is_exact(ops, layout, M, N, K, T16):
    """|sum| <= K <= 192 and the operands added below keep every value an integer below 256: exact in bf16 and fp16."""
    if layout[0] == 1:
        M = (M + 7) // 8 * 8
    A, B, ref, (lda, ldb) = _operands(layout, M, N, K, T16)
    bias = ints(N, lo=-3, hi=4, seed=3, dtype=F32)
    res = ints(M, N, lo=-8, hi=9, seed=4, dtype=T16)
    out = torch.full((M, N), 7.0, dtype=T16, device=dev())
    ops.gemm_grouped([ops.Gemm(A, B, out, M, N, K, lda, ldb, N)], bool(layout[0]), bool(layout[1]))
    assert torch.equal(out.float(), ref), "plain store"
    aux = torch.full((M, N), 5.0, dtype=T16, device=dev())
    ops.gemm_grouped([ops.Gemm(A, B, out, M, N, K, lda, ldb, N, bias=bias, residual=res, aux=aux)], bool(layout[0]), bool(layout[1]))
    assert torch.equal(aux.float(), ref + bias), "aux = acc + bias"
    assert torch.equal(out.float(), ref + bias + res.float()), "bias + residual"
    # the stored-derivative multiply (MH_GEMM_DERIV_AUX: `mul` is multiplied in as it is), alone and together with a residual
    mul = ints(M, N, lo=-1, hi=2, seed=5, dtype=T16)
    ops.gemm_grouped([ops.Gemm(A, B, out, M, N, K, lda, ldb, N, mul=mul, deriv_aux=True)], bool(layout[0]), bool(layout[1]))
    assert torch.equal(out.float(), ref * mul.float()), "mul"
    ops.gemm_grouped([ops.Gemm(A, B, out, M, N, K, lda, ldb, N, bias=bias, mul=mul, residual=res, deriv_aux=True)],
                     bool(layout[0]), bool(layout[1]))
    assert torch.equal(out.float(), (ref + bias) * mul.float() + res.float()), "bias, mul and residual"


@BOTH
def test_f32_output_operands_and_accumulate(ops, T16):
    M, N, K = 394, 320, 128
    A, B, ref, (lda, ldb) = _operands((0, 0), M, N, K, T16)
    bias = ints(N, lo=-3, hi=4, seed=3, dtype=F32)
    res = ints(M, N, lo=-8, hi=9, seed=4, dtype=T16)
    mul = ints(M, N, lo=-2, hi=3, seed=5, dtype=T16)
    out = torch.zeros((M, N), dtype=F32, device=dev())
    ops.gemm_grouped([ops.Gemm(A, B, out, M, N, K, lda, ldb, N, bias=bias, residual=res, mul=mul, deriv_aux=True, alpha=0.5)], False, False)
    want = (0.5 * ref + bias) * mul.float() + res.float()
    assert torch.equal(out, want)
    ops.gemm_grouped([ops.Gemm(A, B, out, M, N, K, lda, ldb, N, accum=True)], False, False)
    assert torch.equal(out, want + ref), "accumulate into f32"


@BOTH
def test_grouped_problems_with_packed_rows(ops, T16):
    """Two problems in one launch, one of them with a device-side live row count: rows past it keep what the output held."""
    K, N = 192, 256
    M1, M2, live = 512, 394, 300
    A1, B1, ref1, _ = _operands((0, 0), M1, N, K, T16)
    A2 = ints(M2, K, seed=7, dtype=T16)
    B2 = ints(N, K, seed=8, dtype=T16)
    ref2 = A2.float() @ B2.float().t()
    res1 = ints(M1, N, lo=-8, hi=9, seed=9, dtype=T16)
    o1 = torch.full((M1, N), 3.0, dtype=T16, device=dev())
    o2 = torch.full((M2, N), 3.0, dtype=T16, device=dev())
    rows = torch.tensor([live], dtype=torch.int32, device=dev())
    ops.gemm_grouped([ops.Gemm(A1, B1, o1, M1, N, K, K, K, N, residual=res1, rows_dev=rows),
                      ops.Gemm(A2, B2, o2, M2, N, K, K, K, N)], False, False)
    assert torch.equal(o1[:live].float(), (ref1 + res1.float())[:live])
    assert torch.equal(o1[live:].float(), torch.full((M1 - live, N), 3.0, device=dev())), "rows past the live count are not stored"
    assert torch.equal(o2.float(), ref2)
    # the weight-gradient layout contracts over the packed rows: rows past the live count must not enter the sum
    dy = ints(M1, N, seed=10, dtype=T16)
    x = ints(M1, K, seed=11, dtype=T16)
    dw = torch.zeros((N, K), dtype=F32, device=dev())
    db = torch.zeros(N, dtype=F32, device=dev())
    ops.gemm_grouped([ops.Gemm(dy, x, dw, N, K, M1, N, K, K, rowsum=db, rows_dev=rows)], True, True)
    assert torch.equal(dw, dy[:live].float().t() @ x[:live].float())
    assert torch.equal(db, dy[:live].float().sum(0))


def test_split_k_slabs_are_exact(ops):
    T16 = torch.bfloat16
    M, N, K = 256, 192, 1024
    A, B, ref, (lda, ldb) = _operands((1, 1), M, N, K, T16)
    from multimodal_propaganda_meme_classification_amd import _lib
    sp = _lib.load("bf16").mh_gemm_ksplit_for(K, 4)
    slabs = torch.full((sp, M, N), 9.0, dtype=F32, device=dev())
    ops.gemm_grouped([ops.Gemm(A, B, slabs, M, N, K, lda, ldb, N, ksplit=sp)], True, True)
    assert torch.equal(slabs.sum(0), ref)


# ---- launches of >= 512 tiles of 256x256 (config 5's forward / dgrad sizes).  With the product library these run gemm_kernel over several
# rounds of resident workgroups; with the lab library (MEMEHIP_LIB / MEMEHIP_LIB_F16 -> libmemehip_lab*.so) they also run the 256x256
# eight-phase kernel (lab variant 13, csrc/lab/gemm_lab.inc) -- the tests that found its epilogue-prefetch bug ----------------------
def _variants(T16):
    from multimodal_propaganda_meme_classification_amd import _lib
    lib = _lib.load("bf16" if T16 == torch.bfloat16 else "fp16")
    return lib, ([-2, 13] if hasattr(lib, "mh_gemm_set_variant") else [None])


@BOTH
@pytest.mark.parametrize("layout", [(0, 0), (0, 1)], ids=["fwd", "dgrad"])
@pytest.mark.parametrize("M,N,K", [(4000, 8200, 64), (4000, 8200, 192), (8192, 4096, 128)])
def test_large_launch_16bit_output_is_exact(ops, layout, M, N, K, T16):
    """ragged M, a partial last column tile, odd and even K-tile counts, every 16-bit epilogue operand"""
    assert ((M + 255) // 256) * ((N + 255) // 256) >= 512
    lib, variants = _variants(T16)
    A, B, ref, (lda, ldb) = _operands(layout, M, N, K, T16)
    bias = ints(N, lo=-3, hi=4, seed=3, dtype=F32)
    res = ints(M, N, lo=-8, hi=9, seed=4, dtype=T16)
    mul = ints(M, N, lo=-1, hi=2, seed=5, dtype=T16)
    try:
        for v in variants:
            if v is not None:
                assert lib.mh_gemm_set_variant(v) == 0
            out = torch.full((M, N), 7.0, dtype=T16, device=dev())
            ops.gemm_grouped([ops.Gemm(A, B, out, M, N, K, lda, ldb, N)], False, bool(layout[1]))
            assert torch.equal(out.float(), ref), f"plain store (variant {v})"
            aux = torch.full((M, N), 5.0, dtype=T16, device=dev())
            ops.gemm_grouped([ops.Gemm(A, B, out, M, N, K, lda, ldb, N, bias=bias, residual=res, aux=aux)], False, bool(layout[1]))
            assert torch.equal(aux.float(), ref + bias) and torch.equal(out.float(), ref + bias + res.float()), f"variant {v}"
            ops.gemm_grouped([ops.Gemm(A, B, out, M, N, K, lda, ldb, N, bias=bias, mul=mul, residual=res, deriv_aux=True)], False, bool(layout[1]))
            assert torch.equal(out.float(), (ref + bias) * mul.float() + res.float()), f"variant {v}"
    finally:
        if variants[0] is not None:
            lib.mh_gemm_set_variant(-2)


@BOTH
@pytest.mark.parametrize("layout", [(0, 0), (0, 1)], ids=["fwd", "dgrad"])
def test_large_launch_long_k_grouped_and_packed_rows(ops, layout, T16):
    """Two problems in one launch at K = 2048 (32 K tiles through the LDS stages: a tile restaged early or read before it landed shows
    as a wrong sum), f32 output so the sums stay exact, one problem with a device-side live row count; run several times -- the result
    must be the same bits every time."""
    K, N = 2048, 4096
    M1, M2, live = 8192, 4104, 5000
    lib, variants = _variants(T16)
    A1, B1, ref1, (lda, ldb) = _operands(layout, M1, N, K, T16)
    A2 = ints(M2, K, seed=7, dtype=T16)
    ref2 = A2.float() @ (B1.float().t() if layout == (0, 0) else B1.float())
    o1 = torch.full((M1, N), 3.0, dtype=F32, device=dev())
    o2 = torch.full((M2, N), 3.0, dtype=F32, device=dev())
    rows = torch.tensor([live], dtype=torch.int32, device=dev())
    try:
        for v in variants:
            if v is not None:
                assert lib.mh_gemm_set_variant(v) == 0
            for _ in range(5):
                o1.fill_(3.0)
                o2.fill_(3.0)
                ops.gemm_grouped([ops.Gemm(A1, B1, o1, M1, N, K, lda, ldb, N, rows_dev=rows), ops.Gemm(A2, B1, o2, M2, N, K, lda, ldb, N)],
                                 False, bool(layout[1]))
                assert torch.equal(o1[:live], ref1[:live]), f"variant {v}"
                assert torch.equal(o1[live:], torch.full((M1 - live, N), 3.0, device=dev())), "rows past the live count are not stored"
                assert torch.equal(o2, ref2), f"variant {v}"
    finally:
        if variants[0] is not None:
            lib.mh_gemm_set_variant(-2)
