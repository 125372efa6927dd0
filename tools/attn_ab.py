"""Same-box A/B of two builds of the library on the attention forward (boxes differ by +-1.5 %, more than many kernel changes):
both .so files are loaded in ONE process and timed interleaved.   usage: attn_ab.py <libA.so> <libB.so>"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

paths = sys.argv[1:3]
libs = []
for p in paths:
    lib = C.CDLL(os.path.abspath(p))
    lib.mh_attn_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_uint32,
                                C.c_void_p]
    lib.mh_attn_fwd.restype = C.c_int
    lib.mh_attn_bwd.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_uint32, C.c_void_p]
    lib.mh_attn_bwd.restype = C.c_int
    libs.append(lib)
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
for (B, S, H) in ((32, 577, 16), (32, 197, 12), (32, 256, 16)):
    qkv = (torch.randn((B * S, 3 * H * 64), device=dev, generator=g) * 1.0).to(torch.float16)
    outs = [torch.empty((B * S, H * 64), dtype=torch.float16, device=dev) for _ in libs]
    lses = [torch.empty((B, H, S), dtype=torch.float32, device=dev) for _ in libs]
    st = torch.cuda.current_stream().cuda_stream
    res = {0: [], 1: []}
    for rep in range(7):
        for i, lib in enumerate(libs):
            for _ in range(2):
                assert lib.mh_attn_fwd(qkv.data_ptr(), None, outs[i].data_ptr(), lses[i].data_ptr(), B, S, H, None, 0.0, 0, st) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.mh_attn_fwd(qkv.data_ptr(), None, outs[i].data_ptr(), lses[i].data_ptr(), B, S, H, None, 0.0, 0, st)
            e1.record()
            torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) * 100)
    a, b = sorted(res[0])[3], sorted(res[1])[3]
    err = float((outs[0].float() - outs[1].float()).abs().max())
    print(f"B={B} S={S} H={H}: fwd A {a:7.1f} us  B {b:7.1f} us  ({(b / a - 1) * 100:+.1f} %)  max |out_A - out_B| = {err:.2e}  "
          f"lse diff {float((lses[0] - lses[1]).abs().max()):.2e}")
    dout = (torch.randn((B * S, H * 64), device=dev, generator=g) * 0.1).to(torch.float16)
    dq = [torch.zeros_like(qkv) for _ in libs]
    dl = [torch.empty((B, H, S), dtype=torch.float32, device=dev) for _ in libs]
    res = {0: [], 1: []}
    for rep in range(7):
        for i, lib in enumerate(libs):
            args = (qkv.data_ptr(), None, outs[0].data_ptr(), dout.data_ptr(), lses[0].data_ptr(), dl[i].data_ptr(), dq[i].data_ptr(), B, S, H,
                    None, 0.0, 0, st)
            for _ in range(2):
                assert lib.mh_attn_bwd(*args) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.mh_attn_bwd(*args)
            e1.record()
            torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) * 100)
    a, b = sorted(res[0])[3], sorted(res[1])[3]
    print(f"               bwd A {a:7.1f} us  B {b:7.1f} us  ({(b / a - 1) * 100:+.1f} %)  max |dqkv_A - dqkv_B| = "
          f"{float((dq[0].float() - dq[1].float()).abs().max()):.2e}")
