"""One-pass attention backward (129..224-token heads) against the two-sweep form, both in ONE process (mh_attn_set_onepass):
time per launch, difference between the two, and each one's error against an fp32 torch reference."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_propaganda_meme_classification_amd import _lib

lib = _lib.load("fp16")
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
for (B, S, H) in ((32, 197, 12), (8, 224, 12), (4, 150, 3), (32, 197, 12)):
    qkv = torch.randn((B * S, 3 * H * 64), device=dev, generator=g).to(torch.float16)
    out = torch.empty((B * S, H * 64), dtype=torch.float16, device=dev)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    assert lib.mh_attn_fwd(qkv.data_ptr(), None, out.data_ptr(), lse.data_ptr(), B, S, H, None, 0.0, 0, st) == 0
    dout = (torch.randn((B * S, H * 64), device=dev, generator=g) * 0.1).to(torch.float16)
    # fp32 reference gradient
    x = qkv.float().view(B, S, 3, H, 64).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(True)
    att = torch.softmax(x[0] @ x[1].transpose(-1, -2) * 0.125, -1) @ x[2]
    att.backward(dout.float().view(B, S, H, 64).permute(0, 2, 1, 3))
    ref = x.grad.permute(1, 3, 0, 2, 4).reshape(B * S, 3 * H * 64)
    res, outs = {}, {}
    for mode in (0, 1):
        lib.mh_attn_set_onepass(mode)
        dq = torch.zeros_like(qkv)
        dl = torch.empty((B, H, S), dtype=torch.float32, device=dev)
        args = (qkv.data_ptr(), None, out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dl.data_ptr(), dq.data_ptr(), B, S, H, None, 0.0, 0, st)
        ts = []
        for rep in range(7):
            for _ in range(2):
                assert lib.mh_attn_bwd(*args) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.mh_attn_bwd(*args)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        res[mode], outs[mode] = sorted(ts)[3], (dq.float(), dl.clone())
    lib.mh_attn_set_onepass(1)
    e0 = float((outs[0][0] - ref).abs().max())
    e1 = float((outs[1][0] - ref).abs().max())
    print(f"B={B} S={S} H={H}: two sweeps {res[0]:7.1f} us  one pass {res[1]:7.1f} us ({(res[1] / res[0] - 1) * 100:+.1f} %)  "
          f"max|d two-one| {float((outs[0][0] - outs[1][0]).abs().max()):.2e}  err vs fp32: two {e0:.2e} one {e1:.2e} (scale {float(ref.abs().max()):.2e})  "
          f"delta equal: {bool(torch.equal(outs[0][1], outs[1][1]))}", flush=True)
