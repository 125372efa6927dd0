"""Does a hipGraph capture survive the process group's watchdog thread?  A process of its own (a failing capture aborts it):

    python tools/capture_mode_check.py thread_local|global [repetitions]

1-rank RCCL group; per repetition: eight 64-MB all-reduces are left in flight (work for ProcessGroupNCCL's watchdog to poll with
hipEventQuery), then a data-parallel GraphedStep captures its segment graphs.  In `global` capture-error mode an event query from
ANOTHER thread while a capture is open invalidates the capture (hipErrorStreamCaptureInvalidated -- what killed `bench.py --force-ddp`
in round 3); in `thread_local` mode only the capturing thread's own calls count.  Whether the watchdog polls inside the capture window
is a race: the test (tests/test_model_gpu.py::test_capture_survives_the_process_groups_watchdog) REQUIRES thread_local to pass and
RECORDS what global does."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_propaganda_meme_classification_amd as pkg  # noqa: E402
from multimodal_propaganda_meme_classification_amd import ddp, model as M  # noqa: E402
from oracle import meme_oracle as O  # noqa: E402  (inputs / initial weights only: test tooling)

mode = sys.argv[1] if len(sys.argv) > 1 else "thread_local"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
assert mode in ("thread_local", "global")
M.CAPTURE_MODE = mode
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29578")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
cfg = O.tiny_config("cls")
text, image, mask, labels = O.synthetic_batch(cfg, 4, 16, seed=21)
dev = [t.cuda() for t in (text, image, mask, labels)]
big = torch.ones(1 << 24, device="cuda")
for r in range(reps):
    m = pkg.MultimodalClassifier.from_config(pkg.ModelConfig.from_dict(cfg.to_dict()), init=False)
    m.load_state_dict(O.init_params(cfg, 22 + r))
    m.to("cuda")
    o = pkg.Adam(m.parameters(), lr=2e-5)
    red = ddp.GradientReducer(m.flat_grads, bucket_cap_elems=1 << 16)
    g = pkg.GraphedStep(m, o, 4, 16, reducer=red, ddp_mode="segments")
    works = [dist.all_reduce(big, async_op=True) for _ in range(8)]      # pending work for the watchdog to poll
    g.load_batch(*dev)
    loss, _ = g.step()                                                      # captures the segment graphs now
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    assert float(loss) == float(loss) and float(big[0]) == 1.0
    print(f"capture {r} ok ({mode})", flush=True)
print(f"CAPTURE OK {mode} x{reps}", flush=True)
ddp.shutdown()
