"""The product guard for the hipGraphLaunch fault that follows an RCCL communicator create / destroy cycle (profiles/r04_segfault_record.md),
in a process of its own (it destroys a communicator, which must not happen inside the test suite's process):

    python tools/pg_destroy_guard_check.py

creates a 1-rank RCCL process group, runs a data-parallel GraphedStep on it, tears everything down through ddp.shutdown() (one cycle: still
fine), creates and destroys a SECOND communicator, then builds a
NEW GraphedStep: it must warn (RuntimeWarning), run with eager launches (use_graph False) and give the same losses and parameters as a
graph-replayed step of a fresh process would -- checked here against the first, graphed, phase on the same inputs."""
import os
import sys
import warnings

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_propaganda_meme_classification_amd as pkg  # noqa: E402
from multimodal_propaganda_meme_classification_amd import ddp  # noqa: E402
from oracle import meme_oracle as O  # noqa: E402  (inputs / initial weights only: test tooling)

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
cfg = O.tiny_config("cls")
params = O.init_params(cfg, 31)
b, S, steps = 4, 16, 3


def make():
    m = pkg.MultimodalClassifier.from_config(pkg.ModelConfig.from_dict(cfg.to_dict()), init=False)
    m.load_state_dict(params)
    return m.to("cuda")


def run(step, model):
    losses = []
    for it in range(steps):
        text, image, mask, labels = O.synthetic_batch(cfg, b, S, seed=200 + it)
        step.load_batch(*(t.cuda() for t in (text, image, mask, labels)))
        loss, _ = step.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    return losses, model.flat_params.detach().clone()


# phase 1: one communicator, graphs allowed
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
m1 = make()
red = ddp.GradientReducer(m1.flat_grads)
s1 = pkg.GraphedStep(m1, pkg.Adam(m1.parameters(), lr=1e-3, model=m1), b, S, reducer=red)
assert s1.use_graph and not ddp.communicator_was_destroyed()
l1, p1 = run(s1, m1)
ddp.shutdown()
assert not ddp.communicator_was_destroyed(), "ONE create / destroy cycle is the supported life cycle: graphs stay on"
# a second communicator, created and destroyed: from here on the recorded crash configuration
os.environ["MASTER_PORT"] = str(int(os.environ["MASTER_PORT"]) + 1)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
ddp.note_process_group()
dist.all_reduce(torch.ones(8, device="cuda"))
torch.cuda.synchronize()
ddp.shutdown()
assert ddp.communicator_was_destroyed()
# phase 2: the configuration that faults -- a fresh multi-stream graph after the destroy.  The guard must turn it into eager launches.
m2 = make()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    s2 = pkg.GraphedStep(m2, pkg.Adam(m2.parameters(), lr=1e-3, model=m2), b, S)
assert any(issubclass(x.category, RuntimeWarning) and "process group has been destroyed" in str(x.message) for x in w), [str(x.message) for x in w]
assert s2.use_graph is False
l2, p2 = run(s2, m2)
assert max(abs(a - c) for a, c in zip(l1, l2)) < 1e-6, (l1, l2)
assert torch.equal(p1, p2), float((p1 - p2).abs().max())
print("PG-DESTROY GUARD OK: eager fallback after the destroy, losses", [round(x, 6) for x in l2], "parameters bit-identical to the graphed run")
