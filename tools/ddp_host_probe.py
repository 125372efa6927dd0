"""How long does the host take to ISSUE one data-parallel step (1-rank RCCL group) vs the GPU to run it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import multimodal_propaganda_meme_classification_amd as pkg
from multimodal_propaganda_meme_classification_amd import ddp
from bench import synthetic_batch

mode = sys.argv[1] if len(sys.argv) > 1 else "stream"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29777")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")
cfg = pkg.ModelConfig(); cfg.compute_dtype = "fp16"
model = pkg.MultimodalClassifier.from_config(cfg, device=dev, seed=0); model.train()
red = None if mode in ("none", "eager1") else ddp.GradientReducer(model.flat_grads)
opt = pkg.Adam(model.parameters(), lr=2e-5, model=model)
step = pkg.GraphedStep(model, opt, 32, 128, reducer=red, ddp_mode=None if red is None else mode, use_graph=(mode != "eager1"))
step.load_batch(*synthetic_batch(cfg, 32, 128, seed=1234, device=dev))
for _ in range(5):
    step.step()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    step.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"mode {mode}: host issue {1e3 * (t1 - t0) / N:.2f} ms/step, wall {1e3 * (t2 - t0) / N:.2f} ms/step")
dist.destroy_process_group()
