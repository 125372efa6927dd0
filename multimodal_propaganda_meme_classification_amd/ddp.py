"""Data-parallel replicas: one process per GPU, gradients summed with RCCL all-reduce over xGMI.

The reference has no distributed code; BASELINE configs 4-5 ask for DDP.
Each meme is independent in forward/backward, so the only exchange is ONE sum of the flat
gradient buffer per step.  The flat layout puts the big matrices first in backward-completion
order, so a layer pair's 57 MB gradient slice is all-reduced (async, on RCCL's stream) while the
next layer's backward runs; the optimizer folds the 1/world_size into its fused update
(Adam.grad_scale).  Works with any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in
the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class GradientReducer:
    def __init__(self, flat_grads: torch.Tensor, group=None, bucket_cap_elems: int = 64 << 20):
        self.G = flat_grads
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cap = bucket_cap_elems
        self.pending: List = []
        self.reduced_elems = 0

    def reduce_range(self, rng: Optional[Tuple[int, int]]):
        """Start the all-reduce (SUM) of G[start:end]; returns immediately with the list of work handles started."""
        started: List = []
        if rng is None or not dist.is_initialized():
            return started
        a, b = rng
        while a < b:
            e = min(b, a + self.cap)
            started.append(dist.all_reduce(self.G[a:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.reduced_elems += e - a
            a = e
        self.pending += started
        return started

    def hook(self, seg_name: str, rng: Optional[Tuple[int, int]]):
        self.reduce_range(rng)

    def gather(self, pairs):
        """all-gather (local, gathered) tensor pairs in rank order (the embedding-gradient exchange)."""
        if not dist.is_initialized():
            for local, full in pairs:
                full.copy_(local.reshape(full.shape))
            return
        for local, full in pairs:
            dist.all_gather_into_tensor(full.view(-1), local.contiguous().view(-1), group=self.group)

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending.clear()

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)


def check_bucket_cover(plan_bucket_after: dict, n_total: int) -> int:
    """The per-segment gradient ranges must tile [0, end) exactly once; returns `end` (= n_total, or the start of
    the embedding tables when their gradients are exchanged as gathered rows instead)."""
    rngs = sorted(plan_bucket_after.values())
    pos = 0
    for a, b in rngs:
        if a != pos:
            raise AssertionError(f"gradient buckets leave a gap/overlap at {pos} (next starts at {a})")
        pos = b
    if pos > n_total:
        raise AssertionError(f"gradient buckets end at {pos} > {n_total}")
    return pos
