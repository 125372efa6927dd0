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

import weakref
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

# everything that holds RCCL work handles, comm streams or hipGraphs built around them: closed, in order, by shutdown()
_LIVE_REDUCERS: "weakref.WeakSet" = weakref.WeakSet()
_LIVE_STEPS: "weakref.WeakSet" = weakref.WeakSet()
# Communicators this process has torn down / seen.  A hipGraph with more than one stream, captured and launched AFTER an RCCL
# communicator has been created and destroyed in the same process, can fault inside hipGraphLaunch on this stack (ROCm 7.2 / torch
# 2.10): reproduced in rounds 2, 3 and -- with every capture thread-local -- 4 (profiles/r04_segfault_record.md); the cause sits below
# the HIP API and is not known.  The product therefore (1) creates ONE communicator per process and (2) refuses to replay graphs once
# one has been destroyed: GraphedStep falls back to eager launches of the same plan (same results, ~same speed: DESIGN.md section 5).
_DESTROYED = 0
_SEEN_GROUP = False


def note_process_group() -> None:
    global _SEEN_GROUP
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        _SEEN_GROUP = True


_SAFE_CYCLES = 1      # one create / destroy cycle has never faulted (every full GPU suite of rounds 3-4 does exactly that); the recorded crash needs several


def communicator_was_destroyed() -> bool:
    """True once MORE THAN ONE RCCL process group has been destroyed in this process -- by shutdown() or behind this module's back
    (a group a GradientReducer / GraphedStep saw is gone, and another one has been created since).  One communicator per process, torn
    down once at the end, is the supported life cycle and leaves hipGraphs usable; repeated create / destroy cycles are the recorded
    crash configuration (profiles/r04_segfault_record.md)."""
    gone_behind_our_back = 1 if (_SEEN_GROUP and _DESTROYED == 0 and not (dist.is_available() and dist.is_initialized())) else 0
    return _DESTROYED + gone_behind_our_back > _SAFE_CYCLES


class _EventWork:
    """Work handle of a compressed bucket on a HIP device: wait() makes the current stream wait for the comm stream."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)


class GradientReducer:
    """``compress="bf16"``: the gradient slice crosses xGMI as bfloat16 -- half the bytes of the fp32 all-reduce (443 MB
    instead of 887 MB per step for config 3) -- WITHOUT summing in bf16: every rank sends shard r of its
    bf16-rounded slice to rank r (all-to-all), rank r adds the `world` shards in fp32, rounds the sum to bf16 once and
    all-gathers it.  Per element: one bf16 rounding of each rank's contribution + one of the sum (relative 2^-9 each); the
    fp32 buffer receives the exact bf16 value, identical on every rank."""

    def __init__(self, flat_grads: torch.Tensor, group=None, bucket_cap_elems: int = 64 << 20, compress: Optional[str] = None):
        if compress not in (None, "bf16"):
            raise ValueError(f"compress must be None or 'bf16', got {compress!r}")
        self.G = flat_grads
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cap = bucket_cap_elems
        self.compress = compress
        self.pending: List = []
        self.reduced_elems = 0
        self.wire_bytes = 0                  # bytes this rank put on the wire per element-pass (accounting for tests / logs)
        self._cstream = torch.cuda.Stream() if (compress and flat_grads.is_cuda) else None
        self._bufs = {}
        self._had_group = dist.is_initialized()
        note_process_group()
        self.closed = False
        _LIVE_REDUCERS.add(self)

    def _check_open(self):
        if self.closed:
            raise RuntimeError("GradientReducer is closed")
        if self._had_group and not dist.is_initialized():
            raise RuntimeError("GradientReducer outlived its process group: call ddp.shutdown() (or reducer.close()) BEFORE "
                               "torch.distributed.destroy_process_group()")

    def close(self):
        """Finish the pending collectives, drop their work handles, the staging buffers and the comm stream.  Must run while the
        process group is still alive (ddp.shutdown() does it in the right order); idempotent."""
        if self.closed:
            return
        self.wait()
        if self.G.is_cuda:
            torch.cuda.synchronize(self.G.device)
        self._bufs.clear()
        self._cstream = None
        self.closed = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _reduce_compressed(self, a: int, e: int):
        """all-to-all (bf16) -> fp32 sum of the received shards -> bf16 -> all-gather -> fp32 gradient slice."""
        W, n = self.world, e - a
        shard = -(-n // (W * 8)) * 8
        key = shard
        if key not in self._bufs:
            dev = self.G.device
            self._bufs[key] = (torch.zeros(W * shard, dtype=torch.bfloat16, device=dev), torch.empty(W * shard, dtype=torch.bfloat16, device=dev),
                               torch.empty(shard, dtype=torch.bfloat16, device=dev), torch.empty(W * shard, dtype=torch.bfloat16, device=dev))
        send, recv, mine, full = self._bufs[key]
        g = self.G[a:e]

        def body():
            if g.is_cuda:
                from . import ops
                ops.cast_f32_bf16(g, send[:n] if n % 4 == 0 else send)      # HIP cast kernel (n is 4-aligned for layout slices)
            else:
                send[:n].copy_(g)
            dist.all_to_all_single(recv, send, group=self.group)
            if g.is_cuda:
                ops.sum_shards_16(recv, mine, W)                            # fp32 accumulation on receipt, one rounding
            else:
                mine.copy_(recv.view(W, shard).float().sum(dim=0))
            dist.all_gather_into_tensor(full, mine, group=self.group)
            if g.is_cuda:
                from . import ops
                ops.cast_bf16_f32(full[:n], g)
            else:
                g.copy_(full[:n])

        self.wire_bytes += 2 * 2 * (W - 1) * shard
        if self._cstream is None:
            body()
            return _EventWork(None)
        self._cstream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._cstream):
            body()
            ev = torch.cuda.Event()
            ev.record(self._cstream)
        return _EventWork(ev)

    def reduce_range(self, rng: Optional[Tuple[int, int]]):
        """Start the all-reduce (SUM) of G[start:end]; returns immediately with the list of work handles started."""
        started: List = []
        self._check_open()
        if rng is None or not dist.is_initialized():
            return started
        a, b = rng
        if self.compress:      # (also on a 1-rank group: the same collectives and kernels, so one GPU can rehearse the path)
            while a < b:
                e = min(b, a + self.cap)
                started.append(self._reduce_compressed(a, e))
                self.reduced_elems += e - a
                a = e
            self.pending += started
            return started
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
        self._check_open()
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


def shutdown(destroy_process_group: bool = True):
    """Tear the data-parallel machinery down in dependency order: every live GraphedStep (its hipGraphs, then the private pool they
    share, then its side streams), then every live GradientReducer (pending work handles, staging buffers, comm stream), a cyclic
    garbage collection so that nothing created around the communicator is left for a later collection, and only then the process
    group.  Objects that hold work handles or graphs built next to a communicator must not outlive it."""
    import gc
    for st in list(_LIVE_STEPS):
        st.close()
    for red in list(_LIVE_REDUCERS):
        red.close()
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if destroy_process_group and dist.is_initialized():
        global _DESTROYED
        if dist.get_backend() == "nccl":
            _DESTROYED += 1
        dist.destroy_process_group()


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
