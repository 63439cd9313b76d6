"""Data parallelism for the temporal train step: replicated model, the trajectory batch split across ranks, the live prefix of the flat
gradient buffer SUM-all-reduced once per step (RCCL through torch.distributed backend "nccl"; gloo for tests) — as ONE collective
(allreduce_flat_gradients), or, from the fused train step, with the slices that are final early in the backward reduced under the rest of it
(OverlappedGradientReduce) — and the 1/world mean folded into the AdamW kernel's grad_scale.  The reference has no distributed code: this is new functionality."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def world_size(group: Optional[dist.ProcessGroup] = None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def rank(group: Optional[dist.ProcessGroup] = None) -> int:
    return dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rank r takes trajectories [r*B/world, (r+1)*B/world) of a global batch (B divisible by world)."""
    B = t.shape[0]
    if B % world != 0:
        raise ValueError(f"global batch {B} is not divisible by world size {world}")
    per = B // world
    return t[rank * per:(rank + 1) * per]


def shard_bounds(B: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of rank r's trajectories in a global batch of ANY size B: the first B % world ranks take one trajectory more (a ragged last batch of an
    epoch — the reference's DataLoader has no drop_last, train/train_temporal.py:84 — still trains on every trajectory; the caller weights the rank's
    mean loss by (hi - lo) * world / B so that the SUM all-reduce times 1 / world is the mean over the global batch)."""
    per, extra = divmod(B, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def rehearse() -> bool:
    """SEA_DP_REHEARSE=1: a process group of ONE rank issues every collective a larger world would (same slices, same order, same streams) instead of
    skipping them — on a one-GPU box this is the only way the RCCL path (backend `nccl`) runs at all: library load, communicator set-up on our
    buffers, the asynchronous slices beside the backward launches, their hand-over to the AdamW launch.  Off: world size 1 issues nothing."""
    return os.environ.get("SEA_DP_REHEARSE", "0") == "1"


def allreduce_flat_gradients(flat_grads: torch.Tensor, n_live: int, group: Optional[dist.ProcessGroup] = None) -> float:
    """SUM-all-reduce the live prefix of the flat gradient buffer in ONE collective; returns the grad_scale (1/world) that turns the
    sum into the mean inside the optimizer kernel."""
    if not dist.is_available() or not dist.is_initialized():
        return 1.0
    world = dist.get_world_size(group)
    if world == 1 and not rehearse():
        return 1.0
    dist.all_reduce(flat_grads[:n_live], op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


class OverlappedGradientReduce:
    """The same SUM all-reduce of grads[:n_live], started early where it can be: `on_bucket(lo, hi)` (called by the backward replay as soon as
    grads[lo:hi] is final) enqueues an asynchronous all-reduce of that slice — RCCL runs it on its own stream behind everything issued so far, beside
    the backward launches that follow — and `finish()` reduces what no bucket covered, waits for the early collectives and returns grad_scale.
    Every element is reduced exactly once by every rank in the same order: ranks stay bit-identical.  SEA_DP_OVERLAP=0 keeps the single collective."""

    def __init__(self, flat_grads: torch.Tensor, n_live: int, group: Optional[dist.ProcessGroup] = None, overlap: Optional[bool] = None):
        self.grads, self.n_live, self.group = flat_grads, n_live, group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # `overlap`: the engine's agreed decision (TemporalEngine.dp_overlap, frozen into the training plan); None (stand-alone use): this rank's switch
        self.live = self.world > 1 or (rehearse() and dist.is_available() and dist.is_initialized())
        self.active = self.live and (os.environ.get("SEA_DP_OVERLAP", "1") != "0" if overlap is None else bool(overlap))
        self.works: List = []
        self.done: List[Tuple[int, int]] = []
        self.calls = 0

    def on_bucket(self, lo: int, hi: int) -> None:
        self.works.append(dist.all_reduce(self.grads[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.done.append((lo, hi))
        self.calls += 1

    def finish(self) -> float:
        if not self.live:
            return 1.0
        pos = 0
        for lo, hi in sorted(self.done) + [(self.n_live, self.n_live)]:
            if lo > pos:
                dist.all_reduce(self.grads[pos:lo], op=dist.ReduceOp.SUM, group=self.group)
                self.calls += 1
            pos = max(pos, hi)
        for w in self.works:
            w.wait()
        return 1.0 / self.world


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, group: Optional[dist.ProcessGroup] = None) -> None:
    """Make every rank start from rank `src`'s parameters (one broadcast of the flat buffer)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)


def parameters_in_sync(flat_params: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> bool:
    """Debug check: the parameter buffers of all ranks are bitwise identical (compares a 64-bit checksum of the raw bits)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return True
    bits = flat_params.view(torch.int32).to(torch.int64)
    idx = torch.arange(1, bits.numel() + 1, device=bits.device, dtype=torch.int64)
    chk = torch.stack(((bits * idx).sum(), bits.sum()))
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))
