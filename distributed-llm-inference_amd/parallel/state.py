"""Process-group state for tensor / expert parallelism.

One process per GPU; ``torch.distributed`` with backend ``nccl`` (= RCCL over xGMI on ROCm) on GPUs
and ``gloo`` on CPU (tests).  The reference has no collective communication at all;
this module is the new stack's N1 layer.  Collectives used by the model code:

* ``all_reduce``  – row-parallel linear outputs / MoE partial sums.  Small decode-sized messages go
  through the custom one-/two-shot xGMI kernel when it is initialised (``parallel/custom_all_reduce``),
  everything else through RCCL.
* ``all_gather``  – vocab-parallel logits.
* ``broadcast_object`` / ``broadcast_tensor`` – per-step batch metadata from the driver rank.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Optional

import torch
import torch.distributed as dist


@dataclass
class ParallelContext:
    tp_size: int = 1
    tp_rank: int = 0
    group: Optional[Any] = None           # torch.distributed process group of the TP ranks
    cpu_group: Optional[Any] = None       # gloo group for object broadcasts
    ep_size: int = 1                      # expert-parallel degree (1 or tp_size)
    custom_ar: Optional[Any] = None       # parallel.custom_all_reduce.CustomAllReduce

    @property
    def is_driver(self) -> bool:
        return self.tp_rank == 0

    @property
    def ep_rank(self) -> int:
        return self.tp_rank if self.ep_size > 1 else 0

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        if self.tp_size == 1:
            return x
        if self.custom_ar is not None and self.custom_ar.should_use(x):
            return self.custom_ar.all_reduce(x)
        dist.all_reduce(x, group=self.group)
        return x

    def all_gather_last_dim(self, x: torch.Tensor) -> torch.Tensor:
        """[..., n] on every rank → [..., tp·n] ordered by rank."""
        if self.tp_size == 1:
            return x
        lead, n = x.shape[:-1], x.shape[-1]
        x2 = x.reshape(-1, n).contiguous()
        rows = x2.shape[0]
        if self.custom_ar is not None and self.custom_ar.should_use(x2):
            out = self.custom_ar.all_gather(x2)
            return out.permute(1, 0, 2).reshape(*lead, self.tp_size * n)
        out = torch.empty((self.tp_size * rows, n), dtype=x.dtype, device=x.device)   # rank-major concat
        dist.all_gather_into_tensor(out, x2, group=self.group)
        return out.view(self.tp_size, rows, n).permute(1, 0, 2).reshape(*lead, self.tp_size * n)

    def broadcast_tensor(self, x: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.tp_size > 1:
            dist.broadcast(x, src=dist.get_global_rank(self.group, src) if self.group is not None else src,
                           group=self.group)
        return x

    def broadcast_object(self, obj, src: int = 0):
        if self.tp_size == 1:
            return obj
        box = [obj]
        g = self.cpu_group if self.cpu_group is not None else self.group
        dist.broadcast_object_list(box, src=dist.get_global_rank(g, src) if g is not None else src, group=g)
        return box[0]

    def barrier(self):
        if self.tp_size > 1:
            dist.barrier(group=self.cpu_group if self.cpu_group is not None else self.group)


SINGLE = ParallelContext()


def init_distributed(backend: str = "auto", device: Optional[torch.device] = None) -> None:
    """Initialise the default process group from the torchrun environment (idempotent)."""
    if dist.is_initialized():
        return
    if backend == "auto":
        backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)


def make_tp_context(tp_size: int, ep_size: int = 1, ranks: Optional[list] = None, with_cpu_group: bool = True,
                    custom_all_reduce: bool = False, device: Optional[torch.device] = None) -> ParallelContext:
    """TP group over ``ranks`` (default: the whole world).  Must be called by every rank of the world."""
    if tp_size == 1:
        return ParallelContext()
    assert dist.is_initialized(), "call init_distributed() first"
    if ranks is None:
        ranks = list(range(dist.get_world_size()))
    assert len(ranks) == tp_size, f"TP group needs {tp_size} ranks, got {ranks}"
    assert ep_size in (1, tp_size), "expert parallelism is either off or spans the TP group"
    group = dist.new_group(ranks)
    cpu_group = None
    if with_cpu_group and dist.get_backend() != "gloo":
        cpu_group = dist.new_group(ranks, backend="gloo")
    ctx = ParallelContext(tp_size=tp_size, tp_rank=ranks.index(dist.get_rank()), group=group, cpu_group=cpu_group,
                          ep_size=ep_size)
    if custom_all_reduce and device is not None and device.type == "cuda":
        from .custom_all_reduce import CustomAllReduce
        ctx.custom_ar = CustomAllReduce.create(ctx, device)
    return ctx
