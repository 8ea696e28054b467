"""Process-group state for tensor / expert parallelism.

One process per GPU; ``torch.distributed`` with backend ``nccl`` (= RCCL over xGMI on ROCm) on GPUs
and ``gloo`` on CPU (tests).  The reference has no collective communication at all;
this module is the new stack's N1 layer.  Collectives used by the model code:

* ``all_reduce``  – row-parallel linear outputs / MoE partial sums.  Small decode-sized messages go
  through the custom one-/two-shot xGMI kernel when it is initialised (``parallel/custom_all_reduce``),
  everything else through RCCL.
* ``all_gather``  – vocab-parallel logits.
* ``all_to_all``  – variable-split row exchange (expert-parallel token dispatch / combine).
* ``broadcast_object`` / ``broadcast_tensor`` – per-step batch metadata from the driver rank.

Every collective is counted (``ParallelContext.comm``: calls and bytes per kind and per path) so that the
benchmark can report the all-reduce calls/bytes per step next to its throughput.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
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
    ep_a2a_min_tokens: int = 0            # MoE steps of at least this many tokens dispatch by all-to-all (0: never)
    # ONE rank of a tp_size-wide group run alone (bench.py --emulate-tp): the model holds exactly TP rank `tp_rank`'s
    # shards (heads, FFN columns, vocabulary slice, experts), and every collective is replaced by a device copy of the
    # same bytes — the rank's GEMM / attention shapes, kernel count and host work per step are the real ones, the link
    # time is not.  Measures what one GPU of an 8-GPU replica has to do, on the one GPU there is.
    emulated: bool = False
    # A TP "group" of ONE rank that still goes through the library for every collective (tp_size == 1): on a one-GPU box
    # this is as much of the RCCL path as can execute — communicator set-up, the stream-ordered calls, their capture in
    # the decode / mixed-step hipGraphs, RCCL on the side stream of the GEMM / all-reduce pipeline, pipelined stepping
    # with an `nccl` device group (tests/test_rccl_gpu.py).  Two ranks on one device are refused by RCCL itself.
    one_rank_group: bool = False
    # Sequence-parallel norms (set per forward pass by the model, see `sp_wanted`): the reductions that follow the
    # row-parallel GEMMs are fused with the residual add + RMSNorm and run on this rank's T / tp rows only
    # (`CustomAllReduce.all_reduce_add_norm`); every rank then carries the residual stream of ITS rows only, so all
    # add-norm sites of one forward pass must take the same path — this flag is that agreement.
    sp_step: bool = False
    sp_ok: bool = os.environ.get("DLI_SP_NORM", "1") != "0"      # cleared by the start-up self-test on a mismatch
    # calls / bytes per collective kind and path since start-up (host-side counts; a hipGraph replay adds the
    # counts recorded at its capture, see ModelRunner)
    comm: dict = field(default_factory=lambda: {"ar_custom_calls": 0, "ar_custom_bytes": 0, "ar_lib_calls": 0,
                                                "ar_lib_bytes": 0, "ag_calls": 0, "ag_bytes": 0, "a2a_calls": 0, "sp_norm_calls": 0,
                                                "a2a_bytes": 0, "bcast_calls": 0, "bcast_bytes": 0})

    @property
    def is_driver(self) -> bool:
        return self.tp_rank == 0

    @property
    def distributed(self) -> bool:
        """More than one process takes part (False for a single rank and for an emulated rank)."""
        return (self.tp_size > 1 or self.one_rank_group) and not self.emulated

    @property
    def solo(self) -> bool:
        """No collective of any kind: one rank, no group."""
        return self.tp_size == 1 and not self.one_rank_group

    @property
    def ep_rank(self) -> int:
        return self.tp_rank if self.ep_size > 1 else 0

    @property
    def device_backend(self) -> str:
        if self.emulated:
            return "emulated"
        if self.solo or self.group is None:
            return "none"
        return dist.get_backend(self.group)

    def sp_wanted(self, T: int, H: int, on_gpu: bool) -> bool:
        """Fused sequence-parallel reduce + add + norm for a forward pass of ``T`` rows?  Needs the custom kernels (the
        GEMMs write their partial sums into the IPC staging buffer), a message in the two-shot regime (one-shot
        all-reduce + a whole-batch norm is faster for decode-sized steps) and rows that fit a workgroup's registers."""
        if not (self.sp_ok and on_gpu and self.tp_size > 1 and H % 8 == 0 and H <= 8192 and T >= self.tp_size):
            return False
        nbytes = T * H * 2
        if self.emulated:
            return nbytes > (256 << 10)
        ar = self.custom_ar
        if ar is None or not ar.enabled or not ar.fits(nbytes):
            return False
        floor = int(os.environ.get("DLI_SP_MIN_BYTES", "0")) or ar.one_shot_max_bytes     # the env value: tests with tiny models
        return nbytes > floor

    def sp_rows(self, T: int):
        """This rank's row range of a sequence-parallel step."""
        per = (T + self.tp_size - 1) // self.tp_size
        lo = min(self.tp_rank * per, T)
        return lo, min(lo + per, T), per

    def _custom_only(self, x: torch.Tensor) -> bool:
        ar = self.custom_ar
        return (ar is not None and ar.enabled and ar.use_max_bytes >= 4096 and x.is_cuda and x.dtype == torch.bfloat16
                and x.is_contiguous() and x.numel() % 8 == 0 and x.data_ptr() % 16 == 0 and self.device_backend != "nccl")

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        if self.solo:
            return x
        c = self.comm
        if self.emulated:
            c["ar_custom_calls"] += 1
            c["ar_custom_bytes"] += x.numel() * x.element_size()
            out = torch.empty_like(x)
            out.copy_(x)                                     # same bytes read + written as the message, no peers
            return out
        if self.custom_ar is not None and self.custom_ar.should_use(x):
            c["ar_custom_calls"] += 1
            c["ar_custom_bytes"] += x.numel() * x.element_size()
            return self.custom_ar.all_reduce(x)
        if self._custom_only(x):
            # no stream-ordered library behind the group (RCCL probe failed / ranks share a GPU): larger messages go
            # through the same kernels in staging-sized pieces, so the step stays free of host-staged collectives
            c["ar_custom_calls"] += 1
            c["ar_custom_bytes"] += x.numel() * x.element_size()
            flat = x.view(-1)
            out = torch.empty_like(flat)
            step = self.custom_ar.use_max_bytes // (2 * 8) * 8
            for lo in range(0, flat.numel(), step):
                out[lo:lo + step] = self.custom_ar.all_reduce(flat[lo:lo + step])
            return out.view(x.shape)
        c["ar_lib_calls"] += 1
        c["ar_lib_bytes"] += x.numel() * x.element_size()
        dist.all_reduce(x, group=self.group)
        return x

    def all_to_all(self, x: torch.Tensor, out_rows: list, in_rows: list) -> torch.Tensor:
        """Row exchange: ``x`` is ``[sum(in_rows), …]`` with ``in_rows[r]`` consecutive rows for TP rank r; returns
        ``[sum(out_rows), …]`` with ``out_rows[r]`` rows received from rank r (``dist.all_to_all_single``: RCCL on
        GPUs).  The split sizes are host integers — the caller has exchanged the counts beforehand."""
        if self.solo:
            return x
        assert not self.emulated, "an emulated rank has no peers to exchange rows with (ep_a2a_min_tokens must be 0)"
        out = torch.empty((int(sum(out_rows)),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        self.comm["a2a_calls"] += 1
        self.comm["a2a_bytes"] += x.numel() * x.element_size()
        if dist.get_backend(self.group) == "gloo" and x.is_cuda:
            raise RuntimeError("device tensors are exchanged by the IPC dispatch kernels (CustomAllReduce.ep_moe), not through gloo")
        dist.all_to_all_single(out, x.contiguous(), [int(n) for n in out_rows], [int(n) for n in in_rows], group=self.group)
        return out

    def all_gather_last_dim(self, x: torch.Tensor) -> torch.Tensor:
        """[..., n] on every rank → [..., tp·n] ordered by rank."""
        if self.solo:
            return x
        lead, n = x.shape[:-1], x.shape[-1]
        x2 = x.reshape(-1, n).contiguous()
        rows = x2.shape[0]
        self.comm["ag_calls"] += 1
        self.comm["ag_bytes"] += x2.numel() * x2.element_size()
        if self.emulated:
            return x2.repeat(1, self.tp_size).reshape(*lead, self.tp_size * n)    # tp copies of the local slice
        if self.custom_ar is not None and self.custom_ar.should_use(x2):
            out = self.custom_ar.all_gather(x2)
            return out.permute(1, 0, 2).reshape(*lead, self.tp_size * n)
        if self._custom_only(x2) and n % 8 == 0:
            out = torch.empty((rows, self.tp_size, n), dtype=x.dtype, device=x.device)
            step = max(self.custom_ar.use_max_bytes // (2 * n), 1)
            for lo in range(0, rows, step):
                out[lo:lo + step] = self.custom_ar.all_gather(x2[lo:lo + step]).permute(1, 0, 2)
            return out.reshape(*lead, self.tp_size * n)
        out = torch.empty((self.tp_size * rows, n), dtype=x.dtype, device=x.device)   # rank-major concat
        dist.all_gather_into_tensor(out, x2, group=self.group)
        return out.view(self.tp_size, rows, n).permute(1, 0, 2).reshape(*lead, self.tp_size * n)

    def all_gather_rows(self, x: torch.Tensor) -> torch.Tensor:
        """``[n, …]`` on every rank → ``[tp·n, …]`` ordered by rank (equal row counts)."""
        if self.solo:
            return x
        x = x.contiguous()
        self.comm["ag_calls"] += 1
        self.comm["ag_bytes"] += x.numel() * x.element_size()
        if self.emulated:
            return x.repeat((self.tp_size,) + (1,) * (x.dim() - 1))
        if self.custom_ar is not None and self.custom_ar.should_use(x):
            return self.custom_ar.all_gather(x).reshape((self.tp_size * x.shape[0],) + tuple(x.shape[1:]))
        out = torch.empty((self.tp_size * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=self.group)
        return out

    def broadcast_tensor(self, x: torch.Tensor, src: int = 0) -> torch.Tensor:
        """In place broadcast from TP rank ``src``: the IPC kernel when it is up (no library call on the step path),
        the device group's broadcast otherwise."""
        if self.distributed:
            self.comm["bcast_calls"] += 1
            self.comm["bcast_bytes"] += x.numel() * x.element_size()
            if self.custom_ar is not None and self.custom_ar.can_broadcast(x):
                self.custom_ar.broadcast(x, src)
                return x
            dist.broadcast(x, src=dist.get_global_rank(self.group, src) if self.group is not None else src,
                           group=self.group)
        return x

    def broadcast_object(self, obj, src: int = 0):
        if not self.distributed:
            return obj
        box = [obj]
        g = self.cpu_group if self.cpu_group is not None else self.group
        dist.broadcast_object_list(box, src=dist.get_global_rank(g, src) if g is not None else src, group=g)
        return box[0]

    def barrier(self):
        if self.distributed:
            dist.barrier(group=self.cpu_group if self.cpu_group is not None else self.group)

    def all_min_int(self, value: int) -> int:
        """Minimum of a host integer over the TP group (KV-cache sizing: every rank must agree on the page count)."""
        if not self.distributed:
            return int(value)
        t = torch.tensor([int(value)], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.cpu_group if self.cpu_group is not None else self.group)
        return int(t[0])

    def disable_custom_ar(self, why: str = "") -> None:
        """Permanent switch to the library collectives (a kernel timed out / a self-test failed)."""
        if self.custom_ar is not None and self.custom_ar.enabled:
            self.custom_ar.enabled = False
            print(f"[dli_amd] tp rank {self.tp_rank}: custom all-reduce switched off ({why}); using {self.device_backend}",
                  flush=True)


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


def drain_collectives() -> None:
    """Call before opening a hipGraph capture: finish the device's work and let every RCCL group's watchdog thread retire the
    collectives it still tracks.

    The watchdog polls the end event of each EAGER collective with hipEventQuery until it has seen it complete, on its own ≈100 ms
    cadence.  Those events sit on the group's internal stream; once that stream has joined a capture (a collective issued inside
    the graph), HIP refuses the query — "operation not permitted on an event last recorded in a capturing stream" — and the
    watchdog terminates the process.  Caught on the one-GPU box (tests/test_rccl_gpu.py), timing-dependent: only when a capture
    opens within the poll interval after an eager warm-up collective.  ``_wait_for_pending_works`` returns when the watchdog's
    list is empty; the works are complete after the synchronize, so it returns within one poll."""
    if not (dist.is_available() and dist.is_initialized() and torch.cuda.is_available()):
        return
    torch.cuda.synchronize()
    from torch.distributed.distributed_c10d import _world
    for pg in list(_world.pg_map):
        try:
            is_rccl = "nccl" in str(dist.get_backend(pg))
        except Exception:
            continue
        if not is_rccl:
            continue
        if hasattr(pg, "_wait_for_pending_works"):
            pg._wait_for_pending_works()
        else:                                                # older torch: out-wait the watchdog's poll interval instead
            import time
            time.sleep(0.3)


def make_tp_context(tp_size: int, ep_size: int = 1, ranks: Optional[list] = None, with_cpu_group: bool = True,
                    custom_all_reduce: bool = False, device: Optional[torch.device] = None,
                    all_groups: Optional[list] = None, device_backend: Optional[str] = None,
                    ep_a2a_min_tokens: int = 0, ar_tuning_rows: Optional[list] = None,
                    one_rank_group: bool = False) -> ParallelContext:
    """TP group over ``ranks`` (default: the whole world).  Must be called by every rank of the world.

    ``all_groups``: every TP group of the job (list of rank lists, ``ranks`` among them) when the world holds several
    replicas — ``dist.new_group`` is collective over the WORLD, so each rank creates all groups in the same order and
    keeps its own.  ``device_backend`` overrides the backend of the device group (``"gloo"`` when the RCCL probe
    failed: the job still runs, through the custom kernels and host-staged collectives)."""
    if tp_size == 1 and not one_rank_group:
        return ParallelContext()
    assert dist.is_initialized(), "call init_distributed() first"
    if ranks is None:
        ranks = list(range(dist.get_world_size())) if not one_rank_group else [dist.get_rank()]
    assert len(ranks) == tp_size, f"TP group needs {tp_size} ranks, got {ranks}"
    assert ep_size in (1, tp_size), "expert parallelism is either off or spans the TP group"
    group = cpu_group = None
    world_backend = dist.get_backend()
    for rs in (all_groups or [ranks]):
        g = dist.new_group(rs, backend=device_backend) if device_backend else dist.new_group(rs)
        cg = None
        if with_cpu_group and (device_backend or world_backend) != "gloo":
            cg = dist.new_group(rs, backend="gloo")
        if list(rs) == list(ranks):
            group, cpu_group = g, cg
    assert group is not None, f"{ranks} is not one of the groups {all_groups}"
    ctx = ParallelContext(tp_size=tp_size, tp_rank=ranks.index(dist.get_rank()), group=group, cpu_group=cpu_group,
                          ep_size=ep_size, ep_a2a_min_tokens=ep_a2a_min_tokens if ep_size > 1 else 0,
                          one_rank_group=one_rank_group and tp_size == 1)
    if custom_all_reduce and tp_size > 1 and device is not None and device.type == "cuda":
        from .custom_all_reduce import CustomAllReduce
        ar = CustomAllReduce.create(ctx, device, tuning_rows=ar_tuning_rows)
        # all or nothing: one rank without the fast path while its peers wait in the kernels' flag barriers is a hang
        if not ctx.all_min_int(1 if ar is not None else 0):
            if ar is not None:
                ar.enabled = False
            ar = None
        ctx.custom_ar = ar
    return ctx
