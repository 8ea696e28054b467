"""Multi-rank job bootstrap: one process per GPU, shared by ``bench.py`` and the tests.

Two ways in, one layout out:

* **self-launch** – ``python bench.py --gpus N`` without a torchrun environment: the parent never touches a GPU
  (no HIP call, not even ``torch.cuda.is_available()``); it starts N fresh child ranks with ``RANK`` /
  ``LOCAL_RANK`` / ``WORLD_SIZE`` / ``MASTER_*`` set, forwards their output, and exits non-zero as soon as any
  child fails (after ending the others) — :func:`self_launch`;
* **torchrun** – the driver's ``python -m torch.distributed.run --nproc-per-node N … bench.py --gpus N``: the ranks
  already exist, the environment is read as is.

Rank layout (:class:`Layout`): ``world = gpus``; ``tp`` consecutive ranks form one tensor-parallel replica,
``dp = gpus // tp`` replicas; TP rank 0 of every replica is its *driver* (scheduler, block manager, request
replay), the others run ``worker_loop``.  Control plane: a gloo world group (barriers, stats, the probe verdict).
Device plane: per replica, RCCL (``nccl``) if the first-contact probe (``parallel/comm_probe.py``) proved it,
else gloo — plus the custom IPC collectives when their self-test passes.  The reference has no multi-process
launcher (its only process is the load generator, `/root/reference/traffic_generator/main.py:315-343`).
"""
from __future__ import annotations

import datetime
import os
import signal
import socket
import subprocess
import sys
import time
from dataclasses import dataclass, field
from typing import Optional


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launched_by_torchrun() -> bool:
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def self_launch(n: int, script: str, argv: list, extra_env: Optional[dict] = None, poll: float = 0.2) -> int:
    """Start ``n`` ranks of ``script`` and wait.  Returns the job's exit code: 0 only if every rank returned 0.
    Must be called before anything in this process initialises the GPU (children are fresh interpreters, so the
    parent's state never leaks into them either way)."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DLI_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or 8) // max(n, 1)))))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))

    def end_all(sig):
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)
                except OSError:
                    pass

    code = 0
    try:
        while True:
            states = [p.poll() for p in procs]
            bad = [c for c in states if c not in (None, 0)]
            if bad:
                code = bad[0] if bad[0] > 0 else 128 - bad[0]
                print(f"[launch] a rank exited with code {bad[0]}; ending the other ranks", file=sys.stderr, flush=True)
                end_all(signal.SIGTERM)
                t0 = time.time()
                while any(p.poll() is None for p in procs) and time.time() - t0 < 10:
                    time.sleep(0.1)
                end_all(signal.SIGKILL)
                break
            if all(c == 0 for c in states):
                break
            time.sleep(poll)
    except KeyboardInterrupt:
        end_all(signal.SIGTERM)
        code = 130
    for p in procs:
        try:
            p.wait(timeout=5)
        except subprocess.TimeoutExpired:
            p.kill()
    return code


@dataclass
class Layout:
    rank: int
    world: int
    local_rank: int
    tp: int = 1
    ep: int = 1

    @property
    def dp(self) -> int:
        return self.world // self.tp

    @property
    def replica(self) -> int:
        return self.rank // self.tp

    @property
    def tp_rank(self) -> int:
        return self.rank % self.tp

    @property
    def is_driver(self) -> bool:
        return self.tp_rank == 0

    @property
    def groups(self) -> list:
        return [list(range(i * self.tp, (i + 1) * self.tp)) for i in range(self.dp)]

    @property
    def driver_ranks(self) -> list:
        return [i * self.tp for i in range(self.dp)]

    @classmethod
    def from_env(cls, gpus: int, tp: int = 1, ep: int = 1) -> "Layout":
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", str(rank)))
        if world != gpus:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}: one rank per GPU")
        if tp < 1 or gpus % tp:
            raise SystemExit(f"--tp {tp} must divide --gpus {gpus}")
        if ep not in (1, tp):
            raise SystemExit(f"--ep {ep}: expert parallelism is either off (1) or spans the TP group ({tp})")
        return cls(rank, world, local, tp, ep)


@dataclass
class Job:
    """What :func:`bootstrap` hands to the caller."""
    layout: Layout
    device: object                       # torch.device
    ctx: object                          # ParallelContext of this rank's replica
    world_group: object = None           # gloo, all ranks (None when world == 1)
    driver_group: object = None          # gloo, the replicas' drivers (None when there is one)
    comm: dict = field(default_factory=dict)   # probe verdict + all-reduce table + what is in use


def share_gpu_requested() -> bool:
    """One switch for every launcher (bench.py, api_server --tp, the tests): ranks may share a device for a
    functional rehearsal on a box with fewer GPUs than ranks.  (The two older names are still honoured.)"""
    return any(os.environ.get(k) == "1" for k in ("DLI_SHARE_GPU", "DLI_BENCH_SHARE_GPU", "DLI_TP_SHARE_GPU")) \
        or os.environ.get("DLI_BENCH_BACKEND") == "gloo"


def bootstrap(layout: Layout, device_kind: str = "cuda", share_gpu: bool = False, custom_all_reduce: bool = True,
              probe_seconds: float = 5.0, probe_timeout: float = 150.0, ep_a2a_min_tokens: int = 0,
              group_timeout_s: float = 900.0, device_base: int = 0, force_backend: Optional[str] = None) -> Job:
    """Join the job: pick this rank's device, build the control plane, probe the device collectives in a child
    process, build the TP context on what the probe proved.  Every rank calls this with the same arguments.
    THE bootstrap of every multi-rank entry point: ``bench.py`` ranks and the TP workers of ``api_server --tp``
    (``runtime/tp_launcher.py``) both come through here.  ``device_base``: first device of this job (a DP replica of the
    router owns devices ``base … base + tp - 1``); ``force_backend="gloo"`` skips RCCL (CPU-style rehearsals)."""
    import torch
    import torch.distributed as dist
    from ..parallel.comm_probe import probe_in_subprocess
    from ..parallel.state import SINGLE, make_tp_context

    on_gpu = device_kind == "cuda"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if on_gpu:
        n_dev = torch.cuda.device_count()                    # does not initialise the GPU on this image
        if n_dev < 1:
            raise SystemExit("no GPU visible")
        if n_dev < device_base + layout.world and not share_gpu:
            raise SystemExit(f"{layout.world} ranks from device {device_base} but {n_dev} GPU(s) visible (DLI_SHARE_GPU=1 lets "
                             "ranks share a device for a functional rehearsal)")
        share_gpu = share_gpu and n_dev < device_base + layout.world
        dev_index = (device_base + layout.local_rank) % n_dev
        device = torch.device("cuda", dev_index)
    else:
        dev_index, device, share_gpu = -1, torch.device("cpu"), False
    job = Job(layout, device, SINGLE)
    comm = job.comm
    comm.update(control_backend="none", device_backend="none", device_ids=[dev_index], rccl_ranks_joined=0,
                custom_ar_enabled=False, ranks_share_gpu=bool(share_gpu))
    if layout.world == 1:
        if on_gpu:
            torch.cuda.set_device(device)
        return job

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=layout.rank, world_size=layout.world,
                            timeout=datetime.timedelta(seconds=group_timeout_s))
    job.world_group = dist.group.WORLD
    comm["control_backend"] = "gloo"
    ids = [None] * layout.world
    dist.all_gather_object(ids, dev_index)
    comm["device_ids"] = ids

    probe = {"rccl_ok": False, "custom_ar_ok": False, "rccl_ranks_joined": 0, "all_reduce_table": []}
    if on_gpu and probe_seconds >= 0:
        port_box = [free_port() if layout.rank == 0 else 0]
        dist.broadcast_object_list(port_box, src=0)
        # the probe touches the GPU before this process does; it has exited by the time the engine is built
        mine = probe_in_subprocess(layout.rank, layout.world, dev_index, port_box[0], share_gpu=share_gpu,
                                   seconds=probe_seconds, timeout=probe_timeout)
        allp = [None] * layout.world
        dist.all_gather_object(allp, mine)
        probe = dict(allp[0])
        probe["rccl_ok"] = all(p.get("rccl_ok") for p in allp) and force_backend in (None, "nccl")
        probe["custom_ar_ok"] = all(p.get("custom_ar_ok") for p in allp)
        probe["rccl_ranks_joined"] = layout.world if probe["rccl_ok"] else 0
        errs = {str(i): {k: v for k, v in p.items() if k.endswith("error")} for i, p in enumerate(allp)}
        errs = {k: v for k, v in errs.items() if v}
        if errs:
            probe["errors_by_rank"] = errs
        probe.pop("rank", None)
        probe.pop("device", None)
    comm["probe"] = probe
    comm["rccl_ranks_joined"] = probe["rccl_ranks_joined"]

    if on_gpu:
        torch.cuda.set_device(device)
    if layout.dp > 1:
        job.driver_group = dist.new_group(layout.driver_ranks, backend="gloo")
    if layout.tp > 1:
        if on_gpu:
            backend = "nccl" if probe["rccl_ok"] else "gloo"
            want_custom = custom_all_reduce and probe["custom_ar_ok"]
        else:
            backend, want_custom = "gloo", False
        my_group = layout.groups[layout.replica]
        try:
            # the probe child of TP rank 0 of THIS replica swept the launch shapes of the same peers: its table is
            # installed instead of a second sweep (a world-wide probe over several replicas measured other group sizes)
            rows = probe.get("tuning_rows") if (layout.dp == 1 and os.environ.get("DLI_AR_RETUNE") != "1") else None
            ctx = make_tp_context(layout.tp, ep_size=layout.ep, ranks=my_group, all_groups=layout.groups,
                                  custom_all_reduce=want_custom, device=device, device_backend=backend,
                                  ep_a2a_min_tokens=ep_a2a_min_tokens, ar_tuning_rows=rows)
        except Exception as e:
            raise SystemExit(f"rank {layout.rank}: could not build the TP group over {backend}: {type(e).__name__}: {e}")
        job.ctx = ctx
        probe.pop("tuning_rows", None)                       # installed (or superseded by the group's own sweep)
        comm["device_backend"] = {"nccl": "rccl"}.get(backend, backend)
        comm["custom_ar_enabled"] = bool(ctx.custom_ar is not None and ctx.custom_ar.enabled)
        if ctx.custom_ar is not None and ctx.custom_ar.tuning:
            t = ctx.custom_ar.tuning
            comm["tp_group_tuning"] = {"source": t.get("source"), "use_custom_up_to_bytes": t["use_custom_up_to_bytes"],
                                       "one_shot_up_to_bytes": t["one_shot_up_to_bytes"],
                                       "sweep_seconds": t["sweep_seconds"],
                                       "best": [{"bytes": r["bytes"], **r["best"], "library_us": r.get("library_us")}
                                                for r in t["rows"]]}
    else:
        # data-parallel replicas exchange no tensors: the control plane is all they share
        comm["device_backend"] = "rccl (probe only; DP replicas exchange no tensors)" if probe["rccl_ok"] else "none"
    return job


def shutdown(job: Job) -> None:
    import torch.distributed as dist
    if job.layout.world > 1 and dist.is_initialized():
        try:
            dist.barrier(group=job.world_group)
        finally:
            dist.destroy_process_group()
