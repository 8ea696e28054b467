"""Tensor-parallel worker processes: one process per GPU, ``torch.distributed`` over RCCL (``nccl``
backend on ROCm) — gloo on CPU for the tests.

The *driver* (TP rank 0) owns the scheduler, the block manager and the HTTP/engine-core loop; every
other rank runs ``worker_loop``: receive the step header + packed step buffer broadcast by the
driver (``ModelRunner._broadcast_step``), execute the same forward pass on its weight shards, take
part in the all-reduces, sample the same token from the all-gathered logits (same Philox stream), and
wait for the next step.  SURVEY.md §3.6 "broadcast metadata to TP ranks".
"""
from __future__ import annotations

import multiprocessing as mp
import os
import socket
from typing import Optional

import torch

from ..config import EngineConfig
from ..parallel.state import SINGLE, ParallelContext


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_job = None            # this process's launch.Job (what /stats reports: backend, custom all-reduce, probe verdict)


def job_info() -> dict:
    """Communication set-up of this process's TP group, as ``bench.py`` prints it (empty for tp = 1)."""
    if _job is None:
        return {}
    c = dict(_job.comm)
    probe = dict(c.get("probe") or {})
    probe.pop("all_reduce_table", None)
    c["probe"] = probe
    return c


def _join_group(cfg: EngineConfig, rank: int, world: int, port: int, base_index: int):
    """Every TP rank of a server — the engine-core process as rank 0, the spawned workers as ranks 1 … — joins through
    ``launch.bootstrap``: the one bootstrap of the code base (gloo control plane, first-contact probe of RCCL and the
    custom IPC collectives in a throw-away child, RCCL → gloo + custom-kernel fall-back, all-or-nothing custom
    all-reduce, launch-shape table from the probe).  ``bench.py`` ranks take the same path."""
    global _job
    from . import launch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    use_gpu = cfg.device == "cuda" or (cfg.device == "auto" and torch.cuda.device_count() > 0)
    layout = launch.Layout(rank, world, rank, tp=world, ep=cfg.parallel.ep if cfg.parallel.ep > 1 else 1)
    probe_s = float(os.environ.get("DLI_COMM_PROBE_SECONDS", "3"))
    _job = launch.bootstrap(layout, device_kind="cuda" if use_gpu else "cpu", share_gpu=launch.share_gpu_requested(),
                            custom_all_reduce=cfg.parallel.custom_all_reduce and os.environ.get("DLI_NO_CUSTOM_AR") != "1",
                            probe_seconds=probe_s, ep_a2a_min_tokens=cfg.parallel.ep_a2a_min_tokens,
                            device_base=base_index,
                            force_backend="gloo" if cfg.parallel.backend == "gloo" else None)
    return _job.ctx, _job.device


def _die_with_parent(parent_pid: int) -> None:
    """A worker must not outlive its driver: blocked in a collective it would hold its GPU (and the KV cache)
    forever.  PR_SET_PDEATHSIG delivers SIGKILL when the parent exits; a watchdog thread covers the window
    before the prctl call and kernels that re-parent differently."""
    import ctypes
    import signal
    import threading
    import time
    try:
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL)        # PR_SET_PDEATHSIG = 1
    except OSError:
        pass

    def watch():
        while True:
            if os.getppid() != parent_pid:
                os._exit(3)
            time.sleep(1.0)

    threading.Thread(target=watch, daemon=True, name="parent-watchdog").start()


def _maybe_dump_stacks_later() -> None:
    """Debug aid: DLI_DUMP_STACKS_AFTER=<seconds> makes every rank print its Python stacks after that long
    (where is a hung TP group waiting?)."""
    sec = int(os.environ.get("DLI_DUMP_STACKS_AFTER", "0") or 0)
    if sec > 0:
        import faulthandler
        faulthandler.dump_traceback_later(sec, repeat=True)


def _worker_main(cfg: EngineConfig, rank: int, world: int, port: int, base_index: int, parent_pid: int = 0):
    if parent_pid:
        _die_with_parent(parent_pid)
    _maybe_dump_stacks_later()
    from ..engine.engine import LLMEngine
    ctx, device = _join_group(cfg, rank, world, port, base_index)
    engine = LLMEngine(cfg, ctx, device)
    engine.warmup()
    worker_loop(engine)


def worker_loop(engine) -> None:
    while engine.runner.worker_step():
        pass


_children: list = []


def maybe_launch_tp(cfg: EngineConfig, base_index: int = 0) -> Optional[ParallelContext]:
    """In the driver process: spawn ``tp-1`` workers and join the group as rank 0.  Returns the
    driver's ``ParallelContext`` (``SINGLE`` when tp == 1)."""
    tp = cfg.parallel.tp
    if tp <= 1:
        return SINGLE
    _maybe_dump_stacks_later()
    port = _free_port()
    spawn = mp.get_context("spawn")
    for rank in range(1, tp):
        p = spawn.Process(target=_worker_main, args=(cfg, rank, tp, port, base_index, os.getpid()), daemon=True, name=f"tp-worker-{rank}")
        p.start()
        _children.append(p)
    ctx, _device = _join_group(cfg, 0, tp, port, base_index)
    return ctx


def current_device():
    """Device this process's rank was given by the bootstrap (None before / without one)."""
    return None if _job is None else _job.device


def check_children():
    """Failure detection for the TP group: raises if a worker process has exited."""
    for p in _children:
        if not p.is_alive():
            raise RuntimeError(f"tensor-parallel worker {p.name} died (exit code {p.exitcode}); aborting the engine")


def join_children(timeout: float = 20.0):
    for p in _children:
        p.join(timeout)
        if p.is_alive():
            p.terminate()
    _children.clear()
