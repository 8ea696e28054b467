"""Asynchronous front of the engine: the API event loop talks to an *engine core loop* through a
pair of message channels ("thread/process boundary: API loop ⇄ engine loop").

Two transports, one protocol:

* ``mode="thread"``  – the core loop runs in a thread of this process (tests, CPU plumbing config);
* ``mode="process"`` – the core loop runs in its own process that owns the GPU, so the per-token
  Python work of the HTTP layer (detokenise, JSON, socket writes for thousands of concurrent
  streams) never competes with the scheduler for the GIL.

Messages to the core:   ("add", req_id, prompt_ids, params_dict, arrival, n) | ("abort", req_id) |
                        ("stats",) | ("shutdown",)
Messages from the core: ("out", [(req_id, token, finished, reason, n_prompt, n_out, n_cached, choice), …]) —
                        ``req_id`` is always the id the client used; ``choice`` numbers the n > 1 samples —
                        one message per engine step — | ("stats", dict) | ("ready", info) | ("dead", err)
"""
from __future__ import annotations

import os

import asyncio
import dataclasses
import multiprocessing as mp
import queue
import threading
import time
import traceback
from typing import AsyncIterator, Optional

from ..config import EngineConfig, SamplingParams


# ------------------------------------------------------------------ core loop (thread or process)
class _QueueChannel:
    """Thread transport with the subset of the mp.Connection API the loops use."""

    def __init__(self, rx: queue.SimpleQueue, tx: queue.SimpleQueue):
        self.rx, self.tx = rx, tx

    def send(self, msg):
        self.tx.put(msg)

    def poll(self, timeout: float = 0.0) -> bool:
        if not self.rx.empty():
            return True
        if timeout <= 0:
            return False
        try:
            item = self.rx.get(timeout=timeout)
        except queue.Empty:
            return False
        self._peek = item
        return True

    def recv(self):
        if getattr(self, "_peek", None) is not None:
            item, self._peek = self._peek, None
            return item
        return self.rx.get()


def _comm_info(engine) -> dict:
    """What the TP group of this engine runs on — the same fields the bench JSON line carries."""
    from ..runtime.tp_launcher import job_info
    ctx = engine.ctx
    info = {"tp": ctx.tp_size, "ep": ctx.ep_size, "device_backend": ctx.device_backend,
            "custom_ar_enabled": bool(ctx.custom_ar is not None and ctx.custom_ar.enabled),
            "pipelined_steps": bool(engine.async_scheduling), "per_kind": dict(ctx.comm)}
    j = job_info()
    for k in ("control_backend", "rccl_ranks_joined", "device_ids", "ranks_share_gpu", "probe", "tp_group_tuning"):
        if k in j:
            info[k] = j[k]
    return info


def engine_core_loop(engine, chan, idle_wait: float = 0.02) -> None:
    """Runs until ("shutdown",): drain commands → one engine step → ship that step's tokens."""
    def stats():
        d = engine.metrics.snapshot()
        d.update(outstanding_tokens=engine.outstanding_tokens(), num_running=engine.num_running,
                 num_waiting=engine.num_waiting, kv_cache_tokens=engine.runner.kv_cache_tokens,
                 graph_steps=engine.runner.stats["graph_steps"], eager_steps=engine.runner.stats["eager_steps"],
                 mixed_graph_steps=engine.runner.stats["mixed_graph_steps"], ar_fallbacks=engine.runner.stats["ar_fallbacks"],
                 selftest=engine.runner.selftest, comm=_comm_info(engine),
                 prometheus=engine.metrics.render_prometheus())
        return d

    running = True
    while running:
        busy = engine.has_unfinished()
        while chan.poll(0.0 if busy else idle_wait):
            msg = chan.recv()
            kind = msg[0]
            if kind == "add":
                _, rid, ids, pd, arrival, n = msg
                try:
                    engine.add_request(rid, prompt_ids=ids, params=SamplingParams(**pd), arrival=arrival, n=n)
                except Exception as e:                       # bad request: report as finished-with-error
                    chan.send(("out", [(rid, None, True, f"error: {e}", 0, 0, 0, c) for c in range(max(n, 1))]))
            elif kind == "abort":
                engine.abort_request(msg[1])
            elif kind == "stats":
                chan.send(("stats", stats()))
            elif kind == "shutdown":
                running = False
                break
            busy = True
        if running and engine.has_unfinished():
            try:
                outs = engine.step()
            except Exception:                                # a failed step must not strand the clients
                chan.send(("dead", traceback.format_exc()))
                raise
            if outs:
                chan.send(("out", [(o.parent_id or o.req_id, o.new_token_ids[0] if o.new_token_ids else None, o.finished,
                                    o.finish_reason, o.num_prompt_tokens, o.num_output_tokens, o.num_cached_tokens,
                                    o.sample_index) for o in outs]))
    engine.shutdown()


def _process_main(cfg: EngineConfig, conn, device_index: int, parent_pid: int = 0):
    try:
        if parent_pid:                                       # never outlive the front end (it owns the clients)
            from ..runtime.tp_launcher import _die_with_parent
            _die_with_parent(parent_pid)
        import torch
        if torch.cuda.is_available() and cfg.device != "cpu":
            torch.cuda.set_device(device_index)
        from .engine import LLMEngine
        from ..runtime.tp_launcher import maybe_launch_tp
        ctx = maybe_launch_tp(cfg, device_index)
        engine = LLMEngine(cfg, ctx)
        engine.warmup()
        conn.send(("ready", {"kv_cache_tokens": engine.runner.kv_cache_tokens, "load_seconds": engine.runner.load_seconds}))
        engine_core_loop(engine, conn)
        from ..runtime.tp_launcher import join_children
        join_children()
    except Exception:
        conn.send(("dead", traceback.format_exc()))
        raise


# ------------------------------------------------------------------ API-side client
@dataclasses.dataclass
class StreamItem:
    token: Optional[int]
    finished: bool
    reason: Optional[str]
    n_prompt: int
    n_out: int
    n_cached: int
    choice: int = 0


class AsyncEngine:
    def __init__(self, cfg: EngineConfig, mode: str = "thread", engine=None, device_index: int = 0):
        self.cfg = cfg
        self.mode = mode
        self._streams: dict[str, asyncio.Queue] = {}
        self._loop: Optional[asyncio.AbstractEventLoop] = None
        self._stats_waiters: list[asyncio.Future] = []
        self._dead: Optional[str] = None
        self.info: dict = {}
        if mode == "thread":
            from .engine import LLMEngine
            self.engine = engine or LLMEngine(cfg)
            a, b = queue.SimpleQueue(), queue.SimpleQueue()
            self._chan = _QueueChannel(rx=b, tx=a)                     # API side: send → a, receive ← b
            core_chan = _QueueChannel(rx=a, tx=b)
            self._core = threading.Thread(target=engine_core_loop, args=(self.engine, core_chan), daemon=True,
                                          name="engine-core")
            self.info = {"kv_cache_tokens": self.engine.runner.kv_cache_tokens}
        elif mode == "process":
            ctx = mp.get_context("spawn")
            parent, child = ctx.Pipe(duplex=True)
            self._chan = parent
            # not a daemon: under TP the core process spawns the worker ranks itself
            self._core = ctx.Process(target=_process_main, args=(cfg, child, device_index, os.getpid()), daemon=False,
                                     name="engine-core")
            self.engine = None
        else:
            raise ValueError(mode)
        self._reader = threading.Thread(target=self._read_loop, daemon=True, name="engine-reader")

    async def start(self, ready_timeout: float = 1800.0):
        self._loop = asyncio.get_running_loop()
        self._core.start()
        if self.mode == "process":
            t0 = time.time()
            while not self._chan.poll(0.2):
                if not self._core.is_alive():
                    raise RuntimeError("engine process died during start-up")
                if time.time() - t0 > ready_timeout:
                    raise TimeoutError("engine process did not become ready")
                await asyncio.sleep(0.05)
            kind, payload = self._chan.recv()
            if kind != "ready":
                raise RuntimeError(f"engine process failed to start:\n{payload}")
            self.info = payload
        self._reader.start()

    def _read_loop(self):
        while True:
            try:
                msg = self._chan.recv()
            except (EOFError, OSError):
                msg = None
            try:
                if msg is None:
                    self._loop.call_soon_threadsafe(self._on_dead, "engine channel closed")
                    return
                if msg[0] == "bye":
                    return
                self._loop.call_soon_threadsafe(self._dispatch, msg)
            except RuntimeError:                             # the server's event loop is gone (shutdown): nobody to tell
                return

    def _on_dead(self, why: str):
        self._dead = why
        for q in self._streams.values():
            q.put_nowait(StreamItem(None, True, f"error: engine {why}", 0, 0, 0))

    def _dispatch(self, msg):
        kind = msg[0]
        if kind == "out":
            for rid, tok, fin, reason, n_p, n_o, n_c, choice in msg[1]:
                q = self._streams.get(rid)
                if q is not None:
                    q.put_nowait(StreamItem(tok, fin, reason, n_p, n_o, n_c, choice))
        elif kind == "stats":
            for f in self._stats_waiters:
                if not f.done():
                    f.set_result(msg[1])
            self._stats_waiters.clear()
        elif kind == "dead":
            self._on_dead(msg[1])

    # ------------------------------------------------------------------ public API
    async def generate(self, req_id: str, prompt_ids, params: SamplingParams, n: int = 1) -> AsyncIterator[StreamItem]:
        """Yields one ``StreamItem`` per generated token (``item.choice`` tells the n > 1 samples apart; the iterator
        ends when all ``n`` have finished); aborts the request if the consumer goes away."""
        if self._dead:
            raise RuntimeError(self._dead)
        q: asyncio.Queue = asyncio.Queue()
        self._streams[req_id] = q
        self._chan.send(("add", req_id, list(map(int, prompt_ids)), dataclasses.asdict(params), time.perf_counter(), int(n)))
        finished = False
        open_choices = max(int(n), 1)
        try:
            while True:
                item = await q.get()
                if item.finished:
                    open_choices -= 1
                finished = open_choices <= 0 or (item.reason or "").startswith("error: engine")
                yield item
                if finished:
                    return
        finally:
            self._streams.pop(req_id, None)
            if not finished:
                self._chan.send(("abort", req_id))

    async def stats(self, timeout: float = 5.0) -> dict:
        fut = self._loop.create_future()
        self._stats_waiters.append(fut)
        self._chan.send(("stats",))
        return await asyncio.wait_for(fut, timeout)

    @property
    def num_streams(self) -> int:
        return len(self._streams)

    async def shutdown(self):
        try:
            self._chan.send(("shutdown",))
        except Exception:
            pass
        if self.mode == "thread":
            self._core.join(timeout=10)
            self._chan.rx.put(("bye",))
        else:
            self._core.join(timeout=20)
            if self._core.is_alive():
                self._core.terminate()
