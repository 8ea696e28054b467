"""Data-parallel request router (BASELINE.json config 3: "Llama-3 8B DP=8 replicas
behind request router, conversations.json multi-turn replay").

A streaming reverse proxy in front of N independent engine replicas (one process per GPU, each a
full ``api_server``).  The harness keeps talking to ONE url; the router picks a replica per request:

* **least outstanding tokens** – every in-flight request is charged ``prompt tokens + max_tokens``
  (released as it completes), so a replica digesting a burst of long prompts is avoided even if its
  request count is low;
* **prefix / session affinity** – the prompt is cut into 256-character blocks with chained hashes;
  the replica that most recently served the longest matching chain holds those KV pages in its prefix
  cache (multi-turn prompts extend the previous turn), so the request goes there unless that replica
  is overloaded relative to the least-loaded one (``affinity_slack``).  ``X-Session-Id`` pins a
  session the same way;
* **health** – a background probe marks replicas down after consecutive failures and routes around
  them; a request that fails before any byte was relayed is retried on another replica, and so is a
  request that a replica sheds with ``503`` (its admission queue is full); when every replica sheds, the
  client gets the ``503`` + ``Retry-After`` (the harness records it as a failed request through
  ``raise_for_status``, `/root/reference/traffic_generator/main.py:258,269-270`);
* **several router processes** (``--workers N``, one port, ``SO_REUSEPORT``) share ONE load table: the
  outstanding-token and in-flight counters of every replica live in shared memory (``SharedLoad``), so each
  worker balances against the traffic of all of them, not just its own.

The wire format is relayed untouched (NDJSON or SSE lines are forwarded as they arrive), so
``log.json`` semantics are identical with or without the router.

    python -m dli_amd.router.router --replicas http://127.0.0.1:8001,http://127.0.0.1:8002 --port 11434
    python -m dli_amd.router.router --spawn 8 --model llama3-8b --port 11434      # one replica per GPU
"""
from __future__ import annotations

import argparse
import asyncio
import json
import os
import subprocess
import sys
import time
import zlib
from collections import OrderedDict
from typing import Optional

import aiohttp
from aiohttp import web

BLOCK_CHARS = 256
MAX_BLOCKS = 64


def prefix_chain(prompt: str) -> list[int]:
    """Chained hashes of the 256-character blocks of ``prompt`` (whole blocks only)."""
    out, h = [], 0
    data = prompt.encode("utf-8", "ignore")
    for i in range(min(len(data) // BLOCK_CHARS, MAX_BLOCKS)):
        h = zlib.crc32(data[i * BLOCK_CHARS:(i + 1) * BLOCK_CHARS], h) ^ (i * 0x9E3779B1 & 0xFFFFFFFF)
        out.append((h << 8) | (i & 0xFF))
    return out


class SharedLoad:
    """Per-replica (outstanding tokens, in-flight requests, served) counters in shared memory, created before the
    router workers fork.  Updates take the array's lock for a few hundred nanoseconds; reads are lock-free (a torn
    read of two counters only perturbs one routing decision)."""

    def __init__(self, n_replicas: int):
        import multiprocessing as mp
        self.n = n_replicas
        self.arr = mp.get_context("fork").Array("q", 3 * n_replicas, lock=True)

    def add(self, idx: int, tokens: int, requests: int, served: int = 0):
        with self.arr.get_lock():
            self.arr[3 * idx] += tokens
            self.arr[3 * idx + 1] += requests
            self.arr[3 * idx + 2] += served

    def get(self, idx: int):
        a = self.arr
        return a[3 * idx], a[3 * idx + 1], a[3 * idx + 2]


class Replica:
    def __init__(self, url: str, index: int = 0, shared: Optional[SharedLoad] = None):
        self.url = url.rstrip("/")
        self.index = index
        self.shared = shared                 # None: this process is the only router
        self._tokens = 0
        self._inflight = 0
        self._served = 0
        self.healthy = True
        self.fails = 0
        self.errors = 0

    # the load figures every routing decision uses: the totals over ALL router workers when they share a table
    @property
    def outstanding_tokens(self) -> int:
        return self.shared.get(self.index)[0] if self.shared is not None else self._tokens

    @outstanding_tokens.setter
    def outstanding_tokens(self, v: int):                      # unit tests poke the load directly
        self._tokens = v

    @property
    def inflight(self) -> int:
        return self.shared.get(self.index)[1] if self.shared is not None else self._inflight

    @inflight.setter
    def inflight(self, v: int):
        self._inflight = v

    @property
    def served(self) -> int:
        return self.shared.get(self.index)[2] if self.shared is not None else self._served

    def acquire(self, cost: int):
        self._tokens += cost
        self._inflight += 1
        if self.shared is not None:
            self.shared.add(self.index, cost, 1)

    def release(self, cost: int, served: bool = False):
        self._tokens -= cost
        self._inflight -= 1
        self._served += int(served)
        if self.shared is not None:
            self.shared.add(self.index, -cost, -1, int(served))

    def load(self) -> float:
        return self.outstanding_tokens + 64 * self.inflight


class RoutingPolicy:
    """Pure decision logic (unit-testable without sockets)."""

    def __init__(self, replicas: list[Replica], affinity_slack: float = 0.5, slack_tokens: int = 4096,
                 table_size: int = 200_000, stateless_affinity: bool = False):
        self.replicas = replicas
        # several router workers (``--workers``) do not share this table: with ``stateless_affinity`` a prompt or
        # session unknown to THIS worker goes to the replica every worker would pick (rendezvous hash of its first
        # prefix block / session id over the live replicas), so that follow-up turns still find their KV pages
        self.stateless_affinity = stateless_affinity
        self.affinity_slack = affinity_slack
        self.slack_tokens = slack_tokens
        self.table: OrderedDict[int, int] = OrderedDict()      # prefix-chain hash / session key → replica index
        self.table_size = table_size
        self.affinity_hits = 0
        self.decisions = 0

    def _remember(self, key: int, idx: int):
        self.table[key] = idx
        self.table.move_to_end(key)
        while len(self.table) > self.table_size:
            self.table.popitem(last=False)

    def choose(self, chain: list[int], session: Optional[str] = None, exclude: frozenset = frozenset()) -> Optional[int]:
        alive = [i for i, r in enumerate(self.replicas) if r.healthy and i not in exclude]
        if not alive:
            return None
        self.decisions += 1
        best = min(alive, key=lambda i: (self.replicas[i].load(), i))
        cand = None
        if session is not None:
            cand = self.table.get(zlib.crc32(session.encode()) | (1 << 60))
        if cand is None:
            for key in reversed(chain):                             # longest cached prefix first
                cand = self.table.get(key)
                if cand is not None:
                    break
        if cand is None and self.stateless_affinity and (session is not None or chain):
            key = zlib.crc32(session.encode()) if session is not None else chain[0]
            cand = max(alive, key=lambda i: zlib.crc32(i.to_bytes(2, "little"), key & 0xFFFFFFFF))
        if cand is not None and cand in alive:
            lb, lc = self.replicas[best].load(), self.replicas[cand].load()
            if lc <= lb * (1.0 + self.affinity_slack) + self.slack_tokens:
                self.affinity_hits += 1
                best = cand
        return best

    def record(self, idx: int, chain: list[int], session: Optional[str] = None):
        for key in chain:
            self._remember(key, idx)
        if session is not None:
            self._remember(zlib.crc32(session.encode()) | (1 << 60), idx)


class Router:
    def __init__(self, replica_urls: list[str], health_interval: float = 2.0, max_fails: int = 2, retries: int = 2,
                 chars_per_token: float = 1.0, shared: Optional[SharedLoad] = None, **policy_kw):
        self.replicas = [Replica(u, i, shared) for i, u in enumerate(replica_urls)]
        self.shed = 0
        self.policy = RoutingPolicy(self.replicas, **policy_kw)
        self.health_interval = health_interval
        self.max_fails = max_fails
        self.retries = retries
        self.chars_per_token = chars_per_token
        self.session: Optional[aiohttp.ClientSession] = None
        self._health_task = None
        self.started = time.time()

    # ------------------------------------------------------------------ lifecycle
    async def start(self):
        conn = aiohttp.TCPConnector(limit=0, keepalive_timeout=30)
        self.session = aiohttp.ClientSession(connector=conn, timeout=aiohttp.ClientTimeout(total=None, sock_connect=5))
        self._health_task = asyncio.create_task(self._health_loop())

    async def stop(self):
        if self._health_task:
            self._health_task.cancel()
        if self.session:
            await self.session.close()

    async def _probe(self, r: Replica):
        try:
            async with self.session.get(r.url + "/health", timeout=aiohttp.ClientTimeout(total=2)) as resp:
                ok = resp.status == 200
        except Exception:
            ok = False
        if ok:
            r.fails, r.healthy = 0, True
        else:
            r.fails += 1
            if r.fails >= self.max_fails:
                r.healthy = False

    async def _health_loop(self):
        while True:
            await asyncio.gather(*(self._probe(r) for r in self.replicas))
            await asyncio.sleep(self.health_interval)

    # ------------------------------------------------------------------ proxying
    def _cost(self, body: dict) -> int:
        prompt = body.get("prompt")
        if prompt is None and isinstance(body.get("messages"), list):
            prompt = "".join(str(m.get("content", "")) for m in body["messages"])
        n_prompt = int(len(prompt) / self.chars_per_token) if isinstance(prompt, str) else len(prompt or ())
        opt = body.get("options") or {}
        max_tokens = body.get("max_tokens") or opt.get("num_predict") or 128
        return n_prompt + int(max_tokens)

    async def proxy(self, request: web.Request):
        raw = await request.read()
        try:
            body = json.loads(raw)
            assert isinstance(body, dict)
        except Exception:
            return web.json_response({"error": "invalid JSON body"}, status=400)
        prompt = body.get("prompt") if isinstance(body.get("prompt"), str) else \
            "".join(str(m.get("content", "")) for m in body.get("messages", []) if isinstance(m, dict))
        chain = prefix_chain(prompt)
        session_id = request.headers.get("X-Session-Id")
        cost = self._cost(body)
        tried: set[int] = set()
        overloaded = None
        for _attempt in range(max(self.retries + 1, 1)):
            idx = self.policy.choose(chain, session_id, frozenset(tried))
            if idx is None:
                break
            tried.add(idx)
            rep = self.replicas[idx]
            rep.acquire(cost)
            resp: Optional[web.StreamResponse] = None
            done = False
            try:
                async with self.session.post(rep.url + request.path, data=raw,
                                             headers={"Content-Type": "application/json"}) as up:
                    if up.status == 503:                              # the replica shed it: try the next one
                        overloaded = (await up.read(), up.headers.get("Retry-After", "1"))
                        continue
                    resp = web.StreamResponse(status=up.status,
                                              headers={"Content-Type": up.headers.get("Content-Type", "application/json"),
                                                       "X-Served-By": str(idx)})
                    await resp.prepare(request)
                    self.policy.record(idx, chain, session_id)
                    try:
                        async for chunk in up.content.iter_any():
                            await resp.write(chunk)
                        await resp.write_eof()
                    except (ConnectionResetError, aiohttp.ClientConnectionResetError):
                        # the CLIENT went away: leaving the `async with` closes the upstream connection, which aborts
                        # the request on the replica (frees its KV pages) — not a replica failure
                        if up.content.exception() is None:
                            return resp
                        raise
                    done = True
                    return resp
            except (aiohttp.ClientConnectionError, asyncio.TimeoutError, aiohttp.ClientPayloadError) as e:
                rep.errors += 1
                rep.fails += 1
                if rep.fails >= self.max_fails:
                    rep.healthy = False
                if resp is not None:                                 # bytes already relayed: cannot retry —
                    if request.transport is not None:                #   break the client stream so it sees a failure
                        request.transport.close()
                    return resp
            finally:
                rep.release(cost, served=done)
        self.shed += 1
        if overloaded is not None:                                   # every replica tried is shedding load
            return web.Response(body=overloaded[0], status=503, content_type="application/json",
                                headers={"Retry-After": overloaded[1]})
        return web.json_response({"error": "no healthy replica"}, status=503, headers={"Retry-After": "1"})

    # ------------------------------------------------------------------ introspection
    async def health(self, request):
        up = sum(r.healthy for r in self.replicas)
        return web.json_response({"status": "ok" if up else "down", "replicas_up": up, "replicas": len(self.replicas)},
                                 status=200 if up else 503)

    def stats_dict(self) -> dict:
        return {"decisions": self.policy.decisions, "affinity_hits": self.policy.affinity_hits, "shed": self.shed,
                "pid": os.getpid(), "shared_load_table": self.replicas[0].shared is not None,
                "replicas": [{"url": r.url, "healthy": r.healthy, "inflight": r.inflight,
                              "outstanding_tokens": r.outstanding_tokens, "served": r.served, "errors": r.errors}
                             for r in self.replicas]}

    async def stats(self, request):
        return web.json_response(self.stats_dict())

    async def metrics(self, request):
        """Router gauges + every replica's Prometheus text, labelled by replica index."""
        lines = ["# TYPE dli_router_decisions_total counter", f"dli_router_decisions_total {self.policy.decisions}",
                 "# TYPE dli_router_affinity_hits_total counter", f"dli_router_affinity_hits_total {self.policy.affinity_hits}"]
        for i, r in enumerate(self.replicas):
            lines.append(f'dli_router_replica_healthy{{replica="{i}"}} {int(r.healthy)}')
            lines.append(f'dli_router_replica_outstanding_tokens{{replica="{i}"}} {r.outstanding_tokens}')
            lines.append(f'dli_router_replica_inflight{{replica="{i}"}} {r.inflight}')

        async def grab(i, r):
            try:
                async with self.session.get(r.url + "/metrics", timeout=aiohttp.ClientTimeout(total=3)) as resp:
                    txt = await resp.text()
            except Exception:
                return []
            out = []
            for ln in txt.splitlines():
                if ln.startswith("#") or not ln.strip():
                    continue
                name, _, val = ln.rpartition(" ")
                out.append(f'{name[:-1]},replica="{i}"}} {val}' if name.endswith("}") else f'{name}{{replica="{i}"}} {val}')
            return out
        for part in await asyncio.gather(*(grab(i, r) for i, r in enumerate(self.replicas) if r.healthy)):
            lines += part
        return web.Response(text="\n".join(lines) + "\n", content_type="text/plain")

    def app(self) -> web.Application:
        app = web.Application(client_max_size=64 << 20)
        for path in ("/api/generate", "/api/chat", "/v1/completions", "/v1/chat/completions"):
            app.router.add_post(path, self.proxy)
        app.router.add_get("/health", self.health)
        app.router.add_get("/router/stats", self.stats)
        app.router.add_get("/metrics", self.metrics)

        async def on_startup(_):
            await self.start()

        async def on_cleanup(_):
            await self.stop()
        app.on_startup.append(on_startup)
        app.on_cleanup.append(on_cleanup)
        return app


async def start_router(router: Router, host: str = "127.0.0.1", port: int = 0):
    runner = web.AppRunner(router.app(), access_log=None)
    await runner.setup()
    site = web.TCPSite(runner, host, port, backlog=4096)
    await site.start()
    return runner, f"http://{host}:{site._server.sockets[0].getsockname()[1]}"


def spawn_replicas(n: int, base_port: int, server_args: list[str], gpus: bool = True) -> list[subprocess.Popen]:
    """One ``api_server`` process per GPU (``HIP_VISIBLE_DEVICES=i``), ports ``base_port + i``.  With more
    replicas than GPUs they wrap around (single-GPU rehearsal of the DP path: give each replica an explicit
    ``--num-kv-blocks`` so that they do not all size their cache from the same free memory)."""
    procs = []
    n_gpu = 1
    if gpus:
        import torch
        n_gpu = max(torch.cuda.device_count(), 1)            # counting devices does not initialise the GPU
    for i in range(n):
        env = dict(os.environ)
        if gpus:
            env["HIP_VISIBLE_DEVICES"] = str(i % n_gpu)
        cmd = [sys.executable, "-m", "dli_amd.server.api_server", "--host", "127.0.0.1", "--port", str(base_port + i)] + server_args
        procs.append(subprocess.Popen(cmd, env=env))
    return procs


async def wait_ready(urls: list[str], timeout: float = 1800.0, procs: list | None = None):
    """Block until every replica answers ``/health``.  A spawned replica that EXITS while we wait (out of memory, bad
    arguments) fails the start-up at once instead of leaving the caller polling a router without replicas."""
    t0 = time.time()
    async with aiohttp.ClientSession() as s:
        for i, u in enumerate(urls):
            while True:
                if procs is not None and procs[i].poll() is not None:
                    raise RuntimeError(f"replica {u} exited with code {procs[i].returncode} during start-up")
                try:
                    async with s.get(u + "/health", timeout=aiohttp.ClientTimeout(total=2)) as r:
                        if r.status == 200:
                            break
                except Exception:
                    pass
                if time.time() - t0 > timeout:
                    raise TimeoutError(f"replica {u} did not come up")
                await asyncio.sleep(0.5)


def main(argv=None):
    ap = argparse.ArgumentParser(description="dli_amd data-parallel router")
    ap.add_argument("--replicas", default="", help="comma-separated replica base URLs")
    ap.add_argument("--spawn", type=int, default=0, help="spawn this many replicas (one per GPU) instead")
    ap.add_argument("--replica-base-port", type=int, default=18000)
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=11434)
    ap.add_argument("--affinity-slack", type=float, default=0.5)
    ap.add_argument("--cpu", action="store_true", help="spawned replicas run on CPU (plumbing tests)")
    ap.add_argument("--workers", type=int, default=1,
                    help="router processes sharing the port (SO_REUSEPORT); one relays roughly 100 k streamed lines/s")
    a, server_args = ap.parse_known_args(argv)
    procs = []
    if a.spawn:
        urls = [f"http://127.0.0.1:{a.replica_base_port + i}" for i in range(a.spawn)]
        procs = spawn_replicas(a.spawn, a.replica_base_port, server_args, gpus=not a.cpu)
        try:
            asyncio.run(wait_ready(urls, procs=procs))
        except BaseException:
            for p in procs:
                p.terminate()
            raise
    else:
        urls = [u for u in a.replicas.split(",") if u]
    if not urls:
        ap.error("give --replicas or --spawn")
    shared = SharedLoad(len(urls)) if a.workers > 1 else None      # created before the fork: one table for all workers

    def serve():
        router = Router(urls, affinity_slack=a.affinity_slack, stateless_affinity=a.workers > 1, shared=shared)
        web.run_app(router.app(), host=a.host, port=a.port, access_log=None, backlog=4096, reuse_port=a.workers > 1,
                    print=None if a.workers > 1 else print)

    workers = []
    try:
        if a.workers > 1:
            import multiprocessing as mp
            workers = [mp.get_context("fork").Process(target=serve, daemon=True) for _ in range(a.workers - 1)]
            for w in workers:
                w.start()
        serve()
    finally:
        for w in workers:
            w.terminate()
        for p in procs:
            p.terminate()


if __name__ == "__main__":
    main()
