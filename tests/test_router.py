"""DP router tests: policy unit tests + an integration run with two CPU replicas (SURVEY.md §4 item 5:
"router tests can spawn N CPU replicas")."""
import asyncio
import json

import os
import sys

import aiohttp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "traffic_generator"))
from fake_server import FakeOllama, start  # noqa: E402

from dli_amd.config import CacheConfig, EngineConfig, SchedulerConfig, get_model_config
from dli_amd.router.router import Replica, Router, RoutingPolicy, prefix_chain, start_router
from dli_amd.server.api_server import ApiServer, start_server


def test_prefix_chain_is_prefix_stable():
    a = "s" * 300 + "x" * 600
    b = a + "more text " * 80
    ca, cb = prefix_chain(a), prefix_chain(b)
    assert len(ca) == 3 and cb[:3] == ca and len(cb) > 3
    assert prefix_chain("short") == []
    assert prefix_chain("t" * 300)[0] != ca[0]


def test_policy_least_outstanding_tokens_and_affinity():
    reps = [Replica(f"http://r{i}") for i in range(3)]
    pol = RoutingPolicy(reps, affinity_slack=0.5, slack_tokens=100)
    reps[0].outstanding_tokens, reps[1].outstanding_tokens, reps[2].outstanding_tokens = 5000, 100, 3000
    assert pol.choose([]) == 1
    chain = prefix_chain("p" * 1000)
    pol.record(2, chain)
    reps[2].outstanding_tokens = 200                     # within slack of the best → affinity wins
    assert pol.choose(chain + [123]) == 2
    reps[2].outstanding_tokens = 50_000                  # overloaded → fall back to least loaded
    assert pol.choose(chain) == 1
    reps[1].healthy = False
    assert pol.choose([]) == 0
    assert pol.choose([], exclude=frozenset({0})) == 2
    pol.record(0, [], session="abc")
    reps[0].outstanding_tokens = 0
    assert pol.choose([], session="abc") == 0
    for r in reps:
        r.healthy = False
    assert pol.choose([]) is None


def _server():
    cfg = EngineConfig(model=get_model_config("tiny-opt", max_position=2048), cache=CacheConfig(num_blocks=256),
                       sched=SchedulerConfig(max_num_seqs=16, max_num_batched_tokens=256, max_model_len=1536),
                       device="cpu", use_hip_graphs=False)
    return ApiServer(cfg, engine_mode="thread")


def test_router_end_to_end_two_replicas_affinity_and_failover():
    async def go():
        s1, s2 = _server(), _server()
        r1, u1 = await start_server(s1)
        r2, u2 = await start_server(s2)
        router = Router([u1, u2], health_interval=0.2, max_fails=1)
        rr, base = await start_router(router)
        try:
            async with aiohttp.ClientSession() as s:
                async def gen(prompt, **kw):
                    async with s.post(base + "/api/generate", json={"prompt": prompt, "max_tokens": 4, **kw}) as r:
                        lines = [json.loads(l) async for l in r.content]
                        return r.status, r.headers.get("X-Served-By"), lines
                # concurrent distinct prompts spread over both replicas
                res = await asyncio.gather(*(gen(f"{i} " + "q" * 40) for i in range(8)))
                assert all(st == 200 and len(l) == 5 and l[-1]["done"] for st, _, l in res)
                assert {by for _, by, _ in res} == {"0", "1"}
                # multi-turn: the second turn extends the first → same replica, prefix cache hit there
                turn1 = "system: be brief.\n" + "user: " + "w" * 600
                _, by1, l1 = await gen(turn1)
                turn2 = turn1 + "".join(x["response"] for x in l1) + "\nuser: and then?"
                _, by2, _ = await gen(turn2)
                assert by1 == by2
                stats = [await (await s.get(u + "/stats")).json() for u in (u1, u2)]
                assert stats[int(by1)]["cached_prompt_tokens"] >= 512
                # OpenAI path is proxied too
                async with s.post(base + "/v1/completions", json={"prompt": "hi", "max_tokens": 2}) as r:
                    assert r.status == 200 and (await r.json())["usage"]["completion_tokens"] == 2
                # failover: kill replica 0; requests keep succeeding on replica 1
                await r1.cleanup()
                res = await asyncio.gather(*(gen(f"after {i}") for i in range(6)))
                assert all(st == 200 and by == "1" for st, by, _ in res)
                h = await (await s.get(base + "/health")).json()
                assert h["replicas_up"] == 1
                rs = await (await s.get(base + "/router/stats")).json()
                assert rs["replicas"][0]["healthy"] is False and rs["affinity_hits"] >= 1
                prom = await (await s.get(base + "/metrics")).text()
                assert 'dli_router_replica_healthy{replica="0"} 0' in prom and 'dli_requests_total{replica="1"}' in prom
        finally:
            await rr.cleanup()
            await r2.cleanup()
    asyncio.run(go())


def test_stateless_affinity_is_consistent_across_router_workers():
    """Two router workers (separate tables) must send an unknown session / prompt prefix to the same replica."""
    from dli_amd.router.router import Replica, RoutingPolicy, prefix_chain
    mk = lambda: RoutingPolicy([Replica(f"http://r{i}") for i in range(8)], stateless_affinity=True)
    a, b = mk(), mk()
    picks = set()
    for k in range(50):
        chain = prefix_chain(f"conversation {k} " * 40)
        pa, pb = a.choose(chain), b.choose(chain)
        assert pa == pb
        assert a.choose([], session=f"s{k}") == b.choose([], session=f"s{k}")
        picks.add(pa)
    assert len(picks) >= 5                                   # spread over the replicas, not one hot spot
    # an overloaded target still yields to the least-loaded replica
    chain = prefix_chain("conversation 0 " * 40)
    target = a.choose(chain)
    a.replicas[target].outstanding_tokens = 10_000_000
    assert a.choose(chain) != target
    # and a dead one is skipped consistently
    a.replicas[target].outstanding_tokens = 0
    a.replicas[target].healthy = b.replicas[target].healthy = False
    assert a.choose(chain) == b.choose(chain) != target


def test_two_router_workers_share_one_load_table_over_eight_replicas():
    """`--workers 2` (two processes on one port): the load counters live in shared memory, so EVERY worker sees the
    in-flight requests of all workers and the eight replicas end up evenly loaded."""
    import socket
    import subprocess
    import sys
    import time
    import os

    async def body():
        fakes = [FakeOllama(token_delay=0.05, n_tokens=12) for _ in range(8)]
        started = [await start(f) for f in fakes]
        urls = [u.rsplit("/api/generate", 1)[0] for _r, u in started]
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        proc = subprocess.Popen([sys.executable, "-m", "dli_amd.router.router", "--replicas", ",".join(urls), "--host", "127.0.0.1",
                                 "--port", str(port), "--workers", "2"], cwd=root,
                                env=dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", "")))
        base = f"http://127.0.0.1:{port}"
        try:
            async with aiohttp.ClientSession(connector=aiohttp.TCPConnector(limit=0, force_close=True)) as s:
                for _ in range(100):
                    try:
                        async with s.get(base + "/health") as r:
                            if r.status == 200:
                                break
                    except aiohttp.ClientError:
                        await asyncio.sleep(0.1)

                async def one(i):
                    async with s.post(base + "/api/generate", json={"model": "m", "prompt": f"p{i}", "max_tokens": 12}) as r:
                        assert r.status == 200
                        return len([l async for l in r.content])

                tasks = [asyncio.create_task(one(i)) for i in range(64)]
                await asyncio.sleep(0.3)                      # all 64 are streaming now (12 tokens × 50 ms each)
                seen = {}
                for _ in range(12):                            # new connections land on either worker
                    async with s.get(base + "/router/stats") as r:
                        st = await r.json()
                    seen[st["pid"]] = st
                    assert st["shared_load_table"] is True
                    assert sum(rep["inflight"] for rep in st["replicas"]) == 64        # the GLOBAL figure, from any worker
                    assert max(rep["inflight"] for rep in st["replicas"]) <= 10
                lines = await asyncio.gather(*tasks)
                assert all(n == 13 for n in lines)
                async with s.get(base + "/router/stats") as r:
                    st = await r.json()
                assert sum(rep["served"] for rep in st["replicas"]) == 64 and all(rep["inflight"] == 0 for rep in st["replicas"])
            assert len(seen) == 2, "the stats probes should have reached both workers"
            per = [len(f.seen) for f in fakes]
            assert sum(per) == 64 and min(per) >= 6 and max(per) <= 10, per
        finally:
            proc.terminate()
            proc.wait(10)
            for r, _u in started:
                await r.cleanup()

    asyncio.run(body())


def test_router_retries_a_shedding_replica_and_relays_503_when_all_shed():
    from aiohttp import web

    async def body():
        async def shed(request):
            return web.json_response({"error": "server overloaded"}, status=503, headers={"Retry-After": "2"})

        async def start_shedder():
            app = web.Application()
            app.router.add_post("/api/generate", shed)
            async def ok(_r):
                return web.json_response({"status": "ok"})
            app.router.add_get("/health", ok)
            runner = web.AppRunner(app)
            await runner.setup()
            site = web.TCPSite(runner, "127.0.0.1", 0)
            await site.start()
            return runner, f"http://127.0.0.1:{site._server.sockets[0].getsockname()[1]}"

        good = FakeOllama(token_delay=0.001, n_tokens=3)
        rg, ug = await start(good)
        rs, us = await start_shedder()
        router = Router([us, ug.rsplit("/api/generate", 1)[0]], health_interval=5.0)
        rr, base = await start_router(router)
        async with aiohttp.ClientSession() as s:
            for i in range(4):                                  # whichever is tried first, the request ends on the good one
                async with s.post(base + "/api/generate", json={"model": "m", "prompt": "x" * 10, "max_tokens": 3}) as r:
                    assert r.status == 200 and r.headers["X-Served-By"] == "1"
                    assert len([l async for l in r.content]) == 4
        await rr.cleanup()
        router2 = Router([us], health_interval=5.0)
        rr2, base2 = await start_router(router2)
        async with aiohttp.ClientSession() as s:
            async with s.post(base2 + "/api/generate", json={"model": "m", "prompt": "x", "max_tokens": 3}) as r:
                assert r.status == 503 and r.headers["Retry-After"] == "2"
        assert router2.shed == 1
        for r in (rr2, rg, rs):
            await r.cleanup()

    asyncio.run(body())


def test_spawned_replica_that_dies_fails_the_router_start():
    """`--spawn N` with replicas that exit during start-up (here: an unknown model) must end the router with an error
    instead of serving a port without replicas (seen on the GPU box: two replicas sized their KV cache from the same
    free memory, both died, the router kept waiting)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "dli_amd.router.router", "--spawn", "2", "--cpu", "--port", "0",
                        "--replica-base-port", "18740", "--model", "no-such-model", "--device", "cpu"],
                       capture_output=True, text=True, timeout=120, cwd=root,
                       env=dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", "")))
    assert p.returncode != 0 and "exited with code" in p.stderr
