"""How many streamed lines per second can ONE router process relay?  (BASELINE.json config 3 puts 8 replicas
behind it: 8 × ~65 req/s × ~300 tokens ≈ 150 k lines/s.)  CPU only: N fake NDJSON replicas (own processes) that
stream `--tokens` lines per request with no delay, the router in its own process, and a closed-loop client with
`--concurrency` streams in this process.

    python benchmarks/router_capacity.py --replicas 2 --concurrency 256 --requests 2000 --tokens 200
"""
import argparse
import asyncio
import json
import multiprocessing as mp
import os
import sys
import time

import aiohttp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "traffic_generator"))


def run_fake(port, tokens):
    from aiohttp import web
    import fake_server
    fake = fake_server.FakeOllama(token_delay=0.0, n_tokens=tokens)
    app = fake.app()

    async def health(_request):
        return web.json_response({"status": "ok"})
    app.router.add_get("/health", health)                 # the router probes it
    web.run_app(app, host="127.0.0.1", port=port, print=None, access_log=None)


def run_router(port, urls, workers):
    from dli_amd.router import router as R
    R.main(["--replicas", ",".join(urls), "--host", "127.0.0.1", "--port", str(port)] + (["--workers", str(workers)] if workers > 1 else []))


async def wait_up(url):
    async with aiohttp.ClientSession() as s:
        for _ in range(200):
            try:
                async with s.get(url, timeout=aiohttp.ClientTimeout(total=1)) as r:
                    if r.status < 500:
                        return
            except Exception:
                await asyncio.sleep(0.1)
    raise TimeoutError(url)


async def client(url, n_req, conc, tokens):
    sem = asyncio.Semaphore(conc)
    lines = 0
    complete = 0
    conn = aiohttp.TCPConnector(limit=0)
    async with aiohttp.ClientSession(connector=conn, timeout=aiohttp.ClientTimeout(total=None)) as s:
        async def one(i):
            nonlocal lines, complete
            async with sem:
                body = {"model": "m", "prompt": f"request {i} " * 40, "max_tokens": tokens, "stream": True}
                n = 0
                async with s.post(url, json=body) as r:
                    async for _ in r.content:
                        n += 1
                lines += n
                complete += n == tokens + 1
        t0 = time.perf_counter()
        await asyncio.gather(*(one(i) for i in range(n_req)))
        dt = time.perf_counter() - t0
    return lines, dt, complete


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=2)
    ap.add_argument("--workers", type=int, default=1)
    ap.add_argument("--concurrency", type=int, default=256)
    ap.add_argument("--requests", type=int, default=2000)
    ap.add_argument("--tokens", type=int, default=200)
    ap.add_argument("--direct", action="store_true", help="bypass the router (client → replica 0): the fake's own ceiling")
    a = ap.parse_args()
    ctx = mp.get_context("spawn")
    procs = []
    urls = []
    for i in range(a.replicas):
        port = 19100 + i
        p = ctx.Process(target=run_fake, args=(port, a.tokens), daemon=True)
        p.start()
        procs.append(p)
        urls.append(f"http://127.0.0.1:{port}")
    rp = ctx.Process(target=run_router, args=(19099, urls, a.workers), daemon=False)   # it forks its own workers
    rp.start()
    procs.append(rp)
    try:
        asyncio.run(wait_up("http://127.0.0.1:19099/health"))
        target = (urls[0] if a.direct else "http://127.0.0.1:19099") + "/api/generate"
        lines, dt, complete = asyncio.run(client(target, a.requests, a.concurrency, a.tokens))
        print(json.dumps({"path": "direct" if a.direct else f"router x{a.workers}", "replicas": a.replicas,
                          "requests": a.requests, "complete_streams": complete, "concurrency": a.concurrency, "lines": lines, "seconds": round(dt, 2),
                          "lines_per_s": round(lines / dt), "req_per_s": round(a.requests / dt, 1)}))
    finally:
        for p in procs:
            p.terminate()


if __name__ == "__main__":
    main()
