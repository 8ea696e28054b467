"""In-process fake of the black-box server the reference harness was written against
(Ollama ``POST /api/generate`` → ``application/x-ndjson`` chunked stream; wire format recorded
in `/root/reference/notebooks/aiohttp_tracing.ipynb:192-197`).  Used only by the harness tests:
configurable per-token delay, serial (Ollama-like, no batching) or concurrent service, and
fault injection (HTTP 500 / abrupt connection reset on chosen request indices).
"""
from __future__ import annotations

import asyncio
import json
import time
from datetime import datetime, timezone

from aiohttp import web


def _now():
    return datetime.now(timezone.utc).isoformat().replace("+00:00", "Z")


class FakeOllama:
    def __init__(self, token_delay: float = 0.001, n_tokens: int | None = None, serial: bool = False,
                 fail_500: set | None = None, reset_after_first: set | None = None,
                 reset_before_headers: set | None = None):
        self.token_delay = token_delay
        self.n_tokens = n_tokens
        self.serial = serial
        self.fail_500 = fail_500 or set()
        self.reset_after_first = reset_after_first or set()
        self.reset_before_headers = reset_before_headers or set()
        self.seen: list[dict] = []
        self._lock = asyncio.Lock()

    async def generate(self, request: web.Request):
        body = await request.json()
        idx = len(self.seen)
        self.seen.append(body)
        if idx in self.fail_500:
            return web.json_response({"error": "injected"}, status=500)
        if idx in self.reset_before_headers:
            request.transport.close()
            return web.Response()
        n = self.n_tokens if self.n_tokens is not None else int(body.get("max_tokens", 8))
        if self.serial:
            await self._lock.acquire()
        try:
            t0 = time.perf_counter_ns()
            resp = web.StreamResponse(headers={"Content-Type": "application/x-ndjson"})
            await resp.prepare(request)
            for i in range(n):
                await asyncio.sleep(self.token_delay)
                line = {"model": body.get("model"), "created_at": _now(), "response": f" t{i}", "done": False}
                await resp.write((json.dumps(line) + "\n").encode())
                if i == 0 and idx in self.reset_after_first:
                    request.transport.close()
                    return resp
            total = time.perf_counter_ns() - t0
            tail = {"model": body.get("model"), "created_at": _now(), "response": "", "done": True,
                    "done_reason": "stop", "context": [], "total_duration": total, "load_duration": 0,
                    "prompt_eval_count": len(body.get("prompt", "")) + 1, "prompt_eval_duration": 0,
                    "eval_count": n, "eval_duration": total}
            await resp.write((json.dumps(tail) + "\n").encode())
            await resp.write_eof()
            return resp
        finally:
            if self.serial:
                self._lock.release()

    def app(self) -> web.Application:
        app = web.Application()
        app.router.add_post("/api/generate", self.generate)

        async def health(_request):
            return web.json_response({"status": "ok"})
        app.router.add_get("/health", health)
        return app


async def start(fake: FakeOllama, host: str = "127.0.0.1", port: int = 0):
    """Starts the fake on an ephemeral port; returns ``(runner, url)``."""
    runner = web.AppRunner(fake.app())
    await runner.setup()
    site = web.TCPSite(runner, host, port)
    await site.start()
    real_port = site._server.sockets[0].getsockname()[1]
    return runner, f"http://{host}:{real_port}/api/generate"


def main(argv=None):
    """``python fake_server.py --port 11434``: stand-alone fake for manual runs and the subprocess tests."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=11434)
    ap.add_argument("--token-delay", type=float, default=0.001)
    ap.add_argument("--n-tokens", type=int, default=None)
    ap.add_argument("--serial", action="store_true")
    a = ap.parse_args(argv)
    web.run_app(FakeOllama(a.token_delay, a.n_tokens, a.serial).app(), host=a.host, port=a.port, print=None)


if __name__ == "__main__":
    main()
