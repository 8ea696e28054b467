"""HTTP front end (aiohttp): the first-party replacement of the black-box server the reference
harness pointed at (`/root/reference/traffic_generator/main.py:306`).

Endpoints
  POST /api/generate          Ollama wire format — what the harness actually calls.  Streaming body is
                              ``application/x-ndjson``: one ``{"model","created_at","response","done":false}``
                              line per token and a final ``done:true`` line with ``done_reason``, ``context``
                              and the ns duration counters recorded in
                              `/root/reference/notebooks/aiohttp_tracing.ipynb:192-197`.
  POST /v1/completions        OpenAI-compatible; SSE (``data: {...}\\n\\n`` … ``data: [DONE]``) when streaming.
  POST /v1/chat/completions   minimal chat wrapper over the same engine.
  GET  /api/tags /v1/models /api/version /health /stats /metrics (Prometheus text)

Behavioural notes:
* top-level ``temperature`` / ``max_tokens`` of the harness payload are honoured (Ollama ignored them);
  Ollama-style ``options.{temperature,num_predict,top_p,top_k,seed}`` are accepted too;
* response headers are flushed at admission, *before* the first token exists, so the harness'
  ``response_headers_received_time`` is a true server acknowledgement time;
* a client that disconnects mid-stream aborts its request, which frees the KV pages.

    python -m dli_amd.server.api_server --model llama3-8b --port 11434
"""
from __future__ import annotations

import argparse
import asyncio
import itertools
import json
import time
from datetime import datetime, timezone

from aiohttp import web

from ..config import (CacheConfig, EngineConfig, ParallelConfig, SamplingParams, SchedulerConfig, get_model_config)
from ..engine.async_engine import AsyncEngine
from ..engine.tokenizer import get_tokenizer


def _now_iso() -> str:
    return datetime.now(timezone.utc).isoformat().replace("+00:00", "Z")



class StopMatcher:
    """Stop strings over the detokenised stream (Ollama ``options.stop``, OpenAI ``stop``).  ``feed`` returns the text that
    may be sent now — everything except a suffix that could still grow into a stop string — and whether a stop string
    was completed (the text before it is returned, the string itself is not, as both APIs specify)."""

    def __init__(self, stops):
        self.stops = [s for s in stops if isinstance(s, str) and s]
        self.hold = ""

    def __bool__(self):
        return bool(self.stops)

    def feed(self, piece: str):
        buf = self.hold + piece
        hits = [i for i in (buf.find(s) for s in self.stops) if i >= 0]
        if hits:
            self.hold = ""
            return buf[:min(hits)], True
        keep = 0
        for s in self.stops:
            for n in range(min(len(s) - 1, len(buf)), keep, -1):
                if buf.endswith(s[:n]):
                    keep = n
                    break
        self.hold = buf[len(buf) - keep:] if keep else ""
        return (buf[:len(buf) - keep] if keep else buf), False

    def flush(self) -> str:
        out, self.hold = self.hold, ""
        return out


class ApiServer:
    def __init__(self, cfg: EngineConfig, engine_mode: str = "thread", served_name: str | None = None,
                 default_max_tokens: int = 128, engine=None, max_pending_requests: int = 0):
        self.cfg = cfg
        self.aengine = AsyncEngine(cfg, mode=engine_mode, engine=engine)
        self.tokenizer = get_tokenizer(cfg.model, cfg.tokenizer_path)
        self.served_name = served_name or cfg.model.name
        self.default_max_tokens = default_max_tokens
        self._ids = itertools.count()
        self.started = time.time()
        # overload shedding at the door: requests admitted over HTTP that have not produced their first token yet
        # (queued or prefilling).  Past the limit a new request gets 503 + Retry-After instead of joining a queue
        # it cannot leave in time (0 = unlimited); the engine's own max_queue_delay is the backstop behind it.
        self.max_pending_requests = max_pending_requests
        self._pending = 0
        self.shed = 0

    # ------------------------------------------------------------------ helpers
    def _params(self, body: dict) -> SamplingParams:
        opt = body.get("options") or {}

        def pick(*names, default=None):
            for n in names:
                if body.get(n) is not None:
                    return body[n]
                if opt.get(n) is not None:
                    return opt[n]
            return default

        max_tokens = pick("max_tokens", "num_predict", "max_completion_tokens", default=self.default_max_tokens)
        if int(max_tokens) <= 0:
            max_tokens = self.default_max_tokens
        seed = pick("seed")
        return SamplingParams(temperature=float(pick("temperature", default=0.8)), top_p=float(pick("top_p", default=1.0)),
                              top_k=int(pick("top_k", default=0)), max_tokens=int(max_tokens),
                              ignore_eos=bool(pick("ignore_eos", default=False)),
                              seed=None if seed is None else int(seed))

    @staticmethod
    def _stops(body: dict) -> list:
        v = body.get("stop")
        if v is None:
            v = (body.get("options") or {}).get("stop")
        if v is None:
            return []
        return [v] if isinstance(v, str) else [x for x in v if isinstance(x, str)][:16]

    def _rid(self, prefix: str) -> str:
        return f"{prefix}-{next(self._ids)}"

    @staticmethod
    def _bad(msg: str, status: int = 400):
        return web.json_response({"error": msg}, status=status)

    def _overloaded(self):
        if self.max_pending_requests and self._pending >= self.max_pending_requests:
            self.shed += 1
            return web.json_response({"error": "server overloaded: admission queue is full, retry later"}, status=503,
                                     headers={"Retry-After": "1"})
        return None

    async def _tracked(self, rid, ids, params, n: int = 1):
        """``aengine.generate`` with the pending-first-token count kept around it."""
        self._pending += 1
        waiting = True
        try:
            async for item in self.aengine.generate(rid, ids, params, n):
                if waiting and (item.token is not None or item.finished):
                    waiting = False
                    self._pending -= 1
                yield item
        finally:
            if waiting:
                self._pending -= 1

    async def _json(self, request):
        try:
            body = await request.json()
            if not isinstance(body, dict):
                raise ValueError("body must be a JSON object")
            return body
        except Exception as e:
            raise web.HTTPBadRequest(text=json.dumps({"error": f"invalid JSON body: {e}"}),
                                     content_type="application/json")

    # ------------------------------------------------------------------ Ollama: /api/generate, /api/chat
    async def api_chat(self, request: web.Request):
        """Ollama chat API (what ``langchain_ollama.ChatOllama`` of the reference's notebooks calls,
        ``/root/reference/notebooks/llm_requests.ipynb:53-57``): ``messages`` in, ``message`` pieces out, same NDJSON
        framing and timing tail as ``/api/generate``."""
        return await self.api_generate(request, chat=True)

    async def api_generate(self, request: web.Request, chat: bool = False):
        body = await self._json(request)
        if chat:
            msgs = body.get("messages")
            if not isinstance(msgs, list) or not msgs or not all(isinstance(m, dict) for m in msgs):
                return self._bad("'messages' must be a non-empty list of {role, content} objects")
            prompt = "".join(f"{m.get('role', 'user')}: {m.get('content', '')}\n" for m in msgs) + "assistant: "
        else:
            prompt = body.get("prompt")
        if not isinstance(prompt, str):
            return self._bad("'prompt' must be a string")
        model = body.get("model") or self.served_name

        def line(piece):                                             # one streamed record
            if chat:
                return {"model": model, "created_at": _now_iso(), "message": {"role": "assistant", "content": piece}, "done": False}
            return {"model": model, "created_at": _now_iso(), "response": piece, "done": False}
        stream = body.get("stream", True)
        busy = self._overloaded()
        if busy is not None:
            return busy
        params = self._params(body)
        ids = self.tokenizer.encode(prompt)
        rid = self._rid("gen")
        t0 = time.perf_counter_ns()
        dec = self.tokenizer.stream_decoder()
        stops = StopMatcher(self._stops(body))
        t_first = None
        out_ids: list[int] = []
        reason = "stop"

        if stream:
            resp = web.StreamResponse(headers={"Content-Type": "application/x-ndjson"})
            await resp.prepare(request)                              # ack now: admission, not first token
            dumps = json.dumps
            try:
                agen = self._tracked(rid, ids, params)
                hit = False
                try:
                    async for item in agen:
                        if item.token is not None:
                            if t_first is None:
                                t_first = time.perf_counter_ns()
                            out_ids.append(item.token)
                            piece = dec.step(item.token)
                            if stops:
                                piece, hit = stops.feed(piece)
                            await resp.write((dumps(line(piece)) + "\n").encode())
                            if hit:                                  # a stop string is complete: the engine drops the request
                                reason = "stop"
                                break
                        if item.finished:
                            reason = item.reason or "stop"
                finally:
                    await agen.aclose()                              # early exit (stop string, client gone): abort in the engine
                t_end = time.perf_counter_ns()
                last = "" if hit else (stops.feed(dec.flush())[0] + stops.flush() if stops else dec.flush())
                tail = self._ollama_tail(model, reason, ids, out_ids, t0, t_first, t_end, last, chat)
                await resp.write((dumps(tail) + "\n").encode())
                await resp.write_eof()
            except (ConnectionResetError, asyncio.CancelledError):
                raise                                                # generator's finally aborts the request
            return resp

        text = []
        agen = self._tracked(rid, ids, params)
        hit = False
        try:
            async for item in agen:
                if item.token is not None:
                    if t_first is None:
                        t_first = time.perf_counter_ns()
                    out_ids.append(item.token)
                    piece = dec.step(item.token)
                    if stops:
                        piece, hit = stops.feed(piece)
                    text.append(piece)
                    if hit:
                        reason = "stop"
                        break
                if item.finished:
                    reason = item.reason or "stop"
        finally:
            await agen.aclose()
        if not hit:
            text.append(stops.feed(dec.flush())[0] + stops.flush() if stops else dec.flush())
        t_end = time.perf_counter_ns()
        return web.json_response(self._ollama_tail(model, reason, ids, out_ids, t0, t_first, t_end, "".join(text), chat))

    @staticmethod
    def _ollama_tail(model, reason, ids, out_ids, t0, t_first, t_end, response_text, chat=False):
        t_first = t_first or t_end
        text = {"message": {"role": "assistant", "content": response_text}} if chat else {"response": response_text}
        if reason == "shed":
            reason = "error: server overloaded, request shed after waiting too long for admission"
        if reason.startswith("error"):
            text = {"message": {"role": "assistant", "content": ""}} if chat else {"response": ""}
            return {"model": model, "created_at": _now_iso(), **text, "done": True, "done_reason": "error", "error": reason}
        return {"model": model, "created_at": _now_iso(), **text, "done": True,
                "done_reason": "stop" if reason == "stop" else reason,
                **({} if chat else {"context": list(ids) + out_ids}), "total_duration": t_end - t0, "load_duration": 0,
                "prompt_eval_count": len(ids), "prompt_eval_duration": t_first - t0,
                "eval_count": len(out_ids), "eval_duration": t_end - t_first}

    # ------------------------------------------------------------------ OpenAI: /v1/completions
    async def v1_completions(self, request: web.Request, chat: bool = False):
        body = await self._json(request)
        if chat:
            msgs = body.get("messages")
            if not isinstance(msgs, list) or not msgs:
                return self._bad("'messages' must be a non-empty list")
            prompt = "".join(f"{m.get('role', 'user')}: {m.get('content', '')}\n" for m in msgs) + "assistant: "
        else:
            prompt = body.get("prompt")
            if isinstance(prompt, list) and prompt and all(isinstance(x, int) for x in prompt):
                prompt = None
            elif isinstance(prompt, list) and len(prompt) == 1:
                prompt = prompt[0]
            if prompt is not None and not isinstance(prompt, str):
                return self._bad("'prompt' must be a string or a list of token ids")
        busy = self._overloaded()
        if busy is not None:
            return busy
        ids = body["prompt"] if prompt is None else self.tokenizer.encode(prompt)
        model = body.get("model") or self.served_name
        params = self._params(body)
        # OpenAI `n`: n samples of ONE prompt — prefilled once, the other n-1 forked off its KV pages (copy-on-write)
        try:
            n = int(body.get("n") or 1)
        except (TypeError, ValueError):
            return self._bad("'n' must be an integer")
        if not 1 <= n <= 16:
            return self._bad("'n' must be between 1 and 16")
        rid = self._rid("chatcmpl" if chat else "cmpl")
        created = int(time.time())
        obj = "chat.completion" if chat else "text_completion"
        decs = [self.tokenizer.stream_decoder() for _ in range(n)]
        # stop strings, per choice.  One choice: the request is dropped at the hit.  n > 1: the choice goes silent and
        # reports finish_reason "stop"; its siblings share the stream and run on
        stops = [StopMatcher(self._stops(body)) for _ in range(n)]
        stopped = [False] * n

        def chunk(piece, finish=None, streaming=True, index=0):
            if chat:
                key = "delta" if streaming else "message"
                ch = {"index": index, key: {"role": "assistant", "content": piece}, "finish_reason": finish}
            else:
                ch = {"index": index, "text": piece, "logprobs": None, "finish_reason": finish}
            return {"id": rid, "object": obj + (".chunk" if chat and streaming else ""), "created": created,
                    "model": model, "choices": [ch]}

        if body.get("stream"):
            resp = web.StreamResponse(headers={"Content-Type": "text/event-stream", "Cache-Control": "no-cache"})
            await resp.prepare(request)
            n_out = n_prompt = seen = 0
            agen = self._tracked(rid, ids, params, n)
            try:
                async for item in agen:
                    finish = None
                    c = item.choice
                    seen += item.token is not None
                    dec = decs[c]
                    if item.finished:
                        finish = "length" if item.reason == "length" else (item.reason or "stop")
                        n_out, n_prompt = n_out + item.n_out, item.n_prompt
                    if stopped[c]:
                        continue
                    piece = dec.step(item.token) if item.token is not None else ""
                    if item.finished:
                        piece += dec.flush()
                    if stops[c]:
                        piece, stopped[c] = stops[c].feed(piece)
                        if stopped[c]:
                            finish = "stop"
                        elif item.finished:
                            piece += stops[c].flush()
                    await resp.write(("data: " + json.dumps(chunk(piece, finish, index=c)) + "\n\n").encode())
                    if stopped[c] and n == 1:
                        n_out, n_prompt = seen, len(ids)
                        break
            finally:
                await agen.aclose()
            if (body.get("stream_options") or {}).get("include_usage"):
                usage = {"id": rid, "object": obj, "created": created, "model": model, "choices": [],
                         "usage": {"prompt_tokens": n_prompt, "completion_tokens": n_out, "total_tokens": n_prompt + n_out}}
                await resp.write(("data: " + json.dumps(usage) + "\n\n").encode())
            await resp.write(b"data: [DONE]\n\n")
            await resp.write_eof()
            return resp

        pieces, finish, n_out, n_prompt = [[] for _ in range(n)], ["stop"] * n, 0, len(ids)
        counted = [0] * n
        agen = self._tracked(rid, ids, params, n)
        try:
            async for item in agen:
                c = item.choice
                if item.finished:
                    n_out, n_prompt = n_out + item.n_out, item.n_prompt or n_prompt
                if stopped[c]:
                    continue
                if item.token is not None:
                    counted[c] += 1
                    piece = decs[c].step(item.token)
                    if stops[c]:
                        piece, stopped[c] = stops[c].feed(piece)
                    pieces[c].append(piece)
                if stopped[c]:
                    finish[c] = "stop"
                    if n == 1:
                        n_out = counted[0]
                        break
                elif item.finished:
                    finish[c] = "length" if item.reason == "length" else (item.reason or "stop")
        finally:
            await agen.aclose()
        for i in range(n):
            if not stopped[i]:
                tail = decs[i].flush()
                pieces[i].append(stops[i].feed(tail)[0] + stops[i].flush() if stops[i] else tail)
        out = chunk("".join(pieces[0]), finish[0], streaming=False)
        for i in range(1, n):
            out["choices"].append(chunk("".join(pieces[i]), finish[i], streaming=False, index=i)["choices"][0])
        out["usage"] = {"prompt_tokens": n_prompt, "completion_tokens": n_out, "total_tokens": n_prompt + n_out}
        return web.json_response(out)

    async def v1_chat(self, request):
        return await self.v1_completions(request, chat=True)

    # ------------------------------------------------------------------ introspection
    async def health(self, request):
        return web.json_response({"status": "ok", "model": self.served_name, "uptime_s": time.time() - self.started,
                                  "streams": self.aengine.num_streams})

    async def stats(self, request):
        s = await self.aengine.stats()
        s.pop("prometheus", None)
        s["streams"] = self.aengine.num_streams
        s["pending_first_token"] = self._pending
        s["shed_503"] = self.shed
        return web.json_response(s)

    async def metrics(self, request):
        s = await self.aengine.stats()
        return web.Response(text=s["prometheus"], content_type="text/plain")

    async def tags(self, request):
        m = self.cfg.model
        return web.json_response({"models": [{"name": self.served_name, "model": self.served_name,
                                              "details": {"family": m.arch, "parameter_size": f"{m.param_count() / 1e9:.1f}B",
                                                          "quantization_level": m.dtype}}]})

    async def models(self, request):
        return web.json_response({"object": "list", "data": [{"id": self.served_name, "object": "model",
                                                              "created": int(self.started), "owned_by": "dli_amd"}]})

    async def version(self, request):
        from .. import __version__
        return web.json_response({"version": __version__})

    # ------------------------------------------------------------------ wiring
    def app(self) -> web.Application:
        app = web.Application(client_max_size=64 << 20)
        r = app.router
        r.add_post("/api/generate", self.api_generate)
        r.add_post("/api/chat", self.api_chat)
        r.add_post("/v1/completions", self.v1_completions)
        r.add_post("/v1/chat/completions", self.v1_chat)
        r.add_get("/health", self.health)
        r.add_get("/stats", self.stats)
        r.add_get("/metrics", self.metrics)
        r.add_get("/api/tags", self.tags)
        r.add_get("/v1/models", self.models)
        r.add_get("/api/version", self.version)

        async def on_startup(_app):
            await self.aengine.start()

        async def on_cleanup(_app):
            await self.aengine.shutdown()

        app.on_startup.append(on_startup)
        app.on_cleanup.append(on_cleanup)
        return app


async def start_server(server: ApiServer, host: str = "127.0.0.1", port: int = 0):
    """Programmatic start (tests, router): returns ``(runner, base_url)``."""
    runner = web.AppRunner(server.app(), access_log=None)
    await runner.setup()
    site = web.TCPSite(runner, host, port, backlog=4096)
    await site.start()
    real_port = site._server.sockets[0].getsockname()[1]
    return runner, f"http://{host}:{real_port}"


def engine_config_from_args(a) -> EngineConfig:
    overrides = {}
    if a.max_position:
        overrides["max_position"] = a.max_position
    if getattr(a, "num_layers", None):
        overrides["num_layers"] = a.num_layers
    return EngineConfig(
        model=get_model_config(a.model, **overrides),
        cache=CacheConfig(kv_dtype=a.kv_dtype, num_blocks=a.num_kv_blocks, gpu_memory_utilization=a.gpu_memory_utilization,
                          enable_prefix_caching=not a.no_prefix_caching,
                          calibration_file=getattr(a, "kv_calibration_file", None)),
        sched=SchedulerConfig(max_num_seqs=a.max_num_seqs, max_num_batched_tokens=a.max_num_batched_tokens,
                              max_model_len=a.max_model_len, long_prefill_chunk=a.prefill_chunk, ttft_slo=a.ttft_slo,
                              tpot_slo=a.tpot_slo, max_queue_delay=a.max_queue_delay),
        parallel=ParallelConfig(tp=a.tp, ep=a.ep, custom_all_reduce=not a.no_custom_all_reduce),
        device=a.device, seed=a.seed, use_hip_graphs=not a.no_hip_graphs, weights_path=a.weights,
        tokenizer_path=a.tokenizer, quantization=getattr(a, "quantization", None),
        library_gemm_tuning=getattr(a, "tune_library_gemms", None))


def add_engine_args(ap: argparse.ArgumentParser):
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--weights", default=None, help="safetensors file/dir (default: random init)")
    ap.add_argument("--tokenizer", default=None, help="HF tokenizer.json (default: byte-level)")
    ap.add_argument("--device", default="auto", choices=["auto", "cuda", "cpu"])
    ap.add_argument("--tp", type=int, default=1)
    ap.add_argument("--ep", type=int, default=1)
    ap.add_argument("--kv-dtype", default="auto", choices=["auto", "fp8"])
    ap.add_argument("--tune-library-gemms", default=None, metavar="FILE",
                    help="PyTorch TunableOp results file: tune the vendor GEMMs per shape at the first start, load the file afterwards")
    ap.add_argument("--quantization", default=None, choices=["fp8"],
                    help="W8A8: dense projections in OCP e4m3 (per-channel weight scales, per-token activation scales) for steps above 64 rows")
    ap.add_argument("--kv-calibration-file", default=None,
                    help="fp8 KV: text file of representative prompts to calibrate the per-head scales on (default: random ids)")
    ap.add_argument("--num-kv-blocks", type=int, default=None)
    ap.add_argument("--gpu-memory-utilization", type=float, default=0.92)
    ap.add_argument("--no-prefix-caching", action="store_true")
    ap.add_argument("--max-num-seqs", type=int, default=2048,
                    help="sequences running at once (rows of a decode step); one MI355X holds 1.9 M tokens of bf16 KV for Llama-3-8B")
    ap.add_argument("--max-num-batched-tokens", type=int, default=4096)
    ap.add_argument("--max-model-len", type=int, default=4096)
    ap.add_argument("--max-position", type=int, default=None)
    ap.add_argument("--num-layers", type=int, default=None, help="override the layer count (rehearsals of big shapes)")
    ap.add_argument("--prefill-chunk", type=int, default=2048)
    ap.add_argument("--ttft-slo", type=float, default=None,
                    help="seconds; waiting requests already past this time-to-first-token deadline yield to fresher ones")
    ap.add_argument("--tpot-slo", type=float, default=None,
                    help="seconds per output token; the prefill share of a step shrinks while decodes would exceed it")
    ap.add_argument("--max-queue-delay", type=float, default=None,
                    help="seconds; a request still not admitted after this long is shed (stream ends with an error line)")
    ap.add_argument("--no-hip-graphs", action="store_true")
    ap.add_argument("--no-custom-all-reduce", action="store_true")
    ap.add_argument("--seed", type=int, default=0)


def main(argv=None):
    ap = argparse.ArgumentParser(description="dli_amd inference server")
    add_engine_args(ap)
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=11434)
    ap.add_argument("--served-model-name", default=None)
    ap.add_argument("--engine-mode", default="process", choices=["process", "thread"])
    ap.add_argument("--default-max-tokens", type=int, default=128)
    ap.add_argument("--max-pending-requests", type=int, default=0,
                    help="answer 503 + Retry-After once this many requests wait for their first token (0: unlimited)")
    a = ap.parse_args(argv)
    cfg = engine_config_from_args(a)
    server = ApiServer(cfg, engine_mode=a.engine_mode, served_name=a.served_model_name,
                       default_max_tokens=a.default_max_tokens, max_pending_requests=a.max_pending_requests)
    web.run_app(server.app(), host=a.host, port=a.port, access_log=None, backlog=4096)


if __name__ == "__main__":
    main()
