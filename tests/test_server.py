"""HTTP server tests on CPU (tiny models) + the end-to-end plumbing run of BASELINE.json config 1:
the preserved harness replays ``data/trace1.csv`` against the first-party server and must produce a
``log.json`` with the reference schema."""
import asyncio
import json
import os
import re
import sys

import aiohttp
import pytest

import main as tg
import analyze
from dli_amd.config import CacheConfig, EngineConfig, SchedulerConfig, get_model_config
from dli_amd.server.api_server import ApiServer, start_server

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_server(model="tiny-opt", mode="thread", **kw):
    cfg = EngineConfig(model=get_model_config(model, max_position=2048), cache=CacheConfig(num_blocks=256),
                       sched=SchedulerConfig(max_num_seqs=16, max_num_batched_tokens=256, max_model_len=1536,
                                             long_prefill_chunk=128),
                       device="cpu", use_hip_graphs=False)
    return ApiServer(cfg, engine_mode=mode, **kw)


def with_server(coro_fn, **kw):
    async def go():
        server = make_server(**kw)
        runner, base = await start_server(server)
        try:
            return await coro_fn(server, base)
        finally:
            await runner.cleanup()
    return asyncio.run(go())


def test_api_generate_stream_ndjson_wire_format():
    async def body(server, base):
        async with aiohttp.ClientSession() as s:
            payload = {"model": "mistral", "prompt": "hello world", "temperature": 0.7, "max_tokens": 7, "stream": True}
            async with s.post(base + "/api/generate", json=payload) as r:
                assert r.status == 200 and r.headers["Content-Type"].startswith("application/x-ndjson")
                assert r.headers.get("Transfer-Encoding") == "chunked"
                lines = [json.loads(l) async for l in r.content]
        return lines
    lines = with_server(body)
    assert len(lines) == 8                                            # max_tokens honoured + done line
    for l in lines[:-1]:
        assert list(l.keys()) == ["model", "created_at", "response", "done"] and l["done"] is False and l["model"] == "mistral"
    tail = lines[-1]
    assert tail["done"] is True and tail["done_reason"] in ("stop", "length") and tail["eval_count"] == 7
    assert tail["prompt_eval_count"] == len("hello world") + 1 and len(tail["context"]) == tail["prompt_eval_count"] + 7
    for k in ("total_duration", "load_duration", "prompt_eval_duration", "eval_duration"):
        assert isinstance(tail[k], int)
    assert re.match(r"\d{4}-\d\d-\d\dT\d\d:\d\d:\d\d", tail["created_at"])


def test_api_generate_non_stream_and_options_and_seed():
    async def body(server, base):
        async with aiohttp.ClientSession() as s:
            p = {"prompt": "abc", "stream": False, "options": {"num_predict": 5, "temperature": 0.9, "seed": 3, "top_k": 20}}
            a = await (await s.post(base + "/api/generate", json=p)).json()
            b = await (await s.post(base + "/api/generate", json=p)).json()
            bad = await s.post(base + "/api/generate", json={"prompt": 5})
            bad2 = await s.post(base + "/api/generate", data=b"not json")
            return a, b, bad.status, bad2.status
    a, b, st, st2 = with_server(body)
    assert a["done"] and a["eval_count"] == 5 and a["response"] == b["response"] and a["context"] == b["context"]
    assert st == 400 and st2 == 400


def test_stop_strings_cut_the_stream_and_release_the_request():
    """Ollama `options.stop` / OpenAI `stop`: generation ends at the first stop string, the text before it is returned
    and the string itself is not; a stop string that straddles two tokens is found; the request is dropped in the engine
    (no stream left open)."""
    from dli_amd.server.api_server import StopMatcher
    m = StopMatcher(["END", "xy"])
    got = [m.feed(p) for p in ("hel", "lo E", "N", "D wor")]
    assert got == [("hel", False), ("lo ", False), ("", False), ("", True)]
    m = StopMatcher(["ab"])
    assert m.feed("xa") == ("x", False) and m.feed("c") == ("ac", False) and m.flush() == ""

    async def body(server, base):
        async with aiohttp.ClientSession() as s:
            p = {"prompt": "abc", "stream": False, "options": {"num_predict": 24, "temperature": 0}}
            full = await (await s.post(base + "/api/generate", json=p)).json()
            text = full["response"]
            assert len(text) >= 6, text
            stop = text[len(text) // 2: len(text) // 2 + 2]            # two characters from the middle
            cut = text[:text.find(stop)]
            a = await (await s.post(base + "/api/generate", json=dict(p, options=dict(p["options"], stop=[stop])))).json()
            lines = []
            async with s.post(base + "/api/generate", json=dict(p, stream=True, stop=[stop])) as r:
                async for raw in r.content:
                    lines.append(json.loads(raw))
            o = await (await s.post(base + "/v1/completions", json={"prompt": "abc", "max_tokens": 24, "temperature": 0,
                                                                     "stop": stop})).json()
            for _ in range(50):                                      # the aborts are processed by the engine thread
                if server.aengine.num_streams == 0:
                    break
                await asyncio.sleep(0.05)
            return full, cut, a, lines, o, server.aengine.num_streams
    full, cut, a, lines, o, open_streams = with_server(body)
    assert a["response"] == cut and a["done_reason"] == "stop" and a["eval_count"] < full["eval_count"]
    assert "".join(l["response"] for l in lines) == cut and lines[-1]["done"] and lines[-1]["done_reason"] == "stop"
    assert o["choices"][0]["text"] == cut and o["choices"][0]["finish_reason"] == "stop"
    assert o["usage"]["completion_tokens"] < full["eval_count"]
    assert open_streams == 0


def test_ollama_chat_endpoint_streams_message_pieces():
    """`/api/chat` — the endpoint `langchain_ollama.ChatOllama` of the reference's notebooks talks to
    (`/root/reference/notebooks/llm_requests.ipynb:53-57`): NDJSON lines with a `message` piece each, a final `done`
    record with the timing tail; non-streamed form; same tokens as `/api/generate` on the flattened prompt."""
    async def body(server, base):
        async with aiohttp.ClientSession() as s:
            msgs = [{"role": "system", "content": "be brief"}, {"role": "user", "content": "hi"}]
            p = {"model": "m", "messages": msgs, "options": {"num_predict": 6, "temperature": 0}}
            lines = []
            async with s.post(base + "/api/chat", json=p) as r:
                assert r.status == 200 and r.headers["Content-Type"].startswith("application/x-ndjson")
                async for raw in r.content:
                    lines.append(json.loads(raw))
            whole = await (await s.post(base + "/api/chat", json=dict(p, stream=False))).json()
            flat = await (await s.post(base + "/api/generate", json={"prompt": "system: be brief\nuser: hi\nassistant: ", "stream": False,
                                                                     "options": p["options"]})).json()
            bad = await s.post(base + "/api/chat", json={"messages": "hi"})
            return lines, whole, flat, bad.status
    lines, whole, flat, bad = with_server(body)
    assert len(lines) == 7 and all(l["message"]["role"] == "assistant" and not l["done"] for l in lines[:-1])
    tail = lines[-1]
    assert tail["done"] and tail["done_reason"] == "length" and tail["eval_count"] == 6 and "context" not in tail
    text = "".join(l["message"]["content"] for l in lines)
    assert whole["message"]["content"] == text == flat["response"] and whole["done"] and bad == 400


def test_openai_completions_sse_and_usage_and_chat():
    async def body(server, base):
        async with aiohttp.ClientSession() as s:
            p = {"model": "m", "prompt": "hi there", "max_tokens": 4, "temperature": 0, "stream": True,
                 "stream_options": {"include_usage": True}}
            async with s.post(base + "/v1/completions", json=p) as r:
                assert r.headers["Content-Type"].startswith("text/event-stream")
                raw = (await r.read()).decode()
            ns = await (await s.post(base + "/v1/completions", json={"prompt": "hi there", "max_tokens": 4, "temperature": 0})).json()
            chat = await (await s.post(base + "/v1/chat/completions",
                                       json={"messages": [{"role": "user", "content": "yo"}], "max_tokens": 3})).json()
            models = await (await s.get(base + "/v1/models")).json()
            tags = await (await s.get(base + "/api/tags")).json()
            return raw, ns, chat, models, tags
    raw, ns, chat, models, tags = with_server(body)
    events = [e for e in raw.split("\n\n") if e]
    assert events[-1] == "data: [DONE]" and all(e.startswith("data: ") for e in events)
    chunks = [json.loads(e[6:]) for e in events[:-1]]
    assert len(chunks) == 5 and chunks[-1]["usage"]["completion_tokens"] == 4
    assert chunks[3]["choices"][0]["finish_reason"] == "length" and chunks[0]["choices"][0]["finish_reason"] is None
    assert "".join(c["choices"][0]["text"] for c in chunks[:4]) == ns["choices"][0]["text"]      # greedy: same text
    assert ns["usage"] == {"prompt_tokens": 9, "completion_tokens": 4, "total_tokens": 13}
    assert chat["choices"][0]["message"]["role"] == "assistant" and chat["usage"]["completion_tokens"] == 3
    assert models["data"][0]["id"] == "tiny-opt" and tags["models"][0]["name"] == "tiny-opt"


def test_metrics_health_and_abort_on_disconnect():
    async def body(server, base):
        async with aiohttp.ClientSession() as s:
            # start a long generation and drop the connection after the first line
            r = await s.post(base + "/api/generate", json={"prompt": "x" * 200, "max_tokens": 400, "ignore_eos": True})
            await r.content.readline()
            r.close()
            await asyncio.sleep(0.3)
            for _ in range(50):
                st = await (await s.get(base + "/stats")).json()
                if st["num_running"] == 0 and st["num_waiting"] == 0:
                    break
                await asyncio.sleep(0.05)
            await (await s.post(base + "/api/generate", json={"prompt": "ok", "max_tokens": 2, "stream": False})).json()
            health = await (await s.get(base + "/health")).json()
            prom = await (await s.get(base + "/metrics")).text()
            st = await (await s.get(base + "/stats")).json()
            return st, health, prom
    st, health, prom = with_server(body)
    assert st["requests_aborted"] == 1 and st["num_running"] == 0 and st["kv_used_frac"] < 0.05
    assert health["status"] == "ok"
    assert "dli_requests_total 2" in prom and "dli_time_to_first_token_seconds_bucket" in prom and "dli_kv_cache_usage" in prom


@pytest.mark.parametrize("mode", ["thread", "process"])
def test_harness_replay_trace1_against_first_party_server(tmp_path, mode):
    """BASELINE.json config 1 (with a tiny OPT so it runs in seconds)."""
    import synth
    pool_path = tmp_path / "conversations.json"
    synth.make_conversations(str(pool_path), n=150, seed=4)
    data = tg.DataLoader().get_data_from_path(str(pool_path))
    sched = tg.Scheduler().get_schedule_from_trace(os.path.join(ROOT, "data", "trace1.csv"), 100)
    sched["Timestamp"] = sched["Timestamp"] * 0.05

    async def body(server, base):
        cfg = dict(tg.config, url=base + "/api/generate", max_tokens=12)
        logger = tg.MetricCollector()
        gen = tg.TrafficGenerator(data, sched, cfg, logger)
        await gen.issue_queries()
        return logger

    logger = with_server(body, mode=mode)
    m = logger.metrics
    assert len(m) == 6 and all(e["success"] for e in m.values())
    keys = ["number_of_input_tokens", "request_start_time", "response_headers_received_time",
            "first_token_arrive_time", "response_end_time", "scheduled_start_time", "success"]
    assert all(list(e.keys()) == keys for e in m.values())
    # headers are acknowledged at admission, before the first token exists
    assert all(e["response_headers_received_time"] <= e["first_token_arrive_time"] for e in m.values())
    path = tmp_path / "log.json"
    logger.save(str(path))
    num = r"-?\d+(?:\.\d+)?(?:e-?\d+)?"
    ref = "/root/reference/logs/log.json"
    if os.path.exists(ref):
        assert re.sub(num, "N", path.read_text()) == re.sub(num, "N", open(ref).read())
    n_out = analyze.n_out_from_sidecar({str(k): v for k, v in logger.sidecar.items()})
    assert set(n_out.values()) == {12}                                # 12 token lines + done line
    # generous thresholds: this checks the plumbing of the analysis, not the speed of a loaded CI box
    s = analyze.summarize(json.loads(path.read_text()), n_out, slo_ttft=120.0, slo_tpot=30.0)
    assert s["success"] == 6 and s["slo_attainment"] == 1.0 and s["tpot"]["p50"] is not None
    # continuous batching: the six requests overlap instead of queueing one behind the other
    ends = sorted(e["response_end_time"] for e in m.values())
    starts = sorted(e["first_token_arrive_time"] for e in m.values())
    assert starts[-1] < ends[0] + 30.0


def test_overload_is_shed_with_503_and_retry_after():
    """`max_pending_requests`: past that many requests waiting for their first token the server answers 503 +
    Retry-After at the door (the harness records it as a failed request via raise_for_status,
    `/root/reference/traffic_generator/main.py:258,269-270`) instead of queueing without bound."""
    async def body(server, base):
        server.max_pending_requests = 2
        async with aiohttp.ClientSession() as s:
            async def one(i):
                async with s.post(base + "/api/generate", json={"model": "m", "prompt": "hello " * 40, "max_tokens": 6}) as r:
                    if r.status != 200:
                        return r.status, r.headers.get("Retry-After")
                    return r.status, len([l async for l in r.content])
            res = await asyncio.gather(*(one(i) for i in range(12)))
            ok = [r for r in res if r[0] == 200]
            shed = [r for r in res if r[0] == 503]
            assert len(ok) >= 2 and len(shed) >= 1 and len(ok) + len(shed) == 12
            assert all(r[1] == "1" for r in shed) and all(r[1] == 7 for r in ok)
            async with s.get(base + "/stats") as r:
                st = await r.json()
            assert st["shed_503"] == len(shed) and st["pending_first_token"] == 0
    with_server(body, mode="thread")


def test_engine_sheds_requests_that_waited_too_long():
    """`max_queue_delay`: a request not admitted in time is finished with reason "shed" — its KV reservation is
    released and its client gets a final record instead of hanging (bounded deferral)."""
    from dli_amd.config import CacheConfig, EngineConfig, SamplingParams, SchedulerConfig, get_model_config
    from dli_amd.engine.engine import LLMEngine
    import time
    cfg = EngineConfig(model=get_model_config("tiny-llama"), cache=CacheConfig(num_blocks=64),
                       sched=SchedulerConfig(max_num_seqs=1, max_num_batched_tokens=64, max_model_len=256, max_queue_delay=0.05),
                       device="cpu", use_hip_graphs=False)
    eng = LLMEngine(cfg)
    p = SamplingParams(temperature=0.0, max_tokens=30, ignore_eos=True)
    for i in range(4):
        eng.add_request(f"r{i}", prompt_ids=list(range(3, 40)), params=p)
    finished = {}
    t0 = time.time()
    while eng.has_unfinished() and time.time() - t0 < 60:
        for o in eng.step():
            if o.finished:
                finished[o.req_id] = o.finish_reason
        time.sleep(0.005)
    assert finished["r0"] == "length"
    assert sorted(finished) == ["r0", "r1", "r2", "r3"] and list(finished.values()).count("shed") >= 2
    assert eng.metrics.requests_shed == list(finished.values()).count("shed")
    assert eng.bm.num_free == eng.bm.num_blocks and not eng.requests


def test_openai_n_choices_share_one_prefill():
    """`n` on /v1/completions: n choices (indexed) from one request, streamed and not."""
    async def body(server, base):
        async with aiohttp.ClientSession() as s:
            req = {"model": "m", "prompt": "tell me three things " * 5, "max_tokens": 6, "temperature": 0.9, "n": 3, "seed": 4}
            async with s.post(base + "/v1/completions", json=req) as r:
                assert r.status == 200
                out = await r.json()
            assert [c["index"] for c in out["choices"]] == [0, 1, 2]
            assert out["usage"]["completion_tokens"] == 18
            async with s.post(base + "/v1/completions", json=dict(req, stream=True)) as r:
                chunks = [json.loads(l[6:]) async for l in r.content if l.startswith(b"data: {")]
            per = {}
            for c in chunks:
                ch = c["choices"][0]
                per.setdefault(ch["index"], []).append(ch["finish_reason"])
            assert sorted(per) == [0, 1, 2] and all(v[-1] == "length" and len(v) == 6 for v in per.values())
            async with s.post(base + "/v1/completions", json=dict(req, n=99)) as r:
                assert r.status == 400
            async with s.get(base + "/stats") as r:
                st = await r.json()
            assert st["num_running"] == 0 and st["pending_first_token"] == 0
    with_server(body, mode="thread")


def _replay_http(tmp_path, extra, timeout=600, env_extra=None):
    """benchmarks/replay_http.py: api_server (own process, engine core + TP workers in further processes) ← the preserved
    harness run as a script from its directory (contract I1) → log.json + summary."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    out_dir = str(tmp_path / "replay")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "replay_http.py"), "--port", str(port), "--trace",
                        "../data/trace1.csv", "--max-trace", "6", "--qps-scale", "20", "--max-tokens", "8", "--out-dir", out_dir]
                       + extra, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    brief = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    return brief, open(os.path.join(out_dir, "log.json")).read()


def test_http_replay_against_a_tp2_server_goes_through_the_one_bootstrap(tmp_path):
    """`api_server --tp 2` joins its ranks through `launch.bootstrap` (here: CPU, gloo), the harness
    replays trace1.csv over HTTP against it, log.json keeps the reference's byte pattern and /stats says what the
    group runs on — the same fields as the bench JSON line."""
    brief, log_text = _replay_http(tmp_path, ["--model", "tiny-llama", "--device", "cpu", "--tp", "2"])
    assert brief["success"] == 6 and brief["tp"] == 2
    assert brief["comm"]["tp"] == 2 and brief["comm"]["device_backend"] == "gloo" and brief["comm"]["custom_ar_enabled"] is False
    num = r"-?\d+(?:\.\d+)?(?:e-?\d+)?"
    ref = "/root/reference/logs/log.json"
    if os.path.exists(ref):
        assert re.sub(num, "N", log_text) == re.sub(num, "N", open(ref).read())
