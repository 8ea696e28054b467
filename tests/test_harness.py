"""Harness unit + integration tests (SURVEY.md §4 items 1–2).

The oracle for the prompt matcher is a literal, slow transcription of the *documented rule*
(nearest len_prompt, then nearest len_output, ties → shorter, row 0 never a donor), not a copy of
the reference routine.  The byte-format fixture is /root/reference/logs/log.json (schema only).
"""
import asyncio
import json
import os
import re
import sys

import numpy as np
import pandas as pd
import pytest

import main as tg
import analyze
import synth
from fake_server import FakeOllama, start

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SUCCESS_KEYS = ["number_of_input_tokens", "request_start_time", "response_headers_received_time",
                "first_token_arrive_time", "response_end_time", "scheduled_start_time", "success"]


# ------------------------------------------------------------------ users / schedule
def test_steady_user_inclusive_end_and_offset():
    u = tg.SteadyUser("u", req_freq=1.0, duration=10.0, delay_start=0.3)
    ts = u.get_timestamps()
    assert len(ts) == 11 and ts[0] == 0.3 and abs(ts[-1] - 10.3) < 1e-9


def test_steady_user_float_accumulation():
    # 0.1 added three times is 0.30000000000000004 > 0.3 → the "inclusive" end stamp is lost
    ts = tg.SteadyUser("u", req_freq=10.0, duration=0.3).get_timestamps()
    assert len(ts) == 3


def test_burst_user():
    assert tg.BurstUser("b", n_req=5, time=2.5).get_timestamps() == [2.5] * 5


def test_schedule_from_users_columns():
    users = [tg.SteadyUser("a", 1.0, 2.0), tg.BurstUser("b", 3, 1.5)]
    df = tg.Scheduler().get_schedule_from_users(users)
    assert list(df.columns) == ["Timestamp", "Request tokens", "Response tokens", "User"]
    assert len(df) == 6 and set(df["Request tokens"]) == {500} and set(df["Response tokens"]) == {500}
    assert list(df.index) == list(range(6))


def test_trace_parse_reference_fixture():
    df = tg.Scheduler().get_schedule_from_trace(os.path.join(ROOT, "data", "trace1.csv"), 100)
    assert list(df.columns) == ["Timestamp", "Request tokens", "Response tokens"]
    assert df["Timestamp"].dtype == float and df["Request tokens"].dtype == np.int64
    assert df["Request tokens"].tolist() == [472, 1087, 417, 1360, 185, 586]
    assert len(tg.Scheduler().get_schedule_from_trace(os.path.join(ROOT, "data", "trace1.csv"), 2)) == 2


def test_trace_extra_columns_tolerated(tmp_path):
    df = synth.make_burstgpt(str(tmp_path / "b.csv"), n=50, seed=3)
    got = tg.Scheduler().get_schedule_from_trace(str(tmp_path / "b.csv"), 20)
    assert len(got) == 20 and "Total tokens" in got.columns
    assert (np.diff(df["Timestamp"]) >= 0).all()


# ------------------------------------------------------------------ nearest fill
def _naive_fill(arr, missing):
    src = list(arr)
    valid = [i for i, v in enumerate(src) if v != missing]
    if not valid:
        return src
    out = []
    for i, v in enumerate(src):
        if v != missing:
            out.append(v)
            continue
        best = min(valid, key=lambda j: (abs(j - i), j))       # ties → left
        out.append(src[best])
    return out


@pytest.mark.parametrize("seed", range(8))
def test_fill_missing_matches_rule(seed):
    rng = np.random.default_rng(seed)
    arr = np.where(rng.random(40) < 0.7, -1, rng.integers(0, 99, 40))
    want = _naive_fill(arr.tolist(), -1)
    got = arr.copy()
    assert tg.Query._fill_missing_idx(got, -1) is None
    assert got.tolist() == want
    as_list = arr.tolist()
    tg.Query._fill_missing_idx(as_list, -1)
    assert as_list == want


def test_fill_missing_tie_goes_left_and_all_missing_untouched():
    a = np.array([7, -1, -1, -1, 9])
    tg.Query._fill_missing_idx(a, -1)
    assert a.tolist() == [7, 7, 7, 9, 9]
    b = np.array([-1, -1, -1])
    tg.Query._fill_missing_idx(b, -1)
    assert b.tolist() == [-1, -1, -1]


def _pool(pairs):
    return [(f"p{i}", lp, lo, "o") for i, (lp, lo) in enumerate(pairs)]


def _rule_match(pool, want_in, want_out, cap=1024):
    want_in, want_out = min(want_in, cap), min(want_out, cap)
    ok = [(i, s) for i, s in enumerate(pool) if s[1] <= cap and s[2] <= cap]
    rows = sorted({s[1] for _, s in ok})
    donors = [r for r in rows if r != 0] or rows
    if want_in in rows:
        row = want_in
    else:
        row = min(donors, key=lambda r: (abs(r - want_in), r))
    in_row = {}
    for i, s in ok:
        if s[1] == row:
            in_row[s[2]] = i                                    # last writer wins
    lo = min(in_row, key=lambda o: (abs(o - want_out), o))
    return in_row[lo]


def test_query_matching_rule_random():
    rng = np.random.default_rng(0)
    pairs = [(int(a), int(b)) for a, b in zip(rng.integers(1, 1300, 60), rng.integers(1, 1300, 60))]
    pairs += [(100, 50), (100, 50), (100, 70)]                  # duplicate → last wins
    pool = _pool(pairs)
    sched = pd.DataFrame({"Timestamp": np.arange(200, dtype=float),
                          "Request tokens": rng.integers(1, 2000, 200),
                          "Response tokens": rng.integers(1, 2000, 200)})
    q = tg.Query(pool, sched)
    assert q.prefill_idx.shape == (1025, 1025)
    for i in range(len(q)):
        prompt, lp, lo, qid, at = q.get_query()
        want = _rule_match(pool, int(sched.at[i, "Request tokens"]), int(sched.at[i, "Response tokens"]))
        assert (prompt, lp, lo) == pool[want][:3]
        assert qid == i and at == float(i)


def test_query_sorts_clamps_and_resets():
    pool = _pool([(1015, 200), (476, 30), (2000, 10)])          # third sample is over the cap → ignored
    sched = pd.DataFrame({"Timestamp": [5.0, 0.0], "Request tokens": [5000, 470],
                          "Response tokens": [9000, 18]})
    q = tg.Query(pool, sched)
    first = q.get_query()
    assert first[1:] == [476, 30, 0, 0.0]                       # sorted by Timestamp
    second = q.get_query()
    assert second[1:] == [1015, 200, 1, 5.0]                    # clamped to 1024 → nearest 1015
    q.reset()
    assert q.get_query()[3] == 0 and len(q) == 2


def test_query_row_zero_never_donates():
    pool = _pool([(0, 5), (10, 6)])
    sched = pd.DataFrame({"Timestamp": [0.0, 1.0], "Request tokens": [3, 0], "Response tokens": [5, 5]})
    q = tg.Query(pool, sched)
    assert q.get_query()[1] == 10                               # 3 is nearer to 0, but 0 cannot donate
    assert q.get_query()[1] == 0                                # exact row 0 still matches itself


def test_dataloader_roundtrip(tmp_path):
    pool = synth.make_conversations(str(tmp_path / "c.json"), n=20, seed=5)
    data = tg.DataLoader().get_data_from_path(str(tmp_path / "c.json"))
    assert len(data) == len(pool)
    prompt, lp, lo, out = data[0]
    assert lp == len(prompt.encode()) + 1 and isinstance(lo, int) and isinstance(out, str)


def test_multiturn_prompts_share_prefix():
    pool = list(synth.make_conversations(n=40, seed=2, multi_turn=True).values())
    shared = sum(1 for a, b in zip(pool, pool[1:]) if b["prompt"].startswith(a["prompt"]))
    assert shared >= 10 and all(p["len_prompt"] <= 1024 for p in pool)


# ------------------------------------------------------------------ replay against the fake server
def _run(fake, n=6, max_tokens=5, extra_cfg=None):
    async def go():
        runner, url = await start(fake)
        try:
            pool = _pool([(476, 18), (1015, 230), (427, 276), (182, 215), (581, 293)])
            sched = tg.Scheduler().get_schedule_from_trace(os.path.join(ROOT, "data", "trace1.csv"), n)
            sched["Timestamp"] = sched["Timestamp"] * 0.02
            cfg = dict(tg.config, url=url, max_tokens=max_tokens, **(extra_cfg or {}))
            logger = tg.MetricCollector()
            gen = tg.TrafficGenerator(pool, sched, cfg, logger)
            await gen.issue_queries()                           # contract I2: awaitable in a running loop
            return logger
        finally:
            await runner.cleanup()
    return asyncio.run(go())


def test_replay_success_schema_order_and_bytes(tmp_path, capsys):
    fake = FakeOllama(token_delay=0.001)
    logger = _run(fake)
    m = logger.metrics
    assert list(m.keys()) == list(range(6))
    for qid, e in m.items():
        assert list(e.keys()) == SUCCESS_KEYS                   # insertion order (contract I7)
        assert e["success"] is True
        assert e["request_start_time"] <= e["response_headers_received_time"] <= e["response_end_time"]
        assert e["first_token_arrive_time"] <= e["response_end_time"]
        assert e["request_start_time"] >= e["scheduled_start_time"] >= 0
    assert [m[i]["number_of_input_tokens"] for i in range(6)] == [476, 1015, 427, 1015, 182, 581]

    path = tmp_path / "log.json"
    logger.save(str(path))
    raw = path.read_bytes()
    assert not raw.endswith(b"\n")
    ref = open("/root/reference/logs/log.json").read() if os.path.exists("/root/reference/logs/log.json") else None
    num = r"-?\d+(?:\.\d+)?(?:e-?\d+)?"
    shape = lambda s: re.sub(num, "N", s)
    if ref is not None:
        assert shape(raw.decode()) == shape(ref)                # same keys, order, separators, bools
    assert json.loads(raw)["0"]["success"] is True

    body = fake.seen[0]                                         # contract I5
    assert list(body.keys()) == ["model", "prompt", "temperature", "max_tokens", "stream"]
    assert body["stream"] is True and body["max_tokens"] == 5 and body["model"] == "mistral"

    out = capsys.readouterr().out                               # contract I8
    assert "[START] ID: 0, Start: 0.0" in out and re.search(r"\[END\] ID: 5, End: \d+\.\d, turnaround: \d+\.\d", out)
    assert "Request tokens" in out
    assert logger.sidecar[0]["n_lines"] == 6                    # 5 tokens + done line


def test_replay_failures_are_null_and_run_continues(capsys):
    fake = FakeOllama(token_delay=0.001, fail_500={1}, reset_before_headers={2}, reset_after_first={3})
    m = _run(fake).metrics
    assert m[0]["success"] and m[4]["success"] and m[5]["success"]
    assert m[1]["success"] is False and m[1]["response_end_time"] is None and m[1]["first_token_arrive_time"] is None
    assert m[2]["success"] is False and m[2]["response_headers_received_time"] is None
    assert m[3]["success"] is False and m[3]["response_end_time"] is None
    assert json.loads(json.dumps(m))["2"]["response_headers_received_time"] is None
    out = capsys.readouterr().out
    assert "ClientResponseError" in out and "[ERROR] ID: 2, Request Exception" in out


def test_replay_timeout_recorded_not_fatal():
    fake = FakeOllama(token_delay=0.2, n_tokens=50)
    m = _run(fake, n=2, extra_cfg={"timeout": 0.3}).metrics
    assert all(e["success"] is False for e in m.values())


def test_serial_server_shows_head_of_line_blocking():
    fake = FakeOllama(token_delay=0.004, n_tokens=10, serial=True)
    m = _run(fake).metrics
    ttft = [m[i]["first_token_arrive_time"] - m[i]["request_start_time"] for i in range(6)]
    assert ttft[5] > ttft[0] * 3                                # queueing grows with position


# ------------------------------------------------------------------ overrides / analysis
def test_overrides_default_is_identity_and_flags_apply():
    assert tg.apply_overrides(tg.config, argv=[], environ={}) == tg.config
    got = tg.apply_overrides(tg.config, argv=["--url", "http://x/y", "--max-trace", "7"],
                             environ={"TG_MODEL": "llama3", "TG_MAX_TRACE": "3"})
    assert got["url"] == "http://x/y" and got["max_trace"] == 7 and got["model"] == "llama3"
    assert set(tg.config) == {"trace_path", "data_path", "max_trace", "url", "no_proxy", "model",
                              "temperature", "max_tokens", "save_log", "log_path"}
    assert (tg.MAX_PROMPT_LEN, tg.MAX_GEN_LEN, tg.STREAM) == (1024, 1024, True)


def test_analyze_on_reference_log():
    path = "/root/reference/logs/log.json"
    if not os.path.exists(path):
        pytest.skip("reference tree not mounted")
    s = analyze.summarize(json.load(open(path)), fixed_n_out=None, slo_ttft=2.0, slo_tpot=0.1)
    assert s["success"] == 6 and abs(s["makespan_s"] - 38.74) < 0.01
    assert abs(s["ttft"]["mean"] - 17.31) < 0.02 and abs(s["ttft"]["p50"] - 18.86) < 0.02
    assert abs(s["request_throughput"] - 0.155) < 0.001
    assert abs(s["slo_goodput"] - 0.026) < 0.001                # 1 of 6 within TTFT ≤ 2 s


def test_analyze_tpot_from_sidecar():
    metrics = {"0": {"number_of_input_tokens": 10, "request_start_time": 0.0,
                     "response_headers_received_time": 0.1, "first_token_arrive_time": 0.2,
                     "response_end_time": 1.2, "scheduled_start_time": 0.0, "success": True}}
    n_out = analyze.n_out_from_sidecar({"0": {"n_lines": 12}})
    s = analyze.summarize(metrics, n_out, slo_ttft=0.5, slo_tpot=0.2)
    assert n_out == {"0": 11} and abs(s["tpot"]["mean"] - 0.1) < 1e-9 and s["slo_attainment"] == 1.0


def test_fresh_clone_invocation_writes_default_log_path(tmp_path):
    """Contract I1 end to end: `cd traffic_generator && python main.py` on a tree WITHOUT a logs/ directory must still
    leave `../logs/log.json` behind (the reference ships the directory, `/root/reference/logs/log.json`; its default
    `log_path` is relative to it, `/root/reference/traffic_generator/main.py:312,343`), byte-pattern-identical to the
    reference's sample."""
    import shutil
    import socket
    import subprocess
    import time
    import synth
    tree = tmp_path / "clone"
    shutil.copytree(os.path.join(ROOT, "traffic_generator"), tree / "traffic_generator",
                    ignore=shutil.ignore_patterns("__pycache__"))
    os.makedirs(tree / "data")
    shutil.copy(os.path.join(ROOT, "data", "trace1.csv"), tree / "data" / "trace1.csv")
    synth.make_conversations(str(tree / "data" / "conversations.json"), n=200, seed=3)
    assert not (tree / "logs").exists()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    tgd = str(tree / "traffic_generator")
    srv = subprocess.Popen([sys.executable, "fake_server.py", "--port", str(port), "--n-tokens", "5"], cwd=tgd)
    try:
        for _ in range(100):
            try:
                socket.create_connection(("127.0.0.1", port), timeout=0.2).close()
                break
            except OSError:
                time.sleep(0.1)
        p = subprocess.run([sys.executable, "main.py", "--url", f"http://127.0.0.1:{port}/api/generate"], cwd=tgd,
                           capture_output=True, text=True, timeout=120)
    finally:
        srv.terminate()
        srv.wait(10)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "[START] ID: 0" in p.stdout and "[END] ID: 5" in p.stdout
    out = (tree / "logs" / "log.json").read_text()
    num = r"-?\d+(?:\.\d+)?(?:e-?\d+)?"
    ref = "/root/reference/logs/log.json"
    if os.path.exists(ref):
        assert re.sub(num, "N", out) == re.sub(num, "N", open(ref).read())
    # the tracked sample (first-party CPU server on the same trace) has the same shape
    assert re.sub(num, "N", out) == re.sub(num, "N", open(os.path.join(ROOT, "logs", "log.json")).read())
