"""Open-loop LLM traffic replayer (trace → streaming HTTP requests → ``log.json``).

This is a fresh implementation of the harness contract documented in SURVEY.md §1.3
(I1–I8).  Parity citations point into the read-only reference tree:

* invocation / config dict ............ /root/reference/traffic_generator/main.py:298-343
* ``SteadyUser`` / ``BurstUser`` ...... main.py:13-37
* ``DataLoader`` ...................... main.py:40-51
* ``Scheduler`` ....................... main.py:53-84
* ``Query`` (prompt matcher) .......... main.py:86-182
* ``MetricCollector`` / ``TraceConfig`` main.py:184-222
* ``TrafficGenerator`` ................ main.py:230-294

Run it exactly like the reference: ``cd traffic_generator && python main.py``.  With no
flags and no ``TG_*`` environment variables the behaviour is driven purely by the in-file
``config`` dict.  Optional overrides (``--url``, ``--trace-path`` … or ``TG_URL`` …) are
layered on top and never change the ``log.json`` schema.

Deliberate deviations from the reference (none alters the I/O schema):

* the exception hook takes the logger from the per-request trace context instead of a module
  global, so it also works when this file is imported (reference bug at main.py:220);
* ``asyncio.TimeoutError`` and other ``aiohttp.ClientError``s (mid-stream payload errors) are
  recorded as a failed request instead of aborting the whole run;
* client timeout / connection limit are configurable (defaults = aiohttp defaults, i.e. what
  the reference silently used);
* an optional *side-car* file (``sidecar_path``) records the number of streamed lines per
  request so TPOT can be derived without adding keys to ``log.json`` (contract I7);
* the default ``url`` points at localhost (the new first-party server binds Ollama's port).
"""
from __future__ import annotations

import argparse
import asyncio
import json
import os
import sys
from time import perf_counter

import aiohttp
import numpy as np
import pandas as pd

# --------------------------------------------------------------------------------------
# Module-level knobs (contract I1: three globals + one dict, edited in place)
# --------------------------------------------------------------------------------------
MAX_PROMPT_LEN = 1024
MAX_GEN_LEN = 1024
STREAM = True

config = {
    'trace_path': '../data/trace1.csv',
    'data_path': '../data/conversations.json',
    'max_trace': 100,
    'url': 'http://127.0.0.1:11434/api/generate',
    'no_proxy': '127.0.0.1',
    'model': 'mistral',
    'temperature': 0.7,
    'max_tokens': 200,
    'save_log': False,
    'log_path': '../logs/log.json'
}

TRACE_COLUMNS = ('Timestamp', 'Request tokens', 'Response tokens')


# --------------------------------------------------------------------------------------
# Synthetic users
# --------------------------------------------------------------------------------------
class SteadyUser:
    """A user firing ``req_freq`` requests per second for ``duration`` seconds.

    The end point is inclusive and the clock is advanced by repeated float addition, which
    is what the reference does (main.py:20-27); both matter for the exact stamp count.
    """

    def __init__(self, name: str, req_freq: float, duration: float, delay_start: float = 0.0):
        self.name = name
        self.req_freq = req_freq
        self.duration = duration
        self.delay_start = delay_start

    def get_timestamps(self) -> list[float]:
        step = 1.0 / self.req_freq
        stamps: list[float] = []
        elapsed = 0.0
        while elapsed <= self.duration:
            stamps.append(self.delay_start + elapsed)
            elapsed += step
        return stamps


class BurstUser:
    """``n_req`` requests that all arrive at the same instant ``time``."""

    def __init__(self, name: str, n_req: int, time: float):
        self.name = name
        self.n_req = n_req
        self.time = time

    def get_timestamps(self) -> list[float]:
        return [self.time for _ in range(self.n_req)]


# --------------------------------------------------------------------------------------
# Inputs: prompt pool and arrival schedule
# --------------------------------------------------------------------------------------
class DataLoader:
    """Reads the ``conversations.json`` prompt pool (contract I4)."""

    def __init__(self, config=None):
        self.config = config

    @staticmethod
    def load_json_from_path(file_path: str):
        with open(file_path, "r") as fh:
            return json.load(fh)

    def get_data_from_path(self, data_path: str) -> list[tuple]:
        records = self.load_json_from_path(data_path)
        pool = []
        for rec in records.values():
            pool.append((rec["prompt"], rec["len_prompt"], rec["len_output"], rec["output"]))
        return pool


class Scheduler:
    """Builds the arrival schedule DataFrame, from a trace CSV or from synthetic users."""

    # fixed lengths used for user-generated schedules (reference main.py:69-70)
    USER_REQUEST_TOKENS = 500
    USER_RESPONSE_TOKENS = 500

    def __init__(self, config=None):
        self.config = config

    def get_schedule_from_trace(self, trace_path: str, max_trace: int) -> pd.DataFrame:
        dtypes = {TRACE_COLUMNS[0]: float, TRACE_COLUMNS[1]: int, TRACE_COLUMNS[2]: int}
        return pd.read_csv(trace_path, nrows=max_trace, dtype=dtypes)

    def get_schedule_from_users(self, users: list) -> pd.DataFrame:
        frames = []
        for user in users:
            stamps = user.get_timestamps()
            n = len(stamps)
            frames.append(pd.DataFrame({
                'Timestamp': stamps,
                'Request tokens': [self.USER_REQUEST_TOKENS] * n,
                'Response tokens': [self.USER_RESPONSE_TOKENS] * n,
                'User': [user.name] * n,
            }))
        return pd.concat(frames).reset_index(drop=True)


class Query:
    """Iterates the schedule and picks, for every row, the pool sample whose
    ``(len_prompt, len_output)`` is nearest to the traced ``(Request, Response)`` lengths.

    Matching rule (reference main.py:131-167): nearest ``len_prompt`` first, then nearest
    ``len_output`` among samples of that prompt length; ties go to the *shorter* length;
    traced lengths are clamped to ``MAX_PROMPT_LEN`` / ``MAX_GEN_LEN``; pool samples longer
    than either limit are ignored; on duplicate ``(len_prompt, len_output)`` the last pool
    entry wins.
    """

    def __init__(self, inputs: list, schedule: pd.DataFrame):
        self.inputs = inputs
        self.schedule = schedule.sort_values(by='Timestamp').reset_index(drop=True)
        self.query_id = -1
        self.query_time = 0
        self.max_prompt_len = MAX_PROMPT_LEN
        self.max_gen_len = MAX_GEN_LEN
        self.prefill_idx = self.get_prefill_idx()

    @staticmethod
    def _fill_missing_idx(arr, missing):
        """In place: replace every ``missing`` entry of 1-D ``arr`` by its nearest
        non-missing neighbour; equidistant → take the left one.  All-missing → untouched.
        """
        n = len(arr)
        if n == 0:
            return
        present = np.asarray(arr) != missing
        if not present.any():
            return
        where = np.arange(n)
        # index of the nearest valid entry at or before i  (-1 → none)
        left_src = np.maximum.accumulate(np.where(present, where, -1))
        # index of the nearest valid entry at or after i   (n → none)
        right_src = np.minimum.accumulate(np.where(present, where, n)[::-1])[::-1]
        far = 2 * n + 1
        d_left = np.where(left_src >= 0, where - left_src, far)
        d_right = np.where(right_src < n, right_src - where, far)
        src = np.where(d_left <= d_right, left_src, right_src)
        filled = np.asarray(arr)[src]
        if isinstance(arr, np.ndarray):
            arr[:] = filled
        else:                       # plain lists are updated element-wise, still in place
            for i in range(n):
                arr[i] = filled[i].item()

    def get_prefill_idx(self):
        n_rows, n_cols = self.max_prompt_len + 1, self.max_gen_len + 1
        table = np.full((n_rows, n_cols), -1, dtype=int)
        have_row = np.zeros(n_rows, dtype=bool)

        for pool_idx, sample in enumerate(self.inputs):
            lp, lo = sample[1], sample[2]
            if lp > self.max_prompt_len or lo > self.max_gen_len:
                continue
            table[lp, lo] = pool_idx
            have_row[lp] = True

        # 1) inside every populated prompt-length row: nearest output length
        for lp in np.flatnonzero(have_row):
            self._fill_missing_idx(table[lp], missing=-1)

        # 2) for unpopulated rows: nearest populated prompt length.  The reference encodes
        #    "row present" as its own (non-zero) index, so row 0 can never act as a donor
        #    (main.py:148-149); keep that.
        donor = have_row * np.arange(n_rows)
        self._fill_missing_idx(donor, missing=0)
        absent = np.flatnonzero(~have_row)
        table[absent] = table[donor[absent]]
        return table

    def get_query(self):
        self.query_id += 1
        row = self.query_id
        self.query_time = self.schedule.at[row, 'Timestamp'].item()

        want_in = min(self.schedule.at[row, 'Request tokens'].item(), self.max_prompt_len)
        want_out = min(self.schedule.at[row, 'Response tokens'].item(), self.max_gen_len)
        sample = self.inputs[self.prefill_idx[want_in][want_out]]

        # [prompt, matched prompt length, matched output length, id, absolute start offset]
        return [sample[0], sample[1], sample[2], self.query_id, self.query_time]

    def reset(self):
        self.query_id = -1
        self.query_time = 0

    def __len__(self):
        return len(self.schedule)


# --------------------------------------------------------------------------------------
# Metrics and aiohttp client tracing
# --------------------------------------------------------------------------------------
class MetricCollector:
    """Per-request timing store.  ``metrics[qid]`` is filled by the trace hooks and by
    ``TrafficGenerator.inference_call``; ``session_start_timestamp`` is attached by the
    driver right before the requests are released (contract I7)."""

    def __init__(self):
        self.trace_config = TraceConfig()
        self.metrics = {}
        self.sidecar = {}

    def save(self, path):
        # the default '../logs/log.json' must work on a fresh clone too (the reference ships its logs/ directory,
        # `/root/reference/logs/log.json`; a missing one would lose the whole replay at the very last line)
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        with open(path, 'w') as fh:
            json.dump(self.metrics, fh)

    def save_sidecar(self, path):
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        with open(path, 'w') as fh:
            json.dump(self.sidecar, fh)


class TraceConfig(aiohttp.TraceConfig):
    """Client-side hooks: request start, headers received, request exception."""

    def __init__(self):
        super().__init__()
        self.on_request_start.append(self.on_request_start_callback)
        self.on_request_end.append(self.on_request_end_callback)
        self.on_request_exception.append(self.on_request_exception_callback)

    @staticmethod
    def _who(ctx):
        rc = ctx.trace_request_ctx
        return rc['logger'], rc['query_id']

    async def on_request_start_callback(self, session, ctx, params):
        logger, query_id = self._who(ctx)
        t = perf_counter() - logger.session_start_timestamp
        logger.metrics[query_id]['request_start_time'] = t
        print(f"[START] ID: {query_id}, Start: {t:.1f}")

    async def on_request_end_callback(self, session, ctx, params):
        # fires once the status line and the headers are in
        logger, query_id = self._who(ctx)
        logger.metrics[query_id]['response_headers_received_time'] = \
            perf_counter() - logger.session_start_timestamp

    async def on_request_exception_callback(self, session, ctx, params):
        logger, query_id = self._who(ctx)
        logger.metrics[query_id]['response_headers_received_time'] = None
        print(f"[ERROR] ID: {query_id}, Request Exception")


# --------------------------------------------------------------------------------------
# Replay driver
# --------------------------------------------------------------------------------------
class TrafficGenerator:
    """Generates LLM inference traffic and sends it to an inference endpoint (open loop:
    request *n* is released at its absolute trace offset regardless of earlier completions)."""

    def __init__(self, data: list, schedule: pd.DataFrame, config: dict, logger: MetricCollector):
        self.queries = Query(inputs=data, schedule=schedule)
        self.config = config
        self.logger = logger

        print(self.queries.schedule)

    def _payload(self, prompt):
        return {
            "model": self.config['model'],
            "prompt": prompt,
            "temperature": self.config['temperature'],
            "max_tokens": self.config['max_tokens'],
            "stream": STREAM
        }

    async def inference_call(self, session, prompt, sleep_time, query_id):
        logger = self.logger
        ctx = {'query_id': query_id, 'logger': logger}
        ok = False
        t_first = None
        t_end = None
        n_lines = 0

        await asyncio.sleep(sleep_time)
        try:
            async with session.post(self.config['url'], json=self._payload(prompt),
                                    trace_request_ctx=ctx) as resp:
                resp.raise_for_status()
                async for _line in resp.content:          # '\n'-delimited iteration
                    if n_lines == 0:
                        t_first = perf_counter() - logger.session_start_timestamp
                    n_lines += 1
            ok = True
            t_end = perf_counter() - logger.session_start_timestamp
            turnaround = t_end - logger.metrics[query_id]['request_start_time']
            print(f"[END] ID: {query_id}, End: {t_end:.1f}, turnaround: {turnaround:.1f}")
        except aiohttp.ClientResponseError as e:
            print(f"ClientResponseError: {e}")
        except aiohttp.ClientConnectionError as e:
            print(f"ClientConnectionError: {e}")
        except aiohttp.ClientError as e:          # e.g. ClientPayloadError on a mid-stream reset
            print(f"ClientError: {e}")
        except asyncio.TimeoutError as e:
            print(f"TimeoutError: {e}")

        entry = logger.metrics[query_id]
        entry['first_token_arrive_time'] = t_first
        entry['response_end_time'] = t_end
        entry['scheduled_start_time'] = sleep_time
        entry['success'] = ok
        logger.sidecar[query_id] = {'n_lines': n_lines}

    def _session_kwargs(self):
        kw = {'trace_configs': [self.logger.trace_config]}
        if self.config.get('timeout') is not None:
            kw['timeout'] = aiohttp.ClientTimeout(total=self.config['timeout'])
        if self.config.get('connector_limit') is not None:
            kw['connector'] = aiohttp.TCPConnector(limit=self.config['connector_limit'])
        return kw

    async def issue_queries(self):
        async with aiohttp.ClientSession(**self._session_kwargs()) as session:
            calls = []
            for _ in range(len(self.queries)):
                prompt, n_in, _n_out, query_id, at = self.queries.get_query()
                calls.append(self.inference_call(session, prompt, at, query_id))
                self.logger.metrics[query_id] = {}
                self.logger.metrics[query_id]['number_of_input_tokens'] = n_in
            self.logger.session_start_timestamp = perf_counter()      # t = 0 for every clock
            await asyncio.gather(*calls)

    def start_profile(self):
        self.queries.reset()
        asyncio.run(self.issue_queries())


# --------------------------------------------------------------------------------------
# Optional overrides (no flags + no TG_* env ⇒ identical to the in-file dict)
# --------------------------------------------------------------------------------------
_OVERRIDABLE = {
    'trace_path': str, 'data_path': str, 'max_trace': int, 'url': str, 'model': str,
    'temperature': float, 'max_tokens': int, 'log_path': str,
    'timeout': float, 'connector_limit': int, 'sidecar_path': str,
}


def apply_overrides(cfg: dict, argv=None, environ=None) -> dict:
    """Returns a copy of ``cfg`` with ``TG_<KEY>`` env vars and ``--key`` flags applied."""
    environ = os.environ if environ is None else environ
    out = dict(cfg)
    for key, typ in _OVERRIDABLE.items():
        env_name = 'TG_' + key.upper()
        if env_name in environ:
            out[key] = typ(environ[env_name])
    parser = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    for key, typ in _OVERRIDABLE.items():
        parser.add_argument('--' + key.replace('_', '-'), dest=key, type=typ, default=None)
    ns = parser.parse_args(sys.argv[1:] if argv is None else argv)
    for key in _OVERRIDABLE:
        val = getattr(ns, key)
        if val is not None:
            out[key] = val
    return out


if __name__ == "__main__":
    config = apply_overrides(config)

    data = DataLoader().get_data_from_path(data_path=config['data_path'])
    schedule = Scheduler().get_schedule_from_trace(trace_path=config['trace_path'],
                                                   max_trace=config['max_trace'])
    logger = MetricCollector()

    # Synthetic-user schedule instead of a trace:
    # users = [SteadyUser(name='u1', req_freq=1.0, duration=10.0),
    #          BurstUser(name='u2', n_req=5, time=2.5)]
    # schedule = Scheduler().get_schedule_from_users(users=users)

    generator = TrafficGenerator(data=data, schedule=schedule, config=config, logger=logger)
    generator.start_profile()

    print(logger.metrics)
    logger.save(path=config['log_path'])
    if config.get('sidecar_path'):
        logger.save_sidecar(config['sidecar_path'])
