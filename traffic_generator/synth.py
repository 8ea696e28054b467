"""Synthetic stand-ins for the two data blobs the reference mirror does not ship
(`/root/reference/.MISSING_LARGE_BLOBS:1-2`): ``data/BurstGPT_1.csv`` and
``data/conversations.json``.

* ``make_burstgpt``   – BurstGPT-shaped CSV (``Timestamp,Model,Request tokens,Response tokens,
  Total tokens,Log Type``; the harness only reads the three columns of contract I3).  Arrivals
  are a Gamma renewal process with shape < 1 (coefficient of variation > 1 → bursty) overlaid
  with occasional dense bursts; token lengths follow heavy-tailed log-normal marginals similar
  to the published BurstGPT statistics (median request ≈ 400 tok, median response ≈ 200 tok).
* ``make_conversations`` – prompt pool per contract I4.  Prompts are ASCII text whose length
  under the stack's byte-level tokenizer (1 token per byte + BOS) is *exactly* ``len_prompt``.
  ``multi_turn=True`` emits sessions whose turn *k* prompt is turn *k-1*'s prompt + answer +
  a new user message, i.e. prompts share long prefixes (exercises prefix caching and the
  router's prefix affinity; BASELINE.json config 3).
* ``make_trace`` – the small ``trace1.csv``-style file described by
  `/root/reference/notebooks/generate_trace.ipynb:11-27` (two groups of ``n_req`` requests, the
  second starting 30 s later, token columns taken from the head of a BurstGPT file).

CLI: ``python synth.py --out-dir ../data`` writes all files.
"""
from __future__ import annotations

import argparse
import json
import os

import numpy as np
import pandas as pd

_WORDS = ("the quick brown fox jumps over lazy dog while distributed inference engines batch "
          "tokens across matrix cores and stream every response line by line to impatient "
          "clients who measure latency goodput throughput fairness under bursty traffic "
          "please summarize explain translate rewrite compare list describe why how what "
          "model cache page block wave lane tile kernel graph reduce gather scatter").split()


def _text_of_length(rng: np.random.Generator, n_bytes: int) -> str:
    """ASCII pseudo-sentence of exactly ``n_bytes`` bytes."""
    if n_bytes <= 0:
        return ""
    parts, size = [], 0
    while size < n_bytes:
        w = _WORDS[int(rng.integers(len(_WORDS)))]
        parts.append(w)
        size += len(w) + 1
    return " ".join(parts)[:n_bytes].ljust(n_bytes, ".")


def _lengths(rng, n, median, sigma, lo, hi):
    x = rng.lognormal(mean=np.log(median), sigma=sigma, size=n)
    return np.clip(np.rint(x), lo, hi).astype(int)


def make_burstgpt(path: str | None = None, n: int = 2000, mean_qps: float = 2.0,
                  burstiness: float = 0.35, seed: int = 0) -> pd.DataFrame:
    """Returns (and optionally writes) a BurstGPT-like trace of ``n`` requests.

    ``burstiness`` is the Gamma shape of the inter-arrival distribution (CV = 1/sqrt(shape));
    the mean inter-arrival is ``1/mean_qps``.
    """
    rng = np.random.default_rng(seed)
    gaps = rng.gamma(shape=burstiness, scale=1.0 / (mean_qps * burstiness), size=n)
    # dense bursts: ~2 % of requests arrive together with their predecessor group
    burst_heads = rng.random(n) < 0.02
    burst_len = rng.integers(4, 24, size=n)
    i = 0
    while i < n:
        if burst_heads[i]:
            gaps[i + 1:i + 1 + burst_len[i]] *= 0.01
            i += burst_len[i]
        i += 1
    stamps = np.cumsum(gaps)
    stamps -= stamps[0]
    if n > 1 and stamps[-1] > 0:
        stamps *= ((n - 1) / mean_qps) / stamps[-1]       # the compressed bursts shorten the trace: restore the mean rate
    req = _lengths(rng, n, median=420, sigma=1.0, lo=4, hi=8000)
    resp = _lengths(rng, n, median=210, sigma=0.9, lo=1, hi=2000)
    df = pd.DataFrame({
        "Timestamp": np.round(stamps, 3),
        "Model": np.where(rng.random(n) < 0.8, "ChatGPT", "GPT-4"),
        "Request tokens": req,
        "Response tokens": resp,
        "Total tokens": req + resp,
        "Log Type": np.where(rng.random(n) < 0.7, "Conversation log", "API log"),
    })
    if path:
        df.to_csv(path, index=False)
    return df


def make_conversations(path: str | None = None, n: int = 3000, seed: int = 1,
                       multi_turn: bool = False, max_len: int = 1024) -> dict:
    """Prompt pool: ``{key: {prompt, len_prompt, len_output, output}}`` (contract I4)."""
    rng = np.random.default_rng(seed)
    pool: dict[str, dict] = {}

    def add(prompt: str, out_len: int):
        pool[str(len(pool))] = {
            "prompt": prompt,
            "len_prompt": len(prompt.encode()) + 1,      # + BOS under the byte tokenizer
            "len_output": int(out_len),
            "output": _text_of_length(rng, max(int(out_len) - 1, 0)),
        }

    if not multi_turn:
        lp = _lengths(rng, n, median=420, sigma=1.0, lo=4, hi=max_len)
        lo = _lengths(rng, n, median=210, sigma=0.9, lo=1, hi=max_len)
        # a coarse deterministic grid guarantees every trace length has a close neighbour
        grid = [(a, b) for a in range(16, max_len + 1, 48) for b in (8, 64, 192, 384, 768, max_len)]
        for a, b in grid:
            add(_text_of_length(rng, a - 1), b)
        for a, b in zip(lp, lo):
            add(_text_of_length(rng, int(a) - 1), int(b))
    else:
        while len(pool) < n:
            history = "system: you are a helpful assistant.\n"
            for _turn in range(int(rng.integers(2, 7))):
                user = "user: " + _text_of_length(rng, int(rng.integers(24, 160))) + "\nassistant: "
                prompt = history + user
                if len(prompt) + 1 > max_len:
                    break
                out_len = int(_lengths(rng, 1, median=120, sigma=0.7, lo=4, hi=400)[0])
                add(prompt, out_len)
                history = prompt + pool[str(len(pool) - 1)]["output"] + "\n"
    if path:
        with open(path, "w") as fh:
            json.dump(pool, fh)
    return pool


def make_trace(burst_df: pd.DataFrame, path: str | None = None, n_req: int = 10,
               gap_s: float = 30.0) -> pd.DataFrame:
    head = burst_df.head(n_req)
    stamps = list(range(n_req)) + [gap_s + i for i in range(n_req)]
    df = pd.DataFrame({
        "Timestamp": stamps,
        "Request tokens": list(head["Request tokens"]) * 2,
        "Response tokens": list(head["Response tokens"]) * 2,
    })
    if path:
        df.to_csv(path, index=False)
    return df


def main(argv=None):
    ap = argparse.ArgumentParser(description="write synthetic BurstGPT_1.csv / conversations.json")
    ap.add_argument("--out-dir", default="../data")
    ap.add_argument("--n-trace", type=int, default=2000)
    ap.add_argument("--n-conv", type=int, default=3000)
    ap.add_argument("--mean-qps", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    os.makedirs(args.out_dir, exist_ok=True)
    burst = make_burstgpt(os.path.join(args.out_dir, "BurstGPT_1.csv"), n=args.n_trace,
                          mean_qps=args.mean_qps, seed=args.seed)
    make_conversations(os.path.join(args.out_dir, "conversations.json"), n=args.n_conv,
                       seed=args.seed + 1)
    make_conversations(os.path.join(args.out_dir, "conversations_multiturn.json"),
                       n=min(args.n_conv, 1200), seed=args.seed + 2, multi_turn=True)
    make_trace(burst, os.path.join(args.out_dir, "trace_2x10.csv"))
    print(f"wrote synthetic data to {args.out_dir}")


if __name__ == "__main__":
    main()
