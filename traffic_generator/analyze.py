"""Derived serving metrics from a harness ``log.json`` (+ optional side-car).

The reference records raw per-request clocks only (`/root/reference/traffic_generator/main.py:274-277`);
it computes nothing from them.  The definitions here are the ones pinned in BASELINE.md §2:

    TTFT       = first_token_arrive_time − request_start_time
    ack        = response_headers_received_time − request_start_time
    E2E        = response_end_time − request_start_time
    client lag = request_start_time − scheduled_start_time
    TPOT       = (response_end_time − first_token_arrive_time) / (n_out − 1)
    goodput    = #{success ∧ TTFT ≤ SLO_TTFT ∧ TPOT ≤ SLO_TPOT} / max(response_end_time)

``n_out`` is not in ``log.json`` (contract I7 forbids adding keys), so it comes from the side-car
written by ``main.py --sidecar-path`` (streamed line count; NDJSON streams carry one trailing
``done`` line, SSE streams carry blank separators — see ``--lines-per-token`` /
``--trailer-lines``) or from ``--fixed-n-out`` when the server generates exactly ``max_tokens``.

CLI: ``python analyze.py ../logs/log.json [--sidecar ../logs/sidecar.json] [--slo-ttft 2 --slo-tpot 0.1]``
"""
from __future__ import annotations

import argparse
import json

import numpy as np

SLOS = {
    "llama3-8b-interactive": (0.5, 0.030),
    "llama3-8b-relaxed": (2.0, 0.100),
    "llama3-70b-interactive": (0.45, 0.040),
    "llama3-70b-relaxed": (2.0, 0.200),
    "mixtral-8x7b": (2.0, 0.200),
}


def _pct(xs, q):
    xs = [x for x in xs if x is not None]
    return float(np.percentile(xs, q)) if xs else None      # linear interpolation (numpy default)


def _stats(xs):
    xs = [x for x in xs if x is not None]
    if not xs:
        return {"mean": None, "p50": None, "p99": None, "max": None}
    return {"mean": float(np.mean(xs)), "p50": _pct(xs, 50), "p99": _pct(xs, 99), "max": float(np.max(xs))}


def per_request(metrics: dict, n_out: dict | None = None, fixed_n_out: int | None = None) -> list[dict]:
    rows = []
    for qid, m in metrics.items():
        ok = bool(m.get("success"))
        row = {"id": str(qid), "success": ok, "n_in": m.get("number_of_input_tokens")}
        start = m.get("request_start_time")
        if ok and start is not None:
            first, end = m["first_token_arrive_time"], m["response_end_time"]
            row["ttft"] = first - start
            row["e2e"] = end - start
            row["stream"] = end - first
            hdr = m.get("response_headers_received_time")
            row["ack"] = None if hdr is None else hdr - start
            row["lag"] = start - m["scheduled_start_time"]
            row["end"] = end
            n = fixed_n_out if fixed_n_out is not None else (n_out or {}).get(str(qid))
            row["n_out"] = n
            row["tpot"] = (end - first) / (n - 1) if n and n > 1 else (0.0 if n == 1 else None)
        rows.append(row)
    return rows


def summarize(metrics: dict, n_out: dict | None = None, fixed_n_out: int | None = None,
              slo_ttft: float = 2.0, slo_tpot: float = 0.1) -> dict:
    rows = per_request(metrics, n_out, fixed_n_out)
    good = [r for r in rows if r["success"] and "ttft" in r]
    makespan = max((r["end"] for r in good), default=0.0)
    have_tpot = [r for r in good if r.get("tpot") is not None]

    def meets(r):
        if r["ttft"] > slo_ttft:
            return False
        return r.get("tpot") is None or r["tpot"] <= slo_tpot

    n_meet = sum(1 for r in good if meets(r))
    out = {
        "requests": len(rows),
        "success": len(good),
        "makespan_s": makespan,
        "request_throughput": len(good) / makespan if makespan else None,
        "input_token_throughput": sum(r["n_in"] or 0 for r in good) / makespan if makespan else None,
        "output_token_throughput": (sum(r["n_out"] for r in have_tpot) / makespan
                                    if makespan and have_tpot else None),
        "ttft": _stats([r["ttft"] for r in good]),
        "tpot": _stats([r["tpot"] for r in have_tpot]),
        "e2e": _stats([r["e2e"] for r in good]),
        "stream": _stats([r["stream"] for r in good]),
        "ack": _stats([r["ack"] for r in good]),
        "client_lag": _stats([r["lag"] for r in good]),
        "slo": {"ttft": slo_ttft, "tpot": slo_tpot, "tpot_known": len(have_tpot)},
        "slo_attainment": n_meet / len(rows) if rows else None,
        "slo_goodput": n_meet / makespan if makespan else None,
    }
    return out


def n_out_from_sidecar(sidecar: dict, lines_per_token: int = 1, trailer_lines: int = 1) -> dict:
    """NDJSON (``/api/generate``): 1 line per token + 1 ``done`` trailer.  SSE
    (``/v1/completions``): 2 lines per event, trailer ``data: [DONE]`` + blank = 2."""
    return {str(k): max((v["n_lines"] - trailer_lines) // lines_per_token, 0) for k, v in sidecar.items()}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("log")
    ap.add_argument("--sidecar")
    ap.add_argument("--fixed-n-out", type=int)
    ap.add_argument("--lines-per-token", type=int, default=1)
    ap.add_argument("--trailer-lines", type=int, default=1)
    ap.add_argument("--slo", choices=sorted(SLOS))
    ap.add_argument("--slo-ttft", type=float, default=2.0)
    ap.add_argument("--slo-tpot", type=float, default=0.1)
    args = ap.parse_args(argv)
    with open(args.log) as fh:
        metrics = json.load(fh)
    n_out = None
    if args.sidecar:
        with open(args.sidecar) as fh:
            n_out = n_out_from_sidecar(json.load(fh), args.lines_per_token, args.trailer_lines)
    ttft, tpot = SLOS[args.slo] if args.slo else (args.slo_ttft, args.slo_tpot)
    print(json.dumps(summarize(metrics, n_out, args.fixed_n_out, ttft, tpot), indent=2))


if __name__ == "__main__":
    main()
