"""Headline benchmark: SLO goodput (req/s) + p50/p99 TTFT & TPOT under BurstGPT replay, Llama-3-8B
(BASELINE.json ``metric`` / ``configs``), bf16, random-init weights, synthetic prompts of the traced
length distribution.

    python bench.py --gpus 1 --steps 1500 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps K --warmup W

One rank per GPU; every rank is a data-parallel Llama-3-8B replica (BASELINE.json config 3) and
serves its share (round-robin, the router's stateless policy) of ONE global BurstGPT-like trace whose
offered load grows with N (weak scaling: ``--qps`` requests/s per GPU).  Arrivals are replayed in
real time, open loop, exactly like ``traffic_generator/main.py`` does over HTTP: a request becomes
visible to the engine at ``t0 + Timestamp`` whatever the engine is doing.  Lengths are clamped to
1024/1024 like the harness (`/root/reference/traffic_generator/main.py:163-165`); each request
generates exactly its traced ``Response tokens`` (ignore-EOS), so TPOT is well defined.

A *step* is one engine iteration (schedule → forward over the continuous batch → sample → advance).
Untimed: weight init, hipGraph capture, a ramp phase that runs the replay until the engine is in
steady state, then ``--warmup`` steps.  Timed: exactly ``--steps`` steps between
barrier + synchronize brackets; the reported value is the max-over-ranks wall time.

``value`` = whole-job SLO goodput = requests that *finished inside the timed window* with
TTFT ≤ 2 s and TPOT ≤ 100 ms (BASELINE.md §3 "Llama-3-8B relaxed"), summed over ranks, per second.
``vs_baseline`` divides by the only goodput figure the reference tree supports: 0.026 req/s at
TTFT ≤ 2 s for its serial Ollama server (derived from logs/log.json).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "traffic_generator"))

BASELINE_GOODPUT = 0.026          # req/s, BASELINE.md §1 "Replay: goodput @ TTFT ≤ 2 s"
SLO_TTFT, SLO_TPOT = 2.0, 0.100   # BASELINE.md §3, Llama-3-8B relaxed
SLO_TTFT_I, SLO_TPOT_I = 0.5, 0.030


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--qps", type=float, default=float(os.environ.get("DLI_BENCH_QPS", "68")),
                    help="offered load per GPU (requests/s)")
    ap.add_argument("--ramp-seconds", type=float, default=30.0,
                    help="untimed replay before the window: ~2 request lifetimes, so that the in-flight population is steady")
    ap.add_argument("--max-num-seqs", type=int, default=2048)
    ap.add_argument("--max-batched-tokens", type=int, default=4096)
    ap.add_argument("--chunk", type=int, default=2048)
    ap.add_argument("--kv-dtype", default="auto")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--fcfs", action="store_true", help="plain FCFS admission (default: requests past their TTFT deadline yield)")
    ap.add_argument("--sync-steps", action="store_true", help="disable pipelined stepping (one step in flight at a time)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--num-kv-blocks", type=int, default=None, help="fix the KV cache size (default: fill free HBM)")
    ap.add_argument("--dump", default="", help="write per-request records of rank 0 to this JSON file")
    return ap.parse_args()


def pct(xs, q):
    return float(np.percentile(xs, q)) if len(xs) else None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU"
    # one rank per GPU; DLI_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the multi-rank path
    backend = os.environ.get("DLI_BENCH_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend == "gloo" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from dli_amd.config import CacheConfig, EngineConfig, SamplingParams, SchedulerConfig, get_model_config
    from dli_amd.engine.engine import LLMEngine
    import synth

    mcfg = get_model_config(args.model)
    cfg = EngineConfig(model=mcfg, cache=CacheConfig(kv_dtype=args.kv_dtype, num_blocks=args.num_kv_blocks),
                       sched=SchedulerConfig(max_num_seqs=args.max_num_seqs,
                                             max_num_batched_tokens=args.max_batched_tokens,
                                             max_model_len=2112, long_prefill_chunk=args.chunk,
                                             ttft_slo=None if args.fcfs else SLO_TTFT),
                       device="cuda", seed=args.seed, use_hip_graphs=not args.no_graphs,
                       async_scheduling=not args.sync_steps)
    t_load = time.perf_counter()
    eng = LLMEngine(cfg, device=dev)
    eng.warmup()
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t_load

    # ---- the global trace (identical on every rank), this rank's share
    n_req = int(args.qps * world * 600) + 1000
    trace = synth.make_burstgpt(None, n=n_req, mean_qps=args.qps * world, seed=args.seed + 17)
    mine = trace.iloc[rank::world]
    stamps = mine["Timestamp"].to_numpy()
    n_in = np.minimum(mine["Request tokens"].to_numpy(), 1024)
    n_out = np.minimum(mine["Response tokens"].to_numpy(), 1024)
    rng = np.random.default_rng(args.seed * 1000 + rank)

    nxt = 0
    t0 = None
    live = {}
    done = []

    def inject(now):
        nonlocal nxt
        while nxt < len(stamps) and stamps[nxt] <= now - t0:
            ids = rng.integers(3, mcfg.vocab_size, size=int(n_in[nxt]))
            p = SamplingParams(temperature=0.7, top_p=1.0, max_tokens=int(n_out[nxt]), ignore_eos=True, seed=nxt)
            live[f"q{nxt}"] = eng.add_request(f"q{nxt}", prompt_ids=ids, params=p, arrival=t0 + stamps[nxt])
            nxt += 1

    def one_step():
        """Exactly one engine step (sleeping through idle gaps does not count as a step)."""
        while True:
            now = time.perf_counter()
            inject(now)
            if eng.has_unfinished():
                break
            time.sleep(max(min(t0 + stamps[nxt] - now, 0.005), 0.0))
        for o in eng.step():
            if o.finished:
                done.append(live.pop(o.req_id))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    sync()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.ramp_seconds:          # untimed ramp to steady state
        one_step()
    for _ in range(args.warmup):
        one_step()
    sync()
    n_done_before = len(done)
    n_injected_before = nxt
    steps_before = (eng.metrics.steps, eng.metrics.step_tokens, eng.metrics.sched_time, eng.metrics.exec_time,
                    eng.metrics.post_time, dict(eng.runner.stats))
    t_start = time.perf_counter()
    beat = t_start
    for i in range(args.steps):
        one_step()
        if rank == 0 and time.perf_counter() - beat > 30.0:         # keep long runs visibly alive
            beat = time.perf_counter()
            print(f"[bench] step {i + 1}/{args.steps}, {beat - t_start:.0f} s", file=sys.stderr, flush=True)
    sync()
    t_end = time.perf_counter()
    elapsed = t_end - t_start

    fin = done[n_done_before:]
    ttft = np.array([r.first_token_time - r.arrival for r in fin])
    tpot = np.array([(r.finish_time - r.first_token_time) / max(r.num_output - 1, 1) for r in fin])
    ok = (ttft <= SLO_TTFT) & (tpot <= SLO_TPOT)
    ok_i = (ttft <= SLO_TTFT_I) & (tpot <= SLO_TPOT_I)
    m = eng.metrics
    local_stats = {
        "elapsed": elapsed, "finished": len(fin), "good": int(ok.sum()), "good_interactive": int(ok_i.sum()),
        "ttft": ttft.tolist(), "tpot": tpot.tolist(),
        "out_tokens": int(sum(r.num_output for r in fin)), "in_tokens": int(sum(r.num_prompt for r in fin)),
        "step_tokens": m.step_tokens - steps_before[1],
        "sched_s": m.sched_time - steps_before[2], "exec_s": m.exec_time - steps_before[3],
        "post_s": m.post_time - steps_before[4],
        "graph_steps": eng.runner.stats["graph_steps"] - steps_before[5]["graph_steps"],
        "eager_steps": eng.runner.stats["eager_steps"] - steps_before[5]["eager_steps"],
        "running": m.running, "waiting": m.waiting, "kv_used": m.kv_used_frac, "preemptions": m.preemptions,
        "backlog_s": float(time.perf_counter() - t0 - (stamps[nxt - 1] if nxt else 0.0)),
        "arrived": nxt - n_injected_before, "late_queue": len(eng.scheduler.late),
    }
    if world > 1:
        allstats = [None] * world
        dist.all_gather_object(allstats, local_stats)
    else:
        allstats = [local_stats]

    if rank == 0:
        T = max(s["elapsed"] for s in allstats)
        all_ttft = np.concatenate([np.array(s["ttft"]) for s in allstats]) if allstats else np.zeros(0)
        all_tpot = np.concatenate([np.array(s["tpot"]) for s in allstats])
        good = sum(s["good"] for s in allstats)
        finished = sum(s["finished"] for s in allstats)
        value = good / T
        line = {
            "metric": f"slo_goodput_req_per_s (TTFT<=2s & TPOT<=100ms, BurstGPT replay, {args.model})",
            "value": round(value, 3), "unit": "req/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(T / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(value / BASELINE_GOODPUT, 1), "dtype": "bf16",
            "data": "synthetic (BurstGPT-like Gamma arrivals + log-normal lengths clamped to 1024/1024, "
                    "random token ids, random-init weights)",
            "config": {"model": args.model, "global_batch": args.max_num_seqs * world, "seq_len": 2048,
                       "parallelism": f"dp{world}", "offered_qps": args.qps * world, "kv_dtype": args.kv_dtype,
                       "max_num_batched_tokens": args.max_batched_tokens, "chunk": args.chunk,
                       "pipelined_steps": not args.sync_steps, "slo_aware_admission": not args.fcfs},
            "requests_finished": finished, "slo_attainment": round(good / max(finished, 1), 4),
            "goodput_interactive_req_per_s": round(sum(s["good_interactive"] for s in allstats) / T, 3),
            "request_throughput": round(finished / T, 3),
            "arrival_rate_in_window": round(sum(s["arrived"] for s in allstats) / T, 3),
            "deferred_past_deadline": sum(s["late_queue"] for s in allstats),
            "ttft_p50_s": pct(all_ttft, 50), "ttft_p99_s": pct(all_ttft, 99),
            "tpot_p50_s": pct(all_tpot, 50), "tpot_p99_s": pct(all_tpot, 99),
            "output_tok_per_s": round(sum(s["out_tokens"] for s in allstats) / T, 1),
            "total_tok_per_s": round(sum(s["step_tokens"] for s in allstats) / T, 1),
            "rank0": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in allstats[0].items()
                      if k not in ("ttft", "tpot")},
            "load_s": round(t_load, 1), "kv_cache_tokens": eng.runner.kv_cache_tokens,
        }
        if args.dump:
            with open(args.dump, "w") as fh:
                json.dump([{"arrival": r.arrival - t0, "n_in": r.num_prompt, "n_out": r.num_output,
                            "ttft": r.first_token_time - r.arrival,
                            "tpot": (r.finish_time - r.first_token_time) / max(r.num_output - 1, 1)} for r in fin], fh)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
