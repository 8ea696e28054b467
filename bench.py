"""Headline benchmark: SLO goodput (req/s) + p50/p99 TTFT & TPOT under BurstGPT replay (BASELINE.json
``metric`` / ``configs``), bf16, random-init weights, synthetic prompts of the traced length distribution.

    python bench.py                                   # 1 GPU, Llama-3-8B (BASELINE.json config 2)
    python bench.py --gpus 8                          # 8 data-parallel replicas (config 3); starts the 8 ranks itself
    python bench.py --gpus 8 --tp 8 --model llama3-70b                        # config 4
    python bench.py --gpus 8 --tp 8 --ep 8 --model mixtral-8x7b --kv-dtype fp8   # config 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps K --warmup W        # the driver's launch: ranks already exist
    python bench.py --sweep                           # SLO frontier: largest offered QPS with ≥ 90 % attainment

One rank per GPU.  ``--tp`` consecutive ranks form one tensor-parallel replica (``dp = gpus / tp`` replicas);
every replica serves its share (round-robin, the router's stateless policy) of ONE global BurstGPT-like trace
whose offered load grows with the replica count (weak scaling: ``--qps`` requests/s per replica).  Arrivals are
replayed in real time, open loop, exactly like ``traffic_generator/main.py`` does over HTTP: a request becomes
visible to the engine at ``t0 + Timestamp`` whatever the engine is doing.  Lengths are clamped to 1024/1024 like
the harness (`/root/reference/traffic_generator/main.py:163-165`); each request generates exactly its traced
``Response tokens`` (ignore-EOS), so TPOT is well defined.

Without a torchrun environment and ``--gpus N > 1`` the parent process starts the N ranks itself *before it
touches a GPU* and exits non-zero if any rank fails (``dli_amd/runtime/launch.py``).  Before the engine is built
every rank probes RCCL and the custom xGMI collectives in a throw-away child process
(``dli_amd/parallel/comm_probe.py``): what fails there is not used here (the JSON line says what is in use), and
the probe's all-reduce latency table is embedded in the line.

A *step* is one engine iteration (schedule → forward over the continuous batch → sample → advance).  Untimed:
weight init, hipGraph capture, start-up self-tests, (``--tp`` > 1 without ``--qps``: a capacity probe that sets the
offered load,) a ramp that replays until the engine IS in steady state — the population of requests in the system
grows by less than 10 % of the arrivals over the last 10 s (``SteadyDetector``; at least ``--ramp-seconds``, at most
``--ramp-cap-seconds``) — then ``--sustained-seconds`` more of replay, then ``--warmup`` steps.  Timed: exactly
``--steps`` steps between barrier + synchronize brackets (``ms_per_step``); ``value_window`` = requests that
*finished inside the timed window* with TTFT and TPOT within the model's relaxed SLO, summed over
replicas, divided by the max-over-ranks wall time.  A window shorter than 10 s or 500 finished requests is a noisy
sample of a rate (20 steps ≈ 1 s ≈ 60 requests): the line then carries ``window_too_short: true`` and ``value`` is
``value_sustained`` — the same quantity over the last ``--sustained-seconds`` of real-time replay, ALL of it after the
steady point and ending with the timed steps, with a 95 % confidence interval; otherwise ``value`` is the timed
window's own figure.  ``steady`` says whether the criterion was met before the cap.  ``vs_baseline`` is null: the
reference publishes no throughput figure to divide by (its only replay served 6 requests serially on
unknown hardware).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "traffic_generator"))

# (TTFT s, TPOT s) per model: BASELINE.md §3
SLOS = {
    "llama3-8b": {"relaxed": (2.0, 0.100), "interactive": (0.5, 0.030)},
    "llama3-70b": {"relaxed": (2.0, 0.200), "interactive": (0.45, 0.040)},
    "mixtral-8x7b": {"relaxed": (2.0, 0.200), "interactive": (0.5, 0.040)},
}
# default offered load per replica (requests/s): just above the measured capacity of one replica, so that the SLO goodput in
# the line is the capacity and not the arrival rate of the window (8B: 65.6 req/s good at 72 offered, 65.9 at 76 offered,
# 100 % of finished requests within the SLO in both — profiles/bench_r3o_offered7{2,6}.json; at 70 offered the stretch
# saw 64.4 arrivals/s and finished 62.4)
DEFAULT_QPS = {"llama3-8b": 72.0, "llama3-70b": 8.0, "mixtral-8x7b": 40.0}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--tp", type=int, default=1, help="tensor-parallel degree of one replica (dp = gpus / tp)")
    ap.add_argument("--ep", type=int, default=1, help="expert-parallel degree (1 or tp; MoE models)")
    ap.add_argument("--emulate-tp", type=int, default=0,
                    help="run ONE rank (TP rank 0) of a replica of this TP degree on one GPU: that rank's exact shards, every "
                         "collective replaced by a same-bytes device copy, load found by the capacity probe.  A workload "
                         "study (GPU-busy, host time per step, kernel mix) — the line is marked INVALID as a throughput figure")
    ap.add_argument("--qps", type=float, default=float(os.environ.get("DLI_BENCH_QPS", "0")),
                    help="offered load per replica (requests/s); 0: the model's default")
    ap.add_argument("--ramp-seconds", type=float, default=30.0,
                    help="shortest untimed replay before the window; the ramp goes on until the steady-state criterion holds")
    ap.add_argument("--ramp-cap-seconds", type=float, default=120.0, help="longest ramp (the line then says steady: false)")
    ap.add_argument("--sustained-seconds", type=float, default=60.0,
                    help="replay between the steady point and the timed steps; value_sustained covers exactly this stretch "
                         "(the trace's arrival rate swings ±15 %% between 10 s bins: 60 s average that out to ≈ ±3 %%)")
    ap.add_argument("--capacity-probe-seconds", type=float, default=20.0,
                    help="--tp > 1 without --qps: saturate the replica this long, offer 0.9 × the measured completion rate")
    ap.add_argument("--max-num-seqs", type=int, default=0, help="0: 2048, or 4096 for --tp > 2")
    ap.add_argument("--max-batched-tokens", type=int, default=0, help="0: 4096, or 8192 for --tp > 2 (1/tp of the GEMM work per GPU)")
    ap.add_argument("--chunk", type=int, default=2048)
    ap.add_argument("--kv-dtype", default="auto")
    ap.add_argument("--slo", default="relaxed", choices=["relaxed", "interactive"], help="SLO row that defines `value`")
    ap.add_argument("--tpot-control", action="store_true",
                    help="TPOT-targeting token budget (SchedulerConfig.tpot_slo = the SLO row's TPOT)")
    ap.add_argument("--max-queue-delay", type=float, default=None,
                    help="shed requests not admitted after this many seconds (default: 4 × the TTFT SLO)")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-gemm-tune", action="store_true", help="skip the start-up GEMM dispatch measurement (A/B)")
    ap.add_argument("--no-mixed-graphs", action="store_true", help="capture pure-decode steps only (A/B of the mixed-step graphs)")
    ap.add_argument("--no-custom-ar", action="store_true")
    ap.add_argument("--quantization", default=None, choices=["fp8"],
                    help="W8A8 dense projections on the fp8 matrix cores (EngineConfig.quantization); NOT the headline config: the line is marked reduced precision")
    ap.add_argument("--tune-library-gemms", default=None, metavar="FILE",
                    help="PyTorch TunableOp results file: tune hipBLASLt / rocBLAS per GEMM shape at warm-up (EngineConfig.library_gemm_tuning)")
    ap.add_argument("--no-ttft-tpot-trade", action="store_true",
                    help="TPOT control as a fixed per-step cap (A/B of SchedulerConfig.ttft_tpot_trade)")
    ap.add_argument("--fcfs", action="store_true", help="plain FCFS admission (default: requests past their TTFT deadline yield)")
    ap.add_argument("--sync-steps", action="store_true", help="disable pipelined stepping (one step in flight at a time)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--num-kv-blocks", type=int, default=None, help="fix the KV cache size (default: fill free HBM)")
    ap.add_argument("--num-layers", type=int, default=None, help="override the layer count (rehearsals only: marks the line INVALID)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"])
    ap.add_argument("--no-comm-probe", action="store_true")
    ap.add_argument("--probe-seconds", type=float, default=5.0)
    ap.add_argument("--ep-a2a-min-tokens", type=int, default=2048)
    ap.add_argument("--sweep", action="store_true", help="SLO frontier instead of the fixed-load run")
    ap.add_argument("--sweep-qps", default="", help="comma-separated offered loads per replica (default: around the model's rate)")
    ap.add_argument("--sweep-window", type=float, default=30.0)
    ap.add_argument("--sweep-rows", default="relaxed,interactive", help="SLO rows the sweep covers")
    ap.add_argument("--sweep-ramp", type=float, default=20.0, help="untimed replay before each sweep point's window")
    ap.add_argument("--dump", default="", help="write per-request records of rank 0 to this JSON file")
    return ap.parse_args(argv)


def pct(xs, q):
    import numpy as np
    return float(np.percentile(xs, q)) if len(xs) else None


class Replay:
    """Open-loop real-time replay of one replica's share of the global trace against an in-process engine."""

    def __init__(self, eng, vocab, stamps, n_in, n_out, seed):
        import numpy as np
        self.eng, self.vocab = eng, vocab
        self.stamps, self.n_in, self.n_out = stamps, n_in, n_out
        self.rng = np.random.default_rng(seed)
        self.nxt = 0
        self.t0 = None
        self.live = {}
        self.done = []                    # finished Request objects, in finish order
        self.shed = 0
        self.tag = 0

    def start(self, t0):
        self.t0 = t0

    def inject(self, now):
        from dli_amd.config import SamplingParams
        st, eng = self.stamps, self.eng
        while self.nxt < len(st) and st[self.nxt] <= now - self.t0:
            i = self.nxt
            ids = self.rng.integers(3, self.vocab, size=int(self.n_in[i]))
            p = SamplingParams(temperature=0.7, top_p=1.0, max_tokens=int(self.n_out[i]), ignore_eos=True, seed=i)
            rid = f"q{self.tag}-{i}"
            self.live[rid] = eng.add_request(rid, prompt_ids=ids, params=p, arrival=self.t0 + st[i])
            self.nxt += 1

    def one_step(self):
        """Exactly one engine step (sleeping through idle gaps does not count as a step)."""
        eng = self.eng
        while True:
            now = time.perf_counter()
            self.inject(now)
            if eng.has_unfinished():
                break
            if self.nxt >= len(self.stamps):
                raise RuntimeError("trace exhausted: raise the request count")
            time.sleep(max(min(self.t0 + self.stamps[self.nxt] - now, 0.005), 0.0))
        for o in eng.step():
            if o.finished:
                r = self.live.pop(o.req_id, None)
                if r is None:
                    continue
                if o.finish_reason == "shed":
                    self.shed += 1
                else:
                    self.done.append(r)

    def run_for(self, seconds):
        self.run_until(time.perf_counter() + seconds)

    def run_until(self, end):
        while time.perf_counter() < end:
            self.one_step()

    @property
    def population(self):
        """Requests in the system (queued or running): injected − finished − shed."""
        return self.nxt - len(self.done) - self.shed

    def drain(self):
        """Abort everything in flight (between sweep points)."""
        eng = self.eng
        for rid in list(eng.requests):
            eng.abort_request(rid)
        while eng.has_unfinished():
            eng.step()
        self.live.clear()


class SteadyDetector:
    """Ramp criterion (shared with the sweep's ``mark_steady``): the replay is in steady state when the number of
    requests in the system grew by less than ``frac`` of the arrivals over the last ``horizon`` seconds — below that
    the latencies being measured are the ones this load ends up with; above it the backlog is still building."""

    def __init__(self, horizon=10.0, frac=0.10):
        self.horizon, self.frac = horizon, frac
        self.samples = []                                    # (t, population, arrivals so far)

    def add(self, t, population, arrived):
        self.samples.append((float(t), int(population), int(arrived)))

    def growth(self):
        """(population growth, arrivals) over the last ``horizon`` seconds, or None before that much was seen."""
        if not self.samples:
            return None
        t1, p1, a1 = self.samples[-1]
        old = [x for x in self.samples if x[0] <= t1 - self.horizon + 1e-9]
        if not old:
            return None
        _, p0, a0 = old[-1]
        return p1 - p0, a1 - a0

    def steady(self):
        g = self.growth()
        return bool(g is not None and g[1] > 0 and g[0] < self.frac * g[1])


def request_stats(reqs, slo):
    import numpy as np
    ttft = np.array([r.first_token_time - r.arrival for r in reqs], dtype=float)
    tpot = np.array([(r.finish_time - r.first_token_time) / max(r.num_output - 1, 1) for r in reqs], dtype=float)
    ok = (ttft <= slo[0]) & (tpot <= slo[1])
    return ttft, tpot, ok


def sustained(done, t_end, seconds, slo):
    """Goodput over the last ``seconds`` of replay: per-second bins → mean and 95 % CI half-width."""
    import numpy as np
    recent = [r for r in done if r.finish_time > t_end - seconds]
    if not recent:
        return {"good_per_bin": [0] * int(seconds), "n": 0, "ttft": [], "tpot": []}
    ttft, tpot, ok = request_stats(recent, slo)
    bins = np.zeros(int(seconds))
    for r, g in zip(recent, ok):
        if g:
            bins[min(int(t_end - r.finish_time), len(bins) - 1)] += 1
    return {"good_per_bin": bins.tolist(), "n": len(recent), "ttft": ttft.tolist(), "tpot": tpot.tolist()}


def main():
    args = parse()
    # ---- stage 0: no GPU touched yet.  Without a torchrun environment, start the ranks ourselves.
    from dli_amd.runtime import launch
    if args.gpus > 1 and not launch.launched_by_torchrun():
        sys.exit(launch.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist

    layout = launch.Layout.from_env(args.gpus, args.tp, args.ep if args.emulate_tp <= 1 else 1)   # emulated rank: ep lives in its context
    # stdout carries exactly ONE line: the JSON of rank 0.  Libraries write there too (gloo prints its connection
    # messages to stdout from C++), so the real stdout is kept aside and file descriptor 1 joins stderr for the run.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if os.environ.get("DLI_DUMP_STACKS_AFTER"):              # debug aid: where is a hung rank waiting?
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DLI_DUMP_STACKS_AFTER"]), repeat=True)

    def note(msg):
        if layout.rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)
    share = launch.share_gpu_requested()
    on_gpu = args.device == "cuda"
    job = launch.bootstrap(layout, device_kind=args.device, share_gpu=share, custom_all_reduce=not args.no_custom_ar,
                           probe_seconds=-1.0 if args.no_comm_probe else args.probe_seconds,
                           ep_a2a_min_tokens=args.ep_a2a_min_tokens)
    dev, ctx = job.device, job.ctx
    rank, world = layout.rank, layout.world
    if args.emulate_tp > 1:
        if world != 1 or args.tp != 1:
            raise SystemExit("--emulate-tp runs one rank on one GPU: use it with --gpus 1 and without --tp")
        from dli_amd.parallel.state import ParallelContext
        ctx = ParallelContext(tp_size=args.emulate_tp, tp_rank=0, ep_size=args.emulate_tp if args.ep > 1 else 1, emulated=True)
        job.comm["device_backend"] = f"emulated (rank 0 of tp{args.emulate_tp}: collectives are same-bytes device copies)"
    note(f"{world} rank(s) up: device backend {job.comm['device_backend']}, custom all-reduce "
         f"{job.comm['custom_ar_enabled']}, devices {job.comm['device_ids']}")

    from dli_amd.config import (CacheConfig, EngineConfig, ParallelConfig, SchedulerConfig, get_model_config)
    from dli_amd.engine.engine import LLMEngine
    from dli_amd.runtime.tp_launcher import worker_loop

    slo_rows = SLOS.get(args.model, SLOS["llama3-8b"])
    slo = slo_rows[args.slo]
    # offered load per replica: --qps, else the model's measured 1-GPU rate; a TP replica without --qps measures its own
    # capacity first (run_fixed → capacity_probe) — the figure here is then only the probe's starting point
    tp_eff = max(args.tp, args.emulate_tp)
    qps = args.qps or DEFAULT_QPS.get(args.model, 20.0) * (1.0 if tp_eff == 1 else 0.6 * tp_eff)
    if not args.max_num_seqs:
        args.max_num_seqs = 2048 if tp_eff <= 2 else 4096
    if not args.max_batched_tokens:
        args.max_batched_tokens = 4096 if tp_eff <= 2 else 8192
    over = {"num_layers": args.num_layers} if args.num_layers else {}
    mcfg = get_model_config(args.model, **over)
    qdelay = args.max_queue_delay if args.max_queue_delay is not None else 4.0 * slo[0]
    cfg = EngineConfig(model=mcfg, cache=CacheConfig(kv_dtype=args.kv_dtype, num_blocks=args.num_kv_blocks),
                       sched=SchedulerConfig(max_num_seqs=args.max_num_seqs,
                                             max_num_batched_tokens=args.max_batched_tokens,
                                             max_model_len=2112, long_prefill_chunk=args.chunk,
                                             ttft_slo=None if args.fcfs else slo[0],
                                             tpot_slo=slo[1] if args.tpot_control else None,
                                             ttft_tpot_trade=not args.no_ttft_tpot_trade,
                                             max_queue_delay=None if args.fcfs else qdelay),
                       parallel=ParallelConfig(tp=tp_eff, ep=args.ep if args.emulate_tp <= 1 else (tp_eff if args.ep > 1 else 1),
                                               custom_all_reduce=not args.no_custom_ar,
                                               ep_a2a_min_tokens=0 if args.emulate_tp > 1 else args.ep_a2a_min_tokens),
                       device=args.device, seed=args.seed, use_hip_graphs=on_gpu and not args.no_graphs,
                       async_scheduling=False if args.sync_steps else None, mixed_step_graphs=not args.no_mixed_graphs,
                       tune_gemms=not args.no_gemm_tune, quantization=args.quantization,
                       library_gemm_tuning=args.tune_library_gemms)
    t_load = time.perf_counter()
    eng = LLMEngine(cfg, ctx, dev)
    note(f"model built in {time.perf_counter() - t_load:.1f} s, {eng.runner.kv_cache_tokens} KV tokens; warming up")
    eng.warmup()
    if on_gpu:
        torch.cuda.synchronize()
    t_load = time.perf_counter() - t_load
    note(f"ready after {t_load:.1f} s ({len(eng.runner.graphs)} graphs, self-test {eng.runner.selftest}); replaying")

    def sync_drivers():
        if on_gpu:
            torch.cuda.synchronize()
        if job.driver_group is not None:
            dist.barrier(group=job.driver_group)

    def agree(v, how):
        """Drivers agree on a small integer: ``min`` of a value, or ``min_bits`` = AND of a bit mask (sum-free, 2 bits)."""
        if job.driver_group is None:
            return v
        if how == "min":
            t = torch.tensor([int(v)], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=job.driver_group)
            return int(t[0])
        t = torch.tensor([(v >> 1) & 1, v & 1], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=job.driver_group)
        return (int(t[0]) << 1) | int(t[1])

    if not layout.is_driver:
        worker_loop(eng)                                     # returns when the replica's driver shuts the group down
        stats = None
    elif args.sweep:
        stats = run_sweep(args, eng, mcfg, layout, slo_rows, qps, sync_drivers)
    else:
        stats = run_fixed(args, eng, mcfg, layout, slo, slo_rows, qps, sync_drivers, on_gpu, agree, note)
    if layout.is_driver:
        eng.shutdown()

    if world > 1:
        allstats = [None] * world
        dist.all_gather_object(allstats, stats, group=job.world_group)
        allstats = [s for s in allstats if s is not None]
    else:
        allstats = [stats]

    if rank == 0:
        if args.sweep:
            line = sweep_line(args, allstats, layout, qps)
        else:
            line = fixed_line(args, allstats, layout, slo, qps, mcfg)
        comm = dict(job.comm)
        table = (comm.get("probe") or {}).pop("all_reduce_table", [])
        line.update({
            "backend": {"control": comm["control_backend"], "device": comm["device_backend"]},
            "rccl_ranks_joined": comm["rccl_ranks_joined"], "device_ids": comm["device_ids"],
            "custom_ar_enabled": comm["custom_ar_enabled"], "ranks_share_gpu": comm["ranks_share_gpu"],
            "selftest": eng.runner.selftest, "comm_probe": comm.get("probe"),
            "tp_group_tuning": comm.get("tp_group_tuning"), "all_reduce_table_us": table,
            "load_s": round(t_load, 1), "kv_cache_tokens": eng.runner.kv_cache_tokens,
            "self_launched": os.environ.get("DLI_SELF_LAUNCHED") == "1",
        })
        if args.emulate_tp > 1:
            line["config"]["parallelism"] = f"tp{args.emulate_tp} rank 0 emulated on 1 GPU" + (f"+ep{args.emulate_tp}" if args.ep > 1 else "")
            line["INVALID_emulation"] = (f"one rank's share of a tp{args.emulate_tp} replica with device copies in place of the "
                                         "collectives: a workload study, not a throughput figure of any configuration")
        if args.quantization:
            line["dtype"] = f"bf16 activations / KV, {args.quantization} (e4m3, per-token × per-channel scales) dense projections above 64 rows"
            line["INVALID_reduced_precision"] = ("--quantization: an optional serving mode, not BASELINE.json's bf16 configuration; "
                                                 "quality on a trained checkpoint is unpinned here")
        if args.num_layers:
            line["INVALID_reduced_model"] = f"--num-layers {args.num_layers}: rehearsal, not the named architecture"
        print(json.dumps(line), file=result_out, flush=True)
    launch.shutdown(job)


# ------------------------------------------------------------------------------------------ fixed-load run
def make_share(args, mcfg, layout, qps, seconds, seed):
    """This replica's share (round-robin) of one global trace offered at ``qps × dp`` requests/s."""
    import numpy as np
    import synth
    dp = layout.dp
    n_req = int(qps * dp * seconds) + 1000
    trace = synth.make_burstgpt(None, n=n_req, mean_qps=qps * dp, seed=seed)
    mine = trace.iloc[layout.replica::dp]
    return (mine["Timestamp"].to_numpy(), np.minimum(mine["Request tokens"].to_numpy(), 1024),
            np.minimum(mine["Response tokens"].to_numpy(), 1024))


def capacity_probe(args, eng, mcfg, layout, qps_guess, sync_drivers, agree, note):
    """Offered load for a replica nobody has measured yet (``--tp`` > 1 without ``--qps``): SATURATE it for
    ``--capacity-probe-seconds`` (4 × the guess — the guess itself is 0.6 × tp × the 1-GPU rate, so this is ≈ 2.4 × what
    perfect scaling could serve; shedding on), take the token rate of the second half and divide by the trace's mean
    tokens per request → requests/s the replica can complete; offer 0.9 × the smallest replica's figure."""
    import numpy as np
    secs = args.capacity_probe_seconds
    probe_qps = 4.0 * qps_guess
    stamps, n_in, n_out = make_share(args, mcfg, layout, probe_qps, secs + 5.0, args.seed + 991)
    rp = Replay(eng, mcfg.vocab_size, stamps, n_in, n_out, args.seed * 1000 + 500 + layout.replica)
    rp.tag = "cap"
    sync_drivers()
    rp.start(time.perf_counter())
    rp.run_for(secs / 2)
    tok0, t_a = eng.metrics.step_tokens, time.perf_counter()
    rp.run_for(secs / 2)
    tok_s = (eng.metrics.step_tokens - tok0) / max(time.perf_counter() - t_a, 1e-6)
    pop_end = rp.population
    rp.drain()
    per_req = float(np.mean(n_in + n_out))
    est = tok_s / per_req
    est_all = agree(int(est * 1000), "min") / 1000.0
    saturated = pop_end > 0.25 * probe_qps * secs / 2          # a backlog built up: the rate is the replica's, not the trace's
    info = {"seconds": secs, "offered_qps_during_probe": round(probe_qps, 3), "saturated": bool(saturated),
            "in_system_at_end": int(pop_end), "tok_per_s": round(tok_s, 1),
            "mean_tokens_per_request": round(per_req, 1), "est_req_per_s": round(est, 3),
            "est_req_per_s_min_over_replicas": round(est_all, 3), "offered_fraction": 0.9}
    note(f"capacity probe: {tok_s:.0f} tok/s ≈ {est:.2f} req/s per replica → offering {0.9 * est_all:.2f}")
    return max(0.9 * est_all, 0.05), info


def run_fixed(args, eng, mcfg, layout, slo, slo_rows, qps, sync_drivers, on_gpu, agree, note):
    import numpy as np
    probe = None
    if (layout.tp > 1 or args.emulate_tp > 1) and not args.qps and args.capacity_probe_seconds > 0:
        qps, probe = capacity_probe(args, eng, mcfg, layout, qps, sync_drivers, agree, note)
    # an upper bound of the replay length: the ramp plus the steps at a pessimistic 0.25 s each
    horizon = args.ramp_cap_seconds + args.sustained_seconds + (args.steps + args.warmup) * 0.25 + 60.0
    stamps, n_in, n_out = make_share(args, mcfg, layout, qps, horizon, args.seed + 17)
    rp = Replay(eng, mcfg.vocab_size, stamps, n_in, n_out, args.seed * 1000 + layout.replica)
    ctx = eng.ctx

    sync_drivers()
    t0 = time.perf_counter()
    rp.start(t0)
    # ---- untimed ramp: until the population is steady on EVERY replica (all drivers check at the same replay times)
    det = SteadyDetector(horizon=min(10.0, max(args.ramp_seconds, 0.5)))
    tick = min(2.5, det.horizon / 2)
    k, steady = 0, False
    det.add(0.0, 0, 0)
    while True:
        k += 1
        rp.run_until(t0 + k * tick)
        t = time.perf_counter() - t0
        det.add(t, rp.population, rp.nxt)
        ok = t >= args.ramp_seconds - 1e-6 and det.steady()
        capped = t >= args.ramp_cap_seconds
        verdict = agree((2 if ok else 0) + (1 if capped else 0), "min_bits")
        if verdict & 2:
            steady = True
            break
        if verdict & 1:
            break
    t_steady = time.perf_counter()
    growth = det.growth()
    # ---- the sustained stretch: all of it after the steady point
    rp.run_until(t_steady + args.sustained_seconds)
    for _ in range(args.warmup):
        rp.one_step()
    sync_drivers()
    n_done_before, n_inj_before, shed_before = len(rp.done), rp.nxt, rp.shed
    late_before = len(eng.scheduler.late)
    m = eng.metrics
    before = (m.steps, m.step_tokens, m.sched_time, m.exec_time, m.post_time, dict(eng.runner.stats), dict(ctx.comm),
              m.launch_time)
    t_start = time.perf_counter()
    beat = t_start
    for i in range(args.steps):
        rp.one_step()
        if layout.rank == 0 and time.perf_counter() - beat > 30.0:      # keep long runs visibly alive
            beat = time.perf_counter()
            print(f"[bench] step {i + 1}/{args.steps}, {beat - t_start:.0f} s", file=sys.stderr, flush=True)
    sync_drivers()
    t_end = time.perf_counter()
    elapsed = t_end - t_start

    fin = rp.done[n_done_before:]
    ttft, tpot, ok = request_stats(fin, slo)
    _, _, ok_i = request_stats(fin, slo_rows["interactive"])
    _, _, ok_r = request_stats(fin, slo_rows["relaxed"])
    comm_delta = {k: ctx.comm[k] - before[6][k] for k in ctx.comm}
    # everything between the steady point and the end of the timed steps (whole seconds, counted back from the end)
    sus_s = int(t_end - t_steady)
    sus = sustained(rp.done, t_end, sus_s, slo) if sus_s >= 1 else None
    stats = {
        "replica": layout.replica, "elapsed": elapsed, "finished": len(fin), "good": int(ok.sum()),
        "good_interactive": int(ok_i.sum()), "good_relaxed": int(ok_r.sum()),
        "shed": rp.shed - shed_before, "late_growth": max(len(eng.scheduler.late) - late_before, 0),
        "ttft": ttft.tolist(), "tpot": tpot.tolist(),
        "out_tokens": int(sum(r.num_output for r in fin)), "in_tokens": int(sum(r.num_prompt for r in fin)),
        "step_tokens": m.step_tokens - before[1],
        "sched_s": m.sched_time - before[2], "exec_s": m.exec_time - before[3], "post_s": m.post_time - before[4],
        "launch_s": m.launch_time - before[7],
        "graph_steps": eng.runner.stats["graph_steps"] - before[5]["graph_steps"],
        "mixed_graph_steps": eng.runner.stats["mixed_graph_steps"] - before[5]["mixed_graph_steps"],
        "eager_steps": eng.runner.stats["eager_steps"] - before[5]["eager_steps"],
        "ar_fallbacks": eng.runner.stats["ar_fallbacks"],
        "running": m.running, "waiting": m.waiting, "kv_used": m.kv_used_frac, "preemptions": m.preemptions,
        "burst_steps": eng.scheduler.num_burst_steps, "burst_seconds": round(eng.scheduler.burst_seconds, 3),
        "backlog_s": float(time.perf_counter() - t0 - (stamps[rp.nxt - 1] if rp.nxt else 0.0)),
        "arrived": rp.nxt - n_inj_before, "late_queue": len(eng.scheduler.late), "comm": comm_delta,
        "sustained": sus, "replay_seconds": t_end - t0, "steady": steady, "ramp_seconds": t_steady - t0,
        "sustained_arrivals": int(((stamps > t_end - t0 - sus_s) & (stamps <= t_end - t0)).sum()),
        "ramp_population": rp.population, "ramp_growth": list(growth) if growth else None,
        "capacity_probe": probe, "offered_qps_per_replica": qps,
    }
    if args.dump and layout.rank == 0:
        with open(args.dump, "w") as fh:
            json.dump([{"arrival": r.arrival - t0, "n_in": r.num_prompt, "n_out": r.num_output,
                        "ttft": r.first_token_time - r.arrival,
                        "tpot": (r.finish_time - r.first_token_time) / max(r.num_output - 1, 1)} for r in fin], fh)
    return stats


def fixed_line(args, allstats, layout, slo, qps, mcfg):
    import numpy as np
    qps = allstats[0].get("offered_qps_per_replica", qps)   # the capacity probe may have replaced the default
    T = max(s["elapsed"] for s in allstats)
    all_ttft = np.concatenate([np.array(s["ttft"]) for s in allstats])
    all_tpot = np.concatenate([np.array(s["tpot"]) for s in allstats])
    good = sum(s["good"] for s in allstats)
    finished = sum(s["finished"] for s in allstats)
    shed = sum(s["shed"] for s in allstats)
    parked = sum(s["late_growth"] for s in allstats)
    value = good / T
    par = f"dp{layout.dp}" if layout.tp == 1 else (f"dp{layout.dp}xtp{layout.tp}" if layout.dp > 1 else f"tp{layout.tp}")
    if layout.ep > 1:
        par += f"+ep{layout.ep}"
    steps = max(args.steps, 1)
    comm = {k: sum(s["comm"][k] for s in allstats) for k in allstats[0]["comm"]}
    per_step = lambda k: round(comm[k] / (steps * len(allstats)), 2)
    line = {
        "metric": f"slo_goodput_req_per_s (TTFT<={slo[0]:g}s & TPOT<={slo[1] * 1e3:g}ms, BurstGPT replay, {args.model})",
        "value": round(value, 3), "unit": "req/s", "n_gpus": layout.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(T / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic (BurstGPT-like Gamma arrivals + log-normal lengths clamped to 1024/1024, "
                "random token ids, random-init weights)",
        "config": {"model": args.model, "global_batch": args.max_num_seqs * layout.dp, "seq_len": 2048,
                   "parallelism": par, "offered_qps": round(qps * layout.dp, 3), "kv_dtype": args.kv_dtype,
                   "max_num_batched_tokens": args.max_batched_tokens, "chunk": args.chunk,
                   "pipelined_steps": not args.sync_steps, "slo_aware_admission": not args.fcfs,
                   "tpot_control": args.tpot_control, "slo_row": args.slo, "layers": mcfg.num_layers},
        "window_s": round(T, 3), "requests_finished": finished,
        # attainment over everything the window had to deal with: finished + shed + newly parked past their deadline
        "slo_attainment": round(good / max(finished + shed + parked, 1), 4),
        "slo_attainment_of_finished": round(good / max(finished, 1), 4),
        "shed_in_window": shed, "parked_past_deadline_in_window": parked,
        "goodput_relaxed_req_per_s": round(sum(s["good_relaxed"] for s in allstats) / T, 3),
        "goodput_interactive_req_per_s": round(sum(s["good_interactive"] for s in allstats) / T, 3),
        "request_throughput": round(finished / T, 3),
        "arrival_rate_in_window": round(sum(s["arrived"] for s in allstats) / T, 3),
        "deferred_past_deadline": sum(s["late_queue"] for s in allstats),
        "ttft_p50_s": pct(all_ttft, 50), "ttft_p99_s": pct(all_ttft, 99),
        "tpot_p50_s": pct(all_tpot, 50), "tpot_p99_s": pct(all_tpot, 99),
        "output_tok_per_s": round(sum(s["out_tokens"] for s in allstats) / T, 1),
        "total_tok_per_s": round(sum(s["step_tokens"] for s in allstats) / T, 1),
        "host_ms_per_step": {k: round(sum(s[k + "_s"] for s in allstats) / (steps * len(allstats)) * 1e3, 3)
                             for k in ("sched", "launch", "post")},
        "all_reduce_calls_per_step": per_step("ar_custom_calls") + per_step("ar_lib_calls"),
        "all_reduce_bytes_per_step": per_step("ar_custom_bytes") + per_step("ar_lib_bytes"),
        "comm_per_step": {k: per_step(k) for k in comm},
        "rank0": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in allstats[0].items()
                  if k not in ("ttft", "tpot", "sustained", "comm")},
    }
    line["window_too_short"] = bool(T < 10.0 or finished < 500)
    line["value_window"] = line["value"]
    line["steady"] = bool(all(s["steady"] for s in allstats))
    line["ramp_s"] = round(max(s["ramp_seconds"] for s in allstats), 1)
    line["population_growth_vs_arrivals_at_ramp_end"] = allstats[0]["ramp_growth"]
    if allstats[0].get("capacity_probe"):
        line["capacity_probe"] = allstats[0]["capacity_probe"]
    if all(s["sustained"] is not None for s in allstats):
        nb = min(len(s["sustained"]["good_per_bin"]) for s in allstats)      # replicas may differ by a second
        bins = np.sum([np.array(s["sustained"]["good_per_bin"][:nb]) for s in allstats], axis=0)
        st = np.concatenate([np.array(s["sustained"]["ttft"]) for s in allstats])
        sp = np.concatenate([np.array(s["sustained"]["tpot"]) for s in allstats])
        line.update({
            "value_sustained": round(float(bins.mean()), 3), "sustained_window_s": len(bins),
            "sustained_n": int(sum(s["sustained"]["n"] for s in allstats)),
            "sustained_ci95": round(float(1.96 * bins.std(ddof=1) / np.sqrt(len(bins))), 3) if len(bins) > 1 else None,
            "sustained_ttft_p50_s": pct(st, 50), "sustained_ttft_p99_s": pct(st, 99),
            "sustained_tpot_p50_s": pct(sp, 50), "sustained_tpot_p99_s": pct(sp, 99),
            "sustained_arrivals_per_s": round(sum(s["sustained_arrivals"] for s in allstats) / max(len(bins), 1), 3),
        })
        if line["window_too_short"]:
            # K steps are a second or so of replay: the rate over the steady stretch that ENDS with them is the figure
            line["value"] = line["value_sustained"]
            line["value_source"] = (f"value_sustained: last {len(bins)} s of replay, all after the steady point "
                                    f"(the {args.steps}-step window alone is value_window)")
    line.setdefault("value_source", "timed window")
    return line


# ------------------------------------------------------------------------------------------ SLO frontier
def run_sweep(args, eng, mcfg, layout, slo_rows, qps, sync_drivers):
    """For each SLO row: replay at increasing offered load (fresh trace per point, ramp + window in real time) and
    record attainment over the requests that ARRIVED in the window (a tail lets them finish; the ones still generating
    are judged on their progress so far).  Loads are swept upwards without draining the engine in between.  A point
    only counts towards the frontier if it is a steady state (``mark_steady``)."""
    import numpy as np
    pts = [float(x) for x in args.sweep_qps.split(",") if x] or \
        [round(qps * f, 1) for f in (0.25, 0.4, 0.55, 0.7, 0.8, 0.9, 1.0, 1.1)]
    sc = eng.scheduler.cfg
    rows = {}
    for row_name, slo in slo_rows.items():
        if row_name not in args.sweep_rows.split(","):
            continue
        sc.ttft_slo = None if args.fcfs else slo[0]
        sc.max_queue_delay = None if args.fcfs else 4.0 * slo[0]
        sc.tpot_slo = slo[1] if (args.tpot_control or row_name == "interactive") else None
        eng.scheduler.dyn_budget = sc.max_num_batched_tokens
        eng.scheduler.step_ema = None
        eng.scheduler.credit = eng.scheduler.granted_inflight = 0.0
        out = []
        for pi, q in enumerate(pts):
            ramp, win, tail = args.sweep_ramp, args.sweep_window, 12.0
            stamps, n_in, n_out = make_share(args, mcfg, layout, q, ramp + win + tail + 30.0, args.seed + 101 + pi)
            rp = Replay(eng, mcfg.vocab_size, stamps, n_in, n_out, args.seed * 1000 + layout.replica + 7 * pi)
            rp.tag = f"{row_name[0]}{pi}"
            sync_drivers()
            t0 = time.perf_counter()
            rp.start(t0)
            rp.run_for(ramp + win + tail)                    # the tail lets the window's arrivals finish
            t_end = time.perf_counter()
            # the cohort: requests that arrived inside [ramp, ramp + win)
            lo, hi = t0 + ramp, t0 + ramp + win
            cohort_idx = np.flatnonzero((stamps >= ramp) & (stamps < ramp + win))
            fin = [r for r in rp.done if lo <= r.arrival < hi]
            ttft, tpot, ok = request_stats(fin, slo)
            # requests of the cohort still generating when the point ends (outputs of up to 1024 tokens outlast the tail):
            # judged on what they have shown so far — first token in time, running inter-token time within the SLO;
            # one that has no first token yet counts as failed once its TTFT budget is spent, and is left out otherwise
            good_open = n_open = 0
            for r in rp.live.values():
                if not (lo <= r.arrival < hi):
                    continue
                if r.first_token_time is None:
                    if t_end - r.arrival > slo[0]:
                        n_open += 1
                    continue
                n_open += 1
                run = (r.last_token_time - r.first_token_time) / max(r.num_output - r.pending - 1, 1)
                good_open += int(r.first_token_time - r.arrival <= slo[0] and run <= slo[1])
            shed_cohort = len(cohort_idx) - len(fin) - sum(1 for r in rp.live.values() if lo <= r.arrival < hi)
            # is the point a steady state?  requests in the system (queued or running) at the window's two ends, and
            # completions per second inside it: a point whose population still grows has not reached its TPOT yet
            served = list(rp.done) + list(rp.live.values())

            def in_system(t):
                return sum(1 for r in served if r.arrival <= t) - sum(1 for r in rp.done if r.finish_time <= t)
            pop_lo, pop_hi = in_system(lo), in_system(hi)
            done_in_win = sum(1 for r in rp.done if lo <= r.finish_time < hi)
            res = {"offered_qps": q * layout.dp, "cohort": len(fin) + n_open + max(shed_cohort, 0), "finished": len(fin),
                   "good": int(ok.sum()) + good_open, "still_running": n_open, "shed": max(shed_cohort, 0),
                   "ttft": ttft.tolist(), "tpot": tpot.tolist(), "window_s": win,
                   "in_system_at_window_start": pop_lo, "in_system_at_window_end": pop_hi, "finished_in_window": done_in_win}
            out.append(res)
            # no drain between the points of a row: loads are swept upwards and the next point starts from this one's
            # population (a point that starts from an empty engine needs 2–3 request lifetimes of ramp to be steady);
            # what is left of this point's requests finishes inside the engine, untracked
            rp.live.clear()
            if layout.rank == 0:
                att = res["good"] / max(res["cohort"], 1)
                print(f"[sweep] {row_name} qps/replica {q:g}: attainment {att:.3f} "
                      f"({res['good']}/{res['cohort']}), TTFT p50 {pct(ttft, 50)}, TPOT p50 {pct(tpot, 50)}",
                      file=sys.stderr, flush=True)
            if res["good"] / max(res["cohort"], 1) < 0.5:
                break                                        # far past the frontier: higher loads only take time
        rows[row_name] = out
        rp.drain()                                           # rows differ in scheduler settings: start the next one empty
    return {"replica": layout.replica, "rows": rows}


def mark_steady(t, window_s):
    """A sweep point is a steady state if the number of requests in the system (queued + running; shed ones have left)
    grows by less than 10 % of the window's arrivals: below that the point's TTFT / TPOT are the ones this load ends up
    with, above it the backlog is still building and the window has not seen the latencies to come."""
    t["population_growth_per_s"] = round((t["in_system_at_window_end"] - t["in_system_at_window_start"]) / window_s, 3)
    t["steady"] = bool(t["population_growth_per_s"] < 0.10 * t["arrivals_per_s"])
    return t


def sweep_line(args, allstats, layout, qps):
    import numpy as np
    table, frontier = {}, {}
    for row_name in allstats[0]["rows"]:
        n_pts = min(len(s["rows"][row_name]) for s in allstats)
        tab = []
        for i in range(n_pts):
            ps = [s["rows"][row_name][i] for s in allstats]
            cohort, good = sum(p["cohort"] for p in ps), sum(p["good"] for p in ps)
            ttft = np.concatenate([np.array(p["ttft"]) for p in ps])
            tpot = np.concatenate([np.array(p["tpot"]) for p in ps])
            tab.append({"offered_qps": ps[0]["offered_qps"], "arrivals": cohort, "good": good,
                        "still_running_at_end": sum(p.get("still_running", 0) for p in ps), "shed": sum(p.get("shed", 0) for p in ps),
                        "attainment": round(good / max(cohort, 1), 4),
                        "goodput_req_per_s": round(good / ps[0]["window_s"], 3),
                        "arrivals_per_s": round(cohort / ps[0]["window_s"], 3),
                        "finished_per_s": round(sum(p["finished_in_window"] for p in ps) / ps[0]["window_s"], 3),
                        "in_system_at_window_start": sum(p["in_system_at_window_start"] for p in ps),
                        "in_system_at_window_end": sum(p["in_system_at_window_end"] for p in ps),
                        "ttft_p50_s": pct(ttft, 50), "ttft_p99_s": pct(ttft, 99),
                        "tpot_p50_s": pct(tpot, 50), "tpot_p99_s": pct(tpot, 99)})
        for t in tab:
            mark_steady(t, ps[0]["window_s"])
        table[row_name] = tab
        okp = [t for t in tab if t["attainment"] >= 0.9 and t["steady"]]
        frontier[row_name] = max(okp, key=lambda t: t["offered_qps"]) if okp else None
    slo = SLOS.get(args.model, SLOS["llama3-8b"])
    best = frontier.get(args.slo)
    par = f"dp{layout.dp}" if layout.tp == 1 else f"dp{layout.dp}xtp{layout.tp}"
    return {
        "metric": f"max_offered_qps_with_90pct_slo_attainment ({args.slo}: TTFT<={slo[args.slo][0]:g}s & "
                  f"TPOT<={slo[args.slo][1] * 1e3:g}ms, BurstGPT replay, {args.model})",
        "value": best["offered_qps"] if best else 0.0, "unit": "req/s", "n_gpus": layout.world, "steps": 0, "warmup": 0,
        "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic (BurstGPT-like Gamma arrivals + log-normal lengths clamped to 1024/1024, random token ids, "
                "random-init weights); attainment over the requests that ARRIVED in each point's window",
        "config": {"model": args.model, "global_batch": args.max_num_seqs * layout.dp, "seq_len": 2048,
                   "parallelism": par, "kv_dtype": args.kv_dtype, "window_s": args.sweep_window, "ramp_s": args.sweep_ramp,
                   "tpot_control": "interactive row" if not args.tpot_control else "both rows"},
        "frontier": frontier, "sweep": table,
    }


if __name__ == "__main__":
    main()
