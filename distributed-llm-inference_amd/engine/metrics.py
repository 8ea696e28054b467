"""Server-side observability (SURVEY.md §5.5 [NEW-BUILD]): counters, gauges and latency histograms
rendered as Prometheus text at ``/metrics``.  The reference only has the client-side ``log.json``."""
from __future__ import annotations

import bisect
import time

_TTFT_BUCKETS = (0.01, 0.025, 0.05, 0.1, 0.25, 0.5, 1.0, 2.0, 5.0, 10.0, 30.0)
_TPOT_BUCKETS = (0.002, 0.005, 0.01, 0.02, 0.03, 0.05, 0.1, 0.2, 0.5, 1.0)
_STEP_BUCKETS = (0.001, 0.002, 0.005, 0.01, 0.02, 0.05, 0.1, 0.25, 1.0)


class Histogram:
    def __init__(self, buckets):
        self.buckets = tuple(buckets)
        self.counts = [0] * (len(self.buckets) + 1)
        self.total = 0.0
        self.n = 0

    def observe(self, v: float):
        self.counts[bisect.bisect_left(self.buckets, v)] += 1
        self.total += v
        self.n += 1

    def render(self, name: str, help_: str) -> list[str]:
        out = [f"# HELP {name} {help_}", f"# TYPE {name} histogram"]
        acc = 0
        for b, c in zip(self.buckets, self.counts):
            acc += c
            out.append(f'{name}_bucket{{le="{b}"}} {acc}')
        out.append(f'{name}_bucket{{le="+Inf"}} {self.n}')
        out.append(f"{name}_sum {self.total}")
        out.append(f"{name}_count {self.n}")
        return out


class EngineMetrics:
    def __init__(self):
        self.started = time.time()
        self.requests_total = 0
        self.requests_finished = 0
        self.requests_aborted = 0
        self.requests_shed = 0
        self.comm_fault_recoveries = 0
        self.prompt_tokens = 0
        self.cached_prompt_tokens = 0
        self.generated_tokens = 0
        self.steps = 0
        self.step_tokens = 0
        self.last_batch_tokens = 0
        self.last_batch_seqs = 0
        self.running = 0
        self.waiting = 0
        self.kv_used_frac = 0.0
        self.preemptions = 0
        self.sched_time = 0.0
        self.launch_time = 0.0
        self.exec_time = 0.0
        self.post_time = 0.0
        self.ttft = Histogram(_TTFT_BUCKETS)
        self.tpot = Histogram(_TPOT_BUCKETS)
        self.queue_time = Histogram(_TTFT_BUCKETS)
        self.step_time = Histogram(_STEP_BUCKETS)

    def on_arrival(self, req):
        self.requests_total += 1
        self.prompt_tokens += req.num_prompt

    def on_finish(self, req, aborted: bool = False):
        if aborted:
            self.requests_aborted += 1
            return
        self.requests_finished += 1
        self.generated_tokens += req.num_output
        self.cached_prompt_tokens += req.num_cached_prompt
        if req.first_token_time is not None:
            self.ttft.observe(req.first_token_time - req.arrival)
            if req.first_scheduled is not None:
                self.queue_time.observe(req.first_scheduled - req.arrival)
            if req.num_output > 1 and req.finish_time is not None:
                self.tpot.observe((req.finish_time - req.first_token_time) / (req.num_output - 1))

    def on_step(self, batch, t_sched, t_exec, t_post, engine):
        self.steps += 1
        self.step_tokens += batch.total_tokens
        self.last_batch_tokens = batch.total_tokens
        self.last_batch_seqs = len(batch.requests)
        self.sched_time += t_sched
        self.exec_time += t_exec
        self.post_time += t_post
        self.step_time.observe(t_sched + t_exec + t_post)
        sch = engine.scheduler
        self.running, self.waiting = len(sch.running), sch.num_waiting
        self.preemptions = sch.num_preemptions
        bm = engine.bm
        self.kv_used_frac = 1.0 - bm.num_free / bm.num_blocks

    def snapshot(self) -> dict:
        return {k: v for k, v in self.__dict__.items() if isinstance(v, (int, float))}

    def render_prometheus(self, extra: dict | None = None) -> str:
        g = [
            ("dli_requests_total", "counter", "requests received", self.requests_total),
            ("dli_requests_finished_total", "counter", "requests completed", self.requests_finished),
            ("dli_requests_aborted_total", "counter", "requests aborted (client disconnect)", self.requests_aborted),
            ("dli_requests_shed_total", "counter", "requests dropped by overload shedding (max_queue_delay)", self.requests_shed),
            ("dli_comm_fault_recoveries_total", "counter", "steps discarded after an all-reduce time-out (sequences recomputed)", self.comm_fault_recoveries),
            ("dli_prompt_tokens_total", "counter", "prompt tokens received", self.prompt_tokens),
            ("dli_prefix_cache_hit_tokens_total", "counter", "prompt tokens served from the prefix cache", self.cached_prompt_tokens),
            ("dli_generated_tokens_total", "counter", "tokens generated", self.generated_tokens),
            ("dli_engine_steps_total", "counter", "engine steps", self.steps),
            ("dli_step_tokens_total", "counter", "tokens computed over all steps", self.step_tokens),
            ("dli_preemptions_total", "counter", "sequences preempted (recompute)", self.preemptions),
            ("dli_running_requests", "gauge", "sequences in the running set", self.running),
            ("dli_waiting_requests", "gauge", "sequences waiting for admission", self.waiting),
            ("dli_kv_cache_usage", "gauge", "fraction of KV pages in use", self.kv_used_frac),
            ("dli_last_batch_tokens", "gauge", "tokens in the most recent step", self.last_batch_tokens),
            ("dli_last_batch_seqs", "gauge", "sequences in the most recent step", self.last_batch_seqs),
            ("dli_schedule_seconds_total", "counter", "host time in the scheduler", self.sched_time),
            ("dli_launch_seconds_total", "counter", "host time staging a step and queueing its kernels / graph launches", self.launch_time),
            ("dli_execute_seconds_total", "counter", "time in model execution (incl. sync)", self.exec_time),
            ("dli_postprocess_seconds_total", "counter", "host time after execution", self.post_time),
        ]
        for k, v in (extra or {}).items():
            g.append((k, "gauge", k, v))
        out = []
        for name, typ, help_, val in g:
            out += [f"# HELP {name} {help_}", f"# TYPE {name} {typ}", f"{name} {val}"]
        out += self.ttft.render("dli_time_to_first_token_seconds", "arrival → first token")
        out += self.tpot.render("dli_time_per_output_token_seconds", "mean inter-token time per request")
        out += self.queue_time.render("dli_queue_time_seconds", "arrival → first scheduled")
        out += self.step_time.render("dli_step_seconds", "engine step wall time")
        return "\n".join(out) + "\n"
