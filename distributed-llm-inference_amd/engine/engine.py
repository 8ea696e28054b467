"""``LLMEngine`` — the synchronous core: request intake, one ``step()`` = schedule → execute →
sample → advance, output records for the streaming layer (SURVEY.md §3.6 hot loop).

The reference has no engine (its server was an external Ollama box); the behaviours the
harness relies on are produced here: one output record per generated token (→ one stream line), and
honouring ``max_tokens`` / ``temperature`` (which Ollama ignored).
"""
from __future__ import annotations

import itertools
import time
import zlib
from typing import Optional

import numpy as np

from ..config import EngineConfig, KV_BLOCK_SIZE, SamplingParams
from ..parallel.state import SINGLE, ParallelContext
from ..runtime.block_manager import BlockManager
from ..runtime.model_runner import ModelRunner
from ..utils.profiling import trace_range
from .metrics import EngineMetrics
from .request import Request, RequestOutput, RequestStatus
from .scheduler import Scheduler
from .tokenizer import get_tokenizer


class LLMEngine:
    def __init__(self, cfg: EngineConfig, ctx: Optional[ParallelContext] = None, device=None):
        self.cfg = cfg
        self.ctx = ctx or SINGLE
        # positions beyond the model's table (RoPE cache / learned positions) do not exist
        cfg.sched.max_model_len = min(cfg.sched.max_model_len, cfg.model.max_position)
        self.runner = ModelRunner(cfg, self.ctx, device)
        self.tokenizer = get_tokenizer(cfg.model, cfg.tokenizer_path)
        self.metrics = EngineMetrics()
        self.requests: dict[str, Request] = {}
        self._seq_ids = itertools.count(1)
        self._inflight = None
        self._warmed = False
        self._last_result_time = 0.0
        self.async_scheduling = self._default_async() if cfg.async_scheduling is None else bool(cfg.async_scheduling)
        if self.ctx.is_driver:
            self.bm = BlockManager(self.runner.num_blocks, KV_BLOCK_SIZE, cfg.cache.enable_prefix_caching)
            self.scheduler = Scheduler(cfg.sched, self.bm)
            self.scheduler.seq_ids = self._seq_ids
            self.scheduler.max_copies_per_step = ModelRunner.MAX_COW
        else:
            self.bm = None
            self.scheduler = None

    def _default_async(self) -> bool:
        """Pipelined stepping is the GPU default.  Under TP it needs stream-ordered collectives (RCCL): the step
        broadcast is then just another operation on the stream.  gloo moves device tensors through the host and
        blocks until the stream has drained — with step N's all-reduce kernels still spinning for their peer that
        wait never ends (seen with Llama-3-8B shapes in the single-GPU rehearsal mode), so gloo groups step
        one at a time."""
        if not self.runner.on_gpu:
            return False
        if self.ctx.distributed:
            import torch.distributed as dist
            return dist.get_backend(self.ctx.group) == "nccl"
        return True

    # ------------------------------------------------------------------ intake
    def add_request(self, req_id: str, prompt=None, params: Optional[SamplingParams] = None, prompt_ids=None,
                    arrival: Optional[float] = None, client_tag=None, n: int = 1) -> Request:
        """``n > 1`` (OpenAI ``n``): the prompt is prefilled ONCE; the other ``n - 1`` samples are forked off its KV pages
        when it completes and appear as requests ``f"{req_id}#{i}"`` (outputs carry ``parent_id`` / ``sample_index``)."""
        if req_id in self.requests:
            raise ValueError(f"duplicate request id {req_id!r}")
        if prompt_ids is None:
            prompt_ids = self.tokenizer.encode(prompt or "")
        max_len = self.cfg.sched.max_model_len
        if len(prompt_ids) >= max_len:                      # keep the tail, leave room for one token
            prompt_ids = prompt_ids[-(max_len - 1):]
        vocab = self.cfg.model.vocab_size
        prompt_ids = np.asarray(prompt_ids, dtype=np.int64) % vocab
        params = (params or SamplingParams()).clamp(max_len, len(prompt_ids))
        req = Request(req_id, next(self._seq_ids), prompt_ids, params, arrival, client_tag)
        req.n_samples = max(1, int(n))
        if params.seed is None:
            req.seed = (zlib.crc32(req_id.encode()) ^ (self.cfg.seed * 0x9E3779B1)) & 0x7FFFFFFF
        self.requests[req_id] = req
        self.scheduler.add(req)
        self.metrics.on_arrival(req)
        return req

    def abort_request(self, req_id: str) -> bool:
        for rid in [r for r, q in self.requests.items() if q.parent_id == req_id]:      # n > 1: the children go with it
            self.abort_request(rid)
        req = self.requests.pop(req_id, None)
        if req is None:
            return False
        req.forked = True                                        # an aborted parent must not fork any more
        self.scheduler.abort(req_id, req)
        self.metrics.on_finish(req, aborted=True)
        return True

    def has_unfinished(self) -> bool:
        return self.scheduler.has_work() or self._inflight is not None

    @property
    def num_running(self) -> int:
        return len(self.scheduler.running)

    @property
    def num_waiting(self) -> int:
        return self.scheduler.num_waiting

    def outstanding_tokens(self) -> int:
        """Load figure used by the DP router: tokens still to be computed or generated."""
        tot = 0
        for r in self.requests.values():
            tot += (r.num_tokens - r.num_computed) + max(r.params.max_tokens - r.num_output, 0)
        return tot

    # ------------------------------------------------------------------ one step
    def step(self) -> list[RequestOutput]:
        """One engine iteration.

        Pipelined (``cfg.async_scheduling``, the GPU default): schedule and queue step N+1 *behind* the
        step N still running on the GPU, then wait for step N's sampled ids and turn them into outputs —
        scheduling, batch staging and output processing all run while the GPU is busy, and the stream never
        drains between steps.  Ids the host has not seen yet are fed from the sampler's device buffer
        (``ops.gather_sampled``); an EOS discovered one step late costs one discarded token.
        Unpipelined: schedule → run → wait → outputs, nothing in flight between calls."""
        if not self._warmed and self.ctx.distributed:
            self.warmup()                                    # the TP workers are waiting in theirs (collective)
        t0 = time.perf_counter()
        with trace_range("schedule"):
            batch = self.scheduler.schedule()
        t1 = time.perf_counter()
        launched = None
        if batch is not None:
            for r in batch.requests:
                if r.first_scheduled is None:
                    r.first_scheduled = t1
            if self.ctx.distributed and self.metrics.steps % 128 == 0:
                from ..runtime.tp_launcher import check_children
                check_children()                             # a dead TP worker would hang the next collective
                # (a timed-out custom all-reduce is seen behind EVERY step without a sync and switches the group to
                #  the library collectives with the next step header: ModelRunner._note_ar_error / _broadcast_step)
            with trace_range(f"launch[T={batch.total_tokens},B={len(batch.requests)}]"):
                handle = self.runner.launch(batch, self.bm, self.scheduler.take_copies())
                self.metrics.launch_time += time.perf_counter() - t1     # host: stage + queue the step's kernels / graphs
                self.scheduler.advance(batch, self.cfg.sched.max_model_len)
                for child in self.scheduler.forked:          # n > 1: children forked off a prompt that just completed
                    self.requests[child.req_id] = child
                    self.metrics.requests_total += 1
                self.scheduler.forked.clear()
            launched = (batch, handle, t1 - t0, t1)
        if self.async_scheduling:
            prev, self._inflight = self._inflight, launched
        else:
            prev = launched
        outs = self._finish_step(prev) if prev is not None else []
        if self.scheduler.shed:
            outs += self._report_shed()
        return outs

    def _report_shed(self) -> list[RequestOutput]:
        """Requests the scheduler gave up on (``max_queue_delay``): one final record each, no token."""
        outs = []
        for req in self.scheduler.shed:
            self.requests.pop(req.req_id, None)
            self.metrics.on_finish(req, aborted=True)
            self.metrics.requests_shed += 1
            # an n > 1 parent that never got to fork answers for ALL its choices: the client's stream ends only after
            # n finished records (AsyncEngine.generate), and the children that would send the others do not exist
            choices = range(req.n_samples) if (req.n_samples > 1 and not req.forked) else (req.sample_index,)
            req.forked = True
            for c in choices:
                outs.append(RequestOutput(req.req_id if c == req.sample_index else f"{req.req_id}#{c}", [], True,
                                          RequestStatus.FINISHED_SHED.value, req.num_prompt, 0, 0, req.client_tag,
                                          req.parent_id if c == req.sample_index else req.req_id, c))
        self.scheduler.shed.clear()
        return outs

    def _finish_step(self, inflight) -> list[RequestOutput]:
        batch, handle, sched_s, t_launch = inflight
        t1 = time.perf_counter()
        with trace_range("collect"):
            sampled = self.runner.collect(handle)
        if self.runner.take_comm_fault():
            return self._recover_from_comm_fault(batch)
        t2 = time.perf_counter()
        with trace_range("update"):
            results = self.scheduler.resolve(batch, sampled, self.cfg.model.eos_token_id)
        outs = []
        for req, tok, status in results:
            if tok is None:
                continue
            if req.first_token_time is None:
                req.first_token_time = t2
            req.last_token_time = t2
            fin = status is not None
            if fin:
                req.finish_time = t2
                self.requests.pop(req.req_id, None)
                self.metrics.on_finish(req)
            outs.append(RequestOutput(req.req_id, [tok], fin, status.value if fin else None, req.num_prompt,
                                      req.num_output, req.num_cached_prompt, req.client_tag, req.parent_id,
                                      req.sample_index))
        # exec = launch → ids on the host (GPU time of the step as the caller sees it)
        self.metrics.on_step(batch, sched_s, t2 - t_launch, time.perf_counter() - t2, self)
        # step-time feedback for the TPOT controller: with a step queued behind this one the GPU never idles, so the
        # time between two consecutive results IS the step time; otherwise launch → result
        dt = t2 - max(t_launch, self._last_result_time) if self._last_result_time else t2 - t_launch
        self._last_result_time = t2
        self.scheduler.observe_step(dt, batch.total_tokens, batch.burst_grant)
        return outs

    def _recover_from_comm_fault(self, batch) -> list[RequestOutput]:
        """The step just collected ran a custom all-reduce that timed out (on this or — the kernels raise the flag in
        every peer's signal block — on any other rank): its sums, hence its sampled ids and the K/V it wrote, are
        garbage, and so is whatever the step queued behind it computed on top.  Neither step's tokens reach the
        clients: both are discarded and every sequence holding pages is recomputed (``Scheduler.reset_after_fault``).
        The group leaves the custom kernels with the next step header (``ModelRunner._broadcast_step``)."""
        batches = [batch]
        if self._inflight is not None:                       # queued behind the faulty step: drain it, drop it
            nxt, handle = self._inflight[0], self._inflight[1]
            self._inflight = None
            self.runner.collect(handle)
            self.runner.take_comm_fault()
            batches.append(nxt)
        n = self.scheduler.reset_after_fault(batches)
        self.metrics.comm_fault_recoveries += 1
        self._last_result_time = 0.0
        print(f"[dli_amd] all-reduce fault: discarded {len(batches)} step(s), recomputing {n} sequence(s)", flush=True)
        return []

    # ------------------------------------------------------------------ conveniences
    def generate(self, prompts, params: Optional[SamplingParams] = None, prompt_ids=None) -> list[list[int]]:
        """Offline helper: run a list of prompts to completion, return the generated ids per prompt."""
        n = len(prompts) if prompts is not None else len(prompt_ids)
        ids = []
        for i in range(n):
            rid = f"gen-{next(self._seq_ids)}-{i}"
            ids.append(rid)
            self.add_request(rid, None if prompts is None else prompts[i], params,
                             None if prompt_ids is None else prompt_ids[i])
        reqs = {rid: self.requests[rid] for rid in ids}
        while self.has_unfinished():
            self.step()
        return [reqs[rid].output_ids.tolist() for rid in ids]

    def warmup(self, capture_graphs: bool = True):
        """Compile/capture everything that would otherwise hit the first requests' latency, then check the fast
        paths against their plain counterparts (``ModelRunner.startup_selftest``).  Under TP every rank calls this
        (the captures and the checks contain collectives); the host barriers bound the rank skew seen by the first
        spin-waiting all-reduce kernel."""
        self._warmed = True
        self.ctx.barrier()
        if self.cfg.cache.kv_dtype == "fp8" and self.cfg.cache.calibrate_fp8_scales:
            ids = None
            if self.cfg.cache.calibration_file:
                with open(self.cfg.cache.calibration_file, encoding="utf-8", errors="replace") as fh:
                    ids = np.asarray(self.tokenizer.encode(fh.read()), dtype=np.int64) % self.cfg.model.vocab_size
            self.runner.calibrate_kv_scales(ids if ids is not None and len(ids) >= 16 else None)
        self.runner.tune_gemms()                             # measured GEMM dispatch: the captured steps hold its choices
        self.runner.selftest_overlap()                       # decides what the captured steps will contain
        self.runner.selftest_sp_norm()                       # (same: sequence-parallel reduce + add + norm, or the plain norms)
        if capture_graphs:
            self.runner.capture_all()
        self.runner.startup_selftest()
        self.ctx.barrier()

    def shutdown(self):
        if not self._warmed and self.ctx.distributed and self.ctx.is_driver:
            self.warmup()                                    # release the workers from their warm-up barrier first
        self.runner.shutdown_workers()
