"""Continuous-batching scheduler with chunked prefill over the paged KV cache.

North-star item (1) of BASELINE.json; the reference's server side was a black box that served
requests strictly one at a time — this scheduler exists to remove that
head-of-line blocking under BurstGPT-style bursts.

Policy, one call per engine step:

1. **Running sequences first** (FCFS order): each decode gets its one token; an unfinished prefill
   gets up to ``long_prefill_chunk`` tokens.  All share the step's ``max_num_batched_tokens`` budget,
   so a burst of new prompts can never starve token generation of already admitted requests (bounds
   TPOT), while decodes leave the rest of the budget to prefill (bounds TTFT).
2. **Admission** of waiting requests while budget, sequence slots and KV pages (above a small
   watermark) remain.  On first admission the prompt is matched against the prefix cache, so
   multi-turn prompts skip their shared prefix.
3. **Preemption by recompute**: if a running sequence cannot get a page, the most recently admitted
   sequence is evicted (pages freed, progress reset, put at the head of the waiting queue).  With
   288 GB of HBM recompute is cheaper than swap.
4. **SLO control** (optional): ``ttft_slo`` — requests already past their TTFT deadline yield to fresher
   ones (under overload FCFS makes every request late); ``max_queue_delay`` — a request that still has
   not been admitted after that long is shed (finished with an error) instead of waiting for ever;
   ``tpot_slo`` — the engine feeds measured step times back (``observe_step``) and the prefill share
   of the token budget follows them, so that a burst of prompts cannot stretch the inter-token time of
   the running decodes past the target; when even the decode-only step is too slow, admission pauses.
   With both set, the two are traded (``ttft_tpot_trade``): TPOT is a per-request mean, so the time that steps stay
   under ``tpot_slo`` accumulates as credit, and a step may run longer — a grant paid from that credit — when
   waiting prompts would otherwise miss their TTFT deadline.

KV accounting lives in the C++ ``BlockManager`` (``csrc/runtime/block_manager.cpp``).
"""
from __future__ import annotations

import time
from collections import deque
from typing import Optional

import numpy as np

from ..config import SchedulerConfig
from .request import Request, RequestStatus, ScheduledBatch


_LPT_DECODE = __import__("os").environ.get("DLI_DECODE_ORDER", "longest_first") != "fifo"   # A/B knob

class Scheduler:
    def __init__(self, cfg: SchedulerConfig, block_manager, watermark: float = 0.01):
        self.cfg = cfg
        self.bm = block_manager
        self.waiting: deque[Request] = deque()
        self.late: deque[Request] = deque()                # ttft_slo only: deadline already missed, oldest first
        self.running: list[Request] = []
        self.watermark_blocks = max(int(block_manager.num_blocks * watermark), 1)
        self.num_preemptions = 0
        self.forked: list[Request] = []                    # children created by n > 1 forks; the engine registers them
        self.pending_copies: list[tuple[int, int]] = []    # (src page, dst page) copy-on-write work for the next launch
        self.seq_ids = None                                # iterator of fresh sequence ids (set by the engine)
        self.shed: list[Request] = []                      # finished with status "shed"; the engine reports them
        self.max_copies_per_step = 256                     # copy-on-write pairs one launch takes (ModelRunner.MAX_COW)
        self.num_shed = 0
        # TPOT controller state
        self.dyn_budget = cfg.max_num_batched_tokens        # current total-token target of a step
        self.step_ema: Optional[float] = None
        self.credit = 0.0                                  # seconds of step time saved under tpot_slo (TTFT ⇄ TPOT trade)
        self.granted_inflight = 0.0                        # grants of launched steps whose time has not been observed yet
        self.tok_rate: Optional[float] = None              # tokens per second of recent steps
        self.num_burst_steps = 0
        self.burst_seconds = 0.0

    # ------------------------------------------------------------------ queue management
    def add(self, req: Request) -> None:
        self.waiting.append(req)

    def abort(self, req_id: str, req: Optional[Request] = None) -> Optional[Request]:
        for q in (self.running, self.waiting, self.late):
            for r in q:
                if r.req_id == req_id:
                    q.remove(r)
                    self._release(r, RequestStatus.FINISHED_ABORTED)
                    return r
        if req is not None and req.finishing and not req.status.finished:
            self._release(req, RequestStatus.FINISHED_ABORTED)     # out of the queues, last token in flight
            return req
        return None

    def has_work(self) -> bool:
        return bool(self.waiting or self.running or self.late)

    @property
    def num_waiting(self) -> int:
        return len(self.waiting) + len(self.late)

    def _release(self, req: Request, status: RequestStatus) -> None:
        self.bm.free_seq(req.seq_id)
        req.status = status

    def finish(self, req: Request, status: RequestStatus) -> None:
        if not req.finishing:
            if req in self.running:
                self.running.remove(req)
            elif req in self.waiting:                      # preempted while its last step was in flight
                self.waiting.remove(req)
            elif req in self.late:
                self.late.remove(req)
        # publish the generated text too: the next turn of a conversation re-sends it as a prefix
        if self.bm.has_seq(req.seq_id):
            self.bm.commit(req.seq_id, req.all_ids, min(req.num_computed, req.num_tokens))
        self._release(req, status)

    def _preempt(self, req: Request) -> None:
        self.bm.free_seq(req.seq_id)
        req.num_computed = 0
        req.status = RequestStatus.PREEMPTED
        req.preemptions += 1
        self.num_preemptions += 1
        self.waiting.appendleft(req)

    def reset_after_fault(self, batches) -> int:
        """A step's results cannot be trusted (a tensor-parallel all-reduce kernel gave up waiting for a peer and
        finished with partial sums): nothing computed by the steps in ``batches`` — sampled ids, K/V written to the
        pages, pages published to the prefix cache — may be used.  Every sequence that holds pages goes back to the
        waiting queue for recomputation from the tokens the host already knew before those steps (placeholders of
        in-flight samples are dropped), and the prefix cache is emptied.  Returns the number of sequences reset."""
        victims = list(self.running)
        seen = {id(r) for r in victims}
        for b in batches:
            for r in b.requests:                            # taken out of `running` at launch (length limit reached)
                if id(r) not in seen and r.finishing and not r.status.finished:
                    victims.append(r)
                    seen.add(id(r))
        for r in victims:
            if self.bm.has_seq(r.seq_id):
                self.bm.free_seq(r.seq_id)                  # no commit: the pages may hold garbage
            r.num_tokens -= r.pending
            r.pending = 0
            r.finishing = False
            r.sample_row = -1
            r.num_computed = 0
            r.status = RequestStatus.PREEMPTED
            r.preemptions += 1
        self.running.clear()
        self.pending_copies.clear()
        for q in (self.waiting, self.late):                 # a prefix matched but not admitted holds references too
            for r in q:
                if self.bm.has_seq(r.seq_id):
                    self.bm.free_seq(r.seq_id)
                    r.num_computed = 0
        self.bm.reset_prefix_cache()
        victims.sort(key=lambda r: r.arrival)
        self.waiting.extendleft(reversed(victims))
        self.num_preemptions += len(victims)
        return len(victims)

    # ------------------------------------------------------------------ TPOT control
    def observe_step(self, seconds: float, tokens: int, grant: float = 0.0) -> None:
        """Feedback from the engine: wall time of a finished step, the tokens it computed and the burst grant it was
        scheduled with.  Multiplicative control of the step's token target towards ``tpot_slo · tpot_slo_margin``
        (+ the grant) seconds per step; the distance to ``tpot_slo`` itself goes to the credit account."""
        cfg = self.cfg
        if cfg.tpot_slo is None or seconds <= 0:
            return
        if tokens > 0:
            rate = tokens / seconds
            self.tok_rate = rate if self.tok_rate is None else 0.7 * self.tok_rate + 0.3 * rate
        cap = cfg.burst_credit_cap * cfg.tpot_slo
        self.credit = min(max(self.credit + 0.95 * cfg.tpot_slo - seconds, -cap), cap)
        self.granted_inflight = max(self.granted_inflight - grant, 0.0)
        if grant <= 0:                                     # a deliberate long step says nothing about the usual ones
            self.step_ema = seconds if self.step_ema is None else 0.6 * self.step_ema + 0.4 * seconds
        target = cfg.tpot_slo * cfg.tpot_slo_margin + grant
        ratio = target / seconds
        if ratio < 1.0:                                    # too slow: shrink relative to what the step really held
            basis = min(self.dyn_budget, max(tokens, 1))
            new = basis * max(ratio, 0.6)
        else:                                              # head-room: grow gently (a burst arrives faster than it leaves)
            new = self.dyn_budget * min(ratio, 1.1)
        self.dyn_budget = int(min(max(new, cfg.min_prefill_tokens), cfg.max_num_batched_tokens))

    def _burst_grant(self, now: float, prefill_left: int) -> float:
        """Seconds by which this step may exceed the TPOT target: as much as the urgent prompts (waiting, or admitted
        but still without their first token, for longer than ``burst_urgent_frac · ttft_slo``) need beyond the step's
        prefill share, bounded by the credit not yet promised to steps in flight."""
        cfg = self.cfg
        if not (cfg.ttft_tpot_trade and cfg.ttft_slo is not None and self.tok_rate):
            return 0.0
        avail = self.credit - self.granted_inflight
        if avail <= 0:
            return 0.0
        cutoff = now - cfg.burst_urgent_frac * cfg.ttft_slo
        need = 0
        for q in (self.waiting, self.running):
            seen = 0
            for r in (q if q is self.waiting else reversed(q)):      # running: the most recently admitted sit at the tail
                seen += 1
                if seen > 64:
                    break
                if r.num_output == 0 and r.preemptions == 0 and r.arrival < cutoff:
                    need += r.num_tokens - r.num_computed
                elif q is self.waiting:
                    break                                             # arrival order: the rest is younger
        extra = need - prefill_left
        if extra <= 0:
            return 0.0
        return min(extra / self.tok_rate, avail)

    def _shed_overdue(self, now: float) -> None:
        lim = self.cfg.max_queue_delay
        if lim is None:
            return
        for q in (self.late, self.waiting):
            # arrival order: overdue requests sit at the head; preempted / partly generated ones and the queued
            # children of an n > 1 request (their parent is being served) are never shed
            while q and q[0].arrival < now - lim and q[0].num_output == 0 and q[0].preemptions == 0 \
                    and q[0].parent_id is None:
                req = q.popleft()
                if self.bm.has_seq(req.seq_id):
                    self.bm.free_seq(req.seq_id)
                req.status = RequestStatus.FINISHED_SHED
                self.shed.append(req)
                self.num_shed += 1

    # ------------------------------------------------------------------ one step
    def schedule(self) -> Optional[ScheduledBatch]:
        cfg = self.cfg
        budget = cfg.max_num_batched_tokens
        picked: list[tuple[Request, int]] = []
        preempted: list[Request] = []
        grant = 0.0
        now = time.perf_counter()
        self._shed_overdue(now)
        # prefill share of this step: everything (no TPOT target), or what the controller leaves after the decodes
        prefill_left = budget
        admit = True
        if cfg.tpot_slo is not None:
            n_dec = sum(1 for r in self.running if r.num_tokens - r.num_computed == 1)
            prefill_left = max(self.dyn_budget - n_dec, 0)
            if self.step_ema is not None and self.step_ema > cfg.tpot_slo and self.dyn_budget <= cfg.min_prefill_tokens:
                admit = False                              # the decode population alone is over the target
            elif prefill_left < cfg.min_prefill_tokens:
                prefill_left = cfg.min_prefill_tokens       # keep prompts trickling in while decodes meet the target
            grant = self._burst_grant(now, prefill_left) if admit else 0.0
            if grant > 0:
                prefill_left += int(grant * self.tok_rate)

        # 1) running sequences
        i = 0
        while i < len(self.running) and budget > 0:
            req = self.running[i]
            remaining = req.num_tokens - req.num_computed
            n_new = min(remaining, cfg.long_prefill_chunk, budget)
            if remaining > 1:                               # a prefill chunk competes for the prefill share
                n_new = min(n_new, prefill_left)
                if n_new <= 0:
                    i += 1
                    continue
            ok = True
            while not self.bm.allocate(req.seq_id, req.num_computed + n_new):
                victim = self.running.pop()                # most recently admitted
                self._preempt(victim)
                preempted.append(victim)
                if victim is req:
                    ok = False
                    break
                # a victim scheduled earlier in this step cannot exist: victims come from the tail
            if not ok:
                break
            picked.append((req, n_new))
            budget -= n_new
            if remaining > 1:
                prefill_left -= n_new
            i += 1

        # 2) admission (never in a step that had to preempt)
        if cfg.ttft_slo is not None and self.waiting:
            deadline = now - cfg.ttft_slo * (1.0 - cfg.ttft_slo_margin)
            while self.waiting and self.waiting[0].arrival < deadline and self.waiting[0].num_output == 0 \
                    and self.waiting[0].preemptions == 0 and self.waiting[0].parent_id is None:
                self.late.append(self.waiting.popleft())    # arrival order = deadline order: late ones sit at the head
        if not preempted and admit:
            while (self.waiting or self.late) and budget > 0 and prefill_left > 0 and len(self.running) < cfg.max_num_seqs:
                queue = self.waiting if self.waiting else self.late
                req = queue[0]
                matched_now = False
                if not self.bm.has_seq(req.seq_id):
                    cached = self.bm.match_prefix(req.seq_id, req.all_ids)
                    matched_now = True
                    req.num_computed = cached
                    if req.preemptions == 0:
                        req.num_cached_prompt = cached
                remaining = req.num_tokens - req.num_computed
                n_new = min(remaining, cfg.long_prefill_chunk, budget, prefill_left)
                need_total = req.num_computed + n_new
                free_after = self.bm.num_free - max((need_total + self.bm.block_size - 1) // self.bm.block_size
                                                    - len(self.bm.block_table(req.seq_id)), 0)
                if ((not cfg.enable_chunked_prefill and n_new < remaining) or free_after < self.watermark_blocks
                        or not self.bm.allocate(req.seq_id, need_total)):
                    if matched_now:
                        # not admitted after all: drop the references on the matched prefix pages, or a request that
                        # is only waiting would pin them against eviction (it is matched again at its next attempt)
                        self.bm.free_seq(req.seq_id)
                        req.num_computed = 0
                    break
                queue.popleft()
                req.status = RequestStatus.RUNNING
                self.running.append(req)
                picked.append((req, n_new))
                budget -= n_new
                prefill_left -= n_new

        if not picked:
            return None
        # tile alignment: give the ragged remainder of a big step back to its largest prefill chunk
        align = cfg.token_alignment
        total = cfg.max_num_batched_tokens - budget
        if align > 1 and cfg.enable_chunked_prefill and total >= 4 * align and total % align:
            cut = total % align
            j = max(range(len(picked)), key=lambda i: picked[i][1])
            if picked[j][1] > cut + 1:                      # stays a multi-token chunk; its pages are already allocated
                picked[j] = (picked[j][0], picked[j][1] - cut)
        # decodes (single token) first, LONGEST CONTEXT FIRST among them: the decode-attention grid is dispatched in row
        # order, one workgroup per (row, kv head), and its work is the row's context — longest-processing-time-first
        # keeps the tail of the launch short (+3–15 % kernel bandwidth on the trace's length mix,
        # profiles/kernels_decode_r2ad.txt).  Prefill chunks keep their order.
        dec = [p for p in picked if p[1] == 1]
        if len(dec) > 1 and _LPT_DECODE:
            dec.sort(key=lambda p: -p[0].num_computed)
        picked = dec + [p for p in picked if p[1] != 1]
        reqs = [p[0] for p in picked]
        num_new = np.fromiter((p[1] for p in picked), dtype=np.int32, count=len(picked))
        num_decode = int((num_new == 1).sum())
        sample_mask = np.fromiter((r.num_computed + n == r.num_tokens for r, n in picked), dtype=bool, count=len(picked))
        total = int(num_new.sum())
        if grant > 0:                                       # charge what the step really took of the grant
            used = max(total - (self.dyn_budget if cfg.tpot_slo is not None else total), 0) / self.tok_rate
            grant = min(grant, used)
            if grant > 0:
                self.granted_inflight += grant
                self.num_burst_steps += 1
                self.burst_seconds += grant
        return ScheduledBatch(reqs, num_new, num_decode, total, sample_mask, preempted, grant)

    # ------------------------------------------------------------------ around the forward pass
    # A step is accounted for in two halves so that step N+1 can be scheduled and queued on the GPU
    # while step N is still running (pipelined stepping, SURVEY.md §7.3 item 5 "CPU overhead"):
    #   advance(batch)  – at launch: the KV of the scheduled tokens is as good as written (everything
    #                     later is ordered behind it on the stream); a sampled token is counted right away
    #                     as a placeholder, and a length limit reached by it takes the request out of
    #                     scheduling;
    #   resolve(batch…) – when the sampled ids arrive: fill the placeholders, detect EOS/stop tokens, finish.
    # An EOS found by resolve(N) may come after step N+1 was queued with the same request in it; that
    # step's extra token is dropped (the request is already finished when it resolves).
    def _fork_children(self, req: Request) -> None:
        """n > 1: the parent's prompt is in the cache now.  Each child shares the parent's pages (ref-counted), gets a
        private copy of the page holding the LAST prompt token (copy-on-write, done on the device before the next
        launch) and recomputes just that token — which yields its own first sample from its own RNG stream without any
        logits plumbing.  Without room for the private pages the children queue as ordinary requests instead (they
        still skip the shared prefix through the prefix cache)."""
        import dataclasses
        req.forked = True
        k = req.n_samples - 1
        room = self.bm.num_free - self.watermark_blocks >= k
        queued = []
        for i in range(1, req.n_samples):
            child = Request(f"{req.req_id}#{i}", next(self.seq_ids), req.tokens[:req.num_prompt],
                            dataclasses.replace(req.params), req.arrival, req.client_tag)
            child.seed = (req.seed + 0x9E3779B1 * i) & 0x7FFFFFFF
            child.parent_id, child.sample_index = req.req_id, i
            child.first_scheduled = req.first_scheduled
            # a child joins the running set only while the step buffers have a row for it (they are sized to
            # max_num_seqs: every running sequence is scheduled every step) and this step's copy-on-write list has a
            # slot; the others queue at the HEAD of the waiting line (they belong to an admitted request) and skip the
            # shared prompt through the prefix cache when their turn comes
            if room and len(self.running) < self.cfg.max_num_seqs and len(self.pending_copies) < self.max_copies_per_step:
                self.bm.fork(req.seq_id, child.seq_id)
                src, dst = self.bm.cow_tail(child.seq_id, req.num_prompt - 1)
                if src >= 0:
                    self.pending_copies.append((src, dst))
                child.num_computed = child.num_cached_prompt = req.num_prompt - 1
                child.status = RequestStatus.RUNNING
                self.running.append(child)
            else:
                queued.append(child)
            self.forked.append(child)
        self.waiting.extendleft(reversed(queued))

    def take_copies(self) -> list:
        out, self.pending_copies = self.pending_copies, []
        return out

    def advance(self, batch: ScheduledBatch, max_model_len: int) -> None:
        j = 0
        for req, n_new, samples in zip(batch.requests, batch.num_new, batch.sample_mask):
            req.num_computed += int(n_new)
            if samples:
                if req.num_computed - n_new < req.num_prompt:   # this step completes the prompt: publish its pages
                    self.bm.commit(req.seq_id, req.tokens[:req.num_prompt], req.num_prompt)
                    if req.n_samples > 1 and not req.forked:
                        self._fork_children(req)
                req.sample_row = j
                j += 1
                req.append(-1)
                req.pending += 1
                if req.num_output >= req.params.max_tokens or req.num_tokens >= max_model_len:
                    req.finishing = True
                    self.running.remove(req)               # holds its pages until the token arrives
            elif req.num_computed % (8 * self.bm.block_size) < n_new:
                # a long prefill crossed a multiple of 8 pages: publish its full pages early
                self.bm.commit(req.seq_id, req.tokens[:req.num_prompt], min(req.num_computed, req.num_prompt))

    def resolve(self, batch: ScheduledBatch, sampled: np.ndarray, eos_token_id: int):
        """Returns ``[(request, new_token or None, finished_status or None)]`` for the step's sequences."""
        out = []
        j = 0
        for req, samples in zip(batch.requests, batch.sample_mask):
            if not samples:
                out.append((req, None, None))
                continue
            tok = int(sampled[j])
            j += 1
            if req.status.finished:                         # stopped/aborted while this step was in flight
                continue
            pos = req.num_tokens - req.pending
            req.tokens[pos] = tok
            req.pending -= 1
            status = None
            p = req.params
            if (not p.ignore_eos) and (tok == eos_token_id or tok in p.stop_token_ids):
                status = RequestStatus.FINISHED_STOPPED
                req.num_tokens = pos + 1                    # drop the placeholder of a step queued behind
                req.pending = 0
            elif req.finishing and req.pending == 0:
                status = RequestStatus.FINISHED_LENGTH
            if status is not None:
                self.finish(req, status)
            out.append((req, tok, status))
        return out

    def update(self, batch: ScheduledBatch, sampled: np.ndarray, eos_token_id: int, max_model_len: int):
        """Both halves back to back (unpipelined stepping)."""
        self.advance(batch, max_model_len)
        return self.resolve(batch, sampled, eos_token_id)
