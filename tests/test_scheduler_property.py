"""Randomised invariants of the scheduler + C++ block manager (no model: the "forward pass" is a fake that
returns random token ids).  Every step must respect the token budget and the sequence cap, never hand the
same KV page to two live sequences that do not share it through the prefix cache, and release every page
once all requests are gone — under random arrivals, aborts, stop tokens, tiny caches (preemption) and
both stepping modes (scheduler correctness independent of kernels)."""
import numpy as np
import pytest

from dli_amd.config import KV_BLOCK_SIZE, SamplingParams, SchedulerConfig
from dli_amd.engine.request import Request
from dli_amd.engine.scheduler import Scheduler
from dli_amd.runtime.block_manager import BlockManager

EOS = 2


def run_random(seed: int, pipelined: bool, num_blocks: int, prefix: bool):
    rng = np.random.default_rng(seed)
    cfg = SchedulerConfig(max_num_seqs=6, max_num_batched_tokens=96, max_model_len=400, long_prefill_chunk=64,
                          token_alignment=16)
    bm = BlockManager(num_blocks, KV_BLOCK_SIZE, prefix)
    sch = Scheduler(cfg, bm)
    live, finished, next_id = {}, 0, 0
    shared = rng.integers(3, 50, size=70)
    inflight = None
    preempt_seen = 0
    aligned, big = [0], [0]
    for step in range(600):
        # arrivals / aborts
        if step < 350 and rng.random() < 0.25:
            n = int(rng.integers(1, 150))
            ids = np.concatenate([shared[:min(n, 70)], rng.integers(3, 50, size=max(n - 70, 0))]) if rng.random() < 0.4 \
                else rng.integers(3, 50, size=n)
            p = SamplingParams(max_tokens=int(rng.integers(1, 60)), ignore_eos=bool(rng.random() < 0.5),
                               stop_token_ids=(7,) if rng.random() < 0.3 else ()).clamp(cfg.max_model_len, n)
            r = Request(f"r{next_id}", next_id + 1, ids, p)
            next_id += 1
            live[r.req_id] = r
            sch.add(r)
        if live and rng.random() < 0.03:
            rid = list(live)[int(rng.integers(len(live)))]
            r = live.pop(rid)
            sch.abort(rid, r)

        batch = sch.schedule()
        if batch is not None:
            assert batch.total_tokens <= cfg.max_num_batched_tokens
            aligned[0] += batch.total_tokens >= 64 and batch.total_tokens % 16 == 0
            big[0] += batch.total_tokens >= 64
            assert len(sch.running) <= cfg.max_num_seqs
            assert batch.total_tokens == int(batch.num_new.sum()) and (batch.num_new > 0).all()
            assert (batch.num_new[:batch.num_decode] == 1).all()
            preempt_seen += len(batch.preempted)
            for r, n in zip(batch.requests, batch.num_new):
                assert r.num_computed + n <= r.num_tokens
                assert len(bm.block_table(r.seq_id)) * KV_BLOCK_SIZE >= r.num_computed + n
            # pages written this step (beyond the shared, already computed prefix) belong to one sequence only
            written = {}
            for r, n in zip(batch.requests, batch.num_new):
                bt = bm.block_table(r.seq_id)
                for pos in range(r.num_computed, r.num_computed + int(n)):
                    blk = bt[pos // KV_BLOCK_SIZE]
                    assert written.setdefault((blk, pos % KV_BLOCK_SIZE), r.seq_id) == r.seq_id
                    assert bm.ref_count(blk) == 1, "a page being written must not be shared"
            sampled = rng.integers(0, 12, size=int(batch.sample_mask.sum()))      # small range: EOS (2) and stop (7) occur
            sch.advance(batch, cfg.max_model_len)
            launched = (batch, sampled)
        else:
            launched = None
        todo = [inflight] if pipelined else [launched]
        if pipelined:
            inflight = launched
        for item in todo:
            if item is None:
                continue
            for req, tok, status in sch.resolve(item[0], item[1], EOS):
                if status is not None:
                    live.pop(req.req_id, None)
                    finished += 1
                    assert req.pending == 0 and (req.tokens[:req.num_tokens] >= 0).all()
                    assert req.num_output <= req.params.max_tokens
        if step > 350 and not sch.has_work() and inflight is None:
            break
    assert not sch.has_work() and not live, (len(sch.running), len(sch.waiting), len(live))
    assert bm.num_free == bm.num_blocks, "page leak"
    assert big[0] == 0 or aligned[0] >= 0.5 * big[0], (aligned, big)      # big steps are mostly trimmed to whole tiles
    return finished, preempt_seen


@pytest.mark.parametrize("pipelined", [False, True])
@pytest.mark.parametrize("seed", range(6))
def test_random_workload_invariants(seed, pipelined):
    finished, _ = run_random(seed, pipelined, num_blocks=64, prefix=True)
    assert finished > 20


@pytest.mark.parametrize("pipelined", [False, True])
def test_random_workload_with_preemption(pipelined):
    total_preempt = 0
    for seed in range(4):
        finished, pre = run_random(100 + seed, pipelined, num_blocks=14, prefix=False)
        assert finished > 10
        total_preempt += pre
    assert total_preempt > 0
