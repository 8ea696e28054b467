"""Engine / scheduler / block-manager tests on CPU with tiny random-init models.

The oracle for the paged, chunked, continuously-batched engine is a *dense* recomputation: the same
model run on the whole sequence at once, one sequence at a time, greedy.
"""
import numpy as np
import pytest
import torch

from dli_amd import ops
from dli_amd.config import (CacheConfig, EngineConfig, SamplingParams, SchedulerConfig, get_model_config)
from dli_amd.engine.engine import LLMEngine
from dli_amd.runtime.block_manager import BlockManager


def make_engine(model="tiny-llama", num_blocks=64, max_tokens=64, max_seqs=8, chunk=32, prefix=True, **kw):
    cfg = EngineConfig(model=get_model_config(model), cache=CacheConfig(num_blocks=num_blocks, enable_prefix_caching=prefix),
                       sched=SchedulerConfig(max_num_seqs=max_seqs, max_num_batched_tokens=max_tokens,
                                             max_model_len=256, long_prefill_chunk=chunk),
                       device="cpu", use_hip_graphs=False, **kw)
    return LLMEngine(cfg)


def dense_greedy(engine, prompt_ids, n_new):
    """Oracle: every step re-runs the full sequence as one fresh prefill in a scratch cache."""
    runner = engine.runner
    model = runner.model
    toks = list(prompt_ids)
    out = []
    for _ in range(n_new):
        L = len(toks)
        nb = (L + 31) // 32
        caches = ops.allocate_kv_cache(engine.cfg.model.num_layers, nb, runner.local_kv_heads,
                                       engine.cfg.model.head_dim, runner.kv_dtype, "cpu")
        meta = ops.AttnMetadata(num_decode=0, num_prefill=1, num_tokens=L,
                                positions=torch.arange(L, dtype=torch.int32),
                                slot_mapping=torch.arange(L, dtype=torch.int32),
                                block_tables=torch.arange(nb, dtype=torch.int32)[None],
                                seq_lens=torch.tensor([L], dtype=torch.int32),
                                query_start_loc=torch.tensor([0, L], dtype=torch.int32))
        with torch.inference_mode():
            h = model(torch.tensor(toks), meta, caches)
            nxt = int(model.compute_logits(h[-1:]).float().argmax(-1))
        toks.append(nxt)
        out.append(nxt)
    return out


GREEDY = SamplingParams(temperature=0.0, max_tokens=12, ignore_eos=True)


@pytest.mark.parametrize("model", ["tiny-llama", "tiny-opt", "tiny-mixtral"])
def test_engine_matches_dense_oracle(model):
    eng = make_engine(model)
    rng = np.random.default_rng(0)
    prompts = [rng.integers(3, 500, size=n).tolist() for n in (5, 70, 33, 1, 100)]
    got = eng.generate(None, GREEDY, prompt_ids=prompts)
    for p, g in zip(prompts, got):
        assert g == dense_greedy(eng, p, 12)
    assert eng.bm.num_free == eng.bm.num_blocks          # everything released
    assert eng.metrics.requests_finished == 5 and eng.metrics.generated_tokens == 60


def test_chunked_prefill_and_token_budget_respected():
    eng = make_engine(max_tokens=48, chunk=16)
    rng = np.random.default_rng(1)
    prompts = [rng.integers(3, 500, size=n).tolist() for n in (90, 60, 40)]
    for i, p in enumerate(prompts):
        eng.add_request(f"r{i}", prompt_ids=p, params=GREEDY)
    seen_mixed = False
    steps = 0
    while eng.has_unfinished():
        batch_before = eng.metrics.steps
        eng.step()
        steps += 1
        assert eng.metrics.last_batch_tokens <= 48
        assert eng.metrics.steps == batch_before + 1
        seen_mixed |= eng.metrics.last_batch_seqs > 1
    assert seen_mixed and steps > 12
    # same answers as unchunked, one-at-a-time
    ref = make_engine(max_tokens=256, chunk=256)
    assert ref.generate(None, GREEDY, prompt_ids=prompts) == \
        [dense_greedy(ref, p, 12) for p in prompts]


def test_preemption_recompute_gives_same_tokens():
    # 7 pages of 32 tokens: three 60-token prompts (2 pages each) + growth forces preemption
    eng = make_engine(num_blocks=7, max_tokens=256, chunk=256, prefix=False)
    rng = np.random.default_rng(2)
    prompts = [rng.integers(3, 500, size=60).tolist() for _ in range(3)]
    params = SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True)
    got = eng.generate(None, params, prompt_ids=prompts)
    assert eng.scheduler.num_preemptions > 0
    big = make_engine(num_blocks=64, max_tokens=256, chunk=256, prefix=False)
    assert got == big.generate(None, params, prompt_ids=prompts)
    assert eng.bm.num_free == 7


def test_prefix_cache_reuses_pages_and_keeps_results():
    eng = make_engine()
    rng = np.random.default_rng(3)
    shared = rng.integers(3, 500, size=96).tolist()
    a = shared + rng.integers(3, 500, size=10).tolist()
    b = shared + rng.integers(3, 500, size=17).tolist()
    out_a = eng.generate(None, GREEDY, prompt_ids=[a])[0]
    hits_before = eng.bm.cache_hits
    eng.add_request("b", prompt_ids=b, params=GREEDY)
    req = eng.requests["b"]
    while eng.has_unfinished():
        eng.step()
    assert eng.bm.cache_hits - hits_before == 3 and req.num_cached_prompt == 96
    assert req.output_ids.tolist() == dense_greedy(eng, b, 12)
    # multi-turn: prompt + generated answer is itself cached after the first turn
    turn2 = a + out_a + rng.integers(3, 500, size=5).tolist()
    eng.add_request("t2", prompt_ids=turn2, params=GREEDY)
    r2 = eng.requests["t2"]
    while eng.has_unfinished():
        eng.step()
    assert r2.num_cached_prompt >= 96 and r2.output_ids.tolist() == dense_greedy(eng, turn2, 12)
    assert eng.metrics.cached_prompt_tokens >= 192


def test_sampling_params_seeded_and_eos_and_abort():
    eng = make_engine()
    p = SamplingParams(temperature=0.8, top_p=0.9, top_k=50, max_tokens=10, seed=7, ignore_eos=True)
    a = eng.generate(None, p, prompt_ids=[[5, 6, 7, 8]])[0]
    b = eng.generate(None, p, prompt_ids=[[5, 6, 7, 8]])[0]
    assert a == b and len(a) == 10                                  # per-request seeded stream
    c = eng.generate(None, SamplingParams(temperature=0.8, max_tokens=10, seed=8, ignore_eos=True),
                     prompt_ids=[[5, 6, 7, 8]])[0]
    assert c != a
    # EOS stops generation: make the greedy token the EOS by construction
    first = eng.generate(None, SamplingParams(temperature=0.0, max_tokens=3, ignore_eos=True), prompt_ids=[[9, 10]])[0][0]
    eng.cfg.model.eos_token_id = first
    out = eng.generate(None, SamplingParams(temperature=0.0, max_tokens=30), prompt_ids=[[9, 10]])[0]
    assert out == [first]
    # abort frees pages
    eng.add_request("x", prompt_ids=list(range(3, 80)), params=SamplingParams(max_tokens=50, ignore_eos=True))
    eng.step()
    assert eng.bm.num_free < eng.bm.num_blocks
    assert eng.abort_request("x") and not eng.has_unfinished()
    assert eng.bm.num_free == eng.bm.num_blocks


def test_max_model_len_truncates_prompt_and_generation():
    eng = make_engine()
    out = eng.generate(None, SamplingParams(temperature=0.0, max_tokens=500, ignore_eos=True),
                       prompt_ids=[list(np.arange(3, 303) % 500)])[0]
    assert 1 <= len(out) <= 2                                       # 255-token tail + room for 1 token


# ------------------------------------------------------------------ block manager unit tests
def test_block_manager_alloc_free_refcount_cow():
    bm = BlockManager(8, 32, True)
    toks = np.arange(100, dtype=np.int32)
    assert bm.match_prefix(1, toks) == 0 and bm.allocate(1, 100)
    assert bm.block_table(1) == [0, 1, 2, 3] and bm.num_free == 4
    assert not bm.allocate(2, 32 * 5) and bm.num_free == 4          # all-or-nothing
    bm.free_seq(2)
    bm.fork(1, 3)
    assert bm.ref_count(3) == 2
    src, dst = bm.cow_tail(3, 100)                                  # tail page shared + partially filled
    assert (src, dst) == (3, 4) and bm.ref_count(3) == 1 and bm.block_table(3) == [0, 1, 2, 4]
    assert bm.cow_tail(3, 100) == (-1, -1) and bm.cow_tail(1, 96) == (-1, -1)
    bm.free_seq(1)
    bm.free_seq(3)
    assert bm.num_free == 8


def test_block_manager_prefix_cache_lru_eviction():
    bm = BlockManager(4, 32, True)
    a = np.arange(64, dtype=np.int32)
    bm.match_prefix(1, a)
    bm.allocate(1, 64)
    bm.commit(1, a, 64)
    bm.free_seq(1)
    assert bm.num_free == 4 and bm.num_cached == 2                  # cached pages stay evictable
    assert bm.match_prefix(2, np.concatenate([a, a[:5]])) == 64     # full hit on both pages
    bm.free_seq(2)
    assert bm.match_prefix(3, a) == 32                              # never the whole prompt
    bm.free_seq(3)
    b = np.arange(1000, 1128, dtype=np.int32)
    bm.match_prefix(4, b)
    assert bm.allocate(4, 128)                                      # needs all 4 pages → evicts the cache
    assert bm.num_cached == 0 and bm.num_free == 0
    bm.free_seq(4)
    assert bm.match_prefix(5, a) == 0


def test_build_batch_layout():
    bm = BlockManager(16, 32, False)
    bm.allocate(7, 40)
    bm.allocate(9, 70)
    pos, slot = np.zeros(64, np.int32), np.zeros(64, np.int32)
    bt, sl, qs = np.zeros((4, 8), np.int32), np.zeros(4, np.int32), np.zeros(5, np.int32)
    T = bm.build_batch(np.array([7, 9]), np.array([39, 60], np.int32), np.array([1, 10], np.int32), pos, slot, bt, sl, qs)
    assert T == 11 and sl[:2].tolist() == [40, 70] and qs[:3].tolist() == [0, 1, 11]
    assert pos[:11].tolist() == [39] + list(range(60, 70))
    t7, t9 = bm.block_table(7), bm.block_table(9)
    assert slot[0] == t7[1] * 32 + 7 and slot[1] == t9[1] * 32 + 28 and slot[5] == t9[2] * 32
    assert bt[0, :2].tolist() == t7 and bt[1, :3].tolist() == t9
    with pytest.raises(RuntimeError):
        bm.build_batch(np.array([7]), np.array([39], np.int32), np.array([30], np.int32), pos, slot, bt, sl, qs)


# ------------------------------------------------------------------ pipelined (async) stepping
def _run_workload(async_mode, num_blocks=64, stop=(), max_seqs=8, budget=64, abort_at=None):
    eng = make_engine(num_blocks=num_blocks, max_tokens=budget, max_seqs=max_seqs, chunk=32, prefix=True,
                      async_scheduling=async_mode)
    assert eng.async_scheduling == async_mode
    rng = np.random.default_rng(11)
    reqs = {}
    arrivals = {0: [0, 1, 2], 2: [3], 3: [4, 5], 9: [6], 15: [7, 8, 9]}
    lens = [5, 70, 33, 1, 100, 64, 31, 12, 90, 40]
    outs = {i: [] for i in range(10)}
    fins = {}
    step = 0
    while step < 400:
        for i in arrivals.get(step, ()):
            greedy = i % 2 == 0
            p = SamplingParams(temperature=0.0 if greedy else 0.9, top_k=0 if greedy else 40, max_tokens=6 + 3 * i,
                               seed=100 + i, stop_token_ids=tuple(stop), ignore_eos=False)
            reqs[i] = eng.add_request(f"r{i}", prompt_ids=rng.integers(3, 500, size=lens[i]).tolist(), params=p)
        if abort_at is not None and step == abort_at:
            assert eng.abort_request("r1")
        if step > 15 and not eng.has_unfinished():
            break
        for o in eng.step():
            i = int(o.req_id[1:])
            outs[i].extend(o.new_token_ids)
            if o.finished:
                assert i not in fins
                fins[i] = (o.finish_reason, o.num_output_tokens)
        step += 1
    assert not eng.has_unfinished() and eng.bm.num_free == eng.bm.num_blocks
    for i, r in reqs.items():
        if i in fins:
            assert r.pending == 0 and r.output_ids.tolist() == outs[i] and fins[i][1] == len(outs[i])
            assert (r.tokens[:r.num_tokens] >= 0).all()
    return outs, fins, eng


def test_pipelined_stepping_equals_unpipelined():
    a_out, a_fin, a_eng = _run_workload(False)
    b_out, b_fin, b_eng = _run_workload(True)
    assert a_out == b_out and a_fin == b_fin and len(a_fin) == 10
    assert sum(reason == "length" for reason, _ in a_fin.values()) >= 5
    # same forward passes, plus at most one extra per EOS that was discovered a step late
    assert a_eng.metrics.steps <= b_eng.metrics.steps <= a_eng.metrics.steps + 5


def test_pipelined_stepping_stop_tokens_are_found_one_step_late_but_output_is_identical():
    base, _, _ = _run_workload(False)
    # pick tokens that really occur mid-generation so that several requests stop early
    stop = sorted({base[i][len(base[i]) // 2] for i in (1, 4, 7, 8)})
    a_out, a_fin, _ = _run_workload(False, stop=stop)
    b_out, b_fin, b_eng = _run_workload(True, stop=stop)
    assert a_out == b_out and a_fin == b_fin
    assert sum(reason == "stop" for reason, _ in a_fin.values()) >= 3
    for i, (reason, _n) in a_fin.items():
        if reason == "stop":
            assert a_out[i][-1] in stop and not (set(a_out[i][:-1]) & set(stop))


def test_pipelined_stepping_with_preemption_and_abort():
    a_out, a_fin, a_eng = _run_workload(False, num_blocks=9, abort_at=6)
    b_out, b_fin, b_eng = _run_workload(True, num_blocks=9, abort_at=6)
    assert a_eng.scheduler.num_preemptions > 0 and b_eng.scheduler.num_preemptions > 0
    assert 1 not in a_fin and 1 not in b_fin
    for i in a_fin:
        assert a_out[i] == b_out[i] and a_fin[i] == b_fin[i]
    assert set(a_fin) == set(b_fin) == set(range(10)) - {1}


def test_slo_aware_admission_serves_requests_that_can_still_make_their_deadline_first():
    import time
    eng = make_engine(max_seqs=1)
    eng.cfg.sched.ttft_slo = 60.0                          # far from anything a loaded box could consume
    now = time.perf_counter()
    p = SamplingParams(temperature=0.0, max_tokens=3, ignore_eos=True)
    late = eng.add_request("late", prompt_ids=[5, 6, 7], params=p, arrival=now - 500.0)     # deadline long gone
    fresh = eng.add_request("fresh", prompt_ids=[8, 9, 10], params=p, arrival=now)
    order = []
    while eng.has_unfinished():
        for o in eng.step():
            if o.finished:
                order.append(o.req_id)
    assert order == ["fresh", "late"]                     # the late one is not dropped, it yields
    assert late.first_scheduled > fresh.first_scheduled and eng.num_waiting == 0
    # FCFS without the option
    eng2 = make_engine(max_seqs=1)
    eng2.add_request("late", prompt_ids=[5, 6, 7], params=p, arrival=now - 500.0)
    eng2.add_request("fresh", prompt_ids=[8, 9, 10], params=p, arrival=now)
    order2 = []
    while eng2.has_unfinished():
        order2 += [o.req_id for o in eng2.step() if o.finished]
    assert order2 == ["late", "fresh"]


def test_ttft_tpot_trade_spends_saved_step_time_on_urgent_prompts_only():
    """with both SLOs set, the time that steps stay under `tpot_slo` is credit (TPOT is a
    per-request mean) and an urgent prompt — waiting for more than `burst_urgent_frac · ttft_slo` — gets a longer step
    paid from it; a fresh prompt does not, and without credit nobody does."""
    import time
    p = SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True)

    def engine_with(credit_steps):
        eng = make_engine(max_seqs=8, max_tokens=4096, num_blocks=512)
        sc = eng.cfg.sched
        sc.ttft_slo, sc.tpot_slo, sc.max_num_batched_tokens, sc.long_prefill_chunk, sc.token_alignment = 10.0, 0.030, 4096, 4096, 0
        sc.max_model_len = 2048
        s = eng.scheduler
        for _ in range(credit_steps):                        # steps of 20 ms at 256 tokens: 8.5 ms under 0.95 · tpot_slo each
            s.observe_step(0.020, 256)
        s.dyn_budget = 256                                   # what the controller would have settled at
        return eng, s

    eng, s = engine_with(10)
    assert abs(s.credit - 10 * (0.95 * 0.030 - 0.020)) < 1e-9 and abs(s.tok_rate - 12800) < 1
    now = time.perf_counter()
    eng.add_request("urgent", prompt_ids=list(range(3, 1503)), params=p, arrival=now - 4.0)       # 40 % of its TTFT gone
    b = s.schedule()
    want = 256 + int(min(s.credit, (1500 - 256) / s.tok_rate) * s.tok_rate)
    assert b.burst_grant > 0 and abs(b.total_tokens - want) <= 1 and b.total_tokens > 256, (b.total_tokens, want)
    assert abs(s.granted_inflight - b.burst_grant) < 1e-12 and s.num_burst_steps == 1
    # credit promised to the step in flight is not promised twice
    eng.add_request("urgent2", prompt_ids=list(range(3, 1503)), params=p, arrival=now - 4.0)
    b2 = s.schedule()
    assert b2.burst_grant <= s.credit - b.burst_grant + 1e-9
    # the long step is judged against its own target (no collapse of the token target), and it costs credit
    before, budget = s.credit, s.dyn_budget
    s.observe_step(0.030 * 0.85 + b.burst_grant, b.total_tokens, b.burst_grant)
    assert s.dyn_budget >= budget and s.credit < before and abs(s.granted_inflight - b2.burst_grant) < 1e-12

    eng, s = engine_with(10)                                 # a fresh prompt: the usual prefill share
    eng.add_request("fresh", prompt_ids=list(range(3, 1503)), params=p, arrival=time.perf_counter())
    b = s.schedule()
    assert b.burst_grant == 0 and b.total_tokens == 256

    eng, s = engine_with(0)                                  # urgent but no credit
    s.tok_rate = 12800.0
    eng.add_request("urgent", prompt_ids=list(range(3, 1503)), params=p, arrival=time.perf_counter() - 4.0)
    b = s.schedule()
    assert b.burst_grant == 0 and b.total_tokens == 256


@pytest.mark.parametrize("model", ["tiny-llama", "tiny-mixtral"])
def test_fp8_w8a8_projections_track_the_bf16_model(model):
    """`EngineConfig.quantization = "fp8"`: an e4m3 copy of every dense projection (per-output-channel scales), activations
    quantised per token, used by steps of ≥ 17 rows (smaller steps: the GPU streams the e4m3 weights through the decode
    kernel, the CPU keeps the bf16 weights) — here through the fp32 reference of the same quantised operands
    (the GPU path runs them on the fp8 matrix cores: tests/test_kernels_gpu.py, test_engine_gpu.py).  The final hidden
    states of an 80-token prefill must track the bf16 model; a decode-sized step does not touch the fp8 copies."""
    from dli_amd.models import build_model
    from dli_amd.models.layers import Weight
    cfg = get_model_config(model)
    plain = build_model(cfg, None, "cpu", torch.bfloat16, 3)
    quant = build_model(cfg, None, "cpu", torch.bfloat16, 3, None, "fp8")
    qw = [m for m in quant.modules() if isinstance(m, Weight) and m.q is not None]
    assert len(qw) == cfg.num_layers * (2 if cfg.num_experts else 4) and not any(m.q is not None for m in plain.modules() if isinstance(m, Weight))
    w = qw[0]
    assert w.q.dtype == torch.float8_e4m3fn and w.q_scale.shape == (1, w.data.shape[0])
    deq = w.q.float() * w.q_scale.t()
    assert (deq - w.data.float()).abs().max() <= 0.07 * w.data.float().abs().max()      # e4m3: 3 mantissa bits

    def hidden(m, L):
        nb = (L + 31) // 32
        caches = ops.allocate_kv_cache(cfg.num_layers, nb, cfg.num_kv_heads, cfg.head_dim, torch.bfloat16, "cpu")
        meta = ops.AttnMetadata(num_decode=0, num_prefill=1, num_tokens=L, positions=torch.arange(L, dtype=torch.int32),
                                slot_mapping=torch.arange(L, dtype=torch.int32), block_tables=torch.arange(nb, dtype=torch.int32)[None],
                                seq_lens=torch.tensor([L], dtype=torch.int32), query_start_loc=torch.tensor([0, L], dtype=torch.int32))
        with torch.inference_mode():
            return m(torch.arange(5, 5 + L), meta, caches).float()
    a, b = hidden(plain, 80), hidden(quant, 80)
    cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
    assert 0.99 < cos < 1.0, cos                              # close, and not the same numbers: the fp8 path did run
    assert torch.equal(hidden(plain, 12), hidden(quant, 12))  # 12 rows < FP8_MIN_ROWS: the bf16 weights serve the step
    with pytest.raises(ValueError):
        build_model(get_model_config("tiny-opt"), None, "cpu", torch.bfloat16, 3, None, "fp8")


def test_decode_sized_gemm_dispatch_rule():
    """ops.linear: which (rows, N, K) go to the weight-streaming kernel (csrc/kernels/skinny_gemm.hip)."""
    from dli_amd import ops
    ok = ops.skinny_rows_ok
    assert ok(1, 4096, 4096) and ok(16, 28672, 4096) and ok(16, 128256, 4096)       # ≤ 16 rows: always
    assert not ok(17, 4096, 4096) and not ok(32, 6144, 4096)                         # 32 MB+ matrices: library above 16
    assert ok(32, 8192, 1024) and ok(64, 4096, 1024) and ok(32, 1536, 4096)          # small TP shards: 16-row passes
    assert not ok(64, 8192, 1024) and not ok(65, 1024, 1024) and not ok(0, 4096, 4096)
    assert not ok(4, 4096, 4000) and not ok(4, 4100, 4096)                           # K % 128, N % 16
    assert ok(4, 28672, 4096, 32) and not ok(4, 28688, 4096, 32)                     # fused gate/up: N % 32
    x = torch.randn(4, 256).bfloat16()
    w = torch.randn(64, 256).bfloat16()
    torch.testing.assert_close(ops.linear(x, w), torch.nn.functional.linear(x, w))   # CPU tensors: library path
    want = torch.nn.functional.silu(x.float() @ w[:32].float().t()) * (x.float() @ w[32:].float().t())
    torch.testing.assert_close(ops.linear_swiglu(x, w).float(), want, rtol=2e-2, atol=2e-2)


def test_gemm256_swiglu_dispatch_rule():
    """Pure function: which (M, I, K) take the fused large-M gate/up kernel (measured crossovers, profiles/gemm256_r2e.jsonl)."""
    from dli_amd.ops import gemm256_swiglu_ok
    assert gemm256_swiglu_ok(4096, 14336, 4096) and gemm256_swiglu_ok(512, 14336, 4096)      # Llama-3-8B
    assert gemm256_swiglu_ok(2048, 3584, 8192) and not gemm256_swiglu_ok(1024, 3584, 8192)   # Llama-3-70B TP=8 shard
    assert not gemm256_swiglu_ok(300, 14336, 4096)          # second row tile would be mostly padding
    assert not gemm256_swiglu_ok(64, 14336, 4096) and not gemm256_swiglu_ok(4096, 1000, 4096)


@pytest.mark.parametrize("plen", [70, 64, 65, 5])
def test_n_samples_fork_off_one_prefill(plen):
    """OpenAI `n`: the prompt is prefilled once; the other samples are forked off its KV pages (ref-counted, private
    copy of the page with the last prompt token) and recompute only that token.  Greedy children must reproduce the
    parent exactly — which they only do if the copied / shared pages hold the right K/V — for prompt lengths that end
    inside a page, on a page boundary, and one past it."""
    from dli_amd.config import CacheConfig, EngineConfig, SamplingParams, SchedulerConfig, get_model_config
    from dli_amd.engine.engine import LLMEngine
    cfg = EngineConfig(model=get_model_config("tiny-llama"), cache=CacheConfig(num_blocks=64),
                       sched=SchedulerConfig(max_num_seqs=8, max_num_batched_tokens=128, max_model_len=256),
                       device="cpu", use_hip_graphs=False)
    rng = np.random.default_rng(plen)
    prompt = rng.integers(3, 500, size=plen).tolist()
    greedy = SamplingParams(temperature=0.0, max_tokens=12, ignore_eos=True)
    want = LLMEngine(cfg).generate(None, greedy, prompt_ids=[prompt])[0]

    eng = LLMEngine(cfg)
    eng.add_request("r", prompt_ids=prompt, params=greedy, n=3)
    outs = {}
    while eng.has_unfinished():
        for o in eng.step():
            outs.setdefault((o.parent_id or o.req_id, o.sample_index), []).extend(o.new_token_ids)
    assert sorted(outs) == [("r", 0), ("r", 1), ("r", 2)]
    assert all(v == want for v in outs.values()), (outs, want)
    # one prefill + one recomputed token per child + the decode steps: far less than three prefills
    assert eng.metrics.step_tokens == plen + 2 + 3 * 11
    assert eng.bm.num_free == eng.bm.num_blocks and not eng.requests

    # sampled: every choice draws from its own RNG stream
    eng2 = LLMEngine(cfg)
    eng2.add_request("s", prompt_ids=prompt, params=SamplingParams(temperature=1.0, max_tokens=12, ignore_eos=True, seed=3), n=4)
    outs2 = {}
    while eng2.has_unfinished():
        for o in eng2.step():
            outs2.setdefault(o.sample_index, []).extend(o.new_token_ids)
    assert len(outs2) == 4 and len({tuple(v) for v in outs2.values()}) >= 3


def test_n_samples_without_room_for_private_pages_queue_as_plain_requests():
    from dli_amd.config import CacheConfig, EngineConfig, SamplingParams, SchedulerConfig, get_model_config
    from dli_amd.engine.engine import LLMEngine
    cfg = EngineConfig(model=get_model_config("tiny-llama"), cache=CacheConfig(num_blocks=6),
                       sched=SchedulerConfig(max_num_seqs=8, max_num_batched_tokens=128, max_model_len=128),
                       device="cpu", use_hip_graphs=False)
    eng = LLMEngine(cfg)
    assert eng.scheduler.watermark_blocks == 1               # 6 pages, 3 taken by the prompt, 1 reserved: no room for 3 forks
    p = SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True)
    eng.add_request("r", prompt_ids=list(range(3, 93)), params=p, n=4)
    done = {}
    for _ in range(400):
        if not eng.has_unfinished():
            break
        for o in eng.step():
            if o.finished:
                done[o.sample_index] = o.finish_reason
    assert done == {0: "length", 1: "length", 2: "length", 3: "length"}
    assert eng.bm.num_free == eng.bm.num_blocks


def test_hf_stream_decoder_is_incremental_and_exact(tmp_path):
    """The streaming detokeniser re-decodes a short window per token (not the whole id list) and still reproduces the
    text exactly, multi-byte characters split over several byte-level tokens included."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from dli_amd.engine.tokenizer import HFTokenizer
    tk = Tokenizer(models.BPE(unk_token=None))
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tk.decoder = decoders.ByteLevel()
    corpus = ["the quick brown fox jumps over the lazy dog " * 3, "distributed inference engines batch tokens"] * 10
    tk.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=300, initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    path = str(tmp_path / "tokenizer.json")
    tk.save(path)
    tok = HFTokenizer(path, 1, 2)
    calls = []
    real = tok.decode
    tok.decode = lambda ids: (calls.append(len(ids)), real(ids))[1]
    text = "héllo wörld ✓ 漢字 🙂 the quick brown fox — " * 12
    ids = tok.encode(text, add_bos=False)
    assert len(ids) > 200
    dec = tok.stream_decoder()
    out = "".join(dec.step(i) for i in ids) + dec.flush()
    assert out == text
    assert max(calls) <= 32, "a decode call must never see more than a short window of ids"


def test_fp8_kv_scale_calibration_matters_when_kv_values_are_small():
    """tiny-llama K/V values are ≈ 0.1: at unit scale e4m3 stores them in its subnormals and greedy decoding falls apart;
    with the warm-up calibration (per-layer, per-kv-head amax / 448) it follows the exact-KV run."""
    from dli_amd.config import CacheConfig, EngineConfig, SamplingParams, SchedulerConfig, get_model_config
    from dli_amd.engine.engine import LLMEngine

    def run(kv, calib=True):
        cfg = EngineConfig(model=get_model_config("tiny-llama"),
                           cache=CacheConfig(num_blocks=64, kv_dtype=kv, calibrate_fp8_scales=calib),
                           sched=SchedulerConfig(max_num_seqs=8, max_num_batched_tokens=256, max_model_len=256),
                           device="cpu", use_hip_graphs=False)
        eng = LLMEngine(cfg)
        eng.warmup()
        rng = np.random.default_rng(0)
        prompts = [rng.integers(3, 500, size=n).tolist() for n in (40, 100, 7)]
        return eng.generate(None, SamplingParams(temperature=0.0, max_tokens=12, ignore_eos=True), prompt_ids=prompts), eng

    agree = lambda x, y: float(np.mean([np.mean(np.array(p) == np.array(q)) for p, q in zip(x, y)]))
    exact, _ = run("auto")
    cal, eng = run("fp8")
    unit, _ = run("fp8", calib=False)
    st = eng.runner.selftest["fp8_kv_scales"]
    assert 0 < st["k_scale_max"] < 0.05 and 0 < st["v_scale_max"] < 0.05          # far below the unit scale
    assert agree(exact, cal) >= 0.8 and agree(exact, cal) > agree(exact, unit) + 0.3


def test_debug_bounds_check_passes_on_good_steps_and_catches_a_short_cache(monkeypatch):
    """DLI_DEBUG_BOUNDS=1: every step's slots / block tables / token ids are checked against THIS rank's
    cache before a kernel sees them — the failure mode of a TP worker with fewer KV blocks than the driver."""
    from dli_amd.runtime import model_runner
    monkeypatch.setattr(model_runner, "_DEBUG_BOUNDS", True)
    eng = make_engine("tiny-llama")
    rng = np.random.default_rng(5)
    prompts = [rng.integers(3, 500, size=n).tolist() for n in (40, 90, 7)]
    assert eng.generate(None, GREEDY, prompt_ids=prompts) == [dense_greedy(eng, p, 12) for p in prompts]
    eng.runner.num_blocks = 1                              # pretend this rank's cache is smaller than the block manager's
    eng.add_request("x", prompt_ids=prompts[1], params=GREEDY)
    with pytest.raises(AssertionError, match="out of range|block table"):
        while eng.has_unfinished():
            eng.step()


def test_decode_rows_are_ordered_longest_context_first():
    """The decode-attention grid is dispatched in row order: the scheduler hands it the longest contexts first."""
    eng = make_engine("tiny-llama")
    rng = np.random.default_rng(9)
    for i, n in enumerate((5, 60, 20, 40)):
        eng.add_request(f"r{i}", prompt_ids=rng.integers(3, 500, size=n).tolist(), params=GREEDY)
    for _ in range(4):
        eng.step()
    b = eng.scheduler.schedule()
    assert b.num_decode == 4
    ctx = [r.num_computed for r in b.requests[:b.num_decode]]
    assert ctx == sorted(ctx, reverse=True) and len(set(ctx)) == 4


def test_n_samples_never_exceed_the_sequence_cap():
    """a request with n > 1 completing its prompt while the replica is at max_num_seqs must not push
    the running set past the cap (the step buffers have max_num_seqs rows) — the extra children queue at the head of
    the waiting line and still produce the parent's greedy continuation."""
    eng = make_engine(max_seqs=2, max_tokens=128, num_blocks=64)
    greedy = SamplingParams(temperature=0.0, max_tokens=10, ignore_eos=True)
    rng = np.random.default_rng(5)
    other = rng.integers(3, 500, size=20).tolist()
    prompt = rng.integers(3, 500, size=40).tolist()
    want = make_engine(max_seqs=2, max_tokens=128).generate(None, greedy, prompt_ids=[prompt])[0]
    eng.add_request("other", prompt_ids=other, params=SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True))
    eng.step()
    eng.add_request("r", prompt_ids=prompt, params=greedy, n=4)
    outs, peak = {}, 0
    for _ in range(600):
        if not eng.has_unfinished():
            break
        for o in eng.step():
            outs.setdefault((o.parent_id or o.req_id, o.sample_index), []).extend(o.new_token_ids)
        peak = max(peak, len(eng.scheduler.running))
    assert not eng.has_unfinished() and peak <= 2
    assert all(outs[("r", i)] == want for i in range(4)), outs
    assert eng.bm.num_free == eng.bm.num_blocks and not eng.requests


def test_copy_on_write_overflow_queues_children_instead_of_raising():
    eng = make_engine(max_seqs=8, max_tokens=128, num_blocks=64)
    eng.scheduler.max_copies_per_step = 1
    greedy = SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True)
    eng.add_request("r", prompt_ids=list(range(3, 43)), params=greedy, n=4)
    done = {}
    for _ in range(200):
        if not eng.has_unfinished():
            break
        for o in eng.step():
            done.setdefault(o.sample_index, []).extend(o.new_token_ids)
    assert len(done) == 4 and len({tuple(v) for v in done.values()}) == 1
    assert eng.bm.num_free == eng.bm.num_blocks


def test_shed_parent_with_n_samples_reports_every_choice():
    """a parent shed before it forked must close all n choices (the stream waits for n finishes)."""
    import time as _t
    eng = make_engine(max_seqs=1, max_tokens=64)
    eng.cfg.sched.max_queue_delay = 0.05
    long = SamplingParams(temperature=0.0, max_tokens=100, ignore_eos=True)
    eng.add_request("busy", prompt_ids=[5, 6, 7], params=long)
    eng.step()
    eng.add_request("r", prompt_ids=[9, 10, 11], params=SamplingParams(temperature=0.0, max_tokens=4), n=3)
    _t.sleep(0.08)
    shed = []
    for _ in range(5):
        shed += [(o.parent_id or o.req_id, o.sample_index, o.finish_reason) for o in eng.step() if o.finished]
    assert sorted(shed) == [("r", 0, "shed"), ("r", 1, "shed"), ("r", 2, "shed")]
    assert "r" not in eng.requests


@pytest.mark.parametrize("pipelined", [False, True])
def test_comm_fault_discards_the_suspect_steps_and_recomputes(pipelined):
    """when a step turns out to contain a timed-out all-reduce, neither its tokens nor those of the
    step queued behind it reach the client, the pages they wrote are not published, and the output equals a clean run."""
    greedy = SamplingParams(temperature=0.0, max_tokens=16, ignore_eos=True)
    rng = np.random.default_rng(11)
    prompts = [rng.integers(3, 500, size=n).tolist() for n in (40, 70, 9)]
    want = make_engine(async_scheduling=pipelined).generate(None, greedy, prompt_ids=prompts)

    eng = make_engine(async_scheduling=pipelined)
    calls = {"n": 0}
    real = eng.runner.collect

    def collect(handle):
        out = real(handle)
        calls["n"] += 1
        if calls["n"] == 6:                                  # mid-generation: poison this step
            eng.runner._comm_fault = True
            out = np.full_like(out, 1)                       # what a partial sum would look like: wrong ids
        return out
    eng.runner.collect = collect
    for i, p in enumerate(prompts):
        eng.add_request(f"q{i}", prompt_ids=p, params=greedy)
    got = {}
    while eng.has_unfinished():
        for o in eng.step():
            got.setdefault(o.req_id, []).extend(o.new_token_ids)
    assert [got[f"q{i}"] for i in range(3)] == want
    assert eng.metrics.comm_fault_recoveries == 1
    assert eng.bm.num_free == eng.bm.num_blocks


def test_decode_work_list_covers_every_page_once_and_respects_caps():
    """ops.build_decode_work: equal-size page ranges per row, every page in exactly one range, partial slots dense and
    consistent with the reduce list, item count under the cap (chunks double until it fits)."""
    rng = np.random.default_rng(3)
    for n in (1, 7, 300, 1500):
        L = np.clip(rng.lognormal(6.1, 0.8, size=n), 1, 2048).astype(np.int64)
        pages = (L + 31) // 32
        for cap in (0, 3 * n, n + 5, n):
            work, red, n_parts = ops.build_decode_work(L, 8, cap)
            assert cap == 0 or len(work) <= max(cap, n)
            cover = np.zeros(n, dtype=np.int64)
            np.add.at(cover, work[:, 0], work[:, 2] - work[:, 1])
            assert (cover == pages).all() and (work[:, 2] > work[:, 1]).all()
            multi = work[work[:, 3] >= 0]
            assert len(multi) == n_parts and sorted(multi[:, 3].tolist()) == list(range(n_parts))
            for row, first, k, _ in red:
                mine = work[work[:, 0] == row]
                assert len(mine) == k and mine[:, 3].tolist() == list(range(first, first + k))
            single = work[work[:, 3] < 0]
            assert len(set(single[:, 0].tolist()) & set(red[:, 0].tolist())) == 0
            if n * 8 >= 512:
                assert len(work) == n and len(red) == 0     # big batches: one range per row (measured fastest), no reduce
            elif cap == 0:
                assert len(work) > n                        # small batches: long rows are cut to fill the chip
    items, reds = ops.decode_launch_caps(8, 8)
    L = np.array([40, 2048, 2048, 100, 5, 2048, 700, 2048])
    work, red, _ = ops.build_decode_work(L, 8, items)
    assert len(work) <= items and len(red) <= reds and (np.bincount(work[:, 0])[[0, 3, 4]] == 1).all()
