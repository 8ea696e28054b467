"""End-to-end GPU checks of the engine on a small D=128 Llama: the HIP-kernel
forward against the fp32 torch-reference forward, hipGraph decode against eager decode, fp8 KV."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dli_amd import ops
from dli_amd.config import CacheConfig, EngineConfig, SamplingParams, SchedulerConfig, get_model_config
from dli_amd.engine.engine import LLMEngine


def make(graphs, kv="auto", buckets=(4, 8), chunk=64, async_mode=None):
    cfg = EngineConfig(model=get_model_config("tiny-llama-d128"), cache=CacheConfig(num_blocks=512, kv_dtype=kv),
                       sched=SchedulerConfig(max_num_seqs=8, max_num_batched_tokens=256, max_model_len=1024,
                                             long_prefill_chunk=chunk),
                       device="cuda", use_hip_graphs=graphs, graph_batch_sizes=buckets, async_scheduling=async_mode)
    return LLMEngine(cfg)


def _prompts(seed=0, sizes=(5, 70, 33, 200, 130)):
    rng = np.random.default_rng(seed)
    return [rng.integers(3, 1000, size=n).tolist() for n in sizes]


def _ref_logits(eng, toks):
    """fp32 torch-reference forward of one whole sequence (CPU copy of the same weights)."""
    from dli_amd.models import build_model
    cpu = build_model(eng.cfg.model, device="cpu", dtype=torch.float32, seed=eng.cfg.seed)
    for (_, w_cpu, _), (_, w_gpu, _) in zip(cpu.named_weights(), eng.runner.model.named_weights()):
        w_cpu.data.copy_(w_gpu.data.float().cpu())
    L = len(toks)
    nb = (L + 31) // 32
    caches = ops.allocate_kv_cache(eng.cfg.model.num_layers, nb, 1, 128, torch.float32, "cpu")
    meta = ops.AttnMetadata(0, 1, L, torch.arange(L, dtype=torch.int32), torch.arange(L, dtype=torch.int32),
                            torch.arange(nb, dtype=torch.int32)[None], torch.tensor([L], dtype=torch.int32),
                            torch.tensor([0, L], dtype=torch.int32))
    with torch.inference_mode():
        return cpu.compute_logits(cpu(torch.tensor(toks), meta, caches)).float()


def test_hip_forward_matches_reference_forward():
    eng = make(graphs=False)
    toks = _prompts(3, (150,))[0]
    want = _ref_logits(eng, toks)
    r = eng.runner
    L = len(toks)
    nb = (L + 31) // 32
    caches = ops.allocate_kv_cache(eng.cfg.model.num_layers, nb, 1, 128, torch.bfloat16, "cuda")
    ts, t0 = ops.build_prefill_tiles(np.array([L]), 0, r.tile_tokens)
    dev = "cuda"
    meta = ops.AttnMetadata(0, 1, L, torch.arange(L, dtype=torch.int32, device=dev),
                            torch.arange(L, dtype=torch.int32, device=dev),
                            torch.arange(nb, dtype=torch.int32, device=dev)[None].contiguous(),
                            torch.tensor([L], dtype=torch.int32, device=dev),
                            torch.tensor([0, L], dtype=torch.int32, device=dev),
                            prefill_tile_seq=torch.from_numpy(ts).to(dev), prefill_tile_q0=torch.from_numpy(t0).to(dev),
                            num_prefill_tiles=len(ts))
    with torch.inference_mode():
        got = r.model.compute_logits(r.model(torch.tensor(toks, device=dev), meta, caches)).float().cpu()
    cos = torch.nn.functional.cosine_similarity(got, want, dim=-1)
    assert cos.min() > 0.995, cos.min()
    assert (got.argmax(-1) == want.argmax(-1)).float().mean() > 0.9


def test_graph_decode_equals_eager_decode():
    p = SamplingParams(temperature=0.0, max_tokens=24, ignore_eos=True)
    eager = make(graphs=False).generate(None, p, prompt_ids=_prompts())
    eng = make(graphs=True)
    graph = eng.generate(None, p, prompt_ids=_prompts())
    assert eng.runner.stats["graph_steps"] >= 20
    agree = np.mean([np.mean(np.array(a) == np.array(b)) for a, b in zip(eager, graph)])
    assert all(a[:4] == b[:4] for a, b in zip(eager, graph)) and agree > 0.9, (agree, eager, graph)


def test_startup_selftest_pins_graph_replay_to_eager_at_logit_level():
    """The start-up self-test replays a captured decode bucket against the same staged batch run eagerly: same
    kernels, same inputs → the logits must agree to bf16 noise (and the engine keeps its graphs)."""
    eng = make(graphs=True)
    eng.warmup()
    st = eng.runner.selftest["graph_vs_eager"]
    assert st["ok"] is True and eng.cfg.use_hip_graphs and eng.runner.graphs
    assert st["max_abs_logit_diff"] <= 2e-3 * max(st["logit_scale"], 1.0), st
    assert st["greedy_agreement"] == 1.0


def test_chunked_prefill_equals_single_shot_on_gpu():
    p = SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)
    a = make(graphs=False, chunk=32).generate(None, p, prompt_ids=_prompts(5))
    b = make(graphs=False, chunk=1024).generate(None, p, prompt_ids=_prompts(5))
    agree = np.mean([np.mean(np.array(x) == np.array(y)) for x, y in zip(a, b)])
    assert agree > 0.9, (a, b)


def test_fp8_kv_cache_engine_runs_and_tracks_bf16():
    p = SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)
    a = make(graphs=True, kv="fp8").generate(None, p, prompt_ids=_prompts(6))
    b = make(graphs=True).generate(None, p, prompt_ids=_prompts(6))
    assert all(len(x) == 8 for x in a)
    assert np.mean([x[0] == y[0] for x, y in zip(a, b)]) >= 0.6      # first tokens mostly agree at e4m3 precision


def test_seeded_sampling_reproducible_across_batch_composition():
    p = SamplingParams(temperature=0.9, top_p=0.95, max_tokens=12, ignore_eos=True, seed=11)
    eng = make(graphs=True)
    alone = eng.generate(None, p, prompt_ids=_prompts(7, (40,)))[0]
    rng_prompts = _prompts(7, (40,)) + _prompts(8, (9, 90))
    eng2 = make(graphs=True)
    eng2.add_request("a", prompt_ids=rng_prompts[0], params=p)
    for i, q in enumerate(rng_prompts[1:]):
        eng2.add_request(f"o{i}", prompt_ids=q, params=SamplingParams(temperature=0.5, max_tokens=20, ignore_eos=True))
    ra = eng2.requests["a"]
    while eng2.has_unfinished():
        eng2.step()
    assert np.mean(np.array(alone) == ra.output_ids) > 0.8           # same Philox stream whatever the batch


def test_mixtral_engine_on_gpu_uses_hip_moe():
    cfg = EngineConfig(model=get_model_config("tiny-mixtral-d128"), cache=CacheConfig(num_blocks=256),
                       sched=SchedulerConfig(max_num_seqs=8, max_num_batched_tokens=256, max_model_len=1024),
                       device="cuda", graph_batch_sizes=(4, 8))
    eng = LLMEngine(cfg)
    p = SamplingParams(temperature=0.0, max_tokens=10, ignore_eos=True)
    outs = eng.generate(None, p, prompt_ids=_prompts(9, (20, 150, 64)))
    assert all(len(o) == 10 for o in outs) and eng.runner.stats["graph_steps"] > 0
    # logits of the HIP path track the fp32 reference model
    toks = _prompts(10, (90,))[0]
    want = _ref_logits(eng, toks)
    L = len(toks)
    nb = (L + 31) // 32
    r = eng.runner
    caches = ops.allocate_kv_cache(cfg.model.num_layers, nb, 1, 128, torch.bfloat16, "cuda")
    ts, t0 = ops.build_prefill_tiles(np.array([L]), 0, r.tile_tokens)
    meta = ops.AttnMetadata(0, 1, L, torch.arange(L, dtype=torch.int32, device="cuda"),
                            torch.arange(L, dtype=torch.int32, device="cuda"),
                            torch.arange(nb, dtype=torch.int32, device="cuda")[None].contiguous(),
                            torch.tensor([L], dtype=torch.int32, device="cuda"),
                            torch.tensor([0, L], dtype=torch.int32, device="cuda"),
                            prefill_tile_seq=torch.from_numpy(ts).cuda(), prefill_tile_q0=torch.from_numpy(t0).cuda(),
                            num_prefill_tiles=len(ts))
    with torch.inference_mode():
        got = r.model.compute_logits(r.model(torch.tensor(toks, device="cuda"), meta, caches)).float().cpu()
    assert torch.nn.functional.cosine_similarity(got, want, dim=-1).min() > 0.99


@pytest.mark.parametrize("graphs", [False, True])
def test_pipelined_stepping_on_gpu_is_bit_identical(graphs):
    """Step N+1 queued behind step N (ids fed on the device by gather_sampled) gives exactly the tokens of
    the one-step-at-a-time engine: same kernels on the same inputs, only the host timing differs."""
    greedy = SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True)
    sampled = SamplingParams(temperature=0.8, top_p=0.9, max_tokens=33, ignore_eos=True, seed=5)

    def run(async_mode):
        eng = make(graphs=graphs, async_mode=async_mode)
        assert eng.async_scheduling == async_mode
        reqs = []
        for i, ids in enumerate(_prompts(12, (5, 70, 33, 200, 130, 64))):
            reqs.append(eng.add_request(f"r{i}", prompt_ids=ids, params=greedy if i % 2 else sampled))
        while eng.has_unfinished():
            eng.step()
        return [r.output_ids.tolist() for r in reqs], eng

    a, _ = run(False)
    b, eng = run(True)
    assert a == b
    assert all(len(x) in (33, 40) for x in b)
    if graphs:
        assert eng.runner.stats["graph_steps"] > 20
    assert eng.bm.num_free == eng.bm.num_blocks
