"""End-to-end GPU checks of the engine on a small D=128 Llama: the HIP-kernel
forward against the fp32 torch-reference forward, hipGraph decode against eager decode, fp8 KV."""
import os
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


def test_mixed_step_graphs_equal_eager_and_cover_the_prefill_steps():
    """steps that hold prefill chunks replay a body graph (token bucket) + a head graph (sampled-row
    bucket) with all attention work taken from device-side lists.  Requests arrive staggered so that most steps mix
    running decodes with new prefill chunks; the greedy output must match the eager engine's, nearly every step must
    be a graph replay, and the start-up self-test pins a mixed step graph-vs-eager at logit level."""
    p = SamplingParams(temperature=0.0, max_tokens=24, ignore_eos=True)
    prompts = _prompts(7, (5, 70, 33, 200, 130, 90, 260, 17))

    def run(graphs):
        eng = make(graphs=graphs, buckets=(1, 2, 4, 8), chunk=64)
        eng.warmup()
        outs = {}
        nxt = 0
        while eng.has_unfinished() or nxt < len(prompts):
            if nxt < len(prompts) and (not eng.has_unfinished() or eng.metrics.steps % 3 == 0):
                eng.add_request(f"r{nxt}", prompt_ids=prompts[nxt], params=p)
                nxt += 1
            for o in eng.step():
                outs.setdefault(o.req_id, []).extend(o.new_token_ids)
        return eng, [outs[f"r{i}"] for i in range(len(prompts))]

    _, eager = run(False)
    eng, graph = run(True)
    st = eng.runner.stats
    assert st["mixed_graph_steps"] >= 10 and st["eager_steps"] <= 2, st
    agree = np.mean([np.mean(np.array(a) == np.array(b)) for a, b in zip(eager, graph)])
    assert all(a[:4] == b[:4] for a, b in zip(eager, graph)) and agree > 0.9, (agree, eager, graph)
    sm = eng.runner.selftest["mixed_graph_vs_eager"]
    assert sm["ok"] is True and sm["greedy_agreement"] == 1.0 and sm["max_abs_logit_diff"] <= 2e-3 * max(sm["logit_scale"], 1.0), sm


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


def test_fp8_kv_calibrated_scales_track_bf16_kv_on_real_shapes():
    """Llama-3-8B layer shapes (4 layers, real vocabulary): with per-(layer, kv head) scales calibrated at warm-up the
    fp8 (e4m3) KV cache tracks the bf16-KV logits at the precision e4m3 allows, where unit scales do not (random-init
    K/V are small: unit-scaled e4m3 pushes them into its subnormals).  What "tracks" can mean here: e4m3 keeps 3 mantissa
    bits (≈ 3.6 % rms relative error per stored element), and with RANDOM weights the attention output dominates the
    residual stream (it is not a small correction to a large skip path as in a trained model), so that error reaches the
    logits almost undamped: 1 − cos ≈ 0.7 % after 4 layers, and top-1 of the near-uniform random logits agrees on 74 %.
    Thresholds are set to that floor (measured: cosine min 0.989 / mean 0.993, printed below); a trained checkpoint is needed to pin a tighter, quality-level bound —
    parity with a published fp8-KV accuracy figure is unpinned (no checkpoint or eval set in this environment)."""
    rng = np.random.default_rng(5)
    ids = rng.integers(3, 100000, size=768).tolist()
    out = {}
    for name, kv, calib in (("bf16", "auto", True), ("fp8", "fp8", True), ("fp8_unit", "fp8", False)):
        cfg = EngineConfig(model=get_model_config("llama3-8b", num_layers=4),
                           cache=CacheConfig(num_blocks=256, kv_dtype=kv, calibrate_fp8_scales=calib),
                           sched=SchedulerConfig(max_num_seqs=8, max_num_batched_tokens=2048, max_model_len=2048),
                           device="cuda", use_hip_graphs=False)
        eng = LLMEngine(cfg)
        eng.warmup()
        out[name] = eng.runner.prefill_logits(ids).cpu()
        if name == "fp8":
            st = eng.runner.selftest["fp8_kv_scales"]
            assert 0 < st["k_scale_min"] <= st["k_scale_max"] < 1.0 and 0 < st["v_scale_min"] < 1.0, st
        del eng
        torch.cuda.empty_cache()
    cos = torch.nn.functional.cosine_similarity(out["fp8"], out["bf16"], dim=-1)
    cos_unit = torch.nn.functional.cosine_similarity(out["fp8_unit"], out["bf16"], dim=-1)
    top1 = (out["fp8"].argmax(-1) == out["bf16"].argmax(-1)).float().mean().item()
    top1_unit = (out["fp8_unit"].argmax(-1) == out["bf16"].argmax(-1)).float().mean().item()
    print(f"fp8 KV vs bf16 KV: cosine min {cos.min():.5f} mean {cos.mean():.5f}, top-1 {top1:.4f}; "
          f"unit scales: cosine min {cos_unit.min():.5f}, top-1 {top1_unit:.4f}")
    assert cos.min() >= 0.98 and cos.mean() >= 0.99, (cos.min(), cos.mean())
    # at these shapes random-init K/V have std ≈ 1.3, inside e4m3's normal range even at unit scale: calibration must
    # not cost anything here (tests/test_engine.py shows the case where it is the difference between right and wrong)
    assert cos.mean() >= cos_unit.mean() - 2e-3 and top1 >= top1_unit - 0.03


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


def test_fp8_w8a8_engine_on_gpu_tracks_the_bf16_engine(tmp_path):
    """`quantization="fp8"` end to end on the GPU: prefill and mixed steps of ≥ 65 rows run the dense projections on the fp8
    matrix cores (captured in the mixed-step hipGraphs, graph-vs-eager self-test included), decode-sized steps stay on the
    bf16 weight-streaming kernels.  First-step logits track the bf16 engine; greedy tokens mostly agree (random-init
    weights: near-uniform logits, so token agreement is a loose bound — parity on a trained checkpoint is unpinned here)."""
    import dli_amd.runtime.model_runner as mr
    from dli_amd.config import CacheConfig, EngineConfig, SamplingParams, SchedulerConfig, get_model_config
    from dli_amd.engine.engine import LLMEngine
    rng = np.random.default_rng(9)
    prompts = [rng.integers(3, 1000, size=n).tolist() for n in (300, 90, 150)]
    outs, logits = {}, {}
    for name, quant in (("bf16", None), ("fp8", "fp8")):
        f = str(tmp_path / f"logits_{name}.pt")
        mr._DUMP_LOGITS = f                                   # (read from DLI_DUMP_LOGITS at import time)
        try:
            cfg = EngineConfig(model=get_model_config("tiny-llama-tp"), cache=CacheConfig(num_blocks=256),
                               sched=SchedulerConfig(max_num_seqs=8, max_num_batched_tokens=1024, max_model_len=1024,
                                                     long_prefill_chunk=1024),
                               device="cuda", graph_batch_sizes=(4, 8), quantization=quant)
            eng = LLMEngine(cfg)
            eng.warmup()
            outs[name] = eng.generate(None, SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True), prompt_ids=prompts)
            st = eng.runner.selftest
            assert st["graph_vs_eager"]["ok"] and st["mixed_graph_vs_eager"]["ok"], st
            assert eng.runner.stats["mixed_graph_steps"] > 0
            eng.shutdown()
        finally:
            mr._DUMP_LOGITS = ""
        logits[name] = torch.load(f, weights_only=True)
    cos = torch.nn.functional.cosine_similarity(logits["bf16"].flatten().float(), logits["fp8"].flatten().float(), dim=0).item()
    assert 0.99 < cos < 1.0, cos
    first = np.mean([a[0] == b[0] for a, b in zip(outs["bf16"], outs["fp8"])])
    assert first >= 2 / 3, (outs, first)


def test_library_gemm_tuning_writes_and_reloads_its_file(tmp_path):
    """`--tune-library-gemms FILE` (PyTorch TunableOp under the measured dispatch): a first start tunes the vendor GEMMs of the
    captured shapes and writes FILE, a second start loads it; both serve.  (Own process: TunableOp is process-global state.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f = str(tmp_path / "tunable.csv")
    lines = []
    for _ in range(2):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--model", "tiny-llama-tp", "--steps", "5", "--warmup", "2", "--qps", "20",
                            "--ramp-seconds", "2", "--ramp-cap-seconds", "2", "--sustained-seconds", "2", "--max-num-seqs", "128",
                            "--max-batched-tokens", "512", "--tune-library-gemms", f], capture_output=True, text=True, timeout=250, cwd=root)
        assert p.returncode == 0, p.stderr[-3000:]
        lines.append(json.loads(p.stdout.strip().splitlines()[-1]))
    for line in lines:
        lt = line["selftest"]["gemm_plan"]["library_tuning"]
        assert lt["file"] == f and lt["entries"] >= 1 and line["rank0"]["finished"] >= 0
    assert os.path.exists(f) and os.path.getsize(f) > 0
    assert lines[1]["selftest"]["gemm_plan"]["library_tuning"]["seconds"] <= lines[0]["selftest"]["gemm_plan"]["library_tuning"]["seconds"] + 5
