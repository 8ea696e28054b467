"""Helper for test_rccl_gpu.py: the engine with a TP "group" of ONE rank on the `nccl` (= RCCL) backend — every
collective of the model, the step protocol and the hipGraphs goes through the library although nothing is reduced."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.distributed as dist

from dli_amd.config import CacheConfig, EngineConfig, ParallelConfig, SamplingParams, SchedulerConfig, get_model_config
from dli_amd.engine.engine import LLMEngine
from dli_amd.parallel.state import drain_collectives, init_distributed, make_tp_context


def main():
    use_rccl = sys.argv[1] == "rccl"
    prompts = json.loads(sys.argv[2])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ctx = None
    extra = {}
    if use_rccl:
        os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"))
        init_distributed("nccl", dev)
        ctx = make_tp_context(1, device=dev, one_rank_group=True)
        # the library calls on their own: eager and captured in a hipGraph
        x = torch.arange(4096, device=dev, dtype=torch.float32).to(torch.bfloat16).reshape(16, 256)
        y = x.clone()
        dist.all_reduce(y, group=ctx.group)
        g_out = torch.empty_like(x)
        dist.all_gather_into_tensor(g_out, x, group=ctx.group)
        a2a = torch.empty_like(x)
        dist.all_to_all_single(a2a, x, [16], [16], group=ctx.group)
        b = x.clone()
        dist.broadcast(b, src=0, group=ctx.group)
        torch.cuda.synchronize()
        extra["eager_collectives_identity"] = bool(torch.equal(y, x) and torch.equal(g_out, x) and torch.equal(a2a, x) and torch.equal(b, x))
        z = x.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            dist.all_reduce(z, group=ctx.group)              # warm-up on the capture stream
        drain_collectives()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
            dist.all_reduce(z, group=ctx.group)
            w = z * 2
        z.copy_(x)
        graph.replay()
        torch.cuda.synchronize()
        extra["captured_all_reduce_replays"] = bool(torch.equal(w, x * 2))
    cfg = EngineConfig(model=get_model_config("tiny-llama-tp"), cache=CacheConfig(num_blocks=512),
                       sched=SchedulerConfig(max_num_seqs=8, max_num_batched_tokens=2048, max_model_len=2048, long_prefill_chunk=2048),
                       parallel=ParallelConfig(tp=1, backend="nccl"), device="cuda", graph_batch_sizes=(4, 8))
    eng = LLMEngine(cfg, ctx, dev)
    eng.warmup()
    from dli_amd.models.layers import RowParallelLinear
    warm = RowParallelLinear.overlap_calls
    out = eng.generate(None, SamplingParams(temperature=0.0, max_tokens=10, ignore_eos=True), prompt_ids=prompts)
    info = dict(extra, greedy=out, graph_steps=eng.runner.stats["graph_steps"], mixed_graph_steps=eng.runner.stats["mixed_graph_steps"],
                eager_steps=eng.runner.stats["eager_steps"], selftest=eng.runner.selftest, pipelined=bool(eng.async_scheduling),
                overlap_calls=RowParallelLinear.overlap_calls - warm, overlap_calls_at_warmup=warm,
                comm=dict(ctx.comm) if ctx is not None else {}, device_backend=ctx.device_backend if ctx is not None else "none")
    eng.shutdown()
    if use_rccl:
        dist.destroy_process_group()
    print("RESULT " + json.dumps(info))


if __name__ == "__main__":
    main()
