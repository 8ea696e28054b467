"""Helper for test_tp_gpu.py: TP=k generation on GPU(s).  With DLI_TP_SHARE_GPU=1 all ranks share one
device and talk over gloo (the custom all-reduce still uses HIP IPC between the processes)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dli_amd.config import (CacheConfig, EngineConfig, ParallelConfig, SamplingParams, SchedulerConfig,
                            get_model_config)
from dli_amd.engine.engine import LLMEngine
from dli_amd.runtime.tp_launcher import join_children, maybe_launch_tp


def main():
    model, tp, ep, backend = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    prompts = json.loads(sys.argv[5])
    over = {"num_layers": int(os.environ["DLI_TEST_LAYERS"])} if os.environ.get("DLI_TEST_LAYERS") else {}
    cfg = EngineConfig(model=get_model_config(model, **over), cache=CacheConfig(num_blocks=512),
                       sched=SchedulerConfig(max_num_seqs=8, max_num_batched_tokens=2048, max_model_len=2048,
                                             long_prefill_chunk=2048),
                       parallel=ParallelConfig(tp=tp, ep=ep, backend=backend,
                                               ep_a2a_min_tokens=int(os.environ.get("DLI_TEST_EP_A2A", "2048")),
                                               custom_all_reduce=os.environ.get("DLI_NO_CUSTOM_AR") != "1"), device="cuda",
                       graph_batch_sizes=(4, 8), quantization=os.environ.get("DLI_TEST_QUANT") or None,
                       async_scheduling={"0": False, "1": True}.get(os.environ.get("DLI_TEST_ASYNC", ""), None))
    ctx = maybe_launch_tp(cfg)
    eng = LLMEngine(cfg, ctx)
    eng.warmup()
    from dli_amd.models.layers import RowParallelLinear
    warm_calls = RowParallelLinear.overlap_calls             # the start-up self-test and the graph captures
    out = eng.generate(None, SamplingParams(temperature=0.0, max_tokens=10, ignore_eos=True), prompt_ids=prompts)
    from dli_amd.runtime.tp_launcher import job_info
    info = {"greedy": out, "graph_steps": eng.runner.stats["graph_steps"],
            "mixed_graph_steps": eng.runner.stats["mixed_graph_steps"], "eager_steps": eng.runner.stats["eager_steps"],
            "custom_ar": bool(ctx is not None and ctx.custom_ar is not None),
            "overlap_calls": RowParallelLinear.overlap_calls - warm_calls, "overlap_calls_at_warmup": warm_calls, "selftest": eng.runner.selftest,
            "a2a_calls": ctx.comm["a2a_calls"] if ctx is not None else 0,
            "sp_norm_calls": ctx.comm["sp_norm_calls"] if ctx is not None else 0,
            "device_backend": job_info().get("device_backend"), "tuning_source": (job_info().get("tp_group_tuning") or {}).get("source")}
    eng.shutdown()
    join_children()
    print("RESULT " + json.dumps(info))


if __name__ == "__main__":
    main()
