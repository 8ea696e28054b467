"""Helper for test_tp.py: two TP ranks whose KV caches come out at DIFFERENT sizes (as happens when every rank sizes
its cache from its own free memory); the group must settle on the minimum before the block manager is built."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dli_amd.config import CacheConfig, EngineConfig, ParallelConfig, SamplingParams, SchedulerConfig, get_model_config
from dli_amd.engine.engine import LLMEngine
from dli_amd.runtime import launch
from dli_amd.runtime.tp_launcher import worker_loop


def main():
    if not launch.launched_by_torchrun():
        sys.exit(launch.self_launch(2, os.path.abspath(__file__), sys.argv[1:]))
    layout = launch.Layout.from_env(2, tp=2)
    job = launch.bootstrap(layout, device_kind="cpu")
    mine = 300 if layout.rank == 0 else 200
    cfg = EngineConfig(model=get_model_config("tiny-llama"), cache=CacheConfig(num_blocks=mine),
                       sched=SchedulerConfig(max_num_seqs=8, max_num_batched_tokens=64, max_model_len=256),
                       parallel=ParallelConfig(tp=2), device="cpu", use_hip_graphs=False)
    eng = LLMEngine(cfg, job.ctx, job.device)
    eng.warmup()
    if layout.is_driver:
        out = eng.generate(None, SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True), prompt_ids=[[5, 6, 7, 8]])
        print("RESULT " + json.dumps({"runner_blocks": eng.runner.num_blocks, "bm_blocks": eng.bm.num_blocks, "n": len(out[0])}),
              flush=True)
        eng.shutdown()
    else:
        worker_loop(eng)
        assert eng.runner.num_blocks == 200
    launch.shutdown(job)


if __name__ == "__main__":
    main()
