"""Helper run as a subprocess by test_tp.py: generate greedily with a TP=k tiny model (gloo, CPU)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dli_amd.config import (CacheConfig, EngineConfig, ParallelConfig, SamplingParams, SchedulerConfig,
                            get_model_config)
from dli_amd.engine.engine import LLMEngine
from dli_amd.runtime.tp_launcher import join_children, maybe_launch_tp


def main():
    model, tp, ep = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    prompts = json.loads(sys.argv[4])
    cfg = EngineConfig(model=get_model_config(model), cache=CacheConfig(num_blocks=64),
                       sched=SchedulerConfig(max_num_seqs=8, max_num_batched_tokens=64, max_model_len=256,
                                             long_prefill_chunk=32),
                       parallel=ParallelConfig(tp=tp, ep=ep, ep_a2a_min_tokens=int(os.environ.get("DLI_TEST_EP_A2A", "0"))),
                       device="cpu", use_hip_graphs=False,
                       async_scheduling=os.environ.get("DLI_TEST_ASYNC") == "1")
    ctx = maybe_launch_tp(cfg)
    eng = LLMEngine(cfg, ctx)
    out = eng.generate(None, SamplingParams(temperature=0.0, max_tokens=10, ignore_eos=True), prompt_ids=prompts)
    sampled = eng.generate(None, SamplingParams(temperature=0.9, top_p=0.9, max_tokens=6, seed=5, ignore_eos=True),
                           prompt_ids=prompts[:2])
    comm = dict(ctx.comm) if ctx is not None else {}
    eng.shutdown()
    join_children()
    print("RESULT " + json.dumps({"greedy": out, "sampled": sampled, "a2a_calls": comm.get("a2a_calls", 0)}))


if __name__ == "__main__":
    main()
