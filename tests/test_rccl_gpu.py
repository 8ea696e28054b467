"""RCCL on the one-GPU box: as much of the `nccl` path as one device can execute.  RCCL refuses two ranks on one
device ("Duplicate GPU detected"), so the group has ONE rank (`ParallelContext.one_rank_group`): the communicator is
real, every collective of the model / step protocol is a library call on the compute (or side) stream, and the decode
and mixed-step hipGraphs capture those calls — what is NOT covered is data moving between ranks (the driver's 8-GPU
runs; the multi-rank logic is covered with gloo on CPU and with the custom IPC kernels on the shared GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run(mode, prompts, env=None):
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_one_rank_script.py"), mode, json.dumps(prompts)],
                       capture_output=True, text=True, timeout=280, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {})))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_engine_over_a_one_rank_rccl_group_equals_the_plain_engine():
    rng = np.random.default_rng(11)
    prompts = [rng.integers(3, 1000, size=n).tolist() for n in (1300, 40, 7)]
    plain = run("plain", prompts)
    rccl = run("rccl", prompts, {"DLI_OVERLAP_FORCE": "1"})          # keep the pipeline on whatever its timing says
    assert rccl["device_backend"] == "nccl" and rccl["eager_collectives_identity"] and rccl["captured_all_reduce_replays"]
    # the model's reductions, the logits gather and the step header all went through the library …
    c = rccl["comm"]
    assert c["ar_lib_calls"] > 0 and c["ag_calls"] > 0 and c["bcast_calls"] > 0 and c["ar_custom_calls"] == 0, c
    # … inside the captured decode and mixed steps, with pipelined stepping on (nccl device group) …
    assert rccl["graph_steps"] > 0 and rccl["mixed_graph_steps"] > 0 and rccl["pipelined"] is True
    assert rccl["selftest"]["graph_vs_eager"]["ok"] and rccl["selftest"]["mixed_graph_vs_eager"]["ok"]
    # … and the GEMM / all-reduce pipeline ran its RCCL branch on the side stream (self-test + the 1300-token prompt)
    st = rccl["selftest"]["overlapped_row_parallel"]
    assert st["ok"] is True and st["reduced_by"].startswith("RCCL"), st
    assert rccl["overlap_calls"] + rccl["overlap_calls_at_warmup"] > 1
    # a sum over one rank is the identity: same tokens as the engine without any group
    assert rccl["greedy"] == plain["greedy"]
