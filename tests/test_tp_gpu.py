"""TP / EP on the GPU kernels: driver/worker protocol, hipGraph decode with collectives inside, the
custom IPC all-reduce and the GEMM/all-reduce overlap path (prompt ≥ 1024 tokens).  On a one-GPU box
the ranks share the device (DLI_TP_SHARE_GPU=1) and bootstrap over gloo."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run(model, tp, ep, prompts):
    share = torch.cuda.device_count() < tp
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if share:
        env["DLI_TP_SHARE_GPU"] = "1"
    backend = "gloo" if share else "nccl"
    p = subprocess.run([sys.executable, os.path.join(HERE, "tp_gpu_script.py"), model, str(tp), str(ep), backend,
                        json.dumps(prompts)], capture_output=True, text=True, timeout=280, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


@pytest.mark.parametrize("model,ep", [("tiny-llama-tp", 1), ("tiny-mixtral-tp", 2)])
def test_tp2_matches_tp1_on_gpu(model, ep):
    rng = np.random.default_rng(0)
    prompts = [rng.integers(3, 1000, size=n).tolist() for n in (40, 1200, 7)]
    one = run(model, 1, 1, prompts)
    two = run(model, 2, ep, prompts)
    assert two["graph_steps"] > 0
    agree = np.mean([np.mean(np.array(a) == np.array(b)) for a, b in zip(one["greedy"], two["greedy"])])
    assert all(a[:2] == b[:2] for a, b in zip(one["greedy"], two["greedy"])) and agree >= 0.8, (one, two)
