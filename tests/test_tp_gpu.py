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


def run(model, tp, ep, prompts, extra_env=None):
    share = torch.cuda.device_count() < tp
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
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


@pytest.mark.parametrize("model,ep", [("tiny-llama-tp8", 1), ("tiny-mixtral-tp8", 8)])
def test_tp8_matches_tp1_on_gpu(model, ep):
    """The degree BASELINE configs 4 and 5 run at: 8 ranks (sharing the device on a one-GPU box), the 8-peer custom
    all-reduce / all-gather / broadcast kernels inside the decode hipGraphs, one expert per rank."""
    rng = np.random.default_rng(2)
    prompts = [rng.integers(3, 1000, size=n).tolist() for n in (40, 300, 7)]
    one = run(model, 1, 1, prompts)
    eight = run(model, 8, ep, prompts)
    assert eight["graph_steps"] > 0 and eight["custom_ar"]
    agree = np.mean([np.mean(np.array(a) == np.array(b)) for a, b in zip(one["greedy"], eight["greedy"])])
    assert all(a[:2] == b[:2] for a, b in zip(one["greedy"], eight["greedy"])) and agree >= 0.8, (one, eight)


@pytest.mark.parametrize("model,ep", [("llama3-8b", 1), ("mixtral-8x7b", 2), ("mixtral-8x7b", 1)])
def test_tp2_real_shapes_logits_track_tp1(model, ep, tmp_path):
    """Full-width layers (H 4096, 32/8 heads, FFN 14336, 8 experts, real vocabularies), 4 of them to keep it
    short: the first sampled row's logits of TP=2 must match TP=1 up to bf16 reduction noise."""
    rng = np.random.default_rng(1)
    prompts = [rng.integers(3, 1000, size=n).tolist() for n in (15, 36)]
    logits = {}
    for tp, e in ((1, 1), (2, ep)):
        f = str(tmp_path / f"logits_tp{tp}.pt")
        out = run(model, tp, e, prompts, {"DLI_TEST_LAYERS": "4", "DLI_DUMP_LOGITS": f})
        assert out["graph_steps"] > 0 and out["custom_ar"] == (tp > 1)
        logits[tp] = torch.load(f, weights_only=True)
    cos = torch.nn.functional.cosine_similarity(logits[1], logits[2], dim=0).item()
    assert cos > 0.9995, cos
