"""Tensor / expert parallel equivalence on CPU (gloo, world_size 2 and 4): a TP=k model built from the
same seed must reproduce the TP=1 greedy tokens (SURVEY.md §4 items 4–5).  Runs the real driver/worker
protocol: rank 0 schedules and broadcasts the packed step buffer, workers execute ``worker_step``."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def run(model, tp, ep, prompts, pipelined=False):
    env = dict(os.environ, OMP_NUM_THREADS="2", DLI_TEST_ASYNC="1" if pipelined else "0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "tp_script.py"), model, str(tp), str(ep), json.dumps(prompts)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def prompts():
    rng = np.random.default_rng(0)
    return [rng.integers(3, 500, size=n).tolist() for n in (5, 70, 33)]


@pytest.mark.parametrize("model,tp,ep", [("tiny-llama", 2, 1), ("tiny-mixtral", 2, 1), ("tiny-mixtral", 2, 2),
                                         ("tiny-opt", 2, 1), ("tiny-llama", 4, 1)])
def test_tp_matches_single_rank(model, tp, ep):
    if model == "tiny-llama" and tp == 4:
        pytest.skip("tiny-llama has 2 KV heads") if False else None
    want = run(model, 1, 1, prompts())
    if model == "tiny-llama" and tp == 4:
        model_tp = "tiny-llama"
        with pytest.raises(AssertionError):
            run(model_tp, tp, ep, prompts())          # 2 KV heads cannot be split 4 ways: must fail loudly
        return
    got = run(model, tp, ep, prompts())
    assert got["greedy"] == want["greedy"]
    assert got["sampled"] == want["sampled"]


def test_tp_pipelined_stepping_matches_single_rank():
    """Driver queues step N+1 before step N's ids are known; the workers feed their own sampler output."""
    want = run("tiny-llama", 1, 1, prompts())
    got = run("tiny-llama", 2, 1, prompts(), pipelined=True)
    assert got == want
