"""bench.py as the driver starts it: `python bench.py --gpus N` WITHOUT a torchrun environment must start its N
ranks itself (parent never touches a GPU), run DP replicas or TP groups (gloo on CPU here), print ONE JSON line
from rank 0, and exit non-zero when a rank fails.  Tiny models, seconds to run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--device", "cpu", "--steps", "25", "--warmup", "2", "--ramp-seconds", "2", "--ramp-cap-seconds", "6", "--sustained-seconds", "2",
          "--qps", "8", "--max-num-seqs", "16", "--max-batched-tokens", "256", "--num-kv-blocks", "512"]


def run_bench(extra, launcher=None, timeout=300, common=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + extra + (COMMON if common is None else common)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_self_launch_dp2():
    p, lines = run_bench(["--gpus", "2", "--model", "tiny-llama"])
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    assert [l for l in p.stdout.splitlines() if l.strip()] == lines, "nothing but that line on stdout (library chatter goes to stderr)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["self_launched"] is True
    assert d["vs_baseline"] is None and d["window_too_short"] is True and "value_sustained" in d
    assert d["backend"]["control"] == "gloo" and d["device_ids"] == [-1, -1]
    assert d["sustained_n"] > 0 and d["steps"] == 25         # (the 25-step window itself is ≈ 0.2 s on CPU)
    assert d["value"] == d["value_sustained"] and "value_window" in d and isinstance(d["steady"], bool)


def test_self_launch_tp2_reports_collectives():
    p, lines = run_bench(["--gpus", "2", "--tp", "2", "--model", "tiny-llama"])
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["config"]["parallelism"] == "tp2" and d["backend"]["device"] == "gloo"
    # 2 layers × 2 row-parallel all-reduces + the embedding: 5 per step, with their bytes
    assert d["all_reduce_calls_per_step"] == pytest.approx(5.0) and d["all_reduce_bytes_per_step"] > 0
    assert d["custom_ar_enabled"] is False and d["rccl_ranks_joined"] == 0


def test_torchrun_dp2_x_tp2_ep2():
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                "127.0.0.1", "--master-port", "29641"]
    p, lines = run_bench(["--gpus", "4", "--tp", "2", "--ep", "2", "--model", "tiny-mixtral", "--ep-a2a-min-tokens", "64"],
                         launcher=launcher)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["config"]["parallelism"] == "dp2xtp2+ep2" and d["self_launched"] is False
    assert d["comm_per_step"]["a2a_calls"] > 0          # large steps dispatched by all-to-all


def test_failing_rank_fails_the_job():
    p, lines = run_bench(["--gpus", "2", "--model", "no-such-model"], timeout=120)
    assert p.returncode != 0 and not lines


def test_gpus_must_match_world():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29643")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "cpu"], capture_output=True,
                       text=True, timeout=60, env=env, cwd=ROOT)
    assert p.returncode != 0 and "one rank per GPU" in p.stderr


def test_sweep_reports_frontier_with_steady_state_check():
    """`--sweep`: one JSON line with, per SLO row, the swept points (attainment over the window's arrivals, arrivals and
    completions per second, population at both ends of the window) and the frontier = largest STEADY point ≥ 90 %."""
    p, lines = run_bench(["--model", "tiny-llama", "--sweep", "--sweep-qps", "2,4", "--sweep-window", "3", "--sweep-ramp", "1",
                          "--sweep-rows", "relaxed"])
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["metric"].startswith("max_offered_qps_with_90pct_slo_attainment") and list(d["sweep"]) == ["relaxed"]
    pts = d["sweep"]["relaxed"]
    assert [t["offered_qps"] for t in pts] == [2.0, 4.0]
    for t in pts:
        assert {"arrivals", "good", "attainment", "arrivals_per_s", "finished_per_s", "in_system_at_window_start",
                "in_system_at_window_end", "steady", "shed", "still_running_at_end"} <= set(t)
        assert t["good"] <= t["arrivals"]
    steady_ok = [t for t in pts if t["steady"] and t["attainment"] >= 0.9]
    assert d["value"] == (max(t["offered_qps"] for t in steady_ok) if steady_ok else 0.0)


@pytest.mark.parametrize("extra,par", [
    (["--gpus", "8", "--model", "tiny-llama"], "dp8"),
    (["--gpus", "8", "--tp", "8", "--model", "tiny-llama-8kv"], "tp8"),
    (["--gpus", "8", "--tp", "8", "--ep", "8", "--model", "tiny-mixtral-8e", "--ep-a2a-min-tokens", "64"], "tp8+ep8"),
])
def test_world_size_8_layouts(extra, par):
    """The three 8-rank layouts the driver runs on the 8-GPU node (BASELINE configs 3, 4, 5), on CPU with gloo: rank
    grouping, driver election, one expert per rank, all-to-all over 8 ranks, one JSON line."""
    # 8 ranks on the CI box's 8 CPUs: a 2 s stretch can pass without one finished request when the box is busy
    common = [a if a != "2" or COMMON[i - 1] != "--sustained-seconds" else "8" for i, a in enumerate(COMMON)]
    p, lines = run_bench(extra, timeout=600, common=common)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == par and d["sustained_n"] > 0
    assert len(d["device_ids"]) == 8
    if "ep8" in par:
        assert d["comm_per_step"]["a2a_calls"] > 0      # prefill-sized steps dispatched by all-to-all over 8 ranks


def test_layout_rank_grouping(monkeypatch):
    """Pure rank arithmetic of the launcher: consecutive ranks form a TP group, tp_rank 0 drives it."""
    from dli_amd.runtime.launch import Layout
    for rank in range(8):
        monkeypatch.setenv("RANK", str(rank))
        monkeypatch.setenv("WORLD_SIZE", "8")
        monkeypatch.setenv("LOCAL_RANK", str(rank))
        dp = Layout.from_env(8)
        assert (dp.dp, dp.replica, dp.tp_rank, dp.is_driver) == (8, rank, 0, True) and dp.driver_ranks == list(range(8))
        l = Layout.from_env(8, tp=4, ep=4)
        assert (l.dp, l.replica, l.tp_rank, l.is_driver) == (2, rank // 4, rank % 4, rank % 4 == 0)
        assert l.groups == [[0, 1, 2, 3], [4, 5, 6, 7]] and l.driver_ranks == [0, 4]
        t = Layout.from_env(8, tp=8)
        assert t.dp == 1 and t.groups == [list(range(8))] and t.driver_ranks == [0]
    for bad in (dict(gpus=4), dict(gpus=8, tp=3), dict(gpus=8, tp=4, ep=2)):
        with pytest.raises(SystemExit):
            Layout.from_env(**bad)


def test_steady_detector_ramp_criterion():
    """bench.py's ramp criterion: steady once the in-system population grew by < 10 % of the
    arrivals over the last 10 s; never before a full horizon has been seen."""
    sys.path.insert(0, ROOT)
    import bench
    d = bench.SteadyDetector(horizon=10.0, frac=0.10)
    d.add(0.0, 0, 0)
    for t in range(1, 10):
        d.add(float(t), 60 * t, 68 * t)                      # almost nothing finishes yet: population tracks arrivals
        assert not d.steady()
    d.add(10.0, 600, 680)
    assert d.growth() == (600, 680) and not d.steady()
    for t in range(11, 40):                                  # completions catch up: growth decays
        d.add(float(t), 600 + int(400 * (1 - 0.85 ** (t - 10))), 68 * t)
    assert d.steady()
    g, a = d.growth()
    assert a == 680 and g < 68
    e = bench.SteadyDetector(horizon=10.0)                   # overload: the backlog keeps growing by 30 % of the arrivals
    for t in range(0, 60):
        e.add(float(t), 20 * t, 68 * t)
    assert not e.steady()


def test_tp_replica_without_qps_probes_its_capacity_and_reports_steady_fields():
    common = [x for i, x in enumerate(COMMON) if x != "--qps" and COMMON[i - 1] != "--qps"]
    p, lines = run_bench(["--gpus", "2", "--tp", "2", "--model", "tiny-llama", "--capacity-probe-seconds", "2"], common=common)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(lines[0])
    cp = d["capacity_probe"]
    assert cp["est_req_per_s"] > 0 and d["config"]["offered_qps"] == pytest.approx(0.9 * cp["est_req_per_s_min_over_replicas"], rel=1e-2)
    assert isinstance(d["steady"], bool) and d["ramp_s"] >= 2.0 and "value_window" in d
    assert d["value"] == d["value_sustained"] and d["value_source"].startswith("value_sustained")
    assert d["config"]["max_num_batched_tokens"] == 256


def test_emulated_tp_rank_runs_alone_with_the_shard_shapes():
    """`bench.py --emulate-tp N`: one process, TP rank 0's shards of an N-wide replica, collectives
    replaced by same-bytes copies — counted like real ones — and the line is marked as not a throughput figure."""
    common = [x for i, x in enumerate(COMMON) if x != "--qps" and COMMON[i - 1] != "--qps"]
    p, lines = run_bench(["--model", "tiny-llama-8kv", "--emulate-tp", "8", "--capacity-probe-seconds", "2"], common=common)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "INVALID_emulation" in d and d["config"]["parallelism"].startswith("tp8 rank 0 emulated")
    assert d["all_reduce_calls_per_step"] == pytest.approx(5.0) and d["comm_per_step"]["ag_calls"] > 0
    assert d["capacity_probe"]["est_req_per_s"] > 0 and set(d["host_ms_per_step"]) == {"sched", "launch", "post"}


def test_emulated_rank_holds_exactly_rank_zero_shards():
    import torch
    from dli_amd.config import get_model_config
    from dli_amd.models import build_model
    from dli_amd.parallel.state import ParallelContext
    cfg = get_model_config("tiny-mixtral-8e")
    full = build_model(cfg, device="cpu", dtype=torch.float32, seed=3)
    emu = build_model(cfg, ParallelContext(tp_size=8, tp_rank=0, ep_size=8, emulated=True), "cpu", torch.float32, 3)
    fw = {n: w for n, w, _ in full.named_weights()}
    for name, w, _ in emu.named_weights():
        assert torch.equal(w.data, w.shard_of(fw[name].data)), name
    l0 = emu.layers[0]
    assert l0.attn.hq == 2 and l0.attn.hkv == 1 and l0.mlp.w13.data.shape[0] == 1 and l0.mlp.local_experts == 1
    x = torch.randn(4, cfg.hidden_size)
    assert emu.ctx.all_reduce(x) is not x and torch.equal(emu.ctx.all_reduce(x), x)
    assert emu.ctx.all_gather_last_dim(torch.ones(3, 5)).shape == (3, 40) and not emu.ctx.distributed


def test_probe_child_environment_drops_the_launcher_rendezvous():
    """Under `torch.distributed.run` every worker carries TORCHELASTIC_USE_AGENT_STORE: with it, a tcp:// rendezvous makes
    ALL ranks clients of a store nobody hosts.  The comm probe's children build their own rendezvous, so their environment
    must not carry it (the GPU rehearsal of the driver's 2-rank invocation lost 150 s in that connect and declared RCCL
    and the custom collectives unusable).  Two children with the scrubbed environment rendezvous and all-reduce."""
    from dli_amd.parallel.comm_probe import child_env
    from dli_amd.runtime.launch import free_port
    dirty = dict(os.environ, TORCHELASTIC_USE_AGENT_STORE="True", TORCHELASTIC_RESTART_COUNT="0", TORCHELASTIC_RUN_ID="x",
                 RANK="5", WORLD_SIZE="9", LOCAL_RANK="5", GROUP_RANK="0", MASTER_PORT="1", LOCAL_WORLD_SIZE="9")
    env = child_env(dirty)
    assert not [k for k in env if k.startswith("TORCHELASTIC_")]
    assert not {"RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE"} & set(env)
    assert env["PATH"] == os.environ["PATH"]
    port = free_port()
    code = ("import sys, datetime, torch, torch.distributed as dist\n"
            "r = int(sys.argv[1])\n"
            f"dist.init_process_group('gloo', init_method='tcp://127.0.0.1:{port}', rank=r, world_size=2, "
            "timeout=datetime.timedelta(seconds=60))\n"
            "x = torch.ones(4) * (r + 1)\n"
            "dist.all_reduce(x)\n"
            "assert float(x[0]) == 3.0\n"
            "dist.destroy_process_group()\n")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out[-2000:]
