"""First-contact probe + micro-benchmark of the device collectives, run in a *throw-away process* per rank.

Why a separate process: the first time more than one GPU is visible, RCCL initialisation, HIP IPC mapping of
peer memory and the custom xGMI kernels all execute for the first time.  An exception there is easy to survive, a
hang or a crash inside a library is not — so every rank of a job starts this module as a child
(``probe_in_subprocess``), bounded by a timeout, and only uses in its own process what the probe has proven to
work.  The same run produces the all-reduce latency table (custom one-shot / two-shot at the tuned workgroup
counts, RCCL eager, RCCL inside a hipGraph; 8 KB … 32 MB) that ``bench.py`` embeds in its JSON line when
``gpus > 1`` and that ``benchmarks/bench_allreduce.py`` prints stand-alone (SURVEY.md §5.8 sizes the whole TP
design on these numbers; the reference has no collective communication at all).

    python -m dli_amd.parallel.comm_probe --rank R --world N --device D --port P --out result.json
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time


def _probe_main(a) -> dict:
    import datetime

    import torch
    import torch.distributed as dist

    res: dict = {"world": a.world, "rank": a.rank, "device": a.device, "rccl_ok": False, "custom_ar_ok": False}
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", a.device)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{a.port}", rank=a.rank, world_size=a.world,
                            timeout=datetime.timedelta(seconds=a.timeout))
    from dli_amd.parallel.state import ParallelContext, drain_collectives

    def vote(ok: bool) -> bool:
        votes = [None] * a.world
        dist.all_gather_object(votes, bool(ok))
        return all(votes)

    # ---- RCCL: group creation + a checked all-reduce
    nccl_group = None
    shared = a.share_gpu or torch.cuda.device_count() < a.world
    if shared:
        res["rccl_skipped"] = "ranks share a GPU (RCCL needs one device per rank)"
    else:
        ok = True
        try:
            nccl_group = dist.new_group(list(range(a.world)), backend="nccl",
                                        timeout=datetime.timedelta(seconds=a.timeout))
            x = torch.full((1024,), float(a.rank + 1), device=dev, dtype=torch.bfloat16)
            dist.all_reduce(x, group=nccl_group)
            torch.cuda.synchronize()
            ok = bool((x.float() == a.world * (a.world + 1) / 2).all())
            if not ok:
                res["rccl_error"] = "all-reduce returned a wrong sum"
        except Exception as e:
            ok = False
            res["rccl_error"] = f"{type(e).__name__}: {e}"
        res["rccl_ok"] = vote(ok)
        if not res["rccl_ok"]:
            nccl_group = None
    res["rccl_ranks_joined"] = a.world if res["rccl_ok"] else 0

    # ---- custom IPC collectives (+ launch-shape sweep = the latency table)
    ctx = ParallelContext(tp_size=a.world, tp_rank=a.rank, group=nccl_group if nccl_group is not None else dist.group.WORLD,
                          cpu_group=dist.group.WORLD)
    table = []
    try:
        from dli_amd.parallel.custom_all_reduce import CustomAllReduce
        ar = CustomAllReduce(ctx, dev, max_bytes=a.max_bytes)
        ok = ar.self_test()
        res["custom_ar_ok"] = ok
        if ok and a.seconds > 0:
            sizes = [s for s in (8 << 10, 32 << 10, 128 << 10, 512 << 10, 2 << 20, 8 << 20, 32 << 20, 64 << 20) if s <= a.max_bytes]
            if a.full:
                sizes = [8 << 10 << i for i in range(13) if (8 << 10 << i) <= a.max_bytes]
            tun = ar.tune(sizes=sizes, iters=a.iters, with_library=nccl_group is not None)
            for row in tun["rows"]:
                one = [c for c in row["candidates"] if c["algo"] == 1]
                two = [c for c in row["candidates"] if c["algo"] == 2]
                table.append({"bytes": row["bytes"],
                              "one_shot_us": min((c["us"] for c in one), default=None),
                              "one_shot_blocks": min(one, key=lambda c: c["us"])["blocks"] if one else None,
                              "two_shot_us": min((c["us"] for c in two), default=None),
                              "two_shot_blocks": min(two, key=lambda c: c["us"])["blocks"] if two else None,
                              "rccl_eager_us": row.get("library_us")})
            res["tuning_rows"] = tun["rows"]                 # the engine installs this table instead of sweeping again
            res["custom_vs_library_crossover_bytes"] = tun["use_custom_up_to_bytes"]
            res["one_shot_up_to_bytes"] = tun["one_shot_up_to_bytes"]
            res["sweep_seconds"] = tun["sweep_seconds"]
    except Exception as e:
        res["custom_ar_ok"] = False
        res["custom_ar_error"] = f"{type(e).__name__}: {e}"
    res["custom_ar_ok"] = vote(res["custom_ar_ok"])

    # ---- RCCL inside a hipGraph (what a captured TP decode step without the custom kernels would replay)
    if nccl_group is not None and a.seconds > 0:
        ok = True
        try:
            for row in table:
                x = torch.ones(row["bytes"] // 2, dtype=torch.bfloat16, device=dev)
                dist.all_reduce(x, group=nccl_group)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream(device=dev)
                s.wait_stream(torch.cuda.current_stream())
                drain_collectives()
                with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                    for _ in range(a.iters):
                        dist.all_reduce(x, group=nccl_group)
                torch.cuda.synchronize()
                dist.barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g.replay()
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                row["rccl_graph_us"] = round(e0.elapsed_time(e1) * 1e3 / a.iters, 2)
        except Exception as e:
            ok = False
            res["rccl_graph_error"] = f"{type(e).__name__}: {e}"
        res["rccl_graph_ok"] = vote(ok)
    res["all_reduce_table"] = table
    dist.barrier()
    dist.destroy_process_group()
    return res


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--device", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--seconds", type=float, default=5.0, help="0: connectivity checks only, no latency table")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--full", action="store_true", help="every power of two from 8 KB to max-bytes")
    ap.add_argument("--max-bytes", type=int, default=64 << 20)
    ap.add_argument("--timeout", type=float, default=120.0)
    a = ap.parse_args(argv)
    res = _probe_main(a)
    with open(a.out + ".tmp", "w") as fh:
        json.dump(res, fh)
    os.replace(a.out + ".tmp", a.out)
    return 0


def child_env(env: dict) -> dict:
    """Environment of the probe child: the probe builds its OWN rendezvous (tcp://127.0.0.1:port, rank 0 hosts the store), so
    nothing of the launcher's may leak in.  Under ``torch.distributed.run`` the workers carry ``TORCHELASTIC_USE_AGENT_STORE``:
    with it set, ``init_process_group`` makes EVERY rank a client of a store it expects the agent to host at that address —
    nobody listens on the probe's port and the child sits in the connect until the probe's time-out (found with the driver's
    invocation, 2 ranks: 150 s lost and RCCL + the custom collectives declared unusable)."""
    drop = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "LOCAL_WORLD_SIZE",
            "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE")
    return {k: v for k, v in env.items() if k not in drop and not k.startswith("TORCHELASTIC_")}


def probe_in_subprocess(rank: int, world: int, device: int, port: int, share_gpu: bool = False, seconds: float = 5.0,
                        timeout: float = 150.0, full: bool = False, iters: int = 10, max_bytes: int = 64 << 20) -> dict:
    """Run the probe for this rank in a child process; never raises.  On a crash, a time-out or a missing result the
    returned dict has ``rccl_ok = custom_ar_ok = False`` and says why."""
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = os.path.join(tempfile.gettempdir(), f"dli_comm_probe_{os.getpid()}_{rank}.json")
    cmd = [sys.executable, "-m", "dli_amd.parallel.comm_probe", "--rank", str(rank), "--world", str(world), "--device",
           str(device), "--port", str(port), "--out", out, "--seconds", str(seconds), "--iters", str(iters), "--max-bytes",
           str(max_bytes), "--timeout", str(max(timeout - 20.0, 30.0))]
    if share_gpu:
        cmd.append("--share-gpu")
    if full:
        cmd.append("--full")
    env = child_env(dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", "")))
    fail = {"world": world, "rank": rank, "device": device, "rccl_ok": False, "custom_ar_ok": False, "rccl_ranks_joined": 0,
            "all_reduce_table": []}
    t0 = time.perf_counter()
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    except OSError as e:
        return dict(fail, probe_error=f"could not start the probe: {e}")
    try:
        log, _ = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, 9)                              # the exact process group started above
        except OSError:
            pass
        p.wait()
        return dict(fail, probe_error=f"probe timed out after {timeout:.0f} s (hung collective initialisation?)")
    if p.returncode != 0 or not os.path.exists(out):
        return dict(fail, probe_error=f"probe exited with code {p.returncode}: {(log or '')[-1500:]}")
    try:
        with open(out) as fh:
            res = json.load(fh)
        os.remove(out)
    except (OSError, ValueError) as e:
        return dict(fail, probe_error=f"unreadable probe result: {e}")
    res["probe_seconds"] = round(time.perf_counter() - t0, 1)
    return res


if __name__ == "__main__":
    sys.exit(main())
