"""All-reduce micro-benchmark: custom one-shot / two-shot over IPC-mapped peer buffers vs RCCL,
8 KB → 32 MB, timed inside hipGraphs (custom, RCCL) and eagerly (RCCL), world = the visible GPUs.

    python benchmarks/bench_allreduce.py                 # one rank per visible GPU (≥ 2)
    python benchmarks/bench_allreduce.py --world 2       # on a one-GPU box the ranks share the device: the IPC
                                                         # protocol is the same, the numbers are NOT xGMI numbers
    python benchmarks/bench_allreduce.py --full --out profiles/allreduce.json

Starts one throw-away process per rank (``dli_amd.parallel.comm_probe``: also the first-contact check that
``bench.py`` runs before it builds a multi-GPU engine), so a hang or crash in a collective library cannot take
the caller down.  The parent never touches a GPU.  Prints a table and the per-size winner the engine's start-up
sweep would install (workgroups, one-/two-shot switch, custom→RCCL crossover)."""
import argparse
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=0, help="ranks (default: visible GPUs, at least 2)")
    ap.add_argument("--full", action="store_true", help="every power of two from 8 KB to 32 MB")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--timeout", type=float, default=240.0)
    ap.add_argument("--out", default="")
    a = ap.parse_args(argv)
    import torch
    from dli_amd.parallel.comm_probe import probe_in_subprocess
    from dli_amd.runtime.launch import free_port
    n_dev = torch.cuda.device_count()                        # no GPU initialisation in this process
    if n_dev < 1:
        raise SystemExit("no GPU visible")
    world = a.world or max(n_dev, 2)
    share = n_dev < world
    port = free_port()
    with ThreadPoolExecutor(world) as pool:
        res = list(pool.map(lambda r: probe_in_subprocess(r, world, r % n_dev, port, share_gpu=share, seconds=5.0,
                                                          timeout=a.timeout, full=a.full, iters=a.iters), range(world)))
    r0 = res[0]
    r0["ranks_share_gpu"] = share
    r0["all_ranks_ok"] = {"rccl": all(r.get("rccl_ok") for r in res), "custom": all(r.get("custom_ar_ok") for r in res)}
    errs = {i: {k: v for k, v in r.items() if k.endswith("error")} for i, r in enumerate(res)}
    r0["errors_by_rank"] = {k: v for k, v in errs.items() if v}
    print(f"world {world}  devices {n_dev}  ranks share a GPU: {share}  rccl_ok {r0['all_ranks_ok']['rccl']}  "
          f"custom_ok {r0['all_ranks_ok']['custom']}")
    print(f"{'bytes':>10} {'one-shot µs':>12} {'(WGs)':>6} {'two-shot µs':>12} {'(WGs)':>6} {'RCCL eager':>11} {'RCCL graph':>11}")
    f = lambda v: "-" if v is None else f"{v:.1f}"
    for row in r0.get("all_reduce_table", []):
        print(f"{row['bytes']:>10} {f(row['one_shot_us']):>12} {str(row['one_shot_blocks'] or '-'):>6} "
              f"{f(row['two_shot_us']):>12} {str(row['two_shot_blocks'] or '-'):>6} {f(row.get('rccl_eager_us')):>11} "
              f"{f(row.get('rccl_graph_us')):>11}")
    print("custom → library crossover:", r0.get("custom_vs_library_crossover_bytes"), "bytes; one-shot up to",
          r0.get("one_shot_up_to_bytes"), "bytes")
    if r0["errors_by_rank"]:
        print("errors:", json.dumps(r0["errors_by_rank"])[:2000])
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(r0, fh, indent=1)
    return 0 if r0["all_ranks_ok"]["custom"] else 1


if __name__ == "__main__":
    sys.exit(main())
