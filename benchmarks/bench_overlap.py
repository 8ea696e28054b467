"""A/B of the GEMM / all-reduce pipeline of ``RowParallelLinear``: plain GEMM → all-reduce against
the chunked pipeline with the reduction on a side stream, for several chunk counts and caps on the all-reduce kernel's
workgroups (the co-resident kernel takes CUs from the GEMM), at the row-parallel shapes of the TP configs.

    python benchmarks/bench_overlap.py --ranks 2            # starts the ranks itself; DLI_SHARE_GPU=1 on a one-GPU box

On a box with one GPU the ranks share it (gloo + the custom IPC kernels): every rank's GEMMs and reduction kernels
compete for the same CUs, so absolute times are about ``ranks ×`` those of one GPU per rank — what the table is for
is the RELATIVE cost of the variants (does a co-resident all-reduce with N workgroups slow the GEMM down, does
chunking pay) and that the pipeline executes and agrees with the plain path.  One JSON line per case from rank 0.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {   # name: (K of the shard, N)
    "8b_o_tp2": (2048, 4096), "8b_down_tp2": (7168, 4096),
    "70b_o_tp8": (1024, 8192), "70b_down_tp8": (3584, 8192),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--rows", default="2048,4096")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    from dli_amd.runtime import launch
    if not launch.launched_by_torchrun():
        sys.exit(launch.self_launch(a.ranks, os.path.abspath(__file__), sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from dli_amd import ops
    from dli_amd.models.layers import RowParallelLinear
    layout = launch.Layout.from_env(a.ranks, a.ranks, 1)
    sys.stdout.flush()
    out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    job = launch.bootstrap(layout, share_gpu=launch.share_gpu_requested(), probe_seconds=2.0)
    ctx, dev = job.ctx, job.device
    cpu_group = ctx.cpu_group if ctx.cpu_group is not None else ctx.group

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        dist.barrier(group=cpu_group)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.iters

    for name, (K, N) in SHAPES.items():
        layer = RowParallelLinear(K * ctx.tp_size, N, ctx, torch.bfloat16, dev)
        layer.weight.data.normal_(0, 0.02)
        for T in [int(x) for x in a.rows.split(",")]:
            x = (torch.randn(T, K, device=dev) * 0.5).to(torch.bfloat16)
            gemm_us = timed(lambda: ops.linear(x, layer.weight.data))
            want = ctx.all_reduce(ops.linear(x, layer.weight.data)).float()
            plain_us = timed(lambda: ctx.all_reduce(ops.linear(x, layer.weight.data)))
            row = {"shape": name, "K_shard": K, "N": N, "rows": T, "ranks": ctx.tp_size,
                   "ranks_share_gpu": job.comm["ranks_share_gpu"], "device_backend": job.comm["device_backend"],
                   "custom_ar": job.comm["custom_ar_enabled"], "gemm_only_us": round(gemm_us, 1),
                   "gemm_then_all_reduce_us": round(plain_us, 1), "pipelined_us": {}}
            for chunks in (2, 4):
                for blocks in (32, 64, 128):
                    RowParallelLinear.OVERLAP_CHUNKS, RowParallelLinear.OVERLAP_AR_BLOCKS = chunks, blocks
                    got = layer._forward_overlapped(x, None).float()
                    torch.cuda.synchronize()
                    err = float((got - want).abs().max())
                    us = timed(lambda: layer._forward_overlapped(x, None))
                    row["pipelined_us"][f"chunks{chunks}_blocks{blocks}"] = round(us, 1)
                    row["max_abs_diff_vs_plain"] = max(row.get("max_abs_diff_vs_plain", 0.0), err)
            best = min(row["pipelined_us"], key=row["pipelined_us"].get)
            row["best"] = best
            row["best_vs_plain"] = round(row["pipelined_us"][best] / plain_us, 3)
            if layout.rank == 0:
                print(json.dumps(row), file=out, flush=True)
    launch.shutdown(job)


if __name__ == "__main__":
    main()
