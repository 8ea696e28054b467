"""Helper for test_custom_ar_gpu.py: one rank of a custom all-reduce group.  All ranks may share one
physical GPU (the IPC mapping path is the same as across GPUs); bootstrap is gloo over TCP."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.distributed as dist


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from dli_amd.parallel.custom_all_reduce import CustomAllReduce
    from dli_amd.parallel.state import ParallelContext
    ctx = ParallelContext(tp_size=world, tp_rank=rank, group=dist.group.WORLD, cpu_group=dist.group.WORLD)
    ar = CustomAllReduce(ctx, dev, max_bytes=4 << 20)
    results = {}
    g = torch.Generator().manual_seed(100 + rank)
    for numel, algo in ((8, 1), (4096, 1), (4096 * 33 + 8, 2), (1 << 20, 2), (1 << 20, 1), (2 << 20, 0)):
        x_cpu = torch.randn(numel, generator=g).to(torch.bfloat16)
        parts = [None] * world
        dist.all_gather_object(parts, x_cpu)
        want = torch.stack([p.float() for p in parts]).sum(0)
        x = x_cpu.to(dev)
        outs = [ar.all_reduce(x, algo) for _ in range(3)]          # back-to-back calls: epoch/flag reuse
        torch.cuda.synchronize()
        err = max(float((o.float().cpu() - want).abs().max()) for o in outs)
        results[f"{numel}/{algo}"] = err
    # skew stress: one rank arrives late (extra GPU work + host sleep), fresh data every call, the result is
    # checked against the exact sum; this is what catches stale reads of a peer's staging buffer
    import time
    bad = 0
    big = torch.randn(2048, 2048, device=dev)
    for it in range(120):
        numel = (1247 * 512, 4096 * 24, 640 * 512)[it % 3]
        base = torch.full((numel,), float(it % 7 + 1), device=dev, dtype=torch.bfloat16)
        x = base * (rank + 1)
        if it % world == rank:                       # the late rank of this round
            for _ in range(3):
                big = (big @ big).clamp_(-1, 1)
            if it % 5 == 0:
                time.sleep(0.003)
        y = ar.all_reduce(x, 1 if it % 2 else 0)
        want = float(it % 7 + 1) * world * (world + 1) / 2
        bad += int((y.float() != want).sum().item() > 0)
        g2 = ar.all_gather(x[:4096])
        bad += int(any(float(g2[r][0]) != float(it % 7 + 1) * (r + 1) for r in range(world)))
    results["skew_stress_bad"] = float(bad)
    # graph replay: the epoch counter must advance on its own
    x = torch.ones(4096, device=dev, dtype=torch.bfloat16) * (rank + 1)
    out = torch.empty_like(x)
    ar.lib.ar_all_reduce(ar.state, x, out, 1)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        ar.lib.ar_all_reduce(ar.state, x, out, 1)
    for _ in range(4):
        gr.replay()
    torch.cuda.synchronize()
    results["graph"] = float((out.float() - world * (world + 1) / 2).abs().max())
    # broadcast of an int32 buffer from rank 0 (the packed step buffer of a TP step)
    b = (torch.arange(4096 * 3, dtype=torch.int32, device=dev) * (1 if rank == 0 else -1)).contiguous()
    ar.broadcast(b, 0)
    results["broadcast"] = float((b != torch.arange(4096 * 3, dtype=torch.int32, device=dev)).sum())
    # launch-shape sweep (tiny): every rank must end up with the same table and still reduce correctly
    tun = ar.tune(sizes=(8 << 10, 128 << 10, 1 << 20), iters=3, with_library=False)
    assert len(tun["rows"]) == 3 and all("best" in r for r in tun["rows"])
    x = torch.ones(1 << 19, device=dev, dtype=torch.bfloat16) * (rank + 1)
    results["after_tune"] = float((ar.all_reduce(x).float() - world * (world + 1) / 2).abs().max())
    ar.queue_error_check()
    torch.cuda.synchronize()
    results["error_flag"] = ar.error() or ar.error_seen
    dist.barrier()
    print("RESULT " + json.dumps(results), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
