"""Probe: does PyTorch TunableOp find faster GEMM solutions than the default hipBLASLt heuristic for the
Llama-3-8B layer shapes at serving-sized M?  Run twice: plain, and with PYTORCH_TUNABLEOP_ENABLED=1."""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

shapes = {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (28672, 4096), "down": (4096, 14336)}
Ms = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1024,2048,2560,3072,4096".split(","))]
dev = "cuda"
res = {}
for name, (N, K) in shapes.items():
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    for M in Ms:
        x = torch.randn(M, K, device=dev).bfloat16()
        for _ in range(3):
            F.linear(x, w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            F.linear(x, w)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res[f"{name}_M{M}"] = round(2 * M * N * K / dt / 1e12, 1)
        print(name, M, f"{dt*1e6:.1f} us", res[f"{name}_M{M}"], "TF", flush=True)
print(json.dumps({"tunable": os.environ.get("PYTORCH_TUNABLEOP_ENABLED", "0"), "tflops": res}))
