"""fused_moe at decode-to-mid batch sizes (the 128-row grouped tile and the skinny path), GPU time in a hipGraph —
used for same-box A/Bs of two builds of ops/_C.so.    python benchmarks/ab_moe_small.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_kernels as bk  # noqa: E402
from dli_amd import ops  # noqa: E402


def main():
    H, I, E = 4096, 14336, 8
    w13 = (torch.randn(E, 2 * I, H, device=bk.DEV) * 0.02).bfloat16()
    w2 = (torch.randn(E, H, I, device=bk.DEV) * 0.02).bfloat16()
    for T in (8, 32, 48, 64, 128, 256, 384):
        x = torch.randn(T, H, device=bk.DEV).bfloat16()
        logits = torch.randn(T, E, device=bk.DEV).bfloat16()
        t = bk.timeit_graph(lambda: ops.fused_moe(x, logits, w13, w2, 2, 0), 8)
        print(json.dumps({"T": T, "us": round(t * 1e6, 1), "TB/s of expert weights": round(E * 3 * I * H * 2 / t / 1e12, 2)}), flush=True)


if __name__ == "__main__":
    main()
