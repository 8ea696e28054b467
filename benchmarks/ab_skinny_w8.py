"""Decode-sized linears: bf16 weight stream vs e4m3 weight stream (W8A16), GPU time in a hipGraph, weights rotated past the
infinity cache.    python benchmarks/ab_skinny_w8.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_kernels as bk  # noqa: E402
from dli_amd import ops  # noqa: E402


def main():
    shapes = [("qkv", 6144, 4096, False), ("o", 4096, 4096, False), ("gate_up", 28672, 4096, True), ("down", 4096, 14336, False)]
    for name, N, K, swiglu in shapes:
        copies = max(2, -(-(1 << 30) // (N * K * 2)))
        ws = [(torch.randn(N, K, device=bk.DEV) * 0.02).bfloat16() for _ in range(copies)]
        qs = [ops.quant_weight_fp8(w) for w in ws]
        for M in (1, 8, 16):
            x = torch.randn(M, K, device=bk.DEV).bfloat16()
            st = {"i": 0}

            def nxt():
                st["i"] = (st["i"] + 1) % copies
                return st["i"]
            t16 = bk.timeit_graph(lambda: ops.hip().skinny_linear(x, ws[nxt()], None, swiglu), 4 * copies)
            t8 = bk.timeit_graph(lambda: ops.linear_w8(x, *qs[nxt()], swiglu=swiglu), 4 * copies)
            print(json.dumps({"shape": name, "M": M, "bf16_us": round(t16 * 1e6, 2), "w8_us": round(t8 * 1e6, 2), "x": round(t16 / t8, 2),
                              "w8 TB/s": round(N * K / t8 / 1e12, 2)}), flush=True)
        del ws, qs


if __name__ == "__main__":
    main()
