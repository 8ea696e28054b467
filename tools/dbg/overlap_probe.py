"""Probe: do the HBM-bound decode attention kernel and an MFMA-bound hipBLASLt GEMM overlap when issued on two
streams?  (Would a two-micro-batch schedule hide decode attention under the GEMMs?)"""
import os, sys, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "benchmarks"))
import torch
import torch.nn.functional as F
import bench_kernels as bk

attn, nbytes, splits, mean = bk._decode_mixed_case(600, 770)
M = 1400
x = torch.randn(M, 4096, device="cuda").bfloat16()
w = (torch.randn(28672, 4096, device="cuda") * 0.02).bfloat16()
y = torch.empty(M, 28672, device="cuda", dtype=torch.bfloat16)
gemm = lambda: torch.mm(x, w.t(), out=y)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

def both():
    with torch.cuda.stream(s1):
        attn()
    with torch.cuda.stream(s2):
        gemm()

def seq():
    attn()
    gemm()

ta, tg, ts, tb = timed(attn), timed(gemm), timed(seq), timed(both)
print(f"attention {ta:.0f} us ({nbytes/ta/1e6:.2f} TB/s)  gemm {tg:.0f} us ({2*M*28672*4096/tg/1e6:.0f} TF)  sequential {ts:.0f} us  two streams {tb:.0f} us")
