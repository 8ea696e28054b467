"""Does PyTorch TunableOp find faster hipBLASLt / rocBLAS solutions for the serving loop's bf16 GEMM shapes?"""
import time
import torch
import torch.nn.functional as F

dev = "cuda"
def t_graph(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        for _ in range(n): fn()
    torch.cuda.synchronize(); g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n

shapes = [(2304, 6144, 4096), (2304, 4096, 4096), (2304, 4096, 14336), (2304, 28672, 4096), (1024, 4096, 14336), (256, 6144, 4096)]
data = {}
for (M, N, K) in shapes:
    copies = max(2, -(-(600 << 20) // (N * K * 2)))
    data[(M, N, K)] = (torch.randn(M, K, device=dev).bfloat16(), [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(copies)])
def bench(tag):
    for (M, N, K), (x, ws) in data.items():
        st = {"i": 0}
        def run():
            st["i"] = (st["i"] + 1) % len(ws)
            return F.linear(x, ws[st["i"]])
        print(tag, M, N, K, round(t_graph(run, 2 * len(ws)), 1), "us", flush=True)
bench("default")
import torch.cuda.tunable as tun
tun.enable(True); tun.tuning_enable(True)
try:
    tun.set_max_tuning_duration(300); tun.set_max_tuning_iterations(20)
except Exception as e:
    print("limits:", e)
t0 = time.time()
for (M, N, K), (x, ws) in data.items():
    t1 = time.time(); F.linear(x, ws[0]); torch.cuda.synchronize(); print("tuned", M, N, K, "in", round(time.time() - t1, 1), "s", flush=True)
print("tuning total", round(time.time() - t0, 1), "s")
tun.tuning_enable(False)
bench("tunable")
try:
    print(tun.get_results()[-8:])
except Exception as e:
    print(e)
