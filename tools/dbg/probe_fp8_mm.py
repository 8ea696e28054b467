"""Does torch._scaled_mm run OCP e4m3 GEMMs on this box, and how fast against bf16 at the 8B shapes?"""
import time
import torch

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

for dt in (torch.float8_e4m3fn, torch.float8_e4m3fnuz):
    try:
        M, N, K = 2304, 28672, 4096
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        xs = x.abs().amax().float() / 448.0; ws = w.abs().amax().float() / 448.0
        xq = (x.float() / xs).to(dt); wq = (w.float() / ws).to(dt)
        y = torch._scaled_mm(xq, wq.t(), scale_a=xs, scale_b=ws, out_dtype=torch.bfloat16)
        ref = x.float() @ w.float().t()
        err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
        print(dt, "ok, rel err", round(err, 4))
        for (M, N, K) in ((2304, 6144, 4096), (2304, 4096, 4096), (2304, 28672, 4096), (2304, 4096, 14336), (256, 28672, 4096)):
            x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
            xq = x.to(dt); wq = w.to(dt); one = torch.tensor(1.0, device=dev)
            tb = t_graph(lambda: torch.nn.functional.linear(x, w))
            t8 = t_graph(lambda: torch._scaled_mm(xq, wq.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16))
            tq = t_graph(lambda: (x.float() / x.abs().amax().float()).to(dt))
            print(M, N, K, "bf16 us", round(tb, 1), "fp8 us", round(t8, 1), "x", round(tb / t8, 2), "act quant us", round(tq, 1))
    except Exception as e:
        print(dt, "FAILED", type(e).__name__, str(e)[:300])
