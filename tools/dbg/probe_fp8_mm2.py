"""Row-wise scaled fp8 GEMM support and accuracy; small-M timing."""
import torch
dev = "cuda"
dt = torch.float8_e4m3fn
def t_graph(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        for _ in range(n): fn()
    torch.cuda.synchronize(); g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
M, N, K = 2304, 6144, 4096
x = torch.randn(M, K, device=dev).bfloat16() * (1 + 10 * torch.rand(M, 1, device=dev)).bfloat16()
w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
ref = x.float() @ w.float().t()
# per-tensor
xs = x.abs().amax().float() / 448; ws = w.abs().amax().float() / 448
y = torch._scaled_mm((x.float() / xs).to(dt), (w.float() / ws).to(dt).t(), scale_a=xs, scale_b=ws, out_dtype=torch.bfloat16)
print("per-tensor: cos", torch.nn.functional.cosine_similarity(y.float().flatten(), ref.flatten(), dim=0).item(),
      "rel rms", ((y.float() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
try:
    xs = (x.abs().amax(dim=1, keepdim=True).float() / 448).contiguous(); ws = (w.abs().amax(dim=1, keepdim=True).float() / 448)
    xq = (x.float() / xs).to(dt); wq = (w.float() / ws).to(dt)
    y = torch._scaled_mm(xq, wq.t(), scale_a=xs, scale_b=ws.t().contiguous(), out_dtype=torch.bfloat16)
    print("row-wise: cos", torch.nn.functional.cosine_similarity(y.float().flatten(), ref.flatten(), dim=0).item(),
          "rel rms", ((y.float() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
    print("row-wise us", round(t_graph(lambda: torch._scaled_mm(xq, wq.t(), scale_a=xs, scale_b=ws.t().contiguous(), out_dtype=torch.bfloat16)), 1))
    one = torch.tensor(1.0, device=dev)
    print("per-tensor us", round(t_graph(lambda: torch._scaled_mm(xq, wq.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)), 1))
except Exception as e:
    print("row-wise FAILED", type(e).__name__, str(e)[:400])
one = torch.tensor(1.0, device=dev)
for (M, N, K) in ((64, 28672, 4096), (128, 28672, 4096), (512, 28672, 4096), (1024, 4096, 14336), (4096, 28672, 4096), (2300, 6144, 4096), (2304, 7168, 8192), (2304, 1280, 8192)):
    try:
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        xq = x.to(dt); wq = w.to(dt)
        tb = t_graph(lambda: torch.nn.functional.linear(x, w))
        t8 = t_graph(lambda: torch._scaled_mm(xq, wq.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16))
        print(M, N, K, "bf16 us", round(tb, 1), "fp8 us", round(t8, 1), "x", round(tb / t8, 2))
    except Exception as e:
        print(M, N, K, "FAILED", str(e)[:200])
