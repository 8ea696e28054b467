"""The fp8-option kernels on serving-loop shapes (a target for tools/profile_kernels.py / rocprofv3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_kernels as bk  # noqa: E402
from dli_amd import ops  # noqa: E402

x = torch.randn(2304, 4096, device=bk.DEV).bfloat16()
gu = torch.randn(2304, 2 * 14336, device=bk.DEV).bfloat16()
w = (torch.randn(28672, 4096, device=bk.DEV) * 0.02).bfloat16()
wq, ws = ops.quant_weight_fp8(w)
x1 = torch.randn(8, 4096, device=bk.DEV).bfloat16()
for _ in range(20):
    q, s = ops.quant_rows_fp8(x)
    a, sa = ops.quant_rows_fp8(gu, silu=True)
    ops.linear_fp8(x, wq, ws, xq=(q, s))
    ops.linear_w8(x1, wq, ws, swiglu=True)
    ops.hip().skinny_linear(x1, w, None, True)
torch.cuda.synchronize()
