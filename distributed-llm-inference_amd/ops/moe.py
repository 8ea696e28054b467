"""MoE dispatch glue: routing/alignment, grouped GEMM, weighted un-permute."""
from __future__ import annotations

import torch


def hip_moe_supported(x: torch.Tensor, w13: torch.Tensor) -> bool:
    from . import hip
    lib = hip()
    return hasattr(lib, "moe_align") and x.dtype == torch.bfloat16 and x.shape[1] % 64 == 0 and w13.shape[1] % 256 == 0


def fused_moe_hip(x, router_logits, w13, w2, top_k, expert_offset):
    raise NotImplementedError
