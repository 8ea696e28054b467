"""MoE dispatch glue: the fused gfx950 path is one C++ op that launches
route → align → grouped GEMM (SwiGLU epilogue) → grouped GEMM (weighted scatter) → sum
(``csrc/kernels/moe.hip``); shapes it does not cover raise instead of silently falling back."""
from __future__ import annotations

import torch


def hip_moe_supported(x: torch.Tensor, w13: torch.Tensor) -> bool:
    if x.dtype != torch.bfloat16:
        raise TypeError(f"the gfx950 MoE kernels take bfloat16 activations, got {x.dtype}")
    H, I = x.shape[1], w13.shape[1] // 2
    if H % 128 or I % 64:
        raise ValueError(f"grouped GEMM needs hidden % 128 == 0 and intermediate % 64 == 0, got {H}, {I}")
    return True


def fused_moe_hip(x, router_logits, w13, w2, top_k, expert_offset, out=None):
    from . import hip
    return hip().fused_moe(x.contiguous(), router_logits.contiguous(), w13, w2, top_k, expert_offset, 0, out)
