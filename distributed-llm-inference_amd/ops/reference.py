"""Plain PyTorch fp32 references of every hot op.

They are (a) the numerics oracle of the HIP kernel tests and (b) the execution
path on CPU tensors (OPT-125m plumbing config, multi-process gloo tests).  On a GPU the engine never
falls back to them silently: ``ops.__init__`` raises if the HIP extension is missing.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .kv_layout import gather_kv_ref, write_kv_ref


def rms_norm_ref(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return (y * weight.float()).to(x.dtype)


def fused_add_rms_norm_ref(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
    """In place: ``residual += x``; ``x = rmsnorm(residual) * weight``."""
    s = x.float() + residual.float()
    residual.copy_(s.to(residual.dtype))
    # normalise the *rounded* residual, as the kernel does (it keeps the sum in fp32 registers: the
    # two differ by one rounding of the residual, well inside bf16 tolerance)
    x.copy_(rms_norm_ref(s, weight, eps).to(x.dtype))


def silu_and_mul_ref(x: torch.Tensor) -> torch.Tensor:
    d = x.shape[-1] // 2
    return (F.silu(x[..., :d].float()) * x[..., d:].float()).to(x.dtype)


def build_cos_sin_cache(max_position: int, rotary_dim: int, theta: float, device="cpu") -> torch.Tensor:
    """``[max_position, rotary_dim]`` fp32: first half cos, second half sin (NeoX pairing i ↔ i+D/2)."""
    inv = 1.0 / (theta ** (torch.arange(0, rotary_dim, 2, dtype=torch.float64) / rotary_dim))
    ang = torch.arange(max_position, dtype=torch.float64)[:, None] * inv[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=-1).float().to(device)


def rope_ref(x: torch.Tensor, positions: torch.Tensor, cos_sin: torch.Tensor) -> torch.Tensor:
    """x ``[T, H, D]`` → rotate-half RoPE in fp32, returned in x.dtype."""
    half = x.shape[-1] // 2
    cs = cos_sin[positions.long()]
    cos, sin = cs[:, None, :half], cs[:, None, half:]
    xf = x.float()
    a, b = xf[..., :half], xf[..., half:]
    return torch.cat([a * cos - b * sin, b * cos + a * sin], dim=-1).to(x.dtype)


def rope_and_cache_ref(qkv: torch.Tensor, positions: torch.Tensor, cos_sin, slot_mapping: torch.Tensor,
                       k_cache: torch.Tensor, v_cache: torch.Tensor, num_heads: int, num_kv_heads: int,
                       head_dim: int, k_scale: float = 1.0, v_scale: float = 1.0):
    """In place on ``qkv [T, (Hq+2Hkv)·D]``: RoPE on q and k (skipped when ``cos_sin`` is None), then
    k, v → paged cache at ``slot_mapping``."""
    T = qkv.shape[0]
    qd, kd = num_heads * head_dim, num_kv_heads * head_dim
    q = qkv[:, :qd].view(T, num_heads, head_dim)
    k = qkv[:, qd:qd + kd].view(T, num_kv_heads, head_dim)
    v = qkv[:, qd + kd:].view(T, num_kv_heads, head_dim)
    if cos_sin is not None:
        q.copy_(rope_ref(q, positions, cos_sin))
        k.copy_(rope_ref(k, positions, cos_sin))
    write_kv_ref(k, v, slot_mapping, k_cache, v_cache, k_scale, v_scale)


def paged_attention_ref(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, block_tables: torch.Tensor,
                        seq_lens: torch.Tensor, query_start_loc: torch.Tensor, scale: float,
                        k_scale: float = 1.0, v_scale: float = 1.0) -> torch.Tensor:
    """Varlen causal attention over the paged cache.  q ``[T, Hq, D]``; sequence i owns query rows
    ``query_start_loc[i]:query_start_loc[i+1]`` which are the *last* tokens of its ``seq_lens[i]``
    cached tokens.  Returns ``[T, Hq·D]`` in q.dtype."""
    T, Hq, D = q.shape
    bs = v_cache.shape[-1]
    Hkv = v_cache.shape[1]
    g = Hq // Hkv
    out = torch.zeros(T, Hq, D, dtype=torch.float32, device=q.device)
    qs = query_start_loc.tolist()
    for i, L in enumerate(seq_lens.tolist()):
        a, b = qs[i], qs[i + 1]
        n = b - a
        if n <= 0 or L <= 0:
            continue
        nb = (L + bs - 1) // bs
        K, V = gather_kv_ref(k_cache, v_cache, block_tables[i, :nb], L, k_scale, v_scale)   # [Hkv, L, D]
        qi = q[a:b].float().permute(1, 0, 2).reshape(Hkv, g, n, D)
        s = torch.einsum("hgnd,hld->hgnl", qi, K) * scale
        qpos = torch.arange(L - n, L, device=q.device)[:, None]
        kpos = torch.arange(L, device=q.device)[None, :]
        s = s.masked_fill(kpos > qpos, float("-inf"))
        p = torch.softmax(s, dim=-1)
        o = torch.einsum("hgnl,hld->hgnd", p, V).reshape(Hq, n, D).permute(1, 0, 2)
        out[a:b] = o
    return out.reshape(T, Hq * D).to(q.dtype)


def sample_ref(logits: torch.Tensor, temperature: torch.Tensor, top_p: torch.Tensor, top_k: torch.Tensor,
               uniform_noise: torch.Tensor | None = None, generator: torch.Generator | None = None) -> torch.Tensor:
    """Reference sampler: temperature → top-k → top-p (nucleus, smallest set with mass ≥ p, the
    most-probable token always kept) → draw.  ``temperature == 0`` rows are greedy.

    The draw uses the Gumbel-max trick (``argmax(log p + g)``), the same formulation as the HIP
    kernel; pass ``uniform_noise [B, V]`` in (0,1) to make it reproducible bit-for-bit.
    """
    B, V = logits.shape
    lf = logits.float()
    greedy = temperature <= 0
    t = torch.where(greedy, torch.ones_like(temperature), temperature).float()
    z = lf / t[:, None]
    # top-k
    k = torch.where(top_k <= 0, torch.full_like(top_k, V), top_k).clamp(max=V).long()
    srt, _ = z.sort(dim=-1, descending=True)
    kth = srt.gather(1, (k - 1)[:, None])
    z = z.masked_fill(z < kth, float("-inf"))
    # top-p on the renormalised distribution
    p = torch.softmax(z, dim=-1)
    ps, idx = p.sort(dim=-1, descending=True)
    csum = ps.cumsum(-1)
    drop_sorted = (csum - ps) >= top_p.float()[:, None]           # mass before this token already ≥ p
    drop_sorted[:, 0] = False
    drop = torch.zeros_like(drop_sorted).scatter(1, idx, drop_sorted)
    logp = torch.log(p).masked_fill(drop, float("-inf"))
    if uniform_noise is None:
        uniform_noise = torch.rand(B, V, generator=generator, device=logits.device).clamp_(1e-10, 1 - 1e-7)
    gumbel = -torch.log(-torch.log(uniform_noise.float()))
    sampled = (logp + gumbel).argmax(-1)
    return torch.where(greedy, lf.argmax(-1), sampled)


def moe_ref(x: torch.Tensor, router_logits: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor, top_k: int,
            expert_offset: int = 0) -> torch.Tensor:
    """Mixtral MoE block: softmax over all experts → top-k → renormalise → Σ w·expert(x).

    ``w13 [E_local, 2·I, H]`` (gate rows then up rows), ``w2 [E_local, H, I]``; only experts
    ``expert_offset … expert_offset+E_local-1`` are evaluated (expert parallelism: the partial sums
    of the ranks are all-reduced by the caller)."""
    probs = torch.softmax(router_logits.float(), dim=-1)
    topv, topi = probs.topk(top_k, dim=-1)
    topv = topv / topv.sum(-1, keepdim=True)
    out = torch.zeros(x.shape, dtype=torch.float32, device=x.device)
    for e in range(w13.shape[0]):
        sel = (topi == e + expert_offset)
        rows = sel.any(-1).nonzero(as_tuple=True)[0]
        if rows.numel() == 0:
            continue
        wgt = (topv * sel)[rows].sum(-1, keepdim=True)
        h = x[rows].float() @ w13[e].float().t()
        i = h.shape[-1] // 2
        h = F.silu(h[:, :i]) * h[:, i:]
        out[rows] += wgt * (h @ w2[e].float().t())
    return out.to(x.dtype)


def attention_scale(head_dim: int) -> float:
    return 1.0 / math.sqrt(head_dim)
