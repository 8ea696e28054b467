"""Physical layout of the paged KV cache (shared by the HIP kernels and the torch references).

A *page* holds ``KV_BLOCK_SIZE = 32`` tokens of one layer.  Inside a page the bytes are ordered so
that every wave-level load of the attention kernels is one contiguous 1 KiB read that lands in
registers already shaped as an MFMA operand (``v_mfma_f32_16x16x32_bf16``):

* **K page** ``[Hkv][D/CH][32 tokens][CH]`` with ``CH = 64 bytes / itemsize`` (32 for bf16, 64 for
  fp8): 16 consecutive token rows × 64 B are the A operand (rows = tokens, k = head-dim slice) of the
  ``Sᵀ = K·Qᵀ`` product — lane ``l`` reads 16 B at ``token (l & 15)``, ``d = CH·c + 8·(l >> 4)``.
* **V page** ``[Hkv][D][32 token slots]`` (transposed, tokens innermost).  Token ``t`` of the page
  lives in slot ``vperm(t) = ((t & 15) >> 2)·8 + (t >> 4)·4 + (t & 3)``, which is exactly the order
  in which the ``Sᵀ`` accumulators of two 16-token tiles sit in a lane's registers — so ``P`` feeds
  the ``Oᵀ = Vᵀ·Pᵀ`` MFMA as the B operand with no cross-lane movement, and 16 ``d`` rows × 64 B are
  its A operand.

Both pages are 8 KiB per KV head at D = 128 / bf16.
"""
from __future__ import annotations

import torch

from ..config import KV_BLOCK_SIZE


def chunk_width(head_dim: int, dtype: torch.dtype) -> int:
    return min(head_dim, 64 // torch.empty((), dtype=dtype).element_size())


def v_slot_perm(block_size: int = KV_BLOCK_SIZE) -> torch.Tensor:
    """``perm[t]`` = slot of token ``t`` inside a V page."""
    assert block_size == 32, "the V page permutation is defined for 32-token pages"
    t = torch.arange(block_size)
    return ((t & 15) >> 2) * 8 + (t >> 4) * 4 + (t & 3)


def kv_cache_shapes(num_blocks: int, num_kv_heads: int, head_dim: int, dtype: torch.dtype,
                    block_size: int = KV_BLOCK_SIZE):
    ch = chunk_width(head_dim, dtype)
    assert head_dim % ch == 0
    return ((num_blocks, num_kv_heads, head_dim // ch, block_size, ch),
            (num_blocks, num_kv_heads, head_dim, block_size))


def allocate_kv_cache(num_layers: int, num_blocks: int, num_kv_heads: int, head_dim: int,
                      dtype: torch.dtype, device, block_size: int = KV_BLOCK_SIZE):
    """One K and one V tensor per layer (zero-filled so that padded page slots are finite)."""
    ks, vs = kv_cache_shapes(num_blocks, num_kv_heads, head_dim, dtype, block_size)
    caches = []
    for _ in range(num_layers):
        if dtype == torch.float8_e4m3fn:
            k = torch.zeros(ks, dtype=torch.uint8, device=device).view(dtype)
            v = torch.zeros(vs, dtype=torch.uint8, device=device).view(dtype)
        else:
            k = torch.zeros(ks, dtype=dtype, device=device)
            v = torch.zeros(vs, dtype=dtype, device=device)
        caches.append((k, v))
    return caches


def bytes_per_block(num_layers: int, num_kv_heads: int, head_dim: int, dtype: torch.dtype,
                    block_size: int = KV_BLOCK_SIZE) -> int:
    return 2 * num_layers * num_kv_heads * head_dim * block_size * torch.empty((), dtype=dtype).element_size()


# ------------------------------------------------------------------ torch reference pack / unpack
def write_kv_ref(k: torch.Tensor, v: torch.Tensor, slot_mapping: torch.Tensor,
                 k_cache: torch.Tensor, v_cache: torch.Tensor, k_scale: float = 1.0, v_scale: float = 1.0):
    """k, v: ``[T, Hkv, D]``.  Slots < 0 are skipped (padding tokens)."""
    bs = v_cache.shape[-1]
    keep = slot_mapping >= 0
    if not bool(keep.all()):
        k, v, slot_mapping = k[keep], v[keep], slot_mapping[keep]
    slot = slot_mapping.long()
    blk, tok = slot // bs, slot % bs
    nchunk, ch = k_cache.shape[2], k_cache.shape[4]
    T, H, _D = k.shape
    if k_cache.dtype == torch.float8_e4m3fn:
        # scales: python floats, or per-kv-head tensors [Hkv] (k, v are [T, Hkv, D])
        ks = k_scale.float().view(1, -1, 1) if torch.is_tensor(k_scale) else k_scale
        vs = v_scale.float().view(1, -1, 1) if torch.is_tensor(v_scale) else v_scale
        k = (k.float() / ks).clamp(-448, 448)
        v = (v.float() / vs).clamp(-448, 448)
    perm = v_slot_perm(bs).to(slot.device)
    kq, vq = k.reshape(T, H, nchunk, ch).to(k_cache.dtype), v.to(v_cache.dtype)
    if k_cache.dtype == torch.float8_e4m3fn:             # fp8 has no index_put: go through the bytes
        k_cache.view(torch.uint8)[blk, :, :, tok] = kq.view(torch.uint8)
        v_cache.view(torch.uint8)[blk, :, :, perm[tok]] = vq.view(torch.uint8)
    else:
        k_cache[blk, :, :, tok] = kq
        v_cache[blk, :, :, perm[tok]] = vq


def gather_kv_ref(k_cache: torch.Tensor, v_cache: torch.Tensor, block_ids: torch.Tensor, seq_len: int,
                  k_scale: float = 1.0, v_scale: float = 1.0):
    """Logical ``K, V`` of one sequence as fp32 ``[Hkv, seq_len, D]``."""
    bs = v_cache.shape[-1]
    ids = block_ids.long()
    if k_cache.dtype == torch.float8_e4m3fn:             # fp8 has no gather: go through the bytes
        ks = k_scale.float().view(1, -1, 1, 1, 1) if torch.is_tensor(k_scale) else k_scale
        vs = v_scale.float().view(1, -1, 1, 1) if torch.is_tensor(v_scale) else v_scale
        kb = k_cache.view(torch.uint8)[ids].view(torch.float8_e4m3fn).float() * ks
        vb = v_cache.view(torch.uint8)[ids].view(torch.float8_e4m3fn).float() * vs
    else:
        kb = k_cache[ids].float() * k_scale              # [nb, H, nchunk, bs, ch]
        vb = v_cache[ids].float() * v_scale              # [nb, H, D, bs(slots)]
    nb, H, nchunk, _, ch = kb.shape
    K = kb.permute(1, 0, 3, 2, 4).reshape(H, nb * bs, nchunk * ch)[:, :seq_len]
    perm = v_slot_perm(bs).to(vb.device)
    V = vb[..., perm].permute(1, 0, 3, 2).reshape(H, nb * bs, vb.shape[2])[:, :seq_len]
    return K, V
