"""Hot ops of the serving stack (SURVEY.md §2.5 K1–K8).

GPU tensors go to the hand-written gfx950 kernels in ``csrc/kernels`` through ``torch.ops.dli``
(built in-tree into ``ops/_C.so`` by ``dli_amd.build``); CPU tensors go to the fp32 torch references
in ``reference.py``.  There is no silent GPU fallback: a missing extension or an unsupported shape on
a GPU tensor raises.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import reference as ref
from .gemm_plan import PLAN as _PLAN, TAIL as _TAIL
from .kv_layout import allocate_kv_cache, bytes_per_block, kv_cache_shapes  # noqa: F401

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_C.so")
_lib = None


def hip():
    """The ``torch.ops.dli`` namespace; loads ``_C.so`` on first use and fails loudly without it."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(f"HIP extension {_SO} is missing: run `python -m dli_amd.build` "
                               "(hipcc --offload-arch=gfx950) before using GPU tensors")
        torch.ops.load_library(_SO)
        _lib = torch.ops.dli
    return _lib


def hip_loaded() -> bool:
    return _lib is not None


def _gpu_kernel_ok(x: torch.Tensor) -> bool:
    if not x.is_cuda:
        return False
    if x.dtype != torch.bfloat16:
        raise TypeError(f"the gfx950 kernels take bfloat16 activations, got {x.dtype}")
    return True


# ------------------------------------------------------------------------------------ metadata
@dataclass
class AttnMetadata:
    """Per-step attention inputs.  Sequences are ordered decodes first (one query token each, rows
    ``0 … num_decode-1`` of the flattened batch), then prefill chunks."""
    num_decode: int
    num_prefill: int
    num_tokens: int
    positions: torch.Tensor            # [T] int32
    slot_mapping: torch.Tensor         # [T] int32 (-1 = padding)
    block_tables: torch.Tensor         # [B, max_blocks] int32
    seq_lens: torch.Tensor             # [B] int32, context length *including* this step's tokens
    query_start_loc: torch.Tensor      # [B+1] int32
    max_seq_len: int = 0
    decode_splits: int = 1             # uniform-split decode (no work list): launch-wide split count
    decode_workspace: Optional[torch.Tensor] = None
    # work-list decode (build_decode_work): device views of the step buffer + the LAUNCHED grid extents (exact for an
    # eager step, the capture-time upper bounds inside a hipGraph; the valid counts are counts[0:2] on the device)
    decode_work: Optional[torch.Tensor] = None      # int32 [items_cap, 4]
    decode_red: Optional[torch.Tensor] = None       # int32 [reds_cap, 4]
    counts: Optional[torch.Tensor] = None           # int32 [4]: decode items, reduce entries, prefill tiles, -
    decode_items: int = 0
    decode_reds: int = 0
    decode_max_parts: int = 0
    counts_on_device: bool = False     # hipGraph step: kernels take the prefill tile count from counts[2] as well
    # prefill work list: one entry per TQ-token tile
    prefill_tile_seq: Optional[torch.Tensor] = None
    prefill_tile_q0: Optional[torch.Tensor] = None
    num_prefill_tiles: int = 0


def prefill_tile_tokens(num_heads: int, num_kv_heads: int) -> int:
    g = num_heads // num_kv_heads
    return 16 * max(8 // g, 1)


def build_prefill_tiles(query_lens: np.ndarray, first_seq: int, tile_tokens: int, seq_lens: np.ndarray | None = None):
    """Host-side work list of the prefill kernel: (sequence index, first query offset) per tile.

    The kernel walks the list from its END (causal tiles late in a chunk see the most context).  With ``seq_lens``
    (context length including the new tokens, per sequence) the list is sorted by the pages a tile reads, ascending, so
    that walk is longest-processing-time-first over ALL sequences of the step — a chunk that continues a 1500-token
    context starts before the tiles of fresh short prompts."""
    seqs, q0s, work = [], [], []
    for i, n in enumerate(query_lens):
        nt = (int(n) + tile_tokens - 1) // tile_tokens
        seqs.append(np.full(nt, first_seq + i, dtype=np.int32))
        q0 = np.arange(nt, dtype=np.int32) * tile_tokens
        q0s.append(q0)
        if seq_lens is not None:
            work.append(int(seq_lens[i]) - int(n) + q0)      # context in front of the tile (+ a constant)
    if not seqs:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    ts, t0 = np.concatenate(seqs), np.concatenate(q0s)
    if seq_lens is not None and len(query_lens) > 1:
        order = np.argsort(np.concatenate(work), kind="stable")
        ts, t0 = ts[order], t0[order]
    return ts, t0


def choose_decode_splits(num_seqs: int, num_kv_heads: int, max_seq_len: int, num_cus: int = 256) -> int:
    """Split the KV range when (sequences × KV heads) alone cannot fill the chip."""
    groups = max(num_seqs * num_kv_heads, 1)
    pages = max((max_seq_len + 31) // 32, 1)
    want = (2 * num_cus + groups - 1) // groups          # aim for ≥ 2 workgroups per CU
    return int(max(1, min(want, (pages + 3) // 4, 16)))   # keep ≥ 4 pages (one per wave) per split


def decode_chunk_pages(seq_lens: np.ndarray, num_kv_heads: int, cap_items: int = 0, num_cus: int = 256) -> int:
    """Pages per work item of the decode kernel.  Measured (``profiles/kernels_decode_r3b.txt``): once (rows × kv heads)
    alone fill the chip with ≥ 2 workgroups per CU, cutting rows further only LOSES bandwidth (B = 1024: whole rows
    5.73 TB/s, 32-page ranges 5.64, 16-page 5.40, 8-page 5.11 — per-workgroup fixed cost and the extra reduce pass
    outweigh the better balance; row order, longest first, is what recovers the tail), so big batches get ONE range per
    row.  Smaller batches cut the longest row into as many ranges as it takes to reach that fill (whole multiples of 4
    pages: one per wave and iteration) — each row by its own length, a short row stays one range.  ``cap_items``:
    ranges are doubled until the list fits (the fixed grid of a hipGraph)."""
    pages = (np.asarray(seq_lens, dtype=np.int64) + 31) >> 5
    n, maxp = len(pages), max(int(pages.max()) if len(pages) else 1, 1)
    want = (2 * num_cus + n * num_kv_heads - 1) // max(n * num_kv_heads, 1)      # ranges the longest row should give
    if want <= 1:
        return (maxp + 3) // 4 * 4
    c = (max((maxp + want - 1) // want, 4) + 3) // 4 * 4
    if cap_items:
        while int(((pages + c - 1) // c).sum()) > cap_items and c < maxp:
            c *= 2
    return c


def decode_launch_caps(rows: int, num_kv_heads: int, num_cus: int = 256):
    """Upper bounds ``(work items, reduce entries)`` of a decode launch over at most ``rows`` rows — the grid a hipGraph
    captures.  Rows are only ever cut while (rows × kv heads) < 2 workgroups per CU, and then into about that many
    ranges in total, so the list never exceeds ``rows + 2·CUs / kv heads`` and at most ``2·CUs / kv heads`` rows are cut."""
    fill = 2 * num_cus // max(num_kv_heads, 1)
    return rows + fill, min(rows, fill)


def build_decode_work(seq_lens: np.ndarray, num_kv_heads: int, cap_items: int = 0, chunk_pages: int = 0):
    """Host-side work list of the decode kernel (no launch-wide split count).

    Every sequence's pages are cut into ``ceil(pages / chunk)`` ranges of EQUAL size (rounded up to 4 pages), so a
    2048-token row becomes several workgroups per kv head and a 40-token row one.  Returns
    ``work int32 [items, 4] = (row, first page, end page, partial slot | -1)``, ``red int32 [r, 4] = (row, first
    partial, partials, 0)`` for the rows with more than one range, and the number of partial slots."""
    L = np.asarray(seq_lens, dtype=np.int64)
    n = len(L)
    if n == 0:
        return np.zeros((0, 4), np.int32), np.zeros((0, 4), np.int32), 0
    pages = np.maximum((L + 31) >> 5, 1)
    c = chunk_pages or decode_chunk_pages(L, num_kv_heads, cap_items)
    k = (pages + c - 1) // c                                   # ranges per row
    per = ((pages + k - 1) // k + 3) // 4 * 4                  # equal-size ranges, whole wave rounds
    k = (pages + per - 1) // per
    total = int(k.sum())
    row = np.repeat(np.arange(n), k)
    first = np.cumsum(k) - k
    j = np.arange(total) - first[row]
    begin = j * per[row]
    end = np.minimum(begin + per[row], pages[row])
    multi = k > 1
    pk = np.where(multi, k, 0)
    pfirst = np.cumsum(pk) - pk
    slot = np.where(multi[row], pfirst[row] + j, -1)
    work = np.stack([row, begin, end, slot], axis=1).astype(np.int32)
    mrows = np.flatnonzero(multi)
    red = np.stack([mrows, pfirst[mrows], k[mrows], np.zeros_like(mrows)], axis=1).astype(np.int32)
    return work, red, int(pk.sum())


def attach_decode_worklist(meta: AttnMetadata, seq_lens: np.ndarray, num_heads: int, num_kv_heads: int, device,
                           cap_items: int = 0, chunk_pages: int = 0, launch_items: int = 0) -> AttnMetadata:
    """Tests / benchmarks: build the decode work list of ``seq_lens`` (host copy of the first ``meta.num_decode`` rows)
    and hang it — device tensors, workspace, counts — on ``meta``.  ``launch_items`` > 0 launches that many workgroups
    per kv head instead of the exact count (what a hipGraph does); the kernel takes the valid count from ``counts``."""
    work, red, _ = build_decode_work(seq_lens, num_kv_heads, cap_items, chunk_pages)
    n_items, n_reds = len(work), len(red)
    li = max(launch_items, n_items)
    lr = max(launch_items, n_reds) if launch_items else n_reds
    w = np.zeros((li + 1, 4), np.int32)
    w[:n_items] = work
    r = np.zeros((lr + 1, 4), np.int32)
    r[:n_reds] = red
    meta.decode_work = torch.from_numpy(w).to(device)
    meta.decode_red = torch.from_numpy(r).to(device)
    meta.counts = torch.tensor([n_items, n_reds, meta.num_prefill_tiles, 0], dtype=torch.int32, device=device)
    meta.decode_items, meta.decode_reds = li, lr
    meta.decode_max_parts = max(li, 1)
    meta.decode_workspace = torch.empty(meta.decode_max_parts * num_heads * 130, dtype=torch.float32, device=device)
    return meta


# ------------------------------------------------------------------------------------ ops
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    if _gpu_kernel_ok(x):
        return hip().rms_norm(x, weight, eps)
    return ref.rms_norm_ref(x, weight, eps)


def fused_add_rms_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> None:
    """In place: ``residual += x``; ``x = rmsnorm(residual) * weight``."""
    if _gpu_kernel_ok(x):
        hip().fused_add_rms_norm(x, residual, weight, eps)
    else:
        ref.fused_add_rms_norm_ref(x, residual, weight, eps)


def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    if _gpu_kernel_ok(x):
        return hip().silu_and_mul(x)
    return ref.silu_and_mul_ref(x)


SKINNY_MAX_ROWS = 0 if os.environ.get("DLI_NO_SKINNY_GEMM") else 16
SKINNY_MULTIPASS_BYTES = 32 << 20


def skinny_rows_ok(m: int, n: int, k: int, row_multiple: int = 16) -> bool:
    """Batch sizes and shapes for which the weight-streaming kernel beats the library GEMM: up to 16 rows always
    (one pass over W at 4–6 TB/s); up to 64 rows in 16-row passes while passes × weight bytes ≤ 32 MiB — the small
    tensor-parallel shards, where re-reading 16 weight rows from L2 is cheaper than the library GEMM's fixed cost
    (GPU time inside a hipGraph: ``profiles/skinny_linear_graph_timed_r1n.jsonl``)."""
    if not (0 < m <= 4 * SKINNY_MAX_ROWS and k % 128 == 0 and n % row_multiple == 0):
        return False
    return m <= SKINNY_MAX_ROWS or (m + 15) // 16 * n * k * 2 <= SKINNY_MULTIPASS_BYTES


def _skinny_ok(x: torch.Tensor, weight: torch.Tensor, row_multiple: int) -> bool:
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16):
        return False
    return (skinny_rows_ok(x.shape[0], weight.shape[0], weight.shape[1], row_multiple) and weight.is_contiguous()
            and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0)


def skinny_w8_ok(x: torch.Tensor, wq: torch.Tensor, row_multiple: int = 16) -> bool:
    """Decode-sized step of a quantised layer: the weight-streaming kernel reads the e4m3 copy (W8A16, half the bytes).
    One pass only (≤ 16 rows): 17 rows and more go to the W8A8 GEMM."""
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 and wq.dtype == torch.float8_e4m3fn and wq.is_contiguous()
            and 0 < x.shape[0] <= SKINNY_MAX_ROWS and wq.shape[1] % 128 == 0 and wq.shape[0] % row_multiple == 0
            and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0)


def linear_w8(x: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor, swiglu: bool = False,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``x @ (wq · scale)ᵀ`` for ≤ 16 rows with the weights streamed as e4m3 (``skinny_linear_kernel<…, W8>``)."""
    return hip().skinny_linear(x, wq, None, swiglu, out, w_scale)


def _tile_operand_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.is_contiguous()
            and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
            and x.shape[0] * x.stride(0) * 2 < 2 ** 32 and weight.numel() * 2 < 2 ** 32)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``x @ weightᵀ + bias``.  Decode-sized GPU inputs take the weight-streaming MFMA kernel
    (``csrc/kernels/skinny_gemm.hip``); above that the start-up measurement (``ops/gemm_plan.py``) says, per layer shape
    and row count, whether the library GEMM (hipBLASLt through torch) or the own 256² tile — with K cut into slices at
    mid-M — runs; shapes it never measured go to the library.  ``out``: 2-D result buffer to write into (the row-parallel
    layers pass a view of the all-reduce's IPC staging buffer)."""
    if _skinny_ok(x, weight, 16):
        return hip().skinny_linear(x, weight, bias, False, out)
    if bias is None and _PLAN.table and x.is_cuda and x.dim() == 2:
        s = _PLAN.splits_for(x.shape[0], weight.shape[0], weight.shape[1], False)
        if s and _tile_operand_ok(x, weight):
            return hip().dense_gemm256(x, weight, False, 1, out, True) if s == _TAIL else hip().dense_gemm256(x, weight, False, s, out)
    if out is not None:
        return torch.mm(x, weight.t(), out=out) if bias is None else torch.addmm(bias, x, weight.t(), out=out)
    return torch.nn.functional.linear(x, weight, bias)


# Stream-capture error mode of every hipGraph capture in the package.  "global" (torch's default) makes a capture-unsafe HIP call
# on ANY thread fail while a capture is open — and the RCCL process group's watchdog thread polls its work events with
# hipEventQuery from its own thread: caught on the one-GPU box with a one-rank RCCL group (tests/test_rccl_gpu.py: "operation
# not permitted when stream is capturing" in the watchdog → terminate), timing-dependent.  "thread_local" polices the capturing
# thread only.  (The second half of the same problem — the watchdog polling an eager collective's event after the group's
# stream has joined a capture — is parallel.state.drain_collectives, called before every capture.)
CAPTURE_MODE = os.environ.get("DLI_CAPTURE_MODE", "thread_local")

FP8_MIN_ROWS = int(os.environ.get("DLI_FP8_MIN_ROWS", "17"))      # below: the weight-streaming kernel (e4m3 weights, bf16 activations)
FP8_MAX = 448.0                                                   # largest finite OCP e4m3 value


def use_fp8(x: torch.Tensor) -> bool:
    """Does a step of this size take the W8A8 GEMM of a quantised layer?  (On the GPU, smaller steps stream the e4m3 weights
    through the decode kernel — ``skinny_w8_ok`` — and keep their activations in bf16.)"""
    return x.dim() == 2 and x.shape[0] >= FP8_MIN_ROWS and x.dtype == torch.bfloat16


def quant_rows_fp8(x: torch.Tensor, silu: bool = False):
    """Per-row dynamic e4m3 quantisation: ``(q [T, K] float8_e4m3fn, scale [T, 1] fp32)`` with ``x ≈ q · scale``.
    ``silu``: ``x = [gate | up]`` and the quantised row is ``silu(gate) * up`` (``csrc/kernels/quant.hip``)."""
    if _gpu_kernel_ok(x) and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0:
        return hip().quant_rows_fp8(x, silu)
    xf = x.float()
    if silu:
        half = x.shape[1] // 2
        xf = torch.nn.functional.silu(xf[:, :half]) * xf[:, half:]
    scale = xf.abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / FP8_MAX
    return (xf / scale).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn), scale


def rms_norm_quant(x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor, eps: float):
    """``residual += x`` (when given), then ``quant_rows_fp8(rmsnorm(residual or x) * weight)`` — one kernel on the GPU
    (``rms_norm_kernel<…, QUANT>``): the normalised row goes straight to e4m3, it is never stored in bf16.  ``x`` is left
    untouched.  Returns ``(q, scale)`` for ``linear_fp8(xq=…)``."""
    if _gpu_kernel_ok(x) and x.shape[1] <= 8192 and x.shape[1] % 16 == 0 and x.stride(1) == 1:
        return hip().rms_norm_quant(x, residual, weight, eps)
    if residual is not None:
        y = x.clone()
        fused_add_rms_norm(y, residual, weight, eps)
    else:
        y = rms_norm(x, weight, eps)
    return quant_rows_fp8(y)


def quant_weight_fp8(w: torch.Tensor):
    """One scale per output channel: ``(q [N, K] float8_e4m3fn, scale [1, N] fp32)``."""
    wf = w.float()
    scale = wf.abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / FP8_MAX
    return (wf / scale).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).contiguous(), scale.t().contiguous()


def linear_fp8(x: Optional[torch.Tensor], wq: torch.Tensor, w_scale: torch.Tensor, out: Optional[torch.Tensor] = None,
               xq=None) -> torch.Tensor:
    """``x @ Wᵀ`` with both operands in e4m3 and fp32 accumulation on the fp8 matrix cores; per-token × per-channel scales
    applied by the GEMM's epilogue (hipBLASLt through ``torch._scaled_mm``), bf16 result.  ``xq``: an already quantised
    input ``(q, scale)`` (the SwiGLU output comes out of ``quant_rows_fp8(silu=True)`` that way)."""
    q, xs = xq if xq is not None else quant_rows_fp8(x)
    if q.is_cuda:
        if out is not None and out.is_contiguous():          # e.g. the all-reduce's IPC staging buffer: no copy pass
            return torch._scaled_mm(q, wq.t(), scale_a=xs, scale_b=w_scale, out_dtype=torch.bfloat16, out=out)
        y = torch._scaled_mm(q, wq.t(), scale_a=xs, scale_b=w_scale, out_dtype=torch.bfloat16)
    else:                                                    # reference (CPU tests): the same quantised operands in fp32
        y = ((q.float() * xs) @ (wq.float() * w_scale.t()).t()).to(torch.bfloat16)
    if out is not None:
        out.copy_(y)
        return out
    return y


GEMM256_SWIGLU = not os.environ.get("DLI_NO_GEMM256_SWIGLU")


def gemm256_swiglu_ok(m: int, n_out: int, k: int) -> bool:
    """Shapes for which the 256×256×64 tile kernel with the SwiGLU epilogue (``csrc/kernels/gemm256.hip``) beats the
    library GEMM + ``silu_and_mul`` (hipGraph-timed A/B on random data, ``profiles/gemm256_r2e.jsonl``: Llama-3-8B
    gate/up 1.04–1.18× at M = 512…8192; the Llama-3-70B TP = 8 shard 1.0–1.05× from M = 2048, slower below): the
    tiles must fill most of the chip (≥ 200 workgroups of one per CU) and their last row tile must not be mostly
    padding."""
    if not GEMM256_SWIGLU or m < 448 or n_out % 128 or k % 64:
        return False
    tiles = (m + 255) // 256
    return tiles * (n_out // 128) >= 200 and m >= 0.8 * tiles * 256


def linear_swiglu(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """``silu(x @ W_gateᵀ) * (x @ W_upᵀ)`` for a fused ``weight = [W_gate; W_up]``: one kernel at decode size
    (SwiGLU in the epilogue of the weight-streaming GEMM) and at prefill size (SwiGLU in the epilogue of the 256² tile
    GEMM: the ``[T, 2·I]`` intermediate never exists), library GEMM + ``silu_and_mul`` in between."""
    if _skinny_ok(x, weight, 32):
        return hip().skinny_linear(x, weight, None, True)
    if _PLAN.table and x.is_cuda and x.dim() == 2:          # measured at start-up for this shape and row count
        s = _PLAN.splits_for(x.shape[0], weight.shape[0], weight.shape[1], True)
        if s is not None:
            if s and _tile_operand_ok(x, weight):
                return hip().dense_gemm256(x, weight, True, 1, None, True) if s == _TAIL else hip().dense_gemm256(x, weight, True, s)
            return silu_and_mul(torch.nn.functional.linear(x, weight))
    if (x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.is_contiguous()
            and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
            and gemm256_swiglu_ok(x.shape[0], weight.shape[0] // 2, weight.shape[1])
            and x.shape[0] * x.stride(0) * 2 < 2 ** 32 and weight.numel() * 2 < 2 ** 32):
        return hip().dense_gemm256(x, weight, True)
    return silu_and_mul(torch.nn.functional.linear(x, weight))


def rope_and_cache(qkv: torch.Tensor, positions: torch.Tensor, cos_sin: Optional[torch.Tensor],
                   slot_mapping: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                   num_heads: int, num_kv_heads: int, head_dim: int,
                   k_scale: float = 1.0, v_scale: float = 1.0, kv_scales: Optional[torch.Tensor] = None) -> None:
    """``kv_scales``: float32 ``[2, Hkv]`` (row 0 K, row 1 V) per-head scales of an fp8 cache — calibrated values living
    on the device (hipGraph replays read them at run time); the scalar ``k_scale`` / ``v_scale`` apply without it."""
    if _gpu_kernel_ok(qkv):
        if head_dim != 128:
            raise ValueError("the gfx950 rope/cache kernel is specialised for head_dim 128")
        hip().rope_and_cache(qkv, positions, cos_sin, slot_mapping, k_cache, v_cache, num_heads, num_kv_heads,
                             k_scale, v_scale, kv_scales)
    else:
        if kv_scales is not None:
            k_scale, v_scale = kv_scales[0], kv_scales[1]
        ref.rope_and_cache_ref(qkv, positions, cos_sin, slot_mapping, k_cache, v_cache, num_heads, num_kv_heads,
                               head_dim, k_scale, v_scale)


# Mixed steps: prefill tiles next to the decode rows on a side stream (same-box serving A/B: TPOT p50 39.4 → 38.4 ms,
# profiles/bench_r3aj_concurrent_attention_{0,1}.json).  DLI_CONCURRENT_ATTN=0: one after the other.
CONCURRENT_ATTN = os.environ.get("DLI_CONCURRENT_ATTN", "1") != "0"
_ATTN_STREAMS: dict = {}


def _attn_stream(device):
    key = (device.type, device.index)
    if key not in _ATTN_STREAMS:
        _ATTN_STREAMS[key] = torch.cuda.Stream(device=device)
    return _ATTN_STREAMS[key]


def paged_attention(qkv: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, meta: AttnMetadata,
                    num_heads: int, num_kv_heads: int, head_dim: int, scale: float,
                    k_scale: float = 1.0, v_scale: float = 1.0, out: Optional[torch.Tensor] = None,
                    kv_scales: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Attention of this step's tokens over the paged cache (which already holds their K/V).
    ``qkv [T, (Hq+2Hkv)·D]`` — only the q columns are read.  Returns ``[T, Hq·D]``."""
    T = qkv.shape[0]
    q = qkv[:, :num_heads * head_dim]
    if _gpu_kernel_ok(qkv):
        if head_dim != 128:
            raise ValueError("the gfx950 attention kernels are specialised for head_dim 128")
        if out is None:
            out = torch.empty(T, num_heads * head_dim, dtype=qkv.dtype, device=qkv.device)
        lib = hip()
        side = None
        if CONCURRENT_ATTN and meta.num_prefill_tiles > 0 and (meta.decode_items > 0 or meta.num_decode > 0):
            # mixed step: the prefill tiles (MFMA / VALU-bound, a few hundred workgroups) run on a side stream next to the
            # decode rows (HBM-bound) — disjoint rows of `out`; inside a capture the side stream forks from and joins the
            # capturing stream, so the step's graph gets two parallel branches per layer
            main = torch.cuda.current_stream()
            side = _attn_stream(qkv.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                lib.paged_attention_prefill(out, q, k_cache, v_cache, meta.block_tables, meta.seq_lens,
                                            meta.query_start_loc, meta.prefill_tile_seq, meta.prefill_tile_q0,
                                            meta.num_prefill_tiles, num_heads, num_kv_heads, scale, k_scale, v_scale, kv_scales,
                                            meta.counts if meta.counts_on_device else None)
        if meta.decode_items > 0:
            lib.paged_attention_decode_worklist(out, q, k_cache, v_cache, meta.block_tables, meta.seq_lens,
                                                meta.decode_work, meta.decode_red, meta.counts, meta.decode_items,
                                                meta.decode_reds, meta.decode_max_parts, num_heads, num_kv_heads, scale,
                                                meta.decode_workspace, k_scale, v_scale, kv_scales)
        elif meta.num_decode > 0 and meta.decode_work is None:
            lib.paged_attention_decode(out, q, k_cache, v_cache, meta.block_tables, meta.seq_lens, meta.num_decode,
                                       num_heads, num_kv_heads, scale, meta.decode_splits, meta.decode_workspace,
                                       k_scale, v_scale, kv_scales)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        elif meta.num_prefill_tiles > 0:
            lib.paged_attention_prefill(out, q, k_cache, v_cache, meta.block_tables, meta.seq_lens,
                                        meta.query_start_loc, meta.prefill_tile_seq, meta.prefill_tile_q0,
                                        meta.num_prefill_tiles, num_heads, num_kv_heads, scale, k_scale, v_scale, kv_scales,
                                        meta.counts if meta.counts_on_device else None)
        return out
    B = meta.num_decode + meta.num_prefill
    if kv_scales is not None:
        k_scale, v_scale = kv_scales[0], kv_scales[1]
    o = ref.paged_attention_ref(q.reshape(T, num_heads, head_dim), k_cache, v_cache, meta.block_tables[:B],
                                meta.seq_lens[:B], meta.query_start_loc[:B + 1], scale, k_scale, v_scale)
    if out is not None:
        out.copy_(o)
        return out
    return o


def sample(logits: torch.Tensor, temperature: torch.Tensor, top_p: torch.Tensor, top_k: torch.Tensor,
           seeds: torch.Tensor, steps: torch.Tensor, noise: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Token ids ``[B]`` (int64).  ``seeds``/``steps`` (int64) key the per-request RNG stream."""
    B = logits.shape[0]
    if logits.is_cuda:
        if out is None:
            out = torch.empty(B, dtype=torch.int64, device=logits.device)
        hip().sample(out, logits, temperature, top_p, top_k, seeds, steps, noise)
        return out
    if noise is None:
        noise = torch.empty(B, logits.shape[1])
        for i in range(B):          # same contract as the kernel: the draw depends on (seed, step) only
            g = torch.Generator().manual_seed(int(seeds[i]) * 1000003 + int(steps[i]))
            noise[i] = torch.rand(logits.shape[1], generator=g)
        noise.clamp_(1e-10, 1 - 1e-7)
    res = ref.sample_ref(logits, temperature, top_p, top_k.long(), uniform_noise=noise)
    if out is not None:
        out.copy_(res)
        return out
    return res


def gather_sampled(input_ids: torch.Tensor, src: torch.Tensor, sampled: torch.Tensor) -> None:
    """In place: ``input_ids[i] = sampled[src[i]]`` where ``src[i] >= 0`` (pipelined stepping)."""
    if input_ids.is_cuda:
        hip().gather_sampled(input_ids, src, sampled)
    else:
        m = src >= 0
        input_ids[m] = sampled[src[m].long()].to(input_ids.dtype)


def copy_pages(cache: torch.Tensor, pairs: torch.Tensor) -> None:
    """Copy-on-write helper: ``cache[dst] = cache[src]`` for every ``(src, dst)`` row of ``pairs``."""
    if cache.is_cuda:
        hip().copy_pages(cache, pairs)
    else:
        p = pairs.long()
        if cache.dtype == torch.float8_e4m3fn:
            c8 = cache.view(torch.uint8)
            c8[p[:, 1]] = c8[p[:, 0]]
        else:
            cache[p[:, 1]] = cache[p[:, 0]]


def fused_moe(x: torch.Tensor, router_logits: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor, top_k: int,
              expert_offset: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Mixtral MoE block for the experts held by this rank (partial sum; caller all-reduces).  ``out``: result buffer."""
    from . import moe as _moe
    if x.is_cuda and os.environ.get('DLI_MOE_REF') != '1' and _moe.hip_moe_supported(x, w13):
        return _moe.fused_moe_hip(x, router_logits, w13, w2, top_k, expert_offset, out)
    res = ref.moe_ref(x, router_logits, w13, w2, top_k, expert_offset)
    if out is not None:
        out.copy_(res)
        return out
    return res
