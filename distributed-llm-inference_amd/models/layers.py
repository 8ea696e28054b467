"""Tensor-parallel building blocks shared by the model families.

Sharding follows the usual Megatron layout, chosen so that one transformer layer needs exactly two
all-reduces (after the attention output projection and after the MLP/MoE down projection):

* ``ColumnParallelLinear`` – output features split over TP ranks (QKV by heads, gate/up by columns);
* ``RowParallelLinear``    – input features split, partial sums all-reduced over xGMI (RCCL or the
  custom one-/two-shot kernel);
* ``VocabParallelEmbedding`` / ``ParallelLMHead`` – vocabulary split, embedding all-reduced, logits
  all-gathered.

Weights are plain ``[out, in]`` tensors consumed by ``torch.nn.functional.linear`` (hipBLASLt on
ROCm — SURVEY.md §2.5 K11: dense GEMMs stay on the vendor library).  Random initialisation is
deterministic per parameter *name* and generated at full (unsharded) shape before slicing, so a TP=k
model holds exactly the shards of the TP=1 model with the same seed — the property the
TP-equivalence tests rely on.
"""
from __future__ import annotations

import zlib
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..parallel.state import ParallelContext


def _param_seed(base_seed: int, name: str) -> int:
    return (base_seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1)


def random_full_tensor(name: str, shape, dtype, device, seed: int, std: float = 0.02) -> torch.Tensor:
    """The *unsharded* random parameter ``name`` (same values whatever the TP degree)."""
    g = torch.Generator(device=device)
    g.manual_seed(_param_seed(seed, name))
    return (torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)


class Weight(nn.Module):
    """A parameter holder that knows how to cut its shard out of the full tensor."""

    def __init__(self, full_shape, shard_dim: Optional[int], ctx: ParallelContext, dtype, device,
                 segments: Optional[list] = None):
        super().__init__()
        self.full_shape = tuple(full_shape)
        self.shard_dim = shard_dim
        self.ctx = ctx
        # `segments`: sizes of independently sharded pieces along shard_dim (q|k|v or gate|up)
        self.segments = segments or ([self.full_shape[shard_dim]] if shard_dim is not None else None)
        shape = list(self.full_shape)
        if shard_dim is not None:
            for s in self.segments:
                assert s % ctx.tp_size == 0, f"segment {s} not divisible by tp={ctx.tp_size}"
            shape[shard_dim] = sum(s // ctx.tp_size for s in self.segments)
        self.data = nn.Parameter(torch.empty(shape, dtype=dtype, device=device), requires_grad=False)
        self.q: Optional[torch.Tensor] = None              # e4m3 copy + per-output-channel scales (EngineConfig.quantization)
        self.q_scale: Optional[torch.Tensor] = None

    def quantize_fp8(self) -> None:
        assert self.data.dim() == 2
        self.q, self.q_scale = ops.quant_weight_fp8(self.data)

    def shard_of(self, full: torch.Tensor) -> torch.Tensor:
        if self.shard_dim is None or self.ctx.tp_size == 1:
            return full
        parts, start = [], 0
        for s in self.segments:
            per = s // self.ctx.tp_size
            parts.append(full.narrow(self.shard_dim, start + self.ctx.tp_rank * per, per))
            start += s
        return torch.cat(parts, dim=self.shard_dim) if len(parts) > 1 else parts[0]

    def load_full(self, full: torch.Tensor):
        assert tuple(full.shape) == self.full_shape, f"expected {self.full_shape}, got {tuple(full.shape)}"
        self.data.copy_(self.shard_of(full).to(self.data.dtype))

    def init_random(self, name: str, seed: int, std: float = 0.02):
        self.load_full(random_full_tensor(name, self.full_shape, self.data.dtype, self.data.device, seed, std))

    def init_const(self, value: float):
        self.data.fill_(value)


class ColumnParallelLinear(nn.Module):
    def __init__(self, in_features: int, out_segments: list, ctx: ParallelContext, dtype, device, bias: bool = False):
        super().__init__()
        self.weight = Weight((sum(out_segments), in_features), 0, ctx, dtype, device, segments=list(out_segments))
        self.bias = Weight((sum(out_segments),), 0, ctx, dtype, device, segments=list(out_segments)) if bias else None

    def forward(self, x, xq=None):
        w = self.weight
        if xq is not None:                                   # the producer (fused norm + quantisation) has the e4m3 rows already
            return ops.linear_fp8(None, w.q, w.q_scale, xq=xq)
        if w.q is not None and self.bias is None:
            if ops.use_fp8(x):
                return ops.linear_fp8(x, w.q, w.q_scale, xq=xq)
            if ops.skinny_w8_ok(x, w.q):
                return ops.linear_w8(x, w.q, w.q_scale)
        return ops.linear(x, w.data, None if self.bias is None else self.bias.data)


class Partial:
    """A row-parallel result whose reduction is still owed (sequence-parallel steps, ``ParallelContext.sp_step``): the
    layer that consumes it fuses the reduction with its residual add + RMSNorm (``add_norm``).
    ``kind == "reduce"``: ``t`` = this rank's partial sums ``[T, H]`` (in the IPC staging buffer; a plain tensor under
    emulation); ``kind == "rows"``: ``t`` = the finished rows of THIS rank's slice only (expert-parallel dispatch path)."""
    __slots__ = ("kind", "t")

    def __init__(self, kind: str, t: torch.Tensor):
        self.kind, self.t = kind, t


def add_norm(h, residual: torch.Tensor, weight: torch.Tensor, eps: float, ctx: ParallelContext) -> torch.Tensor:
    """``residual += h; return rmsnorm(residual) * weight`` — for a plain tensor the in-place kernel, for a ``Partial`` the
    sequence-parallel form: reduce / add / normalise this rank's rows, then gather all rows."""
    if not isinstance(h, Partial):
        ops.fused_add_rms_norm(h, residual, weight, eps)
        return h
    T = residual.shape[0]
    lo, hi, per = ctx.sp_rows(T)
    ctx.comm["sp_norm_calls"] += 1
    if h.kind == "reduce" and not ctx.emulated:
        ctx.comm["ar_custom_calls"] += 1
        ctx.comm["ar_custom_bytes"] += h.t.numel() * 2
        return ctx.custom_ar.all_reduce_add_norm(h.t, residual, weight, eps)
    if h.kind == "reduce":                                   # emulated rank: a same-bytes stand-in for the reads of the peers'
        rows = ctx.all_reduce(h.t)[lo:hi]                    # partial sums, the norm on this rank's rows, the gather's writes
    else:
        rows = h.t[:hi - lo]
    if hi > lo:
        ops.fused_add_rms_norm(rows, residual[lo:hi], weight, eps)
    if rows.shape[0] < per:
        rows = torch.cat([rows, rows.new_zeros(per - rows.shape[0], rows.shape[1])])
    return ctx.all_gather_rows(rows)[:T]


class RowParallelLinear(nn.Module):
    """``y = all_reduce(x_shard @ W_shardᵀ)``.

    For prefill-sized inputs on a GPU the GEMM and the all-reduce are *pipelined*: the rows are cut
    into chunks, chunk i is reduced over xGMI on a side stream while the GEMM of chunk i+1 runs on the
    compute stream (north-star item 3: "all-reduce over xGMI overlapped with the GEMMs").  The reduction of a
    chunk is whichever stream-ordered collective the group has: the custom IPC kernel (chunks up to its staging
    size; with a launch capped at ``OVERLAP_AR_BLOCKS`` workgroups so that the GEMM keeps most of the CUs) or RCCL
    (``nccl`` device group, chunks past the custom kernel's crossover) — so the pipeline also runs, and is tested, when
    the ranks share one GPU with gloo as the library backend.  Both are plain stream work: inside a hipGraph capture
    the side stream forks from and joins the capturing stream, and the overlap is part of the captured step.
    Decode-sized inputs take the single-kernel all-reduce (latency-bound, nothing to overlap with)."""

    OVERLAP_MIN_ROWS = int(__import__('os').environ.get('DLI_OVERLAP_MIN_ROWS', '1024'))
    OVERLAP_CHUNKS = int(__import__('os').environ.get('DLI_OVERLAP_CHUNKS', '4'))
    OVERLAP_AR_BLOCKS = int(__import__('os').environ.get('DLI_OVERLAP_AR_BLOCKS', '64'))
    STAGED_OUTPUT = __import__('os').environ.get('DLI_AR_STAGED_OUTPUT', '1') != '0'   # A/B: GEMM output → IPC staging buffer
    OVERLAP_OK = True          # cleared by ModelRunner.startup_selftest when the pipeline disagrees with the plain path
    overlap_calls = 0          # forward passes that took the pipeline (tests assert the path was taken)

    def __init__(self, in_features: int, out_features: int, ctx: ParallelContext, dtype, device, bias: bool = False,
                 reduce: bool = True):
        super().__init__()
        self.ctx = ctx
        self.reduce = reduce
        self.weight = Weight((out_features, in_features), 1, ctx, dtype, device)
        self.bias = Weight((out_features,), None, ctx, dtype, device) if bias else None

    def _mm(self, x, b, out=None, xq=None):
        """The local GEMM: W8A8 for a quantised layer at more than 64 rows (``xq``: input quantised by the caller)."""
        w = self.weight
        if w.q is not None and b is None:
            if ops.use_fp8(x):
                return ops.linear_fp8(x, w.q, w.q_scale, out=out, xq=xq)
            if ops.skinny_w8_ok(x, w.q):
                return ops.linear_w8(x, w.q, w.q_scale, out=out)
        return ops.linear(x, w.data, b, out=out)

    def forward(self, x, defer: bool = False, xq=None):
        """``defer`` (sequence-parallel step): return the un-reduced partial sums as a ``Partial`` — written straight into
        the IPC staging buffer — for the consumer's fused reduce + add + norm.  ``xq``: ``x`` already quantised to e4m3
        (``(q, scale)``; only meaningful for a quantised layer)."""
        # the bias must be added once, not tp times: only rank 0 adds it before the reduction
        b = None if (self.bias is None or self.ctx.tp_rank != 0) else self.bias.data
        fp8 = self.weight.q is not None and b is None and ops.use_fp8(x)
        if defer and self.reduce and self.ctx.sp_step:
            ar = self.ctx.custom_ar
            N = self.weight.data.shape[0]
            out = None if self.ctx.emulated else ar.staging_tensor(x.shape[0], N, x)
            return Partial("reduce", self._mm(x, b, out, xq))
        if (self.reduce and self.OVERLAP_OK and self.ctx.distributed and x.is_cuda and x.dim() == 2 and not fp8
                and x.shape[0] >= self.OVERLAP_MIN_ROWS and self.overlap_capable()):
            return self._forward_overlapped(x, b)
        ar = self.ctx.custom_ar
        if (self.reduce and self.STAGED_OUTPUT and ar is not None and x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16
                and self.ctx.distributed and ar.fits(x.shape[0] * self.weight.data.shape[0] * 2)):
            # the GEMM writes straight into the IPC staging buffer: the reduction kernel has no copy-in pass left
            y = self._mm(x, b, ar.staging_tensor(x.shape[0], self.weight.data.shape[0], x), xq)
            c = self.ctx.comm
            c["ar_custom_calls"] += 1
            c["ar_custom_bytes"] += y.numel() * 2
            return ar.all_reduce(y)
        y = self._mm(x, b, None, xq)
        return self.ctx.all_reduce(y) if self.reduce else y

    def _library_is_stream_ordered(self) -> bool:
        """RCCL is; gloo moves CUDA tensors through host buffers on its own threads (rank-divergent sums under skew in
        the single-GPU rehearsal mode), so a gloo group pipelines through the custom kernels only."""
        if not hasattr(self, "_nccl"):
            import torch.distributed as dist
            self._nccl = self.ctx.group is not None and dist.get_backend(self.ctx.group) == "nccl"
        return self._nccl

    def overlap_capable(self) -> bool:
        ar = self.ctx.custom_ar
        return self.ctx.distributed and (self._library_is_stream_ordered() or (ar is not None and ar.enabled))

    def _chunk_rows(self, T: int, row_bytes: int) -> int:
        step = (T + self.OVERLAP_CHUNKS - 1) // self.OVERLAP_CHUNKS
        step = (step + 255) // 256 * 256                       # whole MFMA macro-tiles per chunk
        ar = self.ctx.custom_ar
        if ar is not None and ar.enabled and not self._library_is_stream_ordered():
            step = min(step, max(ar.max_bytes // row_bytes // 256 * 256, 256))   # every chunk must fit the IPC staging buffer
        return step

    def _forward_overlapped(self, x, b):
        import torch.distributed as dist
        type(self).overlap_calls += 1
        T = x.shape[0]
        N = self.weight.data.shape[0]
        step = self._chunk_rows(T, N * x.element_size())
        y = torch.empty(T, N, dtype=x.dtype, device=x.device)      # partial sums (GEMM output)
        ar = self.ctx.custom_ar
        use_custom = ar is not None and ar.enabled
        z = torch.empty_like(y) if use_custom else y               # the custom kernel reduces out of place
        main = torch.cuda.current_stream()
        side = _comm_stream(x.device)
        c = self.ctx.comm
        done = []
        for lo in range(0, T, step):
            hi = min(lo + step, T)
            torch.mm(x[lo:hi], self.weight.data.t(), out=y[lo:hi]) if b is None else \
                torch.addmm(b, x[lo:hi], self.weight.data.t(), out=y[lo:hi])
            ready = torch.cuda.Event()
            ready.record(main)
            nbytes = (hi - lo) * N * y.element_size()
            with torch.cuda.stream(side):
                side.wait_event(ready)
                if use_custom and (ar.should_use(y[lo:hi]) or not self._library_is_stream_ordered()):
                    c["ar_custom_calls"] += 1
                    c["ar_custom_bytes"] += nbytes
                    ar.all_reduce(y[lo:hi], blocks=self.OVERLAP_AR_BLOCKS, out=z[lo:hi])   # IPC kernel, GEMM i+1 meanwhile
                else:
                    c["ar_lib_calls"] += 1
                    c["ar_lib_bytes"] += nbytes
                    dist.all_reduce(y[lo:hi], group=self.ctx.group)    # RCCL on the comm stream, GEMM i+1 meanwhile
                    if z is not y:
                        z[lo:hi].copy_(y[lo:hi])
                ev = torch.cuda.Event()
                ev.record(side)
                done.append(ev)
        for ev in done:
            main.wait_event(ev)
        return z


_COMM_STREAMS: dict = {}


def _comm_stream(device):
    key = (device.type, device.index)
    if key not in _COMM_STREAMS:
        _COMM_STREAMS[key] = torch.cuda.Stream(device=device)
    return _COMM_STREAMS[key]


class ReplicatedLinear(nn.Module):
    def __init__(self, in_features: int, out_features: int, ctx: ParallelContext, dtype, device, bias: bool = False):
        super().__init__()
        self.weight = Weight((out_features, in_features), None, ctx, dtype, device)
        self.bias = Weight((out_features,), None, ctx, dtype, device) if bias else None

    def forward(self, x):
        return ops.linear(x, self.weight.data, None if self.bias is None else self.bias.data)


class VocabParallelEmbedding(nn.Module):
    def __init__(self, vocab_size: int, hidden: int, ctx: ParallelContext, dtype, device):
        super().__init__()
        self.ctx = ctx
        self.vocab_padded = (vocab_size + ctx.tp_size - 1) // ctx.tp_size * ctx.tp_size
        self.vocab_size = vocab_size
        self.weight = Weight((self.vocab_padded, hidden), 0, ctx, dtype, device)
        self.per = self.vocab_padded // ctx.tp_size

    def forward(self, ids: torch.Tensor):
        if self.ctx.solo:
            return F.embedding(ids, self.weight.data)
        lo = self.ctx.tp_rank * self.per
        local = ids - lo
        mine = (local >= 0) & (local < self.per)
        out = F.embedding(torch.where(mine, local, torch.zeros_like(local)), self.weight.data)
        out = out * mine.unsqueeze(-1).to(out.dtype)
        return self.ctx.all_reduce(out)


class ParallelLMHead(nn.Module):
    """Logits over the full vocabulary: local ``[n, V/tp]`` GEMM, then an all-gather."""

    def __init__(self, embedding: VocabParallelEmbedding, tied: bool, hidden: int, ctx: ParallelContext, dtype, device):
        super().__init__()
        self.ctx = ctx
        self.vocab_size = embedding.vocab_size
        self.tied = tied
        self._embedding = [embedding]          # not registered twice when tied
        self.weight = None if tied else Weight((embedding.vocab_padded, hidden), 0, ctx, dtype, device)

    def forward(self, h: torch.Tensor) -> torch.Tensor:
        w = self._embedding[0].weight.data if self.tied else self.weight.data
        logits = self.ctx.all_gather_last_dim(ops.linear(h, w))
        return logits[..., :self.vocab_size] if logits.shape[-1] != self.vocab_size else logits
