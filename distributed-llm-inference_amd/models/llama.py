"""Llama-3 (8B / 70B) and Mixtral-8x7B decoder stacks over the paged KV cache.

Per layer:  fused_add_rmsnorm → QKV GEMM → rope+cache write → paged attention →
O GEMM (+all-reduce) → fused_add_rmsnorm → gate/up GEMM → SwiGLU → down GEMM (+all-reduce); Mixtral
replaces the dense MLP by the top-2 MoE block.  Everything between the GEMMs is a hand-written HIP
kernel (``dli_amd.ops``); the GEMMs are hipBLASLt through torch.

Expert parallelism (``ctx.ep_size == tp``): each rank owns ``E/tp`` whole experts.  Two dispatch modes
(SURVEY.md §2.5 K8 "EP dispatch/combine via all-to-all (or all-gather + local mask at small T)"):

* *replicated* (decode-sized steps, hipGraph-capturable): activations are already replicated across the
  TP group after the attention all-reduce, so every rank evaluates its own experts on the tokens routed
  to them and one all-reduce of ``[T, H]`` sums the partial outputs;
* *all-to-all* (steps of ``ctx.ep_a2a_min_tokens`` tokens or more): each rank routes only its ``T/ep``
  slice of the tokens (routing kernel), every (token, expert) row travels to the rank that owns the expert,
  the grouped GEMM runs on what arrived, the expert outputs travel back and are combined; an all-gather of
  the ``T/ep`` result rows replicates the block output.  Per rank that moves ``≈ 2·top_k/ep + 1`` messages of
  ``T·H/ep`` rows' worth instead of the ``≈ 2·(ep-1)/ep · T·H`` of the all-reduce — 3.5× less over xGMI at
  ep = 8, top-2.  The exchange runs on the IPC-mapped peer buffers of the custom collectives
  (``csrc/kernels/custom_all_reduce.hip``: ``ep_plan / ep_pack / ep_exchange / ep_combine``) with the
  per-destination counts kept in device memory: no host synchronisation, capturable in the step's hipGraph.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .. import ops
from ..config import ModelConfig
from ..ops.reference import attention_scale, build_cos_sin_cache
from ..parallel.state import ParallelContext
from .layers import (ColumnParallelLinear, ParallelLMHead, Partial, ReplicatedLinear, RowParallelLinear,
                     VocabParallelEmbedding, Weight, add_norm)


_EP_DEBUG = __import__("os").environ.get("DLI_DEBUG_EP_CHECK") == "1"


def _rank_check(ctx, tag, t):
    """Debug aid (DLI_DEBUG_EP_CHECK=1): replicated activations must be bit-identical on every rank."""
    import torch.distributed as dist
    if not (_EP_DEBUG and ctx.tp_size > 1) or torch.cuda.is_current_stream_capturing():
        return
    rows = t.reshape(t.shape[0], -1).view(torch.int16).to(torch.int64).sum(-1).cpu()
    got = [None] * ctx.tp_size
    dist.all_gather_object(got, rows, group=ctx.cpu_group if ctx.cpu_group is not None else ctx.group)
    if ctx.tp_rank == 0:
        bad = (got[0] != got[1]).nonzero().flatten().tolist()
        if bad:
            print(f"[ep-check] {tag} T={t.shape[0]}: {len(bad)} rows differ, first {bad[:6]} last {bad[-3:]}", flush=True)


def _ep_consistency_check(ctx, h, logits):
    _rank_check(ctx, "moe-input", h)
    _rank_check(ctx, "router-logits", logits)


FUSE_NORM_QUANT = os.environ.get("DLI_FUSE_NORM_QUANT", "1") != "0"     # A/B knob of the W8A8 option


class Attention(nn.Module):
    def __init__(self, cfg: ModelConfig, ctx: ParallelContext, dtype, device, bias: bool = False, rope: bool = True):
        super().__init__()
        tp = ctx.tp_size
        assert cfg.num_heads % tp == 0 and cfg.num_kv_heads % tp == 0, \
            f"tp={tp} must divide heads ({cfg.num_heads}) and kv heads ({cfg.num_kv_heads})"
        self.hq, self.hkv, self.d = cfg.num_heads // tp, cfg.num_kv_heads // tp, cfg.head_dim
        self.qkv = ColumnParallelLinear(cfg.hidden_size,
                                        [cfg.num_heads * self.d, cfg.num_kv_heads * self.d, cfg.num_kv_heads * self.d],
                                        ctx, dtype, device, bias=bias)
        self.o = RowParallelLinear(cfg.num_heads * self.d, cfg.hidden_size, ctx, dtype, device, bias=bias)
        self.scale = attention_scale(self.d)
        self.rope = rope
        self.k_scale = 1.0
        self.v_scale = 1.0
        # fp8 KV cache: per-(layer, kv head) scales [2, Hkv] (K row, V row) on the device, set by the warm-up calibration
        # (ModelRunner.calibrate_kv_scales); the kernels read them at run time, so decode graphs see updates
        self.kv_scales = None
        self.calibrating = False
        self.measuring = False            # held-out pass: count the values the calibrated scales would clip
        self._calibrated = False
        self.sat_count = self.sat_total = None

    def enable_kv_scales(self, device):
        self.kv_scales = torch.ones(2, self.hkv, dtype=torch.float32, device=device)

    @torch.no_grad()
    def _observe(self, qkv):
        """Calibration pass: running per-head amax of K (rotation-invariant bound: the norm of each rotated pair) and V
        of this step's tokens → scale = 1.2 · amax / 448 (e4m3 max), before they are written to the cache."""
        T, hq, hkv, d = qkv.shape[0], self.hq, self.hkv, self.d
        k = qkv[:, hq * d:(hq + hkv) * d].float().view(T, hkv, 2, d // 2)
        v = qkv[:, (hq + hkv) * d:].float().view(T, hkv, d)
        amax = torch.stack([k.pow(2).sum(2).sqrt().amax((0, 2)), v.abs().amax((0, 2))])
        new = amax.clamp_min(1e-4) * (1.2 / 448.0)
        self.kv_scales.copy_(torch.maximum(self.kv_scales, new) if self._calibrated else new)
        self._calibrated = True

    @torch.no_grad()
    def _measure(self, qkv):
        """Held-out pass after calibration: how many K / V elements exceed the e4m3 range under the current scales
        (they are clamped to ±448 by the cache write)."""
        T, hq, hkv, d = qkv.shape[0], self.hq, self.hkv, self.d
        k = qkv[:, hq * d:(hq + hkv) * d].float().view(T, hkv, 2, d // 2).pow(2).sum(2).sqrt()
        v = qkv[:, (hq + hkv) * d:].float().view(T, hkv, d).abs()
        lim = self.kv_scales * 448.0
        n = (k > lim[0].view(1, hkv, 1)).sum() + (v > lim[1].view(1, hkv, 1)).sum()
        self.sat_count = n if self.sat_count is None else self.sat_count + n
        self.sat_total = (self.sat_total or 0) + k.numel() + v.numel()

    def forward(self, h, meta, kv, cos_sin, xq=None):
        """``xq``: the layer input already quantised to e4m3 by the fused norm (W8A8 option); ``h`` is then not read."""
        qkv = self.qkv(h, xq=xq)
        k_cache, v_cache = kv
        if self.calibrating and self.kv_scales is not None:
            self._observe(qkv)
        elif self.measuring and self.kv_scales is not None:
            self._measure(qkv)
        ops.rope_and_cache(qkv, meta.positions, cos_sin if self.rope else None, meta.slot_mapping, k_cache, v_cache,
                           self.hq, self.hkv, self.d, self.k_scale, self.v_scale, self.kv_scales)
        a = ops.paged_attention(qkv, k_cache, v_cache, meta, self.hq, self.hkv, self.d, self.scale,
                                self.k_scale, self.v_scale, kv_scales=self.kv_scales)
        return self.o(a, defer=True)


class SwiGLUMLP(nn.Module):
    def __init__(self, cfg: ModelConfig, ctx: ParallelContext, dtype, device):
        super().__init__()
        self.gate_up = ColumnParallelLinear(cfg.hidden_size, [cfg.intermediate_size, cfg.intermediate_size],
                                            ctx, dtype, device)
        self.down = RowParallelLinear(cfg.intermediate_size, cfg.hidden_size, ctx, dtype, device)

    def forward(self, h, xq=None):
        w13 = self.gate_up.weight
        if xq is not None or (w13.q is not None and ops.use_fp8(h)):
            # W8A8: gate/up on the fp8 matrix cores, then ONE pass that applies SwiGLU and quantises the down projection's input
            gu = ops.linear_fp8(h, w13.q, w13.q_scale, xq=xq)
            return self.down(gu, defer=True, xq=ops.quant_rows_fp8(gu, silu=True))
        if w13.q is not None and ops.skinny_w8_ok(h, w13.q, 32):
            return self.down(ops.linear_w8(h, w13.q, w13.q_scale, swiglu=True), defer=True)     # decode: e4m3 weight stream
        return self.down(ops.linear_swiglu(h, w13.data), defer=True)


class MoEBlock(nn.Module):
    """Mixtral sparse MLP: router (replicated) + experts, expert- or tensor-parallel."""

    def __init__(self, cfg: ModelConfig, ctx: ParallelContext, dtype, device):
        super().__init__()
        self.ctx = ctx
        self.top_k = cfg.top_k_experts
        self.num_experts = cfg.num_experts
        self.router = ReplicatedLinear(cfg.hidden_size, cfg.num_experts, ctx, dtype, device)
        E, I, H = cfg.num_experts, cfg.intermediate_size, cfg.hidden_size
        if ctx.ep_size > 1:
            assert E % ctx.tp_size == 0
            self.local_experts = E // ctx.tp_size
            self.expert_offset = ctx.tp_rank * self.local_experts
            self.w13 = Weight((E, 2 * I, H), 0, ctx, dtype, device)                  # whole experts per rank
            self.w2 = Weight((E, H, I), 0, ctx, dtype, device)
        else:
            self.local_experts, self.expert_offset = E, 0
            self.w13 = Weight((E, 2 * I, H), 1, ctx, dtype, device, segments=[I, I])  # gate|up columns split
            self.w2 = Weight((E, H, I), 2, ctx, dtype, device)

    def forward(self, h):
        ctx = self.ctx
        if ctx.ep_size > 1 and 0 < ctx.ep_a2a_min_tokens <= h.shape[0] and self._a2a_available(h):
            return self._forward_all_to_all(h)
        logits = self.router(h)
        if _EP_DEBUG and self.ctx.tp_size > 1 and not torch.cuda.is_current_stream_capturing():
            _ep_consistency_check(self.ctx, h, logits)
        if ctx.sp_step:                                      # the consumer's fused reduce + add + norm takes the partial sums
            stage = None if ctx.emulated else ctx.custom_ar.staging_tensor(h.shape[0], h.shape[1], h)
            return Partial("reduce", ops.fused_moe(h, logits, self.w13.data, self.w2.data, self.top_k, self.expert_offset, out=stage))
        out = ops.fused_moe(h, logits, self.w13.data, self.w2.data, self.top_k, self.expert_offset)
        return self.ctx.all_reduce(out)

    def _a2a_available(self, h) -> bool:
        """GPU: the dispatch runs on the custom IPC kernels (device-side counts, capturable) — without them the block
        stays on the replicated path, which needs no host either.  CPU (tests): the torch reference of the same algorithm."""
        if not h.is_cuda:
            return self.ctx.distributed
        ar = self.ctx.custom_ar
        return (ar is not None and ar.enabled and h.dtype == torch.bfloat16 and h.shape[1] % 128 == 0
                and ar.ep_tokens_per_call(h.shape[1], self.top_k) >= self.ctx.tp_size * 8)

    def _forward_all_to_all(self, h):
        """Token-sliced routing + dispatch / combine (see the module docstring): each rank routes its ``T / ep`` rows with
        the routing kernel, the IPC dispatch kernels move every (token, expert) row to the expert's owner and the outputs
        back — the per-destination counts never leave the device (no ``.cpu()``, no gloo, no one-hot
        detour in here) — and an all-gather of the result rows replicates the block output.  A step larger than the IPC
        buffers' worst case goes through in several calls."""
        if not h.is_cuda:
            return self._forward_all_to_all_reference(h)
        ctx, ep, k = self.ctx, self.ctx.tp_size, self.top_k
        ar = ctx.custom_ar
        T, H = h.shape
        step = ar.ep_tokens_per_call(H, k)
        outs = []
        for t0 in range(0, T, step):
            hp = h[t0:t0 + step]
            Tp = hp.shape[0]
            per = (Tp + ep - 1) // ep
            lo = min(ctx.tp_rank * per, Tp)
            hl = hp[lo:min(lo + per, Tp)]
            out_l = ar.ep_moe(hl, self.router(hl), self.w13.data, self.w2.data, k, per * k)
            ctx.comm["a2a_calls"] += 2
            ctx.comm["a2a_bytes"] += 2 * hl.shape[0] * k * H * hl.element_size()
            if ctx.sp_step and step >= T:                                    # one call covers the step: the rows stay local,
                return Partial("rows", out_l)                                # the consumer normalises them, then gathers
            if out_l.shape[0] < per:                                         # the last rank's slice may be short
                out_l = torch.cat([out_l, out_l.new_zeros(per - out_l.shape[0], H)])
            outs.append(ctx.all_gather_rows(out_l)[:Tp])
        full = outs[0] if len(outs) == 1 else torch.cat(outs)
        if ctx.sp_step:                                                      # several calls: hand over this rank's rows of the result
            lo, hi, _ = ctx.sp_rows(T)
            return Partial("rows", full[lo:hi].contiguous())
        return full

    def _forward_all_to_all_reference(self, h):
        """CPU reference of the dispatch (gloo; the multi-process CPU tests): same slicing, same owner computation, the
        row exchange through ``dist.all_to_all_single`` with host-side split sizes."""
        import torch.distributed as dist
        ctx, ep, k = self.ctx, self.ctx.tp_size, self.top_k
        T, H = h.shape
        per = (T + ep - 1) // ep
        lo = min(ctx.tp_rank * per, T)
        hl = h[lo:min(lo + per, T)]
        probs = torch.softmax(self.router(hl).float(), dim=-1)
        topv, topi = probs.topk(k, dim=-1)                                   # [n_l, k]
        topv = topv / topv.sum(-1, keepdim=True)
        flat_e = topi.reshape(-1)
        dest = torch.div(flat_e, self.local_experts, rounding_mode="floor")  # owner rank of every (token, k) row
        order = torch.sort(dest, stable=True).indices
        send_counts = torch.bincount(dest, minlength=ep)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=ctx.cpu_group if ctx.cpu_group is not None else ctx.group)
        send_n, recv_n = send_counts.tolist(), recv_counts.tolist()
        tok = torch.div(order, k, rounding_mode="floor")
        x_recv = ctx.all_to_all(hl.index_select(0, tok), recv_n, send_n)     # rows for my experts, grouped by source
        e_recv = ctx.all_to_all(flat_e.index_select(0, order).to(torch.int32), recv_n, send_n)
        R = x_recv.shape[0]
        if R:
            onehot = torch.zeros(R, self.num_experts, dtype=h.dtype)
            onehot.scatter_(1, e_recv.long().unsqueeze(1), 8.0)              # top-1 of a one-hot row = the chosen expert, weight 1
            y = ops.fused_moe(x_recv, onehot, self.w13.data, self.w2.data, 1, self.expert_offset)
        else:
            y = x_recv
        y_back = ctx.all_to_all(y, send_n, recv_n)                            # same order as the rows I sent
        out_l = torch.zeros(per, H, dtype=torch.float32)
        out_l.index_add_(0, tok, y_back.float() * topv.reshape(-1).index_select(0, order).unsqueeze(1))
        return ctx.all_gather_rows(out_l.to(h.dtype))[:T]


class DecoderLayer(nn.Module):
    def __init__(self, cfg: ModelConfig, ctx: ParallelContext, dtype, device):
        super().__init__()
        H = cfg.hidden_size
        self.eps = cfg.rms_eps
        self.input_norm = Weight((H,), None, ctx, dtype, device)
        self.post_norm = Weight((H,), None, ctx, dtype, device)
        self.attn = Attention(cfg, ctx, dtype, device)
        self.mlp = MoEBlock(cfg, ctx, dtype, device) if cfg.num_experts > 0 else SwiGLUMLP(cfg, ctx, dtype, device)

    def forward(self, h, residual, meta, kv, cos_sin):
        ctx = self.attn.o.ctx
        # W8A8 option, dense layers, steps that take the fp8 GEMMs: the norms hand their rows to the projections as e4m3
        # (one kernel: add + norm + per-row quantisation) — unless the input is an un-reduced partial sum (sequence-parallel
        # steps fuse the norm with the reduction instead)
        fuse_q = self.fuse_norm_quant and not isinstance(h, Partial) and ops.use_fp8(h) and h.shape[1] <= 8192
        if residual is None:
            residual = h
            if fuse_q:
                h = self.attn(None, meta, kv, cos_sin, xq=ops.rms_norm_quant(h, None, self.input_norm.data, self.eps))
            else:
                h = self.attn(ops.rms_norm(h, self.input_norm.data, self.eps), meta, kv, cos_sin)
        elif fuse_q:
            h = self.attn(None, meta, kv, cos_sin, xq=ops.rms_norm_quant(h, residual, self.input_norm.data, self.eps))
        else:
            h = add_norm(h, residual, self.input_norm.data, self.eps, ctx)   # h: the previous layer's MLP output, maybe un-reduced
            h = self.attn(h, meta, kv, cos_sin)
        if not isinstance(h, Partial):
            _rank_check(ctx, "attn-out", h)
            if fuse_q:
                return self.mlp(None, xq=ops.rms_norm_quant(h, residual, self.post_norm.data, self.eps)), residual
        h = add_norm(h, residual, self.post_norm.data, self.eps, ctx)
        h = self.mlp(h)
        return h, residual

    @property
    def fuse_norm_quant(self) -> bool:
        return (FUSE_NORM_QUANT and isinstance(self.mlp, SwiGLUMLP) and self.attn.qkv.weight.q is not None
                and self.attn.qkv.bias is None)


class LlamaForCausalLM(nn.Module):
    """Also serves Mixtral (``cfg.num_experts > 0``)."""

    def __init__(self, cfg: ModelConfig, ctx: ParallelContext, dtype, device):
        super().__init__()
        self.cfg, self.ctx = cfg, ctx
        self.embed = VocabParallelEmbedding(cfg.vocab_size, cfg.hidden_size, ctx, dtype, device)
        self.layers = nn.ModuleList(DecoderLayer(cfg, ctx, dtype, device) for _ in range(cfg.num_layers))
        self.final_norm = Weight((cfg.hidden_size,), None, ctx, dtype, device)
        self.lm_head = ParallelLMHead(self.embed, cfg.tie_embeddings, cfg.hidden_size, ctx, dtype, device)
        self.register_buffer("cos_sin", build_cos_sin_cache(cfg.max_position, cfg.head_dim, cfg.rope_theta, device),
                             persistent=False)

    @property
    def local_kv_heads(self) -> int:
        return self.cfg.num_kv_heads // self.ctx.tp_size

    def forward(self, input_ids: torch.Tensor, meta, kv_caches) -> torch.Tensor:
        h = self.embed(input_ids)
        _rank_check(self.ctx, "embed", h)
        # one decision for the whole pass: with sequence-parallel norms every rank keeps the residual of its own rows only
        self.ctx.sp_step = self.ctx.sp_wanted(h.shape[0], h.shape[1], h.is_cuda) and not self._overlap_takes(h.shape[0])
        residual = None
        for layer, kv in zip(self.layers, kv_caches):
            h, residual = layer(h, residual, meta, kv, self.cos_sin)
        h = add_norm(h, residual, self.final_norm.data, self.cfg.rms_eps, self.ctx)
        self.ctx.sp_step = False
        return h

    def _overlap_takes(self, T: int) -> bool:
        """Steps the GEMM / all-reduce pipeline serves (it won the start-up timing on this system) keep the plain norms."""
        return RowParallelLinear.OVERLAP_OK and self.ctx.distributed and T >= RowParallelLinear.OVERLAP_MIN_ROWS

    def compute_logits(self, h: torch.Tensor) -> torch.Tensor:
        return self.lm_head(h)

    # ---- parameter naming (HF checkpoint names → modules) used by init and the safetensors loader
    @torch.no_grad()
    def quantize_fp8(self) -> int:
        """W8A8 (``EngineConfig.quantization = "fp8"``): an e4m3 copy of every dense projection of the decoder layers (qkv, o,
        gate/up, down) next to the bf16 one; embedding, LM head, norms, routers and MoE experts stay bf16.  Returns the
        number of quantised matrices."""
        n = 0
        for l in self.layers:
            ws = [l.attn.qkv.weight, l.attn.o.weight]
            if isinstance(l.mlp, SwiGLUMLP):
                ws += [l.mlp.gate_up.weight, l.mlp.down.weight]
            for w in ws:
                w.quantize_fp8()
                n += 1
        return n

    def named_weights(self):
        yield "model.embed_tokens.weight", self.embed.weight, "normal"
        for i, l in enumerate(self.layers):
            p = f"model.layers.{i}."
            yield p + "input_layernorm.weight", l.input_norm, "ones"
            yield p + "post_attention_layernorm.weight", l.post_norm, "ones"
            yield p + "self_attn.qkv_proj.weight", l.attn.qkv.weight, "normal"
            yield p + "self_attn.o_proj.weight", l.attn.o.weight, "normal"
            if isinstance(l.mlp, MoEBlock):
                yield p + "block_sparse_moe.gate.weight", l.mlp.router.weight, "normal"
                yield p + "block_sparse_moe.experts.w13", l.mlp.w13, "normal"
                yield p + "block_sparse_moe.experts.w2", l.mlp.w2, "normal"
            else:
                yield p + "mlp.gate_up_proj.weight", l.mlp.gate_up.weight, "normal"
                yield p + "mlp.down_proj.weight", l.mlp.down.weight, "normal"
        yield "model.norm.weight", self.final_norm, "ones"
        if self.lm_head.weight is not None:
            yield "lm_head.weight", self.lm_head.weight, "normal"
