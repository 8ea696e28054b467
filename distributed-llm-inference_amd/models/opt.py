"""OPT (125m) — the CPU "plumbing" model of BASELINE.json config 1.

Pre-LayerNorm decoder with learned positions (offset 2), biased linears, ReLU MLP and a tied LM head
(public OPT-125m shapes).  Runs through the same engine, scheduler and paged cache as
the large models; on CPU every op resolves to its torch reference.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..config import ModelConfig
from ..parallel.state import ParallelContext
from .layers import ColumnParallelLinear, ParallelLMHead, RowParallelLinear, VocabParallelEmbedding, Weight
from .llama import Attention


class OPTLayer(nn.Module):
    def __init__(self, cfg: ModelConfig, ctx: ParallelContext, dtype, device):
        super().__init__()
        H = cfg.hidden_size
        self.ln1_w, self.ln1_b = Weight((H,), None, ctx, dtype, device), Weight((H,), None, ctx, dtype, device)
        self.ln2_w, self.ln2_b = Weight((H,), None, ctx, dtype, device), Weight((H,), None, ctx, dtype, device)
        self.attn = Attention(cfg, ctx, dtype, device, bias=True, rope=False)
        self.fc1 = ColumnParallelLinear(H, [cfg.intermediate_size], ctx, dtype, device, bias=True)
        self.fc2 = RowParallelLinear(cfg.intermediate_size, H, ctx, dtype, device, bias=True)

    def forward(self, h, meta, kv):
        x = F.layer_norm(h, (h.shape[-1],), self.ln1_w.data, self.ln1_b.data)
        h = h + self.attn(x, meta, kv, None)
        x = F.layer_norm(h, (h.shape[-1],), self.ln2_w.data, self.ln2_b.data)
        return h + self.fc2(F.relu(self.fc1(x)))


class OPTForCausalLM(nn.Module):
    def __init__(self, cfg: ModelConfig, ctx: ParallelContext, dtype, device):
        super().__init__()
        self.cfg, self.ctx = cfg, ctx
        H = cfg.hidden_size
        self.embed = VocabParallelEmbedding(cfg.vocab_size, H, ctx, dtype, device)
        self.pos = Weight((cfg.max_position + cfg.pos_offset, H), None, ctx, dtype, device)
        self.layers = nn.ModuleList(OPTLayer(cfg, ctx, dtype, device) for _ in range(cfg.num_layers))
        self.final_w, self.final_b = Weight((H,), None, ctx, dtype, device), Weight((H,), None, ctx, dtype, device)
        self.lm_head = ParallelLMHead(self.embed, True, H, ctx, dtype, device)

    @property
    def local_kv_heads(self) -> int:
        return self.cfg.num_kv_heads // self.ctx.tp_size

    def forward(self, input_ids, meta, kv_caches):
        h = self.embed(input_ids) + F.embedding(meta.positions.long() + self.cfg.pos_offset, self.pos.data)
        for layer, kv in zip(self.layers, kv_caches):
            h = layer(h, meta, kv)
        return F.layer_norm(h, (h.shape[-1],), self.final_w.data, self.final_b.data)

    def compute_logits(self, h):
        return self.lm_head(h)

    def named_weights(self):
        yield "model.decoder.embed_tokens.weight", self.embed.weight, "normal"
        yield "model.decoder.embed_positions.weight", self.pos, "normal"
        for i, l in enumerate(self.layers):
            p = f"model.decoder.layers.{i}."
            yield p + "self_attn_layer_norm.weight", l.ln1_w, "ones"
            yield p + "self_attn_layer_norm.bias", l.ln1_b, "zeros"
            yield p + "final_layer_norm.weight", l.ln2_w, "ones"
            yield p + "final_layer_norm.bias", l.ln2_b, "zeros"
            yield p + "self_attn.qkv_proj.weight", l.attn.qkv.weight, "normal"
            yield p + "self_attn.qkv_proj.bias", l.attn.qkv.bias, "zeros"
            yield p + "self_attn.out_proj.weight", l.attn.o.weight, "normal"
            yield p + "self_attn.out_proj.bias", l.attn.o.bias, "zeros"
            yield p + "fc1.weight", l.fc1.weight, "normal"
            yield p + "fc1.bias", l.fc1.bias, "zeros"
            yield p + "fc2.weight", l.fc2.weight, "normal"
            yield p + "fc2.bias", l.fc2.bias, "zeros"
        yield "model.decoder.final_layer_norm.weight", self.final_w, "ones"
        yield "model.decoder.final_layer_norm.bias", self.final_b, "zeros"
