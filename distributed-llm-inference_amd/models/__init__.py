"""Model registry, random initialisation and safetensors checkpoint I/O.

The reference ships no model code or weights; the north star asks for random-init
weights of the named architectures plus a safetensors loader with TP/EP sharding.  Checkpoints use
the public Hugging Face tensor names (separate q/k/v, gate/up and per-expert w1/w3/w2 tensors); the
fused parameters of this stack are assembled from / split into those on the fly.
"""
from __future__ import annotations

import glob
import os
from typing import Optional

import torch

from ..config import ModelConfig
from ..parallel.state import SINGLE, ParallelContext
from .llama import LlamaForCausalLM
from .opt import OPTForCausalLM

_ARCH = {"llama": LlamaForCausalLM, "mixtral": LlamaForCausalLM, "opt": OPTForCausalLM}


def torch_dtype(name: str) -> torch.dtype:
    return {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}[name]


def build_model(cfg: ModelConfig, ctx: Optional[ParallelContext] = None, device="cpu", dtype: Optional[torch.dtype] = None,
                seed: int = 0, weights_path: Optional[str] = None, quantization: Optional[str] = None):
    ctx = ctx or SINGLE
    dtype = dtype or torch_dtype(cfg.dtype)
    device = torch.device(device)
    with torch.no_grad():
        model = _ARCH[cfg.arch](cfg, ctx, dtype, device)
        if weights_path:
            load_safetensors(model, weights_path)
        else:
            init_random(model, seed)
        if quantization:
            if quantization != "fp8" or not hasattr(model, "quantize_fp8"):
                raise ValueError(f"quantization {quantization!r} is not available for {cfg.arch} models (supported: fp8 on llama / mixtral)")
            model.quantize_fp8()
    return model.eval()


@torch.no_grad()
def init_random(model, seed: int = 0) -> None:
    for name, w, kind in model.named_weights():
        if kind == "ones":
            w.init_const(1.0)
        elif kind == "zeros":
            w.init_const(0.0)
        else:
            w.init_random(name, seed)


# ------------------------------------------------------------------ HF name mapping
def _hf_sources(name: str, cfg: ModelConfig):
    """Our fused parameter name → (list of HF tensor names, cat dim) or a per-expert stacking recipe."""
    if name.endswith("self_attn.qkv_proj.weight") or name.endswith("self_attn.qkv_proj.bias"):
        base, kind = name.rsplit("qkv_proj.", 1)
        return [base + f"{p}_proj.{kind}" for p in "qkv"], 0
    if name.endswith("mlp.gate_up_proj.weight"):
        base = name[:-len("gate_up_proj.weight")]
        return [base + "gate_proj.weight", base + "up_proj.weight"], 0
    return [name], 0


def _expert_sources(name: str, cfg: ModelConfig):
    base = name.rsplit("experts.", 1)[0] + "experts."
    if name.endswith("w13"):
        return [[base + f"{e}.w1.weight", base + f"{e}.w3.weight"] for e in range(cfg.num_experts)]
    return [[base + f"{e}.w2.weight"] for e in range(cfg.num_experts)]


class _Shards:
    """Lazy view over one or many ``*.safetensors`` files."""

    def __init__(self, path: str):
        from safetensors import safe_open
        files = sorted(glob.glob(os.path.join(path, "*.safetensors"))) if os.path.isdir(path) else [path]
        if not files:
            raise FileNotFoundError(f"no safetensors files under {path}")
        self.handles = [safe_open(f, framework="pt", device="cpu") for f in files]
        self.where = {k: h for h in self.handles for k in h.keys()}

    def get(self, key: str) -> torch.Tensor:
        if key not in self.where:
            raise KeyError(f"tensor {key!r} not found in checkpoint")
        return self.where[key].get_tensor(key)


@torch.no_grad()
def load_safetensors(model, path: str) -> None:
    cfg = model.cfg
    shards = _Shards(path)
    for name, w, _kind in model.named_weights():
        if ".experts.w" in name:
            full = torch.stack([torch.cat([shards.get(k) for k in ks], dim=0) for ks in _expert_sources(name, cfg)])
        else:
            keys, dim = _hf_sources(name, cfg)
            full = torch.cat([shards.get(k) for k in keys], dim=dim) if len(keys) > 1 else shards.get(keys[0])
        if full.shape[0] != w.full_shape[0] and name.endswith(("embed_tokens.weight", "lm_head.weight")):
            pad = torch.zeros((w.full_shape[0] - full.shape[0],) + tuple(full.shape[1:]), dtype=full.dtype)
            full = torch.cat([full, pad], dim=0)                     # vocab padded to a multiple of tp
        w.load_full(full.to(w.data.device))


@torch.no_grad()
def save_safetensors(model, path: str) -> None:
    """Writes a HF-named checkpoint (TP=1 models only)."""
    from safetensors.torch import save_file
    cfg = model.cfg
    assert model.ctx.tp_size == 1, "save from an unsharded model"
    out = {}
    for name, w, _kind in model.named_weights():
        t = w.data.detach().cpu()
        if ".experts.w" in name:
            for e, keys in enumerate(_expert_sources(name, cfg)):
                parts = t[e].chunk(len(keys), dim=0)
                for k, p in zip(keys, parts):
                    out[k] = p.contiguous()
            continue
        keys, dim = _hf_sources(name, cfg)
        if len(keys) == 1:
            out[keys[0]] = t[:cfg.vocab_size].contiguous() if name.endswith(("embed_tokens.weight", "lm_head.weight")) else t.contiguous()
        elif "qkv_proj" in name:
            d = cfg.head_dim
            sizes = [cfg.num_heads * d, cfg.num_kv_heads * d, cfg.num_kv_heads * d]
            for k, p in zip(keys, t.split(sizes, dim=0)):
                out[k] = p.contiguous()
        else:
            for k, p in zip(keys, t.chunk(len(keys), dim=0)):
                out[k] = p.contiguous()
    save_file(out, path)
