"""Checkpoint I/O (SURVEY.md §5.4 [NEW-BUILD]): HF-named safetensors round trip, fused-parameter
assembly (q/k/v, gate/up, per-expert w1/w3/w2) and TP-sharded loading."""
import numpy as np
import pytest
import torch
from safetensors import safe_open

from dli_amd.config import get_model_config
from dli_amd.models import build_model, save_safetensors
from dli_amd.parallel.state import ParallelContext


@pytest.mark.parametrize("name", ["tiny-llama", "tiny-mixtral", "tiny-opt"])
def test_safetensors_roundtrip_and_hf_names(tmp_path, name):
    cfg = get_model_config(name)
    a = build_model(cfg, seed=3)
    path = str(tmp_path / "model.safetensors")
    save_safetensors(a, path)
    with safe_open(path, framework="pt") as f:
        keys = set(f.keys())
    if name == "tiny-llama":
        assert {"model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.k_proj.weight",
                "model.layers.1.mlp.gate_proj.weight", "model.layers.1.mlp.up_proj.weight", "lm_head.weight",
                "model.embed_tokens.weight", "model.norm.weight"} <= keys
    if name == "tiny-mixtral":
        assert {"model.layers.0.block_sparse_moe.experts.3.w1.weight", "model.layers.0.block_sparse_moe.experts.3.w2.weight",
                "model.layers.0.block_sparse_moe.experts.0.w3.weight", "model.layers.0.block_sparse_moe.gate.weight"} <= keys
    b = build_model(cfg, seed=99, weights_path=path)            # different seed: everything must come from the file
    for (n1, w1, _), (_, w2, _) in zip(a.named_weights(), b.named_weights()):
        assert torch.equal(w1.data, w2.data), n1


def test_tp_sharded_load_takes_the_right_slices(tmp_path):
    cfg = get_model_config("tiny-llama")
    full = build_model(cfg, seed=5)
    path = str(tmp_path / "m.safetensors")
    save_safetensors(full, path)
    tp = 2
    for rank in range(tp):
        ctx = ParallelContext(tp_size=tp, tp_rank=rank)
        shard = build_model(cfg, ctx=ctx, weights_path=path)
        d = cfg.head_dim
        qkv_full = full.layers[0].attn.qkv.weight.data
        hq, hkv = cfg.num_heads * d, cfg.num_kv_heads * d
        want = torch.cat([qkv_full[rank * hq // tp:(rank + 1) * hq // tp],
                          qkv_full[hq + rank * hkv // tp: hq + (rank + 1) * hkv // tp],
                          qkv_full[hq + hkv + rank * hkv // tp: hq + hkv + (rank + 1) * hkv // tp]])
        assert torch.equal(shard.layers[0].attn.qkv.weight.data, want)
        I = cfg.intermediate_size
        assert torch.equal(shard.layers[1].mlp.down.weight.data, full.layers[1].mlp.down.weight.data[:, rank * I // tp:(rank + 1) * I // tp])
        V = cfg.vocab_size
        assert torch.equal(shard.embed.weight.data, full.embed.weight.data[rank * V // tp:(rank + 1) * V // tp])
