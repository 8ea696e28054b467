import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from dli_amd import ops
from dli_amd.ops import reference as ref
from dli_amd.config import get_model_config
from dli_amd.models import build_model
cfg = get_model_config("tiny-mixtral-tp")
m = build_model(cfg, device="cuda", seed=0)
r = np.random.default_rng(0)
toks = np.concatenate([r.integers(3, 1000, size=n) for n in (40, 1200, 7)])
for T in (1200, 1247):
    x = m.embed(torch.tensor(toks[:T], device="cuda"))
    x = ops.rms_norm(x, m.layers[0].input_norm.data, 1e-5)
    moe = m.layers[0].mlp
    logits = moe.router(x)
    w13, w2 = moe.w13.data, moe.w2.data
    full = ops.fused_moe(x, logits, w13, w2, 2, 0).float()
    want = ref.moe_ref(x.float(), logits.float(), w13.float(), w2.float(), 2, 0)
    a = ops.fused_moe(x, logits, w13[:2].contiguous(), w2[:2].contiguous(), 2, 0).float()
    b = ops.fused_moe(x, logits, w13[2:].contiguous(), w2[2:].contiguous(), 2, 2).float()
    sc = want.abs().max()
    top = torch.softmax(logits.float(), -1).topk(2, -1).indices
    print(T, "counts", torch.bincount(top.flatten(), minlength=4).tolist(), "full", float((full - want).abs().max() / sc),
          "a+b", float((a + b - want).abs().max() / sc), "finite", bool(torch.isfinite(a).all() and torch.isfinite(b).all()))
