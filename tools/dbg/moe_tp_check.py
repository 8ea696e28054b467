"""Does the sum of the two TP shards of the MoE block equal the unsharded block (HIP kernels, real Mixtral shapes)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dli_amd import ops
from dli_amd.ops import reference as ref

torch.manual_seed(0)
H, I, E = 4096, 14336, 8
for T in (51, 700):
    x = torch.randn(T, H, device="cuda").bfloat16()
    logits = torch.randn(T, E, device="cuda").bfloat16()
    w13 = (torch.randn(E, 2 * I, H, device="cuda") * 0.02).bfloat16()
    w2 = (torch.randn(E, H, I, device="cuda") * 0.02).bfloat16()
    full = ops.fused_moe(x, logits, w13, w2, 2, 0).float()
    want = ref.moe_ref(x.float(), logits.float(), w13.float(), w2.float(), 2, 0).float()
    parts = []
    for r in range(2):
        h = I // 2
        w13_r = torch.cat([w13[:, r * h:(r + 1) * h], w13[:, I + r * h:I + (r + 1) * h]], dim=1).contiguous()
        w2_r = w2[:, :, r * h:(r + 1) * h].contiguous()
        parts.append(ops.fused_moe(x, logits, w13_r, w2_r, 2, 0).float())
    tp = parts[0] + parts[1]
    cos = lambda a, b: torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
    print(f"T={T}: full vs ref {cos(full, want):.5f}   tp-sum vs ref {cos(tp, want):.5f}   tp-sum vs full {cos(tp, full):.5f}"
          f"   max|tp-ref| {(tp - want).abs().max():.4f}  max|full-ref| {(full - want).abs().max():.4f}  std {want.std():.4f}")
    bad = ((tp - want).abs() > 0.05 * want.abs().max()).nonzero()
    print("   outliers:", bad.shape[0], bad[:5].tolist())
