import sys, os
sys.path.insert(0, os.getcwd())
import torch
from dli_amd import ops
from dli_amd.ops import reference as ref
torch.manual_seed(0)
for T in (7, 47, 1247):
    H, I, E = 512, 1024, 4
    x = torch.randn(T, H, device="cuda").bfloat16()
    logits = (torch.randn(T, E, device="cuda") * 2).bfloat16()
    w13 = (torch.randn(E, 2 * I, H, device="cuda") * 0.03).bfloat16()
    w2 = (torch.randn(E, H, I, device="cuda") * 0.03).bfloat16()
    full = ops.fused_moe(x, logits, w13, w2, 2, 0).float()
    want = ref.moe_ref(x.float(), logits.float(), w13.float(), w2.float(), 2, 0)
    a = ops.fused_moe(x, logits, w13[:2].contiguous(), w2[:2].contiguous(), 2, 0).float()
    b = ops.fused_moe(x, logits, w13[2:].contiguous(), w2[2:].contiguous(), 2, 2).float()
    ra = ref.moe_ref(x.float(), logits.float(), w13[:2].float(), w2[:2].float(), 2, 0)
    rb = ref.moe_ref(x.float(), logits.float(), w13[2:].float(), w2[2:].float(), 2, 2)
    sc = want.abs().max()
    print(T, "full", float((full - want).abs().max() / sc), "a", float((a - ra).abs().max() / sc), "b", float((b - rb).abs().max() / sc),
          "sum", float((a + b - want).abs().max() / sc))
    # non-contiguous expert slices (what a narrow() view would give)
    try:
        c = ops.fused_moe(x, logits, w13[2:], w2[2:], 2, 2).float()
        print("  view ok", float((c - rb).abs().max() / sc))
    except Exception as e:
        print("  view err", str(e)[:100])
