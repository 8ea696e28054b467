"""A/B of the half tiles of the 256-row grouped GEMM (gemm256.hip: an expert's last tile with ≤ 128 routed rows splits K
between the two wave groups instead of multiplying 128 rows of padding) — VERDICT r2 item 7.

The parent runs itself twice (DLI_G256_HALF_TILES=0 / 1: the switch is read once per process) and prints one line per case
with both times.  Cases: random routing at Mixtral-8x7B shapes, and constructed routings in which every expert's last tile
holds a given number of rows.  GPU time inside a hipGraph.

    python benchmarks/ab_moe_half_tiles.py > gpurun_out/ab_moe_half_tiles.jsonl
"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pairs_for_counts(counts):
    """Token → (expert, expert) pairs such that expert e receives exactly counts[e] rows (top-2 routing)."""
    left = list(counts)
    assert sum(left) % 2 == 0 and max(left) <= sum(left) - max(left)
    out = []
    while sum(left):
        a, b = sorted(range(len(left)), key=lambda e: -left[e])[:2]
        assert left[b] > 0
        out.append((a, b))
        left[a] -= 1
        left[b] -= 1
    return out


def logits_for_pairs(pairs, E, device):
    import torch
    logits = torch.full((len(pairs), E), -4.0, device=device)
    idx = torch.tensor(pairs, device=device)
    logits.scatter_(1, idx, torch.tensor([[3.0, 2.5]], device=device).expand(len(pairs), 2))
    return logits.bfloat16()


def child():
    import torch
    from dli_amd import ops
    import bench_kernels as bk
    dev = torch.device("cuda", 0)
    H, I, E = 4096, 14336, 8
    w13 = (torch.randn(E, 2 * I, H, device=dev) * 0.02).bfloat16()
    w2 = (torch.randn(E, H, I, device=dev) * 0.02).bfloat16()
    lib = ops.hip()
    torch.manual_seed(5)
    cases = []
    for T in (1024, 2048, 4096, 8192):
        cases.append((f"random T={T}", torch.randn(T, E, device=dev).bfloat16()))
    for full, tail in ((1, 64), (1, 128), (1, 192), (3, 64), (3, 128), (3, 192), (3, 256)):
        counts = [full * 256 + tail] * E
        cases.append((f"every expert {full} tiles + {tail} rows", logits_for_pairs(pairs_for_counts(counts), E, dev)))
    for name, logits in cases:
        T = logits.shape[0]
        x = torch.randn(T, H, device=dev).bfloat16()
        top = torch.topk(torch.softmax(logits.float(), -1), 2, -1).indices
        per_expert = torch.bincount(top.flatten(), minlength=E).tolist()
        tails = [c % 256 for c in per_expert]
        t = bk.timeit_graph(lambda: lib.fused_moe(x, logits, w13, w2, 2, 0, 256), 6)
        print(json.dumps({"case": name, "T": T, "rows_per_expert": per_expert, "half_tiles": sum(1 for r in tails if 0 < r <= 128),
                          "tiles": sum(-(-c // 256) for c in per_expert), "us": round(t * 1e6, 1),
                          "TFLOPs": round(2 * 2 * T * 3 * I * H / t / 1e12, 1)}), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        return child()
    res = {}
    for flag in ("0", "1"):
        env = dict(os.environ, DLI_G256_HALF_TILES=flag)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=280)
        if out.returncode != 0:
            sys.stderr.write(out.stderr[-3000:])
            sys.exit(out.returncode)
        res[flag] = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    for a, b in zip(res["0"], res["1"]):
        assert a["case"] == b["case"]
        print(json.dumps(dict(b, us_full_tiles_only=a["us"], TFLOPs_full_tiles_only=a["TFLOPs"], speedup=round(a["us"] / b["us"], 3))), flush=True)


if __name__ == "__main__":
    main()
