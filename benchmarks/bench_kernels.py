"""Micro-benchmarks of the gfx950 kernels at Llama-3-8B serving shapes.

Prints one JSON line per case: achieved HBM bandwidth for the bandwidth-bound ops, TFLOP/s for the
prefill attention.  Timing: hip events around ``iters`` back-to-back launches after warm-up, on
random data (cdna guide §5.4 rule 25).

    python benchmarks/bench_kernels.py [--quick] [--only decode]
"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dli_amd import ops  # noqa: E402
from dli_amd.ops.kv_layout import allocate_kv_cache  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def timeit_graph(fn, calls, replays=5):
    """GPU time per call with the launches captured in one hipGraph (no host launch overhead in the number)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(calls):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (replays * calls) * 1e-3


def report(name, seconds, bytes_=None, flops=None, **kw):
    row = {"kernel": name, "us": round(seconds * 1e6, 2)}
    if bytes_:
        row["GB/s"] = round(bytes_ / seconds / 1e9, 1)
    if flops:
        row["TFLOP/s"] = round(flops / seconds / 1e12, 2)
    row.update(kw)
    print(json.dumps(row), flush=True)


def bench_norm(T, H=4096):
    x = torch.randn(T, H, device=DEV).bfloat16()
    r = torch.randn(T, H, device=DEV).bfloat16()
    w = torch.randn(H, device=DEV).bfloat16()
    report("rms_norm", timeit(lambda: ops.rms_norm(x, w, 1e-5)), bytes_=2 * T * H * 2, T=T, H=H)
    report("fused_add_rms_norm", timeit(lambda: ops.fused_add_rms_norm(x, r, w, 1e-5)), bytes_=4 * T * H * 2, T=T, H=H)


def bench_silu(T, I=14336):
    x = torch.randn(T, 2 * I, device=DEV).bfloat16()
    report("silu_and_mul", timeit(lambda: ops.silu_and_mul(x)), bytes_=3 * T * I * 2, T=T, I=I)


def bench_rope(T, Hq=32, Hkv=8):
    qkv = torch.randn(T, (Hq + 2 * Hkv) * 128, device=DEV).bfloat16()
    nblocks = (T + 31) // 32 + 1
    kc, vc = allocate_kv_cache(1, nblocks, Hkv, 128, torch.bfloat16, DEV)[0]
    pos = torch.arange(T, dtype=torch.int32, device=DEV) % 4096
    slots = torch.arange(T, dtype=torch.int32, device=DEV)
    cs = ops.ref.build_cos_sin_cache(4096, 128, 5e5, DEV)
    t = timeit(lambda: ops.rope_and_cache(qkv, pos, cs, slots, kc, vc, Hq, Hkv, 128))
    report("rope_and_cache", t, bytes_=T * ((Hq + Hkv) * 2 + Hkv * 3) * 128 * 2, T=T)


def _decode_case(B, ctx, Hq, Hkv, kv_dtype, splits=None):
    nb_seq = (ctx + 31) // 32
    nblocks = B * nb_seq + 1
    kc, vc = allocate_kv_cache(1, nblocks, Hkv, 128, kv_dtype, DEV)[0]
    if kv_dtype == torch.bfloat16:
        kc.normal_()
        vc.normal_()
    else:
        kc.view(torch.uint8).random_(0, 120)
        vc.view(torch.uint8).random_(0, 120)
    bt = torch.randperm(nblocks - 1, device=DEV).to(torch.int32).reshape(B, nb_seq)
    sl = torch.full((B,), ctx, dtype=torch.int32, device=DEV)
    qsl = torch.arange(B + 1, dtype=torch.int32, device=DEV)
    qkv = torch.randn(B, (Hq + 2 * Hkv) * 128, device=DEV).bfloat16()
    if splits is None:
        splits = ops.choose_decode_splits(B, Hkv, ctx)
    ws = torch.empty(B * Hq * splits * 130, dtype=torch.float32, device=DEV)
    meta = ops.AttnMetadata(B, 0, B, None, None, bt, sl, qsl, decode_splits=splits, decode_workspace=ws)
    out = torch.empty(B, Hq * 128, device=DEV, dtype=torch.bfloat16)
    scale = 1 / math.sqrt(128)
    fn = lambda: ops.paged_attention(qkv, kc, vc, meta, Hq, Hkv, 128, scale, out=out)
    kv_bytes = B * nb_seq * 32 * Hkv * 128 * 2 * kc.element_size()
    return fn, kv_bytes, splits


def _decode_mixed_case(B, mean_ctx, Hq=32, Hkv=8, seed=0, order="random", worklist=None, pages="random"):
    """Serving-shaped batch: log-normal context lengths (the BurstGPT length mix), one decode token each."""
    rng = np.random.default_rng(seed)
    ctx = np.clip(rng.lognormal(np.log(mean_ctx) - 0.32, 0.8, size=B), 8, 2048).astype(np.int64)
    if order == "longest_first":
        ctx = np.sort(ctx)[::-1].copy()
    elif order == "shortest_first":
        ctx = np.sort(ctx)
    nb = (ctx + 31) // 32
    max_nb = int(nb.max())
    nblocks = int(nb.sum()) + 1
    kc, vc = allocate_kv_cache(1, nblocks, Hkv, 128, torch.bfloat16, DEV)[0]
    kc.normal_()
    vc.normal_()
    # page placement: scattered over the cache (a block manager after churn) or each sequence's pages in a row
    perm = (torch.randperm(nblocks - 1) if pages == "random" else torch.arange(nblocks - 1)).to(torch.int32)
    bt = torch.zeros(B, max_nb, dtype=torch.int32)
    o = 0
    for i in range(B):
        bt[i, :nb[i]] = perm[o:o + nb[i]]
        o += int(nb[i])
    bt = bt.to(DEV)
    sl = torch.from_numpy(ctx).to(torch.int32).to(DEV)
    qsl = torch.arange(B + 1, dtype=torch.int32, device=DEV)
    qkv = torch.randn(B, (Hq + 2 * Hkv) * 128, device=DEV).bfloat16()
    splits = ops.choose_decode_splits(B, Hkv, int(ctx.max()))
    ws = torch.empty(B * Hq * splits * 130, dtype=torch.float32, device=DEV)
    meta = ops.AttnMetadata(B, 0, B, None, None, bt, sl, qsl, decode_splits=splits, decode_workspace=ws)
    if worklist is not None:          # dict(chunk_pages=…, launch_items=…): the per-step work list instead of uniform splits
        ops.attach_decode_worklist(meta, ctx, Hq, Hkv, DEV, **worklist)
        splits = f"worklist:{int(meta.counts[0])} items/{meta.decode_items} launched"
    out = torch.empty(B, Hq * 128, device=DEV, dtype=torch.bfloat16)
    fn = lambda: ops.paged_attention(qkv, kc, vc, meta, Hq, Hkv, 128, 1 / math.sqrt(128), out=out)
    return fn, int(nb.sum()) * 32 * Hkv * 128 * 2 * 2, splits, float(ctx.mean())


def bench_decode(quick):
    for B in ([1024] if quick else [256, 1024, 1400]):
        for order in ("random", "longest_first", "shortest_first"):
            fn, nbytes, splits, mean = _decode_mixed_case(B, 770, order=order)
            report("paged_attention_decode(mixed lengths)", timeit(fn), bytes_=nbytes, B=B, mean_ctx=round(mean), splits=splits,
                   row_order=order)
    for B in ([1024] if quick else [256, 1024]):
        fn, nbytes, splits, mean = _decode_mixed_case(B, 770, order="longest_first", pages="contiguous")
        report("paged_attention_decode(mixed lengths)", timeit(fn), bytes_=nbytes, B=B, mean_ctx=round(mean), splits=splits,
               row_order="longest_first", page_placement="contiguous per sequence")
    # work-list decode: equal-size page ranges instead of one launch-wide split count
    for B in ([1024] if quick else [64, 256, 1024, 1400]):
        for order in ("random", "longest_first"):
            for wl in (dict(), dict(chunk_pages=8), dict(chunk_pages=16), dict(chunk_pages=32), dict(launch_items=3 * B)):
                fn, nbytes, splits, mean = _decode_mixed_case(B, 770, order=order, worklist=wl)
                report("paged_attention_decode(mixed lengths, work list)", timeit(fn), bytes_=nbytes, B=B, mean_ctx=round(mean),
                       splits=splits, row_order=order, **{k: v for k, v in wl.items()})
    cases = [(256, 700), (64, 700), (8, 2048), (1, 2048), (512, 1024)]
    if quick:
        cases = cases[:2]
    for B, ctx in cases:
        for kv_dtype in (torch.bfloat16, torch.float8_e4m3fn):
            fn, nbytes, splits = _decode_case(B, ctx, 32, 8, kv_dtype)
            report("paged_attention_decode", timeit(fn), bytes_=nbytes, B=B, ctx=ctx, splits=splits,
                   kv=str(kv_dtype).split(".")[-1])
    fn, nbytes, splits = _decode_case(256, 1024, 8, 1, torch.bfloat16)       # 70B TP=8 shard
    report("paged_attention_decode", timeit(fn), bytes_=nbytes, B=256, ctx=1024, splits=splits, shape="70b-tp8")


def bench_prefill(quick):
    Hq, Hkv = 32, 8
    for nseq, L in ([(4, 1024)] if quick else [(4, 1024), (16, 512), (2, 2048), (1, 4096)]):
        nb_seq = (L + 31) // 32
        nblocks = nseq * nb_seq + 1
        kc, vc = allocate_kv_cache(1, nblocks, Hkv, 128, torch.bfloat16, DEV)[0]
        kc.normal_()
        vc.normal_()
        bt = torch.randperm(nblocks - 1, device=DEV).to(torch.int32).reshape(nseq, nb_seq)
        sl = torch.full((nseq,), L, dtype=torch.int32, device=DEV)
        qsl = (torch.arange(nseq + 1, device=DEV) * L).to(torch.int32)
        T = nseq * L
        qkv = torch.randn(T, (Hq + 2 * Hkv) * 128, device=DEV).bfloat16()
        tq = ops.prefill_tile_tokens(Hq, Hkv)
        ts, t0 = ops.build_prefill_tiles(np.full(nseq, L), 0, tq)
        meta = ops.AttnMetadata(0, nseq, T, None, None, bt, sl, qsl, prefill_tile_seq=torch.from_numpy(ts).to(DEV),
                                prefill_tile_q0=torch.from_numpy(t0).to(DEV), num_prefill_tiles=len(ts))
        out = torch.empty(T, Hq * 128, device=DEV, dtype=torch.bfloat16)
        scale = 1 / math.sqrt(128)
        t = timeit(lambda: ops.paged_attention(qkv, kc, vc, meta, Hq, Hkv, 128, scale, out=out), iters=10)
        flops = 4 * nseq * (L * L / 2) * 128 * Hq
        report("paged_attention_prefill", t, flops=flops, nseq=nseq, L=L)


def bench_sampler(quick):
    for B, V in ([(256, 128256)] if quick else [(256, 128256), (64, 128256), (256, 32000)]):
        logits = torch.randn(B, V, device=DEV).bfloat16()
        one = torch.ones(B, device=DEV)
        zk = torch.zeros(B, dtype=torch.int32, device=DEV)
        seeds = torch.arange(B, dtype=torch.int64, device=DEV)
        steps = torch.zeros(B, dtype=torch.int64, device=DEV)
        out = torch.empty(B, dtype=torch.int64, device=DEV)
        report("sample(T=0.7)", timeit(lambda: ops.sample(logits, one * 0.7, one, zk, seeds, steps, out=out)),
               bytes_=B * V * 2, B=B, V=V)
        report("sample(greedy)", timeit(lambda: ops.sample(logits, one * 0, one, zk, seeds, steps, out=out)),
               bytes_=B * V * 2, B=B, V=V)
        report("sample(top_p=0.9,top_k=50)",
               timeit(lambda: ops.sample(logits, one * 0.7, one * 0.9, zk + 50, seeds, steps, out=out)), B=B, V=V)


def bench_moe(quick):
    H, I, E = 4096, 14336, 8
    for T, El in ([(256, 8)] if quick else [(1, 8), (4, 8), (8, 8), (16, 8), (16, 1), (32, 8), (64, 8), (256, 8), (256, 1), (512, 8), (1024, 8), (4096, 8), (4096, 1)]):
        x = torch.randn(T, H, device=DEV).bfloat16()
        logits = torch.randn(T, E, device=DEV).bfloat16()
        w13 = (torch.randn(El, 2 * I, H, device=DEV) * 0.02).bfloat16()
        w2 = (torch.randn(El, H, I, device=DEV) * 0.02).bfloat16()
        t = timeit(lambda: ops.fused_moe(x, logits, w13, w2, 2, 0), iters=10)
        rows = T * 2 * El / E                                   # expected rows hitting local experts
        hit = int((torch.topk(logits.float(), 2, dim=-1).indices.unique() < El).sum())   # experts whose weights are read
        report("fused_moe", t, flops=2 * rows * 3 * I * H, bytes_=hit * 3 * I * H * 2, T=T, local_experts=El, experts_hit=hit)


def bench_gemm256(quick):
    """Large-M A/B, GPU time inside hipGraphs, random data, weights rotated past the infinity cache:
    (a) gate/up projection: library GEMM + silu_and_mul kernel  vs  the 256² tile kernel with the SwiGLU epilogue;
    (b) plain projections: library GEMM vs the 256² tile kernel;
    (c) MoE block: 128-row tiles (moe.hip) vs 256-row tiles (gemm256.hip)."""
    import torch.nn.functional as F
    lib = ops.hip()
    shapes = [("8b_gate_up", 28672, 4096, True), ("70b_tp8_gate_up", 7168, 8192, True), ("70b_gate_up", 57344, 8192, True), ("8b_qkv", 6144, 4096, False),
              ("8b_o", 4096, 4096, False), ("8b_down", 4096, 14336, False), ("70b_tp8_down", 8192, 3584, False)]
    ms_env = os.environ.get("DLI_BENCH_GEMM_MS")          # e.g. "2304,2816,3328": the step sizes the serving loop produces
    ms = tuple(int(v) for v in ms_env.split(",")) if ms_env else (512, 1024, 2048, 4096, 8192)
    only_shapes = os.environ.get("DLI_BENCH_GEMM_SHAPES")
    if only_shapes:
        shapes = [sh for sh in shapes if sh[0] in only_shapes.split(",")]
    if not only_shapes:
        shapes = [sh for sh in shapes if sh[0] != "70b_gate_up"]       # 0.9 GB per copy: on request only
    for name, N, K, swiglu in (shapes[:2] if quick else shapes):
        if not swiglu and N % 256:
            continue
        copies = max(2, -(-(600 << 20) // (N * K * 2)))
        ws = [(torch.randn(N, K, device=DEV) * 0.02).bfloat16() for _ in range(copies)]
        for M in ((4096,) if quick else ms):
            x = torch.randn(M, K, device=DEV).bfloat16()
            state = {"i": 0}

            def call(fn):
                def run():
                    state["i"] = (state["i"] + 1) % copies
                    return fn(x, ws[state["i"]])
                return run
            if swiglu:
                t_lib = timeit_graph(call(lambda a, w: ops.silu_and_mul(F.linear(a, w))), 2 * copies)
                t_gemm = timeit_graph(call(lambda a, w: F.linear(a, w)), 2 * copies)
            else:
                t_lib = t_gemm = timeit_graph(call(lambda a, w: F.linear(a, w)), 2 * copies)
            t_new = timeit_graph(call(lambda a, w: lib.dense_gemm256(a, w, swiglu)), 2 * copies)
            fl = 2.0 * M * N * K
            report("gemm256(" + name + (",swiglu" if swiglu else "") + ")", t_new, flops=fl, M=M, N=N, K=K,
                   library_us=round(t_lib * 1e6, 1), library_gemm_only_us=round(t_gemm * 1e6, 1),
                   library_TFLOPs=round(fl / t_gemm / 1e12, 1), speedup_vs_library_path=round(t_lib / t_new, 3))
        del ws
    H, I, E = 4096, 14336, 8
    moe_ts = os.environ.get("DLI_BENCH_MOE_TS")              # e.g. "384,512,768": sweep the tile-size crossover
    cases = [(int(v), 8) for v in moe_ts.split(",")] if moe_ts else [(1024, 8), (2048, 8), (4096, 8), (8192, 8), (4096, 1)]
    for T, El in ([(4096, 8)] if quick else cases if (moe_ts or not only_shapes) else []):
        x = torch.randn(T, H, device=DEV).bfloat16()
        logits = torch.randn(T, E, device=DEV).bfloat16()
        w13 = (torch.randn(El, 2 * I, H, device=DEV) * 0.02).bfloat16()
        w2 = (torch.randn(El, H, I, device=DEV) * 0.02).bfloat16()
        rows = T * 2 * El / E
        ts = {tile: timeit_graph(lambda: lib.fused_moe(x, logits, w13, w2, 2, 0, tile), 6) for tile in (128, 256)}
        report("fused_moe(tile 256 vs 128)", ts[256], flops=2 * rows * 3 * I * H, T=T, local_experts=El,
               tile128_us=round(ts[128] * 1e6, 1), tile128_TFLOPs=round(2 * rows * 3 * I * H / ts[128] / 1e12, 1),
               speedup=round(ts[128] / ts[256], 3))


def bench_midm(quick):
    """the library GEMM (hipBLASLt) and the path `ops` dispatches against the roofline
    max(weight bytes / 5.8 TB/s, FLOPs / 2.0 PFLOP/s) for every projection of Llama-3-8B, Llama-3-70B and the 70B TP = 8
    shard at M = 64 … 2048 (where decode-heavy TP = 8 steps live), GPU time inside hipGraphs, weights rotated past the
    infinity cache.  `off_roofline` > 1.25 marks the shapes a hand-written mid-M tile would have to beat."""
    import torch.nn.functional as F
    lib = ops.hip()
    models = {
        "8b": [("qkv", 6144, 4096, False), ("o", 4096, 4096, False), ("gate_up", 28672, 4096, True), ("down", 4096, 14336, False)],
        "70b_tp8": [("qkv", 1280, 8192, False), ("o", 8192, 1024, False), ("gate_up", 7168, 8192, True), ("down", 8192, 3584, False)],
        "70b": [("qkv", 10240, 8192, False), ("o", 8192, 8192, False), ("gate_up", 57344, 8192, True), ("down", 8192, 28672, False)],
    }
    ms_env = os.environ.get("DLI_BENCH_MIDM_MS")           # e.g. "2304,2816,3328": the step sizes of the saturated 8B loop
    only_models = os.environ.get("DLI_BENCH_MIDM_MODELS")
    for model, shapes in (list(models.items())[:1] if quick else models.items()):
        if only_models and model not in only_models.split(","):
            continue
        for name, N, K, swiglu in shapes:
            copies = max(2, -(-(600 << 20) // (N * K * 2)))
            ws = [(torch.randn(N, K, device=DEV) * 0.02).bfloat16() for _ in range(copies)]
            for M in ((256,) if quick else tuple(int(v) for v in ms_env.split(",")) if ms_env else (64, 128, 256, 512, 1024, 2048)):
                x = torch.randn(M, K, device=DEV).bfloat16()
                state = {"i": 0}

                def call(fn):
                    def run():
                        state["i"] = (state["i"] + 1) % copies
                        return fn(x, ws[state["i"]])
                    return run
                t_lib = timeit_graph(call(lambda a, w: F.linear(a, w)), 2 * copies)
                if swiglu:
                    t_lib_path = timeit_graph(call(lambda a, w: ops.silu_and_mul(F.linear(a, w))), 2 * copies)
                    t_ops = timeit_graph(call(lambda a, w: ops.linear_swiglu(a, w)), 2 * copies)
                else:
                    t_lib_path = t_lib
                    t_ops = timeit_graph(call(lambda a, w: ops.linear(a, w)), 2 * copies)
                fl = 2.0 * M * N * K
                roof = max(N * K * 2 / 5.8e12, fl / 2.0e15)
                # the 256² tile with K cut into slices (fp32 partial slabs + reduce pass), whatever `ops` would pick
                sk = {}
                n_out = N // 2 if swiglu else N
                if n_out % (128 if swiglu else 256) == 0 and N * K * 2 < 2 ** 32 and os.environ.get("DLI_BENCH_SPLITK", "1") != "0":
                    for ksp in (1, 2, 4, 8, 16):
                        if ksp <= K // 256:
                            sk[ksp] = round(timeit_graph(call(lambda a, w: lib.dense_gemm256(a, w, swiglu, ksp)), 2 * copies) * 1e6, 1)
                best_sk = min(sk, key=sk.get) if sk else None
                from dli_amd.ops.gemm_plan import GemmPlan
                tail = None
                if sk and GemmPlan.tail_slices(M, N, K, swiglu):     # whole rounds plain, remainder tiles K-split
                    tail = {"slices": GemmPlan.tail_slices(M, N, K, swiglu),
                            "us": round(timeit_graph(call(lambda a, w: lib.dense_gemm256(a, w, swiglu, 1, None, True)), 2 * copies) * 1e6, 1)}
                report(f"midm({model}_{name})", t_ops, flops=fl, M=M, N=N, K=K, library_gemm_us=round(t_lib * 1e6, 1),
                       tile256_splitk_us=sk, best_splitk=best_sk, tile256_tail=tail,
                       best_splitk_vs_library_path=round(t_lib_path * 1e6 / sk[best_sk], 3) if sk else None,
                       library_path_us=round(t_lib_path * 1e6, 1), roofline_us=round(roof * 1e6, 1),
                       library_off_roofline=round(t_lib / roof, 2), dispatched_off_roofline=round(t_ops / roof, 2),
                       dispatched_vs_library_path=round(t_lib_path / t_ops, 3))
            del ws


def bench_linear(quick):
    """Decode-sized linear layers: the weight-streaming kernel against the library GEMM (hipBLASLt via torch).
    Weights rotate through ≥ 1 GB of copies so that no call finds its matrix in the 256 MB infinity cache."""
    import torch.nn.functional as F
    shapes = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336), ("lm_head", 128256, 4096)]
    if os.environ.get("DLI_BENCH_TP_SHAPES"):       # per-rank shards: Llama-3-70B at TP=8, Llama-3-8B at TP=4
        shapes = [("70b_tp8_qkv", 1280, 8192), ("70b_tp8_o", 8192, 1024), ("70b_tp8_gate_up", 7168, 8192),
                  ("70b_tp8_down", 8192, 3584), ("8b_tp4_qkv", 1536, 4096), ("8b_tp4_o", 4096, 1024),
                  ("8b_tp4_gate_up", 7168, 4096), ("8b_tp4_down", 4096, 3584), ("70b_tp8_lm_head", 16032, 8192)]
    for name, N, K in (shapes[1:4] if quick else shapes):
        copies = max(2, -(-(1 << 30) // (N * K * 2)))
        ws = [(torch.randn(N, K, device=DEV) * 0.02).bfloat16() for _ in range(copies)]
        for M in ((1, 16) if quick else (24, 32, 48, 64) if os.environ.get("DLI_BENCH_MULTIPASS") else (1, 8, 16, 32) if os.environ.get("DLI_BENCH_GRAPH") else (1, 4, 8, 16)):
            x = torch.randn(M, K, device=DEV).bfloat16()
            state = {"i": 0}

            def call(fn):
                def run():
                    state["i"] = (state["i"] + 1) % copies
                    return fn(x, ws[state["i"]])
                return run
            if os.environ.get("DLI_BENCH_GRAPH"):       # both sides inside a hipGraph: pure GPU time
                t_lib = timeit_graph(call(lambda a, w: F.linear(a, w)), 4 * copies)
                t_sk = timeit_graph(call(lambda a, w: ops.hip().skinny_linear(a, w, None, False)), 4 * copies)
                report("linear(" + name + ")", t_sk, bytes_=N * K * 2, M=M, N=N, K=K, library_us=round(t_lib * 1e6, 2),
                       speedup=round(t_lib / t_sk, 2), timing="hipGraph")
                continue
            t_lib = timeit(call(lambda a, w: F.linear(a, w)), iters=4 * copies, warmup=copies)
            t_sk = timeit(call(lambda a, w: ops.hip().skinny_linear(a, w, None, False) if M > 16 else ops.linear(a, w)),
                          iters=4 * copies, warmup=copies)
            if name.endswith("gate_up"):
                t_f = timeit(call(lambda a, w: ops.linear_swiglu(a, w)), iters=4 * copies, warmup=copies)
                t_u = timeit(call(lambda a, w: ops.silu_and_mul(ops.linear(a, w))), iters=4 * copies, warmup=copies)
                report("linear_swiglu(gate_up)", t_f, bytes_=N * K * 2, M=M, N=N, K=K, unfused_us=round(t_u * 1e6, 2))
            report("linear(" + name + ")", t_sk, bytes_=N * K * 2, M=M, N=N, K=K, library_us=round(t_lib * 1e6, 2),
                   speedup=round(t_lib / t_sk, 2))
        del ws


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    ops.hip()
    only = [t for t in a.only.split(",") if t]
    run = lambda name: not only or any(t in name for t in only)
    if run("norm"):
        for T in (256, 4096):
            bench_norm(T)
    if run("silu"):
        for T in (256, 4096):
            bench_silu(T)
    if run("rope"):
        for T in (256, 4096):
            bench_rope(T)
    if run("decode"):
        bench_decode(a.quick)
    if run("prefill"):
        bench_prefill(a.quick)
    if run("sampler"):
        bench_sampler(a.quick)
    if run("moe"):
        bench_moe(a.quick)
    if run("gemm256"):
        bench_gemm256(a.quick)
    if run("linear"):
        bench_linear(a.quick)
    if only and run("midm"):                                  # on request only (minutes: 3 models × 4 projections × 6 sizes)
        bench_midm(a.quick)


if __name__ == "__main__":
    main()
