"""HIP kernel numerics vs the plain PyTorch fp32 references.

Every test feeds the *same* bf16 inputs to the gfx950 kernel and to ``ops.reference`` and compares
in fp32.  Shapes follow SURVEY.md §2.5 (Llama-3-8B: H 4096, 32/8 heads, D 128; 70B-TP8: G = 8).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dli_amd import ops
from dli_amd.ops import reference as ref
from dli_amd.ops.kv_layout import allocate_kv_cache

DEV = "cuda"


def _bf(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


def test_extension_is_loaded():
    ops.hip()
    assert ops.hip_loaded()


@pytest.mark.parametrize("T,H", [(1, 4096), (37, 4096), (256, 8192), (5, 16384), (3, 512)])
def test_rms_norm(T, H):
    x, w = _bf(T, H, seed=1), _bf(H, seed=2) * 0.1 + 1
    got = ops.rms_norm(x, w, 1e-5).float()
    want = ref.rms_norm_ref(x.float(), w.float(), 1e-5)
    torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2)


def test_rms_norm_strided_rows():
    big = _bf(9, 3 * 4096, seed=3)
    x = big[:, 4096:8192]
    w = _bf(4096, seed=4)
    torch.testing.assert_close(ops.rms_norm(x, w, 1e-5).float(), ref.rms_norm_ref(x.float(), w.float(), 1e-5),
                               rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("T,H", [(1, 4096), (130, 4096), (64, 8192), (4, 16384)])
def test_fused_add_rms_norm(T, H):
    x, r, w = _bf(T, H, seed=5), _bf(T, H, seed=6), _bf(H, seed=7)
    xs, rs = x.float(), r.float()
    s = xs + rs
    ops.fused_add_rms_norm(x, r, w, 1e-5)
    torch.testing.assert_close(r.float(), s.to(torch.bfloat16).float(), rtol=0, atol=0)
    torch.testing.assert_close(x.float(), ref.rms_norm_ref(s, w.float(), 1e-5), rtol=2e-2, atol=3e-2)


@pytest.mark.parametrize("T,I", [(1, 14336), (77, 14336), (300, 1792), (2, 28672)])
def test_silu_and_mul(T, I):
    x = _bf(T, 2 * I, seed=8, scale=2.0)
    torch.testing.assert_close(ops.silu_and_mul(x).float(), ref.silu_and_mul_ref(x.float()), rtol=2e-2, atol=2e-2)


def _caches(nblocks, hkv, dtype=torch.bfloat16, layers=1):
    return allocate_kv_cache(layers, nblocks, hkv, 128, dtype, DEV)[0]


@pytest.mark.parametrize("Hq,Hkv", [(32, 8), (8, 1), (4, 4)])
@pytest.mark.parametrize("kv_dtype", [torch.bfloat16, torch.float8_e4m3fn])
@pytest.mark.parametrize("layout", ["scattered", "chunk", "mixed"])
def test_rope_and_cache(Hq, Hkv, kv_dtype, layout):
    """scattered: decode-like (unrelated slots, scalar V stores); chunk: a prefill chunk in consecutive slots starting
    mid-group (the 4-token grouped V stores and the unaligned edges); mixed: decode rows followed by a chunk."""
    T, nblocks = 75, 16
    qkv = _bf(T, (Hq + 2 * Hkv) * 128, seed=9)
    pos = torch.randint(0, 4000, (T,), dtype=torch.int32, device=DEV)
    if layout == "scattered":
        slots = torch.randperm(nblocks * 32, device=DEV)[:T].to(torch.int32)
    elif layout == "chunk":
        slots = torch.arange(37, 37 + T, dtype=torch.int32, device=DEV)
    else:
        slots = torch.cat([torch.randperm(256, device=DEV)[:23].to(torch.int32) + 256,
                           torch.arange(64, 64 + T - 23, dtype=torch.int32, device=DEV)])
    slots[5] = -1                                                   # padding token: no cache write
    cs = ref.build_cos_sin_cache(4096, 128, 500000.0, DEV)
    ks, vs = (0.05, 0.07) if kv_dtype == torch.float8_e4m3fn else (1.0, 1.0)
    kc, vc = _caches(nblocks, Hkv, kv_dtype)
    kc_ref, vc_ref = _caches(nblocks, Hkv, kv_dtype)
    qkv_ref = qkv.clone()
    ops.rope_and_cache(qkv, pos, cs, slots, kc, vc, Hq, Hkv, 128, ks, vs)
    ref.rope_and_cache_ref(qkv_ref, pos, cs, slots, kc_ref, vc_ref, Hq, Hkv, 128, ks, vs)
    torch.testing.assert_close(qkv.float(), qkv_ref.float(), rtol=2e-2, atol=2e-2)
    if kv_dtype == torch.bfloat16:
        torch.testing.assert_close(kc.float(), kc_ref.float(), rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(vc.float(), vc_ref.float(), rtol=0, atol=0)
    else:                                                           # one e4m3 ulp = 12.5 %
        torch.testing.assert_close(kc.float(), kc_ref.float(), rtol=0.13, atol=0.02)
        torch.testing.assert_close(vc.float(), vc_ref.float(), rtol=0.13, atol=0.02)


def test_fp8_cache_write_saturates_instead_of_nan():
    """Values past scale × 448 (activation outliers beyond the calibrated head-room) must store ±448, never the e4m3
    NaN pattern: one NaN key poisons every later softmax of its sequence (found with the 80 req/s fp8 bench)."""
    Hq, Hkv, T = 4, 2, 8
    qkv = _bf(T, (Hq + 2 * Hkv) * 128, seed=3) * 100.0            # K, V up to ≈ ±400 with scales 0.05 → ±8000 / scale
    qkv[0, Hq * 128:] = float("inf")
    qkv[1, Hq * 128:] = float("nan")
    pos = torch.zeros(T, dtype=torch.int32, device=DEV)
    slots = torch.arange(T, dtype=torch.int32, device=DEV)
    cs = ref.build_cos_sin_cache(16, 128, 500000.0, DEV)
    kc, vc = _caches(2, Hkv, torch.float8_e4m3fn)
    ops.rope_and_cache(qkv, pos, cs, slots, kc, vc, Hq, Hkv, 128, 0.05, 0.05)
    for c in (kc, vc):
        b = c.view(torch.uint8)
        assert not ((b & 0x7F) == 0x7F).any(), "NaN pattern in the fp8 cache"
        assert torch.isfinite(c.float()).all() and c.float().abs().max() == 448.0


def _fill_cache(Hq, Hkv, seq_lens, kv_dtype, seed=0, scales=(1.0, 1.0)):
    """Random K/V for every sequence written through the reference packer; returns caches + tables."""
    g = torch.Generator().manual_seed(seed)
    B = len(seq_lens)
    max_blocks = max((L + 31) // 32 for L in seq_lens) + 1
    nblocks = sum((L + 31) // 32 for L in seq_lens) + 4
    kc, vc = _caches(nblocks, Hkv, kv_dtype)
    perm = torch.randperm(nblocks, generator=g)
    bt = torch.zeros(B, max_blocks, dtype=torch.int32)
    used = 0
    for i, L in enumerate(seq_lens):
        nb = (L + 31) // 32
        bt[i, :nb] = perm[used:used + nb].to(torch.int32)
        used += nb
        if L == 0:
            continue
        k = (torch.randn(L, Hkv, 128, generator=g)).to(torch.bfloat16).to(DEV)
        v = (torch.randn(L, Hkv, 128, generator=g)).to(torch.bfloat16).to(DEV)
        slots = (bt[i, :nb].long()[:, None] * 32 + torch.arange(32)[None, :]).reshape(-1)[:L].to(torch.int32).to(DEV)
        from dli_amd.ops.kv_layout import write_kv_ref
        write_kv_ref(k, v, slots, kc, vc, *scales)
    return kc, vc, bt.to(DEV)


@pytest.mark.parametrize("Hq,Hkv", [(32, 8), (8, 1), (16, 1)])
@pytest.mark.parametrize("splits", [1, 3, 8])
@pytest.mark.parametrize("kv_dtype", [torch.bfloat16, torch.float8_e4m3fn])
def test_paged_attention_decode(Hq, Hkv, splits, kv_dtype):
    seq_lens = [1, 31, 32, 33, 700, 2048, 129, 0, 1000]
    scales = (0.5, 0.25) if kv_dtype == torch.float8_e4m3fn else (1.0, 1.0)
    kc, vc, bt = _fill_cache(Hq, Hkv, seq_lens, kv_dtype, seed=11, scales=scales)
    B = len(seq_lens)
    qkv = _bf(B, (Hq + 2 * Hkv) * 128, seed=12)
    sl = torch.tensor(seq_lens, dtype=torch.int32, device=DEV)
    qsl = torch.arange(B + 1, dtype=torch.int32, device=DEV)
    ws = torch.empty(B * Hq * splits * 130, dtype=torch.float32, device=DEV)
    meta = ops.AttnMetadata(num_decode=B, num_prefill=0, num_tokens=B, positions=None, slot_mapping=None,
                            block_tables=bt, seq_lens=sl, query_start_loc=qsl, decode_splits=splits,
                            decode_workspace=ws)
    scale = 1 / math.sqrt(128)
    got = ops.paged_attention(qkv, kc, vc, meta, Hq, Hkv, 128, scale, *scales).float()
    want = ref.paged_attention_ref(qkv[:, :Hq * 128].reshape(B, Hq, 128), kc, vc, bt, sl, qsl, scale, *scales).float()
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2)
    assert got[7].abs().max() == 0                                  # empty (padded) sequence → zeros


@pytest.mark.parametrize("Hq,Hkv", [(32, 8), (8, 1)])
@pytest.mark.parametrize("wl", [dict(), dict(chunk_pages=4), dict(chunk_pages=16), dict(cap_items=10), dict(launch_items=64)])
@pytest.mark.parametrize("kv_dtype", [torch.bfloat16, torch.float8_e4m3fn])
def test_paged_attention_decode_work_list(Hq, Hkv, wl, kv_dtype):
    """Work-list decode: ragged page ranges per sequence — one range for short rows, several
    equal-size ones for long rows, combined by the variable-count reduce — against the fp32 reference; also with the
    list squeezed under a cap (bigger chunks) and with a launch grid larger than the list (what a hipGraph replays:
    the surplus workgroups must leave on the device-side count)."""
    seq_lens = [1, 31, 32, 33, 700, 2048, 129, 1000, 65]
    scales = (0.5, 0.25) if kv_dtype == torch.float8_e4m3fn else (1.0, 1.0)
    kc, vc, bt = _fill_cache(Hq, Hkv, seq_lens, kv_dtype, seed=21, scales=scales)
    B = len(seq_lens)
    qkv = _bf(B, (Hq + 2 * Hkv) * 128, seed=22)
    sl = torch.tensor(seq_lens, dtype=torch.int32, device=DEV)
    qsl = torch.arange(B + 1, dtype=torch.int32, device=DEV)
    meta = ops.AttnMetadata(num_decode=B, num_prefill=0, num_tokens=B, positions=None, slot_mapping=None,
                            block_tables=bt, seq_lens=sl, query_start_loc=qsl)
    ops.attach_decode_worklist(meta, np.array(seq_lens), Hq, Hkv, DEV, **wl)
    n_items = int(meta.counts[0])
    if "cap_items" in wl:
        assert n_items <= 10
    if "chunk_pages" in wl:
        assert n_items > B                                          # the long rows really were cut
    scale = 1 / math.sqrt(128)
    out = torch.full((B, Hq * 128), 7.0, dtype=torch.bfloat16, device=DEV)
    got = ops.paged_attention(qkv, kc, vc, meta, Hq, Hkv, 128, scale, *scales, out=out).float()
    want = ref.paged_attention_ref(qkv[:, :Hq * 128].reshape(B, Hq, 128), kc, vc, bt, sl, qsl, scale, *scales).float()
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2)


def test_decode_work_list_equals_uniform_splits_bitwise_on_single_range_rows():
    """Rows that fit one range take the same code path as the uniform kernel with one split: identical bits."""
    Hq, Hkv = 32, 8
    seq_lens = [5, 64, 100, 127]
    kc, vc, bt = _fill_cache(Hq, Hkv, seq_lens, torch.bfloat16, seed=31)
    B = len(seq_lens)
    qkv = _bf(B, (Hq + 2 * Hkv) * 128, seed=32)
    sl = torch.tensor(seq_lens, dtype=torch.int32, device=DEV)
    qsl = torch.arange(B + 1, dtype=torch.int32, device=DEV)
    m1 = ops.AttnMetadata(B, 0, B, None, None, bt, sl, qsl, decode_splits=1)
    m2 = ops.attach_decode_worklist(ops.AttnMetadata(B, 0, B, None, None, bt, sl, qsl), np.array(seq_lens), Hq, Hkv, DEV)
    a = ops.paged_attention(qkv, kc, vc, m1, Hq, Hkv, 128, 0.088)
    b = ops.paged_attention(qkv, kc, vc, m2, Hq, Hkv, 128, 0.088)
    assert torch.equal(a, b)


def test_decode_softmax_rescale_branch():
    """One key dominates late in the sequence: forces a large running-max jump (guide rule 26)."""
    Hq, Hkv, L = 32, 8, 300
    kc, vc, bt = _fill_cache(Hq, Hkv, [L], torch.bfloat16, seed=13)
    qkv = _bf(1, (Hq + 2 * Hkv) * 128, seed=14)
    q = qkv[:, :Hq * 128].reshape(1, Hq, 128)
    # overwrite key 257 of every kv head with 6 × the matching query direction
    from dli_amd.ops.kv_layout import write_kv_ref
    k_spike = (q[0, ::4] * 6).reshape(1, Hkv, 128).to(torch.bfloat16)
    v_spike = torch.full((1, Hkv, 128), 3.0, dtype=torch.bfloat16, device=DEV)
    slot = (bt[0, 257 // 32].long() * 32 + 257 % 32).reshape(1).to(torch.int32)
    write_kv_ref(k_spike, v_spike, slot, kc, vc)
    sl = torch.tensor([L], dtype=torch.int32, device=DEV)
    qsl = torch.tensor([0, 1], dtype=torch.int32, device=DEV)
    meta = ops.AttnMetadata(1, 0, 1, None, None, bt, sl, qsl, decode_splits=1)
    scale = 1 / math.sqrt(128)
    got = ops.paged_attention(qkv, kc, vc, meta, Hq, Hkv, 128, scale).float()
    want = ref.paged_attention_ref(q, kc, vc, bt, sl, qsl, scale).float()
    torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("Hq,Hkv", [(32, 8), (8, 1), (4, 4), (4, 2)])
@pytest.mark.parametrize("kv_dtype", [torch.bfloat16, torch.float8_e4m3fn])
def test_paged_attention_prefill_and_mixed(Hq, Hkv, kv_dtype):
    # (context length incl. new tokens, new tokens): fresh prompts, chunk continuations, 2 decodes first
    seqs = [(500, 1), (64, 1), (300, 300), (1024, 200), (33, 17), (96, 96), (1500, 31), (16, 16)]
    seq_lens = [s[0] for s in seqs]
    qlens = [s[1] for s in seqs]
    scales = (0.5, 0.25) if kv_dtype == torch.float8_e4m3fn else (1.0, 1.0)
    kc, vc, bt = _fill_cache(Hq, Hkv, seq_lens, kv_dtype, seed=21, scales=scales)
    T = sum(qlens)
    qkv = _bf(T, (Hq + 2 * Hkv) * 128, seed=22)
    sl = torch.tensor(seq_lens, dtype=torch.int32, device=DEV)
    qsl = torch.tensor(np.concatenate([[0], np.cumsum(qlens)]), dtype=torch.int32, device=DEV)
    tq = ops.prefill_tile_tokens(Hq, Hkv)
    ts, t0 = ops.build_prefill_tiles(np.array(qlens[2:]), 2, tq)
    meta = ops.AttnMetadata(num_decode=2, num_prefill=len(seqs) - 2, num_tokens=T, positions=None, slot_mapping=None,
                            block_tables=bt, seq_lens=sl, query_start_loc=qsl, decode_splits=1,
                            prefill_tile_seq=torch.from_numpy(ts).to(DEV), prefill_tile_q0=torch.from_numpy(t0).to(DEV),
                            num_prefill_tiles=len(ts))
    scale = 1 / math.sqrt(128)
    got = ops.paged_attention(qkv, kc, vc, meta, Hq, Hkv, 128, scale, *scales).float()
    want = ref.paged_attention_ref(qkv[:, :Hq * 128].reshape(T, Hq, 128), kc, vc, bt, sl, qsl, scale, *scales).float()
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("V", [32000, 128256])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_sampler_matches_reference_with_shared_noise(V, dtype):
    B = 12
    g = torch.Generator().manual_seed(31)
    logits = (torch.randn(B, V, generator=g) * 3).to(dtype).to(DEV)
    temp = torch.tensor([0.0, 0.7, 1.0, 1.3, 0.7, 0.7, 1.0, 0.0, 0.9, 1.0, 0.5, 2.0], device=DEV)
    top_p = torch.tensor([1.0, 1.0, 0.9, 0.5, 1.0, 0.95, 0.1, 0.9, 1.0, 0.8, 0.99, 0.7], device=DEV)
    top_k = torch.tensor([0, 0, 0, 0, 40, 50, 0, 5, 1, 1000, 0, 3], dtype=torch.int32, device=DEV)
    seeds = torch.arange(B, dtype=torch.int64, device=DEV)
    steps = torch.zeros(B, dtype=torch.int64, device=DEV)
    agree = 0
    rounds = 6
    for r in range(rounds):
        noise = torch.rand(B, V, generator=g).clamp_(1e-10, 1 - 1e-7).to(DEV)
        got = ops.sample(logits, temp, top_p, top_k, seeds, steps, noise=noise)
        want = ref.sample_ref(logits, temp, top_p, top_k.long(), uniform_noise=noise)
        agree += int((got == want).sum())
        assert (got[[0, 7]] == logits.float().argmax(-1)[[0, 7]]).all()          # greedy rows
        assert logits[8, got[8]] == logits[8].max()                                 # top_k = 1 (bf16 ties allowed)
    assert agree >= rounds * B - 2, f"only {agree}/{rounds * B} draws agree with the reference"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_sampler_nan_row_yields_a_valid_token(dtype):
    """A NaN row (a bug upstream) must not turn into an out-of-range token id: the next step embeds it."""
    B, V = 6, 128256
    logits = torch.randn(B, V, device=DEV).to(dtype)
    logits[1] = float("nan")
    logits[4] = float("nan")
    temp = torch.tensor([0.0, 0.0, 0.7, 1.0, 0.7, 1.0], device=DEV)
    top_p = torch.tensor([1.0, 1.0, 1.0, 0.9, 0.9, 1.0], device=DEV)
    top_k = torch.tensor([0, 0, 0, 40, 40, 0], dtype=torch.int32, device=DEV)
    seeds = torch.arange(B, dtype=torch.int64, device=DEV)
    steps = torch.zeros(B, dtype=torch.int64, device=DEV)
    got = ops.sample(logits, temp, top_p, top_k, seeds, steps)
    assert ((got >= 0) & (got < V)).all(), got


def test_sampler_philox_distribution_and_determinism():
    V, N = 4096, 4096
    probs = torch.tensor([0.5, 0.25, 0.125, 0.125])
    logits = torch.full((N, V), -30.0)
    logits[:, :4] = probs.log()
    logits = logits.to(DEV)
    one = torch.ones(N, device=DEV)
    zk = torch.zeros(N, dtype=torch.int32, device=DEV)
    seeds = torch.full((N,), 1234, dtype=torch.int64, device=DEV)
    steps = torch.arange(N, dtype=torch.int64, device=DEV)
    a = ops.sample(logits, one, one, zk, seeds, steps)
    b = ops.sample(logits, one, one, zk, seeds, steps)
    assert (a == b).all()                                           # same (seed, step) → same draw
    freq = torch.bincount(a.cpu(), minlength=4)[:4].float() / N
    assert (freq - probs).abs().max() < 0.03
    c = ops.sample(logits, one, one * 0.7, zk, seeds, steps)        # nucleus 0.7 → tokens {0, 1} only
    assert set(c.unique().tolist()) <= {0, 1}


def test_copy_pages():
    kc, _ = _caches(8, 2)
    kc.copy_(torch.randn_like(kc.float()).to(torch.bfloat16))
    want = kc.clone()
    want[5] = want[1]
    want[6] = want[2]
    ops.copy_pages(kc, torch.tensor([[1, 5], [2, 6]], dtype=torch.int32, device=DEV))
    assert torch.equal(kc, want)


@pytest.mark.parametrize("T,H,I,E,El,off", [(1, 512, 1024, 4, 4, 0), (300, 4096, 1792, 8, 8, 0), (77, 1024, 448, 8, 2, 4),
                                             # decode-sized: the weight-streaming path (T ≤ 16, I % 128 == 0)
                                             (3, 4096, 1792, 8, 8, 0), (8, 1024, 14336, 8, 8, 0), (16, 4096, 1792, 8, 8, 0),
                                             (16, 512, 256, 8, 2, 4), (5, 512, 256, 8, 1, 3), (16, 512, 256, 2, 2, 0),
                                             (32, 512, 256, 2, 2, 0),          # 32 rows per expert: two passes each
                                             (32, 4096, 1792, 8, 8, 0), (24, 512, 256, 8, 1, 3),
                                             (12, 512, 448, 8, 8, 0),          # I % 128 != 0 → grouped GEMM
                                             (1000, 512, 256, 8, 1, 3),
                                             (1500, 4096, 448, 8, 8, 0)])     # last: enough row tiles → no split-K
def test_fused_moe_matches_reference(T, H, I, E, El, off):
    x = _bf(T, H, seed=41)
    logits = _bf(T, E, seed=42, scale=2.0)
    w13 = _bf(El, 2 * I, H, seed=43, scale=0.03)
    w2 = _bf(El, H, I, seed=44, scale=0.03)
    got = ops.fused_moe(x, logits, w13, w2, 2, off).float()
    want = ref.moe_ref(x.float(), logits.float(), w13.float(), w2.float(), 2, off).float()
    assert torch.isfinite(got).all()
    err = (got - want).abs().max() / want.abs().max().clamp_min(1e-6)
    assert err < 3e-2, err
    # tokens routed only to non-local experts contribute exactly zero
    if El < E:
        probs = torch.softmax(logits.float(), -1)
        top = probs.topk(2, -1).indices
        none_local = ((top < off) | (top >= off + El)).all(-1)
        assert got[none_local].abs().max() == 0 if none_local.any() else True


@pytest.mark.parametrize("T,H,I,E,El,off", [(1600, 1024, 512, 8, 8, 0), (2100, 512, 256, 8, 2, 4), (900, 512, 256, 2, 2, 0),
                                             (700, 4096, 1792, 8, 1, 5)])
def test_fused_moe_256_row_tiles_match_reference(T, H, I, E, El, off):
    """≥ 384 routed rows per local expert: the 256×256×64 tile kernel (gemm256.hip) runs both grouped GEMMs."""
    x = _bf(T, H, seed=51)
    logits = _bf(T, E, seed=52, scale=2.0)
    w13 = _bf(El, 2 * I, H, seed=53, scale=0.03)
    w2 = _bf(El, H, I, seed=54, scale=0.03)
    got = ops.hip().fused_moe(x, logits, w13, w2, 2, off, 256).float()
    small = ops.hip().fused_moe(x, logits, w13, w2, 2, off, 128).float()
    want = ref.moe_ref(x.float(), logits.float(), w13.float(), w2.float(), 2, off).float()
    scale = want.abs().max().clamp_min(1e-6)
    assert torch.isfinite(got).all()
    assert (got - want).abs().max() / scale < 3e-2
    assert (got - small).abs().max() / scale < 1.5e-2          # the two tilings agree to bf16 rounding of the outputs
    auto = ops.fused_moe(x, logits, w13, w2, 2, off).float()   # the dispatch rule picks the 256-row tiles here
    assert torch.equal(auto, got)


@pytest.mark.parametrize("counts,H,I", [([384, 385, 511, 640], 512, 256),            # last tiles of 128 / 129 / 255 / 128 rows
                                        ([257, 300, 383, 384, 512, 520], 1024, 512),   # 1, 44, 127, 128, 0 (no tail), 8
                                        ([385, 449, 100], 4096, 1792),                 # Mixtral's shapes; an expert with one half tile only
                                        ([128, 127, 129], 768, 384)])                  # nothing but last tiles; 12 and 6 K-steps
def test_fused_moe_256_row_half_tiles_match_reference(counts, H, I):
    """An expert's last 256-row tile with ≤ 128 routed rows runs as a HALF tile (gemm256.hip: the wave groups take 64 rows
    each and the a1 phases of the K-step run empty); routing constructed so that both kinds of last tile
    occur, at the boundary row counts.  Checked against the fp32 reference and against the 128-row tiling."""
    E = len(counts)
    left, pairs = list(counts), []
    while sum(left):
        a, b = sorted(range(E), key=lambda e: -left[e])[:2]
        pairs.append((a, b))
        left[a] -= 1
        left[b] -= 1
    assert min(left) == 0 and max(left) == 0
    T = len(pairs)
    logits = torch.full((T, E), -4.0, device=DEV)
    logits.scatter_(1, torch.tensor(pairs, device=DEV), torch.tensor([[3.0, 2.5]], device=DEV).expand(T, 2))
    logits = logits.bfloat16()
    assert torch.bincount(torch.topk(logits.float(), 2, -1).indices.flatten(), minlength=E).tolist() == counts
    x = _bf(T, H, seed=55)
    w13 = _bf(E, 2 * I, H, seed=56, scale=0.03)
    w2 = _bf(E, H, I, seed=57, scale=0.03)
    got = ops.hip().fused_moe(x, logits, w13, w2, 2, 0, 256).float()
    small = ops.hip().fused_moe(x, logits, w13, w2, 2, 0, 128).float()
    want = ref.moe_ref(x.float(), logits.float(), w13.float(), w2.float(), 2, 0).float()
    scale = want.abs().max().clamp_min(1e-6)
    assert torch.isfinite(got).all()
    assert (got - want).abs().max() / scale < 3e-2
    assert (got - small).abs().max() / scale < 1.5e-2
    # the alignment places rows in index order (no atomics) and a half tile sums in a full tile's order: every launch
    # must give the same bits (the first half-tile version split K between the wave groups and did not)
    for _ in range(6):
        assert torch.equal(ops.hip().fused_moe(x, logits, w13, w2, 2, 0, 256).float(), got)


@pytest.mark.parametrize("M,N,K", [(256, 512, 256), (1000, 1024, 4096), (4096, 256, 512), (1, 256, 64), (300, 768, 1024),
                                   (513, 256, 192)])
def test_dense_gemm256_matches_fp32_reference(M, N, K):
    x = _bf(M, K, seed=61)
    w = _bf(N, K, seed=62, scale=0.05)
    got = ops.hip().dense_gemm256(x, w, False).float()
    want = x.float() @ w.float().t()
    assert got.shape == (M, N) and torch.isfinite(got).all()
    assert (got - want).abs().max() / want.abs().max() < 1e-2


@pytest.mark.parametrize("M,I,K", [(256, 256, 256), (1111, 1792, 4096), (17, 128, 64), (2048, 14336, 512)])
def test_dense_gemm256_swiglu_epilogue(M, I, K):
    x = _bf(M, K, seed=63)
    w = _bf(2 * I, K, seed=64, scale=0.05)
    got = ops.hip().dense_gemm256(x, w, True).float()
    h = x.float() @ w.float().t()
    want = torch.nn.functional.silu(h[:, :I]) * h[:, I:]
    assert got.shape == (M, I) and torch.isfinite(got).all()
    assert (got - want).abs().max() / want.abs().max() < 1.5e-2


@pytest.mark.parametrize("M,N,K,ks", [(128, 512, 1024, 4), (300, 4096, 14336, 16), (64, 256, 512, 2), (511, 1280, 8192, 8)])
def test_dense_gemm256_split_k_matches_fp32_reference(M, N, K, ks):
    """Mid-M path: K cut into slices, fp32 partial slabs, reduce pass — rows past M never stored,
    uneven slices (K / 64 not divisible by the split count) covered."""
    x = _bf(M, K, seed=71)
    w = _bf(N, K, seed=72, scale=0.05)
    got = ops.hip().dense_gemm256(x, w, False, ks).float()
    want = x.float() @ w.float().t()
    assert got.shape == (M, N) and torch.isfinite(got).all()
    assert (got - want).abs().max() / want.abs().max() < 1e-2
    one = ops.hip().dense_gemm256(x, w, False, 1).float()
    assert (got - one).abs().max() / want.abs().max() < 1e-2


@pytest.mark.parametrize("M,I,K,ks", [(128, 256, 1024, 4), (200, 3584, 8192, 8), (33, 128, 320, 5)])
def test_dense_gemm256_split_k_swiglu(M, I, K, ks):
    x = _bf(M, K, seed=73)
    w = _bf(2 * I, K, seed=74, scale=0.05)
    got = ops.hip().dense_gemm256(x, w, True, ks).float()
    h = x.float() @ w.float().t()
    want = torch.nn.functional.silu(h[:, :I]) * h[:, I:]
    assert got.shape == (M, I) and torch.isfinite(got).all()
    assert (got - want).abs().max() / want.abs().max() < 1.5e-2


@pytest.mark.parametrize("M,N,K,swiglu", [(2816, 6144, 4096, False), (1792, 28672, 4096, True), (1200, 28672, 4096, True),
                                           (2100, 8192, 1024, False), (512, 4096, 4096, False)])
def test_dense_gemm256_split_tail_matches_fp32_reference_and_the_plain_tiles(M, N, K, swiglu):
    """Whole rounds of one tile per CU as plain tiles, the remainder tiles K-split into compact fp32 slabs + a per-tile
    reduce (`gemm256_kernel` RANGE mode, `splitk_reduce_tiles_kernel`): every output element is written exactly once (the
    result buffer starts as NaN), ragged last row tiles included; the last case has no remainder to cut (plain path)."""
    from dli_amd.ops.gemm_plan import GemmPlan
    x = _bf(M, K, seed=75)
    w = _bf(N, K, seed=76, scale=0.05)
    n_out = N // 2 if swiglu else N
    out = torch.full((M, n_out), float("nan"), device=DEV, dtype=torch.bfloat16)
    got = ops.hip().dense_gemm256(x, w, swiglu, 1, out, True).float()
    h = x.float() @ w.float().t()
    want = torch.nn.functional.silu(h[:, :n_out]) * h[:, n_out:] if swiglu else h
    assert torch.isfinite(got).all()
    assert (got - want).abs().max() / want.abs().max() < 1.5e-2
    plain = ops.hip().dense_gemm256(x, w, swiglu, 1).float()
    assert (got - plain).abs().max() / want.abs().max() < 1e-2
    cut = GemmPlan.tail_slices(M, N, K, swiglu)
    assert (cut >= 2) == (M != 512)
    if not cut:
        assert torch.equal(got, plain)
    else:                                                   # the whole rounds are the very same tiles: most rows × columns bit-equal
        assert (got == plain).float().mean() > 0.5


@pytest.mark.parametrize("M,K,silu", [(70, 4096, False), (300, 14336, False), (5, 256, False), (129, 14336, True), (33, 28672, False),
                                       (64, 3584, True)])
def test_quant_rows_fp8_matches_reference(M, K, silu):
    """csrc/kernels/quant.hip: per-row scale = amax / 448, values = e4m3(x / scale) — against the torch composite on the same
    input (scales to fp32 rounding, codes equal except where x / scale sits on a rounding boundary); rows longer than the
    register-resident 16384 and the fused SwiGLU form included."""
    x = _bf(M, (2 if silu else 1) * K, seed=81, scale=2.0)
    x[0] = 0                                                  # an all-zero row must not produce NaN / inf
    q, s = ops.hip().quant_rows_fp8(x, silu)
    assert q.dtype == torch.float8_e4m3fn and q.shape == (M, K) and s.shape == (M, 1) and s.dtype == torch.float32
    xf = x.float()
    if silu:
        xf = torch.nn.functional.silu(xf[:, :K]) * xf[:, K:]
    want_s = xf.abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / 448.0
    assert torch.allclose(s[1:], want_s[1:], rtol=2e-2 if silu else 1e-6)            # silu: fast exp2 / rcp in the kernel
    deq = q.float() * s
    assert torch.isfinite(deq).all() and (deq[0] == 0).all()
    err = (deq - xf).abs() / xf.abs().amax(dim=1, keepdim=True).clamp_min(1e-12)
    assert err.max() < 0.04, err.max()                        # half an e4m3 step at the top of the range is 1/32 of amax
    if not silu:
        want_q = (xf / want_s).clamp(-448, 448).to(torch.float8_e4m3fn)
        assert (q.view(torch.uint8) == want_q.view(torch.uint8)).float().mean() > 0.999


@pytest.mark.parametrize("T,H,add", [(70, 4096, True), (300, 8192, True), (33, 1024, False), (5, 4096, True)])
def test_rms_norm_quant_equals_norm_then_quant(T, H, add):
    """W8A8 option: (residual add +) RMSNorm + per-row e4m3 quantisation in ONE kernel against the two kernels it replaces
    — same residual, same scales, same codes (the fused kernel quantises the bf16-rounded normalised row)."""
    x = _bf(T, H, seed=91)
    w = (1 + 0.1 * torch.randn(H, device=DEV)).to(torch.bfloat16)
    res = _bf(T, H, seed=92) if add else None
    r1 = res.clone() if add else None
    q, s = ops.hip().rms_norm_quant(x, r1, w, 1e-5)
    y = x.clone()
    if add:
        r2 = res.clone()
        ops.fused_add_rms_norm(y, r2, w, 1e-5)
        assert torch.equal(r1, r2)
    else:
        y = ops.rms_norm(x, w, 1e-5)
    q2, s2 = ops.hip().quant_rows_fp8(y, False)
    assert torch.equal(s, s2) and torch.equal(q.view(torch.uint8), q2.view(torch.uint8))
    assert torch.equal(x, _bf(T, H, seed=91))                 # the input is left alone


@pytest.mark.parametrize("M,N,K,swiglu", [(1, 6144, 4096, False), (3, 4096, 14336, False), (8, 28672, 4096, True), (16, 28672, 4096, True),
                                           (13, 1280, 8192, False), (4, 128256, 4096, False), (5, 7168, 1024, True), (2, 48, 384, False)])
def test_skinny_linear_w8_matches_the_dequantised_product(M, N, K, swiglu):
    """W8A16 decode (`skinny_linear_kernel<…, W8>`): e4m3 weights streamed through LDS, widened to bf16 in registers (exact),
    bf16 MFMA, per-output-channel scale in the epilogue — against the fp32 product with the dequantised weights."""
    x = _bf(M, K, seed=85)
    w = _bf(N, K, seed=86, scale=0.05)
    wq, ws = ops.quant_weight_fp8(w)
    got = ops.linear_w8(x, wq, ws, swiglu=swiglu).float()
    h = x.float() @ (wq.float() * ws.t()).t()
    want = torch.nn.functional.silu(h[:, :N // 2]) * h[:, N // 2:] if swiglu else h
    assert got.shape == want.shape and torch.isfinite(got).all()
    assert (got - want).abs().max() / want.abs().max() < 1.5e-2
    ref = x.float() @ w.float().t()                           # and the bf16 weights' product within e4m3 noise
    ref = torch.nn.functional.silu(ref[:, :N // 2]) * ref[:, N // 2:] if swiglu else ref
    assert torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item() > 0.997


@pytest.mark.parametrize("M,N,K", [(2304, 6144, 4096), (70, 4096, 14336), (300, 28672, 4096)])
def test_linear_fp8_tracks_fp32_reference(M, N, K):
    """W8A8 GEMM (per-token × per-channel scales, fp8 matrix cores through torch._scaled_mm) against the fp32 product of the
    bf16 operands, and against the fp32 product of the SAME quantised operands (that one tightly: the GEMM itself is exact
    up to accumulation order)."""
    x = _bf(M, K, seed=83)
    w = _bf(N, K, seed=84, scale=0.05)
    wq, ws = ops.quant_weight_fp8(w)
    got = ops.linear_fp8(x, wq, ws).float()
    ref = x.float() @ w.float().t()
    assert got.shape == (M, N) and torch.isfinite(got).all()
    assert torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item() > 0.998
    q, s = ops.quant_rows_fp8(x)
    same = (q.float() * s) @ (wq.float() * ws.t()).t()
    assert (got - same).abs().max() / same.abs().max() < 1e-2


def test_dense_gemm256_identity_check_with_asymmetric_weights():
    """A = I with an asymmetric W: catches a transposed or permuted C write that symmetric data would hide."""
    K = N = 256
    x = torch.eye(K, device=DEV, dtype=torch.bfloat16)
    w = (torch.arange(N * K, device=DEV, dtype=torch.float32).reshape(N, K) % 251 - 125).to(torch.bfloat16)
    got = ops.hip().dense_gemm256(x, w, False)
    assert torch.equal(got.float(), w.float().t())


@pytest.mark.parametrize("M,N,K,bias", [(1, 4096, 4096, False), (3, 6144, 4096, True), (16, 28672, 4096, False),
                                         (7, 4096, 14336, False), (16, 128256, 4096, False), (5, 48, 384, True),
                                         (2, 1024, 3584, False)])
def test_skinny_linear_matches_fp32_reference(M, N, K, bias):
    """K10 (csrc/kernels/skinny_gemm.hip): decode-sized x·Wᵀ against the fp32 product of the same bf16 operands."""
    torch.manual_seed(M * 131 + N)
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16() if bias else None
    y = ops.hip().skinny_linear(x, w, b)
    want = x.float() @ w.float().t() + (b.float() if bias else 0.0)
    assert y.shape == (M, N) and y.dtype == torch.bfloat16
    err = (y.float() - want).abs().max().item()
    assert err <= 0.01 * want.abs().max().item() + 1e-3, err
    # the dispatcher takes the kernel for these shapes; beyond 16 rows the kernel (small weights) or the library GEMM
    assert torch.equal(ops.linear(x, w, b), y)
    x17 = torch.randn(17, K, device="cuda").bfloat16()
    torch.testing.assert_close(ops.linear(x17, w, b).float(), x17.float() @ w.float().t() + (b.float() if bias else 0.0),
                               rtol=2e-2, atol=2e-2 * want.abs().max().item())


@pytest.mark.parametrize("M,N,K,swiglu", [(17, 4096, 1024, False), (40, 1280, 8192, False), (64, 256, 512, False),
                                           (33, 512, 1024, True)])
def test_skinny_linear_multi_pass(M, N, K, swiglu):
    """More than 16 tokens: 16-token passes over the same weight rows."""
    torch.manual_seed(M)
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    y = ops.hip().skinny_linear(x, w, None, swiglu).float()
    want = x.float() @ w.float().t()
    if swiglu:
        want = torch.nn.functional.silu(want[:, :N // 2]) * want[:, N // 2:]
    torch.testing.assert_close(y, want, rtol=2e-2, atol=2e-2 * want.abs().max().item())


def test_skinny_linear_strided_rows_and_rejects_bad_shapes():
    x_full = torch.randn(4, 2 * 512, device="cuda").bfloat16()
    x = x_full[:, :512]                                        # row stride 1024, contiguous rows
    w = (torch.randn(64, 512, device="cuda") * 0.05).bfloat16()
    torch.testing.assert_close(ops.hip().skinny_linear(x, w, None).float(), x.float() @ w.float().t(), rtol=2e-2, atol=2e-2)
    with pytest.raises(RuntimeError):
        ops.hip().skinny_linear(torch.randn(65, 512, device="cuda").bfloat16(), w, None)
    with pytest.raises(RuntimeError):
        ops.hip().skinny_linear(torch.randn(2, 192, device="cuda").bfloat16(), (torch.randn(64, 192, device="cuda")).bfloat16(), None)


@pytest.mark.parametrize("M,I,K", [(1, 14336, 4096), (6, 1792, 4096), (16, 14336, 4096), (3, 64, 256)])
def test_skinny_linear_swiglu_epilogue(M, I, K):
    """Fused [gate; up] projection with SwiGLU in the epilogue against fp32 silu(x·Wgᵀ)·(x·Wuᵀ)."""
    torch.manual_seed(M + I)
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(2 * I, K, device="cuda") * 0.03).bfloat16()
    y = ops.linear_swiglu(x, w)
    gu = x.float() @ w.float().t()
    want = torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]
    assert y.shape == (M, I)
    torch.testing.assert_close(y.float(), want, rtol=2e-2, atol=2e-2 * want.abs().max().item())
    # beyond 16 rows: library GEMM + activation kernel, same function
    x2 = torch.randn(20, K, device="cuda").bfloat16()
    gu2 = x2.float() @ w.float().t()
    want2 = torch.nn.functional.silu(gu2[:, :I]) * gu2[:, I:]
    torch.testing.assert_close(ops.linear_swiglu(x2, w).float(), want2, rtol=2e-2, atol=2e-2 * want2.abs().max().item())
