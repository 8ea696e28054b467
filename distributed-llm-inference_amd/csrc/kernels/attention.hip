// K4 / K5: paged attention for gfx950 — single-query decode (split-KV) and chunked-prefill
// (varlen, causal, paged prefix) — on v_mfma_f32_16x16x32_bf16.
//
// Both kernels use the *transposed* formulation so that nothing ever moves between lanes:
//
//   Sᵀ[token, qcol] = K[token, :] · Qᵀ[:, qcol]      A = K page rows (tokens), B = Qᵀ
//   Oᵀ[d, qcol]    += Vᵀ[d, token] · Pᵀ[token, qcol]  A = Vᵀ page rows (d),    B = Pᵀ
//
// A lane (qcol = lane & 15, g = lane >> 4) therefore owns one query column for the whole kernel: its
// Sᵀ accumulators are 4 tokens of that column, the online-softmax state (m, l) is per lane, and the
// exponentiated accumulators of two 16-token tiles are — after a bf16 pack — exactly the B fragment
// of the P·V MFMA, because the V page stores its 32 tokens in that register order
// (ops/kv_layout.py).  K and V fragments are 16-byte loads straight from the page to VGPRs: every
// wave-level load is one contiguous 1 KiB, each byte of the cache is read once per consumer wave,
// so decode needs no LDS staging (memory-bound GEMV-like regime: every wave walks different pages).
// Prefill tiles share their pages between four waves and stage them through LDS (see below).
//
// qcol meaning:  decode  → query head inside the GQA group (G <= 16, padded columns are zero)
//                prefill → query token inside a 16-token tile, one (head, tile) "column group" per
//                          accumulator set, two column groups per wave.
//
// KV8: pages hold OCP e4m3 (K chunks are 64 wide, see kv_layout); values are widened to bf16 in
// registers, the per-tensor scales are folded into the softmax scale (K) and the epilogue (V).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace dli {

constexpr int HD = 128;                 // head dim
constexpr int PAGE_ELEMS = HD * 32;     // elements per (page, kv head)
constexpr float NEG_BIG = -1.0e30f;


// ---- fragment loads ------------------------------------------------------------------------------
// K fragment for row tile rt (tokens 16rt..16rt+15) and k-step c (0..3) of lane (col, g).
template <bool KV8>
struct KLoader {
  // bf16: page [4][32 tok][32]:  element offset ((c*32 + rt*16 + col)*32 + 8g)
  // fp8 : page [2][32 tok][64]:  byte offset ((c2*32 + rt*16 + col)*64 + 16g) holds k-steps 2c2, 2c2+1
  // NT: streaming (non-temporal) loads — decode reads every K/V byte exactly once per step
  template <bool NT = false>
  static __device__ __forceinline__ void load(const void* page, int col, int g, bf16x8 (&kf)[2][4]) {
    if constexpr (!KV8) {
      const uint16_t* p = reinterpret_cast<const uint16_t*>(page) + col * 32 + g * 8;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bf16x8* src = reinterpret_cast<const bf16x8*>(p + (c * 32 + rt * 16) * 32);
          kf[rt][c] = NT ? __builtin_nontemporal_load(src) : *src;
        }
    } else {
      const uint8_t* p = reinterpret_cast<const uint8_t*>(page) + col * 64 + g * 16;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const u32x4* src = reinterpret_cast<const u32x4*>(p + (c2 * 32 + rt * 16) * 64);
          const u32x4 raw = NT ? __builtin_nontemporal_load(src) : *src;
          const uint32_t w[4] = {raw[0], raw[1], raw[2], raw[3]};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            bf16x8 f;
            fp8x4_to_bf16(w[2 * h], f, 0);
            fp8x4_to_bf16(w[2 * h + 1], f, 4);
            kf[rt][2 * c2 + h] = f;
          }
        }
    }
  }
  // Matching d index of element j of the B (Qᵀ) fragment for k-step c of lane group g.
  static __device__ __forceinline__ int q_d0(int c, int g) {
    if constexpr (!KV8) return 32 * c + 8 * g;
    else return 64 * (c >> 1) + 16 * g + 8 * (c & 1);
  }
};

// V fragment for d tile n (rows 16n..16n+15) of lane (col, g): slots 8g..8g+7 of row d = 16n + col.
template <bool KV8, bool NT = false>
__device__ __forceinline__ bf16x8 load_v(const void* page, int n, int col, int g) {
  if constexpr (!KV8) {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(reinterpret_cast<const uint16_t*>(page) + (16 * n + col) * 32 + 8 * g);
    return NT ? __builtin_nontemporal_load(src) : *src;
  } else {
    const u32x2* src = reinterpret_cast<const u32x2*>(reinterpret_cast<const uint8_t*>(page) + (16 * n + col) * 32 + 8 * g);
    const u32x2 raw = NT ? __builtin_nontemporal_load(src) : *src;
    bf16x8 f;
    fp8x4_to_bf16(raw[0], f, 0);
    fp8x4_to_bf16(raw[1], f, 4);
    return f;
  }
}

// Reductions over the 4 lanes (l, l^16, l^32, l^48) that share a query column, with the gfx950 row-swap VALU
// instructions instead of LDS-crossbar shuffles.  v_permlane16_swap(x, x) returns (rows 0,0,2,2 / rows 1,1,3,3) of
// x, v_permlane32_swap(x, x) returns (upper half twice / lower half twice): in every lane one of the two results
// is the lane's own value and the other its partner's, so max3 / a plain sum of the pair needs no select.
__device__ __forceinline__ float group_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(v, fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(v, fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1])));
}
__device__ __forceinline__ float group_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// One 32-token page against one query-column set: Sᵀ → online softmax → Pᵀ fragment.
// `limit`: tokens with index > limit are masked (limit = L-1 for decode, query position for prefill).
// The scale is applied inside the exponent (one fma per element: exp2(s·scale − m)); the running maximum m is
// kept in scaled units.
template <bool MASK>
__device__ __forceinline__ bf16x8 softmax_step(f32x4 (&s)[2], int tok0, int limit, float scale_log2, float& m,
                                               float& lsum, float& alpha) {
  float raw = NEG_BIG;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = s[rt][r];
      if (MASK) v = (tok0 + rt * 16 + r <= limit) ? v : NEG_BIG;
      s[rt][r] = v;
      raw = fmaxf(raw, v);
    }
  const float mx = group_max(fmaxf(m, raw * scale_log2));
  alpha = fast_exp2(m - mx);
  m = mx;
  bf16x8 pb;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const f32x4 sc = {scale_log2, scale_log2, scale_log2, scale_log2}, nm = {-mx, -mx, -mx, -mx};
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const f32x4 e = __builtin_elementwise_fma(s[rt], sc, nm);            // v_pk_fma_f32: two elements per instruction
    f32x4 p;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = fast_exp2(e[r]);
      if (MASK) p[r] = (tok0 + rt * 16 + r <= limit) ? p[r] : 0.f;
      pb[rt * 4 + r] = (short)f2bf(p[r]);
    }
    acc += p;                                                            // v_pk_add_f32
  }
  const float ps = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  lsum = lsum * alpha + ps;
  return pb;
}

// ==================================================================================================
// Decode: 4 waves per workgroup; wave w takes pages w, w+4, … of the workgroup's page range of one (sequence, kv head).
// The kv head is the fastest grid index: the 8 heads of a sequence (contiguous 64 KB per page) are read at the same
// time by neighbouring workgroups, and a long sequence's work is spread over 8 workgroups that start together instead
// of 8 that start num_seqs ids apart: +3–4 % on mixed-length batches (profiles/kernels_decode_r2aa.txt).
//
// Two ways to say which pages a workgroup reads:
//   WL = true   work list, grid (Hkv, items): `work[item] = {sequence, first page, end page, partial slot}` built by the
//               host per step (ops.build_decode_work) — page ranges of about equal size whatever the sequence's length,
//               so a 2048-token row and a 40-token row no longer get the same split count and the tail of the launch is
//               one chunk long.  slot < 0: the range is the whole sequence, the result goes straight to `out`.  The
//               number of valid items is read from DEVICE memory (`counts[0]`; extra workgroups exit), which is what
//               lets a hipGraph with a fixed grid serve any batch.
//   WL = false  uniform splits, grid (Hkv, num_seqs, num_splits): every sequence is cut into `num_splits` ranges.
// Partial results: `part_o[(p·Hq + head)·128 + d]`, `part_ml[(p·Hq + head)·2 + {m, l}]` with p = slot (WL) or
// seq·num_splits + split; decode_reduce_kernel combines them per sequence.
// ==================================================================================================
template <bool KV8, bool WL, bool NT = false>
__global__ __launch_bounds__(256) void paged_decode_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const void* __restrict__ k_cache, const void* __restrict__ v_cache,
    const int* __restrict__ block_tables, int max_blocks, const int* __restrict__ seq_lens, uint16_t* __restrict__ out,
    int64_t out_stride, float* __restrict__ part_o, float* __restrict__ part_ml, int Hq, int Hkv, int num_splits,
    float scale_log2, float v_scale, const float* __restrict__ kv_scales, const int4* __restrict__ work,
    const int* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int kvh = blockIdx.x;
  int seq, b_begin, b_end, part;                           // part < 0: write `out` directly
  if constexpr (WL) {
    if ((int)blockIdx.y >= counts[0]) return;
    const int4 w = work[blockIdx.y];
    seq = w.x, b_begin = w.y, b_end = w.z, part = w.w;
  } else {
    seq = blockIdx.y;
    part = num_splits == 1 ? -1 : seq * num_splits + (int)blockIdx.z;
  }
  if (KV8 && kv_scales != nullptr) {                       // calibrated per-(layer, kv head) fp8 scales: [k | v][Hkv]
    scale_log2 *= kv_scales[kvh];
    v_scale = kv_scales[Hkv + kvh];
  }
  const int G = Hq / Hkv;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: page loop, block-table reads and page bases in SGPRs
  const int col = lane & 15, g = lane >> 4;
  const int L = seq_lens[seq];
  const int nblk = (L + 31) >> 5;
  if constexpr (!WL) {
    const int per = (nblk + num_splits - 1) / num_splits;
    b_begin = (int)blockIdx.z * per;
    b_end = min(nblk, b_begin + per);
  } else {
    b_end = min(b_end, nblk);
  }
  constexpr int ESZ = KV8 ? 1 : 2;

  bf16x8 qf[4];
  {
    const bool ok = col < G;
    const uint16_t* qp = q + (int64_t)seq * q_stride + (kvh * G + (ok ? col : 0)) * HD;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bf16x8 v = *reinterpret_cast<const bf16x8*>(qp + KLoader<KV8>::q_d0(c, g));
      qf[c] = ok ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  f32x4 o[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = NEG_BIG, lsum = 0.f;

  const int* bt = block_tables + (int64_t)seq * max_blocks;
  const char* kc = reinterpret_cast<const char*>(k_cache);
  const char* vc = reinterpret_cast<const char*>(v_cache);

  // the block-table entry of the NEXT page of this wave is fetched one iteration ahead: the entry → page address → 16
  // fragment loads chain is otherwise paid in full at the top of every iteration
  int blk_next = b_begin + wave < b_end ? bt[b_begin + wave] : 0;
  for (int b = b_begin + wave; b < b_end; b += 4) {
    const int blk_cur = blk_next;
    if (b + 4 < b_end) blk_next = bt[b + 4];
    const int64_t page = ((int64_t)blk_cur * Hkv + kvh) * (PAGE_ELEMS * ESZ);
    bf16x8 kf[2][4];
    KLoader<KV8>::template load<NT>(kc + page, col, g, kf);
    bf16x8 vf[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) vf[n] = load_v<KV8, NT>(vc + page, n, col, g);

    f32x4 s[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      s[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) s[rt] = mfma16(kf[rt][c], qf[c], s[rt]);
    }
    float alpha;
    bf16x8 pb;
    const int tok0 = b * 32 + 4 * g;
    if (b == nblk - 1) pb = softmax_step<true>(s, tok0, L - 1, scale_log2, m, lsum, alpha);
    else pb = softmax_step<false>(s, tok0, L - 1, scale_log2, m, lsum, alpha);
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      o[n] *= alpha;
      o[n] = mfma16(vf[n], pb, o[n]);
    }
  }
  lsum = group_sum(lsum);

  // ---- merge the 4 waves through LDS: [wave][G][128 + 4] fp32 + [wave][G] (m, l).  Rows are padded by one 16-byte
  // slot: the G active lanes of a ds_write_b128 lane group hold the same d range of G different rows, and with
  // 512-byte rows they would all start on bank 0 (4-way conflict: 0.46 conflict cycles per LDS cycle in round 1).
  constexpr int HDP = HD + 4;
  float* sm_o = smem;
  float* sm_m = smem + 4 * G * HDP;
  float* sm_l = sm_m + 4 * G;
  if (col < G) {
    float* dst = sm_o + (wave * G + col) * HDP + 4 * g;
#pragma unroll
    for (int n = 0; n < 8; ++n) *reinterpret_cast<f32x4*>(dst + 16 * n) = o[n];
    if (g == 0) { sm_m[wave * G + col] = m; sm_l[wave * G + col] = lsum; }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < G * HD; idx += 256) {
    const int head = idx >> 7, d = idx & 127;
    float M = NEG_BIG;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, sm_m[w * G + head]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = fast_exp2(sm_m[w * G + head] - M);
      num += f * sm_o[(w * G + head) * HDP + d];
      den += f * sm_l[w * G + head];
    }
    const int qh = kvh * G + head;
    if (part < 0) {
      const float r = den > 0.f ? v_scale / den : 0.f;
      out[(int64_t)seq * out_stride + qh * HD + d] = f2bf(num * r);
    } else {
      const int64_t slot = (int64_t)part * Hq + qh;
      part_o[slot * HD + d] = num;
      if (d == 0) { part_ml[slot * 2] = M; part_ml[slot * 2 + 1] = den; }
    }
  }
}

// Combine the partials of one (sequence, query head): 128 threads, one per d.  Uniform splits: grid (num_seqs, Hq),
// partials seq·num_splits … ; work list: grid (reduce entries, Hq), `red[i] = {sequence, first partial, count}` and the
// number of valid entries in `counts[1]` (device memory, like the decode kernel's item count).
__global__ __launch_bounds__(128) void decode_reduce_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                            uint16_t* __restrict__ out, int64_t out_stride, int Hq,
                                                            int num_splits, float v_scale, const float* __restrict__ kv_scales,
                                                            int Hkv, const int4* __restrict__ red, const int* __restrict__ counts) {
  int seq = blockIdx.x, first = (int)blockIdx.x * num_splits, n = num_splits;
  if (red != nullptr) {
    if ((int)blockIdx.x >= counts[1]) return;
    const int4 r = red[blockIdx.x];
    seq = r.x, first = r.y, n = r.z;
  }
  const int qh = blockIdx.y, d = threadIdx.x;
  if (kv_scales != nullptr) v_scale = kv_scales[Hkv + qh / (Hq / Hkv)];
  float M = NEG_BIG;
  for (int s = 0; s < n; ++s) M = fmaxf(M, part_ml[((int64_t)(first + s) * Hq + qh) * 2]);
  float num = 0.f, den = 0.f;
  for (int s = 0; s < n; ++s) {
    const int64_t slot = (int64_t)(first + s) * Hq + qh;
    const float f = fast_exp2(part_ml[slot * 2] - M);
    num += f * part_o[slot * HD + d];
    den += f * part_ml[slot * 2 + 1];
  }
  const float r = den > 0.f ? v_scale / den : 0.f;
  out[(int64_t)seq * out_stride + qh * HD + d] = f2bf(num * r);
}

// ==================================================================================================
// Prefill: one workgroup of 4 waves per (tile, kv head).  A tile is TQ = 16·(8/G) consecutive query tokens of one sequence; its
// 8 column groups (head, 16-token sub-tile) are spread two per wave = 32 query columns per wave.  All K/V — paged
// prefix and the chunk itself — come from the cache (rope_and_cache has already written the chunk).
//
// Staging.  Unlike decode, all four waves of a tile walk the SAME pages, so K and V pages go through LDS: the block
// copies each 32-token page (K + V, 16 KB bf16 / 8 KB fp8) once by LDS-DMA (global_load_lds, 16 bytes per lane, no
// VGPRs) into a ring of three stages, two pages ahead of the math (one barrier per page, counted vmcnt), and every
// wave reads its fragments with ds_read_b128.  The page keeps its global layout in LDS except that the four 16-byte
// slots of every 64-byte row are permuted by the row (slot ^ h[(row >> 2) & 3], h = {0,2,3,1}; the DMA's LDS image is
// lane-linear, so the permutation sits on the SOURCE address): the 16 lanes the hardware serves together for a
// ds_read_b128 — {0–3, 12–15, 20–27} etc. — then touch 16 different slots of the 256-byte bank row.
//
// Math on v_mfma_f32_32x32x16_bf16: a wave's 32 query columns are ONE MFMA column set.  Lane (c = lane & 31,
// hi = lane >> 5) owns column c:
//   Sᵀ(32 tok × 32 col)  = 8 MFMAs over d (k-step j: d = 16 j + 8 hi …); the lane's 16 accumulators are tokens
//                          (r & 3) + 8 (r >> 2) + 4 hi of the page;
//   P fragments          : accumulators 8 jj … 8 jj + 7 are — after exp2 and a bf16 pack — the B fragment of k-step jj
//                          of P·V, because slots 8 hi + 4 jj … and 16 + 8 hi + 4 jj … of a V page row hold exactly
//                          those tokens (v_slot, ops/kv_layout.py — the layout serves both MFMA shapes): one
//                          ds_read_b128 of slot hi and one of slot 2 + hi give the A fragments of both k-steps;
//   Oᵀ(128 d × 32 col)   = 4 d-tiles × 2 k-steps = 8 MFMAs.
// Round 1 used the 16×16×32 shape here as decode does (two 16-column sets per wave): 32 MFMA instructions per page
// instead of 16, two online-softmax states and four row-swap reductions per lane instead of one and one.  Where the
// time went BEFORE the XCD-aware block mapping below (ablations, L = 4096: docs/KERNELS.md): softmax VALU 36 %,
// issuing — and waiting on — the 4 LDS-DMA instructions per wave and page 24 %, P·V 23 %; the per-page barrier is free.
// ==================================================================================================
__device__ __forceinline__ int slot_swizzle(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

template <bool MASK>
__device__ __forceinline__ void softmax32(f32x16& s, int tok0, int limit, float scale_log2, float& m, float& lsum,
                                          float& alpha, bf16x8 (&pb)[2]) {
  float raw = NEG_BIG;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = s[r];
    if (MASK) v = (tok0 + (r & 3) + 8 * (r >> 2) <= limit) ? v : NEG_BIG;
    s[r] = v;
    raw = fmaxf(raw, v);
  }
  float mx = fmaxf(m, raw * scale_log2);
  auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
  mx = fmaxf(mx, fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
  alpha = fast_exp2(m - mx);
  m = mx;
  float ps = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float p = fast_exp2(__builtin_fmaf(s[r], scale_log2, -mx));
    if (MASK) p = (tok0 + (r & 3) + 8 * (r >> 2) <= limit) ? p : 0.f;
    ps += p;
    pb[r >> 3][r & 7] = (short)f2bf(p);
  }
  lsum = lsum * alpha + ps;
}

template <bool KV8>
__global__ __launch_bounds__(256, 2) void paged_prefill_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const void* __restrict__ k_cache, const void* __restrict__ v_cache,
    const int* __restrict__ block_tables, int max_blocks, const int* __restrict__ seq_lens,
    const int* __restrict__ query_start_loc, const int* __restrict__ tile_seq, const int* __restrict__ tile_q0,
    uint16_t* __restrict__ out, int64_t out_stride, int Hq, int Hkv, int num_tiles, float scale_log2, float v_scale,
    const float* __restrict__ kv_scales, const int* __restrict__ counts) {
  // inside a hipGraph the grid is an upper bound and the step's tile count lives in device memory (counts[2])
  if (counts != nullptr) num_tiles = counts[2];
  constexpr int ESZ = KV8 ? 1 : 2;
  constexpr int PAGE_BYTES = PAGE_ELEMS * ESZ;
  constexpr int NCH = PAGE_BYTES / (16 * 256);
  constexpr int NSTAGE = 3;
  constexpr int NDMA = 2 * NCH;
  __shared__ __attribute__((aligned(16))) char smem[NSTAGE * 2 * PAGE_BYTES];

  // Block → (kv head, tile).  Workgroups are dealt round-robin over the 8 XCDs (private L2s); tiles of one sequence and
  // kv head read the same K/V pages.  XCD x takes the x-th eighth of the order (kv head, tile): with 8 kv heads every
  // XCD serves ONE head — a sequence's pages of that head are fetched into one L2 instead of eight — and with fewer
  // heads (TP shards) a contiguous range of tiles.
  const int total = num_tiles * Hkv;
  const int per_xcd = (total + 7) >> 3;
  const int lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || lin >= total) return;
  const int kvh = lin / num_tiles;
  const int G = Hq / Hkv;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA destinations and the wave's page bounds in SGPRs
  const int c32 = lane & 31, hi = lane >> 5;
  const int tile_id = num_tiles - 1 - (lin - kvh * num_tiles);
  if (KV8 && kv_scales != nullptr) {
    scale_log2 *= kv_scales[kvh];
    v_scale = kv_scales[Hkv + kvh];
  }
  const int seq = tile_seq[tile_id], q0 = tile_q0[tile_id];
  const int qs = query_start_loc[seq];
  const int n_new = query_start_loc[seq + 1] - qs;
  const int L = seq_lens[seq];
  const int ctx = L - n_new;

  // this lane's column: column group cg = 2·wave + (c32 >> 4) is (head cg % G, 16-token sub-tile cg / G)
  const int cg = 2 * wave + (c32 >> 4);
  const int head = cg % G;
  const int qtok = q0 + (cg / G) * 16 + (c32 & 15);
  const bool valid = qtok < n_new;
  const int qpos = ctx + min(qtok, n_new - 1);
  int first_tok = 1 << 30, last_pos = -1;      // wave-uniform bounds over both column groups
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int t0 = q0 + ((2 * wave + i) / G) * 16;
    if (t0 < n_new) {
      first_tok = min(first_tok, ctx + t0);
      last_pos = max(last_pos, ctx + min(t0 + 15, n_new - 1));
    }
  }
  const int my_nblk = last_pos < 0 ? 0 : (last_pos >> 5) + 1;
  const int tile_tokens = 16 * (8 / G);
  const int nblk = ((ctx + min(q0 + tile_tokens, n_new) - 1) >> 5) + 1;

  // Qᵀ fragments: k-step j covers d = 16 j + 8 hi … + 7 (fp8 pages keep d in 64-wide chunks: same d, other address)
  bf16x8 qf[8];
  {
    const uint16_t* qp = q + (int64_t)(qs + min(qtok, n_new - 1)) * q_stride + (kvh * G + head) * HD + 8 * hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = *reinterpret_cast<const bf16x8*>(qp + 16 * j);
  }

  f32x16 o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
  float m = NEG_BIG, lsum = 0.f;

  const int* bt = block_tables + (int64_t)seq * max_blocks;
  const char* kc = reinterpret_cast<const char*>(k_cache);
  const char* vc = reinterpret_cast<const char*>(v_cache);

  int src_k[NCH], src_v[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ci = tid + 256 * i, row = ci >> 2, slot = ci & 3;
    src_k[i] = (row * 4 + (slot ^ slot_swizzle(row))) * 16;
    src_v[i] = KV8 ? ci * 16 : src_k[i];
  }
  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
#define DLI_STAGE(b_, buf_)                                                                                        \
  {                                                                                                                \
    const int64_t page_ = ((int64_t)bt[b_] * Hkv + kvh) * PAGE_BYTES;                                             \
    _Pragma("unroll") for (int i = 0; i < NCH; ++i) {                                                              \
      char* dst_ = smem + (buf_) * 2 * PAGE_BYTES + (wave + 4 * i) * 1024;                                         \
      __builtin_amdgcn_global_load_lds((gptr_t)(kc + page_ + src_k[i]), (lptr_t)dst_, 16, 0, 0);                   \
      __builtin_amdgcn_global_load_lds((gptr_t)(vc + page_ + src_v[i]), (lptr_t)(dst_ + PAGE_BYTES), 16, 0, 0);    \
    }                                                                                                              \
  }
  // Lane offsets inside a staged page.  Rows of 64 bytes (K: token rows of a d chunk; bf16 V: d rows) carry the slot
  // swizzle of their row; it depends on (row >> 2) & 3 only, and every row this lane reads is c32 + a multiple of 32.
  const int swz = slot_swizzle(c32);
  int k_off[2];                                   // bf16: logical slot 2 (j & 1) + hi;  fp8: slot j & 3, byte 8 hi inside
#pragma unroll
  for (int p = 0; p < 2; ++p) k_off[p] = c32 * 64 + (((2 * p + hi) ^ swz) << 4);
  int k8_off[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) k8_off[p] = c32 * 64 + ((p ^ swz) << 4) + 8 * hi;
  const int v_off0 = c32 * 64 + ((hi ^ swz) << 4), v_off1 = c32 * 64 + (((2 + hi) ^ swz) << 4);
  const int v8_off = c32 * 32 + 8 * hi;

  DLI_STAGE(0, 0);
  if (nblk > 1) {
    DLI_STAGE(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  int st = 0;
  for (int b = 0; b < nblk; ++b) {
    const bool ahead = b + 2 < nblk;
    const int st2 = st == 0 ? NSTAGE - 1 : st - 1;
    if (ahead) DLI_STAGE(b + 2, st2);
    if (b < my_nblk) {
      const char* kb = smem + st * 2 * PAGE_BYTES;
      const char* vb = kb + PAGE_BYTES;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bf16x8 kf;
        if constexpr (!KV8) {
          kf = *reinterpret_cast<const bf16x8*>(kb + (j >> 1) * 2048 + k_off[j & 1]);
        } else {
          const uint2 raw = *reinterpret_cast<const uint2*>(kb + (j >> 2) * 2048 + k8_off[j & 3]);
          fp8x4_to_bf16(raw.x, kf, 0);
          fp8x4_to_bf16(raw.y, kf, 4);
        }
        s = mfma32(kf, qf[j], s);
      }
      bf16x8 pb[2];
      float alpha;
      const int tok0 = b * 32 + 4 * hi;
      if (b * 32 + 31 > first_tok) softmax32<true>(s, tok0, qpos, scale_log2, m, lsum, alpha, pb);
      else softmax32<false>(s, tok0, qpos, scale_log2, m, lsum, alpha, pb);
      if (__any(alpha != 1.f)) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int n = 0; n < 4; ++n) o[n] *= alpha;
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        bf16x8 v0, v1;                           // slots hi and 2 + hi of row d = 32 n + c32: [k-step 0 | k-step 1] halves
        if constexpr (!KV8) {
          v0 = *reinterpret_cast<const bf16x8*>(vb + n * 2048 + v_off0);
          v1 = *reinterpret_cast<const bf16x8*>(vb + n * 2048 + v_off1);
        } else {
          const uint2 r0 = *reinterpret_cast<const uint2*>(vb + n * 1024 + v8_off);
          const uint2 r1 = *reinterpret_cast<const uint2*>(vb + n * 1024 + v8_off + 16);
          fp8x4_to_bf16(r0.x, v0, 0);
          fp8x4_to_bf16(r0.y, v0, 4);
          fp8x4_to_bf16(r1.x, v1, 0);
          fp8x4_to_bf16(r1.y, v1, 4);
        }
        const bf16x8 a0 = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const bf16x8 a1 = {v0[4], v0[5], v0[6], v0[7], v1[4], v1[5], v1[6], v1[7]};
        o[n] = mfma32(a0, pb[0], o[n]);
        o[n] = mfma32(a1, pb[1], o[n]);
      }
    }
    if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    st = st == NSTAGE - 1 ? 0 : st + 1;
  }

  {
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
    const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    if (valid) {
      const float rcp = l > 0.f ? v_scale / l : 0.f;
      uint16_t* op = out + (int64_t)(qs + qtok) * out_stride + (kvh * G + head) * HD + 4 * hi;
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          bf16x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (short)f2bf(o[n][4 * r4 + e] * rcp);
          *reinterpret_cast<bf16x4*>(op + 32 * n + 8 * r4) = w;
        }
    }
  }
#undef DLI_STAGE
}

// ---- launchers -------------------------------------------------------------------------------------
// Decode reads every K/V byte exactly once per step: the loads are NON-TEMPORAL (streaming cache policy), so the pages do not
// displace what the GEMMs keep in L2 / the infinity cache and are themselves not written into them: 5.95 → 6.60 TB/s on the
// trace's length mix at B = 1024 (profiles/ab_decode_nt_r3.jsonl).  DLI_DECODE_NT=0: plain loads (A/B knob).
static const bool g_decode_nt = std::getenv("DLI_DECODE_NT") == nullptr || std::atoi(std::getenv("DLI_DECODE_NT")) != 0;

void launch_paged_decode(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                         const int* block_tables, int max_blocks, const int* seq_lens, void* out, int64_t out_stride,
                         float* part_o, float* part_ml, int num_seqs, int Hq, int Hkv, int num_splits, float scale,
                         bool kv_fp8, float k_scale, float v_scale, const float* kv_scales, hipStream_t stream) {
  if (num_seqs == 0) return;
  const int G = Hq / Hkv;
  if (kv_scales != nullptr) k_scale = v_scale = 1.f;        // the kernels read the per-head values themselves
  const float scale_log2 = scale * 1.4426950408889634f * (kv_fp8 ? k_scale : 1.f);
  const size_t lds = (size_t)(4 * G * (HD + 4) + 8 * G) * sizeof(float);
  dim3 grid(Hkv, num_seqs, num_splits);
#define DLI_DECODE_GO(KV8_, NT_)                                                                                              \
  hipLaunchKernelGGL((paged_decode_kernel<KV8_, false, NT_>), grid, dim3(256), lds, stream, (const uint16_t*)q, q_stride, k_cache, \
                     v_cache, block_tables, max_blocks, seq_lens, (uint16_t*)out, out_stride, part_o, part_ml, Hq, Hkv,           \
                     num_splits, scale_log2, KV8_ ? v_scale : 1.f, KV8_ ? kv_scales : nullptr, nullptr, nullptr)
  if (kv_fp8) { if (g_decode_nt) DLI_DECODE_GO(true, true); else DLI_DECODE_GO(true, false); }
  else { if (g_decode_nt) DLI_DECODE_GO(false, true); else DLI_DECODE_GO(false, false); }
#undef DLI_DECODE_GO
  if (num_splits > 1)
    hipLaunchKernelGGL(decode_reduce_kernel, dim3(num_seqs, Hq), dim3(128), 0, stream, part_o, part_ml, (uint16_t*)out,
                       out_stride, Hq, num_splits, kv_fp8 ? v_scale : 1.f, kv_fp8 ? kv_scales : nullptr, Hkv, nullptr, nullptr);
}

// Work-list decode: `items` / `reds` are the LAUNCHED grid extents (an upper bound inside a hipGraph); the valid counts
// are counts[0] / counts[1] on the device.
void launch_paged_decode_worklist(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                                  const int* block_tables, int max_blocks, const int* seq_lens, void* out,
                                  int64_t out_stride, float* part_o, float* part_ml, const int* work, const int* red,
                                  const int* counts, int items, int reds, int Hq, int Hkv, float scale, bool kv_fp8,
                                  float k_scale, float v_scale, const float* kv_scales, hipStream_t stream) {
  if (items <= 0) return;
  const int G = Hq / Hkv;
  if (kv_scales != nullptr) k_scale = v_scale = 1.f;
  const float scale_log2 = scale * 1.4426950408889634f * (kv_fp8 ? k_scale : 1.f);
  const size_t lds = (size_t)(4 * G * (HD + 4) + 8 * G) * sizeof(float);
  dim3 grid(Hkv, items);
#define DLI_DECODE_GO(KV8_, NT_)                                                                                              \
  hipLaunchKernelGGL((paged_decode_kernel<KV8_, true, NT_>), grid, dim3(256), lds, stream, (const uint16_t*)q, q_stride, k_cache,  \
                     v_cache, block_tables, max_blocks, seq_lens, (uint16_t*)out, out_stride, part_o, part_ml, Hq, Hkv, 1,        \
                     scale_log2, KV8_ ? v_scale : 1.f, KV8_ ? kv_scales : nullptr, (const int4*)work, counts)
  if (kv_fp8) { if (g_decode_nt) DLI_DECODE_GO(true, true); else DLI_DECODE_GO(true, false); }
  else { if (g_decode_nt) DLI_DECODE_GO(false, true); else DLI_DECODE_GO(false, false); }
#undef DLI_DECODE_GO
  if (reds > 0)
    hipLaunchKernelGGL(decode_reduce_kernel, dim3(reds, Hq), dim3(128), 0, stream, part_o, part_ml, (uint16_t*)out,
                       out_stride, Hq, 1, kv_fp8 ? v_scale : 1.f, kv_fp8 ? kv_scales : nullptr, Hkv, (const int4*)red, counts);
}

void launch_paged_prefill(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                          const int* block_tables, int max_blocks, const int* seq_lens, const int* query_start_loc,
                          const int* tile_seq, const int* tile_q0, int num_tiles, void* out, int64_t out_stride, int Hq,
                          int Hkv, float scale, bool kv_fp8, float k_scale, float v_scale, const float* kv_scales,
                          const int* counts, hipStream_t stream) {
  if (num_tiles == 0) return;
  if (kv_scales != nullptr) k_scale = v_scale = 1.f;
  const float scale_log2 = scale * 1.4426950408889634f * (kv_fp8 ? k_scale : 1.f);
  dim3 grid((num_tiles * Hkv + 7) / 8 * 8);
  if (kv_fp8)
    hipLaunchKernelGGL(paged_prefill_kernel<true>, grid, dim3(256), 0, stream, (const uint16_t*)q, q_stride, k_cache,
                       v_cache, block_tables, max_blocks, seq_lens, query_start_loc, tile_seq, tile_q0, (uint16_t*)out,
                       out_stride, Hq, Hkv, num_tiles, scale_log2, v_scale, kv_scales, counts);
  else
    hipLaunchKernelGGL(paged_prefill_kernel<false>, grid, dim3(256), 0, stream, (const uint16_t*)q, q_stride, k_cache,
                       v_cache, block_tables, max_blocks, seq_lens, query_start_loc, tile_seq, tile_q0, (uint16_t*)out,
                       out_stride, Hq, Hkv, num_tiles, scale_log2, 1.f, nullptr, counts);
}

}  // namespace dli
