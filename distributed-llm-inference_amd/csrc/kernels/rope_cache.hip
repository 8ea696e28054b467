// K2: NeoX rotate-half RoPE on q and k fused with the paged-KV write.
//
// One 256-thread workgroup per 4 consecutive tokens.  Work items are 8-element (16-byte) chunks: a q/k item owns
// the pair (x[d0:d0+8], x[d0+64:d0+72]) of one head, so both halves of the rotation are produced by
// the same thread; cos/sin come from a host-built fp32 table (no device trig), loaded once per (token, chunk) and
// reused over the heads a thread walks.  Rotated k goes back
// into the qkv buffer (the prefill kernel reads everything from the cache, but the in-place result
// keeps the op's semantics testable) and into the K page as one 16-byte store per chunk; v goes into
// the transposed, slot-permuted V page (layout: ops/kv_layout.py).
//
// KV8 = true stores the cache as OCP e4m3 with per-tensor scales (value / scale, saturating).
#include "common.h"
#include "kernels.h"

namespace dli {

constexpr int D = 128;

__device__ __forceinline__ int v_slot(int t) { return ((t & 15) >> 2) * 8 + (t >> 4) * 4 + (t & 3); }

// TB = tokens per workgroup.  4 for prefill-sized launches (full 256-thread passes, grouped V stores); 1 for
// decode-sized ones, where a launch is a handful of tokens and the kernel is pure latency: 4 tokens per workgroup
// meant 5 dependent load → rotate → store rounds on 1–2 CUs (9 µs at T = 5), one token per workgroup needs 2.
template <bool KV8, int TB>
__global__ __launch_bounds__(256) void rope_cache_kernel(uint16_t* __restrict__ qkv, int64_t row_stride,
                                                         const int* __restrict__ positions,
                                                         const float* __restrict__ cos_sin,  // [max_pos, D] or null
                                                         const int* __restrict__ slot_mapping,
                                                         void* __restrict__ k_cache_, void* __restrict__ v_cache_,
                                                         int T, int Hq, int Hkv, float k_inv_scale_, float v_inv_scale_,
                                                         const float* __restrict__ kv_scales) {   // [k | v][Hkv] or null
  // Workgroup → token group.  Hardware deals workgroups round-robin over the 8 XCDs, each with its own L2, and a
  // 64-byte V page row is completed by 8 consecutive 4-token groups (8 bytes each): with group = blockIdx those eight
  // partial writes sit in eight different L2s and leave as partial lines.  XCD x takes the x-th eighth of the groups
  // instead, so neighbours in the batch are neighbours in one L2 and merge there.
  const int per_xcd = (gridDim.x + 7) >> 3;
  const int grp = TB == 1 ? (int)blockIdx.x : (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (grp * TB >= T) return;
  const int t0 = grp * TB;
  const int nt = min(TB, T - t0);

  // ---- q/k: rotate in place, k also into its K page.  Thread = (token of the workgroup, chunk j of the half-row,
  // head group hg); it loads the 16 cos/sin values of (position, j) ONCE and walks heads hg, hg + NG, … with them
  // (round 1 loaded the table per (head, j) item: 64 bytes of table for 32 bytes of data).  Lanes j = 0..7 of a head
  // cover 128 contiguous bytes, the 8 head groups of a wave 8 such lines: every load and store is whole cache lines.
  constexpr int NG = 256 / (TB * 8);                       // head groups: 8 (prefill form), 32 (decode form)
  {
    const int ti = threadIdx.x / (8 * NG), rem = threadIdx.x - ti * (8 * NG);
    const int hg = rem >> 3, j = rem & 7;
    if (ti < nt) {
      const int64_t t = t0 + ti;
      uint16_t* row = qkv + t * row_stride;
      const int slot = slot_mapping[t];
      f32x4 c0, c1, s0, s1;
      if (cos_sin != nullptr) {
        const float* cs = cos_sin + (int64_t)positions[t] * D + j * 8;
        c0 = *reinterpret_cast<const f32x4*>(cs);
        c1 = *reinterpret_cast<const f32x4*>(cs + 4);
        s0 = *reinterpret_cast<const f32x4*>(cs + 64);
        s1 = *reinterpret_cast<const f32x4*>(cs + 68);
      }
      for (int head = hg; head < Hq + Hkv; head += NG) {
        uint16_t* p = row + head * D + j * 8;
        bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
        bf16x8 b = *reinterpret_cast<const bf16x8*>(p + 64);
        if (cos_sin != nullptr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float c = e < 4 ? c0[e] : c1[e - 4], sn = e < 4 ? s0[e] : s1[e - 4];
            const float x1 = bf2f((uint16_t)a[e]), x2 = bf2f((uint16_t)b[e]);
            a[e] = (short)f2bf(x1 * c - x2 * sn);
            b[e] = (short)f2bf(x2 * c + x1 * sn);
          }
          *reinterpret_cast<bf16x8*>(p) = a;
          *reinterpret_cast<bf16x8*>(p + 64) = b;
        }
        if (head >= Hq && slot >= 0) {             // k head → K page
          const int h = head - Hq, blk = slot >> 5, tok = slot & 31;
          const int64_t page = (int64_t)blk * Hkv + h;
          if (!KV8) {
            uint16_t* kc = reinterpret_cast<uint16_t*>(k_cache_) + page * (D * 32);
            const int d0 = j * 8, d1 = 64 + j * 8;   // chunk c = d/32, offset d%32
            *reinterpret_cast<bf16x8*>(kc + ((d0 >> 5) * 32 + tok) * 32 + (d0 & 31)) = a;
            *reinterpret_cast<bf16x8*>(kc + ((d1 >> 5) * 32 + tok) * 32 + (d1 & 31)) = b;
          } else {
            uint8_t* kc = reinterpret_cast<uint8_t*>(k_cache_) + page * (D * 32);
            const float k_inv_scale = kv_scales != nullptr ? fast_rcp(kv_scales[h]) : k_inv_scale_;
            uint64_t pa = 0, pb = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              pa |= (uint64_t)f_to_fp8(bf2f((uint16_t)a[e]) * k_inv_scale) << (8 * e);
              pb |= (uint64_t)f_to_fp8(bf2f((uint16_t)b[e]) * k_inv_scale) << (8 * e);
            }
            // chunk c = d/64: first half is chunk 0, second half chunk 1
            *reinterpret_cast<uint64_t*>(kc + (0 * 32 + tok) * 64 + j * 8) = pa;
            *reinterpret_cast<uint64_t*>(kc + (1 * 32 + tok) * 64 + j * 8) = pb;
          }
        }
      }
    }
  }

  // ---- v → transposed V page.  Tokens 4k…4k+3 of a page occupy 4 ADJACENT slots of every d row (v_slot), so a
  // prefill chunk — whose TB tokens sit in consecutive cache slots — is written with one 8-byte store per (d row,
  // 4 tokens) instead of four 2-byte ones.  Decode tokens (one per sequence, unrelated slots) take the scalar path.
  const int s_first = slot_mapping[t0];
  bool grouped = TB == 4 && nt == TB && s_first >= 0 && (s_first & 3) == 0;   // the 4-wide store needs 4 tokens
#pragma unroll
  for (int i = 1; i < TB; ++i) grouped = grouped && (i >= nt || slot_mapping[t0 + i] == s_first + i);
  if (grouped) {
    const int blk = s_first >> 5, sl = v_slot(s_first & 31);
    for (int vi = threadIdx.x; vi < Hkv * 16; vi += blockDim.x) {
      const int h = vi >> 4, d0 = (vi & 15) * 8;
      bf16x8 x[TB];
#pragma unroll
      for (int i = 0; i < TB; ++i)
        x[i] = *reinterpret_cast<const bf16x8*>(qkv + (int64_t)(t0 + i) * row_stride + (Hq + Hkv + h) * D + d0);
      const int64_t page = (int64_t)blk * Hkv + h;
      if (!KV8) {
        uint16_t* vc = reinterpret_cast<uint16_t*>(v_cache_) + page * (D * 32) + d0 * 32 + sl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          bf16x4 w;
#pragma unroll
          for (int i = 0; i < TB; ++i) w[i] = x[i][e];
          *reinterpret_cast<bf16x4*>(vc + e * 32) = w;
        }
      } else {
        uint8_t* vc = reinterpret_cast<uint8_t*>(v_cache_) + page * (D * 32) + d0 * 32 + sl;
        const float v_inv_scale = kv_scales != nullptr ? fast_rcp(kv_scales[Hkv + h]) : v_inv_scale_;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          uint32_t w = 0;
#pragma unroll
          for (int i = 0; i < TB; ++i) w |= (uint32_t)f_to_fp8(bf2f((uint16_t)x[i][e]) * v_inv_scale) << (8 * i);
          *reinterpret_cast<uint32_t*>(vc + e * 32) = w;
        }
      }
    }
  } else {
    for (int it = threadIdx.x; it < Hkv * 16 * nt; it += blockDim.x) {
      const int ti = it / (Hkv * 16), vi = it - ti * (Hkv * 16);
      const int64_t t = t0 + ti;
      const int slot = slot_mapping[t];
      if (slot < 0) continue;
      const int h = vi >> 4, d0 = (vi & 15) * 8;
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(qkv + t * row_stride + (Hq + Hkv + h) * D + d0);
      const int64_t page = (int64_t)(slot >> 5) * Hkv + h;
      const int sl = v_slot(slot & 31);
      if (!KV8) {
        uint16_t* vc = reinterpret_cast<uint16_t*>(v_cache_) + page * (D * 32) + d0 * 32 + sl;
#pragma unroll
        for (int e = 0; e < 8; ++e) vc[e * 32] = (uint16_t)a[e];
      } else {
        uint8_t* vc = reinterpret_cast<uint8_t*>(v_cache_) + page * (D * 32) + d0 * 32 + sl;
        const float v_inv_scale = kv_scales != nullptr ? fast_rcp(kv_scales[Hkv + h]) : v_inv_scale_;
#pragma unroll
        for (int e = 0; e < 8; ++e) vc[e * 32] = f_to_fp8(bf2f((uint16_t)a[e]) * v_inv_scale);
      }
    }
  }
}

void launch_rope_and_cache(void* qkv, int64_t row_stride, const int* positions, const float* cos_sin,
                           const int* slot_mapping, void* k_cache, void* v_cache, int T, int Hq, int Hkv,
                           bool kv_fp8, float k_scale, float v_scale, const float* kv_scales, hipStream_t stream) {
  if (T == 0) return;
#define DLI_ROPE(KV8_, TB_, ks_, vs_)                                                                                   \
  hipLaunchKernelGGL((rope_cache_kernel<KV8_, TB_>), dim3((((T + TB_ - 1) / TB_) + 7) / 8 * 8), dim3(256), 0, stream, (uint16_t*)qkv,   \
                     row_stride, positions, cos_sin, slot_mapping, k_cache, v_cache, T, Hq, Hkv, ks_, vs_, kv_fp8 ? kv_scales : nullptr)
  const bool small = T <= 128;                          // ≤ 128 workgroups either way: take the short dependency chain
  if (kv_fp8) {
    if (small) DLI_ROPE(true, 1, 1.f / k_scale, 1.f / v_scale);
    else DLI_ROPE(true, 4, 1.f / k_scale, 1.f / v_scale);
  } else {
    if (small) DLI_ROPE(false, 1, 1.f, 1.f);
    else DLI_ROPE(false, 4, 1.f, 1.f);
  }
#undef DLI_ROPE
}

// K9 helper: device-side page copy for copy-on-write (pairs = [n, 2] int32: src, dst page ids).
__global__ void copy_pages_kernel(uint8_t* __restrict__ cache, const int* __restrict__ pairs, int64_t page_bytes) {
  const int64_t src = pairs[2 * blockIdx.x], dst = pairs[2 * blockIdx.x + 1];
  const uint4* s = reinterpret_cast<const uint4*>(cache + src * page_bytes);
  uint4* d = reinterpret_cast<uint4*>(cache + dst * page_bytes);
  for (int64_t i = threadIdx.x; i < page_bytes / 16; i += blockDim.x) d[i] = s[i];
}

void launch_copy_pages(void* cache, const int* pairs, int n_pairs, int64_t page_bytes, hipStream_t stream) {
  if (n_pairs == 0) return;
  hipLaunchKernelGGL(copy_pages_kernel, dim3(n_pairs), dim3(256), 0, stream, (uint8_t*)cache, pairs, page_bytes);
}

}  // namespace dli
