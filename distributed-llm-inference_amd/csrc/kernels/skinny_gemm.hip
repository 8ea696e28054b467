// K10: decode-sized linear layer, y[M,N] = x[M,K] · W[N,K]ᵀ (+ bias) for M ≤ 16 (up to 64 in 16-token passes).
//
// At M ≤ 16 the layer is a pure stream of W through the chip (2·N·K bytes, ≈ 0 reuse), so the kernel is built
// around HBM, not around the matrix cores: the product is computed transposed, yᵀ = W · xᵀ, with W as the MFMA
// A operand (16 weight rows per v_mfma_f32_16x16x32_bf16) and the ≤ 16 tokens as the 16 B columns — one MFMA
// per KiB of weights, far below the MFMA roof.
//
// What decides the speed is the SHAPE of the global reads.  Loading the A fragments directly (lane = row l&15,
// 16 bytes at k-chunk l>>4) touches 16 rows × 64 B per wave instruction and measured 3.6–4.6 TB/s; rows read in
// 256-byte runs measured 5.2–5.9 TB/s.  So operands go through LDS: a wave fetches a 16-row × 128-k tile with four
// LDS-DMA instructions (global_load_lds, 16 lanes × 16 B = 256 contiguous bytes of each of 4 rows, no VGPRs), and
// reads the MFMA fragments back with ds_read_b128.  One DMA instruction fills 1 KiB of LDS in lane order, so the
// bank swizzle is applied on the source side: LDS chunk p of tile row t holds global 16-byte chunk p ^ t; the
// fragment read of chunk c = 4j + (l>>4) of row t = l&15 goes to slot c ^ t, which is conflict-free for the
// 16-lane groups of ds_read_b128.  The activations take the same path (rows ≥ M repeat the last token; those output
// columns are never stored); with M ≤ 4 / ≤ 8 only 4 / 8 token rows are fetched per slab and the remaining fragment
// columns alias them (x_swz below keeps those reads conflict-free too: SQ_LDS_BANK_CONFLICT = 0,
// profiles/pmc_skinny_r1k.md).
//
// A 256-thread workgroup owns 16 weight rows; its 4 waves split K between them in interleaved 128-element
// slabs, each wave double-buffers its slabs in a private LDS ring (no barriers in the loop: the wave waits on its
// own vmcnt), the 4 partial accumulators meet in LDS at the end and the sum is written as bf16.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace dli {

constexpr int SK_WAVES = 4;
static const bool g_skinny_no_8w = std::getenv("DLI_SKINNY_NO_8W") != nullptr;   // A-B measurements
constexpr int SK_TILE = 4096;    // bytes: 16 rows × 128 bf16
// Cache policy of the weight DMA (the `aux` immediate of global_load_lds): NT for a pass after which the workgroup does
// not come back to these rows — decode streams every weight byte once per step, and a streaming load neither displaces
// what other kernels keep in L2 / the infinity cache nor is written into them: +2…13 % on every Llama-3-8B decode shape
// (gate/up 39.8 → 35.7 µs, down 22.5 → 19.9 µs, LM head 162 → 148 µs; profiles/ab_skinny_nt_r3.txt).  Passes that WILL be
// followed by another one over the same rows (M > 16) keep the default policy: with NT their re-reads miss (0.68–0.9 ×).
constexpr int SK_AUX_NT = 2, SK_AUX_KEEP = 0;
constexpr int MOE_BM = 128;      // rows per expert tile of moe_align (moe.hip: BM)

// XI = DMA instructions spent on the activation tile (4 token rows each): M ≤ 4 needs one, M ≤ 8 two.  A ring
// slot is XI KiB of activations followed by the 4 KiB weight tile.  (Two weight row tiles per workgroup — half the
// activation traffic — measured 20–30 % SLOWER on every Llama shape: fewer, fatter workgroups per CU.)
//
// Swizzle of the activation tile: LDS chunk p of tile row t holds global chunk p ^ x_swz<XI>(t).  With all 16 rows
// loaded it is the weight tile's p ^ t; with 8 or 4 rows (the other fragment columns alias them) the XOR is spread
// so that the 16 (8) distinct addresses of a ds_read_b128 lane group still fall on different 16-byte slots.
template <int XI>
__device__ __forceinline__ int x_swz(int t) { return XI == 4 ? t : XI == 2 ? (t & 7) << 1 : (t & 3) << 2; }

// The streaming loop + cross-wave reduction, shared by the dense and the MoE kernels.  xsrc[i] / wsrc[i]: this
// lane's source of DMA instruction i (tile row 4i + (lane>>4), swizzled chunk already added) at k = 0.
// Returns to thread t the sum for token m = t>>4 and weight row (t&15) of the tile.
// W8: the weights are OCP e4m3 (W8A16 decode of the fp8 option): a weight tile row is 128 bytes, the tile 2 KiB = 2 DMA
// instructions (8 rows × 128 B each: lane → row lane/8, 16-byte chunk lane%8, swizzled by the row like the bf16 tile);
// a lane's fragment of k-block j is 8 bytes, widened to bf16 in registers (exact) for the same bf16 MFMA — half the
// weight bytes per output, and weights are all this kernel streams.  The per-channel scale is the caller's epilogue.
template <int XI, int NW = SK_WAVES, int WAUX = SK_AUX_KEEP, bool W8 = false>
__device__ __forceinline__ float skinny_dot(const char* const (&xsrc)[XI], const char* const (&wsrc)[W8 ? 2 : 4], int K, char* smem) {
  constexpr int WT = W8 ? SK_TILE / 2 : SK_TILE;     // bytes of a weight tile
  constexpr int WI = W8 ? 2 : 4;                      // DMA instructions per weight tile
  constexpr int SLOT = XI * 1024 + WT;                // bytes per slab: activations + weight tile
  constexpr int XROWS = 4 * XI;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  char* ring = smem + wave * (2 * SLOT);              // [wave][2 slots][SLOT]
#define DLI_SK_ISSUE(slab_, slot_)                                                                                 \
  {                                                                                                                \
    char* dst_ = ring + (slot_) * SLOT;                                                                            \
    const int64_t kb_ = (int64_t)(slab_) * 256;                                                                    \
    _Pragma("unroll") for (int i = 0; i < XI; ++i)                                                                 \
        __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[i] + kb_), (lptr_t)(dst_ + i * 1024), 16, 0, 0);            \
    _Pragma("unroll") for (int i = 0; i < WI; ++i)                                                                 \
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + (W8 ? kb_ / 2 : kb_)), (lptr_t)(dst_ + (XI + i) * 1024), 16, 0, WAUX); \
  }

  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const int n_it = K >> 7;
  int frag_off[4], xfrag_off[4];                       // this lane's fragment j inside a weight / activation tile
  const int xr = r & (XROWS - 1);                      // columns ≥ XROWS alias loaded rows (never stored)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (W8) frag_off[j] = XI * 1024 + r * 128 + ((((4 * j + g) >> 1) ^ ((r >> 1) & 7)) << 4) + ((4 * j + g) & 1) * 8;   // key: see the kernel's wsrc
    else frag_off[j] = XI * 1024 + r * 256 + (((4 * j + g) ^ r) << 4);
    xfrag_off[j] = xr * 256 + (((4 * j + g) ^ x_swz<XI>(xr)) << 4);
  }

  if (wave < n_it) DLI_SK_ISSUE(wave, 0);
  int slot = 0;
  for (int it = wave; it < n_it; it += NW, slot ^= 1) {
    // the other slot was consumed by the MFMAs of the previous iteration (they waited for its ds_reads)
    if (it + NW < n_it) {
      DLI_SK_ISSUE(it + NW, slot ^ 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XI + WI) : "memory");    // everything but the slab just issued
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char* base = ring + slot * SLOT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(base + xfrag_off[j]);
      bf16x8 wf;
      if (W8) {
        const u32x2 raw = *reinterpret_cast<const u32x2*>(base + frag_off[j]);
        fp8x4_to_bf16(raw[0], wf, 0);
        fp8x4_to_bf16(raw[1], wf, 4);
      } else {
        wf = *reinterpret_cast<const bf16x8*>(base + frag_off[j]);
      }
      acc = mfma16(wf, xf, acc);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments are in registers before the slot is refilled
    __builtin_amdgcn_sched_barrier(0);
  }
#undef DLI_SK_ISSUE

  // lane holds yᵀ[n = 4g + j][m = r]; the ring is dead once every wave is past its loop
  __syncthreads();
  // [wave][288]: element (j, lane) at 64j + lane + 4j + 2·(lane>>5): the 16 (j, g) runs of 16 tokens start on 16
  // different even banks, so the transposed read below (32 lanes = 16 runs × 2 tokens) hits 32 different banks
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < 4; ++j) red[wave * 288 + j * 64 + lane + 4 * j + 2 * (lane >> 5)] = acc[j];
  __syncthreads();
  const int m = threadIdx.x >> 4, nn = threadIdx.x & 15;     // consecutive threads → consecutive n of one token
  const int src = (nn & 3) * 64 + (nn >> 2) * 16 + m + 4 * (nn & 3) + 2 * (nn >> 3);
  float s = 0.f;
#pragma unroll
  for (int wv_ = 0; wv_ < NW; ++wv_) s += red[wv_ * 288 + src];
  return s;
}

// SWIGLU: W is a fused [gate; up] projection (N = 2·I rows).  A workgroup then takes 8 gate rows and the 8 matching
// up rows as its 16-row tile and stores silu(gate)·up — [M, I] — so the activation kernel and the round trip of the
// [M, 2·I] intermediate disappear from the decode step.
template <int XI, bool SWIGLU, int NW, bool W8 = false>
__global__ __launch_bounds__(NW * 64) void skinny_linear_kernel(const uint16_t* __restrict__ x, int64_t x_stride,
                                                                      const void* __restrict__ w_,
                                                                      const uint16_t* __restrict__ bias,
                                                                      uint16_t* __restrict__ out, int64_t out_stride,
                                                                      int M, int N, int K, const float* __restrict__ w_scale) {
  extern __shared__ __attribute__((aligned(1024))) char sk_smem[];
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  // XCD x streams the x-th eighth of the weight rows: neighbouring 16-column pieces of an output row (32 bytes each)
  // are then written through ONE L2 and leave as whole lines
  const int per_xcd = (gridDim.x + 7) >> 3;
  const int wg = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (wg * 16 >= N) return;
  const int n0 = wg * (SWIGLU ? 8 : 16);
  // DMA sources: instruction i of a tile covers tile rows 4i + (lane>>4); this lane supplies LDS chunk p = lane&15,
  // i.e. global chunk p ^ row.  Rows ≥ M repeat the last token.
  const char* wsrc[W8 ? 2 : 4];
#pragma unroll
  for (int i = 0; i < (W8 ? 2 : 4); ++i) {
    const int row = W8 ? 8 * i + (lane >> 3) : 4 * i + g;
    const int wrow = SWIGLU ? (row < 8 ? n0 + row : (N >> 1) + n0 + row - 8) : n0 + row;
    // e4m3 rows are 128 B = half a bank line: rows of equal parity share their banks, so the swizzle key is (row >> 1) & 7 —
    // distinct over the 8 rows that do (with row & 7 rows r and r + 8 collided: 0.23 LDS conflict cycles per active cycle)
    if (W8) wsrc[i] = reinterpret_cast<const char*>(w_) + (int64_t)wrow * K + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
    else wsrc[i] = reinterpret_cast<const char*>(reinterpret_cast<const uint16_t*>(w_) + (int64_t)wrow * K) + ((r ^ row) << 4);
  }
  // More than 16 tokens: further passes over the same 16 weight rows, 16 tokens each.  The workgroup re-reads the
  // 32–112 KiB it has just streamed; with K ≤ 1024-sized shards those re-reads hit L2 and the pass costs ≈ 40 % of
  // the first, with K = 4096 they mostly miss and it costs as much — the caller decides by size (ops.linear).
  for (int m0 = 0; m0 < M; m0 += 16) {
    const int Ms = min(M - m0, 16);
    const char* xsrc[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int row = 4 * i + g;
      xsrc[i] = reinterpret_cast<const char*>(x + (int64_t)(m0 + min(row, Ms - 1)) * x_stride) + ((r ^ x_swz<XI>(row)) << 4);
    }
    float s;
    if (XI < 4 || m0 + 16 >= M) s = skinny_dot<XI, NW, SK_AUX_NT, W8>(xsrc, wsrc, K, sk_smem);     // the last (or only) pass
    else s = skinny_dot<XI, NW, SK_AUX_KEEP, W8>(xsrc, wsrc, K, sk_smem);
    const int m = threadIdx.x >> 4, nn = threadIdx.x & 15;   // m ≥ 16 (waves 4–7 of the 8-wave form): nothing to store
    if (W8) s *= w_scale[SWIGLU ? (nn < 8 ? n0 + nn : (N >> 1) + n0 + nn - 8) : n0 + nn];     // per-output-channel scale
    if (SWIGLU) {
      const float up = __shfl_xor(s, 8, 64);           // thread nn < 8 holds gate column n0 + nn, thread nn + 8 its up
      if (m < Ms && nn < 8)
        out[(int64_t)(m0 + m) * out_stride + n0 + nn] = f2bf(s * fast_rcp(1.f + fast_exp2(-1.4426950408889634f * s)) * up);
    } else if (m < Ms) {
      if (bias != nullptr) s += bf2f(bias[n0 + nn]);
      out[(int64_t)(m0 + m) * out_stride + n0 + nn] = f2bf(s);
    }
    if (m0 + 16 < M) __syncthreads();                  // the reduction scratch is the next pass's ring
  }
}

// MoE at decode size (few tokens, so an expert sees ≤ 16 rows as a rule): one row tile per (weight row tile, expert
// tile of moe_align).  DOWN = false: gate/up projection, activations gathered by token (sorted id / top_k), raw
// result to gu[tile·16 + m][N] (the SwiGLU epilogue over 8 gate + 8 up rows per workgroup, as in the dense kernel,
// measured 4–5 % SLOWER here: two distant 8-row streams per workgroup instead of one 16-row stream cost more than
// the separate activation kernel).  DOWN = true: activations act[tile·16 + m][K], result scaled by the routing
// weight and stored at its (token, slot) row y[sorted id][N] for moe_sum.
template <int XI, bool DOWN, int NW>
__global__ __launch_bounds__(NW * 64) void moe_skinny_kernel(const uint16_t* __restrict__ a, int64_t a_stride,
                                                                   const uint16_t* __restrict__ w, int64_t w_expert_stride,
                                                                   const int* __restrict__ sorted_ids,
                                                                   const int* __restrict__ tile_expert,
                                                                   const int* __restrict__ meta,
                                                                   const float* __restrict__ topk_w, int top_k,
                                                                   uint16_t* __restrict__ out, int64_t out_stride, int K,
                                                                   int rows_pad) {
  extern __shared__ __attribute__((aligned(1024))) char sk_smem[];
  const int tile = blockIdx.y;
  if (tile >= meta[0]) return;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int* sids = sorted_ids + tile * MOE_BM;
  int M = 0;                                           // valid rows are packed at the front of the tile
  for (int i = 0; i < rows_pad; ++i) M += sids[i] >= 0;
  const uint16_t* we = w + (int64_t)tile_expert[tile] * w_expert_stride;
  const char* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * i + g;
    wsrc[i] = reinterpret_cast<const char*>(we + (int64_t)(n0 + row) * K) + ((r ^ row) << 4);
  }
  // An expert with more than 16 rows (possible once T > 16) takes further passes over its 16 weight rows; the
  // re-read comes from L2 / the infinity cache, and the batch sizes routed here keep that the exception.
  for (int m0 = 0; m0 < M; m0 += 16) {
    const int Ms = min(M - m0, 16);
    const char* xsrc[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int row = 4 * i + g, rr = m0 + min(row, Ms - 1);
      const int64_t arow = DOWN ? (int64_t)tile * rows_pad + rr : sids[rr] / top_k;
      xsrc[i] = reinterpret_cast<const char*>(a + arow * a_stride) + ((r ^ x_swz<XI>(row)) << 4);
    }
    float s;
    if (XI < 4 || m0 + 16 >= M) s = skinny_dot<XI, NW, SK_AUX_NT>(xsrc, wsrc, K, sk_smem);     // the last (or only) pass
    else s = skinny_dot<XI, NW, SK_AUX_KEEP>(xsrc, wsrc, K, sk_smem);
    const int m = threadIdx.x >> 4, nn = threadIdx.x & 15;
    if (m < Ms) {
      if (DOWN) {
        const int sid = sids[m0 + m];
        out[(int64_t)sid * out_stride + n0 + nn] = f2bf(s * topk_w[sid]);
      } else {
        out[((int64_t)tile * rows_pad + m0 + m) * out_stride + n0 + nn] = f2bf(s);
      }
    }
    __syncthreads();                                   // the reduction scratch is the next pass's ring
  }
}

// NW = waves per workgroup = in-workgroup K split.  A workgroup is a latency chain — slabs per wave × HBM round trip —
// so 8 waves (half the slabs per wave, twice the bytes in flight per workgroup, 80–128 KiB of LDS) win wherever they
// do not cost occupancy.  Measured in a hipGraph: −4…−14 % up to 256 workgroups (one per CU either way) at any M, and
// −4…−6 % on every shape at M ≤ 4 (80 KiB: still two workgroups per CU; gate/up 6.0, LM head 6.3 TB/s); with M > 4
// and more workgroups than CUs the lost occupancy costs more than the shorter chain gains.
template <int XI, bool SWIGLU, int NW, bool W8>
static void launch_skinny(void* out, int64_t out_stride, const void* x, int64_t x_stride, const void* w, const void* bias,
                          int M, int N, int K, const float* w_scale, hipStream_t stream) {
  constexpr int lds = NW * 2 * (XI * 1024 + (W8 ? SK_TILE / 2 : SK_TILE));
  hipLaunchKernelGGL((skinny_linear_kernel<XI, SWIGLU, NW, W8>), dim3((N / 16 + 7) / 8 * 8), dim3(NW * 64), lds, stream,
                     (const uint16_t*)x, x_stride, w, (const uint16_t*)bias, (uint16_t*)out, out_stride, M, N, K, w_scale);
}

template <int XI, bool W8>
static void allow_big_lds() {
  constexpr int lds = 8 * 2 * (XI * 1024 + (W8 ? SK_TILE / 2 : SK_TILE));
  if (lds <= 65536) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_linear_kernel<XI, false, 8, W8>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_linear_kernel<XI, true, 8, W8>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

// w_scale != nullptr: `w` is OCP e4m3 [N, K] with one fp32 scale per output channel (W8A16).
void launch_skinny_linear(void* out, int64_t out_stride, const void* x, int64_t x_stride, const void* w, const void* bias,
                          int M, int N, int K, bool swiglu, const float* w_scale, hipStream_t stream) {
  // > 64 KiB of dynamic LDS needs an opt-in per kernel; done for every 8-wave variant on the first call, so that no
  // attribute call can fall inside a later stream capture
  static const bool configured = [] {
    allow_big_lds<1, false>(); allow_big_lds<2, false>(); allow_big_lds<4, false>();
    allow_big_lds<1, true>(); allow_big_lds<2, true>(); allow_big_lds<4, true>();
    return true;
  }();
  (void)configured;
  const int tiles = N / 16;
  const bool narrow = (tiles <= 256 || M <= 4) && K >= 1024 && !g_skinny_no_8w;
#define DLI_SK2(XI_, W8_)                                                                                             \
  (swiglu ? (narrow ? launch_skinny<XI_, true, 8, W8_>(out, out_stride, x, x_stride, w, bias, M, N, K, w_scale, stream)    \
                    : launch_skinny<XI_, true, 4, W8_>(out, out_stride, x, x_stride, w, bias, M, N, K, w_scale, stream))   \
          : (narrow ? launch_skinny<XI_, false, 8, W8_>(out, out_stride, x, x_stride, w, bias, M, N, K, w_scale, stream)   \
                    : launch_skinny<XI_, false, 4, W8_>(out, out_stride, x, x_stride, w, bias, M, N, K, w_scale, stream)))
#define DLI_SK(XI_) (w_scale != nullptr ? DLI_SK2(XI_, true) : DLI_SK2(XI_, false))
  if (M <= 4) DLI_SK(1);
  else if (M <= 8) DLI_SK(2);
  else DLI_SK(4);
#undef DLI_SK
#undef DLI_SK2
}

template <int XI, bool DOWN, int NW>
static void launch_moe_skinny_x(const void* a, int64_t a_stride, const void* w, int64_t w_expert_stride, const int* sorted_ids,
                                const int* tile_expert, const int* meta, const float* topk_w, int top_k, void* out,
                                int64_t out_stride, int K, int N, int max_tiles, int rows_pad, hipStream_t stream) {
  constexpr int lds = NW * 2 * (XI * 1024 + SK_TILE);
  static const bool configured = [] {                  // 8 waves: 80 KiB of dynamic LDS needs the opt-in
    if (lds > 65536)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&moe_skinny_kernel<XI, DOWN, NW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return true;
  }();
  (void)configured;
  hipLaunchKernelGGL((moe_skinny_kernel<XI, DOWN, NW>), dim3(N / 16, max_tiles), dim3(NW * 64), lds, stream,
                     (const uint16_t*)a, a_stride, (const uint16_t*)w, w_expert_stride, sorted_ids, tile_expert, meta, topk_w,
                     top_k, (uint16_t*)out, out_stride, K, rows_pad);
}

// max_rows: upper bound of rows per expert (= T ≤ 128): picks how many activation rows a slab carries; the
// intermediate buffers hold rows_pad = ⌈max_rows / 16⌉·16 rows per expert tile.
void launch_moe_skinny(bool down, const void* a, int64_t a_stride, const void* w, int64_t w_expert_stride,
                       const int* sorted_ids, const int* tile_expert, const int* meta, const float* topk_w, int top_k,
                       void* out, int64_t out_stride, int K, int N, int max_tiles, int max_rows, hipStream_t stream) {
  const int rows_pad = (max_rows + 15) / 16 * 16;
#define DLI_MS(XI_, NW_)                                                                                                 \
  (down ? launch_moe_skinny_x<XI_, true, NW_>(a, a_stride, w, w_expert_stride, sorted_ids, tile_expert, meta, topk_w, top_k, \
                                              out, out_stride, K, N, max_tiles, rows_pad, stream)                        \
        : launch_moe_skinny_x<XI_, false, NW_>(a, a_stride, w, w_expert_stride, sorted_ids, tile_expert, meta, topk_w, top_k, \
                                               out, out_stride, K, N, max_tiles, rows_pad, stream))
  if (max_rows <= 4) DLI_MS(1, 8);                     // as in the dense kernel: 8 waves pay while two workgroups fit a CU
  else if (max_rows <= 8) DLI_MS(2, 4);
  else DLI_MS(4, 4);
#undef DLI_MS
}

}  // namespace dli
