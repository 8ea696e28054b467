// K8: Mixtral-style mixture-of-experts on gfx950: routing, token alignment, grouped GEMM
// on v_mfma_f32_16x16x32_bf16 with fused epilogues, and the weighted un-permute.
//
//   moe_route      softmax over all experts → top-k → renormalise               [T,E] → ids/weights [T,k]
//   moe_align      bucket the (token, slot) pairs of the *local* experts; every expert's segment is
//                  padded to a multiple of the GEMM row tile; all sizes stay on the device, so the
//                  whole block is hipGraph-capturable (grids are launched at their worst-case size and
//                  surplus tiles exit on a device-side count)
//   grouped_gemm   C[rows of expert e] = A[rows] · W[e]ᵀ, 128×128×64 tiles, 4 waves (2×2) of 64×64,
//                  A rows gathered by index straight into the (XOR-swizzled) LDS image, register
//                  prefetch of the next K tile under the MFMAs.
//                  EPI_SWIGLU: the 128 B-rows of a tile are 2×(32 gate + 32 up) weight rows, so each
//                  wave owns matching gate/up columns and applies silu(g)·u in registers — the [rows, 2I]
//                  intermediate never exists.
//                  EPI_SCATTER: rows are scaled by their routing weight and stored at (token·k + slot).
//   moe_sum        out[t] = Σ_slots y[t·k + slot] over the slots served by local experts (no atomics:
//                  bitwise reproducible; non-local slots are summed in by the caller's all-reduce).
#include "common.h"
#include "kernels.h"

namespace dli {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int MOE_MAX_E = 64;

// ------------------------------------------------------------------------------------------ routing
// One wave per token, one lane per expert (E ≤ 64): softmax numerators in registers, K rounds of a wave-wide
// arg-max (ties → lowest expert id, like torch.topk on the reference path), lane k keeps the k-th pick.
__global__ __launch_bounds__(256) void moe_route_kernel(const uint16_t* __restrict__ logits, int T, int E, int K,
                                                        int* __restrict__ ids, float* __restrict__ weights) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= T) return;                                   // whole wave
  const float v = lane < E ? bf2f(logits[(int64_t)t * E + lane]) : -3.0e38f;
  const float mx = wave_max(v);
  float p = lane < E ? __expf(v - mx) : -1.f;           // the softmax denominator cancels in the renormalisation
  float picked = 0.f, my_w = 0.f;
  int my_id = 0;
  for (int k = 0; k < K; ++k) {
    float bv = p;
    int bi = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == k) { my_id = bi; my_w = bv; }
    picked += bv;
    if (lane == bi) p = -2.f;
  }
  if (lane < K) {
    ids[t * K + lane] = my_id;
    weights[t * K + lane] = my_w / picked;              // renormalise over the chosen experts
  }
}

// ------------------------------------------------------------------------------------------ alignment
// meta[0] = number of row tiles, meta[1] = padded row count
__global__ __launch_bounds__(1024) void moe_align_kernel(const int* __restrict__ ids, int n, int expert_offset, int E_local,
                                                         int* __restrict__ sorted_ids, int* __restrict__ tile_expert,
                                                         int* __restrict__ meta, int capacity_rows, int tile_rows) {
  const int BM = tile_rows;                               // row tile of the GEMM that consumes the alignment (128 or 256)
  __shared__ int cnt[MOE_MAX_E], off[MOE_MAX_E + 1], pos[MOE_MAX_E];
  for (int e = threadIdx.x; e < MOE_MAX_E; e += blockDim.x) { cnt[e] = 0; pos[e] = 0; }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int e = ids[i] - expert_offset;
    if (e >= 0 && e < E_local) atomicAdd(&cnt[e], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    off[0] = 0;
    for (int e = 0; e < E_local; ++e) {
      const int padded = (cnt[e] + BM - 1) / BM * BM;
      for (int t = off[e] / BM; t < (off[e] + padded) / BM; ++t) tile_expert[t] = e;
      off[e + 1] = off[e] + padded;
    }
    meta[0] = off[E_local] / BM;
    meta[1] = off[E_local];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < capacity_rows; i += blockDim.x) sorted_ids[i] = -1;
  // Placement in the order of the routed-row index (stable), without atomics: every launch produces the same sorted_ids,
  // so WHICH rows share an expert's last tile never changes from launch to launch and a step replays bit-identically
  // whatever the tile variants do with a partly filled tile (graph-vs-eager self-test, greedy reproducibility; the
  // first half-tile version of gemm256.hip summed K in another order than a full tile and exposed the atomics' order).
  // Per chunk of blockDim ids: rank inside the wave by ballots over the experts present in it, wave totals through
  // LDS, then the running position of the expert.
  // One barrier per chunk: the per-wave counts and the running positions are double-buffered by the chunk's parity.
  __shared__ int wcnt[2][16][MOE_MAX_E], run[2][MOE_MAX_E];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  if (threadIdx.x < MOE_MAX_E) run[0][threadIdx.x] = 0;
  const bool few = E_local <= 16;                          // few experts: one ballot per expert, every wave writes every count
  __syncthreads();
  int par = 0;
  for (int base = 0; base < n; base += blockDim.x, par ^= 1) {
    const int i = base + threadIdx.x;
    int e = i < n ? ids[i] - expert_offset : -1;
    if (e >= E_local) e = -1;
    int rank = 0;
    if (few) {
      for (int t = 0; t < E_local; ++t) {
        const unsigned long long same = __ballot(e == t);
        if (e == t) rank = __popcll(same & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[par][wave][t] = __popcll(same);
      }
    } else {
      static_assert(MOE_MAX_E <= 64, "one lane per expert clears the wave's row");
      if (lane < MOE_MAX_E) wcnt[par][wave][lane] = 0;      // this wave's own row: same-wave LDS order, no barrier needed
      unsigned long long todo = __ballot(e >= 0);
      while (todo) {                                        // one round per expert present in this wave
        const int leader = __ffsll((long long)todo) - 1;
        const int le = __builtin_amdgcn_readlane(e, leader);
        const unsigned long long same = __ballot(e == le);
        if (e == le) rank = __popcll(same & ((1ull << lane) - 1ull));
        if (lane == leader) wcnt[par][wave][le] = __popcll(same);
        todo &= ~same;
      }
    }
    __syncthreads();
    if (e >= 0) {
      int before = run[par][e];
      for (int w = 0; w < wave; ++w) before += wcnt[par][w][e];
      sorted_ids[off[e] + before + rank] = i;
    }
    if (threadIdx.x < E_local) {                            // positions for the next chunk, read after ITS barrier
      int add = run[par][threadIdx.x];
      for (int w = 0; w < n_waves; ++w) add += wcnt[par][w][threadIdx.x];
      run[par ^ 1][threadIdx.x] = add;
    }
  }
}

// ------------------------------------------------------------------------------------------ grouped GEMM
enum { EPI_SWIGLU = 0, EPI_SCATTER = 1, EPI_SCATTER_SPLITK = 2 };   // SPLITK: fp32 partial slabs, summed by moe_sum

// byte offset of 16-byte chunk c (0..7) of row r inside a [128][64] bf16 LDS tile, XOR-swizzled so that
// the 16 rows a ds_read_b128 lane group touches fall on different bank groups
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

template <int EPI>
__global__ __launch_bounds__(256) void grouped_gemm_kernel(
    const uint16_t* __restrict__ A, int64_t a_stride,         // EPI_SWIGLU: x [T, K], rows gathered; EPI_SCATTER: act [P, K]
    const uint16_t* __restrict__ W, int64_t w_expert_stride,  // [E_local, N_total, K]
    const int* __restrict__ sorted_ids, const int* __restrict__ tile_expert, const int* __restrict__ meta,
    const float* __restrict__ topk_w, int top_k, void* __restrict__ out_, int64_t out_stride, int K, int N_half,
    int n_col_blocks, int k_splits, int64_t slab_elems) {
  uint16_t* out = reinterpret_cast<uint16_t*>(out_);
  constexpr int STAGE = 2 * BM * BK * 2;                     // A tile + B tile of one K-step
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];   // double-buffered, filled by LDS-DMA
  // ---- block → (row tile, column block), XCD-aware.  The hardware deals workgroups round-robin over the 8
  // XCDs (id % 8), each with a private 4 MB L2.  Work is cut into super-tiles of 8 row tiles × 8 column
  // blocks; XCD k runs super-tiles k, k+8, …, 64 consecutive workgroups each, so the blocks resident on
  // one XCD at a time share 8 A tiles and (row tiles being sorted by expert) 8–16 W tiles: each operand
  // K-slice is fetched into that L2 once and reused 8×, instead of every block streaming a private
  // 1 MB W tile from HBM (the round-robin default puts the 8 row tiles of an expert on 8 different XCDs).
  // With fewer than 8 row tiles (decode-sized batches) the super-tile narrows to 4/2/1 row tiles × 16/32/64
  // column blocks, so that the live blocks still spread over all XCDs.
  const int n_tiles = meta[0];
  const int gx_log = n_tiles >= 5 ? 3 : n_tiles >= 3 ? 2 : n_tiles >= 2 ? 1 : 0;
  const int gy_log = 6 - gx_log;
  const int j = blockIdx.x >> 3;
  const int super = (j >> 6) * 8 + (blockIdx.x & 7);
  const int n_virtual = n_col_blocks * k_splits;             // split-K slices are extra "columns" (slowest index)
  const int sy_count = (n_virtual + (1 << gy_log) - 1) >> gy_log;
  const int sx = super / sy_count, sy = super - sx * sy_count;
  const int tile = (sx << gx_log) + (j & ((1 << gx_log) - 1));
  const int col_virtual = (sy << gy_log) + ((j & 63) >> gx_log);
  if (tile >= n_tiles || col_virtual >= n_virtual) return;
  const int split = col_virtual / n_col_blocks;
  const int col_block = col_virtual - split * n_col_blocks;
  const int e = tile_expert[tile];
  const uint16_t* We = W + (int64_t)e * w_expert_stride;
  const int n0 = col_block * (EPI == EPI_SWIGLU ? BN / 2 : BN);   // SWIGLU: 64 output columns per tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int col = lane & 15, g = lane >> 4;

  // ---- staging: LDS-DMA (global_load_lds, 16 bytes per lane, no VGPRs, no ds_write).  DMA instruction i of wave w
  // fills the 1 KiB at (w + 4i)·1024 of a tile in lane order, i.e. lane → LDS chunk q = tid + 256 i: row q/8,
  // slot q%8.  The bank swizzle of lds_off is applied on the source side: slot p of row r receives global chunk
  // p ^ (r & 7).
  const uint16_t* a_src[4];
  const uint16_t* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + 256 * i, r = q >> 3, c = (q & 7) ^ (r & 7);
    int64_t arow;
    if (EPI == EPI_SWIGLU) {
      const int sid = sorted_ids[tile * BM + r];
      arow = sid >= 0 ? sid / top_k : 0;                     // padding rows read token 0 (discarded later)
    } else {
      arow = (int64_t)tile * BM + r;
    }
    a_src[i] = A + arow * a_stride + c * 8;
    int wrow;
    if (EPI == EPI_SWIGLU) {
      const int w_ = r >> 6, within = r & 63;                 // wave-column half, position inside it
      wrow = (within < 32 ? 0 : N_half) + n0 + w_ * 32 + (within & 31);
    } else {
      wrow = n0 + r;
    }
    b_src[i] = We + (int64_t)wrow * K + c * 8;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this block's K-steps: slice `split` of k_splits (contiguous, sizes differ by at most one step)
  const int nk_total = K / BK;
  const int kt0 = (int)((int64_t)nk_total * split / k_splits);
  const int nk = (int)((int64_t)nk_total * (split + 1) / k_splits);
  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
#define DLI_GG_STAGE(kt_, buf_)                                                                                   \
  {                                                                                                               \
    char* dst_ = lds + (buf_) * STAGE + wave * 1024;                  /* wave-uniform; + lane·16 by hardware */   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                               \
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + (kt_) * BK), (lptr_t)(dst_ + i * 4096), 16, 0, 0);     \
      __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + (kt_) * BK), (lptr_t)(dst_ + BM * BK * 2 + i * 4096), 16, 0, 0); \
    }                                                                                                             \
  }
  DLI_GG_STAGE(kt0, 0);
  for (int kt = kt0; kt < nk; ++kt) {
    const int buf = (kt - kt0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // my share of K-step kt has landed …
    __syncthreads();                                           // … everyone's has, and everyone is done with kt-1
    if (kt + 1 < nk) DLI_GG_STAGE(kt + 1, buf ^ 1);            // next K-step streams in under the MFMAs
    const char* ldsA = lds + buf * STAGE;
    const char* ldsB = ldsA + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) fa[m] = *reinterpret_cast<const bf16x8*>(ldsA + lds_off(wm * 64 + m * 16 + col, ks * 4 + g));
#pragma unroll
      for (int n = 0; n < 4; ++n) fb[n] = *reinterpret_cast<const bf16x8*>(ldsB + lds_off(wn * 64 + n * 16 + col, ks * 4 + g));
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = mfma16(fb[n], fa[m], acc[m][n]);
      // operands swapped on purpose: D[row = weight column n (4g+r)][col = token row m (lane&15)], i.e. each
      // lane holds 4 *consecutive output columns* of one token row → 8-byte row-contiguous stores
    }
  }

#undef DLI_GG_STAGE

  // ---- epilogue.  acc[m][n][r]: token row = wm*64 + m*16 + col, output column = wn*64 + n*16 + 4g + r
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int prow = tile * BM + wm * 64 + m * 16 + col;
    const int sid = sorted_ids[prow];
    if (EPI == EPI_SWIGLU) {
      // wave columns: n = 0,1 gate (32 cols), n = 2,3 the matching up columns
      uint16_t* dst = out + (int64_t)prow * out_stride + n0 + wn * 32 + 4 * g;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gate = acc[m][n][r], up = acc[m][n + 2][r];
          o[r] = (short)f2bf(gate * fast_rcp(1.f + fast_exp2(-1.4426950408889634f * gate)) * up);
        }
        *reinterpret_cast<bf16x4*>(dst + n * 16) = o;           // padding rows hold finite junk, never read back
      }
    } else if (EPI == EPI_SCATTER_SPLITK) {
      if (sid >= 0) {
        const float w = topk_w[sid];
        float* dst = reinterpret_cast<float*>(out_) + split * slab_elems + (int64_t)sid * out_stride + n0 + wn * 64 + 4 * g;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = acc[m][n][r] * w;
          *reinterpret_cast<f32x4*>(dst + n * 16) = o;
        }
      }
    } else if (sid >= 0) {
      const float w = topk_w[sid];
      uint16_t* dst = out + (int64_t)sid * out_stride + n0 + wn * 64 + 4 * g;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (short)f2bf(acc[m][n][r] * w);
        *reinterpret_cast<bf16x4*>(dst + n * 16) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ un-permute
// SLABS == 0: y is bf16 [T·k, H].  SLABS > 0 (runtime count in n_slabs): y is fp32 [n_slabs][T·k, H] split-K partials.
template <bool F32>
__global__ __launch_bounds__(256) void moe_sum_kernel(const void* __restrict__ y_, const int* __restrict__ ids,
                                                      int expert_offset, int E_local, int top_k, int H, int n_slabs,
                                                      int64_t slab_elems, uint16_t* __restrict__ out) {
  const int t = blockIdx.x;
  for (int v = threadIdx.x; v < H / 8; v += blockDim.x) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < top_k; ++s) {
      const int e = ids[t * top_k + s] - expert_offset;
      if (e < 0 || e >= E_local) continue;
      const int64_t row = ((int64_t)t * top_k + s) * H + v * 8;
      if (F32) {
        const float* y = reinterpret_cast<const float*>(y_);
        for (int k = 0; k < n_slabs; ++k) {                 // fixed order: bitwise reproducible
          const f32x4 a = *reinterpret_cast<const f32x4*>(y + k * slab_elems + row);
          const f32x4 b = *reinterpret_cast<const f32x4*>(y + k * slab_elems + row + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc[j] += a[j]; acc[4 + j] += b[j]; }
        }
      } else {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const uint16_t*>(y_) + row);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += bf2f((uint16_t)a[j]);
      }
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(acc[j]);
    *reinterpret_cast<bf16x8*>(out + (int64_t)t * H + v * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------ launchers
void launch_moe_route(const void* logits, int T, int E, int K, int* ids, float* weights, hipStream_t stream) {
  if (T == 0) return;
  hipLaunchKernelGGL(moe_route_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, (const uint16_t*)logits, T, E, K, ids, weights);
}

void launch_moe_align(const int* ids, int n, int expert_offset, int E_local, int* sorted_ids, int* tile_expert, int* meta,
                      int capacity_rows, int tile_rows, hipStream_t stream) {
  hipLaunchKernelGGL(moe_align_kernel, dim3(1), dim3(1024), 0, stream, ids, n, expert_offset, E_local, sorted_ids, tile_expert,
                     meta, capacity_rows, tile_rows);
}

void launch_grouped_gemm(int epilogue, const void* A, int64_t a_stride, const void* W, int64_t w_expert_stride,
                         const int* sorted_ids, const int* tile_expert, const int* meta, const float* topk_w, int top_k,
                         void* out, int64_t out_stride, int K, int N_out, int max_tiles, int k_splits, int64_t slab_elems,
                         hipStream_t stream) {
  if (max_tiles == 0) return;
  const int n_col_blocks = N_out / (epilogue == EPI_SWIGLU ? BN / 2 : BN);
  const int n_virtual = n_col_blocks * k_splits;
  const int supers = (max_tiles + 7) / 8 * ((n_virtual + 7) / 8);
  dim3 grid((supers + 7) / 8 * 8 * 64);                                      // surplus blocks exit on meta[0]
#define DLI_GG(EPI)                                                                                                       \
  hipLaunchKernelGGL(grouped_gemm_kernel<EPI>, grid, dim3(256), 0, stream, (const uint16_t*)A, a_stride, (const uint16_t*)W, \
                     w_expert_stride, sorted_ids, tile_expert, meta, topk_w, top_k, out, out_stride, K, N_out, n_col_blocks, \
                     k_splits, slab_elems)
  if (epilogue == EPI_SWIGLU) DLI_GG(EPI_SWIGLU);
  else if (epilogue == EPI_SCATTER) DLI_GG(EPI_SCATTER);
  else DLI_GG(EPI_SCATTER_SPLITK);
#undef DLI_GG
}

void launch_moe_sum(const void* y, bool y_fp32, int n_slabs, int64_t slab_elems, const int* ids, int expert_offset, int E_local,
                    int top_k, int T, int H, void* out, hipStream_t stream) {
  if (T == 0) return;
  if (y_fp32)
    hipLaunchKernelGGL(moe_sum_kernel<true>, dim3(T), dim3(256), 0, stream, y, ids, expert_offset, E_local, top_k, H, n_slabs,
                       slab_elems, (uint16_t*)out);
  else
    hipLaunchKernelGGL(moe_sum_kernel<false>, dim3(T), dim3(256), 0, stream, y, ids, expert_offset, E_local, top_k, H, 1,
                       slab_elems, (uint16_t*)out);
}

}  // namespace dli
