// K7: all-reduce / all-gather / broadcast over IPC-mapped peer buffers for the fully connected xGMI mesh
// (SURVEY.md §2.5, §5.8).
//
// MI355X nodes wire every GPU to every other GPU with a private xGMI link, so ring algorithms (one
// link busy per GPU, 2(N-1) hops) are the wrong shape for the decode-sized messages of tensor
// parallelism.  The kernels here are *direct*:
//
//   one-shot  (small):  every rank reads all N staging buffers and reduces locally     — 1 hop
//   two-shot  (larger): rank r reduces slice r from its N-1 peers (all links in use at once), then
//                       every rank gathers the N reduced slices                         — 2 hops
//
// The world size is a template parameter (2, 4, 8; anything else takes the generic instance): the
// per-element loop issues the loads from ALL peers before the first add, so a thread keeps
// world × 16 bytes in flight over world-1 different links instead of walking the peers one
// round-trip at a time.
//
// Protocol per call (all inside one launch, so the op is hipGraph-capturable; the per-block epoch
// counter lives in device memory and advances by itself on every replay):
//   1. copy my input into my staging buffer (peers never read torch-allocator memory) — skipped when the input IS the
//      staging buffer: the row-parallel GEMMs write their output there (ar_staging_tensor);
//   2. barrier "start":  every block tells the same block of every peer "my staging is written";
//   3. reduce (one-shot → out; two-shot → my slice of my result buffer, barrier "mid", gather → out);
//   4. barrier "end": nobody may overwrite staging/result before every peer has finished reading.
// Flags are system-scope release/acquire atomics on *uncached* memory (hipExtMallocWithFlags); data
// stores are followed by a system-scope release fence before the flag that publishes them.  Every spin is
// bounded: on timeout the kernel raises `error` in the Signal block of every rank and finishes.  The host copies
// that word back behind every step (`ar_error_async`, no synchronisation) and, when it is set,
// switches the whole TP group to the library collectives for good — a lost peer can never hang the GPU.
// Flags are compared as "epoch reached" (signed distance ≥ 0), so a rank that timed out once cannot
// leave its peers waiting for an epoch that has already gone by.
// Reduction order is rank 0..N-1 on every rank, so all ranks obtain bit-identical sums.
//
// Launch shape (workgroups, one-/two-shot switch point) is a run-time table (`ar_set_tuning`) filled by
// the start-up sweep in parallel/custom_all_reduce.py; every rank of a group must install the same
// table (a block synchronises with the same block index on its peers).
#include <algorithm>
#include <cstring>

#include "common.h"
#include "kernels.h"

namespace dli {

constexpr int kMaxRanks = 8;
constexpr int kMaxBlocks = 128;
constexpr int kThreads = 512;
constexpr int kMaxTuning = 24;
constexpr unsigned kSpinLimit = 4u * 1000u * 1000u;   // ≈ seconds of polling, then give up with `error` set

struct Signal {
  alignas(128) uint32_t start[kMaxBlocks][kMaxRanks];
  alignas(128) uint32_t mid[kMaxBlocks][kMaxRanks];
  alignas(128) uint32_t end[kMaxBlocks][kMaxRanks];
  alignas(128) uint32_t epoch[kMaxBlocks];
  alignas(128) uint32_t error;
};

struct Peers {
  uint16_t* staging[kMaxRanks];   // [max_elems] input copies
  uint16_t* result[kMaxRanks];    // [max_elems] two-shot reduced slices
  Signal* signal[kMaxRanks];
};

struct ARState {
  Peers peers;
  int rank, world;
  size_t max_bytes;
  // launch-shape table: first entry with bytes <= thr wins; past the last entry the defaults apply
  int n_tuning;
  int64_t thr[kMaxTuning];
  int algo[kMaxTuning];     // 1 one-shot, 2 two-shot
  int blocks[kMaxTuning];
  int default_blocks;
  int64_t one_shot_max_bytes;
};

template <int PHASE>   // 0 start, 1 mid, 2 end
__device__ __forceinline__ void mesh_barrier(const Peers& p, int rank, int world, uint32_t flag) {
  // publish/consume recipe of the CDNA guide (§6 G16): every storing wave drains its stores, the block
  // meets, ONE wave releases (L2 write-back) and waits for it, then raises the flags; after the poll the
  // same wave acquires (cache invalidate) and WAITS for the invalidate before the block is released —
  // buffer_inv completes asynchronously, and without that wait the other waves can read stale lines of a
  // peer's staging buffer (same addresses are reused by every call).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < world) {
    Signal* peer = p.signal[threadIdx.x];
    Signal* self = p.signal[rank];
    uint32_t* dst = PHASE == 0 ? &peer->start[blockIdx.x][rank] : PHASE == 1 ? &peer->mid[blockIdx.x][rank] : &peer->end[blockIdx.x][rank];
    uint32_t* src = PHASE == 0 ? &self->start[blockIdx.x][threadIdx.x] : PHASE == 1 ? &self->mid[blockIdx.x][threadIdx.x] : &self->end[blockIdx.x][threadIdx.x];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                 // system scope
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(dst, flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned spins = 0;
    while ((int32_t)(__hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - flag) < 0) {
      if (++spins > kSpinLimit) {
        // raised in EVERY rank's signal block: the sums of this call are partial on whoever reads this rank's data, so
        // each rank's host must learn of the fault with the step it happened in (the driver discards that step)
        for (int r = 0; r < world; ++r)
          __hip_atomic_store(&p.signal[r]->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

__device__ __forceinline__ void accum(float (&acc)[8], bf16x8 v) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] += bf2f((uint16_t)v[j]);
}
__device__ __forceinline__ bf16x8 pack(const float (&acc)[8]) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(acc[j]);
  return o;
}

// Sum of element i over the staging buffers of all ranks, in rank order.  W > 0: all loads issued first.
template <int W>
__device__ __forceinline__ bf16x8 reduce_at(const Peers& p, int world, int64_t i) {
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (W > 0) {
    bf16x8 v[W];
#pragma unroll
    for (int r = 0; r < W; ++r) v[r] = reinterpret_cast<const bf16x8*>(p.staging[r])[i];
#pragma unroll
    for (int r = 0; r < W; ++r) accum(acc, v[r]);
  } else {
    for (int r = 0; r < world; ++r) accum(acc, reinterpret_cast<const bf16x8*>(p.staging[r])[i]);
  }
  return pack(acc);
}

template <int W, bool TWO_SHOT>
__global__ __launch_bounds__(kThreads) void all_reduce_kernel(Peers p, int rank, int world_rt, const uint16_t* __restrict__ in,
                                                              uint16_t* __restrict__ out, int64_t n_vec) {
  const int world = W > 0 ? W : world_rt;
  Signal* self = p.signal[rank];
  const uint32_t flag = self->epoch[blockIdx.x] + 1;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bf16x8* src = reinterpret_cast<const bf16x8*>(in);
  bf16x8* mine = reinterpret_cast<bf16x8*>(p.staging[rank]);

  // input already IN my staging buffer (the producing GEMM wrote there: RowParallelLinear + ar_staging_tensor) → no copy;
  // the data is an earlier kernel's, visible to the peers once they pass the barrier
  if (src != mine)
    for (int64_t i = tid; i < n_vec; i += stride) mine[i] = src[i];
  mesh_barrier<0>(p, rank, world, flag);

  bf16x8* dst = reinterpret_cast<bf16x8*>(out);
  if (!TWO_SHOT) {
    for (int64_t i = tid; i < n_vec; i += stride) dst[i] = reduce_at<W>(p, world, i);
  } else {
    // slice r = [r*per, min((r+1)*per, n_vec)).  Element i is handled by the thread with i ≡ tid (mod stride) on
    // every rank in every phase — the thread that copied element i into its rank's staging buffer — so the
    // per-block barriers order exactly the data each block consumes (a block only waits for the SAME block of
    // its peers).
    const int64_t per = (n_vec + world - 1) / world;
    const int64_t lo = rank * per, hi = min(lo + per, n_vec);
    bf16x8* res = reinterpret_cast<bf16x8*>(p.result[rank]);
    for (int64_t i = lo + ((tid - lo % stride + stride) % stride); i < hi; i += stride) res[i] = reduce_at<W>(p, world, i);
    mesh_barrier<1>(p, rank, world, flag);
    constexpr int U = 4;                                    // loads in flight per thread in the gather
    for (int64_t i0 = tid; i0 < n_vec; i0 += stride * U) {
      bf16x8 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + u * stride;
        if (i < n_vec) v[u] = reinterpret_cast<const bf16x8*>(p.result[i / per])[i];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + u * stride;
        if (i < n_vec) dst[i] = v[u];
      }
    }
  }
  mesh_barrier<2>(p, rank, world, flag);
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = flag;
}

// Sequence-parallel form of "all-reduce → residual add → RMSNorm" (the pair that follows every row-parallel GEMM of a
// transformer layer), in ONE kernel on the two-shot protocol: rank r reduces ROWS [r·rpr, (r+1)·rpr) of the partial sums
// from all peers' staging buffers, adds its rows of the residual stream (and writes them back: from here on every rank
// carries the residual of its own rows only), normalises them and publishes the normalised rows in its result buffer; after
// the mid barrier every rank gathers all rows.  Against all-reduce + a separate fused_add_rms_norm over all T rows on every
// rank, the norm runs on T / world rows (one TP = 8 rank of Llama-3-70B spends 8.3 % of a saturated step in that kernel:
// profiles/tp8_shard_emulation_r3.md) and the [T, H] intermediate never exists.
// The partial sums must already BE in the staging buffer (the producing GEMM writes there: ar_staging_tensor) — an earlier
// kernel's data, visible to every peer after any block's start barrier.  Row → block mapping (row of a slice ≡ block index
// mod grid) is the same in the reduce and the gather phase on every rank, so the per-block mid barrier orders exactly the
// rows a block gathers.  One 512-thread block holds a row of up to 8192 elements in registers between the two passes.
// Numerics mirror the unfused pair: the sum is rounded to bf16 before the residual add.
template <int W>
__global__ __launch_bounds__(kThreads) void all_reduce_add_norm_kernel(Peers p, int rank, int world_rt, uint16_t* __restrict__ residual,
                                                                       const uint16_t* __restrict__ weight, float eps,
                                                                       uint16_t* __restrict__ out, int T, int H) {
  const int world = W > 0 ? W : world_rt;
  Signal* self = p.signal[rank];
  const uint32_t flag = self->epoch[blockIdx.x] + 1;
  __shared__ float red[kThreads / 64];
  mesh_barrier<0>(p, rank, world, flag);
  const int vpr = H >> 3;
  const int rpr = (T + world - 1) / world;
  const int r_end = min((rank + 1) * rpr, T);
  bf16x8* mine = reinterpret_cast<bf16x8*>(p.result[rank]);
  for (int row = rank * rpr + blockIdx.x; row < r_end; row += gridDim.x) {
    float vals[2][8];
    bf16x8 gw[2];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = threadIdx.x + i * kThreads;
      if (v < vpr) {
        const bf16x8 s = reduce_at<W>(p, world, (int64_t)row * vpr + v);
        gw[i] = *reinterpret_cast<const bf16x8*>(weight + v * 8);
        bf16x8* rp = reinterpret_cast<bf16x8*>(residual + (int64_t)row * H) + v;
        const bf16x8 r = *rp;
        bf16x8 nr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = bf2f((uint16_t)s[j]) + bf2f((uint16_t)r[j]);
          vals[i][j] = f;
          nr[j] = (short)f2bf(f);
          ss += f * f;
        }
        *rp = nr;
      }
    }
    const float inv = rsqrtf(block_sum<kThreads>(ss, red) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = threadIdx.x + i * kThreads;
      if (v < vpr) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(vals[i][j] * inv * bf2f((uint16_t)gw[i][j]));
        mine[(int64_t)row * vpr + v] = o;
      }
    }
  }
  mesh_barrier<1>(p, rank, world, flag);
  bf16x8* dst = reinterpret_cast<bf16x8*>(out);
  for (int q = 0; q < world; ++q) {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(p.result[q]);
    const int q_end = min((q + 1) * rpr, T);
    for (int row = q * rpr + blockIdx.x; row < q_end; row += gridDim.x)
      for (int v = threadIdx.x; v < vpr; v += kThreads) dst[(int64_t)row * vpr + v] = src[(int64_t)row * vpr + v];
  }
  mesh_barrier<2>(p, rank, world, flag);
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = flag;
}

// All-gather with the same protocol: out[r][i] = in_r[i] (rank-major).  Used for the vocab-parallel logits so
// that a TP decode graph contains no library collective at all.
__global__ __launch_bounds__(kThreads) void all_gather_kernel(Peers p, int rank, int world, const uint16_t* __restrict__ in,
                                                              uint16_t* __restrict__ out, int64_t n_vec) {
  Signal* self = p.signal[rank];
  const uint32_t flag = self->epoch[blockIdx.x] + 1;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bf16x8* src = reinterpret_cast<const bf16x8*>(in);
  bf16x8* mine = reinterpret_cast<bf16x8*>(p.staging[rank]);
  for (int64_t i = tid; i < n_vec; i += stride) mine[i] = src[i];
  mesh_barrier<0>(p, rank, world, flag);
  bf16x8* dst = reinterpret_cast<bf16x8*>(out);
  for (int r = 0; r < world; ++r) {
    const bf16x8* peer = reinterpret_cast<const bf16x8*>(p.staging[r]);
    for (int64_t i = tid; i < n_vec; i += stride) dst[r * n_vec + i] = peer[i];
  }
  mesh_barrier<2>(p, rank, world, flag);
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = flag;
}

// Broadcast from `root` (any dtype, whole 16-byte vectors): the driver's packed step buffer reaches the TP
// workers without a library collective.  `buf` is read on the root and written everywhere else.
__global__ __launch_bounds__(kThreads) void broadcast_kernel(Peers p, int rank, int world, int root, uint16_t* __restrict__ buf,
                                                             int64_t n_vec) {
  Signal* self = p.signal[rank];
  const uint32_t flag = self->epoch[blockIdx.x] + 1;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bf16x8* mine = reinterpret_cast<bf16x8*>(buf);
  if (rank == root) {
    bf16x8* stage = reinterpret_cast<bf16x8*>(p.staging[rank]);
    for (int64_t i = tid; i < n_vec; i += stride) stage[i] = mine[i];
  }
  mesh_barrier<0>(p, rank, world, flag);
  if (rank != root) {
    const bf16x8* peer = reinterpret_cast<const bf16x8*>(p.staging[root]);
    for (int64_t i = tid; i < n_vec; i += stride) mine[i] = peer[i];
  }
  mesh_barrier<2>(p, rank, world, flag);
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = flag;
}

// ==================================================================================================
// Expert-parallel dispatch / combine over the same IPC buffers (K8 "EP dispatch/combine via all-to-all"): every rank
// routes ITS slice of the step's tokens, the (token, expert) rows travel to the rank that owns the expert, the expert
// outputs travel back.  No host in the loop: the per-destination counts live in device memory that the peers read.
//
//   ep_plan   (1 workgroup)   bucket my n = tokens·k pairs by destination rank → slot[i] (row in my send area, grouped
//                             by destination) and send_off[0..world] in the EpMeta block at the tail of MY staging buffer
//   ep_pack                   copy the pairs' activation rows, expert ids and gate weights into MY staging buffer at slot[i]
//   ep_exchange (B blocks)    barrier → read every peer's send_off → pull "rows for me" from each peer's staging into a
//                             local compact buffer (ordered by source rank), ids / weights alongside, -1 ids behind the
//                             last valid row; publish recv_off[src] (where each source's rows start) → barrier
//   … grouped GEMMs of the local experts on the received rows, output written into MY result buffer (IPC-visible) …
//   ep_combine (B blocks)     barrier → every pair pulls its output row from the owner's result buffer → barrier
//
// All cross-rank data is produced by an EARLIER kernel of the producing rank and read after a mesh barrier inside the
// consuming kernel — the publish / consume recipe of the all-reduce kernels (release before the flag, acquire + invalidate
// after the wait), so one barrier per kernel suffices and no block pairing is needed.  Same grid on every rank.
// ==================================================================================================
struct EpMeta {
  int send_off[kMaxRanks + 1];   // my staging rows [send_off[d], send_off[d+1]) are for rank d
  int recv_off[kMaxRanks + 1];   // in my result buffer, the rows received from rank s start at recv_off[s]
};
constexpr int kEpMetaBytes = 512;

__global__ __launch_bounds__(1024) void ep_plan_kernel(const int* __restrict__ ids, int n, int local_experts, int world,
                                                       int* __restrict__ slot, EpMeta* __restrict__ meta) {
  __shared__ int cnt[kMaxRanks], off[kMaxRanks + 1], fill[kMaxRanks];
  if (threadIdx.x < kMaxRanks) { cnt[threadIdx.x] = 0; fill[threadIdx.x] = 0; }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&cnt[min(ids[i] / local_experts, world - 1)], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    off[0] = 0;
    for (int d = 0; d < world; ++d) off[d + 1] = off[d] + cnt[d];
    for (int d = 0; d <= world; ++d) meta->send_off[d] = off[d];
  }
  __syncthreads();
  // slots in the order of the routed-row index (stable, no atomics): the receiving rank's alignment keeps that order, and
  // which rows share an expert's last GEMM tile must not change between launches (see moe_align_kernel)
  __shared__ int wcnt[16][kMaxRanks];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  for (int base = 0; base < n; base += blockDim.x) {
    if (threadIdx.x < 16 * kMaxRanks) (&wcnt[0][0])[threadIdx.x] = 0;
    __syncthreads();
    const int i = base + threadIdx.x;
    const int d = i < n ? min(ids[i] / local_experts, world - 1) : -1;
    int rank = 0;
    for (int t = 0; t < world; ++t) {
      const unsigned long long same = __ballot(d == t);
      if (d == t) rank = __popcll(same & ((1ull << lane) - 1ull));
      if (lane == 0) wcnt[wave][t] = __popcll(same);
    }
    __syncthreads();
    if (d >= 0) {
      int before = fill[d];
      for (int w = 0; w < wave; ++w) before += wcnt[w][d];
      slot[i] = off[d] + before + rank;
    }
    __syncthreads();
    if (threadIdx.x < world) {
      int add = 0;
      for (int w = 0; w < n_waves; ++w) add += wcnt[w][threadIdx.x];
      fill[threadIdx.x] += add;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void ep_pack_kernel(const uint16_t* __restrict__ x, int64_t x_stride, const int* __restrict__ ids,
                                                      const float* __restrict__ wts, const int* __restrict__ slot, int n, int top_k,
                                                      int H, uint16_t* __restrict__ rows, int* __restrict__ ids_st,
                                                      float* __restrict__ wts_st) {
  const int vpr = H / 8;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int s = slot[i];
    const bf16x8* src = reinterpret_cast<const bf16x8*>(x + (int64_t)(i / top_k) * x_stride);
    bf16x8* dst = reinterpret_cast<bf16x8*>(rows + (int64_t)s * H);
    for (int v = threadIdx.x; v < vpr; v += blockDim.x) dst[v] = src[v];
    if (threadIdx.x == 0) { ids_st[s] = ids[i]; wts_st[s] = wts[i]; }
  }
}

__global__ __launch_bounds__(kThreads) void ep_exchange_kernel(Peers p, int rank, int world, int H, int64_t ids_off,
                                                               int64_t wts_off, int64_t meta_off, uint16_t* __restrict__ recv_x,
                                                               int* __restrict__ ids_local, float* __restrict__ w_local,
                                                               int r_cap, int* __restrict__ recv_total) {
  Signal* self = p.signal[rank];
  const uint32_t flag = self->epoch[blockIdx.x] + 1;
  mesh_barrier<0>(p, rank, world, flag);
  __shared__ int s_off[kMaxRanks], roff[kMaxRanks + 1];
  if (threadIdx.x < world) {
    const EpMeta* m = reinterpret_cast<const EpMeta*>(reinterpret_cast<const char*>(p.staging[threadIdx.x]) + meta_off);
    s_off[threadIdx.x] = m->send_off[rank];
    roff[threadIdx.x + 1] = m->send_off[rank + 1] - m->send_off[rank];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    roff[0] = 0;
    for (int s = 0; s < world; ++s) roff[s + 1] = min(roff[s] + roff[s + 1], r_cap);
  }
  __syncthreads();
  const int R = roff[world];
  const int vpr = H / 8;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = tid; idx < (int64_t)R * vpr; idx += stride) {
    const int r = (int)(idx / vpr), v = (int)(idx - (int64_t)r * vpr);
    int s = 0;
    while (r >= roff[s + 1]) ++s;
    const int64_t srow = s_off[s] + (r - roff[s]);
    reinterpret_cast<bf16x8*>(recv_x)[idx] = reinterpret_cast<const bf16x8*>(p.staging[s])[srow * vpr + v];
  }
  for (int64_t r = tid; r < r_cap; r += stride) {
    if (r < R) {
      int s = 0;
      while (r >= roff[s + 1]) ++s;
      const int64_t srow = s_off[s] + (r - roff[s]);
      const char* base = reinterpret_cast<const char*>(p.staging[s]);
      ids_local[r] = reinterpret_cast<const int*>(base + ids_off)[srow];
      w_local[r] = reinterpret_cast<const float*>(base + wts_off)[srow];
    } else {
      ids_local[r] = -1;                                     // no expert: the alignment kernel skips the row
      w_local[r] = 0.f;
    }
  }
  if (blockIdx.x == 0) {
    EpMeta* mine = reinterpret_cast<EpMeta*>(reinterpret_cast<char*>(p.staging[rank]) + meta_off);
    if (threadIdx.x <= world) mine->recv_off[threadIdx.x] = roff[threadIdx.x];
    if (threadIdx.x == 0) *recv_total = R;
  }
  mesh_barrier<2>(p, rank, world, flag);
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = flag;
}

__global__ __launch_bounds__(kThreads) void ep_combine_kernel(Peers p, int rank, int world, int H, int64_t meta_off,
                                                              const int* __restrict__ ids, const int* __restrict__ slot, int n,
                                                              int local_experts, uint16_t* __restrict__ y_back) {
  Signal* self = p.signal[rank];
  const uint32_t flag = self->epoch[blockIdx.x] + 1;
  mesh_barrier<0>(p, rank, world, flag);
  __shared__ int send_off[kMaxRanks + 1], peer_roff[kMaxRanks];
  if (threadIdx.x <= world)
    send_off[threadIdx.x] = reinterpret_cast<const EpMeta*>(reinterpret_cast<const char*>(p.staging[rank]) + meta_off)->send_off[threadIdx.x];
  if (threadIdx.x < world)
    peer_roff[threadIdx.x] = reinterpret_cast<const EpMeta*>(reinterpret_cast<const char*>(p.staging[threadIdx.x]) + meta_off)->recv_off[rank];
  __syncthreads();
  const int vpr = H / 8;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = tid; idx < (int64_t)n * vpr; idx += stride) {
    const int i = (int)(idx / vpr), v = (int)(idx - (int64_t)i * vpr);
    const int d = min(ids[i] / local_experts, world - 1);
    const int64_t row = peer_roff[d] + (slot[i] - send_off[d]);
    reinterpret_cast<bf16x8*>(y_back)[idx] = reinterpret_cast<const bf16x8*>(p.result[d])[row * vpr + v];
  }
  mesh_barrier<2>(p, rank, world, flag);
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = flag;
}

// Byte layout of the send area inside a staging buffer for at most n_cap pairs of H-wide rows; < 0 if it does not fit.
int64_t ep_layout(void* st, int64_t n_cap, int H, int64_t* ids_off, int64_t* wts_off, int64_t* meta_off) {
  ARState* s = reinterpret_cast<ARState*>(st);
  const int64_t rows = (n_cap * H * 2 + 255) / 256 * 256;
  *ids_off = rows;
  *wts_off = rows + (n_cap * 4 + 255) / 256 * 256;
  *meta_off = (int64_t)s->max_bytes - kEpMetaBytes;
  return (*wts_off + n_cap * 4 <= *meta_off) ? 0 : -1;
}
void* ar_staging_ptr(void* st) { ARState* s = reinterpret_cast<ARState*>(st); return s->peers.staging[s->rank]; }
void* ep_result_ptr(void* st) { ARState* s = reinterpret_cast<ARState*>(st); return s->peers.result[s->rank]; }
int64_t ep_max_bytes(void* st) { return (int64_t) reinterpret_cast<ARState*>(st)->max_bytes; }

int launch_ep_dispatch(void* st, const void* x, int64_t x_stride, const int* ids, const float* wts, int n, int n_cap, int top_k,
                       int H, int local_experts, int* slot, void* recv_x, int* ids_local, float* w_local, int r_cap,
                       int* recv_total, int blocks, hipStream_t stream) {
  ARState* s = reinterpret_cast<ARState*>(st);
  int64_t ids_off, wts_off, meta_off;
  if (H % 8 || n > n_cap || ep_layout(st, n_cap, H, &ids_off, &wts_off, &meta_off) < 0) return -1;
  if ((int64_t)r_cap * H * 2 > (int64_t)s->max_bytes) return -1;
  char* mine = reinterpret_cast<char*>(s->peers.staging[s->rank]);
  hipLaunchKernelGGL(ep_plan_kernel, dim3(1), dim3(1024), 0, stream, ids, n, local_experts, s->world, slot,
                     reinterpret_cast<EpMeta*>(mine + meta_off));
  if (n > 0)
    hipLaunchKernelGGL(ep_pack_kernel, dim3(std::min(n, 2048)), dim3(256), 0, stream, (const uint16_t*)x, x_stride, ids, wts, slot, n,
                       top_k, H, reinterpret_cast<uint16_t*>(mine), reinterpret_cast<int*>(mine + ids_off),
                       reinterpret_cast<float*>(mine + wts_off));
  blocks = std::max(1, std::min(blocks, kMaxBlocks));
  hipLaunchKernelGGL(ep_exchange_kernel, dim3(blocks), dim3(kThreads), 0, stream, s->peers, s->rank, s->world, H, ids_off, wts_off,
                     meta_off, (uint16_t*)recv_x, ids_local, w_local, r_cap, recv_total);
  return 0;
}

int launch_ep_combine(void* st, const int* ids, const int* slot, int n, int n_cap, int H, int local_experts, void* y_back, int blocks,
                      hipStream_t stream) {
  ARState* s = reinterpret_cast<ARState*>(st);
  int64_t ids_off, wts_off, meta_off;
  if (ep_layout(st, n_cap, H, &ids_off, &wts_off, &meta_off) < 0) return -1;
  blocks = std::max(1, std::min(blocks, kMaxBlocks));
  hipLaunchKernelGGL(ep_combine_kernel, dim3(blocks), dim3(kThreads), 0, stream, s->peers, s->rank, s->world, H, meta_off, ids, slot, n,
                     local_experts, (uint16_t*)y_back);
  return 0;
}

// ---- host API (opaque handle; pointers are exchanged by the python layer through hipIpc handles) ----
void* ar_alloc(size_t bytes, bool uncached) {
  void* p = nullptr;
  hipError_t e = uncached ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) : hipMalloc(&p, bytes);
  if (e != hipSuccess) return nullptr;
  (void)hipMemset(p, 0, bytes);
  (void)hipDeviceSynchronize();
  return p;
}
void ar_free(void* p) { (void)hipFree(p); }

int ar_ipc_handle(void* p, void* handle_out64) {
  return (int)hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle_out64), p);
}
void* ar_ipc_open(const void* handle64) {
  void* p = nullptr;
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle64, sizeof(h));
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return nullptr;
  return p;
}
int ar_ipc_close(void* p) { return (int)hipIpcCloseMemHandle(p); }
size_t ar_signal_bytes() { return sizeof(Signal); }
int ar_max_blocks() { return kMaxBlocks; }

void* ar_create(int rank, int world, const int64_t* staging, const int64_t* result, const int64_t* signal, size_t max_bytes) {
  if (world > kMaxRanks || world < 2 || rank < 0 || rank >= world) return nullptr;
  ARState* s = new ARState();
  s->rank = rank;
  s->world = world;
  s->max_bytes = max_bytes;
  s->n_tuning = 0;
  s->default_blocks = 64;
  s->one_shot_max_bytes = 256 * 1024;
  for (int r = 0; r < world; ++r) {
    s->peers.staging[r] = reinterpret_cast<uint16_t*>(staging[r]);
    s->peers.result[r] = reinterpret_cast<uint16_t*>(result[r]);
    s->peers.signal[r] = reinterpret_cast<Signal*>(signal[r]);
  }
  return s;
}
void ar_destroy(void* st) { delete reinterpret_cast<ARState*>(st); }

int ar_set_tuning(void* st, int n, const int64_t* thr, const int* algo, const int* blocks, int default_blocks,
                  int64_t one_shot_max_bytes) {
  ARState* s = reinterpret_cast<ARState*>(st);
  if (n < 0 || n > kMaxTuning) return -1;
  for (int i = 0; i < n; ++i) {
    if ((algo[i] != 1 && algo[i] != 2) || blocks[i] < 1 || blocks[i] > kMaxBlocks) return -1;
    if (i > 0 && thr[i] <= thr[i - 1]) return -1;
  }
  if (default_blocks < 1 || default_blocks > kMaxBlocks) return -1;
  s->n_tuning = n;
  for (int i = 0; i < n; ++i) {
    s->thr[i] = thr[i];
    s->algo[i] = algo[i];
    s->blocks[i] = blocks[i];
  }
  s->default_blocks = default_blocks;
  s->one_shot_max_bytes = one_shot_max_bytes;
  return 0;
}

int ar_error(void* st) {      // blocking host read of the timeout word (tests, shutdown)
  ARState* s = reinterpret_cast<ARState*>(st);
  uint32_t v = 0;
  (void)hipMemcpy(&v, &s->peers.signal[s->rank]->error, sizeof(v), hipMemcpyDeviceToHost);
  return (int)v;
}

// stream-ordered copy of the timeout word into pinned host memory: read it after the step's event
int ar_error_async(void* st, void* pinned_dst4, hipStream_t stream) {
  ARState* s = reinterpret_cast<ARState*>(st);
  return (int)hipMemcpyAsync(pinned_dst4, &s->peers.signal[s->rank]->error, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
}

static int clamp_blocks(int64_t n_vec, int blocks) {
  const int64_t need = (n_vec + kThreads - 1) / kThreads;
  return (int)std::max<int64_t>(1, std::min<int64_t>({need, (int64_t)blocks, (int64_t)kMaxBlocks}));
}

template <bool TWO>
static void launch_ar(const ARState* s, int blocks, const void* in, void* out, int64_t n_vec, hipStream_t stream) {
  const uint16_t* i = (const uint16_t*)in;
  uint16_t* o = (uint16_t*)out;
  switch (s->world) {
    case 2: hipLaunchKernelGGL((all_reduce_kernel<2, TWO>), dim3(blocks), dim3(kThreads), 0, stream, s->peers, s->rank, 2, i, o, n_vec); break;
    case 4: hipLaunchKernelGGL((all_reduce_kernel<4, TWO>), dim3(blocks), dim3(kThreads), 0, stream, s->peers, s->rank, 4, i, o, n_vec); break;
    case 8: hipLaunchKernelGGL((all_reduce_kernel<8, TWO>), dim3(blocks), dim3(kThreads), 0, stream, s->peers, s->rank, 8, i, o, n_vec); break;
    default: hipLaunchKernelGGL((all_reduce_kernel<0, TWO>), dim3(blocks), dim3(kThreads), 0, stream, s->peers, s->rank, s->world, i, o, n_vec);
  }
}

// algo: 0 = table / default, 1 one-shot, 2 two-shot.  blocks: 0 = table / default.
int launch_all_reduce(void* st, const void* in, void* out, int64_t bytes, int algo, int blocks, hipStream_t stream) {
  ARState* s = reinterpret_cast<ARState*>(st);
  if (bytes <= 0 || bytes % 16 != 0 || (size_t)bytes > s->max_bytes) return -1;
  const int64_t n_vec = bytes / 16;
  int t_algo = 0, t_blocks = s->default_blocks;
  for (int i = 0; i < s->n_tuning; ++i)
    if (bytes <= s->thr[i]) {
      t_algo = s->algo[i];
      t_blocks = s->blocks[i];
      break;
    }
  if (algo == 0) algo = t_algo ? t_algo : (bytes > s->one_shot_max_bytes && s->world > 2 ? 2 : 1);
  blocks = clamp_blocks(n_vec, blocks > 0 ? blocks : t_blocks);
  if (algo == 2)
    launch_ar<true>(s, blocks, in, out, n_vec, stream);
  else
    launch_ar<false>(s, blocks, in, out, n_vec, stream);
  return 0;
}

// staged: the partial sums [T, H] live in this rank's staging buffer.  blocks: same value on every rank.
int launch_all_reduce_add_norm(void* st, void* residual, const void* weight, float eps, void* out, int T, int H, int blocks,
                               hipStream_t stream) {
  ARState* s = reinterpret_cast<ARState*>(st);
  if (T <= 0 || H % 8 != 0 || H > 16 * kThreads || (size_t)T * H * 2 > s->max_bytes) return -1;
  const int rpr = (T + s->world - 1) / s->world;
  blocks = std::max(1, std::min({blocks > 0 ? blocks : 64, rpr, kMaxBlocks}));
  uint16_t* r = (uint16_t*)residual;
  const uint16_t* w = (const uint16_t*)weight;
  uint16_t* o = (uint16_t*)out;
  switch (s->world) {
    case 2: hipLaunchKernelGGL(all_reduce_add_norm_kernel<2>, dim3(blocks), dim3(kThreads), 0, stream, s->peers, s->rank, 2, r, w, eps, o, T, H); break;
    case 4: hipLaunchKernelGGL(all_reduce_add_norm_kernel<4>, dim3(blocks), dim3(kThreads), 0, stream, s->peers, s->rank, 4, r, w, eps, o, T, H); break;
    case 8: hipLaunchKernelGGL(all_reduce_add_norm_kernel<8>, dim3(blocks), dim3(kThreads), 0, stream, s->peers, s->rank, 8, r, w, eps, o, T, H); break;
    default: hipLaunchKernelGGL(all_reduce_add_norm_kernel<0>, dim3(blocks), dim3(kThreads), 0, stream, s->peers, s->rank, s->world, r, w, eps, o, T, H);
  }
  return 0;
}

int launch_all_gather(void* st, const void* in, void* out, int64_t bytes, hipStream_t stream) {
  ARState* s = reinterpret_cast<ARState*>(st);
  if (bytes <= 0 || bytes % 16 != 0 || (size_t)bytes > s->max_bytes) return -1;
  const int64_t n_vec = bytes / 16;
  const int blocks = clamp_blocks(n_vec, s->default_blocks);
  hipLaunchKernelGGL(all_gather_kernel, dim3(blocks), dim3(kThreads), 0, stream, s->peers, s->rank, s->world, (const uint16_t*)in,
                     (uint16_t*)out, n_vec);
  return 0;
}

int launch_broadcast(void* st, void* buf, int64_t bytes, int root, hipStream_t stream) {
  ARState* s = reinterpret_cast<ARState*>(st);
  if (bytes <= 0 || bytes % 16 != 0 || (size_t)bytes > s->max_bytes || root < 0 || root >= s->world) return -1;
  const int64_t n_vec = bytes / 16;
  const int blocks = clamp_blocks(n_vec, s->default_blocks);
  hipLaunchKernelGGL(broadcast_kernel, dim3(blocks), dim3(kThreads), 0, stream, s->peers, s->rank, s->world, root, (uint16_t*)buf,
                     n_vec);
  return 0;
}

}  // namespace dli
