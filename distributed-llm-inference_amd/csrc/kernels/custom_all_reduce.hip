// K7: all-reduce over IPC-mapped peer buffers for the fully connected xGMI mesh.
//
// MI355X nodes wire every GPU to every other GPU with a private xGMI link, so ring algorithms (one
// link busy per GPU, 2(N-1) hops) are the wrong shape for the decode-sized messages of tensor
// parallelism.  Both kernels here are *direct*:
//
//   one-shot  (small):  every rank reads all N staging buffers and reduces locally     — 1 hop
//   two-shot  (larger): rank r reduces slice r from its N-1 peers (all links in use at once), then
//                       every rank gathers the N reduced slices                         — 2 hops
//
// Protocol per call (all inside one launch, so the op is hipGraph-capturable; the per-block epoch
// counter lives in device memory and advances by itself on every replay):
//   1. copy my input into my staging buffer (peers never read torch-allocator memory);
//   2. barrier "start":  every block tells the same block of every peer "my staging is written";
//   3. reduce (one-shot → out; two-shot → my slice of my result buffer, barrier "mid", gather → out);
//   4. barrier "end": nobody may overwrite staging/result before every peer has finished reading.
// Flags are system-scope release/acquire atomics on *uncached* memory (hipExtMallocWithFlags); data
// stores are followed by __threadfence_system() before the flag that publishes them.  Every spin is
// bounded: on timeout the kernel raises `error` in its Signal block and finishes (the host checks the
// word and falls back to RCCL) — a lost peer can never hang the GPU.
// Reduction order is rank 0..N-1 on every rank, so all ranks obtain bit-identical sums.
#include <algorithm>
#include <cstring>

#include "common.h"
#include "kernels.h"

namespace dli {

constexpr int kMaxRanks = 8;
constexpr int kMaxBlocks = 64;
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
    while (__hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != flag) {
      if (++spins > kSpinLimit) {
        __hip_atomic_store(&self->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

template <bool TWO_SHOT>
__global__ __launch_bounds__(512) void all_reduce_kernel(Peers p, int rank, int world, const uint16_t* __restrict__ in,
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
  if (!TWO_SHOT) {
    for (int64_t i = tid; i < n_vec; i += stride) {
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int r = 0; r < world; ++r) accum(acc, reinterpret_cast<const bf16x8*>(p.staging[r])[i]);
      dst[i] = pack(acc);
    }
  } else {
    const int64_t per = (n_vec + world - 1) / world;        // slice r = [r*per, min((r+1)*per, n_vec))
    const int64_t lo = rank * per, hi = min(lo + per, n_vec);
    bf16x8* res = reinterpret_cast<bf16x8*>(p.result[rank]);
    // index i is handled by the SAME (block, thread) on every rank in every phase, so the per-block
    // barriers order exactly the data each block consumes
    for (int64_t i = tid; i < hi; i += stride) {
      if (i < lo) continue;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int r = 0; r < world; ++r) accum(acc, reinterpret_cast<const bf16x8*>(p.staging[r])[i]);
      res[i] = pack(acc);
    }
    mesh_barrier<1>(p, rank, world, flag);
    for (int64_t i = tid; i < n_vec; i += stride) {
      const int owner = (int)(i / per);
      dst[i] = reinterpret_cast<const bf16x8*>(p.result[owner])[i];
    }
  }
  mesh_barrier<2>(p, rank, world, flag);
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = flag;
}

// All-gather with the same protocol: out[r][i] = in_r[i] (rank-major).  Used for the vocab-parallel logits so
// that a TP decode graph contains no library collective at all.
__global__ __launch_bounds__(512) void all_gather_kernel(Peers p, int rank, int world, const uint16_t* __restrict__ in,
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
size_t ar_signal_bytes() { return sizeof(Signal); }

void* ar_create(int rank, int world, const int64_t* staging, const int64_t* result, const int64_t* signal, size_t max_bytes) {
  if (world > kMaxRanks) return nullptr;
  ARState* s = new ARState();
  s->rank = rank;
  s->world = world;
  s->max_bytes = max_bytes;
  for (int r = 0; r < world; ++r) {
    s->peers.staging[r] = reinterpret_cast<uint16_t*>(staging[r]);
    s->peers.result[r] = reinterpret_cast<uint16_t*>(result[r]);
    s->peers.signal[r] = reinterpret_cast<Signal*>(signal[r]);
  }
  return s;
}
void ar_destroy(void* st) { delete reinterpret_cast<ARState*>(st); }

int ar_error(void* st) {      // host-side check of the timeout word (synchronises the stream by itself: call off the hot path)
  ARState* s = reinterpret_cast<ARState*>(st);
  uint32_t v = 0;
  (void)hipMemcpy(&v, &s->peers.signal[s->rank]->error, sizeof(v), hipMemcpyDeviceToHost);
  return (int)v;
}

int launch_all_reduce(void* st, const void* in, void* out, int64_t bytes, int algo, hipStream_t stream) {
  ARState* s = reinterpret_cast<ARState*>(st);
  if (bytes % 16 != 0 || (size_t)bytes > s->max_bytes) return -1;
  const int64_t n_vec = bytes / 16;
  int blocks = (int)std::min<int64_t>((n_vec + 511) / 512, 36);
  if (blocks < 1) blocks = 1;
  const bool two = algo == 2 || (algo == 0 && bytes > 256 * 1024 && s->world > 2);
  if (two)
    hipLaunchKernelGGL(all_reduce_kernel<true>, dim3(blocks), dim3(512), 0, stream, s->peers, s->rank, s->world,
                       (const uint16_t*)in, (uint16_t*)out, n_vec);
  else
    hipLaunchKernelGGL(all_reduce_kernel<false>, dim3(blocks), dim3(512), 0, stream, s->peers, s->rank, s->world,
                       (const uint16_t*)in, (uint16_t*)out, n_vec);
  return 0;
}

int launch_all_gather(void* st, const void* in, void* out, int64_t bytes, hipStream_t stream) {
  ARState* s = reinterpret_cast<ARState*>(st);
  if (bytes % 16 != 0 || (size_t)bytes > s->max_bytes) return -1;
  const int64_t n_vec = bytes / 16;
  int blocks = (int)std::min<int64_t>((n_vec + 511) / 512, 36);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(all_gather_kernel, dim3(blocks), dim3(512), 0, stream, s->peers, s->rank, s->world, (const uint16_t*)in,
                     (uint16_t*)out, n_vec);
  return 0;
}

}  // namespace dli
