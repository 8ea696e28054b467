// 256×256×64 bf16 GEMM tile for gfx950: the large-M path of the MoE grouped GEMM (K8) and of the dense gate/up
// projection with the SwiGLU epilogue (K3 folded into K11).   C[rows, n] = A[rows, :] · W[n, :]ᵀ
//
// Why this shape (cdna guide §5, MI355X_MICROARCH §LDS / §Two waves per SIMD):
//   * 8 waves as 2 (M) × 4 (N); a wave owns 128 × 64 of the output = 8 × 4 accumulators of v_mfma_f32_16x16x32_bf16
//     (the shape that holds the higher clock on random data).  Per K-step of 64 a wave reads 24 fragments
//     (ds_read_b128) for 64 MFMAs — half the LDS bytes per MFMA of the 64×64-per-wave tile in moe.hip, whose loop
//     was LDS-bound at 35 % MFMA-busy.
//   * both operands stream global → LDS by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write.  The
//     LDS image is lane-linear, so the bank swizzle sits on the SOURCE address (slot p of row r receives global
//     chunk p ^ (r & 7)) and on the read — never on the destination (guide rule 21).  With that swizzle the four
//     16-lane groups of a ds_read_b128 hit 16 distinct 16-byte slots: conflict-free.
//   * two K-step buffers of 64 KB (all LDS in ONE array, laid out A0 A1 B0 B1 so that every fragment read is a base
//     register + a 16-bit immediate).  A K-step is four quadrant phases of 16 MFMAs.
//   * PING-PONG between the two waves of a SIMD.  Every phase is a LOAD segment (the quadrant's ds_read_b128s + one
//     128-row half-tile of LDS-DMA = 2 instructions per wave) and an MFMA segment, each closed by a raw s_barrier.
//     Waves 4–7 enter the loop one barrier late, so on every SIMD one wave runs its 16 MFMAs while its partner
//     issues reads and DMA: the matrix pipe never waits for an address unit, and neither wave needs double-buffered
//     fragments (64 operand VGPRs + 128 accumulators).  (v1 of this kernel kept all 8 waves in lockstep with one
//     barrier per K-step: every wave issued its 8 DMAs behind the same barrier and the MFMA pipe idled for
//     ≈ 40 % of the step — profiles/gemm256_r2b_v1_one_barrier_per_kstep.jsonl.)
//   * DMA schedule: a half-tile is re-filled for the step two ahead right after BOTH wave groups have read it —
//     7 phases before its first read.  Every MFMA segment ends with s_waitcnt vmcnt(8): at most the four youngest half-tiles of a
//     wave are in flight, older ones have landed, and the two barriers that follow make that true for both groups
//     before anyone reads them (guide: "read a staged buffer one phase AFTER the wait that retires it", one barrier
//     more for staggered groups).  Never vmcnt(0) in the steady state; the last two K-steps drain.
//     Every LOAD segment ends with lgkmcnt(0) BEFORE its barrier, so "passed the barrier" ⇒ "my reads are done":
//     that is what makes the re-fill of a half-tile safe.
//   * the accumulator is D[weight column][token row] (operands swapped), so a lane holds 4 consecutive output
//     columns of one row: 8-byte row-contiguous stores, and the SwiGLU pair (gate column, up column) of a row
//     lives in the same lane.
//   * block → tile mapping: XCD x runs the x-th eighth of the tile order (8-column groups, row tile, column).
//   * HALF tiles (grouped launches: an expert's last tile with at most 128 routed rows).  The schedule
//     above stays as it is — same DMAs, same barriers, same counted waits — but the wave groups take 64 rows each (group 0
//     rows 0–63, group 1 rows 64–127: both are the FIRST quadrant pair of the K-step for their group) and the two a1
//     phases of a K-step keep their DMA issue, waits and barriers but read no A fragments and issue no MFMAs: a K-step is
//     four MFMA segments instead of eight.  Every output element is summed in the order of a full tile, so a row's bits
//     do not depend on the kind of tile it lands in.  (First version: both groups on all 128 rows, K split 32 / 32 inside
//     every phase and an LDS reduce at the end — 0.81 of a full tile, because a phase's reads, waits and barriers do not
//     halve with its MFMAs; profiles/ab_moe_half_tiles_r3.jsonl.)  The half-tile loop is a second copy of the K loop,
//     chosen per block from the alignment, so a full tile's instruction stream has no condition in it.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace dli {

namespace g256 {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int TILE_BYTES = TM * TK * 2;        // one operand tile of one K-step: 32 KB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + B
#ifndef DLI_G256_W_AUX
#define DLI_G256_W_AUX 0
#endif
constexpr int G256_W_AUX = DLI_G256_W_AUX;     // cache policy of the W-operand DMA; 2 = nt was measured: neutral for one row tile, 3 % SLOWER at M ≥ 2304 (the panel IS re-used by the row tiles of an XCD: profiles/ab_g256_wnt_r3.txt)

enum { EPI_SWIGLU = 0, EPI_SCATTER = 1, EPI_SCATTER_SPLITK = 2, EPI_PLAIN = 3, EPI_PLAIN_SPLITK = 4, EPI_SWIGLU_SPLITK = 5 };
// The two dense split-K epilogues (mid-M GEMMs: one or two row tiles cannot fill 256 CUs, so K is cut into slices) store
// raw fp32 partial sums into slab `split` of a workspace [k_splits][M][ldc]; splitk_reduce_kernel adds the slabs in
// fixed order and applies the real epilogue (bf16 store, or silu(gate)·up over the [gate | up] halves of a row).
constexpr bool is_swiglu(int epi) { return epi == EPI_SWIGLU || epi == EPI_SWIGLU_SPLITK; }

__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

// GROUPED: rows come from the MoE alignment (sorted_ids / tile_expert / meta); otherwise rows 0..M-1 of A, one "expert".
template <int EPI, bool GROUPED>
__global__ __launch_bounds__(512) void gemm256_kernel(
    const uint16_t* __restrict__ A, int64_t a_stride, const uint16_t* __restrict__ W, int64_t w_expert_stride,
    const int* __restrict__ sorted_ids, const int* __restrict__ tile_expert, const int* __restrict__ meta,
    const float* __restrict__ topk_w, int top_k, void* __restrict__ out_, int64_t out_stride, int M, int K, int N_half,
    int n_col_blocks, int k_splits, int64_t slab_elems, int lin_begin, int lin_count) {
  uint16_t* out = reinterpret_cast<uint16_t*>(out_);
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE_BYTES];

  // ---- block → (row tile, column block).  The hardware deals workgroups round-robin over the 8 XCDs (id % 8), each
  // with a private 4 MB L2.  Tiles are put in the linear order (group of 8 column blocks, row tile, column in group)
  // and XCD x runs the x-th eighth of that order: the ≈ 32 blocks resident on an XCD at a time are 4 row tiles × 8
  // column blocks — each operand K-slice is fetched into that L2 once and used 4–8 times — and the eighths differ by
  // at most one tile (whole super-tiles per XCD left XCDs 4–7 idle for the last 1/8 of a 28-super-tile launch).
  // RANGE mode (dense only, lin_count > 0): this launch covers the tiles lin_begin … lin_begin + lin_count − 1 of the
  // un-split tile order, each cut into k_splits K-slices (slowest index).  launch_dense_gemm256 uses it to run the whole
  // rounds of a GEMM (a multiple of the CU count, one 512-thread workgroup per CU) as plain tiles and the REMAINDER
  // tiles K-split over the otherwise idle CUs: 784 tiles are 3 rounds + 16 tiles × 8 slices, not 4 rounds.
  const bool ranged = !GROUPED && lin_count > 0;
  const int n_tiles = GROUPED ? meta[0] : (M + TM - 1) / TM;
  const int n_virtual = ranged ? n_col_blocks : n_col_blocks * k_splits;   // split-K slices are extra "columns" (slowest index)
  const int total = ranged ? lin_count * k_splits : n_tiles * n_virtual;
  const int per_xcd = (total + 7) >> 3;
  const int in_xcd = blockIdx.x >> 3;
  const int j_lin = (blockIdx.x & 7) * per_xcd + in_xcd;
  if (in_xcd >= per_xcd || j_lin >= total) return;
  int split = 0, lin = j_lin, lin_local = 0;
  if (ranged) {
    split = j_lin / lin_count;
    lin_local = j_lin - split * lin_count;
    lin = lin_begin + lin_local;
  }
  constexpr int GW = 8;
  const int grp = lin / (GW * n_tiles), within = lin - grp * (GW * n_tiles);
  const int gw = min(GW, n_virtual - grp * GW);              // the last group may be narrower
  const int tile = within / gw;
  const int col_virtual = grp * GW + (within - tile * gw);
  if (!ranged) split = col_virtual / n_col_blocks;
  const int col_block = ranged ? col_virtual : col_virtual - split * n_col_blocks;
  const int e = GROUPED ? tile_expert[tile] : 0;
  const uint16_t* We = W + (int64_t)e * w_expert_stride;
  const int n0 = col_block * (is_swiglu(EPI) ? TN / 2 : TN);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the LDS-DMA destinations below live in SGPRs
  const int wr = wave >> 2, wc = wave & 3;
  const int col = lane & 15, g = lane >> 4;
  // at most 128 valid rows: the alignment pads an expert's rows at the END of its last tile
  // (grouped launches carry the switch in lin_begin, which only the dense RANGE mode uses otherwise)
  const bool half = GROUPED && lin_begin != 0 && sorted_ids[tile * TM + TM / 2] < 0;
  const int a_row0 = half ? wr * 64 : wr * 128;              // first A row of this wave's group

  // ---- LDS-DMA sources.  A half-tile (a0, a1: 64 rows of each wave row; b0, b1: 32 rows of each wave column) is 128
  // rows = 16 pieces of 8 rows × 128 B = 1 KiB; wave w fills pieces 2w and 2w+1 of every half-tile.  Lane → row
  // lane/8 of the piece, slot lane%8, which receives global chunk slot ^ (row & 7) (swizzle on the source side).
  uint32_t a_src[2][2], b_src[2][2];                         // [half][piece j] byte offset of this lane's 16 bytes at k = 0
  int a_dst[2][2], b_dst[2][2];                              // wave-uniform LDS byte offset of the piece inside its tile
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int lr0 = (2 * wave + jj) * 8;                    // first logical row of the piece inside the half-tile
      const int ar0 = (lr0 >> 6) * 128 + h * 64 + (lr0 & 63), br0 = (lr0 >> 5) * 64 + h * 32 + (lr0 & 31);
      a_dst[h][jj] = ar0 * 128;
      b_dst[h][jj] = br0 * 128;
      const int ar = ar0 + (lane >> 3), br = br0 + (lane >> 3);
      int64_t arow;
      if (GROUPED && is_swiglu(EPI)) {
        const int sid = sorted_ids[tile * TM + ar];
        arow = sid >= 0 ? sid / top_k : 0;                   // padding rows read token 0 (discarded later)
      } else if (GROUPED) {
        arow = (int64_t)tile * TM + ar;
      } else {
        arow = min(tile * TM + ar, M - 1);                   // rows past M repeat the last row (never stored)
      }
      a_src[h][jj] = (uint32_t)((arow * a_stride + (((lane & 7) ^ (ar & 7)) * 8)) * 2);
      int wrow;
      if (is_swiglu(EPI)) {
        const int w_ = br >> 6, within = br & 63;             // wave column, position inside its 32 gate + 32 up rows
        wrow = (within < 32 ? 0 : N_half) + n0 + w_ * 32 + (within & 31);
      } else {
        wrow = n0 + br;
      }
      b_src[h][jj] = (uint32_t)(((int64_t)wrow * K + (((lane & 7) ^ (br & 7)) * 8)) * 2);
    }
  const char* Ab = reinterpret_cast<const char*>(A);
  const char* Wb = reinterpret_cast<const char*>(We);

  f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_total = K / TK;
  const int kt0 = (int)((int64_t)nk_total * split / k_splits);
  const int nk = (int)((int64_t)nk_total * (split + 1) / k_splits) - kt0;      // K-steps of this block

  // one half-tile (2 DMA instructions per wave) of K-step kt_ into buffer buf_
#define G256_STAGE_A(kt_, buf_, h_)                                                                                  \
  _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) __builtin_amdgcn_global_load_lds(                                  \
      (gptr_t)(Ab + a_src[h_][jj] + ((kt_) + kt0) * (TK * 2)), (lptr_t)(lds + (buf_) * TILE_BYTES + a_dst[h_][jj]), 16, 0, 0);
#define G256_STAGE_B(kt_, buf_, h_)                                                                                  \
  _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) __builtin_amdgcn_global_load_lds(                                  \
      (gptr_t)(Wb + b_src[h_][jj] + ((kt_) + kt0) * (TK * 2)), (lptr_t)(lds + STAGE_BYTES + (buf_) * TILE_BYTES + b_dst[h_][jj]), 16, 0, G256_W_AUX);
  // fragment reads: A rows wr·128 + m·16 + col, B rows wc·64 + n·16 + col; chunk ks·4 + g of the 64-wide K-step
  // (row offsets that are multiples of 8 rows leave the swizzle term alone: two base addresses per operand — one per
  //  32-wide k half — and compile-time immediates for everything else)
  const char* a_rd[2] = {lds + lds_off(a_row0 + col, g), lds + lds_off(a_row0 + col, 4 + g)};
  const char* b_rd[2] = {lds + STAGE_BYTES + lds_off(wc * 64 + col, g), lds + STAGE_BYTES + lds_off(wc * 64 + col, 4 + g)};
  // (hm_: the half-tile copy of the loop — no a1 fragments, no a1 MFMAs)
#define G256_RD_A(dst_, buf_, half_, hm_)                                                                            \
  if (!((hm_) && (half_) == 1))                                                                                      \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                    \
        dst_[m][ks] = *reinterpret_cast<const bf16x8*>(a_rd[ks] + (buf_) * TILE_BYTES + ((half_) * 64 + m * 16) * 128);
#define G256_RD_B(dst_, buf_, half_)                                                                                 \
  _Pragma("unroll") for (int n = 0; n < 2; ++n) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                      \
      dst_[n][ks] = *reinterpret_cast<const bf16x8*>(b_rd[ks] + (buf_) * TILE_BYTES + ((half_) * 32 + n * 16) * 128);
  // end of a LOAD segment: my reads are complete before I pass the barrier (→ the half-tile may be re-filled)
#define G256_END_LOAD()                                                                                              \
  {                                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
    __builtin_amdgcn_s_barrier();                                                                                    \
    asm volatile("" ::: "memory");                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }
  // MFMA segment: 16 MFMAs of one quadrant (k-step outermost: 8 independent accumulators between two uses of one),
  // then the counted DMA wait and the barrier
#define G256_MMA(fa_, fb_, mh_, nh_, drain_, hm_)                                                                    \
  {                                                                                                                  \
    if (!((hm_) && (mh_) == 1)) {                                                                                    \
      __builtin_amdgcn_s_setprio(1);                                                                                 \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int m = 0; m < 4; ++m)                  \
          _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                              \
              acc[(mh_) * 4 + m][(nh_) * 2 + n] = mfma16(fb_[n][ks], fa_[m][ks], acc[(mh_) * 4 + m][(nh_) * 2 + n]);  \
      __builtin_amdgcn_s_setprio(0);                                                                                 \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    if (drain_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                            \
    __builtin_amdgcn_s_barrier();                                                                                    \
    asm volatile("" ::: "memory");                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }
  // One K-step from buffer buf_ (= t & 1 = f_, the B half the step starts with).  Quadrants (a0,bf) (a0,bo) (a1,bo)
  // (a1,bf): the last one leaves the registers of bo free, and the NEXT step starts with bo — its first B fragments
  // are read in this step's last LOAD segment, which keeps the segments at 8 / 4 / 8 / 4 reads + 2 DMAs each.
  // DMA of step t+2 into this buffer, each half-tile right after both wave groups have read it: L0: bf (read one phase
  // before the step began), L1: a0, L2: bo, L3: a1 — 7 phases ahead of its first read.
  // (full_: a steady-state step, two more steps follow — no conditions at all in its instruction stream)
#define G256_STEP(t_, buf_, full_, hm_)                                                                                  \
  {                                                                                                                  \
    constexpr int f_ = (buf_), o_ = 1 - (buf_);                                                                      \
    const bool drain_ = !(full_) && (t_) + 2 >= nk;           /* nothing more will be issued: counted waits would lie */ \
    const bool more_ = (full_) || (t_) + 2 < nk;                                                                     \
    G256_RD_A(fa, buf_, 0, hm_);                                                                                          \
    if (more_) G256_STAGE_B((t_) + 2, buf_, f_);                                                                     \
    G256_END_LOAD();                                                                                                 \
    G256_MMA(fa, fb[f_], 0, f_, drain_, hm_);                                                                             \
    G256_RD_B(fb[o_], buf_, o_);                                                                                     \
    if (more_) G256_STAGE_A((t_) + 2, buf_, 0);                                                                      \
    G256_END_LOAD();                                                                                                 \
    G256_MMA(fa, fb[o_], 0, o_, drain_, hm_);                                                                             \
    G256_RD_A(fa, buf_, 1, hm_);                                                                                          \
    if (more_) G256_STAGE_B((t_) + 2, buf_, o_);                                                                     \
    G256_END_LOAD();                                                                                                 \
    G256_MMA(fa, fb[o_], 1, o_, drain_, hm_);                                                                             \
    if ((full_) || (t_) + 1 < nk) G256_RD_B(fb[o_], (buf_) ^ 1, o_);                                                 \
    if (more_) G256_STAGE_A((t_) + 2, buf_, 1);                                                                      \
    G256_END_LOAD();                                                                                                 \
    G256_MMA(fa, fb[f_], 1, f_, drain_, hm_);                                                                             \
  }

  bf16x8 fa[4][2], fb[2][2][2];                              // a0 or a1 (64 rows); fb[h]: b half h (32 weight rows)

  // prologue: steps 0 and 1 completely (the loop issues from step 2 on), then the first B fragments of step 0
  G256_STAGE_B(0, 0, 0);
  G256_STAGE_A(0, 0, 0);
  G256_STAGE_B(0, 0, 1);
  G256_STAGE_A(0, 0, 1);
  if (nk > 1) {
    G256_STAGE_B(1, 1, 1);
    G256_STAGE_A(1, 1, 0);
    G256_STAGE_B(1, 1, 0);
    G256_STAGE_A(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // step 0 landed, step 1 still in flight
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  G256_RD_B(fb[0], 0, 0);
  if (wr == 1) {                                               // the stagger: waves 4–7 run one segment behind
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  __builtin_amdgcn_sched_barrier(0);

  int t = 0;
  if (!half) {
    for (; t + 3 < nk; t += 2) {                               // steady state: both steps have two more behind them
      G256_STEP(t, 0, true, 0);
      G256_STEP(t + 1, 1, true, 0);
    }
    if (t < nk) G256_STEP(t, 0, false, 0);                     // the last (up to three) steps: guarded issue, draining waits
    if (t + 1 < nk) G256_STEP(t + 1, 1, false, 0);
    if (t + 2 < nk) G256_STEP(t + 2, 0, false, 0);
  } else {
    for (; t + 3 < nk; t += 2) {
      G256_STEP(t, 0, true, 1);
      G256_STEP(t + 1, 1, true, 1);
    }
    if (t < nk) G256_STEP(t, 0, false, 1);
    if (t + 1 < nk) G256_STEP(t + 1, 1, false, 1);
    if (t + 2 < nk) G256_STEP(t + 2, 0, false, 1);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();                   // pairs with the extra barrier of waves 4–7
#undef G256_STAGE_A
#undef G256_STAGE_B
#undef G256_RD_A
#undef G256_RD_B
#undef G256_END_LOAD
#undef G256_MMA
#undef G256_STEP

  // ---- epilogue.  acc[m][n][r]: row = wr·128 + m·16 + col, output column = wc·64 + n·16 + 4g + r
  // dense split-K slabs: whole-matrix slabs [split][M][ldc], or — RANGE mode — compact per-tile slabs
  // [split][tile of the range][256 rows][256 floats] (SwiGLU: gate columns 0…127, up columns 128…255)
  float* slab = reinterpret_cast<float*>(out_);
  int64_t slab_ld = out_stride;
  int slab_col0 = n0, slab_half = N_half;
  if (ranged && (EPI == EPI_PLAIN_SPLITK || EPI == EPI_SWIGLU_SPLITK)) {
    slab += ((int64_t)split * lin_count + lin_local) * (TM * TN);
    slab_ld = TN;
    slab_col0 = 0;
    slab_half = TN / 2;
  } else {
    slab += split * slab_elems;
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    if (half && m >= 4) break;                                 // a half tile has no a1 rows
    const int lrow = a_row0 + m * 16 + col;
    const int prow = tile * TM + lrow;
    const int sid = GROUPED ? sorted_ids[prow] : (prow < M ? prow : -1);
    const int64_t srow = (ranged && (EPI == EPI_PLAIN_SPLITK || EPI == EPI_SWIGLU_SPLITK)) ? lrow : prow;
    if (EPI == EPI_PLAIN_SPLITK) {
      if (sid >= 0) {
        float* dst = slab + srow * slab_ld + slab_col0 + wc * 64 + 4 * g;
#pragma unroll
        for (int n = 0; n < 4; ++n) *reinterpret_cast<f32x4*>(dst + n * 16) = acc[m][n];
      }
    } else if (EPI == EPI_SWIGLU_SPLITK) {
      if (sid >= 0) {                                            // raw partials: gate at column c, up at N_half + c of the slab row
        float* dst = slab + srow * slab_ld + slab_col0 + wc * 32 + 4 * g;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          *reinterpret_cast<f32x4*>(dst + n * 16) = acc[m][n];
          *reinterpret_cast<f32x4*>(dst + slab_half + n * 16) = acc[m][n + 2];
        }
      }
    } else if (EPI == EPI_SWIGLU) {
      // wave columns: n = 0,1 gate (32 columns), n = 2,3 the matching up columns
      if (GROUPED || sid >= 0) {
        uint16_t* dst = out + (int64_t)prow * out_stride + n0 + wc * 32 + 4 * g;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gate = acc[m][n][r], up = acc[m][n + 2][r];
            o[r] = (short)f2bf(gate * fast_rcp(1.f + fast_exp2(-1.4426950408889634f * gate)) * up);
          }
          *reinterpret_cast<bf16x4*>(dst + n * 16) = o;        // grouped padding rows hold finite junk, never read back
        }
      }
    } else if (EPI == EPI_SCATTER_SPLITK) {
      if (sid >= 0) {
        const float w = topk_w[sid];
        float* dst = reinterpret_cast<float*>(out_) + split * slab_elems + (int64_t)sid * out_stride + n0 + wc * 64 + 4 * g;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = acc[m][n][r] * w;
          *reinterpret_cast<f32x4*>(dst + n * 16) = o;
        }
      }
    } else if (sid >= 0) {
      const float w = EPI == EPI_SCATTER ? topk_w[sid] : 1.f;
      uint16_t* dst = out + (int64_t)sid * out_stride + n0 + wc * 64 + 4 * g;
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

}  // namespace g256

// epilogue: 0 SwiGLU, 1 scatter (MoE down), 2 scatter split-K (fp32 slabs).  Same contract as launch_grouped_gemm,
// 256-row tiles (moe_align must have padded the expert segments to 256).
void launch_grouped_gemm256(int epilogue, const void* A, int64_t a_stride, const void* W, int64_t w_expert_stride,
                            const int* sorted_ids, const int* tile_expert, const int* meta, const float* topk_w, int top_k,
                            void* out, int64_t out_stride, int K, int N_out, int max_tiles, int k_splits, int64_t slab_elems,
                            hipStream_t stream) {
  using namespace g256;
  if (max_tiles == 0) return;
  static const int half_tiles = [] {                                          // DLI_G256_HALF_TILES=0: every tile as a full tile (A/B)
    const char* v = std::getenv("DLI_G256_HALF_TILES");
    return (v && v[0] == '0') ? 0 : 1;
  }();
  const int n_col_blocks = N_out / (epilogue == EPI_SWIGLU ? TN / 2 : TN);
  const int n_virtual = n_col_blocks * k_splits;
  dim3 grid((max_tiles * n_virtual + 7) / 8 * 8);                            // surplus blocks exit on meta[0]
#define G256_GO(EPI)                                                                                                       \
  hipLaunchKernelGGL((gemm256_kernel<EPI, true>), grid, dim3(512), 0, stream, (const uint16_t*)A, a_stride, (const uint16_t*)W, \
                     w_expert_stride, sorted_ids, tile_expert, meta, topk_w, top_k, out, out_stride, 0, K, N_out, n_col_blocks, \
                     k_splits, slab_elems, half_tiles, 0)
  if (epilogue == EPI_SWIGLU) G256_GO(EPI_SWIGLU);
  else if (epilogue == EPI_SCATTER) G256_GO(EPI_SCATTER);
  else G256_GO(EPI_SCATTER_SPLITK);
#undef G256_GO
}

// out[m][c] = Σ_s slab_s[m][c]  (SWIGLU: silu(Σ gate) · Σ up, gate at c and up at n_out + c of a slab row of 2·n_out)
template <bool SWIGLU>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int64_t slab_elems,
                                                            int64_t ldc, int M, int n_out, uint16_t* __restrict__ out,
                                                            int64_t out_stride) {
  const int vec_per_row = n_out / 4;
  const int64_t total = (int64_t)M * vec_per_row;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / vec_per_row), c = (int)(i - (int64_t)m * vec_per_row) * 4;
    const float* p = slabs + (int64_t)m * ldc + c;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < n_slabs; ++s) {
      a += *reinterpret_cast<const f32x4*>(p + s * slab_elems);
      if (SWIGLU) b += *reinterpret_cast<const f32x4*>(p + s * slab_elems + n_out);
    }
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = SWIGLU ? a[r] * fast_rcp(1.f + fast_exp2(-1.4426950408889634f * a[r])) * b[r] : a[r];
      o[r] = (short)f2bf(v);
    }
    *reinterpret_cast<bf16x4*>(out + (int64_t)m * out_stride + c) = o;
  }
}

// out tile (tile, col_block) of the range = Σ_s compact slab_s, epilogue applied: the reduce pass of the RANGE-mode tail
// (one workgroup per quarter tile: 64 rows)
template <bool SWIGLU>
__global__ __launch_bounds__(256) void splitk_reduce_tiles_kernel(const float* __restrict__ slabs, int n_slabs, int lin_begin,
                                                                  int lin_count, int n_tiles, int n_col_blocks, int M,
                                                                  uint16_t* __restrict__ out, int64_t out_stride) {
  using namespace g256;
  const int lin_local = blockIdx.x >> 2, quarter = blockIdx.x & 3;
  const int lin = lin_begin + lin_local;
  constexpr int GW = 8;
  const int grp = lin / (GW * n_tiles), within = lin - grp * (GW * n_tiles);
  const int gw = min(GW, n_col_blocks - grp * GW);
  const int tile = within / gw;
  const int col_block = grp * GW + (within - tile * gw);
  constexpr int VEC_PER_ROW = (SWIGLU ? TN / 2 : TN) / 4;      // output vectors of 4 columns per tile row
  constexpr int ROWS_PER_PASS = 256 / VEC_PER_ROW;
  const int v = threadIdx.x % VEC_PER_ROW, r_in = threadIdx.x / VEC_PER_ROW;
  const float* base = slabs + (int64_t)lin_local * (TM * TN);
  const int64_t slab_stride = (int64_t)lin_count * (TM * TN);
  const int n0 = col_block * (SWIGLU ? TN / 2 : TN);
  for (int r = quarter * 64 + r_in; r < quarter * 64 + 64; r += ROWS_PER_PASS) {
    const int row = tile * TM + r;
    if (row >= M) break;
    const float* p = base + r * TN + v * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < n_slabs; ++s) {
      a += *reinterpret_cast<const f32x4*>(p + s * slab_stride);
      if (SWIGLU) b += *reinterpret_cast<const f32x4*>(p + s * slab_stride + TN / 2);
    }
    bf16x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float val = SWIGLU ? a[q] * fast_rcp(1.f + fast_exp2(-1.4426950408889634f * a[q])) * b[q] : a[q];
      o[q] = (short)f2bf(val);
    }
    *reinterpret_cast<bf16x4*>(out + (int64_t)row * out_stride + n0 + v * 4) = o;
  }
}

// How launch_dense_gemm256 would cut the REMAINDER tiles of a GEMM (tiles beyond the last whole round of one tile per
// CU): {remainder tiles, K-slices}; slices == 0: no cut (no remainder, no whole round before it, or more than half a
// round of remainder — then slices would only add a reduce pass).
void dense_gemm256_tail_plan(int M, int K, int N_out, bool swiglu, int* tail_tiles, int* tail_slices) {
  using namespace g256;
  constexpr int NUM_CU = 256;
  const int n_lin = ((M + TM - 1) / TM) * (N_out / (swiglu ? TN / 2 : TN));
  const int r = n_lin % NUM_CU;
  int s = 0;
  if (r > 0 && n_lin > NUM_CU && r <= NUM_CU / 2) s = std::min({8, NUM_CU / r, K / TK / 4});
  *tail_tiles = r;
  *tail_slices = s >= 2 ? s : 0;
}

// Dense y = x · Wᵀ (swiglu = false: W [N, K] → y [M, N]) or y = silu(x · W_gateᵀ) ⊙ (x · W_upᵀ)
// (swiglu = true: W = [W_gate; W_up] of 2·N_out rows → y [M, N_out]).  K % 64 == 0, N_out % 256 (128 with swiglu) == 0.
// k_splits > 1: K is cut into that many slices (each at least one K-step), partial sums go through `workspace`
// (fp32, k_splits · M · (swiglu ? 2 : 1) · N_out elements) and a reduce pass — the mid-M path: with one or two row tiles
// the column blocks alone leave most of the 256 CUs without a tile, and the GEMM is a weight stream that needs them all.
// split_tail (k_splits == 1): whole rounds as plain tiles, the remainder tiles K-split per dense_gemm256_tail_plan
// (`workspace`: tail slices · tail tiles · 65536 floats).
void launch_dense_gemm256(bool swiglu, const void* x, int64_t x_stride, const void* W, void* out, int64_t out_stride, int M,
                          int K, int N_out, int k_splits, float* workspace, bool split_tail, hipStream_t stream) {
  using namespace g256;
  if (M == 0) return;
  const int n_tiles = (M + TM - 1) / TM;
  const int n_col_blocks = N_out / (swiglu ? TN / 2 : TN);
#define G256_DENSE(EPI, grid_, out_ptr_, ld_, splits_, slab_, begin_, count_)                                                    \
  hipLaunchKernelGGL((gemm256_kernel<EPI, false>), dim3(grid_), dim3(512), 0, stream, (const uint16_t*)x, x_stride,              \
                     (const uint16_t*)W, (int64_t)0, nullptr, nullptr, nullptr, nullptr, 1, out_ptr_, ld_, M, K, N_out,          \
                     n_col_blocks, splits_, slab_, begin_, count_)
  if (k_splits > 1) {
    const int64_t ldc = (int64_t)N_out * (swiglu ? 2 : 1), slab = (int64_t)M * ldc;
    const int grid = (n_tiles * n_col_blocks * k_splits + 7) / 8 * 8;
    if (swiglu) G256_DENSE(EPI_SWIGLU_SPLITK, grid, workspace, ldc, k_splits, slab, 0, 0);
    else G256_DENSE(EPI_PLAIN_SPLITK, grid, workspace, ldc, k_splits, slab, 0, 0);
    const int blocks = (int)std::min<int64_t>(((int64_t)M * (N_out / 4) + 255) / 256, 2048);
    if (swiglu)
      hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, stream, workspace, k_splits, slab, ldc, M, N_out,
                         (uint16_t*)out, out_stride);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, stream, workspace, k_splits, slab, ldc, M, N_out,
                         (uint16_t*)out, out_stride);
    return;
  }
  int r = 0, s = 0;
  if (split_tail) dense_gemm256_tail_plan(M, K, N_out, swiglu, &r, &s);
  if (s >= 2) {
    const int n_full = n_tiles * n_col_blocks - r;
    const int grid_a = (n_full + 7) / 8 * 8, grid_b = (r * s + 7) / 8 * 8;
    if (swiglu) {
      G256_DENSE(EPI_SWIGLU, grid_a, out, out_stride, 1, (int64_t)0, 0, n_full);
      G256_DENSE(EPI_SWIGLU_SPLITK, grid_b, workspace, (int64_t)TN, s, (int64_t)0, n_full, r);
      hipLaunchKernelGGL(splitk_reduce_tiles_kernel<true>, dim3(r * 4), dim3(256), 0, stream, workspace, s, n_full, r, n_tiles,
                         n_col_blocks, M, (uint16_t*)out, out_stride);
    } else {
      G256_DENSE(EPI_PLAIN, grid_a, out, out_stride, 1, (int64_t)0, 0, n_full);
      G256_DENSE(EPI_PLAIN_SPLITK, grid_b, workspace, (int64_t)TN, s, (int64_t)0, n_full, r);
      hipLaunchKernelGGL(splitk_reduce_tiles_kernel<false>, dim3(r * 4), dim3(256), 0, stream, workspace, s, n_full, r, n_tiles,
                         n_col_blocks, M, (uint16_t*)out, out_stride);
    }
    return;
  }
  const int grid = (n_tiles * n_col_blocks + 7) / 8 * 8;
  if (swiglu) G256_DENSE(EPI_SWIGLU, grid, out, out_stride, 1, (int64_t)0, 0, 0);
  else G256_DENSE(EPI_PLAIN, grid, out, out_stride, 1, (int64_t)0, 0, 0);
#undef G256_DENSE
}

}  // namespace dli
