// K6: temperature / top-k / top-p / multinomial sampling, sort-free.
//
// One 1024-thread workgroup per batch row, logits read with 16-byte loads from L2/HBM:
//   * temperature == 0          → arg-max (lowest index on ties), one pass.
//   * top_k == 0 && top_p >= 1  → Gumbel-max draw  argmax_i(x_i / T + g_i), one pass (shift invariant,
//                                  so neither the row max nor the softmax sum is needed).
//   * otherwise                 → threshold search instead of a sort: the fp32 logit is mapped to an
//                                  order-preserving 32-bit key and the cut-off key is found by a 4-pass
//                                  radix descent (8 bits per pass, 256-bin LDS histogram of *counts*
//                                  for top-k, of *probability mass* for top-p); then the Gumbel-max
//                                  draw runs over the surviving keys only.
// Gumbel noise comes from Philox4x32-10 keyed by the request seed, counter = (element/4, step), so a
// request's draw is independent of the batch it happens to share; a `noise` tensor can be supplied
// instead (tests: bit-exact comparison against the torch reference with the same uniforms).
// Ties at the cut-off key are all kept (the torch reference keeps as many as needed) — only
// observable when several tokens share the exact same logit at the nucleus boundary.
#include "common.h"
#include "kernels.h"

namespace dli {

constexpr int SNT = 1024;

__device__ __forceinline__ uint32_t okey(float f) {   // monotone float → uint
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ uint32_t mulhilo(uint32_t a, uint32_t b, uint32_t& hi) {
  const uint64_t p = (uint64_t)a * b;
  hi = (uint32_t)(p >> 32);
  return (uint32_t)p;
}
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint32_t hi0, hi1;
    const uint32_t lo0 = mulhilo(0xD2511F53u, ctr.x, hi0);
    const uint32_t lo1 = mulhilo(0xCD9E8D57u, ctr.z, hi1);
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += 0x9E3779B9u;
    key.y += 0xBB67AE85u;
  }
  return ctr;
}
__device__ __forceinline__ float u01(uint32_t r) { return ((r >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1)
__device__ __forceinline__ float gumbel(float u) { return -__logf(-__logf(u)); }

struct ArgMax {
  float v;
  int i;
};

// A row of NaN logits never beats the initial candidate and leaves its out-of-range index behind: the next step would
// use it as an embedding row (a GPU fault).  Such a row yields token 0 instead; the NaN itself is the caller's bug.
__device__ __forceinline__ int valid_token(int i, int V) { return (unsigned)i < (unsigned)V ? i : 0; }
__device__ __forceinline__ ArgMax amax(ArgMax a, ArgMax b) {
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ ArgMax block_argmax(ArgMax a, float* sv, int* si) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax b{__shfl_xor(a.v, o, 64), __shfl_xor(a.i, o, 64)};
    a = amax(a, b);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sv[w] = a.v; si[w] = a.i; }
  __syncthreads();
  ArgMax r{sv[0], si[0]};
  for (int k = 1; k < SNT / 64; ++k) r = amax(r, ArgMax{sv[k], si[k]});
  __syncthreads();
  return r;
}
__device__ float block_max(float v, float* sv) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sv[w] = v;
  __syncthreads();
  float r = sv[0];
  for (int k = 1; k < SNT / 64; ++k) r = fmaxf(r, sv[k]);
  __syncthreads();
  return r;
}

template <typename T>
__device__ __forceinline__ float ld(const T* p, int i);
template <>
__device__ __forceinline__ float ld<uint16_t>(const uint16_t* p, int i) { return bf2f(p[i]); }
template <>
__device__ __forceinline__ float ld<float>(const float* p, int i) { return p[i]; }

// Radix descent.  MASS = false: find the key of the k-th largest element (target = k, weights 1).
// MASS = true : find the largest key K with  mass(key >= K) >= target   (weights exp2(z - zmax)).
// Only elements with key >= floor_key take part.  Returns the cut-off key.
template <typename T, bool MASS>
__device__ uint32_t radix_cut(const T* row, int V, float inv_t_log2, float zmax, float target, uint32_t floor_key,
                              float* hist, uint32_t* s_pick, float* s_rem) {
  uint32_t prefix = 0, prefix_mask = 0;
  float remaining = target;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = threadIdx.x; i < 256; i += SNT) hist[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < V; i += SNT) {
      const float z = ld<T>(row, i) * inv_t_log2;
      const uint32_t k = okey(z);
      if (k >= floor_key && (k & prefix_mask) == prefix)
        atomicAdd(&hist[(k >> shift) & 255], MASS ? fast_exp2(z - zmax) : 1.f);
    }
    __syncthreads();
    if (threadIdx.x == 0) {   // walk the bins from the top until the running total reaches the target
      float acc = 0.f;
      int bin = 255;
      for (; bin > 0; --bin) {
        if (acc + hist[bin] >= remaining) break;
        acc += hist[bin];
      }
      *s_pick = (uint32_t)bin;
      *s_rem = remaining - acc;
    }
    __syncthreads();
    prefix |= (*s_pick) << shift;
    prefix_mask |= 255u << shift;
    remaining = *s_rem;
    __syncthreads();
  }
  return prefix;
}

template <typename T>
__global__ __launch_bounds__(SNT) void sample_kernel(const T* __restrict__ logits, int64_t row_stride, int V,
                                                     const float* __restrict__ temperature, const float* __restrict__ top_p,
                                                     const int* __restrict__ top_k, const int64_t* __restrict__ seeds,
                                                     const int64_t* __restrict__ steps, const float* __restrict__ noise,
                                                     int64_t* __restrict__ out) {
  __shared__ float sv[SNT / 64];
  __shared__ int si[SNT / 64];
  __shared__ float hist[256];
  __shared__ uint32_t s_pick;
  __shared__ float s_rem;

  const int b = blockIdx.x;
  const T* row = logits + (int64_t)b * row_stride;
  const float temp = temperature[b];

  if (temp <= 0.f) {
    ArgMax a{-3.0e38f, 0x7fffffff};
    for (int i = threadIdx.x; i < V; i += SNT) a = amax(a, ArgMax{ld<T>(row, i), i});
    a = block_argmax(a, sv, si);
    if (threadIdx.x == 0) out[b] = valid_token(a.i, V);
    return;
  }

  const float inv_t_log2 = 1.4426950408889634f / temp;    // z = x / T in log2 units
  const float p = top_p[b];
  int k = top_k[b];
  if (k <= 0 || k > V) k = V;
  uint32_t floor_key = 0;

  if (k < V || p < 1.f) {
    float zmax = -3.0e38f;
    for (int i = threadIdx.x; i < V; i += SNT) zmax = fmaxf(zmax, ld<T>(row, i) * inv_t_log2);
    zmax = block_max(zmax, sv);
    if (k < V) floor_key = radix_cut<T, false>(row, V, inv_t_log2, zmax, (float)k, 0u, hist, &s_pick, &s_rem);
    if (p < 1.f) {
      float tot = 0.f;                                     // mass of what survived top-k
      for (int i = threadIdx.x; i < V; i += SNT) {
        const float z = ld<T>(row, i) * inv_t_log2;
        if (okey(z) >= floor_key) tot += fast_exp2(z - zmax);
      }
      tot = wave_sum(tot);
      if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = tot;
      __syncthreads();
      tot = 0.f;
      for (int w = 0; w < SNT / 64; ++w) tot += sv[w];
      __syncthreads();
      floor_key = radix_cut<T, true>(row, V, inv_t_log2, zmax, p * tot, floor_key, hist, &s_pick, &s_rem);
    }
  }

  const uint64_t seed = (uint64_t)seeds[b];
  const uint64_t step = (uint64_t)steps[b];
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  ArgMax a{-3.0e38f, 0x7fffffff};
  const float ln2 = 0.6931471805599453f;
  for (int i4 = threadIdx.x; i4 * 4 < V; i4 += SNT) {
    uint4 r = make_uint4(0, 0, 0, 0);
    if (noise == nullptr) r = philox4x32_10(make_uint4((uint32_t)i4, (uint32_t)step, (uint32_t)(step >> 32), 0x5A17u), key);
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i4 * 4 + e;
      if (i < V) {
        const float z = ld<T>(row, i) * inv_t_log2;
        if (okey(z) >= floor_key) {
          const float u = noise != nullptr ? noise[(int64_t)b * V + i] : u01(rr[e]);
          a = amax(a, ArgMax{z * ln2 + gumbel(u), i});
        }
      }
    }
  }
  a = block_argmax(a, sv, si);
  if (threadIdx.x == 0) out[b] = valid_token(a.i, V);
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 fast path (the serving path: logits come out of the LM-head GEMM in bf16, V ≤ 131072, V % 8 == 0).
// The whole row lives in registers — thread t holds 16-byte chunks t, t+1024, … (8 logits each, packed) — so
// every later pass (max, mass, radix, draw) is register work and the row is read from memory exactly once.
// A bf16 logit has only 16 key bits, so the radix descent takes TWO 8-bit passes (not four), and a thread adds
// runs of equal bins to the LDS histogram with one atomic (softmax-shaped rows put most of a thread's elements
// in the same two or three bins; per-element atomics on them serialise the whole workgroup).
// The Philox counter of element i is i/4 exactly as in the generic kernel: both paths draw the same token.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t okey16(uint32_t h) { return (h & 0x8000u) ? (~h & 0xFFFFu) : (h | 0x8000u); }
__device__ __forceinline__ uint32_t half_of(const uint4& c, int e) {   // e-th bf16 (0..7) of a chunk, compile-time e
  const uint32_t w = e < 2 ? c.x : e < 4 ? c.y : e < 6 ? c.z : c.w;
  return (e & 1) ? (w >> 16) : (w & 0xFFFFu);
}

// Every pass re-derives keys and weights from the packed row.  Without this fence the compiler keeps the 128 unpacked
// keys of a thread alive from one pass to the next (CSE) and spills; the empty asm makes the registers opaque.
template <int NCH>
__device__ __forceinline__ void opaque(uint4 (&regs)[NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(regs[c].x), "+v"(regs[c].y), "+v"(regs[c].z), "+v"(regs[c].w));
}

// descending walk over 256 bins by one wave: first bin where the running total reaches `remaining`
__device__ __forceinline__ void pick_bin(const float* hist, float remaining, uint32_t* s_pick, float* s_rem) {
  const int lane = threadIdx.x;                       // called by threads 0..63
  float h[4], local = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { h[j] = hist[255 - 4 * lane - j]; local += h[j]; }
  float incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  const uint64_t hit = __ballot(incl >= remaining);
  const int first = hit ? __ffsll((long long)hit) - 1 : 63;
  if (lane == first) {
    float acc = incl - local;
    int j = 0;
    for (; j < 3; ++j) {
      if (acc + h[j] >= remaining) break;
      acc += h[j];
    }
    if (!hit) { j = 3; acc = incl - h[3]; }           // total < remaining (rounding): lowest bin, like the serial walk
    *s_pick = (uint32_t)(255 - 4 * lane - j);
    *s_rem = remaining - acc;
  }
}

template <int NCH, bool MASS>
__device__ __forceinline__ uint32_t radix_cut16(uint4 (&regs)[NCH], float inv_t_log2, float zmax, float target, uint32_t floor_key,
                                float* hist, uint32_t* s_pick, float* s_rem) {
  uint32_t prefix = 0;
  float remaining = target;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int shift = pass == 0 ? 8 : 0;
    for (int i = threadIdx.x; i < 256; i += SNT) hist[i] = 0.f;
    __syncthreads();
    opaque(regs);
    int cur_bin = -1;
    float cur = 0.f;
#pragma clang loop unroll(full)
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t h = half_of(regs[c], e);
        const uint32_t k = okey16(h);
        const bool take = k >= floor_key && (pass == 0 || (k >> 8) == (prefix >> 8));
        if (take) {
          const float w = MASS ? fast_exp2(bf2f((uint16_t)h) * inv_t_log2 - zmax) : 1.f;
          const int bin = (int)((k >> shift) & 255u);
          if (bin == cur_bin) {
            cur += w;
          } else {
            if (cur_bin >= 0) atomicAdd(&hist[cur_bin], cur);
            cur_bin = bin;
            cur = w;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);      // one chunk at a time: keeps the 64 row registers the only long-lived ones
    }
    if (cur_bin >= 0) atomicAdd(&hist[cur_bin], cur);
    __syncthreads();
    if (threadIdx.x < 64) pick_bin(hist, remaining, s_pick, s_rem);
    __syncthreads();
    prefix |= (*s_pick) << shift;
    remaining = *s_rem;
    __syncthreads();
  }
  return prefix;
}

template <int NCH>
__global__ __launch_bounds__(SNT) void sample_bf16_kernel(const uint16_t* __restrict__ logits, int64_t row_stride, int V,
                                                          const float* __restrict__ temperature, const float* __restrict__ top_p,
                                                          const int* __restrict__ top_k, const int64_t* __restrict__ seeds,
                                                          const int64_t* __restrict__ steps, const float* __restrict__ noise,
                                                          int64_t* __restrict__ out) {
  __shared__ float sv[SNT / 64];
  __shared__ int si[SNT / 64];
  __shared__ float hist[256];
  __shared__ uint32_t s_pick;
  __shared__ float s_rem;

  const int b = blockIdx.x;
  const uint4* row = reinterpret_cast<const uint4*>(logits + (int64_t)b * row_stride);
  const int nchunks = V >> 3;
  uint4 regs[NCH];
#pragma clang loop unroll(full)
  for (int c = 0; c < NCH; ++c) {
    const int idx = c * SNT + threadIdx.x;
    regs[c] = idx < nchunks ? row[idx] : make_uint4(0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u);   // -inf padding
  }
  const float temp = temperature[b];

  if (temp <= 0.f) {
    ArgMax a{-3.0e38f, 0x7fffffff};
#pragma clang loop unroll(full)
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        a = amax(a, ArgMax{bf2f((uint16_t)half_of(regs[c], e)), (c * SNT + (int)threadIdx.x) * 8 + e});
    a = block_argmax(a, sv, si);
    if (threadIdx.x == 0) out[b] = valid_token(a.i, V);
    return;
  }

  const float inv_t_log2 = 1.4426950408889634f / temp;    // z = x / T in log2 units
  const float p = top_p[b];
  int k = top_k[b];
  if (k <= 0 || k > V) k = V;
  uint32_t floor_key = 0;                                  // 16-bit key of the raw bf16 logit (x → x/T is monotone)

  if (k < V || p < 1.f) {
    float zmax = -3.0e38f;
    opaque(regs);
#pragma clang loop unroll(full)
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) zmax = fmaxf(zmax, bf2f((uint16_t)half_of(regs[c], e)));
    zmax = block_max(zmax, sv) * inv_t_log2;
    if (k < V) floor_key = radix_cut16<NCH, false>(regs, inv_t_log2, zmax, (float)k, 0u, hist, &s_pick, &s_rem);
    if (p < 1.f) {
      float tot = 0.f;                                     // mass of what survived top-k
      opaque(regs);
#pragma clang loop unroll(full)
      for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t h = half_of(regs[c], e);
          if (okey16(h) >= floor_key) tot += fast_exp2(bf2f((uint16_t)h) * inv_t_log2 - zmax);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      tot = wave_sum(tot);
      if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = tot;
      __syncthreads();
      tot = 0.f;
      for (int w = 0; w < SNT / 64; ++w) tot += sv[w];
      __syncthreads();
      floor_key = radix_cut16<NCH, true>(regs, inv_t_log2, zmax, p * tot, floor_key, hist, &s_pick, &s_rem);
    }
  }

  const uint64_t seed = (uint64_t)seeds[b];
  const uint64_t step = (uint64_t)steps[b];
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  ArgMax a{-3.0e38f, 0x7fffffff};
  const float ln2 = 0.6931471805599453f;
  opaque(regs);
#pragma clang loop unroll(full)
  for (int c = 0; c < NCH; ++c) {
    const int chunk = c * SNT + (int)threadIdx.x;
    if (chunk < nchunks) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int i4 = chunk * 2 + half;
        uint4 r = make_uint4(0, 0, 0, 0);
        if (noise == nullptr) r = philox4x32_10(make_uint4((uint32_t)i4, (uint32_t)step, (uint32_t)(step >> 32), 0x5A17u), key);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t h = half_of(regs[c], half * 4 + e);
          if (okey16(h) >= floor_key) {
            const int i = i4 * 4 + e;
            const float u = noise != nullptr ? noise[(int64_t)b * V + i] : u01(rr[e]);
            a = amax(a, ArgMax{bf2f((uint16_t)h) * inv_t_log2 * ln2 + gumbel(u), i});
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  a = block_argmax(a, sv, si);
  if (threadIdx.x == 0) out[b] = valid_token(a.i, V);
}

void launch_sample(const void* logits, bool logits_fp32, int64_t row_stride, int B, int V, const float* temperature,
                   const float* top_p, const int* top_k, const int64_t* seeds, const int64_t* steps, const float* noise,
                   int64_t* out, hipStream_t stream) {
  if (B == 0) return;
  const bool rows16 = V % 8 == 0 && row_stride % 8 == 0 && reinterpret_cast<uintptr_t>(logits) % 16 == 0;
  if (logits_fp32)
    hipLaunchKernelGGL(sample_kernel<float>, dim3(B), dim3(SNT), 0, stream, (const float*)logits, row_stride, V,
                       temperature, top_p, top_k, seeds, steps, noise, out);
  else if (rows16 && V <= 4 * SNT * 8)
    hipLaunchKernelGGL(sample_bf16_kernel<4>, dim3(B), dim3(SNT), 0, stream, (const uint16_t*)logits, row_stride, V,
                       temperature, top_p, top_k, seeds, steps, noise, out);
  else if (rows16 && V <= 16 * SNT * 8)
    hipLaunchKernelGGL(sample_bf16_kernel<16>, dim3(B), dim3(SNT), 0, stream, (const uint16_t*)logits, row_stride, V,
                       temperature, top_p, top_k, seeds, steps, noise, out);
  else
    hipLaunchKernelGGL(sample_kernel<uint16_t>, dim3(B), dim3(SNT), 0, stream, (const uint16_t*)logits, row_stride, V,
                       temperature, top_p, top_k, seeds, steps, noise, out);
}

// Pipelined stepping: a decode row whose input id was sampled by the step still in flight reads it from the
// sampler's device output (``src`` = row there, -1 = the host already staged the id).
__global__ void gather_sampled_kernel(int* __restrict__ input, const int* __restrict__ src,
                                      const int64_t* __restrict__ sampled, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int s = src[i];
    if (s >= 0) input[i] = (int)sampled[s];
  }
}

void launch_gather_sampled(int* input, const int* src, const int64_t* sampled, int n, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_sampled_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, input, src, sampled, n);
}

}  // namespace dli
