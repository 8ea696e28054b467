// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dli {

constexpr int kWave = 64;
constexpr int kBlockTokens = 32;   // tokens per KV page (config.KV_BLOCK_SIZE)

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 4 VGPRs: one MFMA A/B fragment
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator fragment
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_mfma;

// D(16x16) += A(16x32) * B(32x16), bf16 in / fp32 accumulate.  Lane l holds A[row l&15][k 8(l>>4)+j],
// B[k 8(l>>4)+j][col l&15], D[row 4(l>>4)+r][col l&15].
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_mfma, a), __builtin_bit_cast(bf16x8_mfma, b),
                                                 c, 0, 0, 0);
}

typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator fragment

// D(32x32) += A(32x16) * B(16x32).  Lane l (c = l & 31, hi = l >> 5) holds A[row c][k 8hi+j], B[k 8hi+j][col c],
// D[row (r & 3) + 8 (r >> 2) + 4 hi][col c] for r = 0..15.
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_mfma, a), __builtin_bit_cast(bf16x8_mfma, b),
                                                 c, 0, 0, 0);
}

__device__ __forceinline__ float bf2f(uint16_t u) { return __uint_as_float(static_cast<uint32_t>(u) << 16); }

// round-to-nearest-even via the native conversion (v_cvt_pk_bf16_f32 on gfx950; keeps NaN a NaN)
__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 b = static_cast<__bf16>(f);
  return __builtin_bit_cast(uint16_t, b);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// OCP e4m3 (gfx950 native) <-> float
__device__ __forceinline__ float fp8_to_f(uint8_t v) {
  return __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(v), 0);
}
__device__ __forceinline__ uint8_t f_to_fp8(float f) {
  // Clamp explicitly (one v_med3_f32): without it a value past ±448 — an activation outlier beyond the calibrated
  // scale's head-room — converts to the e4m3 NaN pattern, and one NaN key poisons every later softmax of its sequence.
  // fmaxf / fminf drop a NaN input as well (→ −448).
  f = fminf(fmaxf(f, -448.f), 448.f);
  int packed = __builtin_amdgcn_cvt_pk_fp8_f32(f, f, 0, false);
  return static_cast<uint8_t>(packed & 0xff);
}

// 4 packed e4m3 bytes → 4 bf16 (elements base..base+3 of a fragment; exact: e4m3 fits bf16)
__device__ __forceinline__ void fp8x4_to_bf16(uint32_t w, bf16x8& f, int base) {
  const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false);
  const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
  f[base + 0] = (short)f2bf(lo[0]);
  f[base + 1] = (short)f2bf(lo[1]);
  f[base + 2] = (short)f2bf(hi[0]);
  f[base + 3] = (short)f2bf(hi[1]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x = NT (multiple of 64).  `red` is NT/64 floats of LDS.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  __syncthreads();
  return t;
}

}  // namespace dli
