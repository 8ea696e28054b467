// W8A8 GEMM inputs: per-row (= per-token) dynamic quantisation of bf16 activations to OCP e4m3 (gfx950's native fp8), one
// pass, the scale chosen from the row's own maximum — no calibration, no cross-row reduction, hipGraph-friendly.
//   q[r][k] = e4m3(x[r][k] · 448 / amax_r),   scale[r] = amax_r / 448          (torch._scaled_mm: scale_a = scale)
// SILU: x is the [gate | up] output of the fused gate/up projection (2·K columns) and the quantised row is
// silu(gate) ⊙ up — the SwiGLU activation never exists in bf16 (the down projection reads fp8).
// One 256-thread workgroup per row; the row (after the activation) stays in registers between the max pass and the
// convert pass for K ≤ 16384, longer rows are re-read (from L2: the producer has just written them).
#include "common.h"
#include "kernels.h"

namespace dli {

namespace {

template <int NT>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) t = fmaxf(t, red[i]);
  __syncthreads();
  return t;
}

template <bool SILU>
__device__ __forceinline__ void load_vals(const uint16_t* __restrict__ xr, int v, int K, float (&f)[8]) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(xr + (int64_t)v * 8);
  if (SILU) {
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(xr + K + (int64_t)v * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float g = bf2f((uint16_t)a[j]);
      f[j] = g * fast_rcp(1.f + fast_exp2(-1.4426950408889634f * g)) * bf2f((uint16_t)b[j]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = bf2f((uint16_t)a[j]);
  }
}

__device__ __forceinline__ u32x2 pack8(const float (&f)[8], float inv) {
  float c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = fminf(fmaxf(f[j] * inv, -448.f), 448.f);   // the convert does not saturate (and NaN → −448)
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  return u32x2{(unsigned)lo, (unsigned)hi};
}

}  // namespace

template <bool SILU>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const uint16_t* __restrict__ x, int64_t x_stride,
                                                             uint8_t* __restrict__ q, int64_t q_stride,
                                                             float* __restrict__ scales, int K) {
  __shared__ float red[4];
  constexpr int MAXV = 8;                                    // vectors of 8 values a thread keeps: K ≤ 16384
  const int row = blockIdx.x;
  const uint16_t* xr = x + (int64_t)row * x_stride;
  uint8_t* qr = q + (int64_t)row * q_stride;
  const int nvec = K >> 3;
  float vals[MAXV][8];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      load_vals<SILU>(xr, v, K, vals[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(vals[i][j]));
    }
  }
  for (int v = threadIdx.x + MAXV * 256; v < nvec; v += 256) {   // the part of a long row that does not fit: max only
    float f[8];
    load_vals<SILU>(xr, v, K, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
  }
  amax = block_max<256>(amax, red);
  amax = fmaxf(amax, 1e-12f);                                // an all-zero row keeps a finite scale
  const float inv = 448.f / amax;
  if (threadIdx.x == 0) scales[row] = amax * (1.f / 448.f);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) *reinterpret_cast<u32x2*>(qr + (int64_t)v * 8) = pack8(vals[i], inv);
  }
  for (int v = threadIdx.x + MAXV * 256; v < nvec; v += 256) {
    float f[8];
    load_vals<SILU>(xr, v, K, f);
    *reinterpret_cast<u32x2*>(qr + (int64_t)v * 8) = pack8(f, inv);
  }
}

// x: [rows, K] bf16 (SILU: [rows, 2·K]) with row stride x_stride (elements); q: [rows, K] e4m3, row stride q_stride (bytes);
// scales: [rows] fp32.  K % 8 == 0.
void launch_quant_rows_fp8(bool silu, const void* x, int64_t x_stride, void* q, int64_t q_stride, float* scales, int rows,
                           int K, hipStream_t stream) {
  if (rows == 0) return;
  if (silu)
    hipLaunchKernelGGL(quant_rows_fp8_kernel<true>, dim3(rows), dim3(256), 0, stream, (const uint16_t*)x, x_stride, (uint8_t*)q,
                       q_stride, scales, K);
  else
    hipLaunchKernelGGL(quant_rows_fp8_kernel<false>, dim3(rows), dim3(256), 0, stream, (const uint16_t*)x, x_stride, (uint8_t*)q,
                       q_stride, scales, K);
}

}  // namespace dli
