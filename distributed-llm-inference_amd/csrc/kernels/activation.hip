// K3: SwiGLU gate, out[t, i] = silu(x[t, i]) * x[t, I + i]   (SURVEY.md §2.5).
// Pure bandwidth kernel: 16-byte vector loads/stores, grid-stride over (token, vector) pairs,
// capped at 2048 workgroups (8 per CU) so a large prefill does not pay launch-side overhead.
#include "common.h"
#include "kernels.h"

namespace dli {

__global__ __launch_bounds__(256) void silu_and_mul_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ x,
                                                           int64_t n_vec, int vec_per_row, int64_t x_stride) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / vec_per_row;
    const int col = (int)(i - row * vec_per_row) * 8;
    const uint16_t* g = x + row * x_stride + col;
    bf16x8 a = *reinterpret_cast<const bf16x8*>(g);
    bf16x8 b = *reinterpret_cast<const bf16x8*>(g + (int64_t)vec_per_row * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gate = bf2f((uint16_t)a[j]);
      const float up = bf2f((uint16_t)b[j]);
      // silu(g) = g / (1 + 2^(-g*log2(e)))
      const float s = gate * fast_rcp(1.f + fast_exp2(-1.4426950408889634f * gate));
      o[j] = (short)f2bf(s * up);
    }
    *reinterpret_cast<bf16x8*>(out + row * (int64_t)vec_per_row * 8 + col) = o;
  }
}

void launch_silu_and_mul(void* out, const void* x, int rows, int inter, int64_t x_stride, hipStream_t stream) {
  if (rows == 0) return;
  const int vec_per_row = inter / 8;
  const int64_t n_vec = (int64_t)rows * vec_per_row;
  int64_t blocks = (n_vec + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(silu_and_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (uint16_t*)out,
                     (const uint16_t*)x, n_vec, vec_per_row, x_stride);
}

}  // namespace dli
