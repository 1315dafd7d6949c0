// AWQ int4 weight dequantisation (bit-exact with the reference).
//
// Replaces awq_dequantize (sgl-kernel/csrc/gemm/awq_kernel.cu:127-221; python gemm.py:7-10) and the Triton
// awq_dequantize_triton the reference uses on HIP (python/sglang/srt/layers/quantization/awq_triton.py:14-108):
//   W[k, 8c + j] = (nib(qweight[k, c], order[j]) - nib(qzeros[k / G, c], order[j])) * scales[k / G, 8c + j]
//   order = [0, 4, 1, 5, 2, 6, 3, 7]  (nibble at bits 4*order[j])
// The difference (w - z) is an exact small integer and the product is rounded once to the scale dtype, which is
// what the reference's half2/bf162 sub+mul produces as well.
// HBM-bound byte work: one int32 (8 weights) per thread -> 4-byte coalesced reads, 16-byte coalesced writes; the
// AWQ nibble order means (q >> 4i) & 0x000F000F is already the pair of adjacent columns (2i, 2i+1).
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void awq_dequant_kernel(const uint32_t* __restrict__ qweight, const T* __restrict__ scales,
                                                          const uint32_t* __restrict__ qzeros, T* __restrict__ out, int K,
                                                          int NC, int G) {
  const int64_t total = (int64_t)K * NC;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int k = (int)(idx / NC), c = (int)(idx - (int64_t)k * NC);
    const int grp = k / G;
    const uint32_t q = qweight[idx];
    const uint32_t z = qzeros[(int64_t)grp * NC + c];
    const u32x4_t sraw = *(const u32x4_t*)(scales + ((int64_t)grp * NC + c) * 8);
    struct S8 { T v[8]; };
    const S8 s = __builtin_bit_cast(S8, sraw);
    S8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t qp = (q >> (4 * i)) & 0x000F000Fu, zp = (z >> (4 * i)) & 0x000F000Fu;
      const int d0 = (int)(qp & 0xF) - (int)(zp & 0xF), d1 = (int)(qp >> 16) - (int)(zp >> 16);
      o.v[2 * i] = (T)((float)d0 * (float)s.v[2 * i]);
      o.v[2 * i + 1] = (T)((float)d1 * (float)s.v[2 * i + 1]);
    }
    *(u32x4_t*)(out + idx * 8) = __builtin_bit_cast(u32x4_t, o);
  }
}

}  // namespace

// qweight [K, N/8] int32, scales [K/G, N] (bf16/f16), qzeros [K/G, N/8] int32 -> out [K, N] in the scale dtype.
extern "C" int sgl_mi355_awq_dequantize(const void* qweight, const void* scales, const void* qzeros, void* out, int K,
                                        int num_packed_cols, int group_size, int scale_dtype, void* stream) {
  SGL_CHECK(K >= 0 && num_packed_cols >= 0, "awq_dequantize: negative shape");
  if (K == 0 || num_packed_cols == 0) return SGL_MI355_OK;
  SGL_CHECK(qweight && scales && qzeros && out, "awq_dequantize: null pointer");
  SGL_CHECK(group_size > 0 && K % group_size == 0, "awq_dequantize: K=%d not a multiple of group_size=%d", K, group_size);
  SGL_CHECK(scale_dtype == SGL_BF16 || scale_dtype == SGL_F16, "awq_dequantize: scales must be f16 or bf16");
  SGL_CHECK(((uintptr_t)scales % 16) == 0 && ((uintptr_t)out % 16) == 0, "awq_dequantize: scales/out must be 16-byte aligned");
  const int64_t total = (int64_t)K * num_packed_cols;
  const int64_t b = (total + 255) / 256;
  const unsigned blocks = (unsigned)(b > 8192 ? 8192 : b);
  hipStream_t st = (hipStream_t)stream;
  if (scale_dtype == SGL_F16)
    hipLaunchKernelGGL((awq_dequant_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, (const uint32_t*)qweight,
                       (const _Float16*)scales, (const uint32_t*)qzeros, (_Float16*)out, K, num_packed_cols, group_size);
  else
    hipLaunchKernelGGL((awq_dequant_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, (const uint32_t*)qweight,
                       (const __bf16*)scales, (const uint32_t*)qzeros, (__bf16*)out, K, num_packed_cols, group_size);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}
