// Dynamic fp8 (OCP e4m3fn -- gfx950 is NOT fnuz) activation quantisation.
//
// Replaces (same arithmetic, so results are bit-identical to the reference's torch restatements):
//   sgl_per_token_quant_fp8   sgl-kernel/csrc/gemm/per_token_quant_fp8.cu:15-228
//       scale = absmax / 448 ; scale_inv = scale == 0 ? 0 : 1 / scale ; q = sat(x * scale_inv)
//   sgl_per_tensor_quant_fp8  sgl-kernel/csrc/gemm/per_tensor_quant_fp8.cu:10-123
//       (dynamic) scale = atomicMax(absmax / 448) ; q = sat(x * (1 / scale))
//   sgl_per_token_group_quant_fp8  sgl-kernel/csrc/gemm/per_token_group_quant_8bit.cu
//       y_s = max(absmax, eps) / fp8_max ; q = clamp(x / y_s, fp8_min, fp8_max)
// Divisions are IEEE (this file is built without fast-math); conversion is the hardware
// v_cvt_pk_fp8_f32 (RNE) after an explicit clamp to +-448.
//
// HBM-bound byte work: 16-byte loads, one wave (or one workgroup for small batches) per token,
// wave-shuffle absmax, second pass served from L2.
#include "common.h"

namespace {

constexpr float kFp8Max = 448.0f;

template <typename T>
struct Vec8 {
  T v[8];
};

template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&f)[8]) {
  const u32x4_t raw = *(const u32x4_t*)p;
  const Vec8<T> x = __builtin_bit_cast(Vec8<T>, raw);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (float)x.v[j];
}

__device__ __forceinline__ float clamp448(float v, float lo = -kFp8Max, float hi = kFp8Max) {
  return fmaxf(fminf(v, hi), lo);
}

// 8 floats -> 8 fp8 bytes (two dwords)
__device__ __forceinline__ u32x2_t pack8_fp8(const float (&f)[8]) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  return u32x2_t{(uint32_t)lo, (uint32_t)hi};
}

__device__ __forceinline__ uint8_t cvt1_fp8(float f) {
  return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(f, 0.f, 0, false) & 0xff);
}

// One workgroup of NT threads per token when TOK_PER_WG == 1, else one wave per token.
template <typename T, int NT, int TOK_PER_WG>
__global__ __launch_bounds__(NT) void per_token_quant_kernel(const T* __restrict__ in, uint8_t* __restrict__ out_q,
                                                             float* __restrict__ out_s, int64_t hidden, int64_t tokens,
                                                             int64_t in_stride) {
  __shared__ float red[NT / 64];
  constexpr int TPT = NT / TOK_PER_WG;  // threads per token
  const int tid = threadIdx.x;
  const int sub = tid / TPT, t = tid % TPT;
  const int64_t token = (int64_t)blockIdx.x * TOK_PER_WG + sub;
  const bool live = token < tokens;
  const T* row = in + (live ? token : 0) * in_stride;
  const int64_t nvec = hidden / 8;
  float amax = 0.f;
  if (live) {
    for (int64_t i = t; i < nvec; i += TPT) {
      float f[8];
      load8(row + i * 8, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
    }
  }
  amax = wave_reduce_max(amax);
  if constexpr (TPT > 64) {
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = red[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) amax = fmaxf(amax, red[i]);
  }
  if (!live) return;
  const float scale = amax / kFp8Max;
  if (t == 0) out_s[token] = scale;
  const float inv = (scale == 0.f) ? 0.f : 1.0f / scale;
  uint8_t* orow = out_q + token * hidden;
  for (int64_t i = t; i < nvec; i += TPT) {
    float f[8];
    load8(row + i * 8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = clamp448(f[j] * inv);
    *(u32x2_t*)(orow + i * 8) = pack8_fp8(f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void per_tensor_absmax_kernel(const T* __restrict__ in, float* __restrict__ out_s, int64_t n) {
  __shared__ float red[4];
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nvec = n / 8;
  float amax = 0.f;
  for (int64_t i = gid; i < nvec; i += stride) {
    float f[8];
    load8(in + i * 8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
  }
  for (int64_t i = nvec * 8 + gid; i < n; i += stride) amax = fmaxf(amax, fabsf((float)in[i]));
  amax = wave_reduce_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // non-negative floats order like their bit patterns
    atomicMax((unsigned int*)out_s, __float_as_uint(amax / kFp8Max));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void per_tensor_quant_kernel(const T* __restrict__ in, uint8_t* __restrict__ out_q,
                                                               const float* __restrict__ scale, int64_t n) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float inv = 1.0f / (*scale);
  const int64_t nvec = n / 8;
  for (int64_t i = gid; i < nvec; i += stride) {
    float f[8];
    load8(in + i * 8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = clamp448(f[j] * inv);
    *(u32x2_t*)(out_q + i * 8) = pack8_fp8(f);
  }
  for (int64_t i = nvec * 8 + gid; i < n; i += stride) out_q[i] = cvt1_fp8(clamp448((float)in[i] * inv));
}

// input_to_float8 (python/sglang/srt/layers/quantization/fp8_utils.py:310-326; weights that arrive unquantised: w8a8_fp8.py:129,
// fp8.py:375): amax = max |x| clamped at 1e-12, scale = 448 / amax (evaluated as torch does: (1 / amax) * 448), q = sat(x * scale) (RNE),
// returns 1 / scale.  The reference's
// arithmetic order, not the per-tensor kernel's (x * (1 / (amax / 448))): the two differ in the last bit of the factor.
template <typename T>
__global__ __launch_bounds__(256) void raw_absmax_kernel(const T* __restrict__ in, float* __restrict__ out_amax, int64_t n) {
  __shared__ float red[4];
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nvec = n / 8;
  float amax = 0.f;
  for (int64_t i = gid; i < nvec; i += stride) {
    float f[8];
    load8(in + i * 8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
  }
  for (int64_t i = nvec * 8 + gid; i < n; i += stride) amax = fmaxf(amax, fabsf((float)in[i]));
  amax = wave_reduce_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax((unsigned int*)out_amax, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}
template <typename T>
__global__ __launch_bounds__(256) void input_to_float8_kernel(const T* __restrict__ in, uint8_t* __restrict__ out_q, const float* __restrict__ amax_p,
                                                              float* __restrict__ out_s, int64_t n) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // `fp_max / amax` with a Python float on the left is Tensor.__rtruediv__ = amax.reciprocal() * fp_max in torch: two roundings
  const float scale = (1.0f / fmaxf(*amax_p, 1e-12f)) * kFp8Max;
  if (gid == 0) *out_s = 1.0f / scale;
  const int64_t nvec = n / 8;
  for (int64_t i = gid; i < nvec; i += stride) {
    float f[8];
    load8(in + i * 8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = clamp448(f[j] * scale);
    *(u32x2_t*)(out_q + i * 8) = pack8_fp8(f);
  }
  for (int64_t i = nvec * 8 + gid; i < n; i += stride) out_q[i] = cvt1_fp8(clamp448((float)in[i] * scale));
}

// 16 lanes per group (reference: threads_per_group = 16), 4 groups per wave.
template <typename T>
__global__ __launch_bounds__(256) void per_token_group_quant_kernel(const T* __restrict__ in, uint8_t* __restrict__ out_q,
                                                                    float* __restrict__ out_s, int group_size,
                                                                    int64_t num_groups, float eps, float qmin, float qmax) {
  const int64_t grp = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l16 = threadIdx.x & 15;
  const bool live = grp < num_groups;
  const T* gin = in + (live ? grp : 0) * (int64_t)group_size;
  const int nvec = group_size / 8;
  float amax = eps;
  if (live)
    for (int i = l16; i < nvec; i += 16) {
      float f[8];
      load8(gin + i * 8, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
    }
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m, WAVE));
  if (!live) return;
  const float ys = amax / qmax;
  if (l16 == 0) out_s[grp] = ys;
  uint8_t* gout = out_q + grp * (int64_t)group_size;
  for (int i = l16; i < nvec; i += 16) {
    float f[8];
    load8(gin + i * 8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = clamp448(f[j] / ys, qmin, qmax);
    *(u32x2_t*)(gout + i * 8) = pack8_fp8(f);
  }
}

// Decode-sized batches (few tokens, one 256-thread workgroup each): the whole row is requested up front and held in registers
// -- with the two strided loops of the general kernel every 16-byte piece was its own dependent L2 round trip (7 per thread at
// hidden = 14336, twice), which is all a kernel with 32 workgroups is made of.  Same arithmetic, bit for bit.
// NT: prefill-sized launches (the row is read once, the codes are read once by a GEMM that streams gigabytes): non-temporal loads and
// stores, so the stream does not evict what the neighbouring kernels re-read
template <typename T, int MAXV, bool NT = false>
__global__ __launch_bounds__(256) void per_token_quant_row_kernel(const T* __restrict__ in, uint8_t* __restrict__ out_q,
                                                                  float* __restrict__ out_s, int hidden, int64_t in_stride) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int64_t token = blockIdx.x;
  const T* row = in + token * in_stride;
  const int nvec = hidden / 8;
  u32x4_t raw[MAXV];
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int i = tid + it * 256;
    if constexpr (NT) raw[it] = __builtin_nontemporal_load((const u32x4_t*)(row + (i < nvec ? i : 0) * 8));
    else raw[it] = *(const u32x4_t*)(row + (i < nvec ? i : 0) * 8);
  }
  float vals[MAXV][8];
  float amax = 0.f;
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    struct V8 { T v[8]; };
    const V8 x = __builtin_bit_cast(V8, raw[it]);
    const bool on = tid + it * 256 < nvec;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      vals[it][j] = (float)x.v[j];
      if (on) amax = fmaxf(amax, fabsf(vals[it][j]));
    }
  }
  amax = wave_reduce_max(amax);
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float scale = amax / kFp8Max;
  if (tid == 0) out_s[token] = scale;
  const float inv = (scale == 0.f) ? 0.f : 1.0f / scale;
  uint8_t* orow = out_q + token * hidden;
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int i = tid + it * 256;
    if (i < nvec) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = clamp448(vals[it][j] * inv);
      if constexpr (NT) __builtin_nontemporal_store(pack8_fp8(f), (u32x2_t*)(orow + i * 8));
      else *(u32x2_t*)(orow + i * 8) = pack8_fp8(f);
    }
  }
}

template <typename T>
int launch_per_token(const void* in, void* q, float* s, int64_t tokens, int64_t hidden, int64_t in_stride, hipStream_t st) {
#ifndef SGL_QUANT_ROW_WIDE
#define SGL_QUANT_ROW_WIDE 1
#endif
  // the register-resident row kernel: decode-sized batches, and (round 3) any batch of WIDE rows -- the general kernel walks a row
  // twice with one wave (the second pass out of L2); at 65 536 x 14 336 it moved its 2.8 GB in 790 us = 3.6 TB/s
  const bool wide = SGL_QUANT_ROW_WIDE && hidden >= 4096;
  if ((tokens <= 512 || wide) && tokens < (1ll << 31) && hidden <= 256 * 8 * 8 && in_stride % 8 == 0 && ((uintptr_t)in % 16) == 0) {
#ifndef SGL_ROW_NT
#define SGL_ROW_NT 1
#endif
    // (tools/debug/quant_time.py: 65 536 x 14 336 520 -> 447 us, 65 536 x 4096 138 -> 124 us, 16 384 x 14 336 128 -> 109 us)
    const bool nt = SGL_ROW_NT && tokens * hidden >= (1ll << 27);
#define SGL_ROWQ(MV)                                                                                                             \
  do {                                                                                                                           \
    if (nt) hipLaunchKernelGGL((per_token_quant_row_kernel<T, MV, true>), dim3((unsigned)tokens), dim3(256), 0, st, (const T*)in, \
                               (uint8_t*)q, s, (int)hidden, in_stride);                                                          \
    else hipLaunchKernelGGL((per_token_quant_row_kernel<T, MV, false>), dim3((unsigned)tokens), dim3(256), 0, st, (const T*)in,   \
                            (uint8_t*)q, s, (int)hidden, in_stride);                                                             \
  } while (0)
    if (hidden <= 256 * 8 * 2) SGL_ROWQ(2);
    else if (hidden <= 256 * 8 * 4) SGL_ROWQ(4);
    else SGL_ROWQ(8);
#undef SGL_ROWQ
    SGL_HIP_LAUNCH_CHECK();
    return SGL_MI355_OK;
  }
  if (tokens >= 2048) {
    hipLaunchKernelGGL((per_token_quant_kernel<T, 256, 4>), dim3((unsigned)((tokens + 3) / 4)), dim3(256), 0, st,
                       (const T*)in, (uint8_t*)q, s, hidden, tokens, in_stride);
  } else {
    hipLaunchKernelGGL((per_token_quant_kernel<T, 256, 1>), dim3((unsigned)tokens), dim3(256), 0, st, (const T*)in,
                       (uint8_t*)q, s, hidden, tokens, in_stride);
  }
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

}  // namespace

extern "C" int sgl_mi355_per_token_quant_fp8(const void* input, int64_t input_stride, void* output_q, float* output_s,
                                             int64_t num_tokens, int64_t hidden_dim, int in_dtype, void* stream) {
  SGL_CHECK(num_tokens >= 0 && hidden_dim >= 0, "per_token_quant_fp8: negative shape");
  if (num_tokens == 0 || hidden_dim == 0) return SGL_MI355_OK;
  SGL_CHECK(input && output_q && output_s, "per_token_quant_fp8: null pointer");
  // reference: "Hidden dimension must be divisible by 8" (per_token_quant_fp8.cu:173)
  SGL_CHECK(hidden_dim % 8 == 0, "Hidden dimension must be divisible by 8, but got %lld", (long long)hidden_dim);
  SGL_CHECK(input_stride % 8 == 0 && ((uintptr_t)input % 16) == 0 && ((uintptr_t)output_q % 8) == 0,
            "per_token_quant_fp8: rows must be 16-byte aligned");
  SGL_CHECK(in_dtype == SGL_BF16 || in_dtype == SGL_F16, "per_token_quant_fp8: input must be bf16 or f16");
  hipStream_t st = (hipStream_t)stream;
  return in_dtype == SGL_BF16 ? launch_per_token<__bf16>(input, output_q, output_s, num_tokens, hidden_dim, input_stride, st)
                              : launch_per_token<_Float16>(input, output_q, output_s, num_tokens, hidden_dim, input_stride, st);
}

extern "C" int sgl_mi355_per_tensor_quant_fp8(const void* input, void* output_q, float* output_s, int64_t num_elements,
                                              int is_static, int in_dtype, void* stream) {
  SGL_CHECK(num_elements >= 0, "per_tensor_quant_fp8: negative size");
  if (num_elements == 0) return SGL_MI355_OK;
  SGL_CHECK(input && output_q && output_s, "per_tensor_quant_fp8: null pointer");
  SGL_CHECK(((uintptr_t)input % 16) == 0 && ((uintptr_t)output_q % 8) == 0, "per_tensor_quant_fp8: misaligned tensor");
  SGL_CHECK(in_dtype == SGL_BF16 || in_dtype == SGL_F16, "per_tensor_quant_fp8: input must be bf16 or f16");
  hipStream_t st = (hipStream_t)stream;
  const int64_t blocks64 = (num_elements / 8 + 255) / 256;
  const unsigned blocks = (unsigned)(blocks64 < 1 ? 1 : (blocks64 > 2048 ? 2048 : blocks64));
  if (in_dtype == SGL_BF16) {
    if (!is_static)
      hipLaunchKernelGGL((per_tensor_absmax_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, (const __bf16*)input, output_s, num_elements);
    hipLaunchKernelGGL((per_tensor_quant_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, (const __bf16*)input,
                       (uint8_t*)output_q, output_s, num_elements);
  } else {
    if (!is_static)
      hipLaunchKernelGGL((per_tensor_absmax_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, (const _Float16*)input, output_s, num_elements);
    hipLaunchKernelGGL((per_tensor_quant_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, (const _Float16*)input,
                       (uint8_t*)output_q, output_s, num_elements);
  }
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_input_to_float8(const void* input, void* output_q, float* output_scale_inv, float* amax_scratch,
                                         int64_t num_elements, int in_dtype, void* stream) {
  SGL_CHECK(num_elements >= 0, "input_to_float8: negative size");
  SGL_CHECK(output_scale_inv && amax_scratch, "input_to_float8: null scale pointer");
  if (num_elements == 0) return SGL_MI355_OK;
  SGL_CHECK(input && output_q, "input_to_float8: null pointer");
  SGL_CHECK(((uintptr_t)input % 16) == 0 && ((uintptr_t)output_q % 8) == 0, "input_to_float8: misaligned tensor");
  SGL_CHECK(in_dtype == SGL_BF16 || in_dtype == SGL_F16, "input_to_float8: input must be bf16 or f16");
  hipStream_t st = (hipStream_t)stream;
  const int64_t blocks64 = (num_elements / 8 + 255) / 256;
  const unsigned blocks = (unsigned)(blocks64 < 1 ? 1 : (blocks64 > 2048 ? 2048 : blocks64));
  if (hipMemsetAsync(amax_scratch, 0, sizeof(float), st) != hipSuccess) {
    snprintf(g_sgl_mi355_err, sizeof(g_sgl_mi355_err), "input_to_float8: hipMemsetAsync failed");
    return SGL_MI355_EHIP;
  }
  if (in_dtype == SGL_BF16) {
    hipLaunchKernelGGL((raw_absmax_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, (const __bf16*)input, amax_scratch, num_elements);
    hipLaunchKernelGGL((input_to_float8_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, (const __bf16*)input, (uint8_t*)output_q,
                       amax_scratch, output_scale_inv, num_elements);
  } else {
    hipLaunchKernelGGL((raw_absmax_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, (const _Float16*)input, amax_scratch, num_elements);
    hipLaunchKernelGGL((input_to_float8_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, (const _Float16*)input, (uint8_t*)output_q,
                       amax_scratch, output_scale_inv, num_elements);
  }
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_per_token_group_quant_fp8(const void* input, void* output_q, float* output_s,
                                                   int64_t num_elements, int group_size, float eps, float fp8_min,
                                                   float fp8_max, int in_dtype, void* stream) {
  SGL_CHECK(group_size > 0 && group_size % 8 == 0, "per_token_group_quant_fp8: group_size %d must be a positive multiple of 8", group_size);
  SGL_CHECK(num_elements >= 0 && num_elements % group_size == 0, "per_token_group_quant_fp8: numel not divisible by group_size");
  if (num_elements == 0) return SGL_MI355_OK;
  SGL_CHECK(input && output_q && output_s, "per_token_group_quant_fp8: null pointer");
  SGL_CHECK(in_dtype == SGL_BF16 || in_dtype == SGL_F16, "per_token_group_quant_fp8: input must be bf16 or f16");
  const int64_t groups = num_elements / group_size;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((groups + 15) / 16);
  if (in_dtype == SGL_BF16)
    hipLaunchKernelGGL((per_token_group_quant_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, (const __bf16*)input,
                       (uint8_t*)output_q, output_s, group_size, groups, eps, fp8_min, fp8_max);
  else
    hipLaunchKernelGGL((per_token_group_quant_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, (const _Float16*)input,
                       (uint8_t*)output_q, output_s, group_size, groups, eps, fp8_min, fp8_max);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}
