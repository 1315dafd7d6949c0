// Elementwise neighbours of the hot path on the decode/extend step: RMSNorm (+fused
// residual add), SiLU*mul, rotary embedding, embedding gather, greedy argmax.
//
// Rounding points follow the reference's torch-native forwards so that an end-to-end step can be
// compared with the torch-native model at the logits level:
//   RMSNorm.forward_native       python/sglang/srt/layers/layernorm.py:135-171
//       xf = f32(x) [+ f32(residual)]; residual' = T(xf); y = T((xf * rsqrt(mean(xf^2) + eps)) * w)
//   SiluAndMul.forward_native    python/sglang/srt/layers/activation.py:60-63
//       T(T(silu(a)) * b)   (silu evaluated in f32, rounded to T, product rounded to T)
//   RotaryEmbedding.forward_native + _apply_rotary_emb  rotary_embedding.py:49-72,138-165
//       cos/sin are cast to T first; every product and the add/sub round to T
// Native CUDA counterparts (not built for ROCm by the reference): sgl-kernel/csrc/elementwise/*.cu.
// All kernels are HBM/L2-bound byte work: 16-byte accesses, one workgroup (or wave) per token row.
#include "common.h"

namespace {

template <typename T>
struct V8 {
  T v[8];
};

template <typename T>
__device__ __forceinline__ V8<T> ld8(const T* p) {
  return __builtin_bit_cast(V8<T>, *(const u32x4_t*)p);
}
template <typename T>
__device__ __forceinline__ void st8(T* p, const V8<T>& x) {
  *(u32x4_t*)p = __builtin_bit_cast(u32x4_t, x);
}

// Round an f32 to T and come back as f32 through the BIT PATTERN, so that the front end cannot keep the value in
// excess precision across the cast (clang evaluates __bf16/_Float16 expressions in float and may elide a
// T -> float -> T round trip); these are the reference's rounding points and must really happen.
template <typename T>
__device__ __forceinline__ float round_via(float x) {
  asm volatile("" : "+v"(x));  // x must exist as an f32 first: no v_fma_mix* single-rounding shortcut (the reference rounds twice)
  const T t = (T)x;
  const uint16_t u = __builtin_bit_cast(uint16_t, t);
  uint16_t v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(u));
  return (float)__builtin_bit_cast(T, v);
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_reduce_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// x, residual updated in place when residual != null (sgl_kernel.fused_add_rmsnorm contract);
// otherwise out = rmsnorm(x).  hidden % 8 == 0.  Up to 8192 columns stay in registers.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_kernel(T* out, const T* x, T* residual, const T* weight, float eps,
                                                      int hidden, int64_t x_stride, int64_t out_stride) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const T* xr = x + row * x_stride;
  T* rr = residual ? residual + row * (int64_t)hidden : nullptr;
  T* orow = out + row * out_stride;
  const int nvec = hidden / 8;
  float vals[MAXV][8];
  float ss = 0.f;
  V8<T> wreg[MAXV];  // the norm weight travels with the inputs (after the reduction it was one more dependent round trip)
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int i = threadIdx.x + it * 256;
    wreg[it] = ld8(weight + (i < nvec ? i : 0) * 8);
  }
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int i = threadIdx.x + it * 256;
    if (i < nvec) {
      const V8<T> a = ld8(xr + i * 8);
      if (rr) {
        const V8<T> r = ld8(rr + i * 8);
        V8<T> ro;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = (float)a.v[j] + (float)r.v[j];
          vals[it][j] = f;
          ro.v[j] = (T)f;
        }
        st8(rr + i * 8, ro);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[it][j] = (float)a.v[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ss = fmaf(vals[it][j], vals[it][j], ss);  // (explicit: see fused_decode.hip)
    }
  }
  const float var = block_sum_256(ss, red) / (float)hidden;
  const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int i = threadIdx.x + it * 256;
    if (i < nvec) {
      V8<T> o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = (T)round_via<T>((vals[it][j] * rs) * (float)wreg[it].v[j]);
      st8(orow + i * 8, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void silu_and_mul_kernel(T* out, const T* x, int64_t tokens, int d) {
  const int nvec = d / 8;
  const int64_t total = tokens * nvec;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t t = idx / nvec;
    const int c = (int)(idx - t * nvec);
    const V8<T> a = ld8(x + t * 2 * d + c * 8);
    const V8<T> b = ld8(x + t * 2 * d + d + c * 8);
    V8<T> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float af = (float)a.v[j];
      const float s = round_via<T>(af / (1.0f + expf(-af)));
      o.v[j] = (T)(s * (float)b.v[j]);
    }
    st8(out + t * d + c * 8, o);
  }
}

// In-place rotary embedding on q [T, Hq, hs] and k [T, Hk, hs] (token strides given), cos_sin_cache f32
// [max_pos, rot_dim] = cat(cos, sin).  One thread per rotation pair.
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T* q, T* k, const int64_t* positions, const float* cache, int64_t tokens,
                                                   int hq, int hk, int head_size, int rot_dim, int64_t q_stride,
                                                   int64_t k_stride, int is_neox) {
  const int half = rot_dim / 2;
  const int pairs_per_token = (hq + hk) * half;
  const int64_t total = tokens * pairs_per_token;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t t = idx / pairs_per_token;
    const int rem = (int)(idx - t * pairs_per_token);
    const int h = rem / half, i = rem - h * half;
    T* base = (h < hq) ? q + t * q_stride + (int64_t)h * head_size : k + t * k_stride + (int64_t)(h - hq) * head_size;
    const float* cs = cache + positions[t] * rot_dim;
    const float c = round_via<T>(cs[i]), s = round_via<T>(cs[half + i]);
    const int i1 = is_neox ? i : 2 * i, i2 = is_neox ? half + i : 2 * i + 1;
    const float x1 = (float)base[i1], x2 = (float)base[i2];
    const float p11 = round_via<T>(x1 * c), p22 = round_via<T>(x2 * s);
    const float p21 = round_via<T>(x2 * c), p12 = round_via<T>(x1 * s);
    base[i1] = (T)(p11 - p22);
    base[i2] = (T)(p21 + p12);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embedding_kernel(T* out, const int64_t* ids, const T* table, int64_t tokens, int hidden,
                                                        int64_t vocab) {
  const int nvec = hidden / 8;
  const int64_t total = tokens * nvec;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t t = idx / nvec;
    const int c = (int)(idx - t * nvec);
    const int64_t id = ids[t];
    // F.embedding raises a device-side assert for an id outside the table; here such a row comes out as NaNs (0x7FC0 is a NaN
    // in bf16 and in f16) -- visible downstream, and never an out-of-range read (which is a GPU memory fault)
    u32x4_t v = u32x4_t{0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u};
    if (id >= 0 && id < vocab) v = *(const u32x4_t*)(table + id * hidden + c * 8);
    *(u32x4_t*)(out + t * hidden + c * 8) = v;
  }
}

// first index of the row maximum (torch.argmax tie-break on CPU/GPU: lowest index)
template <typename T>
__global__ __launch_bounds__(256) void argmax_kernel(int64_t* out, const T* logits, int64_t vocab, int64_t stride) {
  __shared__ float rv[4];
  __shared__ int64_t ri[4];
  const T* row = logits + (int64_t)blockIdx.x * stride;
  float best = -INFINITY;
  int64_t bi = 0x7fffffffffffffffLL;
  for (int64_t i = threadIdx.x; i < vocab; i += 256) {
    const float v = (float)row[i];
    if (argmax_beats(v, i, best, bi)) { best = v; bi = i; }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float ov = __shfl_xor(best, m, WAVE);
    const int64_t oi = __shfl_xor(bi, m, WAVE);
    if (argmax_beats(ov, oi, best, bi)) { best = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = best; ri[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (argmax_beats(rv[w], ri[w], best, bi)) { best = rv[w]; bi = ri[w]; }
    out[blockIdx.x] = bi;
  }
}

inline unsigned grid_for(int64_t work_items) {
  const int64_t b = (work_items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

#define DISPATCH_HALF(dtype, ...)                   \
  if ((dtype) == SGL_BF16) {                        \
    using T = __bf16;                               \
    __VA_ARGS__                                     \
  } else {                                          \
    using T = _Float16;                             \
    __VA_ARGS__                                     \
  }

}  // namespace

// residual == NULL: out = rmsnorm(x) (out may alias x).  residual != NULL: residual += x (rounded to T),
// out = rmsnorm(f32 sum) -- pass out == x for the reference's in-place fused_add_rmsnorm.
extern "C" int sgl_mi355_rmsnorm(void* out, const void* x, void* residual, const void* weight, float eps, int64_t tokens,
                                 int hidden, int64_t x_stride, int64_t out_stride, int dtype, void* stream) {
  SGL_CHECK(tokens >= 0 && hidden > 0, "rmsnorm: bad shape");
  if (tokens == 0) return SGL_MI355_OK;
  SGL_CHECK(out && x && weight, "rmsnorm: null pointer");
  SGL_CHECK(hidden % 8 == 0 && hidden <= 16384, "rmsnorm: hidden=%d must be a multiple of 8 and <= 16384", hidden);
  SGL_CHECK(x_stride % 8 == 0 && out_stride % 8 == 0, "rmsnorm: row strides must be multiples of 8 elements");
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "rmsnorm: dtype must be bf16 or f16");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_HALF(dtype, {
    if (hidden <= 8192)
      hipLaunchKernelGGL((rmsnorm_kernel<T, 4>), dim3((unsigned)tokens), dim3(256), 0, st, (T*)out, (const T*)x, (T*)residual,
                         (const T*)weight, eps, hidden, x_stride, out_stride);
    else
      hipLaunchKernelGGL((rmsnorm_kernel<T, 8>), dim3((unsigned)tokens), dim3(256), 0, st, (T*)out, (const T*)x, (T*)residual,
                         (const T*)weight, eps, hidden, x_stride, out_stride);
  })
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_silu_and_mul(void* out, const void* x, int64_t tokens, int d, int dtype, void* stream) {
  SGL_CHECK(tokens >= 0 && d > 0 && d % 8 == 0, "silu_and_mul: d=%d must be a positive multiple of 8", d);
  if (tokens == 0) return SGL_MI355_OK;
  SGL_CHECK(out && x, "silu_and_mul: null pointer");
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "silu_and_mul: dtype must be bf16 or f16");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_HALF(dtype, {
    hipLaunchKernelGGL((silu_and_mul_kernel<T>), dim3(grid_for(tokens * (d / 8))), dim3(256), 0, st, (T*)out, (const T*)x, tokens, d);
  })
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_rotary_embedding(const int64_t* positions, void* query, void* key, const float* cos_sin_cache,
                                          int64_t tokens, int num_q_heads, int num_k_heads, int head_size, int rot_dim,
                                          int64_t q_stride, int64_t k_stride, int is_neox, int dtype, void* stream) {
  SGL_CHECK(tokens >= 0, "rotary_embedding: negative token count");
  if (tokens == 0) return SGL_MI355_OK;
  SGL_CHECK(positions && query && key && cos_sin_cache, "rotary_embedding: null pointer");
  SGL_CHECK(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, "rotary_embedding: rot_dim=%d invalid for head_size=%d", rot_dim, head_size);
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "rotary_embedding: dtype must be bf16 or f16");
  hipStream_t st = (hipStream_t)stream;
  const int64_t work = tokens * (num_q_heads + num_k_heads) * (rot_dim / 2);
  DISPATCH_HALF(dtype, {
    hipLaunchKernelGGL((rope_kernel<T>), dim3(grid_for(work)), dim3(256), 0, st, (T*)query, (T*)key, positions, cos_sin_cache,
                       tokens, num_q_heads, num_k_heads, head_size, rot_dim, q_stride, k_stride, is_neox);
  })
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_embedding(void* out, const int64_t* ids, const void* table, int64_t tokens, int hidden, int64_t vocab,
                                   int dtype, void* stream) {
  SGL_CHECK(tokens >= 0 && hidden > 0 && hidden % 8 == 0, "embedding: hidden=%d must be a positive multiple of 8", hidden);
  SGL_CHECK(vocab > 0, "embedding: vocab=%lld (the table's row count) must be positive", (long long)vocab);
  if (tokens == 0) return SGL_MI355_OK;
  SGL_CHECK(out && ids && table, "embedding: null pointer");
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "embedding: dtype must be bf16 or f16");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((embedding_kernel<__bf16>), dim3(grid_for(tokens * (hidden / 8))), dim3(256), 0, st, (__bf16*)out, ids,
                     (const __bf16*)table, tokens, hidden, vocab);  // pure 16-bit copy: one instantiation serves both dtypes
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_argmax(int64_t* out, const void* logits, int64_t rows, int64_t vocab, int64_t row_stride, int dtype,
                                void* stream) {
  SGL_CHECK(rows >= 0 && vocab > 0, "argmax: bad shape");
  if (rows == 0) return SGL_MI355_OK;
  SGL_CHECK(out && logits, "argmax: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SGL_F32) {
    hipLaunchKernelGGL((argmax_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, out, (const float*)logits, vocab, row_stride);
  } else {
    SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "argmax: dtype must be f32, bf16 or f16");
    DISPATCH_HALF(dtype, {
      hipLaunchKernelGGL((argmax_kernel<T>), dim3((unsigned)rows), dim3(256), 0, st, out, (const T*)logits, vocab, row_stride);
    })
  }
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

namespace {
// out[c][r] = in[r][c] for 16-bit elements, 64x64 tiles through LDS (both sides coalesced).
__global__ __launch_bounds__(256) void transpose16_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ in, int rows,
                                                          int cols) {
  __shared__ uint16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(int64_t)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 64; i += 4)
    if (c0 + i < cols && r0 + tx < rows) out[(int64_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}
}  // namespace

// out [cols, rows] = in[rows, cols]^T for bf16/f16 (weight re-layout for AWQLinearMethod: awq.py:401-418 multiplies
// by the [K, N] dequantised weight, the GEMMs here stream W as [N, K]).
extern "C" int sgl_mi355_transpose_2d(void* out, const void* in, int rows, int cols, void* stream) {
  SGL_CHECK(rows >= 0 && cols >= 0, "transpose_2d: negative shape");
  if (rows == 0 || cols == 0) return SGL_MI355_OK;
  SGL_CHECK(out && in, "transpose_2d: null pointer");
  hipLaunchKernelGGL(transpose16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (uint16_t*)out, (const uint16_t*)in, rows, cols);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}


// ---------------------------------------------------------------------------------------------------------
// merge_state / merge_state_v2 (sgl-kernel/csrc/attention/merge_attn_states.cu:32-105, python attention.py:12-52):
// LSE-weighted merge of two attention partials (cascade "shared prefix once, private suffix per request" decode):
//   m = max(sa, sb) (+inf -> -inf first); wa = e^(sa-m), wb = e^(sb-m); v = va * wa/(wa+wb) + vb * wb/(wa+wb);
//   s = log(wa + wb) + m.   v [n, h, d] in bf16 / f16 / f32, s f32 [n, h].  One 16-byte pack per thread.
// ---------------------------------------------------------------------------------------------------------
namespace {
template <typename T>
__global__ __launch_bounds__(128) void merge_state_kernel(const T* va, const float* sa, const T* vb, const float* sb, T* vo,
                                                          float* so, int64_t nh, int d) {
  constexpr int PACK = 16 / sizeof(T);
  const int tph = d / PACK;
  const int64_t gid = (int64_t)blockIdx.x * 128 + threadIdx.x;
  if (gid >= nh * tph) return;
  const int64_t th = gid / tph;
  const int pk = (int)(gid - th * tph);
  float p = sa[th], q = sb[th];
  p = isinf(p) ? -INFINITY : p;
  q = isinf(q) ? -INFINITY : q;
  const float m = fmaxf(p, q);
  const float pe = expf(p - m), qe = expf(q - m);
  const float se = pe + qe;
  const float ps = pe / se, qs = qe / se;
  struct P { T v[PACK]; };
  const P a = *(const P*)(va + th * d + pk * PACK), b = *(const P*)(vb + th * d + pk * PACK);
  P o;
#pragma unroll
  for (int i = 0; i < PACK; ++i) o.v[i] = (T)((float)a.v[i] * ps + ((float)b.v[i] * qs));
  *(P*)(vo + th * d + pk * PACK) = o;
  if (so != nullptr && pk == 0) so[th] = logf(se) + m;
}
}  // namespace

extern "C" int sgl_mi355_merge_state(const void* v_a, const float* s_a, const void* v_b, const float* s_b, void* v_merged,
                                     float* s_merged, int64_t num_tokens, int num_heads, int head_size, int dtype,
                                     void* stream) {
  SGL_CHECK(num_tokens >= 0 && num_heads > 0 && head_size > 0, "merge_state: bad shape");
  if (num_tokens == 0) return SGL_MI355_OK;
  SGL_CHECK(v_a && s_a && v_b && s_b && v_merged, "merge_state: null pointer");
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16 || dtype == SGL_F32, "merge_state: dtype must be bf16, f16 or f32");
  const int pack = dtype == SGL_F32 ? 4 : 8;
  SGL_CHECK(head_size % pack == 0, "headsize must be multiple of pack_size: %d", pack);
  SGL_CHECK(((uintptr_t)v_a % 16) == 0 && ((uintptr_t)v_b % 16) == 0 && ((uintptr_t)v_merged % 16) == 0,
            "merge_state: tensors must be 16-byte aligned");
  const int64_t nh = num_tokens * num_heads, threads = nh * (head_size / pack);
  const dim3 grid((unsigned)((threads + 127) / 128));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SGL_BF16)
    hipLaunchKernelGGL((merge_state_kernel<__bf16>), grid, dim3(128), 0, st, (const __bf16*)v_a, s_a, (const __bf16*)v_b, s_b,
                       (__bf16*)v_merged, s_merged, nh, head_size);
  else if (dtype == SGL_F16)
    hipLaunchKernelGGL((merge_state_kernel<_Float16>), grid, dim3(128), 0, st, (const _Float16*)v_a, s_a, (const _Float16*)v_b,
                       s_b, (_Float16*)v_merged, s_merged, nh, head_size);
  else
    hipLaunchKernelGGL((merge_state_kernel<float>), grid, dim3(128), 0, st, (const float*)v_a, s_a, (const float*)v_b, s_b,
                       (float*)v_merged, s_merged, nh, head_size);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}
