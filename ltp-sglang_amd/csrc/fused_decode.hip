// Fused elementwise kernels of the decode step: each one is bit-identical to the sequence of stand-alone kernels it
// replaces (same rounding points), it only removes launches and HBM round trips.  At batch 32 every one of these
// ops is a ~2-5 us launch, 17 of them per layer; fused they are 4.
//
//   fused_add_rmsnorm_quant_fp8 = [splitk_reduce ->] fused_add_rmsnorm -> sgl_per_token_quant_fp8
//       (layernorm.py:135-171 + per_token_quant_fp8.cu:15-228)
//   silu_and_mul_quant_fp8      = silu_and_mul -> sgl_per_token_quant_fp8      (activation.py:60-63)
//   rope_set_kv                 = rotary_embedding -> set_kv_buffer             (rotary_embedding.py:138-165,
//                                                                                memory_pool.py:369-407)
//   decode_merge_quant_fp8      = decode stage-2 LSE merge -> sgl_per_token_quant_fp8
//       (decode_attention.py:492-552)
// All HBM/L2-bound byte work: one 256-thread workgroup per token, 16-byte accesses.
#include "row_helpers.h"
#include "silu_lut.h"

#ifndef SGL_ROW_NT
#define SGL_ROW_NT 1   // 0: ordinary loads / stores in the prefill-sized row kernels (A/B builds)
#endif

namespace {

template <typename T, int MAXV, bool SLABS = true, bool NT = false>  // SLABS = false drops the split-K registers: 4x the occupancy at prefill-sized M;
                                                                       // NT: non-temporal row loads / stores (prefill-sized launches)
__global__ __launch_bounds__(256) void add_rmsnorm_quant_kernel(const T* x, const float* slabs_, int nslabs,
                                                                const float* slab_sx, const float* slab_sw, T* residual,
                                                                const T* weight, float eps, T* out_norm, uint8_t* out_q,
                                                                float* out_s, int tokens, int hidden) {
  __shared__ float red[4];
  const float* slabs = SLABS ? slabs_ : nullptr;
  const int64_t row = blockIdx.x;
  const int nvec = hidden / 8;
  // Every global load of the kernel is issued up front (inputs were just written by other CUs, so each dependent
  // round trip costs ~2 us here): the norm weight travels with the inputs instead of after the first reduction.
  V8<T> wreg[MAXV], rreg[MAXV], xreg[MAXV];
  f32x4_t s0[MAXV][2], s1[MAXV][2], s2[MAXV][2], s3[MAXV][2], swr[MAXV][2];  // (four slabs prefetched: K = 14336 makes 4)
  const float sxm = (slabs && slab_sx) ? slab_sx[row] : 1.0f;
  const int ns = slabs ? nslabs : 0;
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int i = threadIdx.x + it * 256;
    const int ic = i < nvec ? i : 0;  // clamped: lanes past the row read column 0 (harmless) instead of branching
    wreg[it] = ld8(weight + ic * 8);
    if (residual) rreg[it] = NT ? ld8_nt(residual + row * hidden + ic * 8) : ld8(residual + row * hidden + ic * 8);
    if (slabs) {
      const float* sp = slabs + row * hidden + ic * 8;
      s0[it][0] = *(const f32x4_t*)sp;
      s0[it][1] = *(const f32x4_t*)(sp + 4);
      if (ns > 1) {
        const float* sq = sp + (int64_t)tokens * hidden;
        s1[it][0] = *(const f32x4_t*)sq;
        s1[it][1] = *(const f32x4_t*)(sq + 4);
      }
      if (ns > 3) {
        const float* sq = sp + 2 * (int64_t)tokens * hidden;
        s2[it][0] = *(const f32x4_t*)sq;
        s2[it][1] = *(const f32x4_t*)(sq + 4);
        const float* sr = sp + 3 * (int64_t)tokens * hidden;
        s3[it][0] = *(const f32x4_t*)sr;
        s3[it][1] = *(const f32x4_t*)(sr + 4);
      }
      if (slab_sw) {
        swr[it][0] = *(const f32x4_t*)(slab_sw + ic * 8);
        swr[it][1] = *(const f32x4_t*)(slab_sw + ic * 8 + 4);
      }
    } else {
      xreg[it] = NT ? ld8_nt(x + row * hidden + ic * 8) : ld8(x + row * hidden + ic * 8);
    }
  }
  float vals[MAXV][8];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int i = threadIdx.x + it * 256;
    if (i < nvec) {
      float f[8];
      if (slabs) {  // x = T((sum of the split-K partial sums) * sx[m] * sw[n]): the GEMM epilogue, fused
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = s0[it][j >> 2][j & 3];
        if (ns > 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] += s1[it][j >> 2][j & 3];
          if (ns > 3) {  // (same order of additions as the loop below)
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += s2[it][j >> 2][j & 3];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += s3[it][j >> 2][j & 3];
          }
          for (int sI = (ns > 3 ? 4 : 2); sI < ns; ++sI) {
            const float* sp = slabs + ((int64_t)sI * tokens + row) * hidden + i * 8;
            const f32x4_t a0 = *(const f32x4_t*)sp, a1 = *(const f32x4_t*)(sp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[j] += a0[j]; f[4 + j] += a1[j]; }
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = round_via<T>(f[j] * sxm * (slab_sw ? swr[it][j >> 2][j & 3] : 1.0f));
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)xreg[it].v[j];
      }
      if (residual) {
        V8<T> ro;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          f[j] += (float)rreg[it].v[j];
          ro.v[j] = (T)f[j];
        }
        if constexpr (NT) st8_nt(residual + row * hidden + i * 8, ro);
        else st8(residual + row * hidden + i * 8, ro);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        vals[it][j] = f[j];
        ss = fmaf(f[j], f[j], ss);  // explicit fma: left to hipcc, this contraction is made in some kernels and not in others (packed
                                    // f32 mul + add), and the fused forms of this arithmetic must round like this one
      }
    }
  }
  const float var = block_sum(ss, red) / (float)hidden;
  const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int i = threadIdx.x + it * 256;
    if (i < nvec) {
      V8<T> o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        vals[it][j] = round_via<T>((vals[it][j] * rs) * (float)wreg[it].v[j]);
        o.v[j] = (T)vals[it][j];
      }
      if (out_norm) st8(out_norm + row * hidden + i * 8, o);
    }
  }
  if (out_q) quant_row<MAXV, NT>(vals, nvec, out_q + row * hidden, out_s + row, red);
}

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void silu_mul_quant_kernel(const T* x, uint8_t* out_q, float* out_s, int d) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const int nvec = d / 8;
  float vals[MAXV][8];
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int i = threadIdx.x + it * 256;
    if (i < nvec) {
      const V8<T> a = ld8(x + row * 2 * d + i * 8), b = ld8(x + row * 2 * d + d + i * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float af = (float)a.v[j];
        const float sI = round_via<T>(af / (1.0f + expf(-af)));
        vals[it][j] = round_via<T>(sI * (float)b.v[j]);
      }
    }
  }
  quant_row<MAXV>(vals, nvec, out_q + row * d, out_s + row, red);
}

// Prefill-sized form for bf16 (round 3).  The exact silu -- expf + IEEE division, ~25 of the kernel's ~40 VALU instructions per
// element -- makes silu_mul_quant_kernel VALU-bound at 65 536 x 14 336 (1.03 ms for 4.7 GB = 4.6 TB/s; add_rmsnorm_quant moves
// its bytes at 6.3).  T(silu(a)) is a function of a's 16 bits: every workgroup tabulates it ONCE with the exact expression above
// (so the table cannot differ from it), 2 x 37 exponents x 128 mantissas = 9 472 entries = 18.5 KiB of LDS covering 2^-30 <= |a|
// < 128, and walks `rows_per_wg` rows with one ds_read_u16 per element; anything outside the table (zeros, denormals, tiny or
// huge values, inf, nan) takes the exact expression.  Bit-identical to silu_mul_quant_kernel by construction.

template <int MAXV, bool NT = false>   // NT: non-temporal row loads / stores (prefill-sized launches)
__global__ __launch_bounds__(256) void silu_mul_quant_lut_kernel(const __bf16* x, uint8_t* out_q, float* out_s, int d, int tokens,
                                                                 int rows_per_wg) {
  using T = __bf16;
  __shared__ float red[4];
  __shared__ uint16_t lut[kSiluLut];
  for (int i = threadIdx.x; i < kSiluLut; i += 256) lut[i] = silu_lut_entry(i);
  __syncthreads();
  const int nvec = d / 8;
  const int row_end = min(tokens, ((int)blockIdx.x + 1) * rows_per_wg);
  for (int64_t row = (int64_t)blockIdx.x * rows_per_wg; row < row_end; ++row) {
    float vals[MAXV][8];
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
      const int i = threadIdx.x + it * 256;
      if (i < nvec) {
        const V8<T> a = NT ? ld8_nt(x + row * 2 * d + i * 8) : ld8(x + row * 2 * d + i * 8);
        const V8<T> b = NT ? ld8_nt(x + row * 2 * d + d + i * 8) : ld8(x + row * 2 * d + d + i * 8);
        // eight lookups in flight behind ONE wait; a wave with any element outside the table (rare: zeros, tiny / huge values)
        // evaluates the exact expression for the whole vector -- a wave-uniform branch, not a per-element exec mask (with the test
        // per element every ds_read was waited for on its own)
        uint32_t rel[8];
        bool outside = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          rel[j] = silu_lut_index(__builtin_bit_cast(uint16_t, a.v[j]));
          outside = outside || rel[j] >= (uint32_t)kSiluLut;
        }
        float sI[8];
        if (__any(outside)) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            sI[j] = silu_exact_bf16((float)a.v[j]);
          }
        } else {
          uint16_t e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = lut[rel[j]];
#pragma unroll
          for (int j = 0; j < 8; ++j) sI[j] = __uint_as_float((uint32_t)e[j] << 16);
        }
        // T(sI * b): the products as f32 first (an empty asm keeps them from being folded into a mixed-precision fma), rounded in
        // pairs by v_cvt_pk_bf16_f32 and widened again by a shift -- round_via's v_mov per value is what the exact kernel spends here
        V8<T> pr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float m = sI[j] * (float)b.v[j];
          asm volatile("" : "+v"(m));
          pr.v[j] = (T)m;
        }
        u32x4_t pw = __builtin_bit_cast(u32x4_t, pr);
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(pw[q]));   // the rounded bits exist as such
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[it][j] = __uint_as_float((j & 1) ? (pw[j >> 1] & 0xFFFF0000u) : (pw[j >> 1] << 16));
      }
    }
    quant_row<MAXV, NT>(vals, nvec, out_q + row * d, out_s + row, red);
  }
}

// one workgroup per token: rotate q (in place) and k (in place + into the pool), copy v into the pool
template <typename T>
__global__ __launch_bounds__(256) void rope_set_kv_kernel(const int64_t* positions, T* q, T* k, const T* v, const float* cache,
                                                          T* k_buf, T* v_buf, const int64_t* loc, int hq, int hk,
                                                          int head_size, int rot_dim, int64_t q_stride, int64_t k_stride,
                                                          int64_t v_stride, int64_t kb_stride, int64_t vb_stride, int is_neox) {
  const int64_t t = blockIdx.x;
  const int half = rot_dim / 2;
  const float* cs = cache + positions[t] * rot_dim;
  const int64_t slot = loc[t];
  T* kdst = k_buf + slot * kb_stride;
  const int npairs = (hq + hk) * half;
  // neox layout with 16-byte aligned rows: 8 rotation pairs per thread through 16-byte loads / stores (same arithmetic)
  const bool vec = is_neox && half % 8 == 0 && head_size % 8 == 0 && q_stride % 8 == 0 && k_stride % 8 == 0 && kb_stride % 8 == 0 &&
                   ((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)k_buf % 16) == 0 && ((uintptr_t)cache % 16) == 0 &&
                   rot_dim % 4 == 0;
  if (vec) {
    const int hv = half / 8;
    for (int idx = threadIdx.x; idx < (hq + hk) * hv; idx += 256) {
      const int h = idx / hv, i0 = (idx - h * hv) * 8;
      const bool isq = h < hq;
      T* base = isq ? q + t * q_stride + (int64_t)h * head_size : k + t * k_stride + (int64_t)(h - hq) * head_size;
      const V8<T> x1 = ld8(base + i0), x2 = ld8(base + half + i0);
      const f32x4_t c0 = *(const f32x4_t*)(cs + i0), c1 = *(const f32x4_t*)(cs + i0 + 4);
      const f32x4_t s0 = *(const f32x4_t*)(cs + half + i0), s1 = *(const f32x4_t*)(cs + half + i0 + 4);
      V8<T> o1, o2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c = round_via<T>(j < 4 ? c0[j] : c1[j - 4]), sn = round_via<T>(j < 4 ? s0[j] : s1[j - 4]);
        const float a1 = (float)x1.v[j], a2 = (float)x2.v[j];
        o1.v[j] = (T)(round_via<T>(a1 * c) - round_via<T>(a2 * sn));
        o2.v[j] = (T)(round_via<T>(a2 * c) + round_via<T>(a1 * sn));
      }
      st8(base + i0, o1);
      st8(base + half + i0, o2);
      if (!isq) {
        st8(kdst + (h - hq) * head_size + i0, o1);
        st8(kdst + (h - hq) * head_size + half + i0, o2);
      }
    }
  } else
  for (int idx = threadIdx.x; idx < npairs; idx += 256) {
    const int h = idx / half, i = idx - h * half;
    const bool isq = h < hq;
    T* base = isq ? q + t * q_stride + (int64_t)h * head_size : k + t * k_stride + (int64_t)(h - hq) * head_size;
    const float c = round_via<T>(cs[i]), s = round_via<T>(cs[half + i]);
    const int i1 = is_neox ? i : 2 * i, i2 = is_neox ? half + i : 2 * i + 1;
    const float x1 = (float)base[i1], x2 = (float)base[i2];
    const T o1 = (T)(round_via<T>(x1 * c) - round_via<T>(x2 * s));
    const T o2 = (T)(round_via<T>(x2 * c) + round_via<T>(x1 * s));
    base[i1] = o1;
    base[i2] = o2;
    if (!isq) {
      kdst[(h - hq) * head_size + i1] = o1;
      kdst[(h - hq) * head_size + i2] = o2;
    }
  }
  // un-rotated tail of each k head (rot_dim < head_size) and the whole v row
  if (rot_dim < head_size)
    for (int idx = threadIdx.x; idx < hk * (head_size - rot_dim); idx += 256) {
      const int h = idx / (head_size - rot_dim), i = rot_dim + idx % (head_size - rot_dim);
      kdst[h * head_size + i] = k[t * k_stride + (int64_t)h * head_size + i];
    }
  const int vvec = hk * head_size / 8;
  for (int i = threadIdx.x; i < vvec; i += 256)
    *(u32x4_t*)(v_buf + slot * vb_stride + i * 8) = *(const u32x4_t*)(v + t * v_stride + i * 8);
}

// one workgroup per token: merge the split partials of every head (decode stage 2), round to T, per-token fp8 quant
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void decode_merge_quant_kernel(const float* attn_logits, const float* attn_lse,
                                                                 const int32_t* kv_indptr, const int64_t* seq_lens,
                                                                 const int32_t* num_kv_splits, int max_kv_splits, int hq,
                                                                 int dv, T* out_o, uint8_t* out_q, float* out_s) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const int seq_len = kv_indptr ? kv_indptr[b + 1] - kv_indptr[b] : (int)seq_lens[b];
  const int nsplit = max(1, min(num_kv_splits[b], max_kv_splits));
  merge_quant_row<T, MAXV>(b, attn_logits, attn_lse, seq_len, nsplit, max_kv_splits, hq, dv, out_o, out_q, out_s, red);
}

// vectorised greedy argmax: first index of the row maximum
template <typename T>
__global__ __launch_bounds__(1024) void argmax_vec_kernel(int64_t* out, const T* logits, int64_t vocab, int64_t stride) {
  __shared__ float rv[16];
  __shared__ int64_t ri[16];
  const T* row = logits + (int64_t)blockIdx.x * stride;
  float best = -INFINITY;
  int64_t bi = 0x7fffffffffffffffLL;
  const int64_t nvec = vocab / 8;
  // four 16-byte pieces requested per round trip (one workgroup per row: the loop is latency-, not bandwidth-bound)
  for (int64_t i0 = threadIdx.x; i0 < nvec; i0 += 4 * 1024) {
    V8<T> a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t i = i0 + q * 1024;
      a[q] = ld8(row + (i < nvec ? i : i0) * 8);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t i = i0 + q * 1024;
      if (i < nvec) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = (float)a[q].v[j];
          if (argmax_beats(v, i * 8 + j, best, bi)) { best = v; bi = i * 8 + j; }
        }
      }
    }
  }
  for (int64_t i = nvec * 8 + threadIdx.x; i < vocab; i += 1024) {
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
    for (int w = 1; w < 16; ++w)
      if (argmax_beats(rv[w], ri[w], best, bi)) { best = rv[w]; bi = ri[w]; }
    out[blockIdx.x] = bi;
  }
}

#define DISPATCH_HALF(dtype, ...) \
  if ((dtype) == SGL_BF16) {      \
    using T = __bf16;             \
    __VA_ARGS__                   \
  } else {                        \
    using T = _Float16;           \
    __VA_ARGS__                   \
  }

}  // namespace

// x (bf16/f16 [tokens, hidden]) XOR slabs (f32 [nslabs, tokens, hidden] raw split-K sums with their sx[m] / sw[n]);
// residual (in/out) may be NULL (first layer); out_norm and out_q/out_s are each optional.
extern "C" int sgl_mi355_fused_add_rmsnorm_quant_fp8(const void* x, const float* slabs, int nslabs, const float* slab_sx,
                                                     const float* slab_sw, void* residual, const void* weight, float eps,
                                                     void* out_norm, void* out_q, float* out_s, int tokens, int hidden,
                                                     int dtype, void* stream) {
  SGL_CHECK(tokens >= 0 && hidden > 0, "fused_add_rmsnorm_quant_fp8: bad shape");
  if (tokens == 0) return SGL_MI355_OK;
  SGL_CHECK((x != nullptr) != (slabs != nullptr), "fused_add_rmsnorm_quant_fp8: exactly one of x / slabs must be given");
  SGL_CHECK(weight && (out_norm || out_q), "fused_add_rmsnorm_quant_fp8: null pointer");
  SGL_CHECK(!out_q || out_s, "fused_add_rmsnorm_quant_fp8: out_q needs out_s");
  SGL_CHECK(hidden % 8 == 0 && hidden <= 8192, "fused_add_rmsnorm_quant_fp8: hidden=%d must be a multiple of 8 and <= 8192", hidden);
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "fused_add_rmsnorm_quant_fp8: dtype must be bf16 or f16");
  SGL_CHECK(!slabs || nslabs >= 1, "fused_add_rmsnorm_quant_fp8: nslabs must be >= 1");
  hipStream_t st = (hipStream_t)stream;
  // prefill-sized launches stream their rows through once: non-temporal row loads / stores.  Measured (tools/debug/norm_time.py): 65 536 x
  // 4096 322 -> 301 us, 65 536 x 8192 670 -> 605 us; 16 384 x 4096 (0.47 GB of traffic: it fits the 256 MB Infinity Cache half-way) 76 -> 80 us
  const bool nt = SGL_ROW_NT && (int64_t)tokens * hidden >= (1ll << 28) && slabs == nullptr;
#define SGL_NORM_LAUNCH_(MV, SL, NTV)                                                                                        \
  hipLaunchKernelGGL((add_rmsnorm_quant_kernel<T, MV, SL, NTV>), dim3(tokens), dim3(256), 0, st, (const T*)x, slabs, nslabs, slab_sx, \
                     slab_sw, (T*)residual, (const T*)weight, eps, (T*)out_norm, (uint8_t*)out_q, out_s, tokens, hidden)
#define SGL_NORM_LAUNCH(MV, SL)               \
  do {                                        \
    if (nt && !(SL)) SGL_NORM_LAUNCH_(MV, false, true); \
    else SGL_NORM_LAUNCH_(MV, SL, false);     \
  } while (0)
  DISPATCH_HALF(dtype, {
    if (slabs != nullptr) {
      if (hidden <= 4096) {
        SGL_NORM_LAUNCH(2, true);
      } else {
        SGL_NORM_LAUNCH(4, true);
      }
    } else if (hidden <= 2048) {
      SGL_NORM_LAUNCH(1, false);
    } else if (hidden <= 4096) {
      SGL_NORM_LAUNCH(2, false);
    } else {
      SGL_NORM_LAUNCH(4, false);
    }
  })
#undef SGL_NORM_LAUNCH_
#undef SGL_NORM_LAUNCH
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

int g_silu_lut = 1;   // measurement hook: 0 = always the exact-expression kernel
extern "C" int sgl_mi355_silu_and_mul_quant_set_mode(int table) {
  g_silu_lut = table ? 1 : 0;
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_silu_and_mul_quant_fp8(const void* x, void* out_q, float* out_s, int tokens, int d, int dtype,
                                                void* stream) {
  SGL_CHECK(tokens >= 0 && d > 0, "silu_and_mul_quant_fp8: bad shape");
  if (tokens == 0) return SGL_MI355_OK;
  SGL_CHECK(x && out_q && out_s, "silu_and_mul_quant_fp8: null pointer");
  SGL_CHECK(d % 8 == 0 && d <= 32768, "silu_and_mul_quant_fp8: d=%d must be a multiple of 8 and <= 32768", d);
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "silu_and_mul_quant_fp8: dtype must be bf16 or f16");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SGL_BF16 && tokens >= 1024 && g_silu_lut) {   // prefill-sized: the table pays for itself from a few rows per workgroup on
    int rpw = tokens / 2048;
    rpw = rpw < 4 ? 4 : (rpw > 32 ? 32 : rpw);
    const unsigned grid = (unsigned)((tokens + rpw - 1) / rpw);
    const bool nt = SGL_ROW_NT && (int64_t)tokens * d >= (1ll << 27);   // (2 d inputs per token: the add + RMSNorm + quant kernel's threshold in bytes)
    if (d <= 16384) {
      if (nt) hipLaunchKernelGGL((silu_mul_quant_lut_kernel<8, true>), dim3(grid), dim3(256), 0, st, (const __bf16*)x, (uint8_t*)out_q, out_s, d, tokens, rpw);
      else hipLaunchKernelGGL((silu_mul_quant_lut_kernel<8>), dim3(grid), dim3(256), 0, st, (const __bf16*)x, (uint8_t*)out_q, out_s, d, tokens, rpw);
    } else {
      if (nt) hipLaunchKernelGGL((silu_mul_quant_lut_kernel<16, true>), dim3(grid), dim3(256), 0, st, (const __bf16*)x, (uint8_t*)out_q, out_s, d, tokens, rpw);
      else hipLaunchKernelGGL((silu_mul_quant_lut_kernel<16>), dim3(grid), dim3(256), 0, st, (const __bf16*)x, (uint8_t*)out_q, out_s, d, tokens, rpw);
    }
    SGL_HIP_LAUNCH_CHECK();
    return SGL_MI355_OK;
  }
  DISPATCH_HALF(dtype, {
    if (d <= 16384)
      hipLaunchKernelGGL((silu_mul_quant_kernel<T, 8>), dim3(tokens), dim3(256), 0, st, (const T*)x, (uint8_t*)out_q, out_s, d);
    else
      hipLaunchKernelGGL((silu_mul_quant_kernel<T, 16>), dim3(tokens), dim3(256), 0, st, (const T*)x, (uint8_t*)out_q, out_s, d);
  })
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_rope_set_kv(const int64_t* positions, void* query, void* key, const void* value,
                                     const float* cos_sin_cache, void* k_buffer, void* v_buffer, const int64_t* loc,
                                     int64_t tokens, int num_q_heads, int num_k_heads, int head_size, int rot_dim,
                                     int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t k_slot_stride,
                                     int64_t v_slot_stride, int is_neox, int dtype, void* stream) {
  SGL_CHECK(tokens >= 0, "rope_set_kv: negative token count");
  if (tokens == 0) return SGL_MI355_OK;
  SGL_CHECK(positions && query && key && value && cos_sin_cache && k_buffer && v_buffer && loc, "rope_set_kv: null pointer");
  SGL_CHECK(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, "rope_set_kv: rot_dim=%d invalid for head_size=%d", rot_dim, head_size);
  SGL_CHECK((num_k_heads * head_size) % 8 == 0 && v_stride % 8 == 0 && v_slot_stride % 8 == 0 &&
                ((uintptr_t)value % 16) == 0 && ((uintptr_t)v_buffer % 16) == 0,
            "rope_set_kv: v rows must be 16-byte aligned");
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "rope_set_kv: dtype must be bf16 or f16");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_HALF(dtype, {
    hipLaunchKernelGGL((rope_set_kv_kernel<T>), dim3((unsigned)tokens), dim3(256), 0, st, positions, (T*)query, (T*)key,
                       (const T*)value, cos_sin_cache, (T*)k_buffer, (T*)v_buffer, loc, num_q_heads, num_k_heads, head_size,
                       rot_dim, q_stride, k_stride, v_stride, k_slot_stride, v_slot_stride, is_neox);
  })
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// kv_indptr (int32 [batch+1]) or seq_lens (int64 [batch]) gives each request's length (needed to skip empty splits)
extern "C" int sgl_mi355_decode_merge_quant_fp8(const float* attn_logits, const float* attn_lse, const int32_t* kv_indptr,
                                                const int64_t* seq_lens, const int32_t* num_kv_splits, int max_kv_splits,
                                                int batch, int num_q_heads, int v_head_dim, void* out_o, void* out_q,
                                                float* out_s, int dtype, void* stream) {
  SGL_CHECK(batch >= 0, "decode_merge_quant_fp8: negative batch");
  if (batch == 0) return SGL_MI355_OK;
  SGL_CHECK(attn_logits && attn_lse && num_kv_splits && (kv_indptr || seq_lens) && (out_o || out_q),
            "decode_merge_quant_fp8: null pointer");
  SGL_CHECK(!out_q || out_s, "decode_merge_quant_fp8: out_q needs out_s");
  const int row = num_q_heads * v_head_dim;
  SGL_CHECK(v_head_dim % 8 == 0 && row <= 16384, "decode_merge_quant_fp8: Hq*Dv=%d must be <= 16384 and Dv a multiple of 8", row);
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "decode_merge_quant_fp8: dtype must be bf16 or f16");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_HALF(dtype, {
    if (row <= 8192)
      hipLaunchKernelGGL((decode_merge_quant_kernel<T, 4>), dim3(batch), dim3(256), 0, st, attn_logits, attn_lse, kv_indptr,
                         seq_lens, num_kv_splits, max_kv_splits, num_q_heads, v_head_dim, (T*)out_o, (uint8_t*)out_q, out_s);
    else
      hipLaunchKernelGGL((decode_merge_quant_kernel<T, 8>), dim3(batch), dim3(256), 0, st, attn_logits, attn_lse, kv_indptr,
                         seq_lens, num_kv_splits, max_kv_splits, num_q_heads, v_head_dim, (T*)out_o, (uint8_t*)out_q, out_s);
  })
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_argmax_vec(int64_t* out, const void* logits, int64_t rows, int64_t vocab, int64_t row_stride,
                                    int dtype, void* stream) {
  SGL_CHECK(rows >= 0 && vocab > 0, "argmax: bad shape");
  if (rows == 0) return SGL_MI355_OK;
  SGL_CHECK(out && logits, "argmax: null pointer");
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "argmax_vec: dtype must be bf16 or f16");
  SGL_CHECK(row_stride % 8 == 0 && ((uintptr_t)logits % 16) == 0, "argmax_vec: rows must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_HALF(dtype, {
    hipLaunchKernelGGL((argmax_vec_kernel<T>), dim3((unsigned)rows), dim3(1024), 0, st, out, (const T*)logits, vocab, row_stride);
  })
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}
