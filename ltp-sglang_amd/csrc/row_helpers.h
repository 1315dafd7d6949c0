// Row helpers shared by the fused elementwise kernels (fused_decode.hip) and the in-launch split merge of the decode
// attention kernel (decode_attention.hip): 16-byte row access, the forced f32 -> T -> f32 rounding point, per-token fp8
// quantisation of a row held in registers, and the stage-2 LSE merge of one request's split partials.
#pragma once
#include "common.h"

namespace {

constexpr float kFp8Max = 448.0f;

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
// non-temporal forms for prefill-sized row kernels: a row is read once and its results are read once, gigabytes later
template <typename T>
__device__ __forceinline__ V8<T> ld8_nt(const T* p) {
  return __builtin_bit_cast(V8<T>, __builtin_nontemporal_load((const u32x4_t*)p));
}
template <typename T>
__device__ __forceinline__ void st8_nt(T* p, const V8<T>& x) {
  __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, x), (u32x4_t*)p);
}

// f32 -> T -> f32 through the bit pattern (see elementwise.hip: the rounding point must really happen)
template <typename T>
__device__ __forceinline__ float round_via(float x) {
  asm volatile("" : "+v"(x));  // x must exist as an f32 first: no v_fma_mix* single-rounding shortcut (the reference rounds twice)
  const T t = (T)x;
  const uint16_t u = __builtin_bit_cast(uint16_t, t);
  uint16_t v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(u));
  return (float)__builtin_bit_cast(T, v);
}

__device__ __forceinline__ u32x2_t pack8_fp8(const float (&f)[8]) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  return u32x2_t{(uint32_t)lo, (uint32_t)hi};
}

__device__ __forceinline__ float clamp448(float v) { return fmaxf(fminf(v, kFp8Max), -kFp8Max); }

// (block_sum / block_max / quant_row / merge_quant_row: the FIRST 256 threads of the workgroup do the work -- red: 4 floats -- and any
// further waves (the decode kernel's 512-thread form) only pass the barriers: round 4 let them recompute and re-store the rows of
// waves 0-3, correct by accident and twice the merge traffic on the launch's critical path)
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_reduce_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_reduce_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// per-token fp8 quantisation of the row held in vals (already rounded to T), exactly per_token_quant_fp8.cu
template <int MAXV, bool NT = false>
__device__ __forceinline__ void quant_row(const float (&vals)[MAXV][8], int nvec, uint8_t* qrow, float* srow, float* red) {
  const bool on = threadIdx.x < 256;
  float amax = 0.f;
#pragma unroll
  for (int it = 0; it < MAXV; ++it)
    if (on && (int)threadIdx.x + it * 256 < nvec)
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(vals[it][j]));
  amax = block_max(amax, red);
  const float scale = amax / kFp8Max;
  if (threadIdx.x == 0) *srow = scale;
  const float inv = (scale == 0.f) ? 0.f : 1.0f / scale;
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int i = threadIdx.x + it * 256;
    if (on && i < nvec) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = clamp448(vals[it][j] * inv);
      if constexpr (NT) __builtin_nontemporal_store(pack8_fp8(f), (u32x2_t*)(qrow + i * 8));
      else *(u32x2_t*)(qrow + i * 8) = pack8_fp8(f);
    }
  }
}


// One workgroup (its first 256 threads): merge the split partials of every head of request b (decode stage 2,
// decode_attention.py:492-552), round to T, optionally store, optionally per-token fp8 quantise.  `red`: 4 floats of LDS.
template <typename T, int MAXV>
__device__ __forceinline__ void merge_quant_row(int b, const float* attn_logits, const float* attn_lse, int seq_len, int nsplit,
                                                int max_kv_splits, int hq, int dv, T* out_o, uint8_t* out_q, float* out_s,
                                                float* red, bool split0_always = false) {
  const int per0 = (seq_len + nsplit - 1) / nsplit;
  const int per = (per0 + 31) / 32 * 32;
  const int total = nsplit;
  // split0_always: the cascade decode's split 0 carries the shared-prefix state even when the request's private part is empty
  auto live = [&](int sI) -> bool { return sI < total && (sI * per < seq_len || (sI == 0 && split0_always)); };
  const int row_elems = hq * dv, nvec = row_elems / 8;
  float vals[MAXV][8];
  const bool on = threadIdx.x < 256;
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int i = threadIdx.x + it * 256;
    if (on && i < nvec) {
      const int e0 = i * 8, h = e0 / dv, d0 = e0 - h * dv;
      const int64_t slot0 = ((int64_t)b * hq + h) * max_kv_splits;
      LseMerge mg;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      // the first four splits' partials are all requested before the first is consumed (inside the attention launch this
      // merge is the tail of the last workgroup of the request: one L2 round trip instead of one per split); same order of
      // arithmetic as the plain loop
      constexpr int PF = 4;
      float lse_pf[PF];
      f32x4_t a0_pf[PF], a1_pf[PF];
#pragma unroll
      for (int sI = 0; sI < PF; ++sI) {
        const int sc = sI < total ? sI : 0;  // clamped: a split that does not exist re-reads split 0 (never used)
        lse_pf[sI] = attn_lse[slot0 + sc];
        const float* lp = attn_logits + (slot0 + sc) * dv + d0;
        a0_pf[sI] = *(const f32x4_t*)lp;
        a1_pf[sI] = *(const f32x4_t*)(lp + 4);
      }
#pragma unroll
      for (int sI = 0; sI < PF; ++sI) {
        if (live(sI)) {
          mg.begin(lse_pf[sI]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[j] = mg.acc(acc[j], a0_pf[sI][j]);
            acc[4 + j] = mg.acc(acc[4 + j], a1_pf[sI][j]);
          }
        }
      }
      for (int sI = PF; sI < total; ++sI) {
        if (live(sI)) {
          mg.begin(attn_lse[slot0 + sI]);
          const float* lp = attn_logits + (slot0 + sI) * dv + d0;
          const f32x4_t a0 = *(const f32x4_t*)lp, a1 = *(const f32x4_t*)(lp + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[j] = mg.acc(acc[j], a0[j]);
            acc[4 + j] = mg.acc(acc[4 + j], a1[j]);
          }
        }
      }
      V8<T> o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        vals[it][j] = round_via<T>(mg.finish(acc[j]));
        o.v[j] = (T)vals[it][j];
      }
      if (out_o) st8(out_o + (int64_t)b * row_elems + e0, o);
    }
  }
  if (out_q) quant_row<MAXV>(vals, nvec, out_q + (int64_t)b * row_elems, out_s + b, red);
}

}  // namespace
