// Shared device/host helpers for the MI355X (gfx950) hot-path kernels.
// Everything here is written for CDNA4 only: 64-lane wavefronts, MFMA 16x16x32,
// OCP e4m3fn fp8.  There is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

// ---- C-ABI status / error convention (include/sgl_mi355.h) -------------------
// The reference reports errors with TORCH_CHECK -> RuntimeError (message only,
// sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1078-1108).  The C-ABI returns an int
// status and keeps the message in a thread-local buffer; the Python shim raises
// RuntimeError(sgl_mi355_last_error()).
#define SGL_MI355_OK 0
#define SGL_MI355_EINVAL 1
#define SGL_MI355_EHIP 2

extern thread_local char g_sgl_mi355_err[512];

#define SGL_CHECK(cond, ...)                                           \
  do {                                                                 \
    if (!(cond)) {                                                     \
      snprintf(g_sgl_mi355_err, sizeof(g_sgl_mi355_err), __VA_ARGS__); \
      return SGL_MI355_EINVAL;                                         \
    }                                                                  \
  } while (0)

#define SGL_HIP_LAUNCH_CHECK()                                                         \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) {                                                           \
      snprintf(g_sgl_mi355_err, sizeof(g_sgl_mi355_err), "HIP launch failed: %s (%s:%d)", \
               hipGetErrorString(e__), __FILE__, __LINE__);                            \
      return SGL_MI355_EHIP;                                                           \
    }                                                                                  \
  } while (0)

// dtype codes shared with the Python shim
enum SglDtype : int { SGL_BF16 = 0, SGL_F16 = 1, SGL_F32 = 2, SGL_FP8_E4M3 = 3 };

// ---- vector types -------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define WAVE 64

template <typename T>
struct ElemTraits;

template <>
struct ElemTraits<__bf16> {
  typedef bf16x8_t vec8;
  typedef bf16x4_t vec4;
  static __device__ __forceinline__ f32x4_t mfma16(vec8 a, vec8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16_t mfma32(vec8 a, vec8 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f32(__bf16 x) { return (float)x; }
  static __device__ __forceinline__ __bf16 from_f32(float x) { return (__bf16)x; }
};

template <>
struct ElemTraits<_Float16> {
  typedef f16x8_t vec8;
  typedef f16x4_t vec4;
  static __device__ __forceinline__ f32x4_t mfma16(vec8 a, vec8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16_t mfma32(vec8 a, vec8 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f32(_Float16 x) { return (float)x; }
  static __device__ __forceinline__ _Float16 from_f32(float x) { return (_Float16)x; }
};

// Combine a value over the lane pairs (i, i ^ 16) / (i, i ^ 32) with two VALU lane swaps instead of an LDS round trip
// (ds_bpermute, what __shfl_xor compiles to for these distances): v_permlane16_swap(x, x) leaves the even 16-lane rows in one result
// and the odd rows in the other, v_permlane32_swap the two 32-lane halves -- op(first, second) is then op(own, partner) for a
// commutative op, bit for bit (round 3: these sit in the middle of the attention kernels' softmax chains).
__device__ __forceinline__ float pair16_max(float v) {
  const uint32_t b = __builtin_bit_cast(uint32_t, v);
  const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
  return fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
}
__device__ __forceinline__ float pair32_max(float v) {
  const uint32_t b = __builtin_bit_cast(uint32_t, v);
  const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
  return fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
}
__device__ __forceinline__ float pair16_sum(float v) {
  const uint32_t b = __builtin_bit_cast(uint32_t, v);
  const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
  return __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
}
__device__ __forceinline__ float pair32_sum(float v) {
  const uint32_t b = __builtin_bit_cast(uint32_t, v);
  const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
  return __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
}

// wave-wide xor shuffle (ds_bpermute / DPP chosen by the compiler)
__device__ __forceinline__ float wave_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, WAVE); }

// Wave-wide reductions without LDS round trips (round 3): hipcc compiles every __shfl_xor step to a ds_bpermute, six dependent LDS
// round trips per reduction -- a tenth of a decode-sized row kernel's lifetime.  Distances 32 and 16 by lane swaps (above), 8, 4, 2, 1
// by DPP row rotations (within a 16-lane row the partial sums have period 2 x distance, so rotating by the distance reaches the same
// partner values as the xor butterfly): the same pairs are combined in the same order, so max and sum give the same bits as before.
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// value of lane i ^ 8 / i ^ 4 by DPP (one / two VALU moves) instead of a ds_bpermute round trip: within a 16-lane row, i ^ 8 is a
// rotation by 8; i ^ 4 is the half-row mirror (i -> 7 - i within 8 lanes) followed by the quad mirror (j -> 3 - j within 4 lanes)
__device__ __forceinline__ float lane_xor8(float v) { return dpp_row<0x128>(v); }
__device__ __forceinline__ float lane_xor4(float v) { return dpp_row<0x1B>(dpp_row<0x141>(v)); }   // quad_perm [3,2,1,0] after row_half_mirror

__device__ __forceinline__ float wave_reduce_max(float v) {
  v = pair16_max(pair32_max(v));
  v = fmaxf(v, dpp_row<0x128>(v));   // row_ror:8
  v = fmaxf(v, dpp_row<0x124>(v));   // row_ror:4
  v = fmaxf(v, dpp_row<0x122>(v));   // row_ror:2
  v = fmaxf(v, dpp_row<0x121>(v));   // row_ror:1
  return v;
}
__device__ __forceinline__ float wave_reduce_sum(float v) {
  v = pair16_sum(pair32_sum(v));
  v += dpp_row<0x128>(v);
  v += dpp_row<0x124>(v);
  v += dpp_row<0x122>(v);
  v += dpp_row<0x121>(v);
  return v;
}

// One step of the LSE-weighted merge of split-KV partials (decode_attention.py:492-552), written with explicit fmaf so
// that every kernel that merges (stand-alone stage 2, fused merge+quant) rounds identically: left to the compiler,
// a*b + c*d contracts differently from kernel to kernel and flips the last f32 bit now and then.
struct LseMerge {
  float M = -INFINITY, L = 0.f, so = 0.f, sn = 0.f;
  __device__ __forceinline__ void begin(float lse) {
    const float nM = fmaxf(M, lse);
    so = __expf(M - nM);
    sn = __expf(lse - nM);
    L = fmaf(L, so, sn);
    M = nM;
  }
  __device__ __forceinline__ float acc(float a, float v) const { return fmaf(a, so, v * sn); }
  __device__ __forceinline__ float finish(float a) const { return L > 0.f ? a / L : 0.f; }
};

// ---- fp8 (e4m3fn) KV cache conversion (memory_pool.py:385-395): cache.div_(scale) in the source dtype when a scale is given
// (scale <= 0: none), then .to(float8_e4m3fn) = round to nearest even, and what torch turns into NaN (|x| > 464, inf, NaN ->
// 0x7F | sign; c10/util/Float8_e4m3fn.h) instead of the hardware's saturation to 448.  Shared by set_kv_buffer_fp8 and the
// GEMM epilogues that write the pool, so both produce the same bytes.
template <typename T>
__device__ __forceinline__ float kv_fp8_scaled(float v, float scale) {
  if (scale > 0.0f) {  // div_ rounds to the tensor dtype before the cast (forced through the bit pattern: clang keeps 16-bit
    uint32_t bits = __builtin_bit_cast(uint16_t, (T)(v / scale));  // float expressions in excess precision)
    asm volatile("" : "+v"(bits));
    v = (float)__builtin_bit_cast(T, (uint16_t)bits);
  }
  return v;
}
__device__ __forceinline__ bool kv_fp8_is_nan(float f) { return !(fabsf(f) <= 464.0f); }
__device__ __forceinline__ uint32_t kv_fp8_nan_byte(float f) { return 0x7Fu | ((__builtin_bit_cast(uint32_t, f) >> 24) & 0x80u); }
// one value (already a T) -> its pool byte
template <typename T>
__device__ __forceinline__ uint8_t kv_fp8_byte(float v, float scale) {
  const float f = kv_fp8_scaled<T>(v, scale);
  const uint32_t w = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f, f, 0, false);
  return (uint8_t)(kv_fp8_is_nan(f) ? kv_fp8_nan_byte(f) : (w & 0xFFu));
}

static inline int cdiv_i(int a, int b) { return (a + b - 1) / b; }

// torch.argmax's order: NaN is the maximum and the FIRST index among equals wins, so a row of NaNs (or of -inf) has a defined
// answer.  (Round 4: the kernels compared with `v > best` only and returned INT64_MAX for such a row -- the embedding lookup of
// the next decode step then read ~2^63 rows past its table: a GPU memory fault.)
__device__ __forceinline__ bool argmax_beats(float v, int64_t i, float best, int64_t bi) {
  const bool vn = v != v, bn = best != best;
  if (vn || bn) return vn && (!bn || i < bi);
  return v > best || (v == best && i < bi);
}
