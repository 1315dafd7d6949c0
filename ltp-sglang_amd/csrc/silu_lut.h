// T(silu(a)) for bf16 a as a table over a's 16 bits (round 3).  The exact expression -- expf + IEEE division, the rounding points of
// SiluAndMul.forward_native (activation.py:60-63) -- is ~25 VALU instructions per element; the value is a function of 16 bits, so
// a kernel that evaluates it millions of times tabulates it once WITH THAT EXPRESSION (silu_lut_entry: the table cannot differ
// from it) and looks it up: 2 signs x 37 exponents (2^-30 <= |a| < 128) x 128 mantissas = 9 472 entries = 18.5 KiB.  Everything
// outside (zeros, denormals, tiny or huge values, inf, nan) is left to the exact expression by the caller (silu_lut_rel's range test).
#pragma once
#include "row_helpers.h"

namespace {

constexpr int kSiluE0 = 97, kSiluNE = 37;
constexpr int kSiluHalf = kSiluNE * 128, kSiluLut = 2 * kSiluHalf;

__device__ __forceinline__ float silu_exact_bf16(float af) { return round_via<__bf16>(af / (1.0f + expf(-af))); }

// bits of T(silu(a)) for table entry i
__device__ __forceinline__ uint16_t silu_lut_entry(int i) {
  const int sgn = i >= kSiluHalf, r = i - sgn * kSiluHalf;
  const uint16_t bits = (uint16_t)((sgn << 15) | (((r >> 7) + kSiluE0) << 7) | (r & 127));
  const float af = (float)__builtin_bit_cast(__bf16, bits);
  return __builtin_bit_cast(uint16_t, (__bf16)silu_exact_bf16(af));
}

// table index of the bf16 bits `ab`, or a value >= kSiluLut when a is outside the table
__device__ __forceinline__ uint32_t silu_lut_index(uint32_t ab) {
  const uint32_t rel = (ab & 0x7FFFu) - (uint32_t)(kSiluE0 << 7);   // wraps to a huge value below the table
  return rel < (uint32_t)kSiluHalf ? rel + (ab >> 15) * kSiluHalf : 0xFFFFFFFFu;
}

}  // namespace
