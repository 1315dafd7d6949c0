// Compiler trap (ROCm 7.2 hipcc, gfx950; round 4): the f16 halves of the SECOND word of a two-word vector.
//   u32x2_t v = <8-byte load>;  c1 = bit_cast<f16x2_t>(v[1]);  splat(c1[0]), splat(c1[1])
// compiles to reads of word 0's halves (with a buffer load the 8-byte load is even narrowed to one dword): wrong code, silently.
// With the two words held as two uint32_t scalars the same expressions compile correctly, and adjacent scalar loads are merged into
// one 8-byte load later.  csrc/awq.hip (AwqDequant<_Float16>::setup) therefore takes two scalars; test_awq_gemm_vs_oracle caught
// the first version.  Reproduce:
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o - buffer_b64_halves.hip | grep -E "buffer_load|global_load|v_pk"
//   k_vector: buffer_load_dword (not dwordx2); every v_pk_add / v_pk_mul constant comes from the same register
//   k_scalar: the adds take their constants from the second loaded word
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

__global__ void k_vector(const uint32_t* sz, f16x2_t* out, int n) {
  const auto r = __builtin_amdgcn_make_buffer_rsrc((void*)sz, 0, (unsigned)n, 0x00020000);
  const u32x2_t v = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(r, 8u * threadIdx.x, 0, 0));
  const f16x2_t c0 = __builtin_bit_cast(f16x2_t, v[0]), c1 = __builtin_bit_cast(f16x2_t, v[1]);
  const f16x2_t s2 = f16x2_t{c0[0], c0[0]}, nz0 = f16x2_t{c1[0], c1[0]}, nz1 = f16x2_t{c1[1], c1[1]};
  const f16x2_t x = out[threadIdx.x];
  out[threadIdx.x] = (x + nz0) * s2 + (x + nz1) * s2;
}

__global__ void k_scalar(const uint32_t* sz, f16x2_t* out, int n) {
  const uint32_t w0 = sz[2 * threadIdx.x], w1 = sz[2 * threadIdx.x + 1];
  const f16x2_t c0 = __builtin_bit_cast(f16x2_t, w0), c1 = __builtin_bit_cast(f16x2_t, w1);
  const f16x2_t s2 = f16x2_t{c0[0], c0[0]}, nz0 = f16x2_t{c1[0], c1[0]}, nz1 = f16x2_t{c1[1], c1[1]};
  const f16x2_t x = out[threadIdx.x];
  out[threadIdx.x] = (x + nz0) * s2 + (x + nz1) * s2;
}
