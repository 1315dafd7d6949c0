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
#include "gemm_epilogue.h"

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


// ---------------------------------------------------------------------------------------------------------
// Fused int4 dequant + skinny GEMM for decode-sized M (<= 64):  y[m][n] = sum_k x[m][k] * W[k][n] (+ bias), with
// W[k][n] EXACTLY the value awq_dequantize produces (the dequantisation happens in registers, between the global
// load and the MFMA; nothing but int4 weights is ever read from HBM).
//
// Replaces the reference's AWQLinearMethod.apply = awq_dequantize + torch.matmul
// (python/sglang/srt/layers/quantization/awq.py:401-418), which re-materialises the whole bf16 weight every call
// (0.5 B read + 2 B written + 2 B read per parameter; fused: 0.5 B).
//
// The weight is re-laid once at load time (the role awq_marlin_repack plays for the reference's Marlin path,
// sgl-kernel/csrc/gemm/marlin/awq_marlin_repack.cu) into MFMA-fragment order, so that a wave-wide 16-byte load IS the
// B operand of four v_mfma_f32_16x16x32 k-steps:
//   qpacked int32 [N/16][K/128][64 lanes][4]:  word s of lane (a = n % 16, g) of tile n/16, block k/128 holds
//   q[k = 128 b + 32 s + 8 g + e][n], e = 0..7, in nibble (e & 1) * 4 + e / 2, so (word >> 4 i) & 0x000F000F is the pair
//   (e = 2 i, 2 i + 1) in the two 16-bit halves     (1 KiB contiguous per (tile, block): perfectly coalesced)
//   sz     int32 [K/G][N]:  scale bits | (zero point << 16) for bf16 scales; scale bits | (0xE400 | zero point) << 16 for f16
//          scales -- the upper half is then the f16 constant -(1024 + z) the dequantisation subtracts, ready made (round 4)
//          (one 4-byte load per lane, block and scale group)
// Structure = the X-stationary skinny GEMM of skinny_gemm.hip without the LDS re-layout: one 512-thread workgroup per
// CU; wave w owns k-blocks w, w+8, w+16, w+24 of the k-range (blockIdx.y: 32 blocks = 4096 k) for the whole launch and
// keeps their X fragments in registers; the workgroup walks 16-column tiles; the 8 waves' partial sums of TPP tiles
// meet in LDS between two barriers.  k-ranges > 1 (K > 4096) go through f32 slabs and a reduce kernel.  33..64 rows (MT = 4):
// two k-blocks per wave (k-range 2048), so the X fragments stay at 128 VGPRs; two output rows per thread in the epilogue.
// Dequantisation, bit-identical to awq_dequantize's T(float(q - z) * float(s)):
//   f16:  even nibble pairs: (word & 0x000F000F) | 0x64006400 = (1024 + q) as packed f16; odd pairs IN PLACE: (word & 0x00F000F0) |
//         0x54005400 = (64 + q) -- mantissa bits 4..7 of 64.0 weigh 1, 2, 4, 8 -- so ONE shift (by 8) serves the four pairs of a
//         word instead of three (r4; always exact, no condition on the scales); v_pk_add_f16 with -(1024 + z) / -(64 + z) gives
//         q - z exactly; v_pk_mul_f16 by the scale rounds once.  The scale and -(1024 + z) are the two halves of the sz word
//         (op_sel, no instruction), -(64 + z) takes two: 13 VALU lane-ops per 8 weights + 2 per scale group (15 + 8 before);
//   bf16: v_cvt_f32_ubyteN of the nibbles, v_pk_fma_f32 (q * s - z * s is exact in f32), v_cvt_pk_bf16_f32  ->  2.4 per weight.
// ---------------------------------------------------------------------------------------------------------
struct AwqGemmParams {
  const char* x;            // [M][K] T
  int64_t x_stride;         // elements
  const uint32_t* qpacked;  // [N/16][KB][64][4]
  const uint32_t* sz;       // [K/G][N]
  void* y;
  int64_t y_stride;
  const void* bias;
  int M, N, K, G, KB;  // KB = K / 128
#ifdef SGL_AWQ_TIMELINE
  long long* tl;  // tools/microbench/awq_timeline.hip: s_memtime stamps [workgroup][wave][16]
#endif
};

#ifdef SGL_AWQ_TIMELINE
#define AWQ_STAMP(i)                                                                            \
  do {                                                                                          \
    const long long t_ = (long long)__builtin_amdgcn_s_memtime();                               \
    if (lane == 0 && (i) < 16) p.tl[((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + w) * 16 + (i)] = t_; \
  } while (0)
#else
#define AWQ_STAMP(i)
#endif


#ifndef SGL_AWQ_EXACT_WEIGHTS
#define SGL_AWQ_EXACT_WEIGHTS 0  // 1 (tools/build_variant.sh): the round-2 per-weight form everywhere (awq_dequantize's rounded weights)
#endif
template <int B>
__device__ __forceinline__ float cvt_ubyte(uint32_t x) {  // float(byte B of x): one VALU op, no shift / mask
  float f;
  if constexpr (B == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(x));
  else if constexpr (B == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(x));
  else if constexpr (B == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(x));
  else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(x));
  return f;
}

template <typename T>
struct AwqDequant;
// (szw: zero << 16 | scale bits)  ->  per-lane constants of one scale group, then 8 weights of one packed word
template <>
struct AwqDequant<_Float16> {
  f16x2_t s2, nz0, nz1;   // scale, -(1024 + z), -(64 + z) in both halves
  // (one uint32_t, never an element of a u32x2_t: `bit_cast<f16x2_t>(v[1])[0]` of a two-word vector compiles to word 0's half with
  // ROCm 7.2's hipcc -- tools/microbench/buffer_b64_halves.hip)
  __device__ __forceinline__ void setup(uint32_t szw) {
    const f16x2_t c = __builtin_bit_cast(f16x2_t, szw);
    s2 = f16x2_t{c[0], c[0]};      // op_sel on the sz word: no instruction
    nz0 = f16x2_t{c[1], c[1]};     // 0xE400 | z
    const f16x2_t d = __builtin_bit_cast(f16x2_t, ((szw >> 12) & 0xF0u) | 0xD400u);   // -(64 + z): z in mantissa bits 4..7
    nz1 = f16x2_t{d[0], d[0]};
  }
  __device__ __forceinline__ f16x8_t run(uint32_t wq) const {
    f16x2_t o[4];
    const uint32_t w8 = wq >> 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t pr;  // i even: (1024 + q[2i], 1024 + q[2i+1]); i odd: (64 + q[2i], 64 + q[2i+1]); one v_and_or_b32 each
      if (i & 1) asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(pr) : "v"(i < 2 ? wq : w8), "s"(0x00F000F0u), "v"(0x54005400u));
      else asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(pr) : "v"(i < 2 ? wq : w8), "s"(0x000F000Fu), "v"(0x64006400u));
      const f16x2_t d = __builtin_bit_cast(f16x2_t, pr) + ((i & 1) ? nz1 : nz0);   // exact small integers
      o[i] = d * s2;                                                              // one rounding: T(d * s)
    }
    struct P { f16x2_t v[4]; } pk{{o[0], o[1], o[2], o[3]}};
    return __builtin_bit_cast(f16x8_t, pk);
  }
};
template <>
struct AwqDequant<__bf16> {
  float sf, nzs;
  __device__ __forceinline__ void setup(uint32_t szw) {
    sf = (float)__builtin_bit_cast(__bf16, (uint16_t)(szw & 0xFFFFu));
    nzs = -(float)(szw >> 16) * sf;  // exact: 4-bit zero point x 8-bit scale
  }
  __device__ __forceinline__ bf16x8_t run(uint32_t wq) const {
    const uint32_t lo = wq & 0x0F0F0F0Fu, hi = (wq >> 4) & 0x0F0F0F0Fu;  // nibbles 0,2,4,6 = e 0,4,1,5 / 1,3,5,7 = e 2,6,3,7
    const float e0 = cvt_ubyte<0>(lo), e4 = cvt_ubyte<1>(lo);
    const float e1 = cvt_ubyte<2>(lo), e5 = cvt_ubyte<3>(lo);
    const float e2 = cvt_ubyte<0>(hi), e6 = cvt_ubyte<1>(hi);
    const float e3 = cvt_ubyte<2>(hi), e7 = cvt_ubyte<3>(hi);
    struct P { __bf16 v[8]; } pk;
    pk.v[0] = (__bf16)fmaf(e0, sf, nzs); pk.v[1] = (__bf16)fmaf(e1, sf, nzs);
    pk.v[2] = (__bf16)fmaf(e2, sf, nzs); pk.v[3] = (__bf16)fmaf(e3, sf, nzs);
    pk.v[4] = (__bf16)fmaf(e4, sf, nzs); pk.v[5] = (__bf16)fmaf(e5, sf, nzs);
    pk.v[6] = (__bf16)fmaf(e6, sf, nzs); pk.v[7] = (__bf16)fmaf(e7, sf, nzs);
    return __builtin_bit_cast(bf16x8_t, pk);
  }
};

// (round 3) The offset form, used for bf16 operands when one scale group covers a whole 128-k block (SG = 1, G % 128 == 0): the MFMA multiplies X
// by the RAW nibbles as the floating-point integers 16 + q (one shift + one v_and_or_b32 per two weights: the nibble lands in
// the mantissa bits of weight 1 of the constant 16.0), accumulates the block in a temporary f32 tile, and the zero point and the
// scale are applied to that tile:
//     sum_k x[m,k] (q[k,n] - z) s  =  s * ( sum_k x[m,k] (16 + q[k,n])  -  (16 + z) * sum_k x[m,k] )
// with the row sums of X per block (X is stationary: computed once per workgroup by an MFMA against a tile of ones).  8 VALU
// lane-ops per 8 weights + 2 fma per output element and block, against 15 (f16) / 20 (bf16) for the bit-exact per-weight form,
// which was bound by VALU issue ("int4 dequant GEMM").  The weights this form multiplies by are the EXACT
// (q - z) * s, not awq_dequantize's values rounded to the scale dtype: closer to exact arithmetic than the reference, no longer
// bit-identical to dequantise + matmul (parity: the float64 product within the GEMM tolerance; awq_dequantize itself stays
// bit-exact).  16 rather than the classic 1024 (f16) / 128 (bf16) magic: the f32 tile then holds ~24 x sum|x| instead of ~1040 x,
// so the subtraction loses one bit, not seven.
// (No inline asm here: the outputs feed an MFMA directly, and hipcc does not pad the VALU-write -> MFMA-read hazard for an
// instruction hidden in an asm statement -- the first version returned NaNs.  The two constants are laundered through empty asm
// statements once per kernel so that they live in an SGPR / a VGPR and `(x & mask) | magic` can become ONE v_and_or_b32.)
template <typename T>
struct AwqRaw;
template <>
struct AwqRaw<_Float16> {  // 16.0 = 0x4C00: mantissa bits 6..9 have weights 1, 2, 4, 8
  uint32_t mask, magic;
  __device__ __forceinline__ void init() {
    mask = 0x03C003C0u;
    magic = 0x4C004C00u;
    asm volatile("" : "+s"(mask));
    asm volatile("" : "+v"(magic));
  }
  __device__ __forceinline__ f16x8_t run(uint32_t wq) const {
    return __builtin_bit_cast(f16x8_t, u32x4_t{((wq << 6) & mask) | magic, ((wq << 2) & mask) | magic, ((wq >> 2) & mask) | magic,
                                               ((wq >> 6) & mask) | magic});
  }
  static __device__ __forceinline__ f16x8_t ones() { return __builtin_bit_cast(f16x8_t, u32x4_t{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u}); }
};
template <>
struct AwqRaw<__bf16> {  // 16.0 = 0x4180: mantissa bits 3..6 have weights 1, 2, 4, 8
  uint32_t mask, magic;
  __device__ __forceinline__ void init() {
    mask = 0x00780078u;
    magic = 0x41804180u;
    asm volatile("" : "+s"(mask));
    asm volatile("" : "+v"(magic));
  }
  __device__ __forceinline__ bf16x8_t run(uint32_t wq) const {
    return __builtin_bit_cast(bf16x8_t, u32x4_t{((wq << 3) & mask) | magic, ((wq >> 1) & mask) | magic, ((wq >> 5) & mask) | magic,
                                                ((wq >> 9) & mask) | magic});
  }
  static __device__ __forceinline__ bf16x8_t ones() { return __builtin_bit_cast(bf16x8_t, u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}); }
};
template <typename T>
__device__ __forceinline__ float awq_scale_f32(uint32_t szw) { return (float)__builtin_bit_cast(T, (uint16_t)(szw & 0xFFFFu)); }

constexpr int kAwqWaves = 8;
// k-blocks (128 k each) per wave and k-range: 4 (a k-range = 4096 k) while the X fragments of 32 rows fit the register budget,
// 2 (k-range 2048) for 33..64 rows (MT = 4): 128 VGPRs of X fragments either way
constexpr int awq_bpw(int mt) { return mt >= 4 ? 2 : 4; }

// SG = scale groups per 128-k block (1: G % 128 == 0, 2: G = 64, 4: G = 32); EPI: gemm_epilogue.h (single k-range, columns
// interleaved at repack time so that the two values an output needs are 8 columns apart in one 16-column tile)
// EXACTW (round 4, a RUN-TIME choice: sgl_mi355_awq_set_exact_weights / SGL_MI355_AWQ_EXACT_WEIGHTS=1): the per-weight form for
// bf16 too -- the weights awq_dequantize produces, rounded to bf16, i.e. bit-compatible with the reference's
// awq_dequantize -> matmul (awq.py:401-418) for parity runs; the default offset form multiplies by the exact (q - z) s.
template <typename T, int MT, int PD, int TPP, int SG, int EPI = EPI_NONE, bool EXACTW = false>
__global__ __launch_bounds__(kAwqWaves * 64, 1) void awq_gemm_kernel(const AwqGemmParams p, int ntiles, float* slabs,
                                                                     const EpiParams ep = EpiParams{}) {
  typedef ElemTraits<T> Tr;
  typedef typename Tr::vec8 vec8;
  static_assert(TPP == 1 || TPP % PD == 0, "static slot indices");
  constexpr int kAwqBpw = awq_bpw(MT);
  constexpr int EPT = (MT * 16 + 31) / 32;  // output rows per thread and tile (512 threads cover 32 rows x 16 columns)
  static_assert(EPI == EPI_NONE || MT <= 2, "the fused epilogues keep one output row per thread");
  __shared__ float red[TPP][kAwqWaves][MT * 16][16];
  __shared__ __attribute__((aligned(16))) char ximg[kAwqWaves * 4096];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = lane & 15, g = lane >> 4;
  const int kr = blockIdx.y;
  const int b0 = kr * (kAwqWaves * kAwqBpw);
  const int nb = min(kAwqWaves * kAwqBpw, p.KB - b0);  // k-blocks of this range
  const int G_ = gridDim.x;
  const int cnt = (ntiles - (int)blockIdx.x + G_ - 1) / G_;  // tiles of this workgroup: blockIdx.x + j * G_
  AWQ_STAMP(0);  // kernel entry

  // ---- X fragments of this wave's k-blocks, once: coalesced 256-byte row pieces -> swizzled wave-private LDS image ->
  // A-operand registers (lane (a, g) holds X[m = 16 mt + a][128 b + 32 s + 8 g .. + 7]) ----
  char* xl = ximg + w * 4096;                 // 16 rows x 256 B
  const int lc = lane & 15, lr = lane >> 4;   // staging: 16-byte chunk lc of row lr + 4 i
  vec8 xf[MT][kAwqBpw][4];
#pragma unroll
  for (int bi = 0; bi < kAwqBpw; ++bi) {
    const bool bok = w + kAwqWaves * bi < nb;
    const int b = b0 + (bok ? w + kAwqWaves * bi : 0);
    u32x4_t xr[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mt * 16 + lr + 4 * i;
        const u32x4_t v = *(const u32x4_t*)((const T*)p.x + (int64_t)min(m, p.M - 1) * p.x_stride + 128 * b + 8 * lc);
        xr[mt][i] = (m < p.M && bok) ? v : u32x4_t{0u, 0u, 0u, 0u};
      }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = lr + 4 * i;
        *(u32x4_t*)(xl + row * 256 + (((lc ^ row) & 15) << 4)) = xr[mt][i];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
        xf[mt][bi][s] = __builtin_bit_cast(vec8, *(const u32x4_t*)(xl + a * 256 + ((((4 * s + g) ^ a) & 15) << 4)));
    }
  }

  // offset form: row sums of X over each of this wave's k-blocks (rows 16 mt + 4 g + r of the MFMA output, every column alike)
  // bf16 only: with f16 operands the per-weight form is 2 PACKED f16 ops per two weights and the offset form measured 5 % slower
  // in the model (Qwen2-7B AWQ f16, same box: 3.03 vs 3.19 ms/step, with the correction applied one block late as well as in
  // place); with bf16 operands (no packed bf16 arithmetic: cvt_ubyte + fma in f32 + cvt_pk) it is 7 % faster (3.16 vs 3.40).
  constexpr bool OFFS = (SG == 1) && !EXACTW && (SGL_AWQ_EXACT_WEIGHTS == 0) && (sizeof(T) == 2 && !__is_same(T, _Float16));
  AwqRaw<T> raw;
  raw.init();
  // (kept in LDS, 512 bytes per wave, read back as one broadcast ds_read_b128 per block: in registers they were 32 VGPRs too
  // many for the 256-register budget of two waves per SIMD)
  __shared__ f32x4_t xsum_l[OFFS ? kAwqWaves : 1][OFFS ? MT : 1][OFFS ? kAwqBpw : 1][4];
  if constexpr (OFFS) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int bi = 0; bi < kAwqBpw; ++bi) {
        f32x4_t t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) t = Tr::mfma16(xf[mt][bi][s], AwqRaw<T>::ones(), t);
        if (a == 0) xsum_l[w][mt][bi][g] = t;
      }
    // A barrier, although the data is wave private: the stores sit under a lane condition, and without it hipcc moved the loads
    // of the lanes that do not store ahead of the store instruction (NaNs in the single-tile instantiation).
    __syncthreads();
  }
  AWQ_STAMP(1);  // X fragments built
  const int64_t wbytes64 = (int64_t)(p.N / 16) * p.KB * 1024;
  const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.qpacked, 0, (unsigned)min(wbytes64, (int64_t)0xFFFFFFF0ll), 0x00020000);
  u32x4_t wreg[PD][kAwqBpw];
  uint32_t szreg[PD][kAwqBpw][SG];
  auto issue = [&](int slot, int j) {
    const int t = blockIdx.x + j * G_;
    const int tc = min(t, ntiles - 1);  // scale loads stay in range; weight loads past the end are predicated off
#pragma unroll
    for (int bi = 0; bi < kAwqBpw; ++bi) {
      const bool bok = w + kAwqWaves * bi < nb;
      const int b = b0 + (bok ? w + kAwqWaves * bi : 0);
      const unsigned off = (j < cnt && bok) ? (unsigned)((((int64_t)t * p.KB + b) * 64 + lane) * 16) : 0xFFFFFFF0u;
      wreg[slot][bi] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 2));  // nt: read once
#pragma unroll
      for (int q = 0; q < SG; ++q) {
        const int grp = (128 * b + (128 / SG) * q) / p.G;
        szreg[slot][bi][q] = p.sz[(int64_t)grp * p.N + 16 * tc + a];
      }
    }
  };
#pragma unroll
  for (int j = 0; j < PD; ++j) issue(j, j);

  const int em0 = tid >> 4, en = tid & 15;  // this thread's output elements of every tile: rows em0 + 32 e, column en
  const bool has_bias = p.bias != nullptr;
  const T* biasp = has_bias ? (const T*)p.bias : (const T*)p.sz;  // any readable address when absent
  int64_t ep_loc = 0;
  const float* ep_cs = nullptr;
  if constexpr (EPI == EPI_ROPE) {
    ep_loc = ep.loc[min(em0, p.M - 1)];
    ep_cs = ep.cos_sin + ep.positions[min(em0, p.M - 1)] * 128;
  }

  for (int j0 = 0; j0 < cnt; j0 += TPP) {
    uint16_t braw[TPP];
    if (slabs == nullptr) {
#pragma unroll
      for (int jj = 0; jj < TPP; ++jj) {
        const int ne = min((int)(blockIdx.x + min(j0 + jj, cnt - 1) * G_) * 16 + en, p.N - 1);
        braw[jj] = *(const uint16_t*)(biasp + (has_bias ? ne : 0));
      }
    }
#pragma unroll
    for (int jj = 0; jj < TPP; ++jj) {
      const int j = j0 + jj;
      const int slot = (TPP == 1) ? 0 : (jj % PD);
      f32x4_t acc[MT];
      AWQ_STAMP(2 + 7 * (j0 / TPP) + jj);  // tile jj of the phase starts
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if constexpr (OFFS) {
        // The correction of a block reads that block's MFMA results with VALU instructions: applied right behind the MFMAs it
        // stalled on their latency four times per tile (f16: 5 % SLOWER than the per-weight form).  It is applied one block late
        // (two temporary tiles, ping-pong), under the next block's dequantisation and MFMAs.
        f32x4_t tmp[2][MT];
        auto correct = [&](int bi, const f32x4_t (&t)[MT]) __attribute__((always_inline)) {
          const uint32_t szw = szreg[slot][bi][0];
          const float sf = awq_scale_f32<T>(szw), nzc = -(float)(16u + (szw >> 16));
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4_t xs = xsum_l[w][mt][bi][g];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][r] = fmaf(fmaf(nzc, xs[r], t[mt][r]), sf, acc[mt][r]);
          }
        };
#pragma unroll
        for (int bi = 0; bi < kAwqBpw; ++bi) {
          const u32x4_t wq = wreg[slot][bi];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) tmp[bi & 1][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const vec8 wfrag = raw.run(wq[s]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) tmp[bi & 1][mt] = Tr::mfma16(xf[mt][bi][s], wfrag, tmp[bi & 1][mt]);
          }
          if (bi > 0) correct(bi - 1, tmp[(bi - 1) & 1]);
        }
        correct(kAwqBpw - 1, tmp[(kAwqBpw - 1) & 1]);
      } else {
#pragma unroll
        for (int bi = 0; bi < kAwqBpw; ++bi) {
          const u32x4_t wq = wreg[slot][bi];
          AwqDequant<T> dq;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (s % (4 / SG) == 0) dq.setup(szreg[slot][bi][(s * SG) / 4]);
            const vec8 wfrag = dq.run(wq[s]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = Tr::mfma16(xf[mt][bi][s], wfrag, acc[mt]);
          }
        }
      }
      if constexpr (TPP > 1) issue(slot, j + PD);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[jj][w][mt * 16 + 4 * g + r][a] = acc[mt][r];
    }
    AWQ_STAMP(6 + 7 * (j0 / TPP));  // this phase's tiles dequantised and multiplied, partial sums in LDS
    __syncthreads();
    AWQ_STAMP(7 + 7 * (j0 / TPP));  // barrier passed
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int em = em0 + 32 * e;
      if (em >= MT * 16) continue;
#pragma unroll
      for (int jj = 0; jj < TPP; ++jj) {
        const int j = j0 + jj;
        const int n0 = (blockIdx.x + j * G_) * 16;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < kAwqWaves; ++ww) v += red[jj][ww][em][en];
        const bool live = j < cnt && em < p.M && n0 + en < p.N;
        if (slabs != nullptr) {
          if (live) slabs[((int64_t)kr * p.M + em) * p.N + n0 + en] = v;
        } else {
          if (has_bias) v += (float)__builtin_bit_cast(T, braw[jj]);
          if constexpr (EPI == EPI_NONE) {
            if (live) ((T*)p.y)[(int64_t)em * p.y_stride + n0 + en] = (T)v;
          } else {
            float cv = 0.f, sv = 0.f;
            if constexpr (EPI == EPI_ROPE) {
              const int i = 8 * ((n0 & 127) >> 4) + (en & 7);
              cv = ep_cs[i];
              sv = ep_cs[64 + i];
            }
            epi_store<T, EPI>(v, live, em, en, n0, 16, ep, p.y, p.y_stride, cv, sv, ep_loc);
          }
        }
      }
    }
    AWQ_STAMP(8 + 7 * (j0 / TPP));  // outputs stored
    if (TPP > 1) __syncthreads();  // the next phase overwrites red
  }
}

// y[m][n] = T(sum_kr slabs[kr][m][n] + bias[n])
template <typename T>
__global__ __launch_bounds__(256) void awq_slab_reduce_kernel(const float* __restrict__ slabs, int S, const T* bias, T* out,
                                                              int64_t out_stride, int M, int N) {
  const int64_t total = (int64_t)M * (N / 4);
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int m = (int)(idx / (N / 4)), n = (int)(idx - (int64_t)m * (N / 4)) * 4;
    f32x4_t v = *(const f32x4_t*)(slabs + (int64_t)m * N + n);
    for (int sI = 1; sI < S; ++sI) v += *(const f32x4_t*)(slabs + ((int64_t)sI * M + m) * N + n);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float r = v[c];
      if (bias) r += (float)bias[n + c];
      out[(int64_t)m * out_stride + n + c] = (T)r;
    }
  }
}

// qweight [K][N/8] (AWQ nibble order) + qzeros [K/G][N/8] + scales [K/G][N]  ->  qpacked, sz (layouts above)
__global__ __launch_bounds__(256) void awq_repack_w_kernel(const uint32_t* __restrict__ qweight, uint32_t* __restrict__ qpacked,
                                                           int K, int N) {
  const int KB = K / 128, NC = N / 8;
  const int64_t total = (int64_t)(N / 16) * KB * 256;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
    const int s = (int)(o & 3), lane = (int)((o >> 2) & 63);
    const int64_t tb = o >> 8;
    const int b = (int)(tb % KB), t = (int)(tb / KB);
    const int a = lane & 15, g = lane >> 4;
    const int n = 16 * t + a;
    const int shift = 4 * ((n & 1) * 4 + ((n & 7) >> 1));  // AWQ order [0,4,1,5,2,6,3,7]: column j sits in nibble (j&1)*4 + j/2
    uint32_t out = 0;
#pragma unroll
    for (int pq = 0; pq < 8; ++pq) {
      const int k = 128 * b + 32 * s + 8 * g + pq;
      const int nib = (pq & 1) * 4 + (pq >> 1);  // element pq of the lane's k-run sits in nibble (pq & 1) * 4 + pq / 2
      out |= ((qweight[(int64_t)k * NC + (n >> 3)] >> shift) & 0xFu) << (4 * nib);
    }
    qpacked[o] = out;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void awq_repack_sz_kernel(const uint32_t* __restrict__ qzeros, const T* __restrict__ scales,
                                                            uint32_t* __restrict__ sz, int groups, int N) {
  const int NC = N / 8;
  const int64_t total = (int64_t)groups * N;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
    const int n = (int)(o % N);
    const int64_t grp = o / N;
    const int shift = 4 * ((n & 1) * 4 + ((n & 7) >> 1));
    const uint32_t z = (qzeros[grp * NC + (n >> 3)] >> shift) & 0xFu;
    const uint32_t hi = __is_same(T, _Float16) ? (0xE400u | z) : z;   // f16: the bits of -(1024 + z)
    sz[o] = (hi << 16) | (uint32_t)__builtin_bit_cast(uint16_t, scales[o]);
  }
}

int g_awq_exact_weights = 0;   // sgl_mi355_awq_set_exact_weights

inline int awq_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return cus;
}

template <typename T, int MT, int SG, int EPI = EPI_NONE>
int awq_launch(const AwqGemmParams& p, int kranges, float* slabs, hipStream_t st, const EpiParams& ep = EpiParams{}) {
  const int cus = awq_cus();
  const int per_range = cus / kranges > 0 ? cus / kranges : 1;
  const int ntiles = p.N / 16;
  const int gx = ntiles < per_range ? ntiles : per_range;
  const dim3 grid(gx, kranges);
  if constexpr (SG == 1 && !__is_same(T, _Float16) && SGL_AWQ_EXACT_WEIGHTS == 0) {   // the only instantiations the two forms differ in
    if (g_awq_exact_weights) {
      if (ntiles <= gx)
        hipLaunchKernelGGL((awq_gemm_kernel<T, MT, 1, 1, SG, EPI, true>), grid, dim3(kAwqWaves * 64), 0, st, p, ntiles, slabs, ep);
      else
        hipLaunchKernelGGL((awq_gemm_kernel<T, MT, 2, (MT >= 4 ? 2 : 4), SG, EPI, true>), grid, dim3(kAwqWaves * 64), 0, st, p, ntiles, slabs, ep);
      SGL_HIP_LAUNCH_CHECK();
      return SGL_MI355_OK;
    }
  }
  if (ntiles <= gx)
    hipLaunchKernelGGL((awq_gemm_kernel<T, MT, 1, 1, SG, EPI>), grid, dim3(kAwqWaves * 64), 0, st, p, ntiles, slabs, ep);
  else
    hipLaunchKernelGGL((awq_gemm_kernel<T, MT, 2, (MT >= 4 ? 2 : 4), SG, EPI>), grid, dim3(kAwqWaves * 64), 0, st, p, ntiles, slabs, ep);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// Repacked weight -> dense W [N, K] (row-major, the layout the tiled GEMM multiplies by) in ONE pass: for prefill-sized M the
// reference structure is awq_dequantize ([K, N]) + matmul (awq.py:401-418); reading the int4 image that already exists for the
// decode GEMM and writing the transposed rows directly replaces dequantise + a separate 2-D transpose.  One wave per
// (16-row tile, 128-k block): a wave-wide 16-byte load, four dequantised k-steps, four 16-byte stores per lane (64 contiguous
// bytes per row and k-step).  Values are AwqDequant's, i.e. exactly awq_dequantize's.
template <typename T, int SG>
__global__ __launch_bounds__(256) void awq_unpack_nk_kernel(const uint32_t* __restrict__ qpacked, const uint32_t* __restrict__ sz,
                                                            T* __restrict__ out, int N, int K, int G, int KB, int64_t nunits) {
  typedef typename ElemTraits<T>::vec8 vec8;
  const int lane = threadIdx.x & 63;
  const int a = lane & 15, g = lane >> 4;
  for (int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); unit < nunits; unit += (int64_t)gridDim.x * 4) {
    const int t = (int)(unit / KB), b = (int)(unit - (int64_t)t * KB);
    const u32x4_t wq = *(const u32x4_t*)(qpacked + (unit * 64 + lane) * 4);
    uint32_t szw[SG];
#pragma unroll
    for (int q = 0; q < SG; ++q) szw[q] = sz[(int64_t)((128 * b + (128 / SG) * q) / G) * N + 16 * t + a];
    T* orow = out + (int64_t)(16 * t + a) * K + 128 * b + 8 * g;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      AwqDequant<T> dq;
      dq.setup(szw[SG == 1 ? 0 : (SG == 2 ? s4 / 2 : s4)]);
      const vec8 v = dq.run(wq[s4]);
      *(vec8*)(orow + 32 * s4) = v;
    }
  }
}

template <typename T, int EPI = EPI_NONE>
int awq_dispatch(const AwqGemmParams& p, int kranges, float* slabs, hipStream_t st, const EpiParams& ep = EpiParams{}) {
  const int sg = p.G % 128 == 0 ? 1 : 128 / p.G;
#define SGL_AWQ_CASE(MTv)                                                            \
  if (sg == 1) return awq_launch<T, MTv, 1, EPI>(p, kranges, slabs, st, ep);         \
  if (sg == 2) return awq_launch<T, MTv, 2, EPI>(p, kranges, slabs, st, ep);         \
  return awq_launch<T, MTv, 4, EPI>(p, kranges, slabs, st, ep)
  if (p.M <= 16) { SGL_AWQ_CASE(1); }
  if (p.M <= 32 || EPI != EPI_NONE) { SGL_AWQ_CASE(2); }
  if constexpr (EPI == EPI_NONE) { SGL_AWQ_CASE(4); }
  return SGL_MI355_EINVAL;
#undef SGL_AWQ_CASE
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

// Dense W [N, K] (scale dtype) from the repacked image: dequantise + transpose in one pass, for M > 32 (prefill).
extern "C" int sgl_mi355_awq_unpack_nk(const void* qpacked, const void* sz, void* out, int N, int K, int group_size, int dtype,
                                       void* stream) {
  SGL_CHECK(N > 0 && K > 0 && group_size > 0, "awq_unpack_nk: bad shape");
  SGL_CHECK(qpacked && sz && out, "awq_unpack_nk: null pointer");
  SGL_CHECK(K % 128 == 0 && N % 16 == 0, "awq_unpack_nk: needs K %% 128 == 0 and N %% 16 == 0 (K=%d N=%d)", K, N);
  SGL_CHECK(K % group_size == 0 && (group_size % 128 == 0 || group_size == 64 || group_size == 32),
            "awq_unpack_nk: group_size=%d must be 32, 64 or a multiple of 128 dividing K", group_size);
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "awq_unpack_nk: dtype must be bf16 or f16");
  SGL_CHECK(((uintptr_t)qpacked % 16) == 0 && ((uintptr_t)out % 16) == 0, "awq_unpack_nk: qpacked / out must be 16-byte aligned");
  const int KB = K / 128;
  const int64_t nunits = (int64_t)(N / 16) * KB;
  const unsigned blocks = (unsigned)((nunits + 3) / 4 > 16384 ? 16384 : (nunits + 3) / 4);
  const int sg = group_size % 128 == 0 ? 1 : 128 / group_size;
  hipStream_t st = (hipStream_t)stream;
#define SGL_UNPACK(Tv, SGv)                                                                                                \
  hipLaunchKernelGGL((awq_unpack_nk_kernel<Tv, SGv>), dim3(blocks), dim3(256), 0, st, (const uint32_t*)qpacked, (const uint32_t*)sz, \
                     (Tv*)out, N, K, group_size, KB, nunits)
  if (dtype == SGL_F16) {
    if (sg == 1) SGL_UNPACK(_Float16, 1); else if (sg == 2) SGL_UNPACK(_Float16, 2); else SGL_UNPACK(_Float16, 4);
  } else {
    if (sg == 1) SGL_UNPACK(__bf16, 1); else if (sg == 2) SGL_UNPACK(__bf16, 2); else SGL_UNPACK(__bf16, 4);
  }
#undef SGL_UNPACK
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// One-off re-layout of an AWQ weight for sgl_mi355_awq_gemm (see the layout note above).  qpacked: K*N/8 int32, sz: (K/G)*N int32.
extern "C" int sgl_mi355_awq_repack(const void* qweight, const void* scales, const void* qzeros, void* qpacked, void* sz, int K,
                                    int N, int group_size, int scale_dtype, void* stream) {
  SGL_CHECK(K > 0 && N > 0 && group_size > 0, "awq_repack: bad shape");
  SGL_CHECK(qweight && scales && qzeros && qpacked && sz, "awq_repack: null pointer");
  SGL_CHECK(K % 128 == 0 && N % 16 == 0, "awq_repack: needs K %% 128 == 0 and N %% 16 == 0 (K=%d N=%d)", K, N);
  SGL_CHECK(K % group_size == 0 && (group_size % 128 == 0 || group_size == 64 || group_size == 32),
            "awq_repack: group_size=%d must be 32, 64 or a multiple of 128 dividing K", group_size);
  SGL_CHECK(scale_dtype == SGL_BF16 || scale_dtype == SGL_F16, "awq_repack: scales must be f16 or bf16");
  hipStream_t st = (hipStream_t)stream;
  const int64_t tw = (int64_t)K * N / 8, ts = (int64_t)(K / group_size) * N;
  const unsigned bw = (unsigned)((tw + 255) / 256 > 16384 ? 16384 : (tw + 255) / 256);
  const unsigned bs = (unsigned)((ts + 255) / 256 > 16384 ? 16384 : (ts + 255) / 256);
  hipLaunchKernelGGL(awq_repack_w_kernel, dim3(bw), dim3(256), 0, st, (const uint32_t*)qweight, (uint32_t*)qpacked, K, N);
  SGL_HIP_LAUNCH_CHECK();
  if (scale_dtype == SGL_F16)
    hipLaunchKernelGGL((awq_repack_sz_kernel<_Float16>), dim3(bs), dim3(256), 0, st, (const uint32_t*)qzeros,
                       (const _Float16*)scales, (uint32_t*)sz, K / group_size, N);
  else
    hipLaunchKernelGGL((awq_repack_sz_kernel<__bf16>), dim3(bs), dim3(256), 0, st, (const uint32_t*)qzeros, (const __bf16*)scales,
                       (uint32_t*)sz, K / group_size, N);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// f32 [M, N] slabs of workspace sgl_mi355_awq_gemm wants (1: none)
// 1: sgl_mi355_awq_gemm (and its fused forms) multiply by awq_dequantize's weights ROUNDED to the scale dtype for bf16 as well --
// bit-compatible with the reference's awq_dequantize -> matmul (awq.py:401-418), for parity runs; 0 (default): the offset form
// for bf16 operands with one scale group per 128-k block, which multiplies by the exact (q - z) * s (7 % faster, closer to exact
// arithmetic, not bit-identical to dequantise + matmul).  f16 always uses the rounded weights.  Process-wide; not a per-call flag.
extern "C" int sgl_mi355_awq_set_exact_weights(int on) {
  g_awq_exact_weights = on ? 1 : 0;
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_awq_gemm_num_kranges(int M, int K) {
  const int blocks = kAwqWaves * awq_bpw(M > 32 ? 4 : 2);  // 128-k blocks per k-range: 4096 k for M <= 32, 2048 k for 33..64
  return (K / 128 + blocks - 1) / blocks;
}

namespace {
// argument checks and parameter block shared by the awq_gemm entry points
int awq_params(AwqGemmParams& p, const char* who, const void* x, int64_t x_stride_elems, const void* qpacked, const void* sz, void* y,
               int64_t y_stride_elems, const void* bias, int M, int N, int K, int group_size, int dtype) {
  SGL_CHECK(M > 0 && N > 0 && K > 0 && group_size > 0, "%s: bad shape", who);
  SGL_CHECK(M <= 64, "%s: M=%d exceeds 64 (use awq_unpack_nk + the tiled GEMM)", who, M);
  SGL_CHECK(x && qpacked && sz && y, "%s: null pointer", who);
  SGL_CHECK(K % 128 == 0 && N % 16 == 0, "%s: needs K %% 128 == 0 and N %% 16 == 0 (K=%d N=%d)", who, K, N);
  SGL_CHECK(K % group_size == 0 && (group_size % 128 == 0 || group_size == 64 || group_size == 32),
            "%s: group_size=%d must be 32, 64 or a multiple of 128 dividing K", who, group_size);
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "%s: dtype must be bf16 or f16", who);
  SGL_CHECK(x_stride_elems % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)qpacked % 16) == 0,
            "%s: x rows and qpacked must be 16-byte aligned", who);
  SGL_CHECK((int64_t)K * N / 2 < 0xFFFFFFF0ll, "%s: weight larger than 4 GiB", who);
  p.x = (const char*)x; p.x_stride = x_stride_elems;
  p.qpacked = (const uint32_t*)qpacked; p.sz = (const uint32_t*)sz;
  p.y = y; p.y_stride = y_stride_elems; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.G = group_size; p.KB = K / 128;
  return SGL_MI355_OK;
}
}  // namespace

// y [M, N] = x [M, K] . dequant(qpacked, sz) (+ bias); M <= 64; dtype of x / y / bias / scales: SGL_BF16 or SGL_F16.
extern "C" int sgl_mi355_awq_gemm(const void* x, int64_t x_stride_elems, const void* qpacked, const void* sz, void* y,
                                  int64_t y_stride_elems, const void* bias, int M, int N, int K, int group_size, int dtype,
                                  float* workspace, int64_t workspace_floats, void* stream) {
  SGL_CHECK(M >= 0, "awq_gemm: bad shape");
  if (M == 0) return SGL_MI355_OK;
  AwqGemmParams p;
  const int prc = awq_params(p, "awq_gemm", x, x_stride_elems, qpacked, sz, y, y_stride_elems, bias, M, N, K, group_size, dtype);
  if (prc != SGL_MI355_OK) return prc;
  hipStream_t st = (hipStream_t)stream;
  const int kranges = sgl_mi355_awq_gemm_num_kranges(M, K);
  float* slabs = nullptr;
  if (kranges > 1) {
    SGL_CHECK(workspace != nullptr && workspace_floats >= (int64_t)kranges * M * N && N % 4 == 0,
              "awq_gemm: K=%d needs a workspace of %d x M x N floats", K, kranges);
    slabs = workspace;
  }
  const int rc = dtype == SGL_BF16 ? awq_dispatch<__bf16>(p, kranges, slabs, st) : awq_dispatch<_Float16>(p, kranges, slabs, st);
  if (rc != SGL_MI355_OK || kranges == 1) return rc;
  const int64_t items = (int64_t)M * (N / 4);
  const unsigned blocks = (unsigned)((items + 255) / 256 > 2048 ? 2048 : (items + 255) / 256);
  if (dtype == SGL_BF16)
    hipLaunchKernelGGL((awq_slab_reduce_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, slabs, kranges, (const __bf16*)bias,
                       (__bf16*)y, y_stride_elems, M, N);
  else
    hipLaunchKernelGGL((awq_slab_reduce_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, slabs, kranges, (const _Float16*)bias,
                       (_Float16*)y, y_stride_elems, M, N);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// Producer half of the launch-boundary split-K reduce for an int4 weight: raw f32 partial sums [kranges, M, N]
// (kranges = sgl_mi355_awq_gemm_num_kranges(M, K), also 1); the consumer (sgl_mi355_fused_add_rmsnorm_quant_fp8 with slabs and no
// scales) sums them in the order awq_gemm's own reduce kernel does.
extern "C" int sgl_mi355_awq_gemm_slabs(const void* x, int64_t x_stride_elems, const void* qpacked, const void* sz, float* slabs,
                                        int M, int N, int K, int group_size, int dtype, void* stream) {
  AwqGemmParams p;
  const int prc = awq_params(p, "awq_gemm_slabs", x, x_stride_elems, qpacked, sz, slabs, N, nullptr, M, N, K, group_size, dtype);
  if (prc != SGL_MI355_OK) return prc;
  const int kranges = sgl_mi355_awq_gemm_num_kranges(M, K);
  SGL_CHECK(kranges > 1, "awq_gemm_slabs: M=%d K=%d runs as one k-range (no slabs: the kernel would store T-typed outputs into the f32 buffer); use awq_gemm", M, K);
  return dtype == SGL_BF16 ? awq_dispatch<__bf16>(p, kranges, slabs, (hipStream_t)stream)
                           : awq_dispatch<_Float16>(p, kranges, slabs, (hipStream_t)stream);
}

// act [M, N/2] = SiluAndMul(awq_gemm(x, W)) in one launch (AWQLinearMethod.apply awq.py:401-418 -> SiluAndMul activation.py:60-63):
// the weight's packed columns are interleaved BEFORE sgl_mi355_awq_repack so that 16-column tile t = [gate columns 8t..8t+7 | up
// columns 8t..8t+7].  K <= 4096 (one k-range).  Bit-identical to awq_gemm followed by silu_and_mul.
extern "C" int sgl_mi355_awq_gemm_silu_mul(const void* x, int64_t x_stride_elems, const void* qpacked_interleaved,
                                           const void* sz_interleaved, void* act, int64_t act_stride_elems, int M, int N, int K,
                                           int group_size, int dtype, void* stream) {
  AwqGemmParams p;
  const int prc = awq_params(p, "awq_gemm_silu_mul", x, x_stride_elems, qpacked_interleaved, sz_interleaved, act, act_stride_elems,
                             nullptr, M, N, K, group_size, dtype);
  if (prc != SGL_MI355_OK) return prc;
  SGL_CHECK(M <= 32 && sgl_mi355_awq_gemm_num_kranges(M, K) == 1, "awq_gemm_silu_mul: needs M <= 32 and K <= 4096 (one k-range); M=%d K=%d", M, K);
  return dtype == SGL_BF16 ? awq_dispatch<__bf16, EPI_SILU>(p, 1, nullptr, (hipStream_t)stream)
                           : awq_dispatch<_Float16, EPI_SILU>(p, 1, nullptr, (hipStream_t)stream);
}

// qkv_proj (int4) + neox RoPE + set_kv_buffer in one launch: columns interleaved before the repack so that inside every q / k
// head tile u = [columns 8u..8u+7 | columns 64+8u..64+8u+7]; v heads in natural order.  head_dim = rotary_dim = 128, K <= 4096.
// Bit-identical to awq_gemm -> rope_set_kv.
extern "C" int sgl_mi355_awq_qkv_rope_set_kv(const void* x, int64_t x_stride_elems, const void* qpacked_interleaved,
                                             const void* sz_interleaved, void* q_out, int64_t q_stride_elems,
                                             const void* bias_interleaved, const int64_t* positions, const float* cos_sin_cache,
                                             const int64_t* loc, void* k_buffer, void* v_buffer, int64_t k_slot_stride,
                                             int64_t v_slot_stride, int M, int num_q_heads, int num_kv_heads, int head_dim, int K,
                                             int group_size, int dtype, int kv_dtype, float k_scale, float v_scale, void* stream) {
  SGL_CHECK(positions && cos_sin_cache && loc && k_buffer && v_buffer, "awq_qkv_rope_set_kv: null pointer");
  SGL_CHECK(kv_dtype == dtype || kv_dtype == SGL_FP8_E4M3, "awq_qkv_rope_set_kv: the pool holds the activation dtype or float8_e4m3fn (kv_dtype %d)", kv_dtype);
  SGL_CHECK(head_dim == 128, "awq_qkv_rope_set_kv: head_dim (= rotary_dim) must be 128, got %d", head_dim);
  AwqGemmParams p;
  const int prc = awq_params(p, "awq_qkv_rope_set_kv", x, x_stride_elems, qpacked_interleaved, sz_interleaved, q_out, q_stride_elems,
                             bias_interleaved, M, (num_q_heads + 2 * num_kv_heads) * 128, K, group_size, dtype);
  if (prc != SGL_MI355_OK) return prc;
  SGL_CHECK(M <= 32 && sgl_mi355_awq_gemm_num_kranges(M, K) == 1, "awq_qkv_rope_set_kv: needs M <= 32 and K <= 4096 (one k-range); M=%d K=%d", M, K);
  EpiParams ep;
  ep.positions = positions; ep.cos_sin = cos_sin_cache; ep.loc = loc; ep.k_buf = k_buffer; ep.v_buf = v_buffer;
  ep.k_slot_stride = k_slot_stride; ep.v_slot_stride = v_slot_stride; ep.hq = num_q_heads; ep.hkv = num_kv_heads;
  ep.kv_fp8 = kv_dtype == SGL_FP8_E4M3 ? 1 : 0; ep.k_scale = k_scale; ep.v_scale = v_scale;
  return dtype == SGL_BF16 ? awq_dispatch<__bf16, EPI_ROPE>(p, 1, nullptr, (hipStream_t)stream, ep)
                           : awq_dispatch<_Float16, EPI_ROPE>(p, 1, nullptr, (hipStream_t)stream, ep);
}
