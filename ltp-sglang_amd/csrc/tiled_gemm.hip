// LDS-tiled MFMA GEMM for prefill-sized M (> 64): Y[M,N] = (X[M,K] . W[N,K]^T) * sx[m] * sw[n] + bias[n]
//
// Replaces fp8_scaled_mm (sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1071-1146: CUTLASS sm89/sm90/sm100 tile
// dispatch) for e4m3fn x e4m3fn -> f32 -> bf16/f16 with the fused per-row/per-column scale (+bias) epilogue,
// and the unquantised bf16/f16 linear when both scale vectors are NULL.
//
// gfx950 structure: 128x128 output tile per 256-thread workgroup (4 waves, 64x64 = 4x4 MFMA 16x16x32 tiles per
// wave), 128-byte K slices (128 fp8 / 64 half elements), register-staged double-buffered LDS, 16-B-chunk XOR
// swizzle so the ds_read_b128 fragment reads are bank-conflict free, XCD-aware bijective block remap so the
// workgroups that share an X row panel run on the same XCD (one L2).
#include "common.h"
#include "silu_lut.h"
#include <atomic>
#include <type_traits>

namespace {

struct GemmParams {
  const char* x;
  int64_t x_stride;  // bytes
  const char* w;
  int64_t w_stride;  // bytes
  void* y;
  int64_t y_stride;  // elements
  const float* sx;
  const float* sw;
  const void* bias;
  int M, N, kbytes;
  int tiles_m, tiles_n;
  int group_m;  // row tiles per scheduling group (256x256 kernel)
  float* slabs;  // split-K (128x128 kernel, blockIdx.y = K range of `kt_per` slices): raw f32 sums [splits][M][N], else NULL
  int kt_per;
  const uint16_t* silu_lut = nullptr;  // SiluAndMul epilogue of the 256x256 kernel: the table of silu_lut.h in global memory
  int stagger_q, stagger_cus;  // 256x256 kernel: start stagger of each CU's first workgroup (quantum in 1024-cycle units; CU count)
  int* sched = nullptr;  // persistent 256x256 kernel, DYNAMIC tile schedule (round 4): 8 words, one ticket counter per XCD class, ZERO at launch
                         // (the launcher enqueues the memset); NULL = the static schedule (tile j, j + per, ... of the XCD's range)
#ifdef SGL_GEMM_TIMELINE
  long long* tl;  // tools/microbench/gemm256_timeline.hip: s_memtime stamps of workgroup 0, slices 8..11
#endif
};

enum { TG_FP8 = 0, TG_BF16 = 1, TG_F16 = 2 };

template <int ES>
__device__ __forceinline__ void mfma_chunk(const u32x4_t& a, const u32x4_t& b, f32x4_t& acc) {
  if constexpr (ES == TG_FP8) {
    const long a0 = ((long)a[1] << 32) | (long)a[0], a1 = ((long)a[3] << 32) | (long)a[2];
    const long b0 = ((long)b[1] << 32) | (long)b[0], b1 = ((long)b[3] << 32) | (long)b[2];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, acc, 0, 0, 0);
  } else if constexpr (ES == TG_BF16) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
  }
}

// fp8: the block-scaled form v_mfma_scale_f32_16x16x128_f8f6f4 with unit (E8M0 = 127) scales runs at twice the rate
// of the non-scaled 16x16x32 fp8 MFMA; lane (row a, group g) supplies 32 K bytes (hardware order: 32 g .. 32 g + 31).
// The sum over K does not care which 32 bytes a lane group takes as long as A and B agree, so group g takes the 16-byte
// chunks g and 4 + g of the 128-byte slice: the same LDS read pattern as the 16x16x32 path, conflict free under lds_off.
typedef int i32x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mfma_mx(const u32x4_t& a0, const u32x4_t& a1, const u32x4_t& b0, const u32x4_t& b1, f32x4_t& acc) {
  const i32x8_t av = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
  const i32x8_t bv = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

constexpr int BM = 128, BN = 128, BKB = 128;  // BKB: bytes of K per slice
constexpr int TILE_BYTES = BM * BKB;          // 16 KiB per operand per buffer

// byte offset of 16-B chunk `c` (0..7) of row `r` in a [128][128 B] swizzled image
__device__ __forceinline__ int lds_off(int r, int c) { return r * BKB + ((c ^ ((r >> 1) & 7)) << 4); }

template <int ES, typename OutT>
__global__ __launch_bounds__(256, 2) void tiled_gemm_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // ---- XCD-aware bijective remap of the linear block id (8 XCDs, round-robin dispatch) ----
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int q = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
  const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + bid / 8;
  const int tm = wgid / p.tiles_n, tn = wgid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;  // wave's 64x64 sub-tile
  const int a = lane & 15, g = lane >> 4;

  // global->LDS staging: thread loads 16-B chunk (tid & 7) of rows (tid >> 3) + 32 i, i = 0..3, of both operands
  const int ld_c = tid & 7, ld_r = tid >> 3;
  const char* xg[4];
  const char* wg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    xg[i] = p.x + (int64_t)min(m0 + ld_r + 32 * i, p.M - 1) * p.x_stride + ld_c * 16;
    wg[i] = p.w + (int64_t)min(n0 + ld_r + 32 * i, p.N - 1) * p.w_stride + ld_c * 16;
  }
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};
  u32x4_t xr[4], wr[4];
  auto gload = [&](int kt) {
    const int off = kt * BKB;
    const bool ok = off + ld_c * 16 < p.kbytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xr[i] = ok ? *(const u32x4_t*)(xg[i] + off) : zero4;
      wr[i] = ok ? *(const u32x4_t*)(wg[i] + off) : zero4;
    }
  };
  auto lstore = [&](int buf) {
    char* xa = smem + buf * 2 * TILE_BYTES;
    char* wa = xa + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(u32x4_t*)(xa + lds_off(ld_r + 32 * i, ld_c)) = xr[i];
      *(u32x4_t*)(wa + lds_off(ld_r + 32 * i, ld_c)) = wr[i];
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk_all = (p.kbytes + BKB - 1) / BKB;
  const int kt0 = p.slabs ? (int)blockIdx.y * p.kt_per : 0;            // split-K: this workgroup's slice range
  const int nk = p.slabs ? min(nk_all, kt0 + p.kt_per) : nk_all;
  gload(kt0);
  lstore(kt0 & 1);
  __syncthreads();
  for (int kt = kt0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const char* xa = smem + buf * 2 * TILE_BYTES;
    const char* wa = xa + TILE_BYTES;
    if constexpr (ES == TG_FP8) {
      u32x4_t af[4][2], bf[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          af[i][h] = *(const u32x4_t*)(xa + lds_off(wm + 16 * i + a, 4 * h + g));
          bf[i][h] = *(const u32x4_t*)(wa + lds_off(wn + 16 * i + a, 4 * h + g));
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma_mx(af[i][0], af[i][1], bf[j][0], bf[j][1], acc[i][j]);
    } else {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4_t af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *(const u32x4_t*)(xa + lds_off(wm + 16 * i + a, 4 * s + g));
        bf[i] = *(const u32x4_t*)(wa + lds_off(wn + 16 * i + a, 4 * s + g));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma_chunk<ES>(af[i], bf[j], acc[i][j]);
    }
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: acc[i][j][r] -> row m0+wm+16i+4g+r, col n0+wn+16j+a ----
  if (p.slabs) {  // split-K: raw partial sums; scales / bias / rounding happen in the reduce kernel
    float* sl = p.slabs + (int64_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn + 16 * j + a;
      if (n >= p.N) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm + 16 * i + 4 * g + r;
          if (m < p.M) sl[(int64_t)m * p.N + n] = acc[i][j][r];
        }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn + 16 * j + a;
    if (n >= p.N) continue;
    const float swv = p.sw ? p.sw[n] : 1.0f;
    const float bv = p.bias ? (float)((const OutT*)p.bias)[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + 16 * i + 4 * g + r;
        if (m < p.M) {
          float v = acc[i][j][r];
          if (p.sx) v *= p.sx[m];
          v = v * swv + bv;
          ((OutT*)p.y)[(int64_t)m * p.y_stride + n] = (OutT)v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// fp8 256x256 tile: 8 waves (2 along M x 4 along N, 128 x 64 outputs each = 8 x 4 scaled MFMAs per 128-byte K slice),
// both operands staged global -> LDS by the LDS-DMA form of the load (no VGPR staging, swizzle applied on the global
// source address), two 64 KiB buffers, ONE barrier per K slice with the next slice's loads in flight across the whole
// compute phase.  Computed transposed (A operand = W rows, B operand = X rows) so a lane ends up with 4 consecutive n of
// one output row: 8-byte packed stores.
// ---------------------------------------------------------------------------------------------------------
constexpr int T2 = 256;
constexpr int OPB = T2 * BKB;  // 32 KiB per operand per buffer

#ifdef SGL_GEMM_TIMELINE
#define TL_STAMP(i)                                                                            \
  do {                                                                                         \
    if (bid == 0 && kt >= 8 && kt < 12) {                                                      \
      const long long t_ = (long long)__builtin_amdgcn_s_memtime();                            \
      if (lane == 0) p.tl[(((kt - 8) * 8 + w) * 16) + (i)] = t_;                               \
    }                                                                                          \
  } while (0)
#else
#define TL_STAMP(i)
#endif

__device__ __forceinline__ void glds16(const char* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Fused scale / bias epilogue of the transposed-accumulator kernels: acc[j][i][r] -> row mrow0 + 16 i, col ncol0 + 16 j + r.
// Every operand (NI row scales, NJ x 4 column scales and biases) is loaded up front with ONE wave-uniform branch per optional
// pointer; the value loop is then branch-free arithmetic + stores.  (With the pointers tested and the scales loaded per value,
// the epilogue compiled into several hundred basic blocks of one dependent L2 load each -- with one workgroup per CU nothing
// overlaps it, and it cost about as much as a third of a K = 4096 main loop.)  Same arithmetic as before: x * 1.0f is exact.
template <typename OutT, int NI, int NJ>
__device__ __forceinline__ void epilogue_scaled(const GemmParams& p, const f32x4_t (&acc)[NJ][NI], int mrow0, int ncol0) {
  float sxv[NI], swv[NJ][4], bv[NJ][4];
#pragma unroll
  for (int i = 0; i < NI; ++i) sxv[i] = 1.0f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      swv[j][r] = 1.0f;
      bv[j][r] = 0.0f;
    }
  if (p.sx) {
#pragma unroll
    for (int i = 0; i < NI; ++i) sxv[i] = p.sx[min(mrow0 + 16 * i, p.M - 1)];
  }
  if (p.sw) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) swv[j][r] = p.sw[min(ncol0 + 16 * j + r, p.N - 1)];
  }
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = (float)((const OutT*)p.bias)[min(ncol0 + 16 * j + r, p.N - 1)];
  }
  const bool vec_ok = (p.y_stride % 4 == 0) && (((uintptr_t)p.y & 7) == 0);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = mrow0 + 16 * i;
    OutT* yrow = (OutT*)p.y + (int64_t)min(m, p.M - 1) * p.y_stride;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = ncol0 + 16 * j;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[j][i][r] * sxv[i];
        v[r] = v[r] * swv[j][r] + bv[j][r];
      }
      if (m < p.M && n < p.N) {
        if (vec_ok && n + 3 < p.N) {
          typedef OutT o4_t __attribute__((ext_vector_type(4)));
          const o4_t o = {(OutT)v[0], (OutT)v[1], (OutT)v[2], (OutT)v[3]};
          *(o4_t*)(yrow + n) = o;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) yrow[n + r] = (OutT)v[r];
        }
      }
    }
  }
}

// The same epilogue with the output tile staged through LDS (256x256 kernel, LDS free after the main loop): every wave packs
// its 128 x 64 outputs into a private 16 KiB [128][64] image (16-byte chunks XOR-swizzled by the row) and writes them back as
// whole 128-byte rows, 8 rows per store instruction.  Straight from the accumulator layout a store instruction touches 16 rows x 32 bytes; with every CU in
// its epilogue at the same moment those partial-line writes took 6-9 us per tile (measured by compiling the epilogue out),
// as much as 4-6 K slices of MFMA work.  Needs N % 8 == 0 and 16-byte aligned output rows; otherwise epilogue_scaled.
constexpr int kEpiRowB = 128;  // bytes per staged row: 64 outputs of 2 bytes (8 waves x 16 KiB = the kernel's 128 KiB of LDS)
// PASSES > 1: the wave's rows go through a private image of NI / PASSES row tiles, one pass after the other (same-wave LDS operations
// are ordered) -- the persistent kernel has one 64 KiB slice buffer free for the epilogue, not both.
template <typename OutT, int NI, int NJ, int PASSES = 1>
__device__ __forceinline__ void epilogue_scaled_lds(const GemmParams& p, const f32x4_t (&acc)[NJ][NI], int mrow_base, int ncol_base,
                                                    char* wave_lds, int lane) {
  static_assert(NJ == 4 && (NI == 8 || NI == 4) && NI % PASSES == 0, "one wave = 128 (or 64) rows x 64 columns");
  constexpr int NIP = NI / PASSES;
  const int a = lane & 15, g = lane >> 4;
  float sxv[NI], swv[NJ][4], bv[NJ][4];
#pragma unroll
  for (int i = 0; i < NI; ++i) sxv[i] = 1.0f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      swv[j][r] = 1.0f;
      bv[j][r] = 0.0f;
    }
  if (p.sx) {
#pragma unroll
    for (int i = 0; i < NI; ++i) sxv[i] = p.sx[min(mrow_base + a + 16 * i, p.M - 1)];
  }
  if (p.sw) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) swv[j][r] = p.sw[min(ncol_base + 4 * g + 16 * j + r, p.N - 1)];
  }
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = (float)((const OutT*)p.bias)[min(ncol_base + 4 * g + 16 * j + r, p.N - 1)];
  }
  const int rsub = lane >> 3, c16 = lane & 7;  // 8 lanes cover one 128-byte row
  const bool col_ok = ncol_base + 8 * c16 + 7 < p.N;
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
#pragma unroll
    for (int ii = 0; ii < NIP; ++ii)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int i = ps * NIP + ii;
        typedef OutT o4_t __attribute__((ext_vector_type(4)));
        o4_t o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[j][i][r] * sxv[i];
          v = v * swv[j][r] + bv[j][r];
          o[r] = (OutT)v;
        }
        *(o4_t*)(wave_lds + (16 * ii + a) * kEpiRowB + (((2 * j + (g >> 1)) ^ (a & 7)) << 4) + ((g & 1) << 3)) = o;
      }
    // (same-wave LDS operations are ordered: no barrier between the image's writes and reads)
#pragma unroll
    for (int it = 0; it < 2 * NIP; ++it) {
      const int row = 8 * it + rsub;
      const u32x4_t v = *(const u32x4_t*)(wave_lds + row * kEpiRowB + ((c16 ^ (row & 7)) << 4));
      const int m = mrow_base + 16 * NIP * ps + row;
      // NON-TEMPORAL stores: the output (hundreds of MB per prefill GEMM) is not read again by this launch and would otherwise
      // displace the X / W panels that the other tiles of the XCD re-read from L2.  Same-box A/B (round 2): prefill 1 946 -> 2 000 TFLOP/s.
      if (m < p.M && col_ok) __builtin_nontemporal_store(v, (u32x4_t*)((OutT*)p.y + (int64_t)m * p.y_stride + ncol_base + 8 * c16));
    }
  }
}

// gate_up_proj + SiluAndMul (models/llama.py:94-96, activation.py:60-63) as the epilogue of the 256x256 kernel (round 3): W's rows
// are interleaved in 16-row tiles (8 gate rows, 8 up rows: the packing of the decode path's fused kernel, sgl_kernel.fused.
// interleave_gate_up_rows), so an MFMA column block holds gate columns on lanes g = 0, 1 and the up columns of the same indices on
// lanes g = 2, 3.  Per value exactly the unfused sequence: y = bf16(acc * sx[m] * sw[n]) (fp8_scaled_mm's epilogue), s = bf16(silu(
// y_gate)) through the table of silu_lut.h (exact expression outside it), act = bf16(s * y_up).  The rounded y travel as bf16 pairs
// through ONE v_permlane32_swap per two column blocks (lower half-wave: its own gate + the upper half's up of the even block; upper
// half-wave: the lower half's gate of the odd block + its own up), the 256 x 128 act tile is staged in LDS (256-byte rows, 16-byte
// chunks XOR-swizzled by the row) and leaves as whole rows with non-temporal stores.  With the exact silu in this place the
// epilogue cost more than the separate kernel it replaces (round 1: -4 %); the table is what makes it pay.  Needs N % 256 == 0.
constexpr int kSiluLdsOff = 2 * 2 * 256 * 128;   // the table sits above the two 64 KiB operand buffers for the whole launch
template <typename OutT>
__device__ __forceinline__ void epilogue_silu_lds(const GemmParams& p, const f32x4_t (&acc)[4][8], int m0, int wm, int n0, int wn, char* smem,
                                                  int tid, int lane, char* stg = nullptr) {
  if (stg == nullptr) stg = smem;   // the 64 KiB the act tile is staged in (the persistent kernel passes the slice buffer that is free)
  static_assert(sizeof(OutT) == 2 && !__is_same(OutT, _Float16), "the table is over bf16 bits");
  const int a = lane & 15, g = lane >> 4;
  const uint16_t* lut = (const uint16_t*)(smem + kSiluLdsOff);   // copied there at kernel entry, under the first slice's flight
  float sxv[8], swv[4][4];   // (one wave-uniform branch per optional pointer, as in epilogue_scaled: tested per value, every load is a basic block)
#pragma unroll
  for (int i = 0; i < 8; ++i) sxv[i] = 1.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) swv[j][r] = 1.0f;
  if (p.sx) {
#pragma unroll
    for (int i = 0; i < 8; ++i) sxv[i] = p.sx[min(m0 + wm + a + 16 * i, p.M - 1)];
  }
  if (p.sw) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4_t v = *(const f32x4_t*)(p.sw + n0 + wn + 16 * j + 4 * g);   // (N % 256 == 0: whole, 16-byte aligned groups of four)
#pragma unroll
      for (int r = 0; r < 4; ++r) swv[j][r] = v[r];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = wm + 16 * i + a;
    // y = OutT(acc * sx * sw + 0): the arithmetic of epilogue_scaled, packed as bf16 pairs (r = 0, 1 | r = 2, 3); one row tile at a
    // time, so that the accumulators die as they are consumed (all 64 pairs up front cost the persistent kernel 250 spills)
    uint32_t ypi[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v0 = acc[j][i][2 * h] * sxv[i], v1 = acc[j][i][2 * h + 1] * sxv[i];
        v0 = v0 * swv[j][2 * h] + 0.0f;
        v1 = v1 * swv[j][2 * h + 1] + 0.0f;
        struct P2 { OutT lo, hi; };
        ypi[j][h] = __builtin_bit_cast(uint32_t, P2{(OutT)v0, (OutT)v1});
      }
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      uint32_t gt[2], up[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // x = even block, y = odd block: afterwards lanes 0-31 hold (gate, up) of the even block, lanes 32-63 of the odd block
        const auto sw2 = __builtin_amdgcn_permlane32_swap(ypi[2 * jp][h], ypi[2 * jp + 1][h], false, false);
        gt[h] = sw2[0];
        up[h] = sw2[1];
      }
      uint32_t idx[4];
      bool outside = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        idx[r] = silu_lut_index((r & 1) ? gt[r >> 1] >> 16 : gt[r >> 1] & 0xFFFFu);
        outside = outside || idx[r] >= (uint32_t)kSiluLut;
      }
      float sI[4];
      if (__any(outside)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sI[r] = silu_exact_bf16(__uint_as_float((r & 1) ? gt[r >> 1] & 0xFFFF0000u : gt[r >> 1] << 16));
      } else {
        uint16_t e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = lut[idx[r]];
#pragma unroll
        for (int r = 0; r < 4; ++r) sI[r] = __uint_as_float((uint32_t)e[r] << 16);
      }
      uint32_t o[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float m0v = sI[2 * h] * __uint_as_float(up[h] << 16), m1v = sI[2 * h + 1] * __uint_as_float(up[h] & 0xFFFF0000u);
        asm volatile("" : "+v"(m0v), "+v"(m1v));   // f32 products first: no mixed-precision fma shortcut (the reference rounds twice)
        struct P2 { OutT lo, hi; };
        o[h] = __builtin_bit_cast(uint32_t, P2{(OutT)m0v, (OutT)m1v});
      }
      // act column (within the tile's 128) = wn / 2 + 16 jp + 4 g + r  ->  byte wn + 32 jp + 8 g of the 256-byte row
      const int c = (wn >> 4) + 2 * jp + (g >> 1);
      *(u32x2_t*)(stg + row * 256 + ((c ^ (row & 15)) << 4) + ((g & 1) << 3)) = u32x2_t{o[0], o[1]};
    }
  }
  __syncthreads();
  const int pos = tid & 15, r0 = tid >> 4;   // 16 lanes per row, 32 rows per pass
  OutT* ybase = (OutT*)p.y + (n0 >> 1);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = 32 * it + r0;
    const u32x4_t v = *(const u32x4_t*)(stg + row * 256 + (pos << 4));
    const int m = m0 + row;
    if (m < p.M) __builtin_nontemporal_store(v, (u32x4_t*)(ybase + (int64_t)m * p.y_stride + 8 * (pos ^ (row & 15))));
  }
}

// ES: TG_FP8 (block-scaled MFMA over the whole 128-byte slice) or TG_BF16 / TG_F16 (two 16x16x32 k-steps per slice: the
// same LDS reads, chunk 4 h + g being exactly k-step h's fragment)
// WN: W rows (output columns) per tile.  256, or 128 with four waves -- the same 128 x 64 wave tile and slice pipeline over a
// 256 x 128 output tile, for launches whose 256 x 256 tiles are fewer than the CUs (chunked prefill, M ~ 1024-4096)
template <typename OutT, int NWV, bool DMA = true, int ES = TG_FP8, bool SILU = false, int WN = 256>  // NWV = 8: waves 2 (M) x 4 (N), 128 x 64 outputs each; 4: 2 x 2
__global__ __launch_bounds__(NWV * 64, 1) void fp8_gemm256_kernel(const GemmParams p) {
  constexpr int WCOLS = (NWV == 8 && WN == 256) ? 4 : 2;  // waves along N
  constexpr int JN = WN / WCOLS / 16;         // 16-column W fragments per wave
  constexpr int MR = 256 / (NWV / WCOLS);     // X rows (output rows) per wave: 128, or 64 for eight waves over a 256 x 128 tile
  constexpr int NI = MR / 16;                 // 16-row X fragments per wave
  constexpr int IG = NI / 4;                  // ... per MFMA group (four groups per slice)
  static_assert(!SILU || (WN == 256 && NWV == 8), "the SiLU epilogue is written for the 256 x 256 tile");
  constexpr int RPW = 256 / NWV;              // X staging rows per wave
  constexpr int WRPW = WN / NWV;              // W staging rows per wave
  constexpr int WOPB = WN * BKB;              // W bytes per buffer
  constexpr int BUFB = WOPB + OPB;            // one buffer: [W | X 32 KiB]
  constexpr int NBUF = WN == 128 ? 3 : 2;     // slice buffers: a ring of three 48 KiB buffers for the narrow tile (a slice is half the MFMA time: one slice of prefetch distance does not cover the fetch)
  constexpr int DMA_PER_SLICE = (RPW + WRPW) / 8;  // LDS-DMA instructions per wave and slice
  static_assert(WRPW % 16 == 0 && RPW % 16 == 0, "the staging swizzle below assumes 16-row aligned wave bases");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][W | X]
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int q = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
  const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + bid / 8;
  // grouped order: consecutive workgroups (the ~32 resident on one XCD at a time) cover a block of GM row tiles x 8 column
  // tiles, so an X panel is fetched into that XCD's L2 once per 8 and a W panel once per GM of them instead of one W panel
  // per workgroup (W re-reads from the Infinity Cache were the limiter: 7.5 GB per gate_up GEMM)
  const int GM = p.group_m;
  const int per_group = GM * p.tiles_n;
  const int grp = wgid / per_group, in_grp = wgid - grp * per_group;
  const int gsz = min(p.tiles_m - grp * GM, GM);
  const int tm = grp * GM + in_grp % gsz, tn = in_grp / gsz;
  const int m0 = tm * T2, n0 = tn * WN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (w / WCOLS) * MR, wn = (w % WCOLS) * (16 * JN);
  // Phase stagger: tiles of one launch take the same time on every CU, so without it all 256 CUs reach their epilogues together
  // and the burst of 256 x 128 KiB of output is write-bandwidth bound (5-6 us per tile with the MFMA pipes idle).  The first
  // workgroup of every CU starts up to p.stagger_us late; the workgroups that follow on that CU inherit its phase.
  if (p.stagger_q > 0 && (int)blockIdx.x < p.stagger_cus) {
    const int mask = (p.stagger_q & 64) ? 31 : 15;
    const int steps = ((blockIdx.x >> 3) & mask) * (p.stagger_q & 63);  // units of 64 x 8 cycles
    for (int t = 0; t < steps; ++t) __builtin_amdgcn_s_sleep(8);
  }
  const int a = lane & 15, g = lane >> 4;

  // LDS-DMA staging (buffer_load ... lds): wave w fills rows RPW w .. RPW w + RPW - 1 of both operands, 8 rows (1 KiB) per
  // instruction.  The lane that writes position (lane & 7) of row R loads the chunk the swizzle keeps there; that chunk
  // only depends on the parity of the 8-row group, so two per-lane offsets per operand + wave-uniform scalar offsets cover
  // every instruction.  Rows past M / N are out of the buffer's range and arrive as zeros.
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)((int64_t)p.N * p.w_stride), 0x00020000);
  const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)((int64_t)p.M * p.x_stride), 0x00020000);
  unsigned wvo[2], xvo[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int rl = lane >> 3;                                      // row of 8-row group 0 within the wave's rows (parity `par` adds 8 par rows)
    const int chunk = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);  // ((row >> 1) & 7) for row = base + rl + 8 t, t & 1 == par, base % 16 == 0
    wvo[par] = (unsigned)((int64_t)(n0 + WRPW * w + rl) * p.w_stride) + chunk * 16;
    xvo[par] = (unsigned)((int64_t)(m0 + RPW * w + rl) * p.x_stride) + chunk * 16;
  }
  const int nk = p.kbytes / BKB;
  static_assert(DMA || WN == 256, "register staging is only kept for the 256 x 256 tile");
  u32x4_t sreg[DMA ? 1 : RPW / 8][2];  // register staging (DMA == false): one 16-byte piece per 8-row group and operand
  auto stage = [&](int kt, int buf) {
    const int off = kt * BKB;
    if constexpr (DMA) {
      auto* wb = (__attribute__((address_space(3))) char*)(smem + buf * BUFB + (WRPW * w) * BKB);
      auto* xb = (__attribute__((address_space(3))) char*)(smem + buf * BUFB + WOPB + (RPW * w) * BKB);
#pragma unroll
      for (int t = 0; t < RPW / 8; ++t) {
        if (t < WRPW / 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, wb + t * 8 * BKB, 16, wvo[t & 1], (int)(t * 8 * p.w_stride) + off, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, xb + t * 8 * BKB, 16, xvo[t & 1], (int)(t * 8 * p.x_stride) + off, 0, 0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < RPW / 8; ++t) {
        sreg[t][0] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo[t & 1], (int)(t * 8 * p.w_stride) + off, 0));
        sreg[t][1] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvo[t & 1], (int)(t * 8 * p.x_stride) + off, 0));
      }
    }
  };
  auto commit = [&](int buf) {  // register staging: the pieces land where the LDS-DMA would have put them
    if constexpr (!DMA) {
      char* wb = smem + buf * BUFB + (RPW * w) * BKB + lane * 16;
#pragma unroll
      for (int t = 0; t < RPW / 8; ++t) {
        *(u32x4_t*)(wb + t * 8 * BKB) = sreg[t][0];
        *(u32x4_t*)(wb + WOPB + t * 8 * BKB) = sreg[t][1];
      }
    }
  };

  f32x4_t acc[JN][NI];
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // Software pipeline: every LDS read of slice kt is issued before the slice's single barrier, the fragments of the next
  // MFMA group are fetched while the current group runs, and the barrier sits before the LAST group, so that its wait,
  // the re-staging of the buffer just drained and the first fragment reads of slice kt + 1 all hide behind 8-16 MFMAs.
  constexpr int NG = 4;                 // MFMA groups per slice: 2 X row-tiles (32 rows) x JN column tiles each
  u32x4_t wf[DMA ? 1 : 2][JN][2], xf[3][IG][2];  // X fragments in a 3-slot rotation (LDS-DMA path: one W fragment set)
  auto load_w = [&](int buf, int ws) {
    const char* wa = smem + buf * BUFB;
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) wf[ws][j][h] = *(const u32x4_t*)(wa + lds_off(wn + 16 * j + a, 4 * h + g));
  };
  auto load_x = [&](int buf, int grp, int slot) {
    const char* xa = smem + buf * BUFB + WOPB;
#pragma unroll
    for (int i = 0; i < IG; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) xf[slot][i][h] = *(const u32x4_t*)(xa + lds_off(wm + 16 * IG * grp + 16 * i + a, 4 * h + g));
  };
  auto mma = [&](int grp, int slot, int ws) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < IG; ++i)
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        if constexpr (ES == TG_FP8) {
          mfma_mx(wf[ws][j][0], wf[ws][j][1], xf[slot][i][0], xf[slot][i][1], acc[j][IG * grp + i]);
        } else {
          mfma_chunk<ES>(wf[ws][j][0], xf[slot][i][0], acc[j][IG * grp + i]);
          mfma_chunk<ES>(wf[ws][j][1], xf[slot][i][1], acc[j][IG * grp + i]);
        }
      }
    __builtin_amdgcn_s_setprio(0);
  };
  if constexpr (DMA) {
    stage(0, 0);
    if constexpr (SILU) {   // the silu table -> LDS once per workgroup, behind the first slice's DMA
      for (int i = tid; i < kSiluLut / 8; i += NWV * 64) *(u32x4_t*)(smem + kSiluLdsOff + 16 * i) = *(const u32x4_t*)(p.silu_lut + 8 * i);
    }
    if constexpr (NBUF == 3) {   // ring of three: two slices stay in flight behind the one being multiplied
      stage(min(1, nk - 1), 1);
      stage(min(2, nk - 1), 2);
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * DMA_PER_SLICE) : "memory");
    } else {
      __syncthreads();
      stage(min(1, nk - 1), 1);
    }
    load_w(0, 0);
    load_x(0, 0, 0);
    // One slice = 4 MFMA groups g0..g3 (2 X row tiles x JN column tiles each).  X fragments rotate through three slots so
    // that TWO groups (16 MFMAs per wave) are already in registers when the slice's single barrier is reached: the
    // barrier wait, the re-staging of the drained buffer and the first fragment reads of the next slice all hide behind
    // them.  Slot roles advance by one per slice and the W fragment set alternates: six static variants, picked by kt % 6.
    auto slice = [&](int kt, int buf, auto r_) {
      constexpr int R = decltype(r_)::value;
      constexpr int S0 = R, S1 = (R + 1) % 3, S2 = (R + 2) % 3;  // g0 in S0 on entry; g1 -> S1, g2 -> S2, g3 -> S0
      const int nb = NBUF == 3 ? (buf + 1) % 3 : buf ^ 1;        // the next slice's buffer
      load_x(buf, 1, S1);
      mma(0, S0, 0);
      load_x(buf, 2, S2);
      mma(1, S1, 0);
      load_x(buf, 3, S0);
      TL_STAMP(0);  // (stamps only where the wave drains lgkmcnt anyway: an s_memtime elsewhere serialises the LDS reads)
      // slice kt + 1 has landed for every wave (vmcnt) and nobody reads buf any more (lgkmcnt)
      if constexpr (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DMA_PER_SLICE) : "memory");   // (slice kt + 2 may still be in flight)
      else __syncthreads();
      TL_STAMP(1);
      load_x(nb, 0, S1);                       // next slice's g0 -> S1 = its S0 (after the last slice: a re-staged copy, never used)
      stage(min(kt + NBUF, nk - 1), buf);      // branch free: the last iterations re-stage the final slice into drained buffers
      mma(2, S2, 0);
      mma(3, S0, 0);
      load_w(nb, 0);                           // the W fragments are free once the slice's last MFMA has issued
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    int kt = 0;
    for (; kt + 2 < nk; kt += 3) {
      slice(kt, NBUF == 3 ? 0 : kt & 1, I0{});
      slice(kt + 1, NBUF == 3 ? 1 : (kt + 1) & 1, I1{});
      slice(kt + 2, NBUF == 3 ? 2 : kt & 1, I2{});
    }
    if (kt < nk) slice(kt, NBUF == 3 ? 0 : kt & 1, I0{});
    if (kt + 1 < nk) slice(kt + 1, NBUF == 3 ? 1 : (kt + 1) & 1, I1{});
  } else {
    stage(0, 0);
    commit(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      stage(min(kt + 1, nk - 1), buf ^ 1);  // global loads of the next slice fly during this slice's MFMAs
      load_w(buf, 0);
      load_x(buf, 0, 0);
#pragma unroll
      for (int grp = 0; grp < NG; ++grp) {
        if (grp + 1 < NG) load_x(buf, grp + 1, (grp + 1) & 1);
        mma(grp, grp & 1, 0);
      }
      commit(buf ^ 1);
      __syncthreads();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may be in flight when the workgroup ends

  // ---- epilogue: acc[j][i][r] -> row m0+wm+16i+a, col n0+wn+16j+4g+r ----
  if constexpr (SILU) {
    __syncthreads();  // every wave is past its last fragment read and every staged slice has landed: the LDS is free
    epilogue_silu_lds<OutT>(p, acc, m0, wm, n0, wn, smem, tid, lane);
    return;
  }
  if constexpr (JN == 4) {   // 128 x 64 wave tiles: rows staged through the wave's own 16 KiB of LDS, written as whole 128-byte rows
    if (p.N % 8 == 0 && p.y_stride % 8 == 0 && ((uintptr_t)p.y & 15) == 0) {
      __syncthreads();  // every wave is past its last fragment read and every staged slice has landed: the LDS is free
      epilogue_scaled_lds<OutT, NI, JN>(p, acc, m0 + wm, n0 + wn, smem + w * (MR * kEpiRowB), lane);
      return;
    }
  }
  epilogue_scaled<OutT, NI, JN>(p, acc, m0 + wm + a, n0 + wn + 4 * g);
}

// ---------------------------------------------------------------------------------------------------------
// PING-PONG form of the 256x256 fp8 kernel (round 5; the guide's "256^2 8-phase" structure on the block-scaled 16x16x128 loop).
// Same tile, wave layout (2 x 4 waves of 128 x 64 outputs), LDS image, staging instructions, k order per output (bit-identical
// results) and epilogues as fp8_gemm256_kernel; what differs is the schedule.  A K slice is four PHASES (one 32-row group of the
// wave's X rows each: 8 MFMAs = 256 cycles of the matrix pipe), every phase is
//     LOAD: fragment reads of the phase (+ the slice's W fragments in phase 0) | this wave's share of the staging | B1 | lgkmcnt(0)
//     | 8 MFMAs at priority 1 | B2
// and waves 4-7 run ONE BARRIER behind waves 0-3: while one wave of a SIMD is in its MFMA block the other is in its LOAD part, so
// the two never compete for the pipe and neither's LDS round trip or barrier wait leaves it idle (round 4's counters: all eight
// waves in one phase, 27-31 % of wave cycles parked at the slice's single barrier, 48-50 % stalled at issue behind the partner).
// Buffers: two slices of [W 32 KiB | X 32 KiB], restaged REGION BY REGION as soon as a region's last reader is through -- W of slice
// s + 2 in phase 2 of slice s, wave c's 32 X rows (read by its group in phase c) in the phase after -- so every DMA has 1.5-1.75
// slices to land, as in the one-barrier kernel.  vmcnt: counted, once per slice (phase 3), never 0 in the loop.
// What bounds the loop now (timing-only builds, -DSGL_PP_NOWAIT / NOREAD / NOSTAGE; K = 14 336, M = 65 536): without the counted waits
// 2 711 vs 2 701 us, without the fragment reads 2 767 vs 2 796, WITHOUT THE IN-LOOP STAGING 2 223 vs 2 796 -- the delivery of 64 KiB per
// slice and CU out of the L2 (and what it costs in power) is a fifth of the loop; the same bytes through registers (plain loads issued a
// slice ahead, ds_write_b128 at the DMA form's issue points; built, bit-identical) are 15-25 % SLOWER than LDS-DMA, as the guide says.
template <typename OutT, bool SILU = false, int PH = 2, int ES = TG_FP8, int WLATE = 0>   // PH: phases per K slice (2 blocks of 16 MFMAs -- the default -- or 4 of 8); ES as
                                                                                          // fp8_gemm256_kernel; WLATE: X fragments of the second block issued in front of the W staging
__global__ __launch_bounds__(512, 1) void fp8_gemm256pp_kernel(const GemmParams p) {
  constexpr int NWV = 8, WN = 256, WCOLS = 4, JN = 4, MR = 128, NI = 8, IG = 2;
  constexpr int RPW = 32, WRPW = 32;          // staging rows per wave and operand
  constexpr int WOPB = WN * BKB, BUFB = WOPB + OPB;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][W | X]
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int q = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
  const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + bid / 8;
  const int GM = p.group_m;
  const int per_group = GM * p.tiles_n;
  const int grp_ = wgid / per_group, in_grp = wgid - grp_ * per_group;
  const int gsz = min(p.tiles_m - grp_ * GM, GM);
  const int tm = grp_ * GM + in_grp % gsz, tn = in_grp / gsz;
  const int m0 = tm * T2, n0 = tn * WN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (w / WCOLS) * MR, wn = (w % WCOLS) * (16 * JN);
  const int c = w & 3;           // this wave's 32 X rows are the ones its group reads in phase c
  const bool late = w >= 4;      // waves 4-7: one barrier behind
  if (p.stagger_q > 0 && (int)blockIdx.x < p.stagger_cus) {   // phase stagger of the CUs' first tiles (see fp8_gemm256_kernel)
    const int mask = (p.stagger_q & 64) ? 31 : 15;
    const int steps = ((blockIdx.x >> 3) & mask) * (p.stagger_q & 63);
    for (int t = 0; t < steps; ++t) __builtin_amdgcn_s_sleep(8);
  }
  const int a = lane & 15, g = lane >> 4;
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)((int64_t)p.N * p.w_stride), 0x00020000);
  const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)((int64_t)p.M * p.x_stride), 0x00020000);
  unsigned wvo[2], xvo[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int rl = lane >> 3;
    const int chunk = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);
    wvo[par] = (unsigned)((int64_t)(n0 + WRPW * w + rl) * p.w_stride) + chunk * 16;
    xvo[par] = (unsigned)((int64_t)(m0 + RPW * w + rl) * p.x_stride) + chunk * 16;
  }
  const int nk = p.kbytes / BKB;
  auto stage_w = [&](int kt, int buf) {
#ifdef SGL_PP_NOSTAGE   // timing-only build (wrong results): no staging inside the loop
    if (kt > 1) return;
#endif
    const int off = kt * BKB;
    auto* wb = (__attribute__((address_space(3))) char*)(smem + buf * BUFB + (WRPW * w) * BKB);
#pragma unroll
    for (int t = 0; t < WRPW / 8; ++t) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, wb + t * 8 * BKB, 16, wvo[t & 1], (int)(t * 8 * p.w_stride) + off, 0, 0);
  };
  auto stage_x = [&](int kt, int buf) {
#ifdef SGL_PP_NOSTAGE
    if (kt > 1) return;
#endif
    const int off = kt * BKB;
    auto* xb = (__attribute__((address_space(3))) char*)(smem + buf * BUFB + WOPB + (RPW * w) * BKB);
#pragma unroll
    for (int t = 0; t < RPW / 8; ++t) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, xb + t * 8 * BKB, 16, xvo[t & 1], (int)(t * 8 * p.x_stride) + off, 0, 0);
  };

  f32x4_t acc[JN][NI];
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  constexpr int XG = 4 / PH;   // 32-row X groups per phase
  u32x4_t wf[JN][2], xf[XG * IG][2];
  auto load_w = [&](int buf) {
#ifdef SGL_PP_NOREAD   // timing-only build (wrong results): fragments are read once
    if (buf) return;
#endif
    const char* wa = smem + buf * BUFB;
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) wf[j][h] = *(const u32x4_t*)(wa + lds_off(wn + 16 * j + a, 4 * h + g));
  };
  auto load_x = [&](int buf, int grp, int slot = 0) {
#ifdef SGL_PP_NOREAD
    if (buf) return;
#endif
    const char* xa = smem + buf * BUFB + WOPB;
#pragma unroll
    for (int i = 0; i < IG; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) xf[slot * IG + i][h] = *(const u32x4_t*)(xa + lds_off(wm + 16 * IG * grp + 16 * i + a, 4 * h + g));
  };
  // (the MFMA builtin touches no memory and hipcc moves such calls across raw s_barriers -- it gathered the four blocks of a slice behind
  // the last barrier: the phase's X fragments are made opaque in front of the block and its accumulators behind it, which ties the
  // block to the volatile statements, i.e. to the two barriers, around it)
  auto mma = [&](int grp0, int i0 = 0, int i1 = 4) {   // the phase's XG groups grp0 .. grp0 + XG - 1: X fragments i0 .. i1 - 1 of them (at most XG IG <= 4)
#pragma unroll
    for (int i = 0; i < XG * IG; ++i)
      if (i >= i0 && i < i1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(xf[i][h]));
      }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < XG * IG; ++i)
      if (i >= i0 && i < i1) {
#pragma unroll
        for (int j = 0; j < JN; ++j) {
          if constexpr (ES == TG_FP8) {
            mfma_mx(wf[j][0], wf[j][1], xf[i][0], xf[i][1], acc[j][IG * grp0 + i]);
          } else {   // 16-bit operands: two 16x16x32 k-steps per slice, in the one-barrier kernel's order
            mfma_chunk<ES>(wf[j][0], xf[i][0], acc[j][IG * grp0 + i]);
            mfma_chunk<ES>(wf[j][1], xf[i][1], acc[j][IG * grp0 + i]);
          }
        }
      }
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int i = 0; i < XG * IG; ++i)
      if (i >= i0 && i < i1) {
#pragma unroll
        for (int j = 0; j < JN; ++j) asm volatile("" : "+v"(acc[j][IG * grp0 + i]));
      }
  };

  stage_w(0, 0);
  stage_x(0, 0);
  if constexpr (SILU) {   // the silu table -> LDS once per workgroup, behind the first slice's DMA
    for (int i = tid; i < kSiluLut / 8; i += NWV * 64) *(u32x4_t*)(smem + kSiluLdsOff + 16 * i) = *(const u32x4_t*)(p.silu_lut + 8 * i);
  }
  stage_w(1, 1);
  stage_x(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (late) __builtin_amdgcn_s_barrier();   // waves 4-7 fall one barrier behind; waves 0-3 take the matching one after the loop

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  if constexpr (PH == 4) {
  // one phase.  P: 0..3; the staging of the phase goes between the fragment reads and the first barrier (the MFMA block of the other
  // group is running: the DMA issue costs it nothing)
  auto phase = [&](int s, int buf, auto p_) {
    constexpr int P = decltype(p_)::value;
    if constexpr (P == 0) load_w(buf);
    load_x(buf, P);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (P == 3) {
      // slice s + 1 must have landed before anyone reads it in the next phase; still allowed in flight: this wave's W (and, for the
      // waves whose X rows free early, X) of slice s + 2
      if (c < 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    if constexpr (P == 2) stage_w(min(s + 2, nk - 1), buf);              // W of this buffer: every wave read it in phase 0
    if (c == (P + 3) % 4) {                                              // this wave's X rows were read in phase c = P - 1
      if constexpr (P == 0) { if (s > 0) stage_x(min(s + 1, nk - 1), buf ^ 1); }   // (rows of slice s - 1, read in its phase 3)
      else stage_x(min(s + 2, nk - 1), buf);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);   // (MFMAs touch no memory: without the pins hipcc gathers the blocks of several phases behind one barrier)
    mma(P);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int s = 0; s < nk; ++s) {
    const int buf = s & 1;
    phase(s, buf, I0{});
    phase(s, buf, I1{});
    phase(s, buf, I2{});
    phase(s, buf, I3{});
  }
  } else {
  // two phases per slice: blocks of 16 MFMAs, half as many barriers.  Q0 reads W and X groups 0, 1 (the rows of waves c = 0, 1), Q1
  // groups 2, 3.  Restaging: wave c < 2 puts its X rows of slice s + 2 in Q1 of slice s, wave c >= 2 in Q0 of slice s + 1; W of slice
  // s + 2 goes behind Q1's first barrier (by then both groups' W reads of slice s have retired).  Every DMA has one slice to land.
  auto phase2 = [&](int s, int buf, auto q_) {
    constexpr int Q = decltype(q_)::value;
    if constexpr (Q == 0) load_w(buf);
    load_x(buf, 2 * Q, 0);
    load_x(buf, 2 * Q + 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Q == 1) {
      // W and the X rows of groups 0, 1 of slice s + 1 are read two barriers from here: this wave's share must have landed.  Waves
      // c >= 2 may still have their X rows of slice s + 1 (issued in Q0 of this slice, read in Q1 of the next) in flight.
#ifndef SGL_PP_NOWAIT   // timing-only build (wrong results): how much of the loop is the counted wait
      if (c < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#endif
      if (c < 2) stage_x(min(s + 2, nk - 1), buf);
    } else {
      // (waves c >= 2) the X rows of slice s issued a slice ago are read in Q1: landed by now; W of slice s + 1 may be in flight
      if (c >= 2) {
#ifndef SGL_PP_NOWAIT
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#endif
        if (s > 0) stage_x(min(s + 1, nk - 1), buf ^ 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Q == 1 && WLATE > 0) {
      // W of slice s + 2: behind this barrier (both groups' W reads of slice s have retired) and BEHIND the block's first MFMAs -- in
      // front of them the four DMA issues hold the matrix pipe idle once per slice (one DMA behind each quarter of the block instead: no
      // further difference)
      mma(2 * Q, 0, WLATE);
      __builtin_amdgcn_sched_barrier(0);
      stage_w(min(s + 2, nk - 1), buf);
      __builtin_amdgcn_sched_barrier(0);
      mma(2 * Q, WLATE, XG * IG);
    } else {
      if constexpr (Q == 1) stage_w(min(s + 2, nk - 1), buf);
    // (measured and dropped: the closing barrier in FRONT of the block's last 4 / 8 MFMAs, so that the hand-over of the pipe overlaps
    // them -- every hazard-relevant operation lies before the block, so it is legal: +3.6 / +9.5 % slower.  The exclusivity is the point.)
      mma(2 * Q);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int s = 0; s < nk; ++s) {
    const int buf = s & 1;
    phase2(s, buf, I0{});
    phase2(s, buf, I1{});
  }
  }
  if (!late) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may be in flight when the workgroup ends

  if constexpr (SILU) {
    __syncthreads();
    epilogue_silu_lds<OutT>(p, acc, m0, wm, n0, wn, smem, tid, lane);
    return;
  }
  if (p.N % 8 == 0 && p.y_stride % 8 == 0 && ((uintptr_t)p.y & 15) == 0) {
    __syncthreads();
    epilogue_scaled_lds<OutT, NI, JN>(p, acc, m0 + wm, n0 + wn, smem + w * (MR * kEpiRowB), lane);
    return;
  }
  epilogue_scaled<OutT, NI, JN>(p, acc, m0 + wm + a, n0 + wn + 4 * g);
}

// ---------------------------------------------------------------------------------------------------------
// PERSISTENT form of the 256x256 kernel for launches of many tiles per CU (prefill): one workgroup per CU walks its XCD's tile list
// (tile j, j + W, j + 2 W, ... for the W workgroups of an XCD: the order the dispatcher produces for equal tile times, so the
// grouped L2 reuse is the same), and the NEXT tile's first K slice is staged while the current tile's last slice is multiplied:
// slice nk - 2's drained buffer takes it, the last slice stages nothing, the epilogue goes through the buffer the last slice
// leaves (64 KiB: the SiLU act tile as is, the plain output in two passes of 64 rows per wave).  What it removes per tile: the
// workgroup hand-over on the CU, the first slice's full fetch latency with nothing to overlap, the table copy of the SiLU form.
// Same k order and epilogue arithmetic per output: the same bits as the one-tile-per-workgroup kernel.
// ---------------------------------------------------------------------------------------------------------
template <typename OutT, int ES = TG_FP8, bool SILU = false>
__global__ __launch_bounds__(512, 1) void fp8_gemm256p_kernel(const GemmParams p) {
  constexpr int NWV = 8, WCOLS = 4, JN = 4, RPW = 32, BUFB = 2 * OPB;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][W 32 KiB | X 32 KiB] (+ the silu table)
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int q = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
  const int base = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;   // this XCD's tiles: base .. base + cnt - 1
  const int cnt = q + (xcd < r8 ? 1 : 0);
  const int per = (int)gridDim.x / 8;   // workgroups per XCD (the launcher makes the grid a multiple of 8)
  int ti = bid / 8;
  if (ti >= cnt) return;
  const int GM = p.group_m;
  const int per_group = GM * p.tiles_n;
  auto origin = [&](int t, int& m0, int& n0) {   // grouped order, as in fp8_gemm256_kernel
    const int wgid = base + t;
    const int grp = wgid / per_group, in_grp = wgid - grp * per_group;
    const int gsz = min(p.tiles_m - grp * GM, GM);
    m0 = (grp * GM + in_grp % gsz) * T2;
    n0 = (in_grp / gsz) * T2;
  };
  int m0, n0;
  origin(ti, m0, n0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (w / WCOLS) * 128, wn = (w % WCOLS) * (16 * JN);
  if (p.stagger_q > 0 && (int)blockIdx.x < p.stagger_cus) {   // phase stagger of the CUs' first tiles (see fp8_gemm256_kernel)
    const int mask = (p.stagger_q & 64) ? 31 : 15;
    const int steps = ((blockIdx.x >> 3) & mask) * (p.stagger_q & 63);
    for (int t = 0; t < steps; ++t) __builtin_amdgcn_s_sleep(8);
  }
  const int a = lane & 15, g = lane >> 4;

  const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)((int64_t)p.N * p.w_stride), 0x00020000);
  const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)((int64_t)p.M * p.x_stride), 0x00020000);
  // per-lane offsets within a tile (the same for every tile: two VGPRs per operand); the tile's origin travels in the scalar offset
  unsigned wvo[2], xvo[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int rl = RPW * w + (lane >> 3);
    const int chunk = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);
    wvo[par] = (unsigned)((int64_t)rl * p.w_stride) + chunk * 16;
    xvo[par] = (unsigned)((int64_t)rl * p.x_stride) + chunk * 16;
  }
  const int nk = p.kbytes / BKB;   // >= 3 (launcher)
  // wo / xo: byte offsets of the tile's first W / X row (launcher: operands < 2 GiB), in the SCALAR offset.  The range check of gfx950
  // covers vector + scalar offset (tools/microbench/buffer_soffset_check.hip: a load whose vector offset is inside the descriptor's
  // range and whose sum with the scalar offset is not returns zeros, LDS-DMA form included), so a ragged last tile still gets zeros
  // for its rows past N / M and nothing behind the operand is touched.
  auto stage_at = [&](int wo, int xo, int kt, int buf) {
    const int off = kt * BKB;
    auto* wb = (__attribute__((address_space(3))) char*)(smem + buf * BUFB + (RPW * w) * BKB);
#pragma unroll
    for (int t = 0; t < RPW / 8; ++t) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, wb + t * 8 * BKB, 16, wvo[t & 1], wo + (int)(t * 8 * p.w_stride) + off, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, wb + OPB + t * 8 * BKB, 16, xvo[t & 1], xo + (int)(t * 8 * p.x_stride) + off, 0, 0);
    }
  };
  int wo0 = (int)((int64_t)n0 * p.w_stride), xo0 = (int)((int64_t)m0 * p.x_stride), wo1 = wo0, xo1 = xo0;   // this tile / the next one

  f32x4_t acc[JN][8];
  u32x4_t wf[JN][2], xf[3][2][2];
  auto load_w = [&](int buf) {
    const char* wa = smem + buf * BUFB;
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) wf[j][h] = *(const u32x4_t*)(wa + lds_off(wn + 16 * j + a, 4 * h + g));
  };
  auto load_x = [&](int buf, int grp, int slot) {
    const char* xa = smem + buf * BUFB + OPB;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) xf[slot][i][h] = *(const u32x4_t*)(xa + lds_off(wm + 32 * grp + 16 * i + a, 4 * h + g));
  };
  auto mma = [&](int grp, int slot) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        if constexpr (ES == TG_FP8) {
          mfma_mx(wf[j][0], wf[j][1], xf[slot][i][0], xf[slot][i][1], acc[j][2 * grp + i]);
        } else {
          mfma_chunk<ES>(wf[j][0], xf[slot][i][0], acc[j][2 * grp + i]);
          mfma_chunk<ES>(wf[j][1], xf[slot][i][1], acc[j][2 * grp + i]);
        }
      }
    __builtin_amdgcn_s_setprio(0);
  };
  // one K slice (see fp8_gemm256_kernel); TAIL 0: inside the tile, 1: slice nk - 2 (its drained buffer takes the next tile's slice
  // 0), 2: the last slice (stages nothing and fetches no fragments: its buffer is the epilogue's)
  auto slice = [&](int kt, int buf, auto r_, auto tail_, auto pin_) {
    constexpr int R = decltype(r_)::value, TAIL = decltype(tail_)::value;
    constexpr bool PIN = decltype(pin_)::value != 0;
    constexpr int S0 = R, S1 = (R + 1) % 3, S2 = (R + 2) % 3;
    // (outside the k loop the slices are straight-line code up to the epilogue, one scheduling region: left alone, the scheduler
    // hoists every fragment read of the last three slices to the top and parks the fragments in scratch -- pin the order there)
#define SGL_PIN() do { if constexpr (PIN) __builtin_amdgcn_sched_barrier(0); } while (0)
    load_x(buf, 1, S1);
    SGL_PIN();
    mma(0, S0);
    SGL_PIN();
    load_x(buf, 2, S2);
    SGL_PIN();
    mma(1, S1);
    SGL_PIN();
    load_x(buf, 3, S0);
    __syncthreads();
    if constexpr (TAIL != 2) load_x(buf ^ 1, 0, S1);
    if constexpr (TAIL == 0) stage_at(wo0, xo0, kt + 2, buf);
    if constexpr (TAIL == 1) stage_at(wo1, xo1, 0, buf);
    SGL_PIN();
    mma(2, S2);
    mma(3, S0);
    SGL_PIN();
    if constexpr (TAIL != 2) load_w(buf ^ 1);
    SGL_PIN();
#undef SGL_PIN
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  int b0 = 0;   // the buffer that holds slice 0 of the current tile
  stage_at(wo0, xo0, 0, 0);
  if constexpr (SILU) {   // the silu table -> LDS once per workgroup, behind the first slice's DMA
    for (int i = tid; i < kSiluLut / 8; i += NWV * 64) *(u32x4_t*)(smem + kSiluLdsOff + 16 * i) = *(const u32x4_t*)(p.silu_lut + 8 * i);
  }
  // Dynamic tile schedule (round 4; measured in round 3 at -1.7 ... -5.8 % over the static one, not kept then for want of per-launch
  // storage): the first tile of a workgroup is static (tile bid / 8 of its XCD's range); every later one is `per_eff` + a ticket drawn
  // from the XCD class's counter -- the workgroups that finish early take more tiles, the order of tiles inside the XCD's range (and
  // with it the grouped L2 reuse) is unchanged.  Tickets are drawn ONE TILE AHEAD by thread 0, at the point where the tile's second
  // slice is staged: the returning atomic rides under the same wait as that DMA; it reaches the other waves through two words of LDS
  // and the barriers the tile loop already has.  No workgroup ever waits for another: a ticket past the range ends the loop.
  __shared__ int sched_l[2];
  const int per_eff = min(per, cnt);
  int next_ti = ti + per;
  if (p.sched != nullptr && tid == 0) sched_l[0] = per_eff + __hip_atomic_fetch_add(p.sched + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (p.sched != nullptr) next_ti = sched_l[0];
  int iter = 0;
  for (;;) {
    stage_at(wo0, xo0, 1, b0 ^ 1);
    int ticket = 0;
    if (p.sched != nullptr && tid == 0) ticket = __hip_atomic_fetch_add(p.sched + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the tile after next
    const bool has_next = next_ti < cnt;
    int m1 = m0, n1 = n0;
    if (has_next) origin(next_ti, m1, n1);
    wo1 = (int)((int64_t)n1 * p.w_stride);   // (no next tile: this tile's own slice 0 once more, into a drained buffer nobody reads)
    xo1 = (int)((int64_t)m1 * p.x_stride);
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // nk = 3 G + r slices: the r odd ones come FIRST, at rotation phases 3 - r .. 2, so that the groups of three -- and with them the
    // tile's last two slices, which differ (TAIL) -- sit at static phases 0, 1, 2.  (With the odd slices at the end the tail needed
    // one copy per value of r; the three-way join cost the register allocator a thousand spills.)
    const int r = nk % 3, G = nk / 3;   // G >= 1 (launcher)
    load_w(b0);
    load_x(b0, 0, 0);   // the first slice's phase is 0, 2 or 1 (r = 0, 1, 2): its X fragments go to all three slots -- eight extra LDS reads
    load_x(b0, 0, 2);   // per tile, and every slot is defined on every path (a slot loaded on one branch only would be carried around
    load_x(b0, 0, 1);   // the tile loop, through the epilogue, as a live value)
    int kt = 0;
    if (r == 2) {
      slice(kt, (kt + b0) & 1, I1{}, I0{}, I0{});
      ++kt;
    }
    if (r >= 1) {
      slice(kt, (kt + b0) & 1, I2{}, I0{}, I0{});
      ++kt;
    }
    for (int gi = 0; gi + 1 < G; ++gi, kt += 3) {
      slice(kt, (kt + b0) & 1, I0{}, I0{}, I0{});
      slice(kt + 1, (kt + 1 + b0) & 1, I1{}, I0{}, I0{});
      slice(kt + 2, (kt + b0) & 1, I2{}, I0{}, I0{});
    }
    slice(kt, (kt + b0) & 1, I0{}, I0{}, I1{});
    slice(kt + 1, (kt + 1 + b0) & 1, I1{}, I1{}, I1{});
    slice(kt + 2, (kt + b0) & 1, I2{}, I2{}, I1{});
    // ---- epilogue through the last slice's buffer; the other one holds (or is receiving) the next tile's slice 0 ----
    char* ebuf = smem + ((nk - 1 + b0) & 1) * BUFB;
    // The accumulators are "used" here: otherwise the IR-level sinking pass moves the last slices' MFMAs down to their first use
    // inside the epilogue's branches (the SiLU form's table / exact-expression split), below the barrier, and the fragments they
    // read -- loaded above it -- travel there through scratch (225 spilled dwords, the MFMA blocks of the tail slices empty).
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(acc[j][i]));
    __syncthreads();  // every wave is past its last fragment read
    // the epilogue's per-lane addressing is the same for every tile; derived from an opaque copy of the lane id so that it is NOT
    // hoisted out of the tile loop (hoisted, it stayed live across the main loop and was spilled: scratch reloads between the
    // LDS-DMA issues, each dragging a vmcnt(0) behind it)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    if constexpr (SILU) {
      epilogue_silu_lds<OutT>(p, acc, m0, wm, n0, wn, smem, lane_e + 64 * w, lane_e, ebuf);
    } else {
      if (p.N % 8 == 0 && p.y_stride % 8 == 0 && ((uintptr_t)p.y & 15) == 0)
        epilogue_scaled_lds<OutT, 8, JN, 2>(p, acc, m0 + wm, n0 + wn, ebuf + w * (64 * kEpiRowB), lane_e);
      else
        epilogue_scaled<OutT, 8, JN>(p, acc, m0 + wm + (lane_e & 15), n0 + wn + 4 * (lane_e >> 4));
    }
    if (!has_next) break;
    if (p.sched != nullptr && tid == 0) sched_l[(iter + 1) & 1] = per_eff + ticket;
    ti = next_ti;
    m0 = m1;
    n0 = n1;
    wo0 = wo1;
    xo0 = xo1;
    b0 = (nk + b0) & 1;   // = the buffer of slice nk - 2
    __syncthreads();      // the epilogue's LDS reads are done: its buffer may take slice 1
    next_ti = p.sched != nullptr ? sched_l[(iter + 1) & 1] : ti + per;
    ++iter;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may be in flight when the workgroup ends
}

// ---------------------------------------------------------------------------------------------------------
// fp8 "streaming" tile for decode-sized M (64 < M <= 256, the continuous-batching regime): the weights are read once and
// that read is the whole cost, so what matters is bytes in flight per CU, not MFMA rate.  128x128 tile, 128-byte K slices,
// LDS-DMA into a FOUR-deep ring of 32 KiB stages (three slices in flight while one is computed: a counted s_waitcnt, never
// vmcnt(0), in the loop), 8 waves (2 along M x 4 along N, 64 x 32 outputs each), block-scaled MFMA, split-K over
// blockIdx.y with f32 slabs when the launch has fewer tiles than CUs.  Same LDS image, swizzle and fragment order as the
// 256x256 kernel.
// ---------------------------------------------------------------------------------------------------------
constexpr int S_BM = 128, S_BN = 128, S_STAGES = 4;
constexpr int S_OPB = 128 * BKB;  // 16 KiB per operand and stage

template <typename OutT>
__global__ __launch_bounds__(512, 1) void fp8_gemm128s_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [4 stages][W 16 KiB | X 16 KiB]
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int q = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
  const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + bid / 8;
  // Row tiles FASTEST while they are few (M <= 1024; round 4): the row tiles of one column tile then sit next to each other inside
  // ONE XCD's contiguous range, so the W tile they all stream is fetched from HBM once and hit in that XCD's L2 by the others.  With
  // column tiles fastest (rounds 2-3) the two row tiles of M = 256 landed on XCDs x and x + 4 and every weight byte was fetched
  // twice: profiles/round4_pmc_mid_m_gemm.json, 243.6 MB fetched for the 117.4 MB gate_up matrix -- at 53 us that fetch WAS the
  // time.  X (M x K <= 4 MiB here) is small enough to be shared by every XCD.  Same arithmetic per tile: the same bits.
  int tm, tn;
  if (p.tiles_m <= 8) {
    tn = wgid / p.tiles_m;
    tm = wgid - tn * p.tiles_m;
  } else {
    tm = wgid / p.tiles_n;
    tn = wgid - tm * p.tiles_n;
  }
  const int m0 = tm * S_BM, n0 = tn * S_BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (w >> 2) * 64, wn = (w & 3) * 32;
  const int a = lane & 15, g = lane >> 4;

  // staging: wave w fills rows 16 w .. 16 w + 15 of both operands, 8 rows per instruction (see the 256x256 kernel)
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)((int64_t)p.N * p.w_stride), 0x00020000);
  const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)((int64_t)p.M * p.x_stride), 0x00020000);
  unsigned wvo[2], xvo[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int rl = 16 * w + (lane >> 3);
    const int chunk = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);
    wvo[par] = (unsigned)((int64_t)(n0 + rl) * p.w_stride) + chunk * 16;
    xvo[par] = (unsigned)((int64_t)(m0 + rl) * p.x_stride) + chunk * 16;
  }
  const int nk_all = p.kbytes / BKB;
  const int kt0 = p.slabs ? (int)blockIdx.y * p.kt_per : 0;
  const int nk = (p.slabs ? min(nk_all, kt0 + p.kt_per) : nk_all) - kt0;  // slices of this workgroup
  auto stage = [&](int i) {  // slice kt0 + min(i, nk - 1) -> stage i % 4 (past the end: a harmless re-stage, branch free)
    auto* wb = (__attribute__((address_space(3))) char*)(smem + (i % S_STAGES) * 2 * S_OPB + (16 * w) * BKB);
    const int off = (kt0 + min(i, nk - 1)) * BKB;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, wb + t * 8 * BKB, 16, wvo[t], (int)(t * 8 * p.w_stride) + off, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, wb + S_OPB + t * 8 * BKB, 16, xvo[t], (int)(t * 8 * p.x_stride) + off, 0, 0);
    }
  };

  f32x4_t acc[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  stage(0);
  stage(1);
  stage(2);
  for (int i = 0; i < nk; ++i) {
    // slice i has landed once at most the two younger stages (8 instructions of this wave) are outstanding; the barrier
    // makes every wave's pieces visible and proves nobody still reads the stage that slice i + 3 is about to overwrite
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    stage(i + 3);
    const char* wa = smem + (i % S_STAGES) * 2 * S_OPB;
    const char* xa = wa + S_OPB;
    u32x4_t wf[2][2], xf[4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) wf[j][h] = *(const u32x4_t*)(wa + lds_off(wn + 16 * j + a, 4 * h + g));
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int h = 0; h < 2; ++h) xf[ii][h] = *(const u32x4_t*)(xa + lds_off(wm + 16 * ii + a, 4 * h + g));
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < 2; ++j) mfma_mx(wf[j][0], wf[j][1], xf[ii][0], xf[ii][1], acc[j][ii]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may be in flight when the workgroup ends

  // ---- epilogue: acc[j][i][r] -> row m0+wm+16i+a, col n0+wn+16j+4g+r ----
  if (p.slabs) {
    const auto srs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.slabs + (int64_t)blockIdx.y * p.M * p.N), 0,
                                                       (unsigned)((int64_t)p.M * p.N * 4), 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm + 16 * i + a;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + 16 * j + 4 * g;
        const unsigned off = (m < p.M && n + 3 < p.N) ? (unsigned)(((int64_t)m * p.N + n) * 4) : 0xFFFFFFF0u;  // (N % 4 == 0 in slab mode)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[j][i]), srs, off, 0, 0);
      }
    }
    // the consumer (the reduce launch, or a kernel that takes slabs) combines them.  (Round 3 also had an in-launch combine --
    // the S workgroups of a tile waiting for each other on a counter; it measured slower than the reduce launch at every shape
    // tried (DESIGN.md section 5 "Round 3" item 3a) and was removed in round 4 together with its counters.)
    return;
  }
  epilogue_scaled<OutT, 4, 2>(p, acc, m0 + wm + a, n0 + wn + 4 * g);
}

int tg_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return cus;
}
int g_tiled_group_m = 8;   // r3 sweep (tools/debug/gm_sweep.py, M = 8192 / 16 384 / 65 536 x the four Llama-3-8B shapes): 8 is as fast as 4 or up to 5 % faster (4 until round 3)
int g_tiled_stagger = 1;  // x 1024 cycles per step, 16 steps: CUs spread over ~6.5 us
int g_tiled_force = 0;  // test hook: 1 = always the 128x128 kernel, 2 = the 256x256 kernel whenever its shape rules allow

// the ping-pong schedule (fp8_gemm256pp_kernel) for fp8 256 x 256 launches, one tile per workgroup: 1 (default) = every such launch with
// two phases per K slice -- measured (tools/debug/pingpong_ab.py, profiles/round5_ab_gemm_pingpong.json, M = 65 536, against round 4's
// default = the one-barrier schedule, persistent where that won): K = 14336 -9.6 %, K = 4096 x N 6144 / 4096 / 28672 -3.1 / -3.1 / -3.3 %,
// the SiluAndMul form -3.7 %; with four phases per slice (mode 2) -5.4 % at K = 14336 and a tie at K = 4096.  0 = round 4's choice.
// (force_tile 5000 / 5001 / 5002 / 5003: off / default / four phases / two phases with the W staging in front of the second block)
int g_tiled_pingpong = 1;
template <typename OutT, int NWV, bool DMA = true, int ES = TG_FP8, bool SILU = false, int WN = 256>
int launch256(GemmParams& p, hipStream_t st) {
  constexpr int smem = (WN == 128 ? 3 : 2) * (WN * BKB + OPB) + (SILU ? kSiluLut * 2 : 0);  // 128 KiB (+ 18.5 KiB: the silu table); 3 x 48 KiB for 256 x 128 tiles
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)fp8_gemm256_kernel<OutT, NWV, DMA, ES, SILU, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  p.tiles_m = (p.M + T2 - 1) / T2;
  p.tiles_n = (p.N + WN - 1) / WN;
  p.group_m = g_tiled_group_m;
  p.stagger_cus = tg_cus();
  p.stagger_q = (p.tiles_m * p.tiles_n >= 2 * p.stagger_cus) ? g_tiled_stagger : 0;  // needs a second round to pay off
  if constexpr (NWV == 8 && DMA && WN == 256) {
    if (p.kbytes >= 4 * BKB && g_tiled_pingpong >= 1) {
#define SGL_PP_LAUNCH(PHV, WL)                                                                                                         \
  do {                                                                                                                                \
    static bool set_ = false;                                                                                                         \
    if (!set_) {                                                                                                                      \
      (void)hipFuncSetAttribute((const void*)fp8_gemm256pp_kernel<OutT, SILU, PHV, ES, WL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); \
      set_ = true;                                                                                                                    \
    }                                                                                                                                 \
    hipLaunchKernelGGL((fp8_gemm256pp_kernel<OutT, SILU, PHV, ES, WL>), dim3(p.tiles_m * p.tiles_n), dim3(512), smem, st, p);          \
  } while (0)
      if (g_tiled_pingpong == 2) SGL_PP_LAUNCH(4, 0);        // measurement hooks: four phases per slice;
      else if (g_tiled_pingpong == 3) SGL_PP_LAUNCH(2, 0);   // two, W staged in front of the second block (-0.3 ... -0.9 % behind its first four MFMAs)
      else SGL_PP_LAUNCH(2, 1);
#undef SGL_PP_LAUNCH
      SGL_HIP_LAUNCH_CHECK();
      return SGL_MI355_OK;
    }
  }
  hipLaunchKernelGGL((fp8_gemm256_kernel<OutT, NWV, DMA, ES, SILU, WN>), dim3(p.tiles_m * p.tiles_n), dim3(NWV * 64), smem, st, p);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

int g_tiled_persistent = 1;  // measurement hook (force_tile 3000 / 3001): 0 = one tile per workgroup always
int g_tiled_dynamic = 1;  // measurement hook (force_tile 4000 / 4001): 0 = the static tile schedule even where per-launch counters were given
template <typename OutT, int ES = TG_FP8, bool SILU = false>
int launch256p(GemmParams& p, hipStream_t st, int* sched = nullptr) {
  const int cus = tg_cus() / 8 * 8;
  const int64_t tiles = (int64_t)((p.M + T2 - 1) / T2) * ((p.N + T2 - 1) / T2);
  if (!g_tiled_dynamic) sched = nullptr;
  {   // round 5: the ping-pong schedule, one tile per workgroup, beats the one-barrier forms at every shape measured (fp8 and 16-bit operands).
    // (A persistent form of it -- the slice stream running on across tiles, the epilogue through the last slice's buffer -- was built,
    // bit-identical, and measured -1.8 ... +2.6 % against this at M = 65 536: profiles/round5_ab_gemm_pingpong_persistent.log.  Removed.)
    // Very short K keeps the persistent one-barrier kernel where that applies (TP-8 shard shapes at M = 65 536: K = 512 183 vs 215 us,
    // K = 1792 426 vs 435 us: there the tile boundary is most of a tile and hiding the prologue is worth more than the schedule).
    if (p.kbytes >= 4 * BKB && (g_tiled_pingpong >= 2 || (g_tiled_pingpong == 1 && p.kbytes >= 2048))) {
      return launch256<OutT, 8, true, ES, SILU>(p, st);
    }
  }
  if constexpr (SILU) {
    // the SiluAndMul form measured a TIE between the static persistent schedule and one tile per workgroup (round 3); it takes the
    // persistent kernel only with the dynamic schedule, i.e. when the caller gave per-launch counters
    if (sched == nullptr) return launch256<OutT, 8, true, ES, SILU>(p, st);
  }
  // (measured, tools/debug/persistent_256.py / persistent_silu.py, M = 65 536: K = 4096 x N = 6144 / 4096 / 28672 -5 % / -5 % / -2.5 %, K = 384-640
  // -10...-16 %; K = 14336 a tie (the hand-over is 4 % of a tile there and the static tile schedule gives up the dispatcher's load
  // balancing); the SiLU form a tie as well -- both stay on one tile per workgroup)
  if (!g_tiled_persistent || (p.kbytes > 8192 && !(sched != nullptr && g_tiled_dynamic >= 2)) || cus < 8 || tiles < 2 * (int64_t)cus || p.kbytes < 3 * BKB || (int64_t)p.N * p.w_stride >= (1ll << 31) ||
      (int64_t)p.M * p.x_stride >= (1ll << 31))   // (the tile origin travels in a signed 32-bit scalar offset)
    return launch256<OutT, 8, true, ES, SILU>(p, st);
  constexpr int smem = 2 * 2 * OPB + (SILU ? kSiluLut * 2 : 0);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)fp8_gemm256p_kernel<OutT, ES, SILU>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  p.tiles_m = (p.M + T2 - 1) / T2;
  p.tiles_n = (p.N + T2 - 1) / T2;
  p.group_m = g_tiled_group_m;
  p.stagger_cus = cus;
  p.stagger_q = g_tiled_stagger;
  p.sched = sched;
  if (sched != nullptr) {   // the counters start every launch at zero whatever an aborted launch left there (a memset node: capturable)
    const hipError_t e = hipMemsetAsync(sched, 0, 8 * sizeof(int), st);
    SGL_CHECK(e == hipSuccess, "fp8_gemm: hipMemsetAsync of the tile counters failed: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL((fp8_gemm256p_kernel<OutT, ES, SILU>), dim3(cus), dim3(512), smem, st, p);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// y[m][n] = OutT((sum_s slabs[s][m][n]) * sx[m] * sw[n] + bias[n]): the same arithmetic as the fused epilogue
template <typename OutT>
__global__ __launch_bounds__(256) void tiled_splitk_reduce_kernel(const float* __restrict__ slabs, int S, const float* sx,
                                                                  const float* sw, const OutT* bias, OutT* y, int64_t y_stride,
                                                                  int M, int N) {
  const int64_t total = (int64_t)M * (N / 4);
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int m = (int)(idx / (N / 4)), n = (int)(idx - (int64_t)m * (N / 4)) * 4;
    f32x4_t v = *(const f32x4_t*)(slabs + (int64_t)m * N + n);
    for (int sI = 1; sI < S; ++sI) v += *(const f32x4_t*)(slabs + ((int64_t)sI * M + m) * N + n);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float r = v[c];
      if (sx) r *= sx[m];
      r = r * (sw ? sw[n + c] : 1.0f) + (bias ? (float)bias[n + c] : 0.0f);
      y[(int64_t)m * y_stride + n + c] = (OutT)r;
    }
  }
}

int tg_cus();

template <int ES, typename OutT>
int launch(GemmParams& p, hipStream_t st, float* workspace = nullptr, int64_t workspace_floats = 0) {
  constexpr int smem = 2 * 2 * TILE_BYTES;  // 64 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)tiled_gemm_kernel<ES, OutT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  // Few output tiles (decode at 64 < M <= 256, the continuous-batching regime: the weights are streamed once and a tile per
  // CU is all there is): split K over blockIdx.y so every CU streams a share of W; raw sums meet in f32 slabs.
  const int tiles = p.tiles_m * p.tiles_n, nk = (p.kbytes + BKB - 1) / BKB, cus = tg_cus();
  int splits = 1;
  if (workspace != nullptr && tiles < cus && p.N % 4 == 0) {
    splits = (2 * cus) / tiles;  // two resident workgroups per CU
    if (splits > 8) splits = 8;
    if (splits > nk / 8) splits = nk / 8;  // at least 8 slices (1 KiB of K) per workgroup: more splits only move the cost into slabs
    while (splits > 1 && (int64_t)splits * p.M * p.N > workspace_floats) --splits;
  }
  if (splits > 1) {
    p.kt_per = (nk + splits - 1) / splits;
    splits = (nk + p.kt_per - 1) / p.kt_per;
    p.slabs = workspace;
    hipLaunchKernelGGL((tiled_gemm_kernel<ES, OutT>), dim3(tiles, splits), dim3(256), smem, st, p);
    SGL_HIP_LAUNCH_CHECK();
    const int64_t items = (int64_t)p.M * (p.N / 4);
    const unsigned blocks = (unsigned)((items + 255) / 256 > 4096 ? 4096 : (items + 255) / 256);
    hipLaunchKernelGGL((tiled_splitk_reduce_kernel<OutT>), dim3(blocks), dim3(256), 0, st, workspace, splits, p.sx, p.sw,
                       (const OutT*)p.bias, (OutT*)p.y, p.y_stride, p.M, p.N);
    SGL_HIP_LAUNCH_CHECK();
    return SGL_MI355_OK;
  }
  p.slabs = nullptr;
  p.kt_per = 0;
  hipLaunchKernelGGL((tiled_gemm_kernel<ES, OutT>), dim3(tiles), dim3(256), smem, st, p);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// split-K ranges of the streaming 128x128 tile for this shape (1: none) and the slices per range
// floats of a caller's workspace that may hold split-K slabs: all but the ticket ring in its last 512 words (gemm_entry)
inline int64_t slab_capacity(int64_t workspace_floats) { return workspace_floats >= 4096 ? workspace_floats - 512 : workspace_floats; }

inline int splits128s(int M, int N, int kbytes, bool have_workspace, int64_t workspace_floats, int* kt_per) {
  const int tiles = ((M + S_BM - 1) / S_BM) * ((N + S_BN - 1) / S_BN), nk = kbytes / BKB, cus = tg_cus();
  int splits = 1;
  if (have_workspace && tiles < cus && N % 4 == 0) {
    splits = cus / tiles;                  // one 128 KiB workgroup per CU
    if (splits > nk / 8) splits = nk / 8;  // at least 8 slices (1 KiB of K) per workgroup: more splits only move the cost into slabs
    if (splits < 1) splits = 1;
    while (splits > 1 && (int64_t)splits * M * N > workspace_floats) --splits;
  }
  if (splits > 1) {
    *kt_per = (nk + splits - 1) / splits;
    return (nk + *kt_per - 1) / *kt_per;
  }
  *kt_per = nk;
  return 1;
}

// slabs_only: leave the raw f32 partial sums [splits][M][N] in `workspace` for a consumer kernel that combines them
// (sgl_mi355_fused_add_rmsnorm_quant_fp8), instead of running the reduce kernel into p.y; requires splits > 1.
template <typename OutT>
int launch128s(GemmParams& p, hipStream_t st, float* workspace, int64_t workspace_floats, bool slabs_only = false) {
  constexpr int smem = S_STAGES * 2 * S_OPB;  // 128 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)fp8_gemm128s_kernel<OutT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  p.tiles_m = (p.M + S_BM - 1) / S_BM;
  p.tiles_n = (p.N + S_BN - 1) / S_BN;
  const int tiles = p.tiles_m * p.tiles_n;
  const int splits = splits128s(p.M, p.N, p.kbytes, workspace != nullptr, workspace_floats, &p.kt_per);
  p.slabs = splits > 1 ? workspace : nullptr;
  SGL_CHECK(!slabs_only || splits > 1, "fp8_gemm_slabs: this shape runs as one k-range (no slabs): M=%d N=%d", p.M, p.N);
  hipLaunchKernelGGL((fp8_gemm128s_kernel<OutT>), dim3(tiles, splits), dim3(512), smem, st, p);
  SGL_HIP_LAUNCH_CHECK();
  if (splits > 1 && !slabs_only) {
    const int64_t items = (int64_t)p.M * (p.N / 4);
    const unsigned blocks = (unsigned)((items + 255) / 256 > 4096 ? 4096 : (items + 255) / 256);
    hipLaunchKernelGGL((tiled_splitk_reduce_kernel<OutT>), dim3(blocks), dim3(256), 0, st, workspace, splits, p.sx, p.sw,
                       (const OutT*)p.bias, (OutT*)p.y, p.y_stride, p.M, p.N);
    SGL_HIP_LAUNCH_CHECK();
  }
  return SGL_MI355_OK;
}

// Which tile does sgl_mi355_fp8_gemm give this fp8 problem?  The streaming 128x128 tile always at decode-sized M (<= 256: the weights
// are read once and bytes in flight decide).  Above that the 256x256 tile wins whenever it fills the chip, but below that lies a band
// where it leaves CUs idle: 256 < M <= ~1024 with few 256-wide tiles, where the streaming tile (with split-K when even its tiles are
// fewer than CUs) is up to 2x faster (tools/debug/mid_m.py, round 3: M = 1024, N = 4096, K = 14336: 79 us against 167 us), and the
// chunked-prefill band M ~ 1024-4096 with N <= 8192, where the 256x128 tile (eight waves of 64x64 outputs, the same slice pipeline
// over a ring of three 48 KiB buffers) fills the chip with half-size tiles (tools/debug/tile_256x128.py, round 3: M = 2048, N = 4096,
// K = 4096: 33 us against 46 us for 256x256 and 51 us for streaming; K = 14336: 109 us against 168 / 150 us).  The choice is the
// smallest of three measured cost lines per 4096 bytes of K: one round of 256x256 tiles takes 46-50 us, one round of 256x128 tiles
// 29-32 us, one round of streaming tiles 24 us (a partly filled last round of those costs about half its share: fewer CUs contend for
// L2 / HBM).  Launches of many rounds never reach the comparison: the larger tile has the higher arithmetic intensity.
enum TileChoice { kTile256 = 0, kTile128s = 1, kTile256x128 = 2, kTileOld = 3 };
inline TileChoice choose_tile(int M, int N, int kbytes, int64_t x_stride_b, int64_t w_stride_b, bool have_workspace = true,
                              int64_t workspace_floats = (1ll << 40)) {
  const bool can256 = kbytes % BKB == 0 && kbytes >= BKB && (int64_t)N * w_stride_b < 0xFFFFFFF0ll && (int64_t)M * x_stride_b < 0xFFFFFFF0ll;
  if (!can256) return kTile256;   // (run() sends these to the old 128x128 kernel)
  if (g_tiled_force == 5) return kTile128s;
  if (g_tiled_force == 7) return kTile256x128;
  if (g_tiled_force != 0) return kTile256;
  if (M <= 256) return kTile128s;
  const double cus = tg_cus(), k4 = kbytes / 4096.0;
  const int64_t tm256 = (M + T2 - 1) / T2;
  const int64_t t256 = tm256 * ((N + T2 - 1) / T2), tnar = tm256 * ((N + 127) / 128), t128 = (int64_t)((M + S_BM - 1) / S_BM) * ((N + S_BN - 1) / S_BN);
  if (t128 > 8 * (int64_t)cus) return kTile256;   // many rounds either way
  int kt_per = 0;
  const int splits = splits128s(M, N, kbytes, have_workspace, workspace_floats, &kt_per);
  const double r = (double)t128 * splits / cus, rounds = r <= 1.0 ? 1.0 : 0.5 * (ceil(r) + r);
  const double cost128 = rounds * 24.0 * k4 / splits + (splits > 1 ? 7.0 : 0.0);
  const double cost256 = ceil((double)t256 / cus) * 46.0 * k4;   // (49 until round 5: the ping-pong schedule; M = 2048, N = 6144: one round in 46.4 us)
  const double costnar = ceil((double)tnar / cus) * 31.0 * k4;
  if (cost128 < cost256 && cost128 <= costnar) return kTile128s;
  return costnar < cost256 ? kTile256x128 : kTile256;
}
// The same for 16-bit operands (dense_gemm), per 8192 bytes of K (tools/debug/tile_256x128_dense.py, round 3): one round of 256x256
// tiles takes 85-92 us, one round of 256x128 tiles 53-57 us, the 128x128 kernel (kTileOld: split-K when its tiles are fewer than half
// the CUs, where it stays) about 20 + 47 r us for r = tiles / CUs >= 0.5.  Until round 3 the 256x256 kernel ran only with at least one
// tile per CU: M = 2048, N = 6144, K = 4096 took 150 us on the 128x128 kernel against 92 us.
inline TileChoice choose_tile16(int M, int N, int kbytes, int64_t x_stride_b, int64_t w_stride_b) {
  const bool can256 = kbytes % BKB == 0 && kbytes >= BKB && (int64_t)N * w_stride_b < 0xFFFFFFF0ll && (int64_t)M * x_stride_b < 0xFFFFFFF0ll;
  if (!can256 || g_tiled_force == 1) return kTileOld;
  if (g_tiled_force == 2) return kTile256;
  if (g_tiled_force == 7) return kTile256x128;
  const double cus = tg_cus(), k8 = kbytes / 8192.0;
  const int64_t tm256 = (M + T2 - 1) / T2;
  const int64_t t256 = tm256 * ((N + T2 - 1) / T2), tnar = tm256 * ((N + 127) / 128), t128 = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const double r = (double)t128 / cus;
  if (M <= 256 || r < 0.5) return kTileOld;
  if (t128 > 8 * (int64_t)cus) return kTile256;
  const double cost_old = (20.0 + 47.0 * r) * k8;
  const double cost256 = ceil((double)t256 / cus) * 84.0 * k8;   // (88 until round 5: the ping-pong schedule, -4 ... -9 % on 16-bit operands)
  const double costnar = ceil((double)tnar / cus) * 55.0 * k8;
  if (cost_old < cost256 && cost_old <= costnar) return kTileOld;
  return costnar < cost256 ? kTile256x128 : kTile256;
}
inline bool takes128s(int M, int N, int kbytes, int64_t x_stride_b, int64_t w_stride_b, bool have_workspace = true,
                      int64_t workspace_floats = (1ll << 40)) {
  return choose_tile(M, N, kbytes, x_stride_b, w_stride_b, have_workspace, workspace_floats) == kTile128s;
}

int run(const void* x, int64_t xs, const void* w, int64_t ws, void* y, int64_t ys, const float* sx, const float* sw,
        const void* bias, int M, int N, int K, int in_dtype, int out_dtype, void* stream, const char* who,
        float* workspace = nullptr, int64_t workspace_floats = 0) {
  SGL_CHECK(M >= 0 && N >= 0 && K >= 0, "%s: negative shape", who);
  if (M == 0 || N == 0) return SGL_MI355_OK;
  SGL_CHECK(x && w && y, "%s: null pointer", who);
  SGL_CHECK(out_dtype == SGL_BF16 || out_dtype == SGL_F16, "%s: out_dtype must be bf16 or f16", who);
  const int es = in_dtype == SGL_FP8_E4M3 ? 1 : 2;
  SGL_CHECK((K * es) % 16 == 0 && (xs * es) % 16 == 0 && (ws * es) % 16 == 0 && ((uintptr_t)x % 16) == 0 &&
                ((uintptr_t)w % 16) == 0,
            "%s: rows must be 16-byte aligned (K=%d)", who, K);
  SGL_CHECK((int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN) < (1ll << 31), "%s: grid too large", who);
  GemmParams p;
  p.x = (const char*)x; p.x_stride = xs * es;
  p.w = (const char*)w; p.w_stride = ws * es;
  p.y = y; p.y_stride = ys;
  p.sx = sx; p.sw = sw; p.bias = bias;
  p.M = M; p.N = N; p.kbytes = K * es;
  hipStream_t st = (hipStream_t)stream;
  // per-launch tile counters of the persistent kernel's dynamic schedule: one of 64 eight-word slots in the LAST 512 words of the
  // caller's workspace, taken round robin (zeroed by the launcher's memset node) -- launches that overlap on two streams of a process
  // therefore do not share tickets unless 64 further persistent GEMMs were launched in between (the split-K slabs at the workspace's
  // head have no such ring: one workspace must not serve two streams at once, include/sgl_mi355.h)
  static std::atomic<unsigned> sched_slot{0};
  int* sched = (workspace != nullptr && workspace_floats >= 4096)
                   ? (int*)(workspace + workspace_floats - 512) + 8 * (sched_slot.fetch_add(1, std::memory_order_relaxed) & 63u)
                   : nullptr;
  // split-K slabs are sized against the workspace WITHOUT those 512 words (round 4 sized them against all of it: a slab set that
  // filled the buffer overlapped the ticket words, and a later persistent launch's memset would have zeroed live partial sums)
  workspace_floats = slab_capacity(workspace_floats);
  if (in_dtype == SGL_FP8_E4M3) {
    // both LDS-DMA kernels want whole 128-byte K slices
    const bool can256 = p.kbytes % BKB == 0 && p.kbytes >= BKB && (int64_t)N * p.w_stride < 0xFFFFFFF0ll &&
                        (int64_t)M * p.x_stride < 0xFFFFFFF0ll;  // 32-bit buffer offsets
    const TileChoice tile = choose_tile(M, N, p.kbytes, p.x_stride, p.w_stride, workspace != nullptr, workspace_floats);
    // decode-sized M: the streaming tile (weights read once; bytes in flight decide)
    if (tile == kTile128s)
      return out_dtype == SGL_BF16 ? launch128s<__bf16>(p, st, workspace, workspace_floats)
                                   : launch128s<_Float16>(p, st, workspace, workspace_floats);
    if (tile == kTile256x128)
      return out_dtype == SGL_BF16 ? launch256<__bf16, 8, true, TG_FP8, false, 128>(p, st) : launch256<_Float16, 8, true, TG_FP8, false, 128>(p, st);
    if (can256 && g_tiled_force == 4) return out_dtype == SGL_BF16 ? launch256<__bf16, 8, false>(p, st) : launch256<_Float16, 8, false>(p, st);
    if (can256 && g_tiled_force == 3) return out_dtype == SGL_BF16 ? launch256<__bf16, 4>(p, st) : launch256<_Float16, 4>(p, st);
    // (the old 128x128 kernel is left with the shapes neither LDS-DMA kernel accepts: K not whole 128-byte slices, > 4 GiB operands)
    if (can256 && g_tiled_force != 1)
      return out_dtype == SGL_BF16 ? launch256p<__bf16>(p, st, sched) : launch256p<_Float16>(p, st, sched);
    return out_dtype == SGL_BF16 ? launch<TG_FP8, __bf16>(p, st, workspace, workspace_floats)
                                 : launch<TG_FP8, _Float16>(p, st, workspace, workspace_floats);
  }
  {  // 16-bit inputs: the LDS-DMA kernels (256x256 / 256x128 tiles) for whole 128-byte K slices, by measured cost lines
    const TileChoice tile = choose_tile16(M, N, p.kbytes, p.x_stride, p.w_stride);
    if (tile == kTile256x128) {
      if (in_dtype == SGL_BF16)
        return out_dtype == SGL_BF16 ? launch256<__bf16, 8, true, TG_BF16, false, 128>(p, st) : launch256<_Float16, 8, true, TG_BF16, false, 128>(p, st);
      return out_dtype == SGL_BF16 ? launch256<__bf16, 8, true, TG_F16, false, 128>(p, st) : launch256<_Float16, 8, true, TG_F16, false, 128>(p, st);
    }
    if (tile == kTile256) {
      if (in_dtype == SGL_BF16)
        return out_dtype == SGL_BF16 ? launch256p<__bf16, TG_BF16>(p, st, sched) : launch256p<_Float16, TG_BF16>(p, st, sched);
      return out_dtype == SGL_BF16 ? launch256p<__bf16, TG_F16>(p, st, sched) : launch256p<_Float16, TG_F16>(p, st, sched);
    }
  }
  if (in_dtype == SGL_BF16)
    return out_dtype == SGL_BF16 ? launch<TG_BF16, __bf16>(p, st, workspace, workspace_floats)
                                 : launch<TG_BF16, _Float16>(p, st, workspace, workspace_floats);
  return out_dtype == SGL_BF16 ? launch<TG_F16, __bf16>(p, st, workspace, workspace_floats)
                               : launch<TG_F16, _Float16>(p, st, workspace, workspace_floats);
}

// the table of silu_lut.h in global memory (one copy per device: __device__ variables are per-device), filled by four 256-thread
// workgroups (~5 us with the launch) once per device -- see sgl_mi355_internal_tiled_gemm_silu_mul
__device__ uint16_t g_silu_lut[kSiluLut];
__global__ __launch_bounds__(256) void silu_lut_fill_kernel() {
  for (int i = threadIdx.x + 256 * blockIdx.x; i < kSiluLut; i += 256 * gridDim.x) g_silu_lut[i] = silu_lut_entry(i);
}
constexpr int kMaxSiluDevices = 64;
std::atomic<bool> g_silu_table_ready[kMaxSiluDevices];   // set by sgl_mi355_silu_table_init once the fill has COMPLETED

}  // namespace

// Per-device initialisation of the SiluAndMul epilogue's table (csrc/silu_lut.h): runs the fill kernel on `stream` of the CURRENT
// device and WAITS for it (a one-off outside any capture) -- the table is ready for every stream of the device when this returns.
// (Round 4 set the flag when the fill was enqueued: a C-ABI caller on another stream could pass the readiness check and read a
// partly filled table)  Idempotent; refused under stream capture (a captured fill would run at replay, not now).
extern "C" int sgl_mi355_silu_table_init(void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int dev = -1;
  SGL_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxSiluDevices, "silu_table_init: cannot identify the current device");
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  SGL_CHECK(hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone,
            "silu_table_init: must not be called while the stream is capturing");
  if (g_silu_table_ready[dev].load(std::memory_order_acquire)) return SGL_MI355_OK;
  hipLaunchKernelGGL(silu_lut_fill_kernel, dim3(4), dim3(256), 0, st);
  SGL_HIP_LAUNCH_CHECK();
  if (hipStreamSynchronize(st) != hipSuccess) {
    snprintf(g_sgl_mi355_err, sizeof(g_sgl_mi355_err), "silu_table_init: the fill kernel did not complete");
    return SGL_MI355_EHIP;
  }
  g_silu_table_ready[dev].store(true, std::memory_order_release);
  return SGL_MI355_OK;
}

// gate_up_proj + SiluAndMul for M > 64 (called by sgl_mi355_gemm_silu_mul, skinny_gemm.hip): fp8 operands, bf16 out, 16-row
// interleaving, N % 256 == 0, whole 128-byte K slices.  Returns SGL_MI355_EINVAL with a message otherwise.
int sgl_mi355_internal_tiled_gemm_silu_mul(const void* x, int64_t x_stride_b, const void* w, int64_t w_stride_b, void* act,
                                           int64_t act_stride, const float* sx, const float* sw, int M, int N, int K, hipStream_t st,
                                           int* sched) {
  SGL_CHECK(M > 0 && N > 0 && N % 256 == 0 && K % BKB == 0 && K >= BKB, "gemm_silu_mul: the prefill form needs N %% 256 == 0 and K %% 128 == 0 (N=%d K=%d)", N, K);
  SGL_CHECK((int64_t)N * w_stride_b < 0xFFFFFFF0ll && (int64_t)M * x_stride_b < 0xFFFFFFF0ll, "gemm_silu_mul: operands above 4 GiB");
  SGL_CHECK(act_stride % 8 == 0 && ((uintptr_t)act & 15) == 0, "gemm_silu_mul: act rows must be 16-byte aligned");
  GemmParams p;
  p.x = (const char*)x; p.x_stride = x_stride_b;
  p.w = (const char*)w; p.w_stride = w_stride_b;
  p.y = act; p.y_stride = act_stride;
  p.sx = sx; p.sw = sw; p.bias = nullptr;
  p.M = M; p.N = N; p.kbytes = K;
  p.slabs = nullptr; p.kt_per = 0;
  void* lut = nullptr;
  if (hipGetSymbolAddress(&lut, HIP_SYMBOL(g_silu_lut)) != hipSuccess || lut == nullptr) {
    snprintf(g_sgl_mi355_err, sizeof(g_sgl_mi355_err), "gemm_silu_mul: cannot resolve the silu table");
    return SGL_MI355_EINVAL;
  }
  p.silu_lut = (const uint16_t*)lut;
  // The table is a constant of the device, filled by sgl_mi355_silu_table_init (an explicit per-device init entry point: this
  // launcher neither synchronises nor keeps launch-time state, like every other one).
  int dev = -1;
  SGL_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxSiluDevices, "gemm_silu_mul: cannot identify the current device");
  SGL_CHECK(g_silu_table_ready[dev].load(std::memory_order_acquire),
            "gemm_silu_mul: the silu table of device %d is not initialised -- call sgl_mi355_silu_table_init(stream) once per device "
            "(outside stream capture) and order later work after it", dev);
  // one tile per workgroup, or -- given per-launch counters -- the persistent kernel on the dynamic tile schedule (launch256p decides)
  return launch256p<__bf16, TG_FP8, true>(p, st, sched);
}

extern "C" int sgl_mi355_fp8_gemm_force_tile(int mode) {
  if (mode >= 3000 && mode < 4000) {  // measurement hook: 3001 = persistent 256x256 kernel where a launch has at least two tiles per CU (default), 3000 = never
    g_tiled_persistent = mode - 3000;
    return SGL_MI355_OK;
  }
  if (mode >= 5000) {  // measurement hook: 5001 = ping-pong schedule for fp8 256 x 256 launches (default: two phases per slice), 5002 = four phases, 5003 = two, 5000 = round 4's kernels
    g_tiled_pingpong = mode - 5000;
    return SGL_MI355_OK;
  }
  if (mode >= 4000) {  // measurement hook: 4001 = dynamic tile schedule where counters are available (default), 4000 = static schedule
    g_tiled_dynamic = mode - 4000;
    return SGL_MI355_OK;
  }
  SGL_CHECK(mode < 2000 || mode >= 3000, "fp8_gemm_force_tile: the 2000 range (in-launch split-K combine) was removed in round 4");
  if (mode >= 1000) {  // measurement hook: 1000 + q sets the start-stagger quantum of the 256x256 kernel (0 = off)
    g_tiled_stagger = mode - 1000;
    return SGL_MI355_OK;
  }
  if (mode >= 100) {  // measurement hook: 100 + GM sets the scheduling group height of the 256x256 kernel
    g_tiled_group_m = mode - 100 > 0 ? mode - 100 : 1;
    return SGL_MI355_OK;
  }
  g_tiled_force = mode;
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_fp8_gemm(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems, void* y,
                                  int64_t y_stride_elems, const float* scales_x, const float* scales_w, const void* bias,
                                  int M, int N, int K, int out_dtype, float* workspace, int64_t workspace_floats,
                                  void* stream) {
  return run(x, x_stride_elems, w, w_stride_elems, y, y_stride_elems, scales_x, scales_w, bias, M, N, K, SGL_FP8_E4M3,
             out_dtype, stream, "fp8_gemm", workspace, workspace_floats);
}

// Which kernel sgl_mi355_fp8_gemm picks for contiguous rows (host logic, no GPU work): 0 = 256x256 tile (or, for K that is not whole
// 128-byte slices, the old 128x128 kernel), 1 = streaming 128x128 tile, 2 = 256x128 tile.
extern "C" int sgl_mi355_fp8_gemm_tile_choice(int M, int N, int K, int64_t workspace_floats) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (int)choose_tile(M, N, K, K, K, true, slab_capacity(workspace_floats));
}

// How many f32 [M, N] slabs sgl_mi355_fp8_gemm sums for this shape when given `workspace_floats` of scratch (1: it runs as one
// k-range; > 1 only where takes128s() picks the streaming tile and its tiles are fewer than CUs).  Contiguous rows assumed.
extern "C" int sgl_mi355_fp8_gemm_num_slabs(int M, int N, int K, int64_t workspace_floats) {
  workspace_floats = slab_capacity(workspace_floats);
  if (M <= 0 || N <= 0 || K <= 0 || !takes128s(M, N, K, K, K, true, workspace_floats)) return 1;
  int kt_per = 0;
  return splits128s(M, N, K, true, workspace_floats, &kt_per);
}

// Producer half of the launch-boundary split-K reduce for 64 < M <= 256: the raw f32 partial sums [num_slabs][M][N] of
// fp8_scaled_mm, no scales (the consumer, sgl_mi355_fused_add_rmsnorm_quant_fp8 with slabs, applies sx[m] * sw[n]).  The k-range
// partition is sgl_mi355_fp8_gemm's for the same `workspace_floats`, so both sum identically; fails unless num_slabs > 1.
extern "C" int sgl_mi355_fp8_gemm_slabs(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems, float* slabs,
                                        int M, int N, int K, int64_t workspace_floats, void* stream) {
  SGL_CHECK(x && w && slabs && M > 0 && N > 0 && K > 0, "fp8_gemm_slabs: bad arguments");
  SGL_CHECK(K % 16 == 0 && x_stride_elems % 16 == 0 && w_stride_elems % 16 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0,
            "fp8_gemm_slabs: rows must be 16-byte aligned (K=%d)", K);
  workspace_floats = slab_capacity(workspace_floats);
  SGL_CHECK(takes128s(M, N, K, x_stride_elems, w_stride_elems, true, workspace_floats), "fp8_gemm_slabs: M=%d N=%d K=%d is not a streaming-tile shape", M, N, K);
  GemmParams p;
  p.x = (const char*)x; p.x_stride = x_stride_elems;
  p.w = (const char*)w; p.w_stride = w_stride_elems;
  p.y = nullptr; p.y_stride = 0;
  p.sx = nullptr; p.sw = nullptr; p.bias = nullptr;
  p.M = M; p.N = N; p.kbytes = K;
  return launch128s<__bf16>(p, (hipStream_t)stream, slabs, workspace_floats, true);
}

extern "C" int sgl_mi355_dense_gemm(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems, void* y,
                                    int64_t y_stride_elems, const void* bias, int M, int N, int K, int in_dtype,
                                    int out_dtype, float* workspace, int64_t workspace_floats, void* stream) {
  if (!(in_dtype == SGL_BF16 || in_dtype == SGL_F16)) {
    snprintf(g_sgl_mi355_err, sizeof(g_sgl_mi355_err), "dense_gemm: in_dtype must be bf16 or f16");
    return SGL_MI355_EINVAL;
  }
  return run(x, x_stride_elems, w, w_stride_elems, y, y_stride_elems, nullptr, nullptr, bias, M, N, K, in_dtype, out_dtype,
             stream, "dense_gemm", workspace, workspace_floats);
}
