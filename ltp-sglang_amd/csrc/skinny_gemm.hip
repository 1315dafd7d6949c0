// Weight-streaming "skinny-M" GEMM for decode (M <= 64): Y[M,N] = (X[M,K] . W[N,K]^T) * sx[m] * sw[n] + bias[n]
//
// Replaces on the decode path:
//   * fp8_scaled_mm (sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1071-1146; python gemm.py:34-42):
//     mat_a [M,K] e4m3 row-major, mat_b [K,N] e4m3 column-major (== W[N,K] row-major),
//     scales_a [M] f32 per token, scales_b [N] f32 per channel, optional bias[N], out bf16/f16;
//   * the unquantised bf16/f16 linear (lm_head, layers/quantization/unquant.py) with sx = sw = null.
//
// At M <= 64 the GEMM is a pure weight stream (decode: 6.98 GB of fp8 weights per step at
// Llama-3-8B) so the design is the same as the decode-attention K path: every wave owns whole
// 16-row x 1 KiB chunks of W, gathers them with 16-byte loads where the 64 lanes of one
// instruction cover 1 KiB of ONE weight row (fully coalesced), stages them through a
// wave-private XOR-swizzled LDS image (no workgroup barrier in the main loop) and feeds
// v_mfma_f32_16x16x32_{fp8_fp8,bf16,f16}.  X is tiny (M*K bytes, L2 resident) and is read
// straight into MFMA A-fragments.  The 4 waves of a workgroup split K; one LDS reduction and
// the fused scale/bias epilogue finish the tile.
#include "common.h"
#include "gemm_epilogue.h"

int sgl_mi355_internal_tiled_gemm_silu_mul(const void* x, int64_t x_stride_b, const void* w, int64_t w_stride_b, void* act,
                                           int64_t act_stride, const float* sx, const float* sw, int M, int N, int K, hipStream_t st,
                                           int* sched);

namespace {

struct SkinnyParams {
  const char* x;
  int64_t x_stride;  // bytes per row of X
  const char* w;
  int64_t w_stride;  // bytes per row of W (row n holds K contiguous elements)
  void* y;
  int64_t y_stride;  // elements per row of Y
  const float* sx;   // [M] or null
  const float* sw;   // [N] or null
  const void* bias;  // [N] in the output dtype, or null
  int M, N, K;
  int kbytes;  // K * element size
#ifdef SGL_SKINNY_TIMELINE
  long long* tl;  // tools/microbench/skinny_timeline.hip: s_memtime stamps [workgroup][wave][8]
  int x_tiled = 0;  // experiment (decode_gemm_timeline): X laid out [k-slice of a wave][row][bytes of the slice] instead of row-major
#endif
};

enum { ES_FP8 = 0, ES_BF16 = 1, ES_F16 = 2 };

#ifdef SGL_SKINNY_TIMELINE
#define SK_STAMP(i)                                                                  \
  do {                                                                               \
    const long long t_ = (long long)__builtin_amdgcn_s_memrealtime();  /* 100 MHz */ \
    if (lane == 0) p.tl[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + w) * 8 + (i)] = t_; \
  } while (0)
#else
#define SK_STAMP(i)
#endif

template <int ES>
struct MfmaOp;
template <>
struct MfmaOp<ES_FP8> {
  static __device__ __forceinline__ void run(const u32x4_t& xa, const u32x4_t& wb, f32x4_t& acc) {
    // 16 bytes = two k-steps of 8 fp8 per lane
    const long a0 = ((long)xa[1] << 32) | (long)xa[0], a1 = ((long)xa[3] << 32) | (long)xa[2];
    const long b0 = ((long)wb[1] << 32) | (long)wb[0], b1 = ((long)wb[3] << 32) | (long)wb[2];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, acc, 0, 0, 0);
  }
};
template <>
struct MfmaOp<ES_BF16> {
  static __device__ __forceinline__ void run(const u32x4_t& xa, const u32x4_t& wb, f32x4_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, xa), __builtin_bit_cast(bf16x8_t, wb), acc, 0, 0, 0);
  }
};
template <>
struct MfmaOp<ES_F16> {
  static __device__ __forceinline__ void run(const u32x4_t& xa, const u32x4_t& wb, f32x4_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, xa), __builtin_bit_cast(f16x8_t, wb), acc, 0, 0, 0);
  }
};

bool g_skinny_force_v1 = false;  // test hook: exercise the generic kernel on shapes the v2 kernel would take

constexpr int kChunkB = 1024;  // bytes of one weight row per chunk = one wave-wide 16-B load
constexpr int kRows = 16;      // weight rows (output columns) per workgroup
constexpr int kWaves = 4;

template <int ES, int MT, typename OutT>
__global__ __launch_bounds__(kWaves * 64, 2) void skinny_gemm_kernel(const SkinnyParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * kRows;
  char* wl = smem + w * (kRows * kChunkB);

  const int nchunks = (p.kbytes + kChunkB - 1) / kChunkB;
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};

  f32x4_t acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // row pointers: clamp rows past N (their outputs are never stored)
  const char* wrow[kRows];
#pragma unroll
  for (int i = 0; i < kRows; ++i) wrow[i] = p.w + (int64_t)min(n0 + i, p.N - 1) * p.w_stride + lane * 16;
  const char* xrow[MT];
  bool xok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + a;
    xok[mt] = m < p.M;
    xrow[mt] = p.x + (int64_t)min(m, p.M - 1) * p.x_stride + g * 16;
  }

  u32x4_t wreg[kRows];
  auto issue = [&](int c) {
    const int off = c * kChunkB;
    const bool ok = off + lane * 16 < p.kbytes;
#pragma unroll
    for (int i = 0; i < kRows; ++i) wreg[i] = ok ? *(const u32x4_t*)(wrow[i] + off) : zero4;
  };

  int c = w;
  if (c < nchunks) issue(c);
  for (; c < nchunks; c += kWaves) {
#pragma unroll
    for (int i = 0; i < kRows; ++i) *(u32x4_t*)(wl + i * kChunkB + ((lane ^ i) << 4)) = wreg[i];
    if (c + kWaves < nchunks) issue(c + kWaves);
    const int off = c * kChunkB;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int chunk = 4 * s + g;  // 16-byte chunk of the row: bytes [64 s + 16 g, +16)
      const u32x4_t wf = *(const u32x4_t*)(wl + a * kChunkB + ((chunk ^ a) << 4));
      const int xoff = off + 64 * s;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bool ok = xok[mt] && (xoff + g * 16 < p.kbytes);
        const u32x4_t xf = ok ? *(const u32x4_t*)(xrow[mt] + xoff) : zero4;
        MfmaOp<ES>::run(xf, wf, acc[mt]);
      }
    }
  }

  // cross-wave K reduction + epilogue.  acc[mt][r]: m = 16 mt + 4 g + r, n = a
  __syncthreads();
  float* red = (float*)smem;  // [kWaves][MT*16][16]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(w * MT * 16 + mt * 16 + 4 * g + r) * 16 + a] = acc[mt][r];
  __syncthreads();
  for (int o = tid; o < MT * 16 * 16; o += kWaves * 64) {
    const int m = o >> 4, n = o & 15;
    if (m < p.M && n0 + n < p.N) {
      float v = 0.f;
#pragma unroll
      for (int ww = 0; ww < kWaves; ++ww) v += red[(ww * MT * 16 + m) * 16 + n];
      if (p.sx) v *= p.sx[m];
      if (p.sw) v *= p.sw[n0 + n];
      if (p.bias) v += (float)((const OutT*)p.bias)[n0 + n];
      ((OutT*)p.y)[(int64_t)m * p.y_stride + n0 + n] = (OutT)v;
    }
  }
}

template <int ES, int MT, typename OutT>
int launch(const SkinnyParams& p, hipStream_t st) {
  constexpr int smem = kWaves * kRows * kChunkB;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)skinny_gemm_kernel<ES, MT, OutT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((skinny_gemm_kernel<ES, MT, OutT>), dim3((p.N + kRows - 1) / kRows), dim3(kWaves * 64), smem, st, p);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}


// ---------------------------------------------------------------------------------------------------------
// v2: "X-stationary" persistent form (the decode path's kernel).  One 512-thread workgroup per CU.  K is cut into
// k-ranges of 8 * KW bytes (KW = 64 * DS bytes per wave); wave w of a workgroup owns the slice
// [kr * 8 KW + w * KW, + KW) of k-range kr for the whole launch, so its X fragments are staged ONCE
// (coalesced rows -> swizzled LDS image -> MT x DS A-fragment registers) and never re-read.  The workgroup walks
// output tiles of `rpt` (8 or 16) weight rows; each wave streams its slice of the rows with coalesced buffer loads,
// PD register sets ahead of the math, re-lays them through its private LDS image and runs MT x DS (x2 for fp8)
// MFMAs; the 8 K-slice partials of TPP tiles meet in LDS between two barriers.
//   * one k-range (K bytes <= 8 KW): every thread finishes one output element with the fused scale/bias epilogue;
//   * several k-ranges (down_proj K = 14336, bf16 lm_head): blockIdx.y = k-range, raw f32 partial sums go to slab
//     [kr][M][N] and are combined by the consumer (sgl_mi355_splitk_reduce, or fused into the next RMSNorm) -- the
//     launch-boundary split-K reduce.
// Notes from measurements on MI355X (tools/bench_skinny.py, tools/microbench/stream_patterns.hip):
//  * global loads must be COALESCED (the lanes of a quarter wave read one contiguous >= 128-byte run of a row) and
//    re-laid into MFMA fragment order through LDS; fragment-shaped global loads (16 rows x 64 B per instruction) are
//    bound by the texture-address path, for W and for the one-off X fragments;
//  * the loop body is BRANCH-FREE around memory operations (conditional loads make hipcc's wait-count insertion
//    fall back to s_waitcnt vmcnt(0)); tiles past the workgroup's last one become out-of-range buffer offsets,
//    which the hardware answers with zeros without touching memory;
//  * waves run free for TPP tiles between workgroup barriers; a barrier per tile made every tile wait for the
//    slowest of the 8 waves' loads.
// ---------------------------------------------------------------------------------------------------------
constexpr int kV2Waves = 8;
#ifndef SGL_SKINNY_PD
#define SGL_SKINNY_PD 0
#endif
int g_skinny_slab_pd2 = 0;  // measurement hook (force_generic(4 / 5)): 1 = slab-mode launches keep TWO tiles in flight per wave (measured slower)
int g_skinny_allin = 1;  // measurement hook (sgl_mi355_skinny_gemm_force_generic(2 / 3)): 0 = the one-tile-ahead form for 8-row tiles too
constexpr int g_skinny_pd_test = SGL_SKINNY_PD;  // (A/B hook, tools/build_variant.sh: force the prefetch depth of the epilogue-fused launches)

// R8 (round 4): an instantiation for 8-row tiles only (rpt == 8).  The general kernel issues DS load instructions per tile
// whatever the tile height (the upper half re-reads row lr: an L1 hit, but a staging register); with R8 a tile is DS / 2 loads and
// DS / 2 registers per lane, which lets a workgroup with <= 3 tiles (qkv_proj of Llama-3-8B: 768 tiles of 8 rows over 256
// workgroups) request ALL of them before the first is consumed (PD = TPP = 3: ONE HBM round trip instead of three dependent
// 32 KiB ones).  Same k order per wave, same cross-wave sum: the same bits.
template <int ES, int MT, int DS, int PD, int TPP, typename OutT, int EPI = EPI_NONE, int NWV = kV2Waves, bool R8 = false>
__global__ __launch_bounds__(NWV * 64, 1) void skinny_gemm_v2_kernel(const SkinnyParams p, int rpt, int ntiles,
                                                                          float* slabs, const EpiParams ep = EpiParams{}) {
  constexpr int kw = DS * 64;             // bytes of K per wave
  constexpr int LPR = kw / 16;            // lanes per weight row in one load instruction
  constexpr int RPI = 64 / LPR;           // rows per load instruction
  constexpr int NLD = 16 / RPI;           // load instructions per 16-row tile (== DS)
  constexpr int IMG = 16 * kw;            // bytes of one wave's LDS image (16 rows x kw)
  static_assert(NLD == DS, "one 16-byte load per lane and k double-step");
  static_assert(TPP == 1 || TPP % PD == 0, "static slot indices");
  constexpr int EPT = (MT * 256 + NWV * 64 - 1) / (NWV * 64);  // output elements per thread and tile
  __shared__ __attribute__((aligned(16))) char wimg[NWV * IMG];
  __shared__ float red[TPP][NWV][MT * 16][16];
  constexpr bool SCALED = (ES == ES_FP8);  // fp8_scaled_mm always carries both scale vectors
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = lane & 15, g = lane >> 4;
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};
  char* wl = wimg + w * IMG;
  SK_STAMP(0);  // kernel entry
  const int lc = lane % LPR, lr = lane / LPR;  // staging: 16-byte chunk lc of row lr + RPI * i
  const int kr = blockIdx.y;
  const int koff = kr * (NWV * kw) + w * kw + lc * 16;  // this lane's byte offset inside a row
  const bool kok = koff < p.kbytes;                          // K tail: lanes past the row end contribute zeros

  const int G = gridDim.x;
  const int cnt = (ntiles - (int)blockIdx.x + G - 1) / G;  // tiles of this workgroup: blockIdx.x + j * G, >= 1
  constexpr int WN = R8 ? DS / 2 : DS;                     // staging registers (load instructions) per tile
  const int nload = (R8 || rpt != 16) ? DS / 2 : DS;       // 8-row tiles load half the instructions
  // this thread's output elements of every tile are (m = tid / 16 + e * NWV * 4, n = tid % 16)
  const int em0 = tid >> 4, en = tid & 15;
  const bool has_bias = p.bias != nullptr;
  const OutT* biasp = has_bias ? (const OutT*)p.bias : (const OutT*)p.w;  // any readable address when absent
  float sxv[EPT];
  int64_t ep_pos[EPT], ep_loc[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int em = min(em0 + e * NWV * 4, p.M - 1);
    sxv[e] = 1.0f;
    if constexpr (SCALED) {
      if (slabs == nullptr) sxv[e] = p.sx[em];  // slab mode leaves the scales to the consumer kernel
    }
    ep_pos[e] = ep_loc[e] = 0;
    if constexpr (EPI == EPI_ROPE) {
      ep_pos[e] = ep.positions[em];
      ep_loc[e] = ep.loc[em];
    }
  }

  const unsigned wbytes = (unsigned)min((int64_t)p.N * p.w_stride, (int64_t)0xFFFFFFF0ll);
  const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, wbytes, 0x00020000);
  // ---- X rows of this wave's K slice first (L2 hits, back within a microsecond), THEN the first W tiles (HBM): the
  // counted wait below covers only the X loads, so the fragments are built while the weights are still in flight.
  // At most two 16-row tiles of X are staged per round trip (register budget at MT = 4). ----
  constexpr int XG = MT >= 2 ? 2 : 1;
  u32x4_t xr[XG][DS];
  // every workgroup needs the same X at the same moment: walking its rows in the same order makes all CUs hammer one L2
  // channel at a time (the X staging of this kernel ran at 18 B/clk/CU); each workgroup starts at a different row instead
  const int xrot = blockIdx.x + blockIdx.y;
  auto load_x = [&](int mt0) {
#pragma unroll
    for (int q = 0; q < XG; ++q)
#pragma unroll
      for (int i = 0; i < DS; ++i) {
        const int m = (mt0 + q) * 16 + lr + RPI * ((i + xrot) & (DS - 1));
#ifdef SGL_SKINNY_TIMELINE
        const char* xsrc = p.x_tiled ? p.x + ((int64_t)(kr * NWV + w) * (MT * 16) + min(m, p.M - 1)) * kw + lc * 16
                                     : p.x + (int64_t)min(m, p.M - 1) * p.x_stride + (kok ? koff : 0);
        const u32x4_t v = *(const u32x4_t*)xsrc;
#else
        const u32x4_t v = *(const u32x4_t*)(p.x + (int64_t)min(m, p.M - 1) * p.x_stride + (kok ? koff : 0));
#endif
        xr[q][i] = (m < p.M && kok) ? v : zero4;
      }
  };
  load_x(0);
  u32x4_t wreg[PD][WN];
  auto issue = [&](int slot, int j) {
    const int n0t = (blockIdx.x + j * G) * rpt;
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      const int row = (i < nload) ? lr + RPI * i : lr;  // (8-row tiles: the upper half re-reads row lr, an L1 hit)
      const unsigned off = (j < cnt && kok) ? (unsigned)((int64_t)min(n0t + row, p.N - 1) * p.w_stride) + koff : 0xFFFFFFF0u;
      // cache policy 2 = nt: each weight byte is read by ONE CU, once per decode step (MI355X_MICROARCH.md "nt-weights").
      // Same-box A/B (round 2): decode step 4.58 -> 4.49 ms from this alone, 4.58 -> 4.28 ms together with the nt K/V gather.
      wreg[slot][i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 2));
    }
  };
#pragma unroll
  for (int j = 0; j < PD; ++j) issue(j, j);

  // coalesced rows -> swizzled image -> A-fragment registers, once per workgroup
  u32x4_t xf[MT][DS];
#pragma unroll
  for (int mt0 = 0; mt0 < MT; mt0 += XG) {
    if (mt0 > 0) load_x(mt0);
#pragma unroll
    for (int q = 0; q < XG; ++q) {
#pragma unroll
      for (int i = 0; i < DS; ++i) {
        const int row = lr + RPI * ((i + xrot) & (DS - 1));
        *(u32x4_t*)(wl + row * kw + (((lc ^ row) & (LPR - 1)) << 4)) = xr[q][i];
      }
#pragma unroll
      for (int sI = 0; sI < DS; ++sI)
        xf[mt0 + q][sI] = *(const u32x4_t*)(wl + a * kw + ((((4 * sI + g) ^ a) & (LPR - 1)) << 4));
    }
  }
  const int arow = (rpt == 16) ? a : (a & 7);
  SK_STAMP(1);  // X fragments built

  for (int j0 = 0; j0 < cnt; j0 += TPP) {
    // epilogue operands of this phase's tiles (fetched now, consumed after the phase's MFMAs)
    float swv[TPP];
    uint16_t braw[TPP];
    if (slabs == nullptr) {
#pragma unroll
      for (int jj = 0; jj < TPP; ++jj) {
        const int ne = min((int)(blockIdx.x + min(j0 + jj, cnt - 1) * G) * rpt + en, p.N - 1);
        swv[jj] = 1.0f;
        if constexpr (SCALED) swv[jj] = p.sw[ne];
        braw[jj] = *(const uint16_t*)(biasp + (has_bias ? ne : 0));
      }
    }
    // EPI_ROPE: the cos / sin pair of every (row, tile) this thread finishes travels with them (loaded after the MFMAs it was
    // one more dependent L2 round trip at the end of the launch)
    float csc[EPI == EPI_ROPE ? EPT : 1][EPI == EPI_ROPE ? TPP : 1], css[EPI == EPI_ROPE ? EPT : 1][EPI == EPI_ROPE ? TPP : 1];
    if constexpr (EPI == EPI_ROPE) {
      const int H = rpt >> 1;
#pragma unroll
      for (int jj = 0; jj < TPP; ++jj) {
        const int n0 = (blockIdx.x + min(j0 + jj, cnt - 1) * G) * rpt;
        const int i = H * ((n0 & 127) / rpt) + (en & (H - 1));
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const float* cs = ep.cos_sin + ep_pos[e] * 128;
          csc[e][jj] = cs[i];
          css[e][jj] = cs[64 + i];
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < TPP; ++jj) {
      const int j = j0 + jj;
      const int slot = (TPP == 1) ? 0 : (jj % PD);
      // staged registers -> swizzled image (wave private: same-wave LDS ops are ordered, no barrier needed)
#pragma unroll
      for (int i = 0; i < WN; ++i) {
        const int row = lr + RPI * i;
        *(u32x4_t*)(wl + row * kw + (((lc ^ row) & (LPR - 1)) << 4)) = wreg[slot][i];
      }
      if constexpr (TPP > 1) issue(slot, j + PD);
      if (j == 0) { SK_STAMP(5); }  // first weight tile landed and staged
      f32x4_t acc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sI = 0; sI < DS; ++sI) {
        const u32x4_t wf = *(const u32x4_t*)(wl + arow * kw + ((((4 * sI + g) ^ arow) & (LPR - 1)) << 4));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) MfmaOp<ES>::run(xf[mt][sI], wf, acc[mt]);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[jj][w][mt * 16 + 4 * g + r][a] = acc[mt][r];
    }
    SK_STAMP(2);  // this phase's MFMAs issued, partial sums in LDS
    __syncthreads();
    SK_STAMP(3);  // barrier passed
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int em = em0 + e * NWV * 4;
      if (em >= MT * 16) continue;
#pragma unroll
      for (int jj = 0; jj < TPP; ++jj) {
        const int j = j0 + jj;
        const int n0 = (blockIdx.x + j * G) * rpt;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < NWV; ++ww) v += red[jj][ww][em][en];
        const bool live = j < cnt && em < p.M && en < rpt && n0 + en < p.N;
        if (slabs != nullptr) {
          if (live) slabs[((int64_t)kr * p.M + em) * p.N + n0 + en] = v;
        } else {
          const float bv = has_bias ? (float)__builtin_bit_cast(OutT, braw[jj]) : 0.0f;
          v = v * sxv[e] * swv[jj] + bv;
          if constexpr (EPI == EPI_NONE) {
            if (live) ((OutT*)p.y)[(int64_t)em * p.y_stride + n0 + en] = (OutT)v;
          } else {
            float cv = 0.f, sv = 0.f;
            if constexpr (EPI == EPI_ROPE) { cv = csc[e][jj]; sv = css[e][jj]; }
            epi_store<OutT, EPI>(v, live, em, en, n0, rpt, ep, p.y, p.y_stride, cv, sv, ep_loc[e]);
          }
        }
      }
    }
    SK_STAMP(4);  // outputs stored
    if (TPP > 1) __syncthreads();  // the next phase overwrites red
  }
}

// out[m][n] = (sum_kr slabs[kr][m][n]) * sx[m] * sw[n] + bias[n]: the combine step of a split-K launch when no
// consumer kernel fuses it.
template <typename OutT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int S, const float* sx,
                                                            const float* sw, const OutT* bias, OutT* out,
                                                            int64_t out_stride, int M, int N) {
  const int64_t total = (int64_t)M * (N / 4);
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int m = (int)(idx / (N / 4)), n = (int)(idx - (int64_t)m * (N / 4)) * 4;
    f32x4_t v = *(const f32x4_t*)(slabs + (int64_t)m * N + n);
    for (int sI = 1; sI < S; ++sI) v += *(const f32x4_t*)(slabs + ((int64_t)sI * M + m) * N + n);
    const float sm = sx ? sx[m] : 1.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float r = v[c] * sm * (sw ? sw[n + c] : 1.0f);
      if (bias) r += (float)bias[n + c];
      out[(int64_t)m * out_stride + n + c] = (OutT)r;
    }
  }
}

inline int v2_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) cus = 256;
    else cus = prop.multiProcessorCount;
  }
  return cus;
}

// number of k-ranges the v2 kernel needs for this problem, 0 if it cannot take it
template <int ES>
inline int v2_kranges(const SkinnyParams& p, int* ds_out) {
  if (p.M > 64 || g_skinny_force_v1 || (int64_t)p.N * p.w_stride >= 0xFFFFFFF0ll) return 0;
  if ((ES == ES_FP8) != (p.sx != nullptr) || (!p.sx != !p.sw)) return 0;
  int ds;
  if (p.kbytes <= 1024) ds = 2;
  else if (p.kbytes <= 2048) ds = 4;
  // 1 KiB of K per wave only where it makes ONE k-range (4 KiB < K <= 8 KiB, M <= 32: no slabs at all); beyond that more, smaller
  // k-ranges halve the X bytes every workgroup stages (K = 14336: 4 ranges of 4 KiB instead of 2 of 8 KiB, 17.0 -> 14.5 us).
  // MT = 4 keeps 128 VGPRs of X fragments: 512 B of K per wave at most.
  else ds = (p.kbytes <= 4096 || p.M > 32 || p.kbytes > 8192) ? 8 : 16;
  if (p.kbytes % 64 != 0) return 0;
  *ds_out = ds;
  const int range = kV2Waves * ds * 64;
  return (p.kbytes + range - 1) / range;
}

template <int ES, int MT, int DS, typename OutT>
int launch_v2(const SkinnyParams& p, int kranges, float* slabs, hipStream_t st) {
  const int cus = v2_cus();
  // 8-row tiles when 16-row tiles would leave the last round of workgroups mostly idle
  const int per_range_wgs = cus / kranges > 0 ? cus / kranges : 1;
  const int t16 = (p.N + 15) / 16;
  const int rounds16 = (t16 + per_range_wgs - 1) / per_range_wgs;
  // (fp8 only: with 16-bit operands an 8-row tile issues the same 16 load instructions per wave for half the rows -- the
  // unquantised qkv_proj of Llama-3-8B, 384 tiles: decode step 5.47 ms with 8-row tiles, 5.29 ms with 16-row tiles)
  const bool use8 = ES == ES_FP8 && rounds16 < 4 && (t16 % per_range_wgs) != 0 && (t16 % per_range_wgs) < (3 * per_range_wgs) / 4;
  const int rpt = use8 ? 8 : 16;
  const int ntiles = (p.N + rpt - 1) / rpt;
  const int gx = ntiles < per_range_wgs ? ntiles : per_range_wgs;
  const dim3 grid(gx, kranges);
  constexpr int PD = 1;
  constexpr int TPP = (DS >= 16 || MT >= 4) ? 2 : 4;  // (LDS: the cross-wave reduce buffer grows with MT)
  if (ntiles <= gx) {
    hipLaunchKernelGGL((skinny_gemm_v2_kernel<ES, MT, DS, 1, 1, OutT>), grid, dim3(kV2Waves * 64), 0, st, p, rpt, ntiles, slabs);
    SGL_HIP_LAUNCH_CHECK();
    return SGL_MI355_OK;
  }
  if constexpr (ES == ES_FP8 && DS >= 4 && DS <= 8 && MT <= 2) {
    if (use8 && ntiles <= 3 * gx && g_skinny_allin) {
      // two or three 8-row tiles per workgroup: all of them requested up front (the R8 instantiation, round 4) -- e.g. the plain
      // qkv_proj shapes of the reference's table at M <= 32 (N 6144 / 4608: 768 / 576 tiles over 256 workgroups: 10.0 -> 9.2 us,
      // 9.0 -> 7.8 us).  Not at 33..64 rows (MT = 4: 252 registers and all 160 KiB of LDS -- 6144 x 4096 a tie, 1280 x 8192 16 % slower)
      hipLaunchKernelGGL((skinny_gemm_v2_kernel<ES, MT, DS, 3, 3, OutT, EPI_NONE, kV2Waves, true>), grid, dim3(kV2Waves * 64), 0, st, p,
                         rpt, ntiles, slabs);
      SGL_HIP_LAUNCH_CHECK();
      return SGL_MI355_OK;
    }
  }
  if constexpr (ES == ES_FP8 && DS == 8 && MT <= 2) {
    // slab mode (raw partial sums for a consumer kernel: down_proj of the decode step, 4 k-ranges x 4 tiles per workgroup) with TWO
    // tiles in flight per wave: a measurement hook, OFF by default.  No epilogue operands ride in this launch's memory queue, so
    // the round-3 explanation of why deeper prefetch lost in the epilogue-fused launches does not apply -- and it still lost:
    // same-box A/B of the headline step (round 4, order 4 5 4 5): 4.180 / 4.165 ms with one tile in flight, 4.199 / 4.180 with two.
    // The X rows of the same CU share the queue with the deeper weight stream, and the stream is bandwidth-, not latency-bound.
    if (slabs != nullptr && g_skinny_slab_pd2 && ntiles >= 2 * gx) {
      hipLaunchKernelGGL((skinny_gemm_v2_kernel<ES, MT, DS, 2, TPP, OutT>), grid, dim3(kV2Waves * 64), 0, st, p, rpt, ntiles, slabs);
      SGL_HIP_LAUNCH_CHECK();
      return SGL_MI355_OK;
    }
  }
  hipLaunchKernelGGL((skinny_gemm_v2_kernel<ES, MT, DS, PD, TPP, OutT>), grid, dim3(kV2Waves * 64), 0, st, p, rpt, ntiles, slabs);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

template <int ES, int MT, typename OutT>
int launch_v2_ds(const SkinnyParams& p, int ds, int kranges, float* slabs, hipStream_t st) {
  switch (ds) {
    case 16:
      if constexpr (MT < 4) return launch_v2<ES, MT, 16, OutT>(p, kranges, slabs, st);  // (never chosen for M > 32)
      return SGL_MI355_EINVAL;
    case 8: return launch_v2<ES, MT, 8, OutT>(p, kranges, slabs, st);
    case 4: return launch_v2<ES, MT, 4, OutT>(p, kranges, slabs, st);
    default: return launch_v2<ES, MT, 2, OutT>(p, kranges, slabs, st);
  }
}

template <int ES, typename OutT>
int launch_mt(const SkinnyParams& p, float* workspace, int64_t workspace_floats, hipStream_t st) {
  int ds = 0;
  const int kranges = (sizeof(OutT) == 2) ? v2_kranges<ES>(p, &ds) : 0;
  auto by_m = [&](int kr, float* slabs) {
    if (p.M <= 16) return launch_v2_ds<ES, 1, OutT>(p, ds, kr, slabs, st);
    if (p.M <= 32) return launch_v2_ds<ES, 2, OutT>(p, ds, kr, slabs, st);
    return launch_v2_ds<ES, 4, OutT>(p, ds, kr, slabs, st);
  };
  if (kranges == 1) return by_m(1, nullptr);
  if (kranges > 1 && workspace != nullptr && workspace_floats >= (int64_t)kranges * p.M * p.N && p.N % 4 == 0) {
    const int rc = by_m(kranges, workspace);
    if (rc != SGL_MI355_OK) return rc;
    const int64_t items = (int64_t)p.M * (p.N / 4);
    const unsigned blocks = (unsigned)((items + 255) / 256 > 2048 ? 2048 : (items + 255) / 256);
    hipLaunchKernelGGL((splitk_reduce_kernel<OutT>), dim3(blocks), dim3(256), 0, st, workspace, kranges, p.sx, p.sw,
                       (const OutT*)p.bias, (OutT*)p.y, p.y_stride, p.M, p.N);
    SGL_HIP_LAUNCH_CHECK();
    return SGL_MI355_OK;
  }
  if (p.M <= 16) return launch<ES, 1, OutT>(p, st);
  if (p.M <= 32) return launch<ES, 2, OutT>(p, st);
  return launch<ES, 4, OutT>(p, st);
}

}  // namespace

extern "C" int sgl_mi355_skinny_gemm_force_generic(int on) {
  if (on == 2 || on == 3) {   // measurement hook: 2 = 8-row tiles one tile ahead (the round-3 form), 3 = all tiles up front (default)
    g_skinny_allin = on - 2;
    return SGL_MI355_OK;
  }
  if (on == 4 || on == 5) {   // measurement hook: 4 = slab-mode launches one tile ahead (default), 5 = two tiles in flight
    g_skinny_slab_pd2 = on - 4;
    return SGL_MI355_OK;
  }
  g_skinny_force_v1 = on != 0;
  return SGL_MI355_OK;
}

// in_dtype: SGL_FP8_E4M3 / SGL_BF16 / SGL_F16 (X and W share it); out_dtype: SGL_BF16 / SGL_F16.
extern "C" int sgl_mi355_skinny_gemm_num_kranges(int M, int N, int K, int in_dtype) {
  // how many f32 [M, N] slabs of workspace sgl_mi355_skinny_gemm wants for this shape (0 or 1: none needed)
  if (M > 64 || g_skinny_force_v1) return 0;
  const int kbytes = K * (in_dtype == SGL_FP8_E4M3 ? 1 : 2);
  if (kbytes % 64 != 0 || kbytes <= 4096) return kbytes % 64 == 0 ? 1 : 0;
  const int range = (M > 32 || kbytes > 8192) ? 4096 : 8192;  // K bytes one workgroup covers
  return (kbytes + range - 1) / range;
}

// number of f32 [M, N] slabs sgl_mi355_skinny_gemm_slabs writes for K fp8 elements (K bytes of every row)
extern "C" int sgl_mi355_skinny_gemm_slabs_count(int M, int K) {
  const int ds = K <= 1024 ? 2 : (K <= 2048 ? 4 : ((K <= 4096 || M > 32 || K > 8192) ? 8 : 16));
  const int range = kV2Waves * ds * 64;
  return (K + range - 1) / range;
}

extern "C" int sgl_mi355_skinny_gemm(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems, void* y,
                                     int64_t y_stride_elems, const float* scales_x, const float* scales_w,
                                     const void* bias, int M, int N, int K, int in_dtype, int out_dtype,
                                     float* workspace, int64_t workspace_floats, void* stream) {
  SGL_CHECK(M >= 0 && N >= 0 && K >= 0, "skinny_gemm: negative shape");
  if (M == 0 || N == 0) return SGL_MI355_OK;
  SGL_CHECK(M <= 64, "skinny_gemm: M=%d exceeds 64 (use the tiled GEMM)", M);
  SGL_CHECK(x && w && y, "skinny_gemm: null pointer");
  SGL_CHECK(in_dtype == SGL_FP8_E4M3 || in_dtype == SGL_BF16 || in_dtype == SGL_F16, "skinny_gemm: bad in_dtype %d", in_dtype);
  SGL_CHECK(out_dtype == SGL_BF16 || out_dtype == SGL_F16, "skinny_gemm: out_dtype must be bf16 or f16");
  const int es = in_dtype == SGL_FP8_E4M3 ? 1 : 2;
  SGL_CHECK((K * es) % 16 == 0 && (x_stride_elems * es) % 16 == 0 && (w_stride_elems * es) % 16 == 0 &&
                ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0,
            "skinny_gemm: rows must be 16-byte aligned (K=%d)", K);
  SkinnyParams p;
  p.x = (const char*)x; p.x_stride = x_stride_elems * es;
  p.w = (const char*)w; p.w_stride = w_stride_elems * es;
  p.y = y; p.y_stride = y_stride_elems;
  p.sx = scales_x; p.sw = scales_w; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.kbytes = K * es;
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == SGL_FP8_E4M3)
    return out_dtype == SGL_BF16 ? launch_mt<ES_FP8, __bf16>(p, workspace, workspace_floats, st)
                                 : launch_mt<ES_FP8, _Float16>(p, workspace, workspace_floats, st);
  if (in_dtype == SGL_BF16)
    return out_dtype == SGL_BF16 ? launch_mt<ES_BF16, __bf16>(p, workspace, workspace_floats, st)
                                 : launch_mt<ES_BF16, _Float16>(p, workspace, workspace_floats, st);
  return out_dtype == SGL_BF16 ? launch_mt<ES_F16, __bf16>(p, workspace, workspace_floats, st)
                               : launch_mt<ES_F16, _Float16>(p, workspace, workspace_floats, st);
}

// Raw split-K partial sums only: slabs f32 [kranges, M, N] (kranges = sgl_mi355_skinny_gemm_slabs_count(M, K * element size)),
// no scales; the consumer kernel (sgl_mi355_fused_add_rmsnorm_quant_fp8 with slabs) combines them at its own launch boundary.
// in_dtype SGL_FP8_E4M3 / SGL_BF16 / SGL_F16 (X and W share it); strides in elements.
// min_kranges > 1 asks for at least that many k-ranges (a shorter K slice per wave: 2048 / 1024 bytes per workgroup instead of
// 4096): each workgroup then stages a fraction of X before its first MFMA, which is what a short launch is made of (o_proj of
// Llama-3-8B at batch 32: 8.1 us with one k-range, 7.4 / 6.8 us with two / four) -- worth it where a
// consumer kernel sums the slabs anyway.  ..._slabs_count_min gives the number of slabs written for the same arguments.
extern "C" int sgl_mi355_skinny_gemm_slabs_count_min(int M, int K, int min_kranges) {
  int ds = K <= 1024 ? 2 : (K <= 2048 ? 4 : ((K <= 4096 || M > 32 || K > 8192) ? 8 : 16));
  while (ds > 2 && (K + kV2Waves * ds * 64 - 1) / (kV2Waves * ds * 64) < min_kranges) ds >>= 1;
  const int range = kV2Waves * ds * 64;
  return (K + range - 1) / range;
}

extern "C" int sgl_mi355_skinny_gemm_slabs_min(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems,
                                               float* slabs, int M, int N, int K, int in_dtype, int min_kranges, void* stream) {
  SGL_CHECK(M > 0 && M <= 64 && N > 0 && K > 0, "skinny_gemm_slabs: needs 0 < M <= 64");
  SGL_CHECK(x && w && slabs, "skinny_gemm_slabs: null pointer");
  SGL_CHECK(in_dtype == SGL_FP8_E4M3 || in_dtype == SGL_BF16 || in_dtype == SGL_F16, "skinny_gemm_slabs: bad in_dtype %d", in_dtype);
  const int es = in_dtype == SGL_FP8_E4M3 ? 1 : 2;
  SGL_CHECK((K * es) % 64 == 0 && (x_stride_elems * es) % 16 == 0 && (w_stride_elems * es) % 16 == 0 && ((uintptr_t)x % 16) == 0 &&
                ((uintptr_t)w % 16) == 0 && (int64_t)N * w_stride_elems * es < 0xFFFFFFF0ll,
            "skinny_gemm_slabs: unsupported shape/alignment (K=%d)", K);
  SkinnyParams p;
  p.x = (const char*)x; p.x_stride = x_stride_elems * es;
  p.w = (const char*)w; p.w_stride = w_stride_elems * es;
  p.y = nullptr; p.y_stride = 0;
  p.sx = nullptr; p.sw = nullptr; p.bias = nullptr;  // slab mode never reads scales or bias
  p.M = M; p.N = N; p.K = K; p.kbytes = K * es;
  const int kb = p.kbytes;
  int ds = kb <= 1024 ? 2 : (kb <= 2048 ? 4 : ((kb <= 4096 || M > 32 || kb > 8192) ? 8 : 16));  // = sgl_mi355_skinny_gemm's choice
  while (ds > 2 && (kb + kV2Waves * ds * 64 - 1) / (kV2Waves * ds * 64) < min_kranges) ds >>= 1;
  const int range = kV2Waves * ds * 64;
  const int kranges = (kb + range - 1) / range;
  hipStream_t st = (hipStream_t)stream;
#define SGL_SLABS_BY_M(ES, T)                                                      \
  do {                                                                              \
    if (M <= 16) return launch_v2_ds<ES, 1, T>(p, ds, kranges, slabs, st);          \
    if (M <= 32) return launch_v2_ds<ES, 2, T>(p, ds, kranges, slabs, st);          \
    return launch_v2_ds<ES, 4, T>(p, ds, kranges, slabs, st);                       \
  } while (0)
  if (in_dtype == SGL_FP8_E4M3) SGL_SLABS_BY_M(ES_FP8, __bf16);
  if (in_dtype == SGL_BF16) SGL_SLABS_BY_M(ES_BF16, __bf16);
  SGL_SLABS_BY_M(ES_F16, _Float16);
#undef SGL_SLABS_BY_M
}

extern "C" int sgl_mi355_skinny_gemm_slabs(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems,
                                           float* slabs, int M, int N, int K, int in_dtype, void* stream) {
  return sgl_mi355_skinny_gemm_slabs_min(x, x_stride_elems, w, w_stride_elems, slabs, M, N, K, in_dtype, 1, stream);
}

namespace {
template <int ES, int MT, int DS, typename OutT, int EPI>
int launch_v2_epi(const SkinnyParams& p, const EpiParams& ep, int rpt, hipStream_t st) {
  const int cus = v2_cus();
  const int ntiles = p.N / rpt;
  const int gx = ntiles < cus ? ntiles : cus;
  constexpr int TPP = (DS >= 16 || MT >= 4) ? 2 : 4;  // (the plain launcher's rule: LDS budget of the reduce buffer)
  // PD register sets in flight per wave.  One: with two (SGL_SKINNY_PD=2, tools/build_variant.sh) the three 8-row tiles of
  // qkv_proj need two dependent HBM round trips instead of three, but the deeper queue delays every other load of the CU (X rows,
  // epilogue operands): same-box A/B in the model (round 3) 4.39 ms/step with two sets against 4.26 with one.
  constexpr int PDV = (g_skinny_pd_test > 0 && TPP % (g_skinny_pd_test > 0 ? g_skinny_pd_test : 1) == 0) ? g_skinny_pd_test : 1;
  if (ntiles <= gx) {
    hipLaunchKernelGGL((skinny_gemm_v2_kernel<ES, MT, DS, 1, 1, OutT, EPI>), dim3(gx, 1), dim3(kV2Waves * 64), 0, st, p, rpt,
                       ntiles, (float*)nullptr, ep);
    SGL_HIP_LAUNCH_CHECK();
    return SGL_MI355_OK;
  }
  if constexpr (DS >= 4 && DS <= 8 && MT <= 2) {
    if (rpt == 8 && ntiles <= 3 * gx && g_skinny_allin) {
      // two or three 8-row tiles per workgroup: every tile requested up front (see R8 above)
      hipLaunchKernelGGL((skinny_gemm_v2_kernel<ES, MT, DS, 3, 3, OutT, EPI, kV2Waves, true>), dim3(gx, 1), dim3(kV2Waves * 64), 0, st,
                         p, rpt, ntiles, (float*)nullptr, ep);
      SGL_HIP_LAUNCH_CHECK();
      return SGL_MI355_OK;
    }
  }
  hipLaunchKernelGGL((skinny_gemm_v2_kernel<ES, MT, DS, PDV, TPP, OutT, EPI>), dim3(gx, 1), dim3(kV2Waves * 64), 0, st,
                     p, rpt, ntiles, (float*)nullptr, ep);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// fp8 operands: either 16-bit output; 16-bit operands: the output has the operand dtype (the unquantised linear)
template <int ES, int MT, int DS, int EPI>
int launch_v2_epi_out(const SkinnyParams& p, const EpiParams& ep, int out_dtype, int rpt, hipStream_t st) {
  if constexpr (ES == ES_FP8)
    return out_dtype == SGL_BF16 ? launch_v2_epi<ES, MT, DS, __bf16, EPI>(p, ep, rpt, st)
                                 : launch_v2_epi<ES, MT, DS, _Float16, EPI>(p, ep, rpt, st);
  else if constexpr (ES == ES_BF16)
    return launch_v2_epi<ES, MT, DS, __bf16, EPI>(p, ep, rpt, st);
  else
    return launch_v2_epi<ES, MT, DS, _Float16, EPI>(p, ep, rpt, st);
}

template <int ES, int EPI>
int run_epi_es(SkinnyParams& p, const EpiParams& ep, int out_dtype, int tile_rows, hipStream_t st, const char* who) {
  int ds = 0;
  SGL_CHECK(v2_kranges<ES>(p, &ds) == 1,
            "%s: K=%d (%d bytes) at M=%d does not fit one k-range of the weight-streaming kernel (<= 4096 bytes, <= 8192 at M <= 32)",
            who, p.K, p.kbytes, p.M);
#define SGL_EPI_CASE(MTv, DSv) return launch_v2_epi_out<ES, MTv, DSv, EPI>(p, ep, out_dtype, tile_rows, st)
#define SGL_EPI_BY_DS(MTv)              \
  if (ds == 8) { SGL_EPI_CASE(MTv, 8); } \
  if (ds == 4) { SGL_EPI_CASE(MTv, 4); } \
  if (ds == 2) { SGL_EPI_CASE(MTv, 2); }
  if (p.M <= 16) {
    SGL_EPI_BY_DS(1)
    SGL_EPI_CASE(1, 16);
  }
  if (p.M <= 32) {
    SGL_EPI_BY_DS(2)
    SGL_EPI_CASE(2, 16);
  }
  SGL_EPI_BY_DS(4)
  return SGL_MI355_EINVAL;  // (ds = 16 is never chosen for M > 32)
#undef SGL_EPI_BY_DS
#undef SGL_EPI_CASE
}

// in_dtype SGL_FP8_E4M3 (both scale vectors required, out bf16 / f16) or SGL_BF16 / SGL_F16 (no scales, out_dtype == in_dtype)
template <int EPI>
int run_epi(SkinnyParams& p, const EpiParams& ep, int in_dtype, int out_dtype, int tile_rows, hipStream_t st, const char* who) {
  SGL_CHECK(tile_rows == 16 || tile_rows == 8, "%s: tile_rows must be 16 or 8 (got %d)", who, tile_rows);
  SGL_CHECK(p.M > 0 && p.M <= 64, "%s: needs 0 < M <= 64 (got %d)", who, p.M);
  SGL_CHECK(in_dtype == SGL_FP8_E4M3 || in_dtype == SGL_BF16 || in_dtype == SGL_F16, "%s: bad in_dtype %d", who, in_dtype);
  SGL_CHECK(out_dtype == SGL_BF16 || out_dtype == SGL_F16, "%s: out_dtype must be bf16 or f16", who);
  SGL_CHECK(in_dtype == SGL_FP8_E4M3 || out_dtype == in_dtype, "%s: 16-bit operands give an output of the same dtype", who);
  SGL_CHECK((in_dtype == SGL_FP8_E4M3) == (p.sx != nullptr) && (!p.sx == !p.sw), "%s: scales go with fp8 operands, and only with them", who);
  const int es = in_dtype == SGL_FP8_E4M3 ? 1 : 2;
  p.kbytes = p.K * es;
  p.x_stride *= es;
  p.w_stride *= es;
  SGL_CHECK(p.kbytes % 64 == 0 && p.x_stride % 16 == 0 && p.w_stride % 16 == 0 && ((uintptr_t)p.x % 16) == 0 && ((uintptr_t)p.w % 16) == 0,
            "%s: rows must be 16-byte aligned and K a multiple of 64 bytes (K=%d)", who, p.K);
  SGL_CHECK(p.N % 16 == 0 && (int64_t)p.N * p.w_stride < 0xFFFFFFF0ll, "%s: N=%d must be a multiple of 16", who, p.N);
  if (in_dtype == SGL_FP8_E4M3) return run_epi_es<ES_FP8, EPI>(p, ep, out_dtype, tile_rows, st, who);
  if (in_dtype == SGL_BF16) return run_epi_es<ES_BF16, EPI>(p, ep, out_dtype, tile_rows, st, who);
  return run_epi_es<ES_F16, EPI>(p, ep, out_dtype, tile_rows, st, who);
}
}  // namespace

// act[M, N/2] = T(T(silu(g)) * u) with [g | u] = linear(x, w_interleaved) -- gate_up_proj + SiluAndMul in one launch.
// in_dtype SGL_FP8_E4M3: fp8_scaled_mm operands with both scale vectors; SGL_BF16 / SGL_F16: the unquantised linear (scales null,
// out_dtype == in_dtype).  w_interleaved [N, K] / scales_w [N]: tile t of tile_rows = 2 H rows (16 or 8) = gate rows H t .. H t + H - 1
// then the up rows of the same indices.  8-row tiles balance the workgroups when N / 16 is between one and a few times the CU
// count.  K must fit one k-range of the weight-streaming kernel (4096 bytes; 8192 at M <= 32).  Strides in elements.
static int gemm_silu_mul_impl(const void* x, int64_t x_stride_elems, const void* w_interleaved, int64_t w_stride_elems,
                              void* act, int64_t act_stride_elems, const float* scales_x,
                              const float* scales_w_interleaved, int M, int N, int K, int in_dtype, int out_dtype,
                              int tile_rows, void* stream, int* sched) {
  SGL_CHECK(x && w_interleaved && act, "gemm_silu_mul: null pointer");
  if (M > 64) {   // prefill-sized: the 256x256 tile with the SiluAndMul epilogue (tiled_gemm.hip)
    SGL_CHECK(in_dtype == SGL_FP8_E4M3 && out_dtype == SGL_BF16 && tile_rows == 16 && scales_x && scales_w_interleaved,
              "gemm_silu_mul: M > 64 takes fp8 operands with both scale vectors, a bf16 result and 16-row interleaving");
    SGL_CHECK(x_stride_elems % 16 == 0 && w_stride_elems % 16 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w_interleaved % 16) == 0,
              "gemm_silu_mul: rows must be 16-byte aligned");
    return sgl_mi355_internal_tiled_gemm_silu_mul(x, x_stride_elems, w_interleaved, w_stride_elems, act, act_stride_elems, scales_x,
                                                  scales_w_interleaved, M, N, K, (hipStream_t)stream, sched);
  }
  SkinnyParams p;
  p.x = (const char*)x; p.x_stride = x_stride_elems; p.w = (const char*)w_interleaved; p.w_stride = w_stride_elems;
  p.y = act; p.y_stride = act_stride_elems; p.sx = scales_x; p.sw = scales_w_interleaved; p.bias = nullptr;
  p.M = M; p.N = N; p.K = K; p.kbytes = 0;
  return run_epi<EPI_SILU>(p, EpiParams{}, in_dtype, out_dtype, tile_rows, (hipStream_t)stream, "gemm_silu_mul");
}

extern "C" int sgl_mi355_gemm_silu_mul(const void* x, int64_t x_stride_elems, const void* w_interleaved, int64_t w_stride_elems,
                                       void* act, int64_t act_stride_elems, const float* scales_x,
                                       const float* scales_w_interleaved, int M, int N, int K, int in_dtype, int out_dtype,
                                       int tile_rows, void* stream) {
  return gemm_silu_mul_impl(x, x_stride_elems, w_interleaved, w_stride_elems, act, act_stride_elems, scales_x, scales_w_interleaved, M, N, K,
                            in_dtype, out_dtype, tile_rows, stream, nullptr);
}

// The same with per-launch scratch (round 4): `sched` = 8 int32 words (one per XCD) of device memory that no other launch in flight uses (one
// buffer per stream is enough: launches of a stream are ordered).  With it the M > 64 form runs the persistent 256 x 256 kernel on a
// DYNAMIC per-XCD tile schedule (workgroups that finish early take more tiles; the launcher zeroes the words with a memset node, the
// kernel leaves them dirty) -- the same bits as without; NULL or M <= 64: exactly sgl_mi355_gemm_silu_mul.
extern "C" int sgl_mi355_gemm_silu_mul_ws(const void* x, int64_t x_stride_elems, const void* w_interleaved, int64_t w_stride_elems,
                                          void* act, int64_t act_stride_elems, const float* scales_x,
                                          const float* scales_w_interleaved, int M, int N, int K, int in_dtype, int out_dtype,
                                          int tile_rows, void* sched, void* stream) {
  SGL_CHECK(sched == nullptr || ((uintptr_t)sched & 3) == 0, "gemm_silu_mul_ws: sched must be 4-byte aligned");
  return gemm_silu_mul_impl(x, x_stride_elems, w_interleaved, w_stride_elems, act, act_stride_elems, scales_x, scales_w_interleaved, M, N, K,
                            in_dtype, out_dtype, tile_rows, stream, (int*)sched);
}

extern "C" int sgl_mi355_fp8_gemm_silu_mul(const void* x, int64_t x_stride_elems, const void* w_interleaved,
                                           int64_t w_stride_elems, void* act, int64_t act_stride_elems,
                                           const float* scales_x, const float* scales_w_interleaved, int M, int N, int K,
                                           int out_dtype, int tile_rows, void* stream) {
  SGL_CHECK(scales_x && scales_w_interleaved, "fp8_gemm_silu_mul: null pointer");
  return sgl_mi355_gemm_silu_mul(x, x_stride_elems, w_interleaved, w_stride_elems, act, act_stride_elems, scales_x,
                                 scales_w_interleaved, M, N, K, SGL_FP8_E4M3, out_dtype, tile_rows, stream);
}

// qkv_proj + neox rotary embedding + set_kv_buffer in one launch (in_dtype as for sgl_mi355_gemm_silu_mul).  w_interleaved /
// scales / bias rows: inside every q and k head (128 rows) tile u of tile_rows = 2 H rows = rows H u .. H u + H - 1 then rows
// 64 + H u ..; v heads in natural order.  q (rotated) -> q_out [M, Hq*128]; k (rotated) and v -> pool rows loc[m] of k_buffer /
// v_buffer ([slots, Hkv, 128], strides in elements).  kv_dtype: the pool's dtype -- out_dtype, or SGL_FP8_E4M3 with set_kv_buffer's
// conversion (k_scale / v_scale: the layer's scales, <= 0: none; memory_pool.py:385-395).
extern "C" int sgl_mi355_qkv_rope_set_kv(const void* x, int64_t x_stride_elems, const void* w_interleaved, int64_t w_stride_elems,
                                         void* q_out, int64_t q_stride_elems, const float* scales_x,
                                         const float* scales_w_interleaved, const void* bias_interleaved,
                                         const int64_t* positions, const float* cos_sin_cache, const int64_t* loc,
                                         void* k_buffer, void* v_buffer, int64_t k_slot_stride, int64_t v_slot_stride, int M,
                                         int num_q_heads, int num_kv_heads, int head_dim, int K, int in_dtype, int out_dtype,
                                         int tile_rows, int kv_dtype, float k_scale, float v_scale, void* stream) {
  SGL_CHECK(x && w_interleaved && q_out && positions && cos_sin_cache && loc && k_buffer && v_buffer, "qkv_rope_set_kv: null pointer");
  SGL_CHECK(kv_dtype == out_dtype || kv_dtype == SGL_FP8_E4M3, "qkv_rope_set_kv: the pool holds the output dtype or float8_e4m3fn (kv_dtype %d)", kv_dtype);
  SGL_CHECK(head_dim == 128, "qkv_rope_set_kv: head_dim (= rotary_dim) must be 128, got %d", head_dim);
  SkinnyParams p;
  p.x = (const char*)x; p.x_stride = x_stride_elems; p.w = (const char*)w_interleaved; p.w_stride = w_stride_elems;
  p.y = q_out; p.y_stride = q_stride_elems; p.sx = scales_x; p.sw = scales_w_interleaved; p.bias = bias_interleaved;
  p.M = M; p.N = (num_q_heads + 2 * num_kv_heads) * 128; p.K = K; p.kbytes = 0;
  EpiParams ep;
  ep.positions = positions; ep.cos_sin = cos_sin_cache; ep.loc = loc; ep.k_buf = k_buffer; ep.v_buf = v_buffer;
  ep.k_slot_stride = k_slot_stride; ep.v_slot_stride = v_slot_stride; ep.hq = num_q_heads; ep.hkv = num_kv_heads;
  ep.kv_fp8 = (kv_dtype == SGL_FP8_E4M3 && out_dtype != SGL_FP8_E4M3) ? 1 : 0; ep.k_scale = k_scale; ep.v_scale = v_scale;
  return run_epi<EPI_ROPE>(p, ep, in_dtype, out_dtype, tile_rows, (hipStream_t)stream, "qkv_rope_set_kv");
}

extern "C" int sgl_mi355_fp8_qkv_rope_set_kv(const void* x, int64_t x_stride_elems, const void* w_interleaved,
                                             int64_t w_stride_elems, void* q_out, int64_t q_stride_elems,
                                             const float* scales_x, const float* scales_w_interleaved,
                                             const void* bias_interleaved, const int64_t* positions,
                                             const float* cos_sin_cache, const int64_t* loc, void* k_buffer, void* v_buffer,
                                             int64_t k_slot_stride, int64_t v_slot_stride, int M, int num_q_heads,
                                             int num_kv_heads, int head_dim, int K, int out_dtype, int tile_rows,
                                             void* stream) {
  SGL_CHECK(scales_x && scales_w_interleaved, "fp8_qkv_rope_set_kv: null pointer");
  return sgl_mi355_qkv_rope_set_kv(x, x_stride_elems, w_interleaved, w_stride_elems, q_out, q_stride_elems, scales_x,
                                   scales_w_interleaved, bias_interleaved, positions, cos_sin_cache, loc, k_buffer, v_buffer,
                                   k_slot_stride, v_slot_stride, M, num_q_heads, num_kv_heads, head_dim, K, SGL_FP8_E4M3, out_dtype,
                                   tile_rows, out_dtype, -1.f, -1.f, stream);
}
