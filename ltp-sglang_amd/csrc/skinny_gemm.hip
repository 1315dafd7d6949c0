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
};

enum { ES_FP8 = 0, ES_BF16 = 1, ES_F16 = 2 };

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

template <int ES, typename OutT>
int launch_mt(const SkinnyParams& p, hipStream_t st) {
  if (p.M <= 16) return launch<ES, 1, OutT>(p, st);
  if (p.M <= 32) return launch<ES, 2, OutT>(p, st);
  return launch<ES, 4, OutT>(p, st);
}

}  // namespace

// in_dtype: SGL_FP8_E4M3 / SGL_BF16 / SGL_F16 (X and W share it); out_dtype: SGL_BF16 / SGL_F16.
extern "C" int sgl_mi355_skinny_gemm(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems, void* y,
                                     int64_t y_stride_elems, const float* scales_x, const float* scales_w,
                                     const void* bias, int M, int N, int K, int in_dtype, int out_dtype, void* stream) {
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
    return out_dtype == SGL_BF16 ? launch_mt<ES_FP8, __bf16>(p, st) : launch_mt<ES_FP8, _Float16>(p, st);
  if (in_dtype == SGL_BF16)
    return out_dtype == SGL_BF16 ? launch_mt<ES_BF16, __bf16>(p, st) : launch_mt<ES_BF16, _Float16>(p, st);
  return out_dtype == SGL_BF16 ? launch_mt<ES_F16, __bf16>(p, st) : launch_mt<ES_F16, _Float16>(p, st);
}
