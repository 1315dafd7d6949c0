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

  const int nk = (p.kbytes + BKB - 1) / BKB;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const char* xa = smem + buf * 2 * TILE_BYTES;
    const char* wa = xa + TILE_BYTES;
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
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: acc[i][j][r] -> row m0+wm+16i+4g+r, col n0+wn+16j+a ----
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

template <int ES, typename OutT>
int launch(GemmParams& p, hipStream_t st) {
  constexpr int smem = 2 * 2 * TILE_BYTES;  // 64 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)tiled_gemm_kernel<ES, OutT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((tiled_gemm_kernel<ES, OutT>), dim3(p.tiles_m * p.tiles_n), dim3(256), smem, st, p);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

int run(const void* x, int64_t xs, const void* w, int64_t ws, void* y, int64_t ys, const float* sx, const float* sw,
        const void* bias, int M, int N, int K, int in_dtype, int out_dtype, void* stream, const char* who) {
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
  if (in_dtype == SGL_FP8_E4M3) return out_dtype == SGL_BF16 ? launch<TG_FP8, __bf16>(p, st) : launch<TG_FP8, _Float16>(p, st);
  if (in_dtype == SGL_BF16) return out_dtype == SGL_BF16 ? launch<TG_BF16, __bf16>(p, st) : launch<TG_BF16, _Float16>(p, st);
  return out_dtype == SGL_BF16 ? launch<TG_F16, __bf16>(p, st) : launch<TG_F16, _Float16>(p, st);
}

}  // namespace

extern "C" int sgl_mi355_fp8_gemm(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems, void* y,
                                  int64_t y_stride_elems, const float* scales_x, const float* scales_w, const void* bias,
                                  int M, int N, int K, int out_dtype, void* stream) {
  return run(x, x_stride_elems, w, w_stride_elems, y, y_stride_elems, scales_x, scales_w, bias, M, N, K, SGL_FP8_E4M3,
             out_dtype, stream, "fp8_gemm");
}

extern "C" int sgl_mi355_dense_gemm(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems, void* y,
                                    int64_t y_stride_elems, const void* bias, int M, int N, int K, int in_dtype,
                                    int out_dtype, void* stream) {
  if (!(in_dtype == SGL_BF16 || in_dtype == SGL_F16)) {
    snprintf(g_sgl_mi355_err, sizeof(g_sgl_mi355_err), "dense_gemm: in_dtype must be bf16 or f16");
    return SGL_MI355_EINVAL;
  }
  return run(x, x_stride_elems, w, w_stride_elems, y, y_stride_elems, nullptr, nullptr, bias, M, N, K, in_dtype, out_dtype,
             stream, "dense_gemm");
}
