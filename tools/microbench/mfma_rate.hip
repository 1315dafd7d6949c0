// Issue cadence of v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 chains on gfx950 (cycles per MFMA, s_memtime), by where the
// accumulator lives (VGPR or AGPR), how many independent accumulator chains alternate (1, 2, 4), and waves per SIMD (1: 256-thread
// workgroups, 2: 512-thread).  Random-ish bf16 operands (cdna_hip_programming.md rule 25).  One workgroup per CU, every CU busy.
// build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4v;
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// MODE 0: 32x32x16, VGPR accumulators, NCH chains;  MODE 1: 32x32x16, AGPR accumulators (asm-owned a[0:63]);  MODE 2: 16x16x32 VGPR;
// MODE 3: v_mfma_scale_f32_16x16x128_f8f6f4 on e4m3 operands (the prefill GEMMs' instruction), 32 independent accumulator tiles
template <int MODE, int NCH>
__global__ __launch_bounds__(512, 2) void k(const bf8* in, float* out, long long* clk, int iters) {
  bf8 a = in[threadIdx.x], b = in[threadIdx.x + 512];
  long long t0, t1;
  if constexpr (MODE == 0) {
    f16v c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0; c1[r] = 0; c2[r] = 0; c3[r] = 0; }
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
      if constexpr (NCH == 1) { REP16(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));) }
      if constexpr (NCH == 2) { REP4(REP4(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));)) }
      if constexpr (NCH == 4) { REP4(REP4(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));)) }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else if constexpr (MODE == 1) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
      if constexpr (NCH == 1) { REP16(asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(a), "v"(b) : "a0","a15");) }
      if constexpr (NCH == 2) { REP4(REP4(asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]\n v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" :: "v"(a), "v"(b) : "a0","a31");)) }
      if constexpr (NCH == 4) { REP4(REP4(asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]\n v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]\n v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]\n v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" :: "v"(a), "v"(b) : "a0","a63");)) }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a3" : "=v"(x));
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  } else if constexpr (MODE == 3) {
    typedef int i8v __attribute__((ext_vector_type(8)));
    const i8v av = {(int)in[threadIdx.x][0], (int)in[threadIdx.x][1] * 77, (int)in[threadIdx.x][2] * 13, (int)in[threadIdx.x][3], (int)0x3a41c23fu, (int)0x45b8373cu, (int)(threadIdx.x * 0x01010101u ^ 0x38b4c13du), (int)0x3c3c3c3cu};
    const i8v bv = {(int)0x3839b8c0u, (int)(threadIdx.x * 0x00010203u ^ 0x41b23940u), (int)0x37c2b93eu, (int)0x3d3a4438u, (int)0x40b0c138u, (int)0x39394142u, (int)0xb8c0373du, (int)0x3e41b9c2u};
    f4v c[32];
    for (int i = 0; i < 32; ++i) c[i] = f4v{0, 0, 0, 0};
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) c[(j * 2) & 31] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c[(j * 2) & 31], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
    for (int i = 0; i < 32; ++i) sum += c[i][i & 3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  } else {
    f4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
      if constexpr (NCH == 1) { REP16(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));) }
      if constexpr (NCH == 2) { REP4(REP4(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1" : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));)) }
      if constexpr (NCH == 4) { REP4(REP4(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n v_mfma_f32_16x16x32_bf16 %3, %4, %5, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));)) }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  }
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE, int NCH>
void run(const char* name, int threads, const bf8* in, float* out, long long* clk) {
  const int iters = 256, grid = 256, nw = grid * threads / 64;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MODE, NCH>), dim3(grid), dim3(threads), 0, 0, in, out, clk, iters);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NCH>), dim3(grid), dim3(threads), 0, 0, in, out, clk, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(nw);
  hipMemcpy(h.data(), clk, nw * sizeof(long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double n = iters * 16.0 * (MODE == 3 ? 1 : NCH);
  const double flop = (MODE == 3 ? 16.0 * 16 * 128 * 2 : MODE == 2 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2) * n * nw;
  printf("%-44s %d waves/SIMD: %6.1f cycles per MFMA and wave (median), %7.1f TFLOP/s, %.2f GHz\n", name, threads / 256, h[nw / 2] / n, flop / ms / 1e9,
         h[nw / 2] / (ms * 1e6));
}

int main() {
  bf8* in; float* out; long long* clk;
  hipMalloc(&in, 1024 * sizeof(bf8)); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8 * 8);
  std::vector<unsigned short> h(1024 * 8);
  unsigned s = 12345;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (unsigned short)(0x3c00 + ((s >> 16) & 0x3ff) | ((s >> 3) & 0x8000)); }   // ~[-2, 2)
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    run<0, 1>("32x32x16 VGPR acc, 1 chain", threads, in, out, clk);
    run<0, 2>("32x32x16 VGPR acc, 2 chains", threads, in, out, clk);
    run<0, 4>("32x32x16 VGPR acc, 4 chains", threads, in, out, clk);
    run<1, 1>("32x32x16 AGPR acc, 1 chain", threads, in, out, clk);
    run<1, 2>("32x32x16 AGPR acc, 2 chains", threads, in, out, clk);
    run<1, 4>("32x32x16 AGPR acc, 4 chains", threads, in, out, clk);
    run<2, 1>("16x16x32 VGPR acc, 1 chain", threads, in, out, clk);
    run<2, 2>("16x16x32 VGPR acc, 2 chains", threads, in, out, clk);
    run<2, 4>("16x16x32 VGPR acc, 4 chains", threads, in, out, clk);
    run<3, 1>("16x16x128 f8f6f4 (e4m3), 16 accumulators", threads, in, out, clk);
  }
  return 0;
}
