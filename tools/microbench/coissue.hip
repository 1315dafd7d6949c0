// Do the two waves of a SIMD overlap one's MFMAs with the other's VALU work?  (gfx950)  Workgroups of 512 threads: waves 0-3 run role A,
// waves 4-7 (their SIMD partners) role B; roles: M = back-to-back v_mfma_f32_32x32x16_bf16, V = independent v_fma_f32, E = v_exp_f32,
// X = mixed stream (per MFMA: 2 v_exp + 2 v_add + 1 v_cvt_pk), I = idle (exits at once).  Prints cycles per role unit for each pairing.
// build: hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
enum { M = 0, V = 1, E = 2, X = 3, I = 4 };

template <int ROLE>
__device__ __forceinline__ float role(const bf8* in, int iters) {
  bf8 a = in[threadIdx.x & 511], b = in[(threadIdx.x & 511) + 512];
  float r = 0.f;
  if constexpr (ROLE == M) {
    f16v c0;
    for (int i = 0; i < 16; ++i) c0[i] = 0;
    for (int i = 0; i < iters; ++i) { REP16(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));) }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    r = c0[3];
  } else if constexpr (ROLE == V) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < iters; ++i) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(0.5f));) }
    r = x0 + x1 + x2 + x3;
  } else if constexpr (ROLE == E) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < iters; ++i) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));) }
    r = x0 + x1 + x2 + x3;
  } else if constexpr (ROLE == X) {
    f16v c0;
    for (int i = 0; i < 16; ++i) c0[i] = 0;
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, s0 = 0, s1 = 0;
    unsigned pk = 0;
    for (int i = 0; i < iters; ++i) {
      REP16(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %2\n v_cvt_pk_bf16_f32 %5, %1, %2"
                         : "+v"(c0), "+v"(x0), "+v"(x1), "+v"(s0), "+v"(s1), "+v"(pk) : "v"(a), "v"(b));)
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    r = c0[3] + s0 + s1 + pk;
  }
  return r;
}

template <int RA, int RB>
__global__ __launch_bounds__(512, 2) void k(const bf8* in, float* out, long long* clk, int iters) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long t0 = __builtin_amdgcn_s_memtime();
  float r;
  if (w < 4) r = role<RA>(in, iters);
  else r = role<RB>(in, iters);
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + w] = t1 - t0;
}

template <int RA, int RB>
void run(const char* name, const bf8* in, float* out, long long* clk) {
  const int iters = 256, grid = 256;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<RA, RB>), dim3(grid), dim3(512), 0, 0, in, out, clk, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(grid * 8);
  hipMemcpy(h.data(), clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  std::vector<long long> a, b;
  for (int i = 0; i < grid * 8; ++i) ((i & 7) < 4 ? a : b).push_back(h[i]);
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  const double n = iters * 16.0;
  printf("%-34s waves 0-3: %6.1f cycles per unit    waves 4-7: %6.1f cycles per unit\n", name, a[a.size() / 2] / n, b[b.size() / 2] / n);
}

int main() {
  bf8* in; float* out; long long* clk;
  hipMalloc(&in, 1024 * sizeof(bf8)); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8 * 8);
  std::vector<unsigned short> h(1024 * 8);
  unsigned s = 12345;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (unsigned short)(0x3c00 + ((s >> 16) & 0x3ff) | ((s >> 3) & 0x8000)); }
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  printf("unit: M = 1 MFMA 32x32x16 (32 cycles of the matrix pipe); V = 4 v_fma_f32; E = 4 v_exp_f32; X = 1 MFMA + 2 v_exp + 2 v_add + 1 v_cvt_pk\n");
  run<M, I>("M alone", in, out, clk);
  run<V, I>("V alone", in, out, clk);
  run<E, I>("E alone", in, out, clk);
  run<X, I>("X alone", in, out, clk);
  run<M, M>("M with M", in, out, clk);
  run<M, V>("M with V", in, out, clk);
  run<M, E>("M with E", in, out, clk);
  run<V, V>("V with V", in, out, clk);
  run<E, E>("E with E", in, out, clk);
  run<X, X>("X with X", in, out, clk);
  run<X, M>("X with M", in, out, clk);
  return 0;
}
