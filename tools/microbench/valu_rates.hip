// Issue rate of the VALU instructions the int4 dequantisation is made of (gfx950): cycles per wave64 instruction with one
// and two waves per SIMD.  build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
template <int OP>
__global__ __launch_bounds__(512) void k(unsigned* out, long long* clk, int iters) {
  unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = 0x3c003c00u + a0, c = 0x64006400u;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP == 0) { REP16(asm volatile("v_pk_mul_f16 %0, %0, %4\n v_pk_mul_f16 %1, %1, %4\n v_pk_mul_f16 %2, %2, %4\n v_pk_mul_f16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if constexpr (OP == 1) { REP16(asm volatile("v_pk_add_f16 %0, %0, %4\n v_pk_add_f16 %1, %1, %4\n v_pk_add_f16 %2, %2, %4\n v_pk_add_f16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if constexpr (OP == 2) { REP16(asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(0x000F000Fu), "v"(c));) }
    if constexpr (OP == 3) { REP16(asm volatile("v_lshrrev_b32 %0, 4, %0\n v_lshrrev_b32 %1, 4, %1\n v_lshrrev_b32 %2, 4, %2\n v_lshrrev_b32 %3, 4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if constexpr (OP == 4) { REP16(asm volatile("v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if constexpr (OP == 5) { REP16(asm volatile("v_fma_mixlo_f16 %0, %0, %4, %5 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %1, %1, %4, %5 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %2, %2, %4, %5 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %3, %3, %4, %5 op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if constexpr (OP == 6) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if constexpr (OP == 7) { REP16(asm volatile("v_and_b32 %0, %4, %0\n v_and_b32 %1, %4, %1\n v_and_b32 %2, %4, %2\n v_and_b32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
    if constexpr (OP == 8) { REP16(asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte2 %2, %2\n v_cvt_f32_ubyte3 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if constexpr (OP == 9) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2" : "+v"(*(unsigned long long*)&a0), "+v"(*(unsigned long long*)&a2) : "v"(*(unsigned long long*)&b));) }
    if constexpr (OP == 10) { REP16(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if constexpr (OP == 11) { REP16(asm volatile("v_bfe_u32 %0, %0, 4, 4\n v_bfe_u32 %1, %1, 4, 4\n v_bfe_u32 %2, %2, 4, 4\n v_bfe_u32 %3, %3, 4, 4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if constexpr (OP == 12) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %4\n v_cvt_pk_bf16_f32 %1, %1, %4\n v_cvt_pk_bf16_f32 %2, %2, %4\n v_cvt_pk_bf16_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
  }
  if constexpr (OP >= 13) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    h8 xa, xb;
    for (int j = 0; j < 8; ++j) { xa[j] = (_Float16)(float)(a0 & 3); xb[j] = (_Float16)(float)(a1 & 3); }
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
      // OP 13: 2 MFMA per group; OP 14: 2 MFMA + 16 independent packed VALU ops per group; OP 15: the 16 VALU ops alone
#define GROUP                                                                                                              \
      if constexpr (OP != 15) asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1" : "+v"(acc0), "+v"(acc1) : "v"(xa), "v"(xb)); \
      if constexpr (OP != 13) { REP4(asm volatile("v_pk_mul_f16 %0, %0, %4\n v_pk_add_f16 %1, %1, %4\n v_pk_mul_f16 %2, %2, %4\n v_pk_add_f16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
      REP16(GROUP)
    }
    a0 += (unsigned)(acc0[0] + acc1[1]);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int OP>
void run(const char* name, int per_iter) {
  unsigned* out; long long* clk;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 8);
  for (int threads : {256, 512}) {  // one workgroup per CU: 1 or 2 waves per SIMD
    const int iters = 200;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * per_iter;
    printf("%-20s %d waves/SIMD: %6.2f ticks per instruction and wave, %6.2f ns per instruction and SIMD\n", name, threads / 256, c / n,
           ms * 1e6 / (n * (threads / 256)));
  }
}
int main() {
  run<0>("v_pk_mul_f16", 64); run<1>("v_pk_add_f16", 64); run<2>("v_and_or_b32", 64); run<3>("v_lshrrev_b32", 64);
  run<4>("v_pk_fma_f16", 64); run<5>("v_fma_mixlo_f16", 64); run<6>("v_fma_f32", 64); run<7>("v_and_b32", 64);
  run<8>("v_cvt_f32_ubyteN", 64); run<9>("v_pk_fma_f32", 32); run<10>("v_perm_b32", 64); run<11>("v_bfe_u32", 64);
  run<12>("v_cvt_pk_bf16_f32", 64);
  run<13>("2 mfma16x16x32", 16); run<15>("16 pk valu", 16); run<14>("2 mfma + 16 pk valu", 16);
  return 0;
}
