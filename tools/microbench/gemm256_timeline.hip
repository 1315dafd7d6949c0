// In-kernel timeline of the 256x256 fp8 GEMM main loop (s_memtime stamps of workgroup 0, K slices 8..11, all 8 waves).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSGL_GEMM_TIMELINE -I ../../ltp-sglang_amd/csrc gemm256_timeline.hip -o gemm256_timeline
#include "tiled_gemm.hip"
#include <vector>
thread_local char g_sgl_mi355_err[512] = {0};
int main() {
  const int M = 16384, N = 6144, K = 4096;
  char *x, *w; void* y; long long* tl;
  hipMalloc(&x, (size_t)M * K); hipMalloc(&w, (size_t)N * K); hipMalloc(&y, (size_t)M * N * 2); hipMalloc(&tl, 4 * 8 * 16 * 8);
  std::vector<unsigned char> h((size_t)M * K);
  unsigned s = 12345;
  for (auto& b : h) { s = s * 1664525u + 1013904223u; b = (unsigned char)((s >> 24) & 0x7F) % 0x78; }  // finite e4m3 values
  hipMemcpy(x, h.data(), (size_t)M * K, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), (size_t)N * K, hipMemcpyHostToDevice);
  hipMemset(tl, 0, 4 * 8 * 16 * 8);
  GemmParams p;
  p.x = x; p.x_stride = K; p.w = w; p.w_stride = K; p.y = y; p.y_stride = N; p.sx = nullptr; p.sw = nullptr; p.bias = nullptr;
  p.M = M; p.N = N; p.kbytes = K; p.tl = tl;
  for (int rep = 0; rep < 3; ++rep) launch256<__bf16, 8, true>(p, 0);
  hipDeviceSynchronize();
  long long ht[4 * 8 * 16];
  hipMemcpy(ht, tl, sizeof(ht), hipMemcpyDeviceToHost);
  const long long t0 = ht[0];
  for (int wv = 0; wv < 8; ++wv) {
    printf("wave %d:", wv);
    for (int sl = 0; sl < 4; ++sl) {
      printf("  |");
      for (int i = 0; i < 2; ++i) printf(" %6lld", ht[((sl * 8 + wv) * 16) + i] - t0);
    }
    printf("\n");
  }
  return 0;
}
