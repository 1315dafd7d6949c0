// In-kernel timeline of the weight-streaming skinny GEMM (o_proj shape: M 32, N 4096, K 4096, one 16-row tile per workgroup).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSGL_SKINNY_TIMELINE -I ../../ltp-sglang_amd/csrc skinny_timeline.hip -o skinny_timeline
#include "skinny_gemm.hip"
#include <vector>
thread_local char g_sgl_mi355_err[512] = {0};
int main() {
  const int M = 32, N = 4096, K = 4096, L = 8;
  char *x, *w; void* y; float *sx, *sw; long long* tl;
  hipMalloc(&x, (size_t)M * K); hipMalloc(&w, (size_t)L * N * K); hipMalloc(&y, (size_t)M * N * 2);
  hipMalloc(&sx, M * 4); hipMalloc(&sw, N * 4); hipMalloc(&tl, 256 * 8 * 8 * 8);
  hipMemset(x, 0x38, (size_t)M * K); hipMemset(w, 0x38, (size_t)L * N * K); hipMemset(sx, 0, M * 4); hipMemset(sw, 0, N * 4);
  SkinnyParams p;
  p.x = x; p.x_stride = K; p.y = y; p.y_stride = N; p.sx = sx; p.sw = sw; p.bias = nullptr; p.M = M; p.N = N; p.K = K; p.kbytes = K;
  p.w_stride = K; p.tl = tl;
  for (int l = 0; l < L; ++l) {  // distinct weights per launch: cold in cache, like consecutive layers
    p.w = w + (size_t)l * N * K;
    launch_v2<ES_FP8, 2, 8, __bf16>(p, 1, nullptr, 0);
  }
  hipDeviceSynchronize();
  std::vector<long long> h(256 * 8 * 8);
  hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost);
  long long t0 = h[0];
  for (int i = 0; i < 256 * 8; ++i) if (h[i * 8] < t0) t0 = h[i * 8];
  for (int wg : {0, 1, 100, 255}) {
    for (int wv : {0, 7}) {
      printf("wg %3d wave %d:", wg, wv);
      for (int i = 0; i < 5; ++i) printf(" %7lld", h[(wg * 8 + wv) * 8 + i] - t0);
      printf("\n");
    }
  }
  // duration of the launch as a function of the K split (slab mode for > 1 k-range: the consumer combines)
  float* slabs; hipMalloc(&slabs, (size_t)8 * M * N * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int ds : {8, 4, 2}) {
    const int kr = K / (8 * ds * 64);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      for (int l = 0; l < L; ++l) {
        p.w = w + (size_t)l * N * K;
        if (ds == 8) launch_v2<ES_FP8, 2, 8, __bf16>(p, 1, nullptr, 0);
        if (ds == 4) launch_v2<ES_FP8, 2, 4, __bf16>(p, kr, slabs, 0);
        if (ds == 2) launch_v2<ES_FP8, 2, 2, __bf16>(p, kr, slabs, 0);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms / L < best) best = ms / L;
    }
    printf("K bytes per wave %4d, %d k-range(s): %.2f us per launch (back-to-back launches, not graph)\n", ds * 64, kr, best * 1e3);
  }
  // fewer, fatter workgroups: the X broadcast (workgroups x 128 KiB through L2) shrinks, each workgroup streams more tiles
  for (int gx : {256, 128, 64}) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      for (int l = 0; l < L; ++l) {
        p.w = w + (size_t)l * N * K;
        if (gx == 256) hipLaunchKernelGGL((skinny_gemm_v2_kernel<ES_FP8, 2, 8, 1, 1, __bf16>), dim3(gx, 1), dim3(512), 0, 0, p, 16, N / 16, (float*)nullptr, EpiParams{});
        else hipLaunchKernelGGL((skinny_gemm_v2_kernel<ES_FP8, 2, 8, 1, 4, __bf16>), dim3(gx, 1), dim3(512), 0, 0, p, 16, N / 16, (float*)nullptr, EpiParams{});
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms / L < best) best = ms / L;
    }
    printf("%3d workgroups (%d tiles each): %.2f us per launch\n", gx, N / 16 / gx, best * 1e3);
  }
  long long last = 0;
  for (int i = 0; i < 256 * 8; ++i) if (h[i * 8 + 4] > last) last = h[i * 8 + 4];
  printf("last store stamp: %lld ticks after the first entry (100 MHz ticks if s_memtime is the constant clock)\n", last - t0);
  return 0;
}
