// In-kernel timeline of the fused int4 dequant GEMM (Qwen2-7B gate_up shape: M 32, K 3584, N 37888, f16).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSGL_AWQ_TIMELINE -I ../../ltp-sglang_amd/csrc awq_timeline.hip -o awq_timeline
#include "awq.hip"
#include <vector>
thread_local char g_sgl_mi355_err[512] = {0};
int main(int argc, char** argv) {
  const int M = 32, K = argc > 1 ? atoi(argv[1]) : 3584, N = argc > 2 ? atoi(argv[2]) : 37888, G = 128, L = 6;
  char *x, *qp, *sz; void* y; long long* tl;
  const size_t wbytes = (size_t)K * N / 2, szbytes = (size_t)(K / G) * N * 4;
  hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&qp, L * wbytes); hipMalloc(&sz, L * szbytes); hipMalloc(&y, (size_t)M * N * 2);
  hipMalloc(&tl, 256 * 8 * 16 * 8);
  hipMemset(x, 0x38, (size_t)M * K * 2); hipMemset(qp, 0x5a, L * wbytes); hipMemset(sz, 0x1c, L * szbytes); hipMemset(tl, 0, 256 * 8 * 16 * 8);
  AwqGemmParams p;
  p.x = x; p.x_stride = K; p.y = y; p.y_stride = N; p.bias = nullptr; p.M = M; p.N = N; p.K = K; p.G = G; p.KB = K / 128; p.tl = tl;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int l = 0; l < L; ++l) {  // distinct weights per launch: cold in cache, like consecutive layers
      p.qpacked = (const uint32_t*)(qp + l * wbytes); p.sz = (const uint32_t*)(sz + l * szbytes);
      awq_launch<_Float16, 2, 1>(p, 1, nullptr, 0);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%.2f us per launch (back-to-back, not graph)\n", ms / L * 1e3);
  }
  std::vector<long long> h(256 * 8 * 16);
  hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost);
  long long t0 = h[0];
  for (int i = 0; i < 256 * 8; ++i) if (h[i * 16] && h[i * 16] < t0) t0 = h[i * 16];
  printf("stamps (10 ns ticks from the first entry): entry, X built, then per phase: start of tiles 0..3 | tiles done | barrier passed | outputs stored\n");
  for (int wg : {0, 1, 63, 100, 255}) {
    for (int wv : {0, 3, 7}) {
      printf("wg %3d wave %d:", wg, wv);
      const long long base = h[(wg * 8) * 16];  // (s_memtime is per XCD: ticks relative to this workgroup's wave 0 entry)
      for (int i = 0; i < 16; ++i) printf(" %6lld", h[(wg * 8 + wv) * 16 + i] ? h[(wg * 8 + wv) * 16 + i] - base : -1);
      printf("\n");
    }
  }
  return 0;
}
