// Microbenchmark: per-kernel cost of a chain of tiny dependent kernels -- eager vs hipGraph replay (MI355X).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void tiny(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
int main() {
  float* buf; hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 400;
  for (int grid : {32, 256}) {
    // eager
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, st);
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, st, buf, grid * 256);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("grid %d eager : %.2f us per kernel\n", grid, ms * 1e3 / N);
    }
    // graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, st, buf, grid * 256);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, st);
      hipGraphLaunch(ge, st);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("grid %d graph : %.2f us per kernel\n", grid, ms * 1e3 / N);
    }
  }
  return 0;
}
