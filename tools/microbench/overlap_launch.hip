// Microbenchmark (round 3): what does overlapping LAUNCHES buy on MI355X?  A chain of dependent weight-streaming kernels (each
// reads its predecessor's output, then streams its own bytes) replayed from a HIP graph in two forms:
//   serial  : one stream, every kernel a dependent graph node (a barrier packet between kernels) -- the decode step's form today;
//   overlap : kernels alternate between TWO streams with no cross-stream edges; kernel i requests its first tiles, then polls a
//             device counter that kernel i - 1's workgroups bump after their stores have drained (sc1 stores / sc1 loads for the
//             bytes handed over; bounded spin; stream order keeps at most two kernels in flight).
// Build: hipcc --offload-arch=gfx950 -O3 -o overlap_launch overlap_launch.hip     Run: ./overlap_launch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void stream_k(const u32x4* __restrict__ w, size_t n16_per_wg, const uint32_t* prev_out, uint32_t* out,
                                                const uint32_t* prev_cnt, uint32_t prev_target, uint32_t* my_cnt, uint32_t* err) {
  const u32x4* p = w + (size_t)blockIdx.x * n16_per_wg;
  const int tid = threadIdx.x;
  // first tiles requested BEFORE the dependency is known to be satisfied (weights do not depend on it)
  u32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  size_t i = tid;
  if (i + 1536 < n16_per_wg) {
    a0 = __builtin_nontemporal_load(p + i);
    a1 = __builtin_nontemporal_load(p + i + 512);
    a2 = __builtin_nontemporal_load(p + i + 1024);
    a3 = __builtin_nontemporal_load(p + i + 1536);
    i += 2048;
  }
  __shared__ uint32_t dep;
  if (prev_cnt != nullptr) {
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(prev_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < prev_target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 15)) { atomicAdd(err, 1u); break; }   // ~ 4 ms: a launch order that cannot make progress shows up as timeouts, not as a hang
      }
    }
    __syncthreads();
  }
  if (tid == 0) {   // the predecessor's output (write-through stored there, sc1-loaded here: the L2s of the XCDs are not coherent)
    uint32_t v = 0;
    if (prev_out) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(prev_out + blockIdx.x) : "memory");
    dep = v;
  }
  __syncthreads();
  for (; i + 1536 < n16_per_wg; i += 2048) {
    const u32x4 b0 = __builtin_nontemporal_load(p + i), b1 = __builtin_nontemporal_load(p + i + 512);
    const u32x4 b2 = __builtin_nontemporal_load(p + i + 1024), b3 = __builtin_nontemporal_load(p + i + 1536);
    a0 ^= b0; a1 ^= b1; a2 ^= b2; a3 ^= b3;
  }
  const u32x4 a = a0 ^ a1 ^ a2 ^ a3;
  uint32_t r = a[0] ^ a[1] ^ a[2] ^ a[3] ^ dep;
  for (int m = 32; m >= 1; m >>= 1) r ^= __shfl_xor(r, m, 64);
  __shared__ uint32_t red[8];
  if ((tid & 63) == 0) red[tid >> 6] = r;
  __syncthreads();
  if (tid == 0) {
    uint32_t t = 0;
    for (int k = 0; k < 8; ++k) t ^= red[k];
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(out + blockIdx.x), "v"(t) : "memory");
    __hip_atomic_fetch_add(my_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  const int G = 256, N = 32;   // workgroups per kernel, kernels per chain
  const size_t sizes_mb[] = {1, 16, 58, 117};
  uint32_t *outs, *cnt, *err;
  hipMalloc(&outs, (size_t)(N + 1) * G * 4); hipMemset(outs, 0, (size_t)(N + 1) * G * 4);
  hipMalloc(&cnt, (N + 1) * 4); hipMalloc(&err, 4); hipMemset(err, 0, 4);
  hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
  hipEvent_t e0, e1, fork, join; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&fork, hipEventDisableTiming);
  hipEventCreateWithFlags(&join, hipEventDisableTiming);
  for (size_t mb : sizes_mb) {
    const size_t bytes = mb << 20, n16 = bytes / 16, per = n16 / G;
    std::vector<u32x4*> w(8);   // eight distinct weight sets so that no kernel re-reads a cached one
    for (auto& p : w) { hipMalloc(&p, bytes); hipMemset(p, 1, bytes); }
    for (int mode = 0; mode < 3; ++mode) {   // 0 serial graph, 1 two-stream graph with flags, 2 two real streams (eager launches) with flags
      hipMemset(err, 0, 4);
      auto enqueue = [&](bool flags) {
        for (int k = 0; k < N; ++k) {
          hipStream_t st = (flags && (k & 1)) ? s1 : s0;
          hipLaunchKernelGGL(stream_k, dim3(G), dim3(512), 0, st, w[k % 8], per, k ? outs + (size_t)(k - 1) * G : nullptr, outs + (size_t)k * G,
                             (flags && k) ? cnt + (k - 1) : nullptr, (uint32_t)G, cnt + k, err);
        }
      };
      float best = 1e9f;
      if (mode < 2) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal);
        hipMemsetAsync(cnt, 0, (N + 1) * 4, s0);
        if (mode == 1) { hipEventRecord(fork, s0); hipStreamWaitEvent(s1, fork, 0); }
        enqueue(mode == 1);
        if (mode == 1) { hipEventRecord(join, s1); hipStreamWaitEvent(s0, join, 0); }
        hipStreamEndCapture(s0, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int rep = 0; rep < 4; ++rep) {
          hipEventRecord(e0, s0);
          hipGraphLaunch(ge, s0);
          hipEventRecord(e1, s0); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep >= 1 && ms < best) best = ms;
        }
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
      } else {
        for (int rep = 0; rep < 4; ++rep) {
          hipMemsetAsync(cnt, 0, (N + 1) * 4, s0);
          hipEventRecord(fork, s0); hipStreamWaitEvent(s1, fork, 0);
          hipEventRecord(e0, s0);
          enqueue(true);
          hipEventRecord(join, s1); hipStreamWaitEvent(s0, join, 0);
          hipEventRecord(e1, s0); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep >= 1 && ms < best) best = ms;
        }
      }
      uint32_t herr = 0, chk = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&chk, outs + (size_t)(N - 1) * G, 4, hipMemcpyDeviceToHost);
      const char* names[3] = {"serial graph        ", "two-stream graph    ", "two streams (eager) "};
      printf("%4zu MB per kernel, %s: %7.2f us per kernel (%.2f TB/s)  [spin timeouts %u, check %08x]\n", mb, names[mode],
             best * 1e3 / N, bytes / (best * 1e-3 / N) / 1e12, herr, chk);
      fflush(stdout);
    }
    for (auto& p : w) hipFree(p);
  }
  return 0;
}
