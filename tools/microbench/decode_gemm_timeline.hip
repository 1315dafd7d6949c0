// In-kernel timeline (round 4) of the four weight-streaming GEMMs of the Llama-3-8B w8a8 decode step at batch 32, each in the form
// the step launches (qkv + RoPE + KV write on 8-row tiles, o_proj, gate_up + SiluAndMul, down_proj into 4 k-range slabs), replayed
// from a HIP graph behind a small dependent kernel (as in the step: a row kernel precedes every GEMM), eight cold weight sets.
// Stamps (s_memrealtime, 10 ns): 0 entry | 1 X fragments built | 5 first weight tile staged | 2 last phase's MFMAs issued |
// 3 its barrier passed | 4 outputs stored.  Printed: median / max over all (workgroup, wave) in us after the first entry.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSGL_SKINNY_TIMELINE -I ../../ltp-sglang_amd/csrc decode_gemm_timeline.hip -o decode_gemm_timeline
#include "skinny_gemm.hip"
#include <algorithm>
#include <vector>
thread_local char g_sgl_mi355_err[512] = {0};
int sgl_mi355_internal_tiled_gemm_silu_mul(const void*, int64_t, const void*, int64_t, void*, int64_t, const float*, const float*, int, int, int, hipStream_t) { return 1; }

__global__ void small_k(float* o) { if (threadIdx.x == 0) o[blockIdx.x] += 1.0f; }

int main() {
  const int M = 32, L = 8;
  struct G { const char* name; int N, K, kind, tiled; } gs[] = {{"qkv + RoPE (8-row tiles, all up front)", 6144, 4096, 0, 0}, {"qkv + RoPE (8-row tiles, one ahead)", 6144, 4096, 4, 0},
                                                          {"o_proj", 4096, 4096, 1, 0}, {"gate_up + SiluAndMul", 28672, 4096, 2, 0}, {"down_proj -> slabs", 4096, 14336, 3, 0},
                                                          {"o_proj, X tiled per wave slice", 4096, 4096, 1, 1}, {"qkv all up front, X tiled", 6144, 4096, 0, 1},
                                                          {"gate_up, X tiled", 28672, 4096, 2, 1}, {"down_proj, X tiled", 4096, 14336, 3, 1}, {"o_proj again (row-major)", 4096, 4096, 1, 0}};
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float* dummy; hipMalloc(&dummy, 4096); hipMemset(dummy, 0, 4096);
  for (auto& g : gs) {
    const int N = g.N, K = g.K;
    char *x, *w; void* y; float *sx, *sw, *slabs, *cs; long long* tl; int64_t *pos, *loc; void *kb, *vb;
    hipMalloc(&x, (size_t)M * K); hipMalloc(&w, (size_t)L * N * K); hipMalloc(&y, (size_t)M * N * 2); hipMalloc(&slabs, (size_t)8 * M * N * 4);
    hipMalloc(&sx, M * 4); hipMalloc(&sw, N * 4); hipMalloc(&tl, 1024 * 8 * 8 * 8); hipMalloc(&cs, 8192 * 128 * 4);
    hipMalloc(&pos, M * 8); hipMalloc(&loc, M * 8); hipMalloc(&kb, (size_t)4096 * 8 * 128 * 2); hipMalloc(&vb, (size_t)4096 * 8 * 128 * 2);
    hipMemset(x, 0x38, (size_t)M * K); hipMemset(w, 0x38, (size_t)L * N * K); hipMemset(sx, 0, M * 4); hipMemset(sw, 0, N * 4);
    hipMemset(cs, 0, 8192 * 128 * 4); hipMemset(tl, 0, 1024 * 8 * 8 * 8);
    std::vector<int64_t> hp(M), hl(M); for (int i = 0; i < M; ++i) { hp[i] = 2048 + i; hl[i] = 1 + 37 * i; }
    hipMemcpy(pos, hp.data(), M * 8, hipMemcpyHostToDevice); hipMemcpy(loc, hl.data(), M * 8, hipMemcpyHostToDevice);
    SkinnyParams p;
    p.x = x; p.x_stride = K; p.y = y; p.y_stride = g.kind == 2 ? N / 2 : N; p.sx = sx; p.sw = sw; p.bias = nullptr; p.M = M; p.N = N; p.K = K; p.kbytes = K;
    p.w_stride = K; p.tl = tl; p.x_tiled = g.tiled;
    EpiParams ep; ep.positions = pos; ep.cos_sin = cs; ep.loc = loc; ep.k_buf = kb; ep.v_buf = vb; ep.k_slot_stride = 8 * 128; ep.v_slot_stride = 8 * 128; ep.hq = 32; ep.hkv = 8;
    auto launch_one = [&](int l) {
      p.w = w + (size_t)l * N * K;
      hipLaunchKernelGGL(small_k, dim3(32), dim3(256), 0, st, dummy);
      if (g.kind == 0) { g_skinny_allin = 1; launch_v2_epi<ES_FP8, 2, 8, __bf16, EPI_ROPE>(p, ep, 8, st); }
      if (g.kind == 4) { g_skinny_allin = 0; launch_v2_epi<ES_FP8, 2, 8, __bf16, EPI_ROPE>(p, ep, 8, st); }
      if (g.kind == 1) launch_v2<ES_FP8, 2, 8, __bf16>(p, 1, nullptr, st);
      if (g.kind == 2) launch_v2_epi<ES_FP8, 2, 8, __bf16, EPI_SILU>(p, ep, 16, st);
      if (g.kind == 3) { SkinnyParams q = p; q.sx = nullptr; q.sw = nullptr; launch_v2<ES_FP8, 2, 8, __bf16>(q, 4, slabs, st); }
    };
    hipGraph_t gr; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int l = 0; l < L; ++l) launch_one(l);
    hipStreamEndCapture(st, &gr);
    hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    const int wgs = g.kind == 3 ? 256 : (g.kind == 1 ? 256 : 256);
    std::vector<long long> h((size_t)wgs * 8 * 8);
    hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost);   // stamps of the LAST launch of the last replay
    long long t0 = h[0];
    for (int i = 0; i < wgs * 8; ++i) t0 = std::min(t0, h[(size_t)i * 8]);
    printf("%-42s N %5d K %5d: %6.2f us per (small kernel + GEMM) pair in the graph\n", g.name, N, K, best * 1e3 / L);
    const int order[6] = {0, 1, 5, 2, 3, 4};
    const char* nm[6] = {"entry", "X fragments built", "first W tile staged", "last MFMAs issued", "barrier passed", "outputs stored"};
    for (int k = 0; k < 6; ++k) {
      std::vector<long long> v;
      for (int i = 0; i < wgs * 8; ++i) v.push_back(h[(size_t)i * 8 + order[k]] - t0);
      std::sort(v.begin(), v.end());
      printf("    %-22s median %6.2f  p10 %6.2f  p90 %6.2f  max %6.2f us\n", nm[k], v[v.size() / 2] * 0.01, v[v.size() / 10] * 0.01, v[v.size() * 9 / 10] * 0.01, v.back() * 0.01);
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(gr);
    hipFree(x); hipFree(w); hipFree(y); hipFree(slabs); hipFree(sx); hipFree(sw); hipFree(tl); hipFree(cs); hipFree(pos); hipFree(loc); hipFree(kb); hipFree(vb);
  }
  return 0;
}
