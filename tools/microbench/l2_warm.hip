// Microbenchmark (round 4): cross-kernel cache warm-up.  The decode step alternates 5 us row kernels (32
// workgroups, HBM idle) with cold weight streams that pay ~2 us before their first bytes arrive.  Question: if the row kernel is
// launched with 256 workgroups and the spare ones issue default-policy loads of the NEXT stream's first P bytes -- chosen so that
// the lines land in the L2 of the XCD whose consumer workgroup will read them (blockIdx % 8 class) -- how much shorter is the pair?
//   pair = row_k (32 row workgroups: load 2 rows, two dependent reductions, store; + optional prefetch workgroups) -> stream_k
//          (256 x 512 threads, workgroup c reads 64 KiB tiles c, c + 256, ...; nt or default loads), 32 pairs in one HIP graph over
//          distinct buffers (> 512 MiB in rotation: nothing survives in the Infinity Cache from the previous round).
//   variants: base (row grid 32) | idle (row grid 256, spare workgroups exit) | same-XCD prefetch of P MiB | prefetch shifted to
//             the XCD class + 4 (if that gains the same, the gain is the Infinity Cache's, not L2's).
// Kill criterion (stated before the run): < 1 us per pair -> the lever is dropped.
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_warm l2_warm.hip     Run: ./l2_warm
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kTile = 64 * 1024;      // bytes per consumer tile (16 weight rows x 4096 bytes)
constexpr int kRowWgs = 32;

__device__ __forceinline__ uint32_t wave_xor(uint32_t r) {
  for (int m = 32; m >= 1; m >>= 1) r ^= __shfl_xor(r, m, 64);
  return r;
}

// map: 0 = no prefetch, 1 = same XCD class as the consumer, 2 = class + 4
__global__ __launch_bounds__(256) void row_k(const u32x4* __restrict__ x, uint32_t* __restrict__ out, const char* __restrict__ next_w,
                                             int pf_tiles, int map, uint32_t* xcc_log) {
  const int tid = threadIdx.x, b = blockIdx.x;
  __shared__ uint32_t red[4];
  if (b < kRowWgs) {
    if (b == 0 && tid == 0 && xcc_log) *xcc_log = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;
    // add + RMSNorm + quant stand-in: two 8 KiB rows in, two dependent workgroup reductions, 12 KiB out
    const u32x4 a0 = x[(size_t)b * 1024 + tid], a1 = x[(size_t)b * 1024 + 256 + tid];
    const u32x4 r0 = x[(size_t)(b + 32) * 1024 + tid], r1 = x[(size_t)(b + 32) * 1024 + 256 + tid];
    const u32x4 s0 = a0 + r0, s1 = a1 + r1;
    uint32_t v = wave_xor(s0[0] ^ s0[1] ^ s0[2] ^ s0[3] ^ s1[0] ^ s1[1] ^ s1[2] ^ s1[3]);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const uint32_t sum = red[0] ^ red[1] ^ red[2] ^ red[3];
    __syncthreads();
    uint32_t m = wave_xor((s0[0] * sum) ^ (s1[3] + sum));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    const uint32_t mx = red[0] ^ red[1] ^ red[2] ^ red[3];
    u32x4* o = (u32x4*)out + (size_t)b * 768;
    o[tid] = s0 * mx;
    o[256 + tid] = s1 * mx;
    o[512 + tid] = (s0 ^ s1) + mx;
    return;
  }
  if (map == 0) return;
  if (map >= 3) {
    // TLB warm-up only (map = 3 + log2(stride / 4 KiB)): the first eight spare workgroups -- one per XCD class -- touch ONE 16-byte
    // word every `stride` bytes of the WHOLE next weight (pf_tiles * kTile bytes): translations for the consuming XCD, almost no data
    if (b >= kRowWgs + 8) return;
    const size_t stride = (size_t)4096 << (map - 3), total = (size_t)pf_tiles * kTile;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t off = (size_t)tid * stride; off < total; off += 256 * stride) acc ^= *(const u32x4*)(next_w + off);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u) out[1u << 20] = 1;
    return;
  }
  // prefetch workgroup: class = blockIdx % 8 (+ 4 when shifted); tiles t of that class with t < pf_tiles, dealt round-robin over the
  // (gridDim - 32) / 8 workgroups of the class
  const int per_class = (gridDim.x - kRowWgs) / 8;
  const int cls = ((b & 7) + (map == 2 ? 4 : 0)) & 7;
  const int rank = (b - kRowWgs) >> 3;
  u32x4 acc = {0, 0, 0, 0};
  for (int i = rank; 8 * i + cls < pf_tiles; i += per_class) {
    const u32x4* p = (const u32x4*)(next_w + (size_t)(8 * i + cls) * kTile);
#pragma unroll
    for (int j = 0; j < kTile / 16 / 256; ++j) acc ^= p[j * 256 + tid];   // 16 loads of 16 B per thread and tile, default policy
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u) out[1u << 20] = 1;   // never true: keeps the loads
}

template <bool NT>
__global__ __launch_bounds__(512) void stream_k(const char* __restrict__ w, int ntiles, const uint32_t* __restrict__ prev, uint32_t* __restrict__ out,
                                                uint32_t* xcc_log) {
  const int tid = threadIdx.x, c = blockIdx.x;
  if (c == 0 && tid == 0 && xcc_log) *xcc_log = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;
  u32x4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = u32x4{0, 0, 0, 0};
  for (int t = c; t < ntiles; t += gridDim.x) {
    const u32x4* p = (const u32x4*)(w + (size_t)t * kTile);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u32x4 v = NT ? __builtin_nontemporal_load(p + j * 512 + tid) : p[j * 512 + tid];
      acc[j] ^= v;
    }
  }
  u32x4 a = acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ acc[4] ^ acc[5] ^ acc[6] ^ acc[7];
  uint32_t r = wave_xor(a[0] ^ a[1] ^ a[2] ^ a[3]);
  __shared__ uint32_t red[8];
  if ((tid & 63) == 0) red[tid >> 6] = r;
  __syncthreads();
  if (tid == 0) {
    uint32_t t = prev[c & 31];
    for (int k = 0; k < 8; ++k) t ^= red[k];
    out[c] = t;
  }
}

int main(int argc, char** argv) {
  const int N = 32;
  // default: the 8B decode streams; `./l2_warm shard` = the TP-8 shard streams (2 / 3 / 7 / 15 MB: the whole weight fits the L2 slice of
  // its consumers, so the prefetch variants cover ALL of it)
  const bool shard = argc > 1 && argv[1][0] == 's';
  std::vector<size_t> sizes_mb = shard ? std::vector<size_t>{2, 3, 7, 15} : std::vector<size_t>{17, 25, 59, 117};
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  u32x4* x; uint32_t *rout, *sout, *xlog;
  hipMalloc(&x, 64 * 1024 * 16); hipMemset(x, 3, 64 * 1024 * 16);
  hipMalloc(&rout, (1u << 20) * 4 + 64); hipMalloc(&sout, 256 * 4 * N); hipMalloc(&xlog, 4 * 2 * N); hipMemset(xlog, 0xff, 4 * 2 * N);
  for (size_t mb : sizes_mb) {
    const size_t bytes = (mb << 20) / kTile * kTile;
    const int ntiles = (int)(bytes / kTile);
    const int nb = (int)((640 + mb - 1) / mb) < 8 ? 8 : (int)((640 + mb - 1) / mb);
    std::vector<char*> w(nb);
    for (auto& p : w) { hipMalloc(&p, bytes); hipMemset(p, 1, bytes); }
    struct V { const char* name; int grid; int map; int pf_mb; };
    std::vector<V> vs = {{"base: row grid 32            ", 32, 0, 0}, {"idle: row grid 256, no loads ", 256, 0, 0}};
    if (shard) vs.push_back({"prefetch same XCD class (all)", 256, 1, (int)mb});
    else for (int pf : {4, 8}) if ((size_t)pf <= mb) vs.push_back({"prefetch same XCD class      ", 256, 1, pf});
    vs.push_back({"prefetch XCD class + 4       ", 256, 2, shard ? (int)mb : 8});
    vs.push_back({"TLB touch, stride 4 KiB      ", 256, 3, (int)mb});
    vs.push_back({"TLB touch, stride 64 KiB     ", 256, 7, (int)mb});
    vs.push_back({"TLB touch, stride 2 MiB      ", 256, 12, (int)mb});
    vs.push_back({"base again                   ", 32, 0, 0});
    for (int nt = 1; nt >= 1; --nt) {
      for (const V& v : vs) {
        const int pf_tiles = (int)(((size_t)v.pf_mb << 20) / kTile) > ntiles ? ntiles : (int)(((size_t)v.pf_mb << 20) / kTile);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int k = 0; k < N; ++k) {
          hipLaunchKernelGGL(row_k, dim3(v.grid), dim3(256), 0, st, x, rout, w[k % nb], pf_tiles, v.map, xlog + 2 * k);
          if (nt) hipLaunchKernelGGL(stream_k<true>, dim3(256), dim3(512), 0, st, w[k % nb], ntiles, rout, sout + 256 * k, xlog + 2 * k + 1);
          else hipLaunchKernelGGL(stream_k<false>, dim3(256), dim3(512), 0, st, w[k % nb], ntiles, rout, sout + 256 * k, xlog + 2 * k + 1);
        }
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        float best = 1e9f, sum = 0;
        const int reps = 6;
        for (int rep = 0; rep < reps; ++rep) {
          hipEventRecord(e0, st);
          hipGraphLaunch(ge, st);
          hipEventRecord(e1, st); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep >= 1) { sum += ms; if (ms < best) best = ms; }
        }
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
        uint32_t xl[2 * N]; hipMemcpy(xl, xlog, sizeof(xl), hipMemcpyDeviceToHost);
        int same = 0; for (int k = 0; k < N; ++k) same += xl[2 * k] == xl[2 * k + 1];
        printf("%4zu MB %s consumer | %s P=%3d MB : %7.2f us per pair (mean %7.2f) | block 0 on the same XCC in row and stream kernel: %d/%d (xcc %u %u %u %u)\n",
               mb, nt ? "nt     " : "default", v.name, v.pf_mb, best * 1e3 / N, sum / (reps - 1) * 1e3 / N, same, N, xl[0], xl[1], xl[2], xl[3]);
        fflush(stdout);
      }
    }
    for (auto& p : w) hipFree(p);
  }
  return 0;
}
