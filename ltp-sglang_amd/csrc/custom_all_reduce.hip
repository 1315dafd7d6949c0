// One-shot P2P all-reduce over IPC-mapped peer buffers (xGMI between the 8 GPUs of a node; between processes of ONE GPU in
// the validation rig), gfx950.
//
// Replaces, for the tensor-parallel decode path (two [M, hidden] bf16 sums per layer: 256 KiB at Llama-3-8B batch 32,
// 2 MiB at Llama-3-70B batch 128), the reference's custom all-reduce
//   sgl-kernel/csrc/allreduce/custom_all_reduce_hip.cuh:261,294 (cross_device_reduce_1stage), :543-549 (dispatch),
//   python/sglang/srt/distributed/device_communicators/custom_all_reduce.py (buffer registration, should_custom_ar),
// which GroupCoordinator.all_reduce prefers over RCCL for small messages (parallel_state.py:480-500): at these sizes a ring
// collective is per-hop latency bound, a one-shot read of every peer is one xGMI round trip.
//
// Design (written for this hardware, not translated):
//  * every rank owns ONE uncached device allocation (hipDeviceMallocUncached: neither the owner's nor a peer's L2 keeps its
//    lines, so a peer's read always sees what the owner's drained stores wrote) = signal block + two data halves; peers map it
//    with hipIpcOpenMemHandle;
//  * a call copies the rank's input into its own half `epoch & 1`, drains, and block b raises flag[b][rank] = epoch in EVERY
//    rank's signal block (system-scope release stores, one lane per peer); block b then waits until its own flag[b][r] has
//    reached epoch for all r (system-scope acquire polls, one lane per peer) and sums the peers' halves IN RANK ORDER with f32
//    accumulation -- every rank computes the same bits;
//  * the two halves alternate by call, which removes the closing barrier: a rank can enter call n + 1 (writing the other
//    half) while a slow peer still reads call n, but cannot pass call n + 1's opening barrier -- and so cannot touch half
//    n & 1 again -- before every peer has arrived there, i.e. has finished reading call n;
//  * epochs live in device memory and are advanced by the kernel, so a captured HIP graph replays correctly;
//  * every spin is bounded: a peer that never arrives sets the signal block's error word instead of hanging the GPU.
//  * block b of every rank copies and reduces the SAME element range, so a block only ever reads data that the matching block
//    of a peer published before raising the flag it waits on.
//  * (round 3) that argument needs the packet -> block map to be the SAME in every call that shares a pair of halves: block b may
//    re-enter a half only because the peers' blocks b have left it.  The plain / all-gather kernel maps packet i to block
//    (i / 512) % 64, the fused kernels map row r to a block, the two-stage kernels map 8-KiB chunks to (owner, block): each of
//    these four FAMILIES therefore has its own flags, epochs and data halves (a mixed sequence -- one decode step at batch 1, the
//    next at batch 32, hidden 8192 -- could otherwise overwrite rows a lagging peer's other block is still reading).
//
// Two-stage form (reduce-scatter + all-gather; custom_all_reduce_hip.cuh:294 cross_device_reduce_2stage, dispatch :543-549: at 8
// ranks the reference leaves the one-stage kernel at 256 KiB): every rank sums only the chunks it owns (owner of 8-KiB chunk c =
// c % world -- a map that does not depend on the message size) from the peers' published operands, publishes those sums, and
// after a second flag round every rank collects the other owners' sums: 2 x 7/8 of the message come in per rank instead of
// 7 x, and every element is summed ONCE (in rank order), so all ranks hold the same bits.  The fused form gives ROWS to owners
// (owner of row r = r % world): the owner finishes the row -- sum, residual add, RMSNorm, per-token quant -- and the others
// collect the finished row (residual, normed, fp8 + scale) instead of the sum, so the row arithmetic also runs once.
#include "row_helpers.h"

namespace {

constexpr int kMaxRanks = 8;
constexpr int kMaxBlocks = 64;
constexpr int kThreads = 512;
constexpr uint32_t kSpinLimit = 1u << 26;  // polls of ~64 ns each: a few seconds, then give up

constexpr int kFamilies = 4;  // 0: one-shot plain + all-gather, 1: one-shot fused, 2: two-stage plain, 3: two-stage fused
enum { FAM_PLAIN = 0, FAM_FUSED = 1, FAM_2S = 2, FAM_2S_FUSED = 3 };
struct alignas(256) CarSignal {
  uint32_t flag[kFamilies][kMaxBlocks][kMaxRanks];  // written by the peers: flag[f][b][r] = last epoch rank r's block b has published
  uint32_t epoch[kFamilies][kMaxBlocks];            // own counter per family and block
  uint32_t error;                                   // set when a spin ran out
};
// Data layout behind the signal block, in units of max_bytes (H): family 0: 2 halves; 1: 2; 2: 2 operand halves + 2 result halves;
// 3: 2 operand halves + 2 result halves of 3 H (a finished row = residual + normed + fp8 + scale: at most 5 h + 16 <= 6 h bytes).
constexpr int kDataUnits = 2 + 2 + 4 + 2 + 6;
__host__ __device__ inline int64_t fam_off(int fam, int64_t H) {
  const int64_t units = fam == FAM_PLAIN ? 0 : fam == FAM_FUSED ? 2 : fam == FAM_2S ? 4 : 8;
  return (int64_t)sizeof(CarSignal) + units * H;
}

struct CarParams {
  char* buf[kMaxRanks];       // each rank's allocation (signal block first)
  void* inout;
  void* gather_out;           // all-gather: [rows, world * row_bytes] output; NULL = all-reduce
  int64_t row16;              // all-gather: 16-byte packets per input row
  int64_t n16;                // 16-byte packets
  int64_t half_bytes;
  int64_t data_off;           // offset of the family's first data half inside an allocation
  int rank, world;
};

// One flag round of family `fam`, block b: lane r raises flag[fam][b][rank] = epoch at peer r, then waits for peer r's flag here.
// Returns false (and sets the error word) if a peer never arrived.  `failed`: an int in LDS.
__device__ __forceinline__ bool flag_round(char* const* buf, int rank, int world, int fam, int b, uint32_t epoch, int tid, int* failed) {
  CarSignal* me = (CarSignal*)buf[rank];
  if (tid == 0) *failed = 0;
  __syncthreads();
  if (tid < world) {
    CarSignal* peer = (CarSignal*)buf[tid];
    __hip_atomic_store(&peer->flag[fam][b][rank], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t spins = 0;
    // (epochs only grow; "< epoch" tolerates a peer that is already one round ahead)
    while ((int32_t)(__hip_atomic_load(&me->flag[fam][b][tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > kSpinLimit) {
        *failed = 1;
        me->error = 1u;
        break;
      }
    }
  }
  __syncthreads();
  return *failed == 0;
}

template <typename T>
__device__ __forceinline__ void accumulate(float (&acc)[8], const u32x4_t& v) {
  struct P8 { T v[8]; };
  const P8 x = __builtin_bit_cast(P8, v);
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] += (float)x.v[j];
}
template <>
__device__ __forceinline__ void accumulate<float>(float (&acc)[8], const u32x4_t& v) {
  struct F4 { float v[4]; };
  const F4 x = __builtin_bit_cast(F4, v);   // (element-wise bit_cast of v[j] miscompiles: every lane reads element 0)
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] += x.v[j];
}
template <typename T>
__device__ __forceinline__ u32x4_t pack(const float (&acc)[8]) {
  struct P8 { T v[8]; };
  P8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o.v[j] = (T)acc[j];
  return __builtin_bit_cast(u32x4_t, o);
}
template <>
__device__ __forceinline__ u32x4_t pack<float>(const float (&acc)[8]) {
  struct F4 { float v[4]; };
  const F4 o = {{acc[0], acc[1], acc[2], acc[3]}};
  return __builtin_bit_cast(u32x4_t, o);
}

// 16-byte accesses that bypass every cache level that is not coherent with a peer (sc0 sc1 = system scope): the importing
// process's mapping of a peer's allocation need not inherit the owner's uncached memory type, so neither side relies on it.
constexpr int kSysScope = 1 | 16;  // buffer cache policy: sc0 (bit 0) + sc1 (bit 4)
__device__ __forceinline__ u32x4_t load_sys(const char* base, int64_t byte_off) {
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + byte_off), 0, 16, 0x00020000);
  return __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, 0, 0, kSysScope));
}
__device__ __forceinline__ void store_sys(char* base, int64_t byte_off, const u32x4_t& v) {
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + byte_off), 0, 16, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rsrc, 0, 0, kSysScope);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void one_shot_all_reduce_kernel(const CarParams p) {
  const int b = blockIdx.x, tid = threadIdx.x;
  CarSignal* me = (CarSignal*)p.buf[p.rank];
  const uint32_t epoch = me->epoch[FAM_PLAIN][b] + 1;
  const int64_t half = p.data_off + (int64_t)(epoch & 1u) * p.half_bytes;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  // 1. publish this rank's operand in its own (uncached) half
  for (int64_t i = (int64_t)b * kThreads + tid; i < p.n16; i += stride)
    store_sys(p.buf[p.rank], half + i * 16, ((const u32x4_t*)p.inout)[i]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the stores above have left this GPU's write path
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // 2. one lane per peer: raise my flag there, then wait for that peer's flag here
  __shared__ int failed_lds;
  const bool failed = !flag_round(p.buf, p.rank, p.world, FAM_PLAIN, b, epoch, tid, &failed_lds);
  // 3a. all-gather along the last dimension: out[row][r * row_bytes + ...] = rank r's input row
  if (!failed && p.gather_out != nullptr) {
    for (int64_t i = (int64_t)b * kThreads + tid; i < p.n16; i += stride) {
      const int64_t row = i / p.row16, col = i - row * p.row16;
      for (int r = 0; r < p.world; ++r)
        ((u32x4_t*)p.gather_out)[(row * p.world + r) * p.row16 + col] = load_sys(p.buf[r], half + i * 16);
    }
  }
  // 3. sum the peers' halves in rank order (identical arithmetic on every rank)
  if (!failed && p.gather_out == nullptr) {
    for (int64_t i = (int64_t)b * kThreads + tid; i < p.n16; i += stride) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < p.world; ++r) {
        accumulate<T>(acc, load_sys(p.buf[r], half + i * 16));
      }
      ((u32x4_t*)p.inout)[i] = pack<T>(acc);
    }
  }
  __syncthreads();
  if (tid == 0) me->epoch[FAM_PLAIN][b] = epoch;
}

// All-reduce fused with the op sequence that follows it on the decode path (linear.py:1302-1303 -> layernorm.py:135-171 ->
// per_token_quant_fp8.cu): x = all_reduce(partial); residual += x; y = rmsnorm(residual) * w; (y_q, y_s) = per-token fp8 quant.
// One 256-thread workgroup per token row (rows beyond the grid are walked with a stride); block b exchanges flags with block b
// of every peer exactly as the plain kernel does, so the row a block sums is the row the peers' matching blocks published.
// Arithmetic = the plain all-reduce's (f32 sum in rank order, one rounding to T) followed by add_rmsnorm_quant_kernel's, so the
// result is bit-identical to the unfused pair; what it removes is one launch and one [M, hidden] round trip per all-reduce site.
struct CarNormParams {
  char* buf[kMaxRanks];
  const void* partial;   // [rows, hidden] T: this rank's partial sums -- or NULL and the next four (round 4):
  const float* slabs;    // [nslabs][rows][hidden] f32 split-K partial sums of this rank's GEMM (raw accumulators, no scales yet)
  int nslabs;            // the kernel forms x = T((slab 0 + slab 1 + ...) * slab_sx[row] * slab_sw[col]) -- add_rmsnorm_quant_kernel's
  const float* slab_sx;  // [rows] or NULL     slab path, i.e. what the GEMM's own reduce launch would have written -- while it
  const float* slab_sw;  // [hidden] or NULL   publishes the row, so that launch and its [rows, hidden] round trip go away
  void* residual;        // [rows, hidden] T, updated in place (may be NULL: no residual add)
  const void* weight;    // [hidden] T
  void* out_norm;        // optional [rows, hidden] T
  uint8_t* out_q;        // optional [rows, hidden] e4m3
  float* out_s;          // [rows]
  int64_t half_bytes, data_off;
  int rows, hidden;
  float eps;
  int rank, world;
};

// packet i (8 elements) of row `row` of this rank's operand: from `partial`, or formed from the split-K slabs
template <typename T>
__device__ __forceinline__ u32x4_t car_operand_packet(const CarNormParams& p, int row, int nvec, int i) {
  if (p.slabs == nullptr) return *((const u32x4_t*)p.partial + (int64_t)row * nvec + i);
  const int64_t plane = (int64_t)p.rows * p.hidden;
  const float* sp = p.slabs + (int64_t)row * p.hidden + i * 8;
  float f[8];
  {
    const f32x4_t a0 = *(const f32x4_t*)sp, a1 = *(const f32x4_t*)(sp + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = a0[j]; f[4 + j] = a1[j]; }
  }
  for (int sI = 1; sI < p.nslabs; ++sI) {   // ascending slab order: add_rmsnorm_quant_kernel's
    const f32x4_t a0 = *(const f32x4_t*)(sp + sI * plane), a1 = *(const f32x4_t*)(sp + sI * plane + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] += a0[j]; f[4 + j] += a1[j]; }
  }
  const float sxm = p.slab_sx ? p.slab_sx[row] : 1.0f;
  V8<T> o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o.v[j] = (T)round_via<T>(f[j] * sxm * (p.slab_sw ? p.slab_sw[i * 8 + j] : 1.0f));
  return __builtin_bit_cast(u32x4_t, o);
}

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void all_reduce_add_rmsnorm_quant_kernel(const CarNormParams p) {
  __shared__ float red[4];
  __shared__ int failed_lds;
  const int b = blockIdx.x, tid = threadIdx.x;
  CarSignal* me = (CarSignal*)p.buf[p.rank];
  const uint32_t epoch = me->epoch[FAM_FUSED][b] + 1;
  const int64_t half = p.data_off + (int64_t)(epoch & 1u) * p.half_bytes;
  const int nvec = p.hidden / 8;
  // 1. publish this rank's rows b, b + grid, ...
  for (int row = b; row < p.rows; row += gridDim.x)
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
      const int i = tid + it * 256;
      if (i < nvec) store_sys(p.buf[p.rank], half + ((int64_t)row * nvec + i) * 16, car_operand_packet<T>(p, row, nvec, i));
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const bool failed = !flag_round(p.buf, p.rank, p.world, FAM_FUSED, b, epoch, tid, &failed_lds);
  if (!failed) {
    for (int row = b; row < p.rows; row += gridDim.x) {
      // operands first (norm weight, residual, every peer's packet), then the arithmetic of the two unfused kernels
      V8<T> wreg[MAXV], rreg[MAXV];
      float vals[MAXV][8];
      float ss = 0.f;
#pragma unroll
      for (int it = 0; it < MAXV; ++it) {
        const int i = tid + it * 256;
        const int ic = i < nvec ? i : 0;
        wreg[it] = ld8((const T*)p.weight + ic * 8);
        if (p.residual) rreg[it] = ld8((const T*)p.residual + (int64_t)row * p.hidden + ic * 8);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < p.world; ++r) accumulate<T>(acc, load_sys(p.buf[r], half + ((int64_t)row * nvec + ic) * 16));
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[it][j] = round_via<T>(acc[j]);   // = the all-reduce's output element
      }
#pragma unroll
      for (int it = 0; it < MAXV; ++it) {
        const int i = tid + it * 256;
        if (i < nvec) {
          if (p.residual) {
            V8<T> ro;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              vals[it][j] += (float)rreg[it].v[j];
              ro.v[j] = (T)vals[it][j];
            }
            st8((T*)p.residual + (int64_t)row * p.hidden + i * 8, ro);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) ss = fmaf(vals[it][j], vals[it][j], ss);  // (explicit: a contraction hipcc may or may not make per kernel)
        }
      }
      const float var = block_sum(ss, red) / (float)p.hidden;
      const float rs = 1.0f / sqrtf(var + p.eps);
#pragma unroll
      for (int it = 0; it < MAXV; ++it) {
        const int i = tid + it * 256;
        if (i < nvec) {
          V8<T> o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            vals[it][j] = round_via<T>((vals[it][j] * rs) * (float)wreg[it].v[j]);
            o.v[j] = (T)vals[it][j];
          }
          if (p.out_norm) st8((T*)p.out_norm + (int64_t)row * p.hidden + i * 8, o);
        }
      }
      if (p.out_q) quant_row<MAXV>(vals, nvec, p.out_q + (int64_t)row * p.hidden, p.out_s + row, red);
      __syncthreads();   // red is reused by the next row
    }
  }
  __syncthreads();
  if (tid == 0) me->epoch[FAM_FUSED][b] = epoch;
}

// ---- two-stage all-reduce (reduce-scatter + all-gather), see the header ----
constexpr int kChunk = kThreads;  // packets per chunk = one block iteration (8 KiB)

template <typename T>
__global__ __launch_bounds__(kThreads) void two_stage_all_reduce_kernel(const CarParams p) {
  __shared__ int failed_lds;
  const int b = blockIdx.x, tid = threadIdx.x, G = gridDim.x, W = p.world;
  CarSignal* me = (CarSignal*)p.buf[p.rank];
  const uint32_t e0 = me->epoch[FAM_2S][b];
  const int64_t par = (int64_t)((e0 >> 1) & 1u);
  const int64_t in_half = p.data_off + par * p.half_bytes, res_half = p.data_off + (2 + par) * p.half_bytes;
  const int64_t nchunks = (p.n16 + kChunk - 1) / kChunk, nq = (nchunks + W - 1) / W;  // chunk c = q * W + owner
  // 1. publish this rank's whole operand (every owner will read its own chunks of it)
  for (int64_t q = b; q < nq; q += G)
    for (int r = 0; r < W; ++r) {
      const int64_t i = (q * W + r) * kChunk + tid;
      if (i < p.n16) store_sys(p.buf[p.rank], in_half + i * 16, ((const u32x4_t*)p.inout)[i]);
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bool ok = flag_round(p.buf, p.rank, W, FAM_2S, b, e0 + 1, tid, &failed_lds);
  // 2. reduce-scatter: the chunks this rank owns, summed in rank order, published in the result half
  if (ok) {
    for (int64_t q = b; q < nq; q += G) {
      const int64_t i = (q * W + p.rank) * kChunk + tid;
      if (i < p.n16) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < W; ++r) accumulate<T>(acc, load_sys(p.buf[r], in_half + i * 16));
        store_sys(p.buf[p.rank], res_half + i * 16, pack<T>(acc));
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ok = flag_round(p.buf, p.rank, W, FAM_2S, b, e0 + 2, tid, &failed_lds) && ok;
  // 3. all-gather: every owner's sums (its own included: the same bits everywhere)
  if (ok) {
    for (int64_t q = b; q < nq; q += G)
      for (int r = 0; r < W; ++r) {
        const int64_t i = (q * W + r) * kChunk + tid;
        if (i < p.n16) ((u32x4_t*)p.inout)[i] = load_sys(p.buf[r], res_half + i * 16);
      }
  }
  __syncthreads();
  if (tid == 0) me->epoch[FAM_2S][b] = e0 + 2;
}

// Two-stage form of all_reduce_add_rmsnorm_quant_kernel: the owner of a row (row % world) sums it, adds the residual, normalises
// and quantises it ONCE and publishes the finished row; the other ranks collect it.  Record of a finished row in the result
// half: [residual T x h | normed T x h | fp8 x h | scale f32, padded to 16 bytes].  Same arithmetic as the one-shot fused kernel
// on the owner, so bit-identical to it and to the unfused pair.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void two_stage_all_reduce_add_rmsnorm_quant_kernel(const CarNormParams p) {
  __shared__ float red[4];
  __shared__ int failed_lds;
  const int b = blockIdx.x, tid = threadIdx.x, G = gridDim.x, W = p.world;
  CarSignal* me = (CarSignal*)p.buf[p.rank];
  const uint32_t e0 = me->epoch[FAM_2S_FUSED][b];
  const int64_t par = (int64_t)((e0 >> 1) & 1u);
  const int64_t in_half = p.data_off + par * p.half_bytes, res_half = p.data_off + 2 * p.half_bytes + par * 3 * p.half_bytes;
  const int nvec = p.hidden / 8;
  const int64_t rec = (int64_t)p.hidden * 5 + 16;  // bytes of a finished row's record
  const int nq = (p.rows + W - 1) / W;               // row = q * W + owner
  // 1. publish this rank's partial rows
  for (int q = b; q < nq; q += G)
    for (int r = 0; r < W; ++r) {
      const int row = q * W + r;
      if (row < p.rows) {
#pragma unroll
        for (int it = 0; it < MAXV; ++it) {
          const int i = tid + it * 256;
          if (i < nvec) store_sys(p.buf[p.rank], in_half + ((int64_t)row * nvec + i) * 16, car_operand_packet<T>(p, row, nvec, i));
        }
      }
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bool ok = flag_round(p.buf, p.rank, W, FAM_2S_FUSED, b, e0 + 1, tid, &failed_lds);
  // 2. the rows this rank owns: all-reduce, add, RMSNorm, quant (add_rmsnorm_quant_kernel's arithmetic), local outputs + record
  if (ok) {
    for (int q = b; q < nq; q += G) {
      const int row = q * W + p.rank;
      if (row >= p.rows) continue;  // (block-uniform)
      char* recp = p.buf[p.rank] + res_half + (int64_t)row * rec;
      V8<T> wreg[MAXV], rreg[MAXV];
      float vals[MAXV][8];
      float ss = 0.f;
#pragma unroll
      for (int it = 0; it < MAXV; ++it) {
        const int i = tid + it * 256;
        const int ic = i < nvec ? i : 0;
        wreg[it] = ld8((const T*)p.weight + ic * 8);
        if (p.residual) rreg[it] = ld8((const T*)p.residual + (int64_t)row * p.hidden + ic * 8);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < W; ++r) accumulate<T>(acc, load_sys(p.buf[r], in_half + ((int64_t)row * nvec + ic) * 16));
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[it][j] = round_via<T>(acc[j]);   // = the all-reduce's output element
      }
#pragma unroll
      for (int it = 0; it < MAXV; ++it) {
        const int i = tid + it * 256;
        if (i < nvec) {
          V8<T> ro;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (p.residual) vals[it][j] += (float)rreg[it].v[j];
            ro.v[j] = (T)vals[it][j];
          }
          if (p.residual) st8((T*)p.residual + (int64_t)row * p.hidden + i * 8, ro);
          store_sys(recp, (int64_t)i * 16, __builtin_bit_cast(u32x4_t, ro));
#pragma unroll
          for (int j = 0; j < 8; ++j) ss = fmaf(vals[it][j], vals[it][j], ss);  // (explicit: a contraction hipcc may or may not make per kernel)
        }
      }
      const float var = block_sum(ss, red) / (float)p.hidden;
      const float rs = 1.0f / sqrtf(var + p.eps);
      float amax = 0.f;
#pragma unroll
      for (int it = 0; it < MAXV; ++it) {
        const int i = tid + it * 256;
        if (i < nvec) {
          V8<T> o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            vals[it][j] = round_via<T>((vals[it][j] * rs) * (float)wreg[it].v[j]);
            o.v[j] = (T)vals[it][j];
            amax = fmaxf(amax, fabsf(vals[it][j]));
          }
          if (p.out_norm) st8((T*)p.out_norm + (int64_t)row * p.hidden + i * 8, o);
          store_sys(recp, (int64_t)p.hidden * 2 + (int64_t)i * 16, __builtin_bit_cast(u32x4_t, o));
        }
      }
      // per-token quant (quant_row's arithmetic), 16 fp8 bytes per store: threads pair up
      amax = block_max(amax, red);
      const float scale = amax / kFp8Max;
      const float inv = (scale == 0.f) ? 0.f : 1.0f / scale;
#pragma unroll
      for (int it = 0; it < MAXV; ++it) {
        const int i = tid + it * 256;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = clamp448(vals[it][j] * inv);
        const u32x2_t qv = pack8_fp8(f);
        if (i < nvec) {
          if (p.out_q) *(u32x2_t*)(p.out_q + (int64_t)row * p.hidden + i * 8) = qv;
          __hip_atomic_store((unsigned long long*)(recp + (int64_t)p.hidden * 4 + (int64_t)i * 8), ((unsigned long long)qv[1] << 32) | qv[0],
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      if (tid == 0) {
        if (p.out_s) p.out_s[row] = scale;
        __hip_atomic_store((uint32_t*)(recp + (int64_t)p.hidden * 5), __builtin_bit_cast(uint32_t, scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __syncthreads();   // red is reused by the next row
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ok = flag_round(p.buf, p.rank, W, FAM_2S_FUSED, b, e0 + 2, tid, &failed_lds) && ok;
  // 3. collect the other owners' finished rows
  if (ok) {
    for (int q = b; q < nq; q += G)
      for (int r = 0; r < W; ++r) {
        const int row = q * W + r;
        if (r == p.rank || row >= p.rows) continue;
        const char* recp = p.buf[r] + res_half + (int64_t)row * rec;
#pragma unroll
        for (int it = 0; it < MAXV; ++it) {
          const int i = tid + it * 256;
          if (i < nvec) {
            if (p.residual) *((u32x4_t*)p.residual + (int64_t)row * nvec + i) = load_sys(recp, (int64_t)i * 16);
            if (p.out_norm) *((u32x4_t*)p.out_norm + (int64_t)row * nvec + i) = load_sys(recp, (int64_t)p.hidden * 2 + (int64_t)i * 16);
          }
        }
        if (p.out_q) {
          for (int i = tid; i < p.hidden / 16; i += 256)
            *((u32x4_t*)(p.out_q + (int64_t)row * p.hidden) + i) = load_sys(recp, (int64_t)p.hidden * 4 + (int64_t)i * 16);
          if (tid == 0)
            p.out_s[row] = __builtin_bit_cast(float, __hip_atomic_load((const uint32_t*)(recp + (int64_t)p.hidden * 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        }
      }
  }
  __syncthreads();
  if (tid == 0) me->epoch[FAM_2S_FUSED][b] = e0 + 2;
}

// custom_all_reduce_hip.cuh:543-549: two ranks always take the one-stage kernel; up to 4 ranks below 512 KiB; up to 8 below 256 KiB
inline bool use_two_stage(int world, int64_t bytes) {
  if (world <= 2) return false;
  if (world <= 4) return bytes >= 512 * 1024;
  return bytes >= 256 * 1024;
}

}  // namespace

// Allocates a rank's uncached buffer (signal block + the data halves of the four kernel families: 16 x max_bytes), zeroes the
// signal block and returns its 64-byte IPC handle.
extern "C" int sgl_mi355_car_alloc(int64_t max_bytes, void** ptr_out, void* handle_out) {
  SGL_CHECK(max_bytes > 0 && max_bytes % 16 == 0 && ptr_out && handle_out, "car_alloc: bad arguments");
  const size_t total = sizeof(CarSignal) + (size_t)kDataUnits * (size_t)max_bytes;
  void* ptr = nullptr;
  hipError_t e = hipExtMallocWithFlags(&ptr, total, hipDeviceMallocUncached);
  SGL_CHECK(e == hipSuccess, "car_alloc: hipExtMallocWithFlags(%zu, uncached) failed: %s", total, hipGetErrorString(e));
  e = hipMemset(ptr, 0, sizeof(CarSignal));
  SGL_CHECK(e == hipSuccess, "car_alloc: hipMemset failed: %s", hipGetErrorString(e));
  hipIpcMemHandle_t h;
  e = hipIpcGetMemHandle(&h, ptr);
  SGL_CHECK(e == hipSuccess, "car_alloc: hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 is required on this driver)",
            hipGetErrorString(e));
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  memcpy(handle_out, &h, sizeof(h));
  *ptr_out = ptr;
  (void)hipDeviceSynchronize();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_car_open(const void* handle, void** ptr_out) {
  SGL_CHECK(handle && ptr_out, "car_open: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* ptr = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
  SGL_CHECK(e == hipSuccess, "car_open: hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
  *ptr_out = ptr;
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_car_close(void* peer_ptr) {
  const hipError_t e = hipIpcCloseMemHandle(peer_ptr);
  SGL_CHECK(e == hipSuccess, "car_close: hipIpcCloseMemHandle failed: %s", hipGetErrorString(e));
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_car_free(void* own_ptr) {
  const hipError_t e = hipFree(own_ptr);
  SGL_CHECK(e == hipSuccess, "car_free: hipFree failed: %s", hipGetErrorString(e));
  return SGL_MI355_OK;
}

// Reads (and clears) the error word of a rank's own signal block: non-zero = a spin ran out since the last call.
extern "C" int sgl_mi355_car_error(void* own_ptr) {
  uint32_t v = 0;
  CarSignal* s = (CarSignal*)own_ptr;
  if (hipMemcpy(&v, &s->error, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (v) (void)hipMemset(&s->error, 0, sizeof(v));
  return (int)v;
}

// In-place sum of `inout` (num_elements of dtype bf16 / f16 / f32, 16-byte aligned, byte count a multiple of 16 and at most
// the max_bytes of car_alloc) over the `world` ranks whose allocations are peer_bufs[0 .. world) (this rank's own pointer at
// [rank], the others as returned by car_open).  Every rank must call with the same size, in the same order.
// algo: 0 = the reference's dispatch rule (custom_all_reduce_hip.cuh:543-549: one stage below 512 KiB at <= 4 ranks / 256 KiB at
// <= 8 ranks, always at 2), 1 = one-shot, 2 = two-stage (reduce-scatter + all-gather).  All ranks must pass the same value.
extern "C" int sgl_mi355_car_all_reduce_algo(void* inout, int64_t num_elements, int dtype, const void* const* peer_bufs, int rank,
                                             int world, int64_t max_bytes, int algo, void* stream) {
  SGL_CHECK(inout && peer_bufs, "car_all_reduce: null pointer");
  SGL_CHECK(algo >= 0 && algo <= 2, "car_all_reduce: algo %d (0 = rule, 1 = one-shot, 2 = two-stage)", algo);
  SGL_CHECK(world >= 2 && world <= kMaxRanks && rank >= 0 && rank < world, "car_all_reduce: rank %d / world %d unsupported", rank, world);
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16 || dtype == SGL_F32, "car_all_reduce: dtype code %d unsupported", dtype);
  const int64_t bytes = num_elements * (dtype == SGL_F32 ? 4 : 2);
  SGL_CHECK(bytes > 0 && bytes % 16 == 0 && bytes <= max_bytes && ((uintptr_t)inout % 16) == 0,
            "car_all_reduce: %lld bytes must be a positive multiple of 16, at most %lld, 16-byte aligned", (long long)bytes, (long long)max_bytes);
  CarParams p;
  for (int r = 0; r < kMaxRanks; ++r) p.buf[r] = (char*)(r < world ? peer_bufs[r] : peer_bufs[0]);
  p.inout = inout;
  p.gather_out = nullptr;
  p.row16 = 1;
  p.n16 = bytes / 16;
  p.half_bytes = max_bytes;
  p.rank = rank;
  p.world = world;
  hipStream_t st = (hipStream_t)stream;
  const bool two = algo == 2 || (algo == 0 && use_two_stage(world, bytes));
  if (two) {
    p.data_off = fam_off(FAM_2S, max_bytes);
    // chunk c = q * world + owner; block = q % grid: the grid is a function of the size only, the same on every rank
    const int64_t nq = ((p.n16 + kChunk - 1) / kChunk + world - 1) / world;
    const unsigned blocks = (unsigned)(nq < 1 ? 1 : (nq > kMaxBlocks ? kMaxBlocks : nq));
    if (dtype == SGL_BF16) hipLaunchKernelGGL((two_stage_all_reduce_kernel<__bf16>), dim3(blocks), dim3(kThreads), 0, st, p);
    else if (dtype == SGL_F16) hipLaunchKernelGGL((two_stage_all_reduce_kernel<_Float16>), dim3(blocks), dim3(kThreads), 0, st, p);
    else hipLaunchKernelGGL((two_stage_all_reduce_kernel<float>), dim3(blocks), dim3(kThreads), 0, st, p);
  } else {
    p.data_off = fam_off(FAM_PLAIN, max_bytes);
    // the grid is a function of the size only: every rank launches the same number of blocks
    const int64_t want = (p.n16 + kThreads - 1) / kThreads;
    const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > kMaxBlocks ? kMaxBlocks : want));
    if (dtype == SGL_BF16) hipLaunchKernelGGL((one_shot_all_reduce_kernel<__bf16>), dim3(blocks), dim3(kThreads), 0, st, p);
    else if (dtype == SGL_F16) hipLaunchKernelGGL((one_shot_all_reduce_kernel<_Float16>), dim3(blocks), dim3(kThreads), 0, st, p);
    else hipLaunchKernelGGL((one_shot_all_reduce_kernel<float>), dim3(blocks), dim3(kThreads), 0, st, p);
  }
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_car_all_reduce(void* inout, int64_t num_elements, int dtype, const void* const* peer_bufs, int rank,
                                        int world, int64_t max_bytes, void* stream) {
  return sgl_mi355_car_all_reduce_algo(inout, num_elements, dtype, peer_bufs, rank, world, max_bytes, 0, stream);
}

// All-gather along the last dimension through the same buffers and protocol: out [rows, world * row_bytes] receives rank r's
// contiguous input [rows, row_bytes] at column offset r * row_bytes (the logits all-gather of a vocab-sharded lm_head,
// python/sglang/srt/layers/logits_processor.py:471-500).  row_bytes % 16 == 0, rows * row_bytes <= max_bytes.
extern "C" int sgl_mi355_car_all_gather(const void* in, void* out, int64_t rows, int64_t row_bytes, const void* const* peer_bufs,
                                        int rank, int world, int64_t max_bytes, void* stream) {
  SGL_CHECK(in && out && peer_bufs, "car_all_gather: null pointer");
  SGL_CHECK(world >= 2 && world <= kMaxRanks && rank >= 0 && rank < world, "car_all_gather: rank %d / world %d unsupported", rank, world);
  const int64_t bytes = rows * row_bytes;
  SGL_CHECK(rows > 0 && row_bytes > 0 && row_bytes % 16 == 0 && bytes <= max_bytes && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0,
            "car_all_gather: rows=%lld row_bytes=%lld must give 16-byte packets within %lld bytes", (long long)rows, (long long)row_bytes,
            (long long)max_bytes);
  CarParams p;
  for (int r = 0; r < kMaxRanks; ++r) p.buf[r] = (char*)(r < world ? peer_bufs[r] : peer_bufs[0]);
  p.inout = (void*)in;
  p.gather_out = out;
  p.row16 = row_bytes / 16;
  p.n16 = bytes / 16;
  p.half_bytes = max_bytes;
  p.data_off = fam_off(FAM_PLAIN, max_bytes);
  p.rank = rank;
  p.world = world;
  const int64_t want = (p.n16 + kThreads - 1) / kThreads;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > kMaxBlocks ? kMaxBlocks : want));
  hipLaunchKernelGGL((one_shot_all_reduce_kernel<__bf16>), dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, p);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// x = all_reduce(partial) ; residual += x (in place, when given) ; y = rmsnorm(residual or x) * weight ; optional outputs
// out_norm (T) and out_q / out_s (per-token e4m3 + f32 scale): sgl_mi355_car_all_reduce followed by
// sgl_mi355_fused_add_rmsnorm_quant_fp8 in one launch, bit-identical to that pair.  rows * hidden * 2 <= max_bytes,
// hidden % 8 == 0, hidden <= 8192.
// algo as for sgl_mi355_car_all_reduce_algo (two-stage needs hidden % 16 == 0, else the one-shot form is taken).  On one
// communicator every fused call must use the same `hidden` (the row -> block map of the no-closing-barrier protocol).
static int car_fused_impl(const void* partial, const float* slabs, int nslabs, const float* slab_sx, const float* slab_sw, void* residual,
                          const void* weight, float eps, void* out_norm, void* out_q, float* out_s, int rows, int hidden, int dtype,
                          const void* const* peer_bufs, int rank, int world, int64_t max_bytes, int algo, void* stream) {
  SGL_CHECK((partial != nullptr) != (slabs != nullptr), "car_all_reduce_add_rmsnorm_quant: exactly one of partial / slabs");
  SGL_CHECK(slabs == nullptr || (nslabs >= 1 && nslabs <= 64 && ((uintptr_t)slabs % 16) == 0), "car_all_reduce_add_rmsnorm_quant: bad slabs (nslabs=%d)", nslabs);
  SGL_CHECK(weight && peer_bufs && (out_norm || out_q) && (!out_q || out_s), "car_all_reduce_add_rmsnorm_quant: null pointer");
  SGL_CHECK(algo >= 0 && algo <= 2, "car_all_reduce_add_rmsnorm_quant: algo %d (0 = rule, 1 = one-shot, 2 = two-stage)", algo);
  SGL_CHECK(world >= 2 && world <= kMaxRanks && rank >= 0 && rank < world, "car_all_reduce_add_rmsnorm_quant: rank %d / world %d unsupported", rank, world);
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "car_all_reduce_add_rmsnorm_quant: dtype must be bf16 or f16");
  SGL_CHECK(rows > 0 && hidden > 0 && hidden % 8 == 0 && hidden <= 8192 && (int64_t)rows * hidden * 2 <= max_bytes &&
                (partial == nullptr || ((uintptr_t)partial % 16) == 0),
            "car_all_reduce_add_rmsnorm_quant: rows=%d hidden=%d unsupported (hidden %% 8, <= 8192, rows * hidden * 2 <= %lld)", rows,
            hidden, (long long)max_bytes);
  CarNormParams p;
  for (int r = 0; r < kMaxRanks; ++r) p.buf[r] = (char*)(r < world ? peer_bufs[r] : peer_bufs[0]);
  p.partial = partial; p.residual = residual; p.weight = weight; p.out_norm = out_norm; p.out_q = (uint8_t*)out_q; p.out_s = out_s;
  p.slabs = slabs; p.nslabs = nslabs; p.slab_sx = slab_sx; p.slab_sw = slab_sw;
  p.half_bytes = max_bytes;
  p.rows = rows; p.hidden = hidden; p.eps = eps; p.rank = rank; p.world = world;
  hipStream_t st = (hipStream_t)stream;
  const bool two = hidden % 16 == 0 && (algo == 2 || (algo == 0 && use_two_stage(world, (int64_t)rows * hidden * 2)));
  if (two) {
    p.data_off = fam_off(FAM_2S_FUSED, max_bytes);
    const int nq = (rows + world - 1) / world;  // row = q * world + owner; block = q % grid
    const unsigned blocks = (unsigned)(nq > kMaxBlocks ? kMaxBlocks : nq);
#define SGL_CARN2(Tt, MV) hipLaunchKernelGGL((two_stage_all_reduce_add_rmsnorm_quant_kernel<Tt, MV>), dim3(blocks), dim3(256), 0, st, p)
    if (dtype == SGL_BF16) {
      if (hidden <= 2048) SGL_CARN2(__bf16, 1); else if (hidden <= 4096) SGL_CARN2(__bf16, 2); else SGL_CARN2(__bf16, 4);
    } else {
      if (hidden <= 2048) SGL_CARN2(_Float16, 1); else if (hidden <= 4096) SGL_CARN2(_Float16, 2); else SGL_CARN2(_Float16, 4);
    }
#undef SGL_CARN2
  } else {
    p.data_off = fam_off(FAM_FUSED, max_bytes);
    const unsigned blocks = (unsigned)(rows > kMaxBlocks ? kMaxBlocks : rows);
#define SGL_CARN(Tt, MV) hipLaunchKernelGGL((all_reduce_add_rmsnorm_quant_kernel<Tt, MV>), dim3(blocks), dim3(256), 0, st, p)
    if (dtype == SGL_BF16) {
      if (hidden <= 2048) SGL_CARN(__bf16, 1); else if (hidden <= 4096) SGL_CARN(__bf16, 2); else SGL_CARN(__bf16, 4);
    } else {
      if (hidden <= 2048) SGL_CARN(_Float16, 1); else if (hidden <= 4096) SGL_CARN(_Float16, 2); else SGL_CARN(_Float16, 4);
    }
#undef SGL_CARN
  }
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_car_all_reduce_add_rmsnorm_quant_algo(const void* partial, void* residual, const void* weight, float eps,
                                                               void* out_norm, void* out_q, float* out_s, int rows, int hidden,
                                                               int dtype, const void* const* peer_bufs, int rank, int world,
                                                               int64_t max_bytes, int algo, void* stream) {
  SGL_CHECK(partial, "car_all_reduce_add_rmsnorm_quant: null pointer");
  return car_fused_impl(partial, nullptr, 0, nullptr, nullptr, residual, weight, eps, out_norm, out_q, out_s, rows, hidden, dtype, peer_bufs,
                        rank, world, max_bytes, algo, stream);
}

// The same with this rank's operand given as the split-K slabs of its GEMM (sgl_mi355_fp8_gemm_slabs / skinny slabs: raw f32
// accumulators [nslabs][rows][hidden]) and the GEMM's scale vectors: x = T((slab 0 + slab 1 + ...) * sx[row] * sw[col]) is formed while
// the row is published -- bit-identical to the GEMM's reduce launch followed by the entry point above.
extern "C" int sgl_mi355_car_all_reduce_add_rmsnorm_quant_slabs(const float* slabs, int nslabs, const float* slab_sx, const float* slab_sw,
                                                                void* residual, const void* weight, float eps, void* out_norm,
                                                                void* out_q, float* out_s, int rows, int hidden, int dtype,
                                                                const void* const* peer_bufs, int rank, int world, int64_t max_bytes,
                                                                int algo, void* stream) {
  SGL_CHECK(slabs, "car_all_reduce_add_rmsnorm_quant_slabs: null pointer");
  return car_fused_impl(nullptr, slabs, nslabs, slab_sx, slab_sw, residual, weight, eps, out_norm, out_q, out_s, rows, hidden, dtype,
                        peer_bufs, rank, world, max_bytes, algo, stream);
}

extern "C" int sgl_mi355_car_all_reduce_add_rmsnorm_quant(const void* partial, void* residual, const void* weight, float eps,
                                                          void* out_norm, void* out_q, float* out_s, int rows, int hidden,
                                                          int dtype, const void* const* peer_bufs, int rank, int world,
                                                          int64_t max_bytes, void* stream) {
  return sgl_mi355_car_all_reduce_add_rmsnorm_quant_algo(partial, residual, weight, eps, out_norm, out_q, out_s, rows, hidden, dtype,
                                                         peer_bufs, rank, world, max_bytes, 0, stream);
}
