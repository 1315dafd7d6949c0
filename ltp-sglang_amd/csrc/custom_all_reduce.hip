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
#include "row_helpers.h"

namespace {

constexpr int kMaxRanks = 8;
constexpr int kMaxBlocks = 64;
constexpr int kThreads = 512;
constexpr uint32_t kSpinLimit = 1u << 26;  // polls of ~64 ns each: a few seconds, then give up

struct alignas(128) CarSignal {
  uint32_t flag[kMaxBlocks][kMaxRanks];  // written by the peers: flag[b][r] = last epoch rank r's block b has published
  uint32_t epoch[kMaxBlocks];            // own counter per block
  uint32_t error;                        // set when a spin ran out
};

struct CarParams {
  char* buf[kMaxRanks];       // each rank's allocation (signal block first)
  void* inout;
  void* gather_out;           // all-gather: [rows, world * row_bytes] output; NULL = all-reduce
  int64_t row16;              // all-gather: 16-byte packets per input row
  int64_t n16;                // 16-byte packets
  int64_t half_bytes;
  int64_t data_off;           // offset of the first data half inside an allocation
  int rank, world;
};

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
  const uint32_t epoch = me->epoch[b] + 1;
  const int64_t half = p.data_off + (int64_t)(epoch & 1u) * p.half_bytes;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  // 1. publish this rank's operand in its own (uncached) half
  for (int64_t i = (int64_t)b * kThreads + tid; i < p.n16; i += stride)
    store_sys(p.buf[p.rank], half + i * 16, ((const u32x4_t*)p.inout)[i]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the stores above have left this GPU's write path
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // 2. one lane per peer: raise my flag there, then wait for that peer's flag here
  __shared__ int failed;
  if (tid == 0) failed = 0;
  __syncthreads();
  if (tid < p.world) {
    CarSignal* peer = (CarSignal*)p.buf[tid];
    __hip_atomic_store(&peer->flag[b][p.rank], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t spins = 0;
    // (epochs only grow; "< epoch" tolerates a peer that is already one call ahead)
    while ((int32_t)(__hip_atomic_load(&me->flag[b][tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > kSpinLimit) {
        failed = 1;
        me->error = 1u;
        break;
      }
    }
  }
  __syncthreads();
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
  if (tid == 0) me->epoch[b] = epoch;
}

// All-reduce fused with the op sequence that follows it on the decode path (linear.py:1302-1303 -> layernorm.py:135-171 ->
// per_token_quant_fp8.cu): x = all_reduce(partial); residual += x; y = rmsnorm(residual) * w; (y_q, y_s) = per-token fp8 quant.
// One 256-thread workgroup per token row (rows beyond the grid are walked with a stride); block b exchanges flags with block b
// of every peer exactly as the plain kernel does, so the row a block sums is the row the peers' matching blocks published.
// Arithmetic = the plain all-reduce's (f32 sum in rank order, one rounding to T) followed by add_rmsnorm_quant_kernel's, so the
// result is bit-identical to the unfused pair; what it removes is one launch and one [M, hidden] round trip per all-reduce site.
struct CarNormParams {
  char* buf[kMaxRanks];
  const void* partial;   // [rows, hidden] T: this rank's partial sums
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

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void all_reduce_add_rmsnorm_quant_kernel(const CarNormParams p) {
  __shared__ float red[4];
  __shared__ int failed;
  const int b = blockIdx.x, tid = threadIdx.x;
  CarSignal* me = (CarSignal*)p.buf[p.rank];
  const uint32_t epoch = me->epoch[b] + 1;
  const int64_t half = p.data_off + (int64_t)(epoch & 1u) * p.half_bytes;
  const int nvec = p.hidden / 8;
  // 1. publish this rank's rows b, b + grid, ...
  for (int row = b; row < p.rows; row += gridDim.x)
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
      const int i = tid + it * 256;
      if (i < nvec) store_sys(p.buf[p.rank], half + ((int64_t)row * nvec + i) * 16, *((const u32x4_t*)p.partial + (int64_t)row * nvec + i));
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) failed = 0;
  __syncthreads();
  if (tid < p.world) {
    CarSignal* peer = (CarSignal*)p.buf[tid];
    __hip_atomic_store(&peer->flag[b][p.rank], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t spins = 0;
    while ((int32_t)(__hip_atomic_load(&me->flag[b][tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > kSpinLimit) {
        failed = 1;
        me->error = 1u;
        break;
      }
    }
  }
  __syncthreads();
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
          for (int j = 0; j < 8; ++j) ss += vals[it][j] * vals[it][j];
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
  if (tid == 0) me->epoch[b] = epoch;
}

}  // namespace

// Allocates a rank's uncached buffer (signal block + two data halves of max_bytes each), zeroes the signal block and
// returns its 64-byte IPC handle.
extern "C" int sgl_mi355_car_alloc(int64_t max_bytes, void** ptr_out, void* handle_out) {
  SGL_CHECK(max_bytes > 0 && max_bytes % 16 == 0 && ptr_out && handle_out, "car_alloc: bad arguments");
  const size_t total = sizeof(CarSignal) + 2 * (size_t)max_bytes;
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
extern "C" int sgl_mi355_car_all_reduce(void* inout, int64_t num_elements, int dtype, const void* const* peer_bufs, int rank,
                                        int world, int64_t max_bytes, void* stream) {
  SGL_CHECK(inout && peer_bufs, "car_all_reduce: null pointer");
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
  p.data_off = (int64_t)sizeof(CarSignal);
  p.rank = rank;
  p.world = world;
  // the grid is a function of the size only: every rank launches the same number of blocks
  const int64_t want = (p.n16 + kThreads - 1) / kThreads;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > kMaxBlocks ? kMaxBlocks : want));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SGL_BF16) hipLaunchKernelGGL((one_shot_all_reduce_kernel<__bf16>), dim3(blocks), dim3(kThreads), 0, st, p);
  else if (dtype == SGL_F16) hipLaunchKernelGGL((one_shot_all_reduce_kernel<_Float16>), dim3(blocks), dim3(kThreads), 0, st, p);
  else hipLaunchKernelGGL((one_shot_all_reduce_kernel<float>), dim3(blocks), dim3(kThreads), 0, st, p);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
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
  p.data_off = (int64_t)sizeof(CarSignal);
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
extern "C" int sgl_mi355_car_all_reduce_add_rmsnorm_quant(const void* partial, void* residual, const void* weight, float eps,
                                                          void* out_norm, void* out_q, float* out_s, int rows, int hidden,
                                                          int dtype, const void* const* peer_bufs, int rank, int world,
                                                          int64_t max_bytes, void* stream) {
  SGL_CHECK(partial && weight && peer_bufs && (out_norm || out_q) && (!out_q || out_s), "car_all_reduce_add_rmsnorm_quant: null pointer");
  SGL_CHECK(world >= 2 && world <= kMaxRanks && rank >= 0 && rank < world, "car_all_reduce_add_rmsnorm_quant: rank %d / world %d unsupported", rank, world);
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "car_all_reduce_add_rmsnorm_quant: dtype must be bf16 or f16");
  SGL_CHECK(rows > 0 && hidden > 0 && hidden % 8 == 0 && hidden <= 8192 && (int64_t)rows * hidden * 2 <= max_bytes &&
                ((uintptr_t)partial % 16) == 0,
            "car_all_reduce_add_rmsnorm_quant: rows=%d hidden=%d unsupported (hidden %% 8, <= 8192, rows * hidden * 2 <= %lld)", rows,
            hidden, (long long)max_bytes);
  CarNormParams p;
  for (int r = 0; r < kMaxRanks; ++r) p.buf[r] = (char*)(r < world ? peer_bufs[r] : peer_bufs[0]);
  p.partial = partial; p.residual = residual; p.weight = weight; p.out_norm = out_norm; p.out_q = (uint8_t*)out_q; p.out_s = out_s;
  p.half_bytes = max_bytes; p.data_off = (int64_t)sizeof(CarSignal);
  p.rows = rows; p.hidden = hidden; p.eps = eps; p.rank = rank; p.world = world;
  const unsigned blocks = (unsigned)(rows > kMaxBlocks ? kMaxBlocks : rows);
  hipStream_t st = (hipStream_t)stream;
#define SGL_CARN(Tt, MV) hipLaunchKernelGGL((all_reduce_add_rmsnorm_quant_kernel<Tt, MV>), dim3(blocks), dim3(256), 0, st, p)
  if (dtype == SGL_BF16) {
    if (hidden <= 2048) SGL_CARN(__bf16, 1); else if (hidden <= 4096) SGL_CARN(__bf16, 2); else SGL_CARN(__bf16, 4);
  } else {
    if (hidden <= 2048) SGL_CARN(_Float16, 1); else if (hidden <= 4096) SGL_CARN(_Float16, 2); else SGL_CARN(_Float16, 4);
  }
#undef SGL_CARN
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}
