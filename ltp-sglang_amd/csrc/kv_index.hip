// Integer / byte kernels around the paged token_to_kv_pool.  Every result here is BIT-EXACT with
// the reference (these are index tensors and raw row copies):
//   create_kv_indices   python/sglang/srt/layers/attention/utils.py:10-45 (create_flashinfer_kv_indices_triton)
//   compute_position    python/sglang/srt/model_executor/forward_batch_info.py:885-955
//   write_req_to_token  python/sglang/srt/managers/schedule_batch.py:1920-1955 (write_req_to_token_pool_triton)
//   get_last_loc        python/sglang/srt/managers/schedule_batch.py:1958-2028
//   set_kv_buffer       python/sglang/srt/mem_cache/memory_pool.py:369-407 (two index_put launches -> one kernel)
//   num_kv_splits       python/sglang/srt/layers/attention/triton_backend.py:876-924 (get_num_kv_splits_triton)
//   kv_indptr           triton_backend.py:172 (cumsum of seq_lens into kv_indptr[1:bs+1])
// The reference walks the batch serially inside each program to get its output offset
// ("NOTE: This can be slow for large bs"); here the prefix offset is a workgroup reduction.
// Index tensors arrive as int32 or int64 exactly as the reference's callers hold them
// (SURVEY.md 8a "Index dtypes"); a width flag per array selects the load.
#include "common.h"

namespace {

__device__ __forceinline__ int64_t ld_idx(const void* p, int64_t i, int is64) {
  return is64 ? ((const int64_t*)p)[i] : (int64_t)((const int32_t*)p)[i];
}

// sum_{i < pid} lens[i], computed by the whole workgroup
__device__ __forceinline__ int64_t prefix_before(const void* lens, int is64, int pid, int64_t* red) {
  int64_t s = 0;
  for (int i = threadIdx.x; i < pid; i += blockDim.x) s += ld_idx(lens, i, is64);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, WAVE);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  int64_t t = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  return t;
}

__global__ __launch_bounds__(256) void create_kv_indices_kernel(const int32_t* req_to_token, int64_t stride,
                                                                const void* req_pool_indices, int rpi64,
                                                                const void* page_kernel_lens, int pkl64,
                                                                const int32_t* kv_indptr, const void* kv_start_idx,
                                                                int ksi64, int32_t* kv_indices) {
  const int pid = blockIdx.x;
  const int64_t req = ld_idx(req_pool_indices, pid, rpi64);
  const int64_t out0 = kv_indptr[pid];
  const int32_t kv_start = kv_start_idx ? (int32_t)ld_idx(kv_start_idx, pid, ksi64) : 0;
  const int32_t n = (int32_t)ld_idx(page_kernel_lens, pid, pkl64);
  const int32_t* src = req_to_token + req * stride + kv_start;
  for (int i = threadIdx.x; i < n; i += 256) kv_indices[out0 + i] = src[i];
}

__global__ __launch_bounds__(256) void compute_position_kernel(int64_t* positions, int32_t* extend_start_loc,
                                                               const void* prefix_lens, int pl64,
                                                               const void* seq_lens, int sl64) {
  __shared__ int64_t red[4];
  const int pid = blockIdx.x;
  const int64_t pre = prefix_lens ? ld_idx(prefix_lens, pid, pl64) : 0;
  const int64_t n = ld_idx(seq_lens, pid, sl64);
  const int64_t start = prefix_before(seq_lens, sl64, pid, red);
  for (int64_t i = threadIdx.x; i < n; i += 256) positions[start + i] = pre + i;
  if (threadIdx.x == 0) extend_start_loc[pid] = (int32_t)start;
}

__global__ __launch_bounds__(256) void write_req_to_token_kernel(int32_t* req_to_token, int64_t stride,
                                                                 const void* req_pool_indices, int rpi64,
                                                                 const void* pre_lens, int pl64, const void* seq_lens,
                                                                 int sl64, const void* extend_lens, int el64,
                                                                 const int64_t* out_cache_loc) {
  __shared__ int64_t red[4];
  const int pid = blockIdx.x;
  const int64_t req = ld_idx(req_pool_indices, pid, rpi64);
  const int64_t pre = ld_idx(pre_lens, pid, pl64);
  const int64_t n = ld_idx(seq_lens, pid, sl64) - pre;
  const int64_t start = prefix_before(extend_lens, el64, pid, red);
  int32_t* dst = req_to_token + req * stride + pre;
  for (int64_t i = threadIdx.x; i < n; i += 256) dst[i] = (int32_t)out_cache_loc[start + i];
}

__global__ __launch_bounds__(256) void get_last_loc_kernel(const int32_t* req_to_token, int64_t stride,
                                                           const void* req_pool_indices, int rpi64,
                                                           const void* prefix_lens, int pl64, void* result, int res64,
                                                           int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t pre = ld_idx(prefix_lens, i, pl64);
  const int64_t v = pre > 0 ? (int64_t)req_to_token[ld_idx(req_pool_indices, i, rpi64) * stride + pre - 1] : -1;
  if (res64)
    ((int64_t*)result)[i] = v;
  else
    ((int32_t*)result)[i] = (int32_t)v;
}

// One wave per token: copies the token's K row and V row (row_bytes each) into the pool slot loc[t].
__global__ __launch_bounds__(256) void set_kv_buffer_kernel(char* k_buf, char* v_buf, int64_t k_slot_bytes,
                                                            int64_t v_slot_bytes, const int64_t* loc, const char* cache_k,
                                                            const char* cache_v, int64_t ck_stride_bytes,
                                                            int64_t cv_stride_bytes, int k_row_bytes, int v_row_bytes,
                                                            int64_t tokens) {
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= tokens) return;
  const int lane = threadIdx.x & 63;
  const int64_t slot = loc[t];
  const char* ks = cache_k + t * ck_stride_bytes;
  const char* vs = cache_v + t * cv_stride_bytes;
  char* kd = k_buf + slot * k_slot_bytes;
  char* vd = v_buf + slot * v_slot_bytes;
  for (int o = lane * 16; o < k_row_bytes; o += 64 * 16) *(u32x4_t*)(kd + o) = *(const u32x4_t*)(ks + o);
  for (int o = lane * 16; o < v_row_bytes; o += 64 * 16) *(u32x4_t*)(vd + o) = *(const u32x4_t*)(vs + o);
}

// Single workgroup: kv_indptr[1..bs] = inclusive cumsum(seq_lens) (int32, as torch.cumsum into the
// int32 buffer does), and the reference's per-request split heuristic.
__global__ __launch_bounds__(1024) void decode_meta_kernel(int32_t* kv_indptr, int32_t* num_kv_splits, const void* seq_lens,
                                                           int sl64, int num_seq, int num_group, int num_head,
                                                           int num_kv_head, int max_kv_splits, int device_core_count,
                                                           int static_splits) {
  __shared__ int64_t scan[1024];
  __shared__ int32_t smax[16], smin[16];
  const int tid = threadIdx.x;
  // ---- cumsum (chunked Hillis-Steele over 1024 lanes) ----
  if (kv_indptr) {
    int64_t carry = 0;
    if (tid == 0) kv_indptr[0] = 0;
    for (int base = 0; base < num_seq; base += 1024) {
      const int i = base + tid;
      int64_t v = i < num_seq ? ld_idx(seq_lens, i, sl64) : 0;
      scan[tid] = v;
      __syncthreads();
      for (int off = 1; off < 1024; off <<= 1) {
        const int64_t add = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
      }
      if (i < num_seq) kv_indptr[i + 1] = (int32_t)(carry + scan[tid]);
      carry += scan[1023];
      __syncthreads();
    }
  }
  if (!num_kv_splits) return;
  if (static_splits == 2 && device_core_count > 0) {
    // MI355X balance rule: ONE round of resident workgroups (2 x 256-thread workgroups per CU).  Split length T = total key
    // rows x (kv heads x head chunks) / resident, rounded up to whole 32-token tiles; each request gets ceil(len / T) splits.
    // (The kernel alone streams equally fast with one or two rounds; with the stage-2 merge inside the launch every extra
    // split is partial-result traffic and merge work at the tail, and an A/B of whole decode steps on one box favours one
    // round at every shape tried: 8B fp8 bs 32 / 48 / 64 +1.5 / +2.4 / +0.5 %, fp8 KV +2.9 %, Qwen2-7B AWQ +3.7 %, bs 128 equal.)
    int64_t tot = 0;
    for (int i = tid; i < num_seq; i += 1024) tot += ld_idx(seq_lens, i, sl64);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) tot += __shfl_xor(tot, m, WAVE);
    __syncthreads();
    if ((tid & 63) == 0) scan[tid >> 6] = tot;
    __syncthreads();
    tot = 0;
    for (int w = 0; w < 16; ++w) tot += scan[w];
    const int kv_group = num_head / num_kv_head;
    const int64_t units = (int64_t)num_kv_head * ((kv_group + 15) / 16) * num_group;
    // (round 4) at most one unit per two CUs: the kernel runs ONE 512-thread workgroup per CU (csrc/decode_attention.hip launch_mfma
    // makes the same test), so half as many splits fill the chip
    const int64_t target_wgs = (2 * units * num_seq <= device_core_count ? 1ll : 2ll) * device_core_count;
    int64_t T = (tot * units + target_wgs - 1) / target_wgs;
    T = (T + 31) / 32 * 32;
    if (T < 64) T = 64;
    // (round 5, measured and NOT kept: lengthening T for ragged batches until sum_i ceil(len_i / T) units fit one round -- BASELINE
    // config 5's shard, batch 128 x U(512, 4096), one kv head: 309 -> 255 units, attention launch 44.5 -> 46.4 us: the launch time
    // follows the longest unit, which grew with T; profiles/round5_configs.json)
    for (int i = tid; i < num_seq; i += 1024) {
      const int64_t len = ld_idx(seq_lens, i, sl64);
      int ns = (int)((len + T - 1) / T);
      ns = ns < 1 ? 1 : (ns > max_kv_splits ? max_kv_splits : ns);
      for (int gI = 0; gI < num_group; ++gI) num_kv_splits[i * num_group + gI] = ns;
    }
    return;
  }
  if (static_splits == 1 || device_core_count <= 0) {
    for (int i = tid; i < num_seq * num_group; i += 1024) num_kv_splits[i] = max_kv_splits;
    return;
  }
  // ---- max / min of seq_lens ----
  int32_t mx = 0;
  for (int i = tid; i < num_seq; i += 1024) mx = max(mx, (int32_t)ld_idx(seq_lens, i, sl64));
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) mx = max(mx, __shfl_xor(mx, m, WAVE));
  if ((tid & 63) == 0) smax[tid >> 6] = mx;
  __syncthreads();
  mx = 0;
  for (int w = 0; w < 16; ++w) mx = max(mx, smax[w]);
  int32_t mn = mx;
  for (int i = tid; i < num_seq; i += 1024) mn = min(mn, (int32_t)ld_idx(seq_lens, i, sl64));
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) mn = min(mn, __shfl_xor(mn, m, WAVE));
  if ((tid & 63) == 0) smin[tid >> 6] = mn;
  __syncthreads();
  mn = mx;
  for (int w = 0; w < 16; ++w) mn = min(mn, smin[w]);
  if ((int64_t)mx * 8 < (int64_t)mn * 10) mn = mx;
  auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
  if (mx <= 0 || mn <= 0) {  // degenerate batch (the reference would divide by zero)
    for (int i = tid; i < num_seq * num_group; i += 1024) num_kv_splits[i] = 1;
    return;
  }
  const int splits1 = min(cdiv(mx, mn), max_kv_splits);
  const int chunk1 = cdiv(mx, splits1);
  const float ext_seq = (float)mx / 64.0f;
  const int ext_cores = (int)((float)device_core_count * fmaxf(log2f(ext_seq), 1.0f));
  int block_h = 16;
  const int kv_group = num_head / num_kv_head;
  int token_grid;
  if (kv_group == 1) {
    token_grid = num_seq * num_group * num_head;
  } else {
    block_h = min(block_h, kv_group);
    token_grid = num_seq * num_group * cdiv(num_head, block_h);
  }
  const int splits2 = max(1, min(cdiv(ext_cores, token_grid), max_kv_splits));
  const int chunk2 = cdiv(mx, splits2);
  for (int i = tid; i < num_seq; i += 1024) {
    const int len = (int32_t)ld_idx(seq_lens, i, sl64);
    const int ns = max(cdiv(len, chunk1), cdiv(len, chunk2));
    for (int gI = 0; gI < num_group; ++gI) num_kv_splits[i * num_group + gI] = ns;
  }
}

}  // namespace

extern "C" int sgl_mi355_create_kv_indices(const int32_t* req_to_token, int64_t req_to_token_stride,
                                           const void* req_pool_indices, int req_pool_indices_is64,
                                           const void* page_kernel_lens, int page_kernel_lens_is64,
                                           const int32_t* kv_indptr, const void* kv_start_idx, int kv_start_idx_is64,
                                           int32_t* kv_indices, int batch, void* stream) {
  SGL_CHECK(batch >= 0, "create_kv_indices: negative batch");
  if (batch == 0) return SGL_MI355_OK;
  SGL_CHECK(req_to_token && req_pool_indices && page_kernel_lens && kv_indptr, "create_kv_indices: null pointer");
  hipLaunchKernelGGL(create_kv_indices_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, req_to_token,
                     req_to_token_stride, req_pool_indices, req_pool_indices_is64, page_kernel_lens, page_kernel_lens_is64,
                     kv_indptr, kv_start_idx, kv_start_idx_is64, kv_indices);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_compute_position(int64_t* positions, int32_t* extend_start_loc, const void* extend_prefix_lens,
                                          int prefix_is64, const void* extend_seq_lens, int seq_is64, int batch,
                                          void* stream) {
  SGL_CHECK(batch >= 0, "compute_position: negative batch");
  if (batch == 0) return SGL_MI355_OK;
  SGL_CHECK(positions && extend_start_loc && extend_seq_lens, "compute_position: null pointer");
  hipLaunchKernelGGL(compute_position_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, positions, extend_start_loc,
                     extend_prefix_lens, prefix_is64, extend_seq_lens, seq_is64);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_write_req_to_token(int32_t* req_to_token, int64_t req_to_token_stride,
                                            const void* req_pool_indices, int req_pool_indices_is64, const void* pre_lens,
                                            int pre_is64, const void* seq_lens, int seq_is64, const void* extend_lens,
                                            int ext_is64, const int64_t* out_cache_loc, int batch, void* stream) {
  SGL_CHECK(batch >= 0, "write_req_to_token: negative batch");
  if (batch == 0) return SGL_MI355_OK;
  SGL_CHECK(req_to_token && req_pool_indices && pre_lens && seq_lens && extend_lens && out_cache_loc,
            "write_req_to_token: null pointer");
  hipLaunchKernelGGL(write_req_to_token_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, req_to_token,
                     req_to_token_stride, req_pool_indices, req_pool_indices_is64, pre_lens, pre_is64, seq_lens, seq_is64,
                     extend_lens, ext_is64, out_cache_loc);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// One launch for the per-step host work of a graph-replayed decode batch: prepare_for_decode (schedule_batch.py:1560-1590:
// req_to_token[req, seq_len] = new slot, seq_lens += 1) followed by replay_prepare's copies into the graph's static buffers
// (cuda_graph_runner.py:700-760: input_ids, req_pool_indices, seq_lens, out_cache_loc, positions = seq_lens - 1).
namespace {
__global__ __launch_bounds__(256) void decode_prepare_kernel(const int64_t* req_pool_indices, int64_t* seq_lens,
                                                             const int64_t* out_cache_loc, const int64_t* next_ids,
                                                             int32_t* req_to_token, int64_t r2t_stride, int64_t* b_input_ids,
                                                             int64_t* b_req_pool_indices, int64_t* b_seq_lens,
                                                             int64_t* b_out_cache_loc, int64_t* b_positions, int bs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= bs) return;
  const int64_t s = seq_lens[i], req = req_pool_indices[i], loc = out_cache_loc[i];
  req_to_token[req * r2t_stride + s] = (int32_t)loc;
  seq_lens[i] = s + 1;
  b_input_ids[i] = next_ids[i];
  b_req_pool_indices[i] = req;
  b_seq_lens[i] = s + 1;
  b_out_cache_loc[i] = loc;
  b_positions[i] = s > 0 ? s : 0;  // clamp(seq_lens_new - 1, 0), forward_batch_info.py:958-960
}
}  // namespace

extern "C" int sgl_mi355_decode_prepare(const int64_t* req_pool_indices, int64_t* seq_lens, const int64_t* out_cache_loc,
                                        const int64_t* next_ids, int32_t* req_to_token, int64_t req_to_token_stride,
                                        int64_t* buf_input_ids, int64_t* buf_req_pool_indices, int64_t* buf_seq_lens,
                                        int64_t* buf_out_cache_loc, int64_t* buf_positions, int batch, void* stream) {
  SGL_CHECK(batch >= 0, "decode_prepare: negative batch");
  if (batch == 0) return SGL_MI355_OK;
  SGL_CHECK(req_pool_indices && seq_lens && out_cache_loc && next_ids && req_to_token && buf_input_ids && buf_req_pool_indices &&
                buf_seq_lens && buf_out_cache_loc && buf_positions,
            "decode_prepare: null pointer");
  hipLaunchKernelGGL(decode_prepare_kernel, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, req_pool_indices, seq_lens,
                     out_cache_loc, next_ids, req_to_token, req_to_token_stride, buf_input_ids, buf_req_pool_indices, buf_seq_lens,
                     buf_out_cache_loc, buf_positions, batch);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_get_last_loc(const int32_t* req_to_token, int64_t req_to_token_stride,
                                      const void* req_pool_indices, int req_pool_indices_is64, const void* prefix_lens,
                                      int prefix_is64, void* result, int result_is64, int64_t n, void* stream) {
  SGL_CHECK(n >= 0, "get_last_loc: negative size");
  if (n == 0) return SGL_MI355_OK;
  SGL_CHECK(req_to_token && req_pool_indices && prefix_lens && result, "get_last_loc: null pointer");
  hipLaunchKernelGGL(get_last_loc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, req_to_token,
                     req_to_token_stride, req_pool_indices, req_pool_indices_is64, prefix_lens, prefix_is64, result,
                     result_is64, n);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// ---- fp8 (e4m3fn) KV cache: set_kv_buffer with kv_cache_dtype = fp8_e4m3 (memory_pool.py:385-395) ----
// (conversion rules: common.h "fp8 (e4m3fn) KV cache conversion")
namespace {
template <typename T>
__device__ __forceinline__ uint32_t cvt4_to_fp8(const T* x, float inv_or_zero, float scale) {
  float f[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = kv_fp8_scaled<T>((float)x[i], scale);
  (void)inv_or_zero;
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
  uint32_t u = (uint32_t)w;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (kv_fp8_is_nan(f[i])) u = (u & ~(0xFFu << (8 * i))) | (kv_fp8_nan_byte(f[i]) << (8 * i));
  }
  return u;
}

template <typename T>
__global__ __launch_bounds__(256) void set_kv_buffer_fp8_kernel(uint8_t* k_buffer, uint8_t* v_buffer, int64_t k_slot_bytes,
                                                                int64_t v_slot_bytes, const int64_t* loc, const T* cache_k,
                                                                const T* cache_v, int64_t ck_stride, int64_t cv_stride,
                                                                int k_row, int v_row, int64_t tokens, float k_scale,
                                                                float v_scale) {
  const int kch = k_row / 8, vch = v_row / 8;  // 8-element chunks per row
  const int64_t total = tokens * (kch + vch);
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t t = idx / (kch + vch);
    const int c = (int)(idx - t * (kch + vch));
    const int64_t slot = loc[t];
    const bool is_k = c < kch;
    const int cc = is_k ? c : c - kch;
    const T* src = (is_k ? cache_k + t * ck_stride : cache_v + t * cv_stride) + 8 * cc;
    const u32x4_t raw = *(const u32x4_t*)src;
    struct P8 { T v[8]; };
    const P8 x = __builtin_bit_cast(P8, raw);
    const float sc = is_k ? k_scale : v_scale;
    u32x2_t out;
    out[0] = cvt4_to_fp8<T>(x.v, 0.f, sc);
    out[1] = cvt4_to_fp8<T>(x.v + 4, 0.f, sc);
    uint8_t* dst = (is_k ? k_buffer + slot * k_slot_bytes : v_buffer + slot * v_slot_bytes) + 8 * cc;
    *(u32x2_t*)dst = out;
  }
}
}  // namespace

// strides / row sizes in ELEMENTS of the source dtype (bf16 / f16); pool slot strides in bytes; k_scale / v_scale <= 0: none
extern "C" int sgl_mi355_set_kv_buffer_fp8(void* k_buffer, void* v_buffer, int64_t k_slot_bytes, int64_t v_slot_bytes,
                                           const int64_t* loc, const void* cache_k, const void* cache_v,
                                           int64_t cache_k_stride, int64_t cache_v_stride, int k_row, int v_row,
                                           int64_t tokens, int src_dtype, float k_scale, float v_scale, void* stream) {
  SGL_CHECK(tokens >= 0, "set_kv_buffer_fp8: negative token count");
  if (tokens == 0) return SGL_MI355_OK;
  SGL_CHECK(k_buffer && v_buffer && loc && cache_k && cache_v, "set_kv_buffer_fp8: null pointer");
  SGL_CHECK(src_dtype == SGL_BF16 || src_dtype == SGL_F16, "set_kv_buffer_fp8: source dtype must be bf16 or f16");
  SGL_CHECK(k_row % 8 == 0 && v_row % 8 == 0 && k_slot_bytes % 8 == 0 && v_slot_bytes % 8 == 0 && cache_k_stride % 8 == 0 &&
                cache_v_stride % 8 == 0 && ((uintptr_t)k_buffer % 8) == 0 && ((uintptr_t)v_buffer % 8) == 0 &&
                ((uintptr_t)cache_k % 16) == 0 && ((uintptr_t)cache_v % 16) == 0,
            "set_kv_buffer_fp8: rows must be multiples of 8 elements and 16-byte aligned at the source");
  const int64_t total = tokens * (k_row / 8 + v_row / 8);
  const unsigned blocks = (unsigned)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == SGL_BF16)
    hipLaunchKernelGGL((set_kv_buffer_fp8_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, (uint8_t*)k_buffer, (uint8_t*)v_buffer,
                       k_slot_bytes, v_slot_bytes, loc, (const __bf16*)cache_k, (const __bf16*)cache_v, cache_k_stride,
                       cache_v_stride, k_row, v_row, tokens, k_scale, v_scale);
  else
    hipLaunchKernelGGL((set_kv_buffer_fp8_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, (uint8_t*)k_buffer,
                       (uint8_t*)v_buffer, k_slot_bytes, v_slot_bytes, loc, (const _Float16*)cache_k, (const _Float16*)cache_v,
                       cache_k_stride, cache_v_stride, k_row, v_row, tokens, k_scale, v_scale);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// k_buffer/v_buffer: pool bases for one layer; slot strides in bytes.  cache_k/cache_v: [tokens, row] with byte strides.
extern "C" int sgl_mi355_set_kv_buffer(void* k_buffer, void* v_buffer, int64_t k_slot_bytes, int64_t v_slot_bytes,
                                       const int64_t* loc, const void* cache_k, const void* cache_v,
                                       int64_t cache_k_stride_bytes, int64_t cache_v_stride_bytes, int k_row_bytes,
                                       int v_row_bytes, int64_t tokens, void* stream) {
  SGL_CHECK(tokens >= 0, "set_kv_buffer: negative token count");
  if (tokens == 0) return SGL_MI355_OK;
  SGL_CHECK(k_buffer && v_buffer && loc && cache_k && cache_v, "set_kv_buffer: null pointer");
  SGL_CHECK(k_row_bytes % 16 == 0 && v_row_bytes % 16 == 0 && k_slot_bytes % 16 == 0 && v_slot_bytes % 16 == 0 &&
                cache_k_stride_bytes % 16 == 0 && cache_v_stride_bytes % 16 == 0 && ((uintptr_t)k_buffer % 16) == 0 &&
                ((uintptr_t)v_buffer % 16) == 0 && ((uintptr_t)cache_k % 16) == 0 && ((uintptr_t)cache_v % 16) == 0,
            "set_kv_buffer: rows must be 16-byte aligned (k_row_bytes=%d, v_row_bytes=%d)", k_row_bytes, v_row_bytes);
  hipLaunchKernelGGL(set_kv_buffer_kernel, dim3((unsigned)((tokens + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (char*)k_buffer, (char*)v_buffer, k_slot_bytes, v_slot_bytes, loc, (const char*)cache_k,
                     (const char*)cache_v, cache_k_stride_bytes, cache_v_stride_bytes, k_row_bytes, v_row_bytes, tokens);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// kv_indptr and/or num_kv_splits may be NULL to skip that part.  static_splits: 0 = the reference's heuristic
// (get_num_kv_splits_triton), 1 = max_kv_splits for every request (SGLANG_TRITON_DECODE_ATTN_STATIC_KV_SPLITS),
// 2 = the MI355X balance rule (any split count gives the same attention up to fp association: SURVEY.md a12).
extern "C" int sgl_mi355_decode_metadata(int32_t* kv_indptr, int32_t* num_kv_splits, const void* seq_lens, int seq_is64,
                                         int num_seq, int num_group, int num_head, int num_kv_head, int max_kv_splits,
                                         int device_core_count, int static_splits, void* stream) {
  SGL_CHECK(num_seq >= 0, "decode_metadata: negative batch");
  if (num_seq == 0) return SGL_MI355_OK;
  SGL_CHECK(seq_lens, "decode_metadata: null seq_lens");
  SGL_CHECK(num_group >= 1 && num_kv_head >= 1 && num_head >= num_kv_head && max_kv_splits >= 1, "decode_metadata: bad head/split arguments");
  hipLaunchKernelGGL(decode_meta_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, kv_indptr, num_kv_splits, seq_lens,
                     seq_is64, num_seq, num_group, num_head, num_kv_head, max_kv_splits, device_core_count, static_splits);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

namespace {

// Sorted unit list of a decode batch (round 5; consumer: decode_attention.hip, SCHED kernels).  The reference launches a grid of
// (batch, heads, max_kv_splits) programs of which each request uses its first num_kv_splits[b] (decode_attention.py:677-728): over
// ragged lengths the longest split decides when the launch ends, and most of the grid exits at once.  Here the same (request, split)
// units -- split boundaries exactly as the reference's kernel computes them from (length, num_kv_splits[b]) -- become a LIST, sorted by
// length, longest first: the hardware dispatcher hands out workgroups in list order, so long units start first and short ones fill the
// slots that free up (longest-processing-time-first), and the grid is as long as the list.
//   T = 32-token multiple of ceil(total / target), at least 64;  num_kv_splits[b] = clamp(ceil(len / T), 1, max_kv_splits);
//   sched = {T, units, total, capacity} then per unit {request, split | splits << 16, kv_indptr[request], its length}.
// One workgroup; kv_indptr is written by the same launch (cumsum as decode_meta_kernel).  capacity <= 4096 (the sort runs in LDS).
constexpr int kSchedMaxUnits = 4096;
__global__ __launch_bounds__(1024) void decode_schedule_kernel(int32_t* kv_indptr, int32_t* num_kv_splits, int32_t* sched,
                                                               const void* seq_lens, int sl64, int num_seq, int max_kv_splits,
                                                               int target_even, int target_ragged, int capacity) {
  __shared__ int64_t scan[1024];
  __shared__ uint64_t keys[kSchedMaxUnits];   // length << 32 | (0xffff - split) << 16 | (0xffff - request): sorted descending
  __shared__ int32_t smax[16], smin[16];
  const int tid = threadIdx.x;
  int64_t carry = 0;
  int32_t mx = 0, mn = 0x7fffffff;
  if (tid == 0) kv_indptr[0] = 0;
  for (int base = 0; base < num_seq; base += 1024) {
    const int i = base + tid;
    const int64_t v = i < num_seq ? ld_idx(seq_lens, i, sl64) : 0;
    if (i < num_seq) { mx = max(mx, (int32_t)v); mn = min(mn, (int32_t)v); }
    scan[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int64_t add = tid >= off ? scan[tid - off] : 0;
      __syncthreads();
      scan[tid] += add;
      __syncthreads();
    }
    if (i < num_seq) kv_indptr[i + 1] = (int32_t)(carry + scan[tid]);
    carry += scan[1023];
    __syncthreads();
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    mx = max(mx, __shfl_xor(mx, m, WAVE));
    mn = min(mn, __shfl_xor(mn, m, WAVE));
  }
  if ((tid & 63) == 0) { smax[tid >> 6] = mx; smin[tid >> 6] = mn; }
  __syncthreads();
  for (int w = 0; w < 16; ++w) { mx = max(mx, smax[w]); mn = min(mn, smin[w]); }
  // lengths within 25 % of each other (the reference's own test, triton_backend.py:895-897): one round of equal units, the rule the
  // whole-step A/Bs of rounds 2-4 chose; ragged: smaller units (target_ragged of them), which the sorted dispatch packs
  const bool even = (int64_t)mx * 8 < (int64_t)mn * 10;
  const int target = even ? target_even : target_ragged;
  const int64_t tot = carry;
  int64_t T = (tot + target - 1) / target;
  T = (T + 31) / 32 * 32;
  if (T < 64) T = 64;
  // splits per request and their exclusive prefix sum = each request's first slot of the unsorted list
  int64_t ucarry = 0;
  for (int base = 0; base < num_seq; base += 1024) {
    const int i = base + tid;
    int ns = 0, len = 0;
    if (i < num_seq) {
      len = (int)ld_idx(seq_lens, i, sl64);
      ns = (int)((len + T - 1) / T);
      ns = ns < 1 ? 1 : (ns > max_kv_splits ? max_kv_splits : ns);
      num_kv_splits[i] = ns;
    }
    scan[tid] = ns;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int64_t add = tid >= off ? scan[tid - off] : 0;
      __syncthreads();
      scan[tid] += add;
      __syncthreads();
    }
    if (i < num_seq) {
      const int u0 = (int)(ucarry + scan[tid]) - ns;
      const int per0 = (len + ns - 1) / ns;
      const int per = (per0 + 31) / 32 * 32;   // (decode_attention.py:90-94, split_len in decode_attention.hip)
      for (int j = 0; j < ns; ++j) {
        if (u0 + j < kSchedMaxUnits) {
          int l = len - j * per;
          l = l < 0 ? 0 : (l > per ? per : l);
          keys[u0 + j] = ((uint64_t)(uint32_t)l << 32) | ((uint64_t)(0xffffu - (uint32_t)j) << 16) | (uint64_t)(0xffffu - (uint32_t)i);
        }
      }
    }
    ucarry += scan[1023];
    __syncthreads();
  }
  int units = (int)ucarry;   // <= target + num_seq <= capacity by the choice of T (every request's last split is a remainder)
  units = units > capacity ? capacity : units;
  int n2 = 1;
  while (n2 < units) n2 <<= 1;
  for (int u = units + tid; u < n2; u += 1024) keys[u] = 0;
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1) {       // bitonic sort, descending: longest unit first; equal lengths in the grid's own order
    for (int j = k >> 1; j > 0; j >>= 1) {  // (split index slowest, request fastest)
      for (int x = tid; x < n2; x += 1024) {
        const int y = x ^ j;
        if (y > x) {
          const uint64_t a = keys[x], b2 = keys[y];
          const bool desc = (x & k) == 0;
          if (desc ? a < b2 : a > b2) { keys[x] = b2; keys[y] = a; }
        }
      }
      __syncthreads();
    }
  }
  if (tid == 0) {
    sched[0] = (int32_t)T; sched[1] = units; sched[2] = (int32_t)tot; sched[3] = capacity;
  }
  for (int x = tid; x < units; x += 1024) {
    const uint64_t key = keys[x];
    const int b = (int)(0xffffu - (uint32_t)(key & 0xffffu)), j = (int)(0xffffu - (uint32_t)((key >> 16) & 0xffffu));
    const int s0 = kv_indptr[b];
    sched[4 + 4 * x] = b;
    sched[5 + 4 * x] = j | (num_kv_splits[b] << 16);
    sched[6 + 4 * x] = s0;
    sched[7 + 4 * x] = kv_indptr[b + 1] - s0;
  }
}

}  // namespace

int sgl_mi355_internal_decode_sched_target(int batch, int num_q_heads, int num_kv_heads, int rounds_pct);   // decode_attention.hip

// Capacity (units, = the attention launch's grid y) of the list for a batch of this geometry: resident workgroups x rounds_pct / 100
// per (kv head x head chunk) -- the ragged target; a batch of near-equal lengths is cut for ONE round -- + one remainder unit per
// request.  0: not available (more than 4096 units, or batch > 65535).
extern "C" int sgl_mi355_decode_schedule_units(int num_seq, int num_head, int num_kv_head, int rounds_pct) {
  if (num_seq <= 0 || num_seq > 65535 || num_kv_head <= 0 || num_head < num_kv_head || num_head % num_kv_head || rounds_pct < 25 || rounds_pct > 1600)
    return 0;
  const int pct = rounds_pct < 100 ? 100 : rounds_pct;   // (the even target is 100)
  const long long cap = (long long)sgl_mi355_internal_decode_sched_target(num_seq, num_head, num_kv_head, pct) + num_seq;
  return cap > kSchedMaxUnits ? 0 : (int)cap;
}

// kv_indptr, num_kv_splits and the sorted unit list of a decode batch in one launch (the kernel above); the consumer is
// sgl_mi355_decode_attention_scheduled.  sched: int32, 16-byte aligned, 4 + 4 x sgl_mi355_decode_schedule_units(...) words.
extern "C" int sgl_mi355_decode_schedule(int32_t* kv_indptr, int32_t* num_kv_splits, int32_t* sched, int sched_units,
                                         const void* seq_lens, int seq_is64, int num_seq, int num_head, int num_kv_head,
                                         int max_kv_splits, int rounds_pct, void* stream) {
  SGL_CHECK(num_seq >= 0, "decode_schedule: negative batch");
  if (num_seq == 0) return SGL_MI355_OK;
  SGL_CHECK(kv_indptr && num_kv_splits && sched && seq_lens, "decode_schedule: null pointer");
  SGL_CHECK(max_kv_splits >= 1 && max_kv_splits <= 32767, "decode_schedule: max_kv_splits=%d outside [1, 32767]", max_kv_splits);
  const int cap = sgl_mi355_decode_schedule_units(num_seq, num_head, num_kv_head, rounds_pct);
  SGL_CHECK(cap > 0, "decode_schedule: no unit list for this geometry (batch %d, heads %d / %d, rounds_pct %d): more than %d units or bad arguments",
            num_seq, num_head, num_kv_head, rounds_pct, kSchedMaxUnits);
  SGL_CHECK(sched_units == cap && ((uintptr_t)sched % 16) == 0, "decode_schedule: the unit list needs %d units (got %d), 16-byte aligned", cap, sched_units);
  hipLaunchKernelGGL(decode_schedule_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, kv_indptr, num_kv_splits, sched, seq_lens,
                     seq_is64, num_seq, max_kv_splits, sgl_mi355_internal_decode_sched_target(num_seq, num_head, num_kv_head, 100),
                     sgl_mi355_internal_decode_sched_target(num_seq, num_head, num_kv_head, rounds_pct), cap);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Paged allocator index kernels (page_size > 1): bit-exact with alloc_extend_kernel / alloc_decode_kernel of
// python/sglang/srt/mem_cache/allocator.py:275-395.  A request first fills the tail of its last partial page
// (last_loc + 1 ...), then takes whole new pages from the free list, then a new partial page; the per-request offsets
// into out_indices / free_pages are prefix sums over the batch (a workgroup reduction here, a serial walk there).
// ret_values[0] = (sum of new pages << 32) | sum of extend tokens   (extend)   /   sum of new pages   (decode)
// ---------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ int64_t cdivp(int64_t a, int64_t b) { return (a + b - 1) / b; }

__global__ __launch_bounds__(256) void alloc_extend_kernel(const void* pre_lens, int pl64, const void* seq_lens, int sl64,
                                                           const void* last_loc, int ll64, const int64_t* free_pages,
                                                           int64_t* out_indices, int64_t* ret_values, int page_size, int bs) {
  __shared__ int64_t red[4];
  const int pid = blockIdx.x;
  // prefix sums over requests before pid (extend tokens and new pages) and, for the last request, the batch totals
  int64_t ext_before = 0, pages_before = 0;
  for (int i = threadIdx.x; i < pid; i += 256) {
    const int64_t s = ld_idx(seq_lens, i, sl64), p = ld_idx(pre_lens, i, pl64);
    ext_before += s - p;
    pages_before += cdivp(s, page_size) - cdivp(p, page_size);
  }
  for (int m = 32; m >= 1; m >>= 1) {
    ext_before += __shfl_xor(ext_before, m, WAVE);
    pages_before += __shfl_xor(pages_before, m, WAVE);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ext_before;
  __syncthreads();
  ext_before = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pages_before;
  __syncthreads();
  pages_before = red[0] + red[1] + red[2] + red[3];

  const int64_t seq = ld_idx(seq_lens, pid, sl64), pre = ld_idx(pre_lens, pid, pl64);
  const int64_t ext = seq - pre;
  const int64_t pages_self = cdivp(seq, page_size) - cdivp(pre, page_size);
  if (pid == bs - 1 && threadIdx.x == 0) ret_values[0] = ((pages_before + pages_self) << 32) | (ext_before + ext);

  int64_t* out = out_indices + ext_before;
  const int64_t last = ld_idx(last_loc, pid, ll64);
  const int64_t pre_up = cdivp(pre, page_size) * page_size;
  const int64_t part1 = (seq < pre_up ? seq : pre_up) - pre;  // tail of the old partial page
  for (int64_t k = threadIdx.x; k < part1; k += 256) out[k] = last + 1 + k;
  if (pre + part1 == seq) return;
  const int64_t part2 = seq / page_size * page_size - pre_up;  // whole new pages
  for (int64_t k = threadIdx.x; k < part2; k += 256)
    out[part1 + k] = free_pages[pages_before + k / page_size] * page_size + k % page_size;
  if (pre + part1 + part2 == seq) return;
  const int64_t part3 = seq - seq / page_size * page_size;     // new partial page
  const int64_t start_page = free_pages[pages_before + pages_self - 1];
  for (int64_t k = threadIdx.x; k < part3; k += 256) out[part1 + part2 + k] = start_page * page_size + k;
}

__global__ __launch_bounds__(256) void alloc_decode_kernel(const void* seq_lens, int sl64, const void* last_loc, int ll64,
                                                           const int64_t* free_pages, int64_t* out_indices,
                                                           int64_t* ret_values, int page_size, int bs) {
  __shared__ int64_t red[4];
  const int pid = blockIdx.x;
  int64_t pages_before = 0;
  for (int i = threadIdx.x; i < pid; i += 256) {
    const int64_t s = ld_idx(seq_lens, i, sl64);
    pages_before += cdivp(s, page_size) - cdivp(s - 1, page_size);
  }
  for (int m = 32; m >= 1; m >>= 1) pages_before += __shfl_xor(pages_before, m, WAVE);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pages_before;
  __syncthreads();
  pages_before = red[0] + red[1] + red[2] + red[3];
  if (threadIdx.x != 0) return;
  const int64_t seq = ld_idx(seq_lens, pid, sl64);
  const int64_t pages_self = cdivp(seq, page_size) - cdivp(seq - 1, page_size);
  if (pid == bs - 1) ret_values[0] = pages_before + pages_self;
  out_indices[pid] = pages_self == 0 ? ld_idx(last_loc, pid, ll64) + 1 : free_pages[pages_before] * page_size;
}

}  // namespace

extern "C" int sgl_mi355_alloc_extend(const void* prefix_lens, int prefix_is64, const void* seq_lens, int seq_is64,
                                      const void* last_loc, int last_loc_is64, const int64_t* free_pages,
                                      int64_t* out_indices, int64_t* ret_values, int page_size, int batch, void* stream) {
  SGL_CHECK(batch >= 0 && page_size >= 1, "alloc_extend: bad batch/page_size");
  if (batch == 0) return SGL_MI355_OK;
  SGL_CHECK(prefix_lens && seq_lens && last_loc && free_pages && out_indices && ret_values, "alloc_extend: null pointer");
  hipLaunchKernelGGL(alloc_extend_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, prefix_lens, prefix_is64, seq_lens,
                     seq_is64, last_loc, last_loc_is64, free_pages, out_indices, ret_values, page_size, batch);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

extern "C" int sgl_mi355_alloc_decode(const void* seq_lens, int seq_is64, const void* last_loc, int last_loc_is64,
                                      const int64_t* free_pages, int64_t* out_indices, int64_t* ret_values, int page_size,
                                      int batch, void* stream) {
  SGL_CHECK(batch >= 0 && page_size >= 1, "alloc_decode: bad batch/page_size");
  if (batch == 0) return SGL_MI355_OK;
  SGL_CHECK(seq_lens && last_loc && free_pages && out_indices && ret_values, "alloc_decode: null pointer");
  hipLaunchKernelGGL(alloc_decode_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, seq_lens, seq_is64, last_loc,
                     last_loc_is64, free_pages, out_indices, ret_values, page_size, batch);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

// ---- move_kv_cache: MHATokenToKVPool.move_kv_cache -> copy_all_layer_kv_cache (memory_pool.py:409-417, 1046-1081) ----
// buf[tgt_loc[i]] = buf[src_loc[i]] for every K and V buffer of every layer, IN PLACE: as in the reference kernel all source
// rows of a column block are read before any target row of that block is written (a location may be both a source and a
// target).  One workgroup per (buffer, column block): the block's n x cw bytes are gathered into LDS, a barrier, then
// scattered.  cw (a multiple of 16 bytes) is the widest block that keeps n x cw within 64 KiB, so a handful of moved
// tokens copy whole rows with coalesced 16-byte accesses.
namespace {
constexpr int kMoveLds = 64 * 1024;

__global__ __launch_bounds__(256) void move_kv_cache_kernel(const uint64_t* __restrict__ data_ptrs, const int64_t* __restrict__ strides,
                                                            const void* tgt_loc, int tgt64, const void* src_loc, int src64, int n,
                                                            int cw) {
  extern __shared__ __attribute__((aligned(16))) char stage[];
  const int64_t stride = strides[blockIdx.x];
  const int64_t c0 = (int64_t)blockIdx.y * cw;
  if (c0 >= stride) return;  // (buffers with a shorter row than the widest one)
  char* base = (char*)data_ptrs[blockIdx.x];
  const int pieces = (int)(min((int64_t)cw, stride - c0) / 16);  // 16-byte pieces of this block per row
  const int total = n * pieces;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int i = idx / pieces, pc = idx - i * pieces;
    *(u32x4_t*)(stage + (int64_t)idx * 16) = *(const u32x4_t*)(base + ld_idx(src_loc, i, src64) * stride + c0 + pc * 16);
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int i = idx / pieces, pc = idx - i * pieces;
    *(u32x4_t*)(base + ld_idx(tgt_loc, i, tgt64) * stride + c0 + pc * 16) = *(const u32x4_t*)(stage + (int64_t)idx * 16);
  }
}
}  // namespace

// data_ptrs uint64 [num_buffers] (device) = the buffers' base addresses, data_strides int64 [num_buffers] (device) = bytes per
// row of each (multiples of 16), max_stride_bytes = the largest of them: MHATokenToKVPool.data_ptrs / data_strides
// (memory_pool.py:241-256).  tgt_loc / src_loc: int32 or int64 [num_locs] (device), num_locs <= 4096 per call.
extern "C" int sgl_mi355_move_kv_cache(const void* data_ptrs, const int64_t* data_strides, int num_buffers, int64_t max_stride_bytes,
                                       const void* tgt_loc, int tgt_is64, const void* src_loc, int src_is64, int num_locs,
                                       void* stream) {
  SGL_CHECK(num_buffers >= 0 && num_locs >= 0, "move_kv_cache: negative count");
  if (num_buffers == 0 || num_locs == 0) return SGL_MI355_OK;
  SGL_CHECK(data_ptrs && data_strides && tgt_loc && src_loc, "move_kv_cache: null pointer");
  SGL_CHECK(num_locs <= kMoveLds / 16, "move_kv_cache: at most %d locations per call (got %d)", kMoveLds / 16, num_locs);
  SGL_CHECK(max_stride_bytes > 0 && max_stride_bytes % 16 == 0, "move_kv_cache: rows must be multiples of 16 bytes (got %lld)",
            (long long)max_stride_bytes);
  int64_t cw = (kMoveLds / num_locs) / 16 * 16;
  if (cw > max_stride_bytes) cw = max_stride_bytes;
  const int64_t nblk = (max_stride_bytes + cw - 1) / cw;
  SGL_CHECK(nblk <= 65535, "move_kv_cache: row too long");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)move_kv_cache_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMoveLds);
    attr_set = true;
  }
  hipLaunchKernelGGL(move_kv_cache_kernel, dim3(num_buffers, (unsigned)nblk), dim3(256), (size_t)num_locs * cw, (hipStream_t)stream,
                     (const uint64_t*)data_ptrs, data_strides, tgt_loc, tgt_is64, src_loc, src_is64, num_locs, (int)cw);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}
