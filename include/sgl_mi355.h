/*
 * sgl_mi355.h -- C ABI of the MI355X (gfx950) hot path: attention over the paged token_to_kv_pool,
 * fp8 / int4-AWQ dequant GEMM and their integer/elementwise neighbours.
 *
 * Conventions:
 *  - every tensor argument is a raw DEVICE pointer plus sizes/strides (strides in ELEMENTS unless the
 *    name says bytes); no torch types cross this boundary;
 *  - every launcher is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *    performs no allocation and no host synchronisation, so it can be captured into a hipGraph
 *    (reference requirement: cuda_graph_runner.py:280,618,760);
 *  - return value: 0 = ok, 1 = invalid argument, 2 = HIP launch error; the text is available from
 *    sgl_mi355_last_error() (thread local).  The reference signals the same conditions with
 *    TORCH_CHECK -> RuntimeError (sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1078-1108); the Python shim
 *    (ltp-sglang_amd/_cabi.py) converts a non-zero status into RuntimeError(message).
 *  - dtype codes: 0 = bf16, 1 = f16, 2 = f32, 3 = fp8 e4m3fn (OCP; gfx950 is not fnuz).
 *  - "*_is64" flags give the width of an index tensor (0 = int32, 1 = int64), because the reference
 *    holds the same logical tensor as int64 on the scheduler side and int32 in CUDA-graph buffers.
 *
 * Each entry point cites the reference interface it replaces.
 */
#ifndef SGL_MI355_H
#define SGL_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGL_MI355_BF16 0
#define SGL_MI355_F16 1
#define SGL_MI355_F32 2
#define SGL_MI355_FP8_E4M3 3

/* ---- plumbing --------------------------------------------------------------------------- */
const char* sgl_mi355_last_error(void);
int sgl_mi355_abi_version(void);
/* python/sglang/srt/utils.py get_device_core_count (feeds triton_backend.py:124-158) */
int sgl_mi355_device_cu_count(int device);

/* ---- attention -------------------------------------------------------------------------- */
/* Split-KV GQA decode attention + LSE merge.
 * Replaces decode_attention_fwd (python/sglang/srt/layers/attention/triton_ops/decode_attention.py:677-728)
 * when (kv_indptr, kv_indices) are given, and decode_attention_cpu's addressing
 * (sgl-kernel/csrc/cpu/decode.cpp:1375-1575, schema torch_extension_cpu.cpp:264-268) when
 * (req_to_token, req_pool_indices, seq_lens) are given instead (kv_indices == NULL).
 * q [batch, Hq, D] (heads contiguous), k/v_buffer [slots, Hkv, D|Dv], o [batch, Hq, Dv],
 * attn_logits f32 [batch, Hq, max_kv_splits, Dv], attn_lse f32 [batch, Hq, max_kv_splits]. */
int sgl_mi355_decode_attention(const void* q, int64_t q_stride_t, const void* k_buffer, const void* v_buffer,
                               int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h, void* o,
                               int64_t o_stride_t, const int32_t* kv_indptr, const int32_t* kv_indices,
                               const int32_t* req_to_token, int64_t req_to_token_stride, const int64_t* req_pool_indices,
                               const int64_t* seq_lens, float* attn_logits, float* attn_lse, const int32_t* num_kv_splits,
                               int max_kv_splits, int batch, int num_q_heads, int num_kv_heads, int head_dim,
                               int v_head_dim, float sm_scale, float logit_cap, int dtype, int kv_dtype, float k_scale,
                               float v_scale, void* stream);
/* kv_dtype: the q dtype, or SGL_FP8_E4M3 (3) for kv_cache_dtype = fp8_e4m3 pools (memory_pool.py:385-395; head_dim 64 / 128):
 * rows are converted exactly to the q dtype on the way into LDS, K_true = K_fp8 * k_scale, V_true = V_fp8 * v_scale
 * (RadixAttention.k_scale / v_scale, radix_attention.py:73-76; the Triton backend leaves both at 1). */

/* decode_attention_fwd (stage 1) with stage 2 done IN THE SAME LAUNCH: the last workgroup of each request to finish merges the
 * request's splits (decode_attention.py:492-552), writes the merged row (out_o, optional) and its per-token fp8 quantisation
 * (out_q e4m3 [batch, Hq*Dv] + out_s f32 [batch], optional; sgl_per_token_quant_fp8) -- the decode step's
 * attention -> o_proj hand-off in one kernel.  merge_counters: int32 [batch], zero on entry, left zero.  Results are
 * bit-identical to sgl_mi355_decode_attention(o = NULL) + sgl_mi355_decode_merge_quant_fp8.  head_dim 64 / 128. */
int sgl_mi355_decode_attention_merge_quant(const void* q, int64_t q_stride_t, const void* k_buffer, const void* v_buffer,
                                           int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h,
                                           const int32_t* kv_indptr, const int32_t* kv_indices, float* attn_logits,
                                           float* attn_lse, const int32_t* num_kv_splits, int max_kv_splits, int batch,
                                           int num_q_heads, int num_kv_heads, int head_dim, int v_head_dim, float sm_scale,
                                           float logit_cap, int dtype, int kv_dtype, float k_scale, float v_scale,
                                           int32_t* merge_counters, void* out_o, void* out_q, float* out_s, void* stream);
/* sgl_mi355_decode_attention_merge_quant with each workgroup's (request, split) unit read from the sorted list
 * sgl_mi355_decode_schedule wrote for this batch (sched, sched_units = its capacity; num_kv_splits from the same call): the same
 * splits, arithmetic and outputs bit for bit -- another dispatch order (longest unit first, so short units fill the slots long ones
 * leave: BASELINE config 5's ragged batch) and a grid of (kv heads x head chunks, sched_units) without never-live workgroups. */
int sgl_mi355_decode_attention_scheduled(const void* q, int64_t q_stride_t, const void* k_buffer, const void* v_buffer,
                                         int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h,
                                         const int32_t* kv_indptr, const int32_t* kv_indices, float* attn_logits, float* attn_lse,
                                         const int32_t* num_kv_splits, int max_kv_splits, const int32_t* sched, int sched_units,
                                         int batch, int num_q_heads, int num_kv_heads, int head_dim, int v_head_dim, float sm_scale,
                                         float logit_cap, int dtype, int kv_dtype, float k_scale, float v_scale,
                                         int32_t* merge_counters, void* out_o, void* out_q, float* out_s, void* stream);

/* merge_state / merge_state_v2 (sgl-kernel/csrc/attention/merge_attn_states.cu:32-105; python attention.py:12-52): LSE-weighted
 * merge of two attention partials v [n, h, d] (bf16 / f16 / f32) with s f32 [n, h]; s_merged may be NULL. */
int sgl_mi355_merge_state(const void* v_a, const float* s_a, const void* v_b, const float* s_b, void* v_merged, float* s_merged,
                          int64_t num_tokens, int num_heads, int head_size, int dtype, void* stream);

/* Cascade shared-prefix decode ("shared-prefix once, unique-suffix per request"): every request of the batch
 * shares its first prefix_len KV slots -- one RadixCache node (python/sglang/srt/mem_cache/radix_cache.py:370-412).  The
 * prefix rows are attended once for the query heads of ALL requests (a launch of the extend kernel with the batch's decode
 * queries as its query block, at most prefix_splits splits), each request's private part (kv_indptr / kv_indices over the slots
 * AFTER the prefix; at least the new token) separately, its first split continuing the online softmax from the prefix state
 * -- the log-sum-exp combination of merge_state / merge_state_v2 (sgl-kernel/csrc/attention/merge_attn_states.cu,
 * python/sgl_kernel/attention.py:12-52).  Outputs as sgl_mi355_decode_attention_merge_quant.  attn_logits / attn_lse:
 * max_kv_splits slots per (request, head), of which the LAST prefix_splits hold the prefix partials: prefix_splits +
 * max(num_kv_splits) <= max_kv_splits.  head_dim 64 / 128.  merge_counters: int32 [batch], zero on entry, left zero. */
int sgl_mi355_decode_attention_cascade(const void* q, int64_t q_stride_t, const void* k_buffer, const void* v_buffer,
                                       int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h,
                                       const int32_t* prefix_indices, int prefix_len, int prefix_splits,
                                       const int32_t* kv_indptr, const int32_t* kv_indices, float* attn_logits,
                                       float* attn_lse, const int32_t* num_kv_splits, int max_kv_splits, int batch,
                                       int num_q_heads, int num_kv_heads, int head_dim, int v_head_dim, float sm_scale,
                                       float logit_cap, int dtype, int kv_dtype, float k_scale, float v_scale,
                                       int32_t* merge_counters, void* out_o, void* out_q, float* out_s, void* stream);

/* Kernel form of the MFMA decode path.  0 (default) = the waves of a workgroup share one (request, kv head, split) and merge in
 * LDS: four waves per workgroup, or eight (one 512-thread workgroup per CU, round 4) when the launch has at most one (request,
 * kv head, head chunk) unit per two CUs -- sgl_mi355_decode_metadata's balance rule (static_splits = 2) applies the same test and then
 * sizes the splits for one workgroup per CU; 1 = every wave owns one unit (no LDS merge; 1 kv head per rank); measurement hooks:
 * 2 = always four waves, 3 = always eight. */
int sgl_mi355_decode_attention_set_mode(int mode);

/* Kernel choice of sgl_mi355_extend_attention for 16-bit K/V, D = 128, no mask / cap / cascade (all forms implement
 * extend_attention_fwd, extend_attention.py:41-438, with the same arithmetic order and bit-identical outputs): set_mode 0 = always
 * the register-staged kernel, 1 (default) = the LDS-DMA kernel with 4 or 8 waves per workgroup by the mean number of keys a query
 * block attends to (8 from 1536 on), 2 / 3 = always 8 / 4 waves.  set_kv_hint: that mean for the next calls, from host-side lengths
 * (per request prefix + extend / 2); 0 = unknown (no prefix assumed). */
int sgl_mi355_extend_attention_set_mode(int mode);
int sgl_mi355_extend_attention_set_kv_hint(int mean_keys_per_query_block);
/* Extend (prefill / chunked prefill / prefix-cache hit) attention: cached prefix from the paged pool + causal
 * triangle over the contiguous new K/V.  Replaces extend_attention_fwd
 * (python/sglang/srt/layers/attention/triton_ops/extend_attention.py:306-438) with (qo_indptr, kv_indptr,
 * kv_indices), and extend_attention_cpu's addressing (sgl-kernel/csrc/cpu/extend.cpp:579-723, schema
 * torch_extension_cpu.cpp:270-275) with (req_to_token, req_pool_indices, seq_lens, extend_seq_lens,
 * extend_start_loc) when qo_indptr == NULL.  q/o [T, Hq, D], k/v_extend [T, Hkv, D] (heads contiguous).
 * custom_mask (u8 / bool, request b's [ext_len, prefix_len + ext_len] block at mask_indptr[b]; NULL = none),
 * skip_prefix_custom_mask and sliding_window_size (<= 0 = none) follow extend_attention.py:131-203 (prefix phase: mask
 * unless skipped, window q_idx <= kv_idx + W over the kv_indices the caller windowed, triton_backend.py:927-955) and
 * :205-284 (extend phase: the custom mask REPLACES the causal rule). */
int sgl_mi355_extend_attention(const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend,
                               int64_t q_stride_t, int64_t k_stride_t_ext, int64_t v_stride_t_ext, int64_t o_stride_t,
                               const void* k_buffer, const void* v_buffer, int64_t k_stride_t, int64_t k_stride_h,
                               int64_t v_stride_t, int64_t v_stride_h, const int32_t* qo_indptr,
                               const int32_t* kv_indptr, const int32_t* kv_indices, const int32_t* req_to_token,
                               int64_t req_to_token_stride, const int64_t* req_pool_indices, const int64_t* seq_lens,
                               const int32_t* extend_seq_lens, const int32_t* extend_start_loc, int batch,
                               int total_q_tokens, int max_len_extend, int num_q_heads, int num_kv_heads, int head_dim,
                               int v_head_dim, float sm_scale, float logit_cap, int is_causal, int dtype, int kv_dtype,
                               float k_scale, float v_scale, const uint8_t* custom_mask, const int64_t* mask_indptr,
                               int skip_prefix_custom_mask, int sliding_window_size, void* stream);

/* ---- KV pool / index kernels (bit-exact) -------------------------------------------------- */
/* create_flashinfer_kv_indices_triton, python/sglang/srt/layers/attention/utils.py:10-45 */
int sgl_mi355_create_kv_indices(const int32_t* req_to_token, int64_t req_to_token_stride, const void* req_pool_indices,
                                int req_pool_indices_is64, const void* page_kernel_lens, int page_kernel_lens_is64,
                                const int32_t* kv_indptr, const void* kv_start_idx, int kv_start_idx_is64,
                                int32_t* kv_indices, int batch, void* stream);
/* compute_position_triton / compute_position_torch, model_executor/forward_batch_info.py:885-955 */
int sgl_mi355_compute_position(int64_t* positions, int32_t* extend_start_loc, const void* extend_prefix_lens,
                               int prefix_is64, const void* extend_seq_lens, int seq_is64, int batch, void* stream);
/* write_req_to_token_pool_triton, managers/schedule_batch.py:1920-1955 */
int sgl_mi355_write_req_to_token(int32_t* req_to_token, int64_t req_to_token_stride, const void* req_pool_indices,
                                 int req_pool_indices_is64, const void* pre_lens, int pre_is64, const void* seq_lens,
                                 int seq_is64, const void* extend_lens, int ext_is64, const int64_t* out_cache_loc,
                                 int batch, void* stream);
/* Graph-replayed decode step, host side in ONE launch: prepare_for_decode (schedule_batch.py:1560-1590: req_to_token[req,
 * seq_len] = out_cache_loc, seq_lens += 1 in place) + replay_prepare's copies into the graph's static input buffers
 * (cuda_graph_runner.py:700-760; positions = clamp(seq_lens - 1, 0), forward_batch_info.py:958-960). */
int sgl_mi355_decode_prepare(const int64_t* req_pool_indices, int64_t* seq_lens, const int64_t* out_cache_loc,
                             const int64_t* next_ids, int32_t* req_to_token, int64_t req_to_token_stride,
                             int64_t* buf_input_ids, int64_t* buf_req_pool_indices, int64_t* buf_seq_lens,
                             int64_t* buf_out_cache_loc, int64_t* buf_positions, int batch, void* stream);
/* get_last_loc_triton / get_last_loc_torch, managers/schedule_batch.py:1958-2028 */
int sgl_mi355_get_last_loc(const int32_t* req_to_token, int64_t req_to_token_stride, const void* req_pool_indices,
                           int req_pool_indices_is64, const void* prefix_lens, int prefix_is64, void* result,
                           int result_is64, int64_t n, void* stream);
/* MHATokenToKVPool.set_kv_buffer, mem_cache/memory_pool.py:369-407 (K and V scatter in one launch) */
int sgl_mi355_set_kv_buffer(void* k_buffer, void* v_buffer, int64_t k_slot_bytes, int64_t v_slot_bytes,
                            const int64_t* loc, const void* cache_k, const void* cache_v, int64_t cache_k_stride_bytes,
                            int64_t cache_v_stride_bytes, int k_row_bytes, int v_row_bytes, int64_t tokens, void* stream);
/* The same scatter for kv_cache_dtype = fp8_e4m3 (memory_pool.py:385-395): cache_k.div_(k_scale) in the source dtype when a
 * scale is given (k_scale / v_scale <= 0: none), then .to(float8_e4m3fn) with torch's rounding and NaN rule.  Strides and
 * row sizes in elements of the bf16 / f16 source; pool slot strides in bytes. */
int sgl_mi355_set_kv_buffer_fp8(void* k_buffer, void* v_buffer, int64_t k_slot_bytes, int64_t v_slot_bytes, const int64_t* loc,
                                const void* cache_k, const void* cache_v, int64_t cache_k_stride, int64_t cache_v_stride,
                                int k_row, int v_row, int64_t tokens, int src_dtype, float k_scale, float v_scale,
                                void* stream);
/* MHATokenToKVPool.move_kv_cache -> copy_all_layer_kv_cache (memory_pool.py:409-417, 1046-1081): buf[tgt_loc[i]] = buf[src_loc[i]]
 * for every buffer in data_ptrs (uint64 [num_buffers], device) with rows of data_strides[b] bytes (int64 [num_buffers], device;
 * multiples of 16), in place: every source row of a column block is read before any target row of it is written, as in the
 * reference kernel.  Byte-exact.  num_locs <= 4096 per call. */
int sgl_mi355_move_kv_cache(const void* data_ptrs, const int64_t* data_strides, int num_buffers, int64_t max_stride_bytes,
                            const void* tgt_loc, int tgt_is64, const void* src_loc, int src_is64, int num_locs, void* stream);
/* kv_indptr[1:bs+1] = cumsum(seq_lens) (triton_backend.py:172) and get_num_kv_splits_triton
 * (triton_backend.py:876-924); either output may be NULL.  static_splits: 0 = reference heuristic, 1 = max everywhere,
 * 2 = MI355X balance rule (about two rounds of resident workgroups). */
int sgl_mi355_decode_metadata(int32_t* kv_indptr, int32_t* num_kv_splits, const void* seq_lens, int seq_is64,
                              int num_seq, int num_group, int num_head, int num_kv_head, int max_kv_splits,
                              int device_core_count, int static_splits, void* stream);

/* Sorted unit list of a decode batch (replaces get_num_kv_splits_triton, triton_backend.py:876-924, for the launch
 * sgl_mi355_decode_attention_scheduled): kv_indptr[1:bs+1] = cumsum(seq_lens); num_kv_splits[b] = clamp(ceil(len / T), 1,
 * max_kv_splits) with T = the 32-token multiple of total / (resident workgroups x rounds_pct / 100 per kv head x head chunk), >= 64;
 * sched = {T, units, total, capacity} + per (request, split) unit {request, split | splits << 16, kv_indptr[request], length},
 * sorted by the unit's token count, longest first (split boundaries as decode_attention.py:90-94 makes them).
 * sgl_mi355_decode_schedule_units: the list's capacity = the attention launch's grid y (0: geometry not served, use
 * sgl_mi355_decode_metadata); sched holds 4 + 4 x capacity int32 words, 16-byte aligned. */
int sgl_mi355_decode_schedule_units(int num_seq, int num_head, int num_kv_head, int rounds_pct);
int sgl_mi355_decode_schedule(int32_t* kv_indptr, int32_t* num_kv_splits, int32_t* sched, int sched_units, const void* seq_lens,
                              int seq_is64, int num_seq, int num_head, int num_kv_head, int max_kv_splits, int rounds_pct,
                              void* stream);

/* alloc_extend_kernel / alloc_decode_kernel of PagedTokenToKVPoolAllocator, mem_cache/allocator.py:275-395:
 * page-aligned slot assignment; ret_values[0] = (new pages << 32 | extend tokens) resp. new pages. */
int sgl_mi355_alloc_extend(const void* prefix_lens, int prefix_is64, const void* seq_lens, int seq_is64,
                           const void* last_loc, int last_loc_is64, const int64_t* free_pages, int64_t* out_indices,
                           int64_t* ret_values, int page_size, int batch, void* stream);
int sgl_mi355_alloc_decode(const void* seq_lens, int seq_is64, const void* last_loc, int last_loc_is64,
                           const int64_t* free_pages, int64_t* out_indices, int64_t* ret_values, int page_size, int batch,
                           void* stream);

/* ---- fp8 activation quantisation (bit-exact with the reference's torch restatements) -------- */
/* sgl_per_token_quant_fp8, sgl-kernel/csrc/gemm/per_token_quant_fp8.cu:15-228; python gemm.py:140-145 */
int sgl_mi355_per_token_quant_fp8(const void* input, int64_t input_stride, void* output_q, float* output_s,
                                  int64_t num_tokens, int64_t hidden_dim, int in_dtype, void* stream);
/* sgl_per_tensor_quant_fp8, sgl-kernel/csrc/gemm/per_tensor_quant_fp8.cu:10-123; python gemm.py:129-137.
 * Dynamic mode (is_static == 0) atomically maxes into *output_s, which the caller zero-initialises. */
int sgl_mi355_per_tensor_quant_fp8(const void* input, void* output_q, float* output_s, int64_t num_elements,
                                   int is_static, int in_dtype, void* stream);
/* input_to_float8, python/sglang/srt/layers/quantization/fp8_utils.py:310-326 (weights that arrive unquantised: w8a8_fp8.py:129,
 * fp8.py:375): amax = max |x| clamped at 1e-12, scale = 448 / amax, q = sat(x * scale); *output_scale_inv = 1 / scale.  The reference's
 * arithmetic order (sgl_per_tensor_quant_fp8 multiplies by 1 / (amax / 448) instead).  amax_scratch: one float of device memory. */
int sgl_mi355_input_to_float8(const void* input, void* output_q, float* output_scale_inv, float* amax_scratch,
                              int64_t num_elements, int in_dtype, void* stream);
/* sgl_per_token_group_quant_fp8, sgl-kernel/csrc/gemm/per_token_group_quant_8bit.cu; python gemm.py:100-112
 * (row-major float scales; scale_ue8m0 / column-major layouts are DeepSeek-only and out of scope) */
int sgl_mi355_per_token_group_quant_fp8(const void* input, void* output_q, float* output_s, int64_t num_elements,
                                        int group_size, float eps, float fp8_min, float fp8_max, int in_dtype,
                                        void* stream);

/* ---- GEMM --------------------------------------------------------------------------------- */
/* Weight-streaming GEMM for M <= 64: Y = (X . W^T) * scales_x[m] * scales_w[n] + bias[n].
 * in_dtype fp8: fp8_scaled_mm (sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1071-1146, python gemm.py:34-42) with
 * W = mat_b^T stored [N, K] row-major; in_dtype bf16/f16 with NULL scales: the unquantised linear
 * (python/sglang/srt/layers/quantization/unquant.py) used by lm_head.
 * K * esize > 4096 bytes is split into k-ranges whose f32 partial sums go through `workspace`
 * (sgl_mi355_skinny_gemm_num_kranges(...) slabs of M*N floats; NULL or too small -> the slower any-K kernel). */
int sgl_mi355_skinny_gemm(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems, void* y,
                          int64_t y_stride_elems, const float* scales_x, const float* scales_w, const void* bias, int M,
                          int N, int K, int in_dtype, int out_dtype, float* workspace, int64_t workspace_floats,
                          void* stream);
int sgl_mi355_skinny_gemm_num_kranges(int M, int N, int K, int in_dtype);
/* Producer half of the launch-boundary split-K reduce: raw f32 partial sums [kranges, M, N], scales left to the consumer
 * (sgl_mi355_fused_add_rmsnorm_quant_fp8 with slabs != NULL). fp8, M <= 64. */
/* how many slabs the next function writes for rows of K BYTES (K elements for fp8, 2 K for bf16 / f16): the same k-range
 * partition as sgl_mi355_skinny_gemm's, so both sum identically.  in_dtype of the next function: fp8 / bf16 / f16. */
int sgl_mi355_skinny_gemm_slabs_count(int M, int K);
int sgl_mi355_skinny_gemm_slabs(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems, float* slabs,
                                int M, int N, int K, int in_dtype, void* stream);
/* The same with at least `min_kranges` k-ranges (shorter K slices per workgroup; for a consumer that sums the slabs anyway:
 * o_proj's partial sums into the post-attention add + RMSNorm + quant); ..._count_min = the number of slabs written. */
int sgl_mi355_skinny_gemm_slabs_count_min(int M, int K, int min_kranges);
int sgl_mi355_skinny_gemm_slabs_min(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems, float* slabs,
                                    int M, int N, int K, int in_dtype, int min_kranges, void* stream);

/* Test hook: route every skinny GEMM through the generic (any-K) kernel instead of the X-stationary one. */
int sgl_mi355_skinny_gemm_force_generic(int on);
/* Test hook for sgl_mi355_fp8_gemm's tile choice: 0 = by shape, 1 = always 128x128 tiles, 2 = 256x256 tiles whenever K is
  * a multiple of 128 (8 waves), 3 = the same tile with 4 waves of 128x128 outputs (accumulators in AGPRs), 4 = register-staged 256x256, 5 = the 4-stage
  * streaming 128x128 tile (default for M <= 256), 7 = the 256x128 tile (eight waves of 64x64 outputs; default where 256x256 tiles are
  * fewer than the CUs and half-size tiles fill them); 100 + g = scheduling group height g of the 256x256 kernel; 3001 / 3000 =
  * persistent form of the 256x256 kernel where a launch has at least two tiles per CU (default) / one tile per workgroup always;
  * 5001 / 5000 = ping-pong schedule of the fp8 256x256 kernel (default) / round 4's one-barrier kernels (persistent where that won); 5002 / 5003 =
  * the ping-pong schedule with four / two phases per K slice. */
int sgl_mi355_fp8_gemm_force_tile(int mode);
/* Tiled MFMA GEMM for M > 64 with the same contract as sgl_mi355_skinny_gemm's fp8 case:
 * fp8_scaled_mm, sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1071-1146 (CUTLASS tile dispatch :303-440,739-796). */
int sgl_mi355_fp8_gemm(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems, void* y,
                       int64_t y_stride_elems, const float* scales_x, const float* scales_w, const void* bias, int M,
                       int N, int K, int out_dtype, float* workspace, int64_t workspace_floats, void* stream);
/* Split-K of sgl_mi355_fp8_gemm where it runs the streaming tile with fewer tiles than CUs (64 < M <= 256 always; larger M when the
 * 256x256 tile would leave CUs idle) left to the consumer: num_slabs = how many f32 [M, N]
 * partial sums it forms for this shape and scratch size (1: none); fp8_gemm_slabs writes them raw (no scales) for
 * sgl_mi355_fused_add_rmsnorm_quant_fp8 (slabs + sx + sw), with fp8_gemm's own k-range partition. */
int sgl_mi355_fp8_gemm_num_slabs(int M, int N, int K, int64_t workspace_floats);
/* Which kernel sgl_mi355_fp8_gemm picks for this shape (contiguous rows; host logic only): 0 = 256x256 tile, 1 = streaming 128x128
 * tile, 2 = 256x128 tile.  The counterpart of the reference's shape-keyed CUTLASS tile dispatch (fp8_gemm_kernel.cu:303-440). */
int sgl_mi355_fp8_gemm_tile_choice(int M, int N, int K, int64_t workspace_floats);
int sgl_mi355_fp8_gemm_slabs(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems, float* slabs, int M,
                             int N, int K, int64_t workspace_floats, void* stream);
/* (workspace: optional f32 scratch for split-K when a launch has fewer 128x128 output tiles than half the CUs -- decode at
 * 64 < M <= 256, the continuous-batching regime, where the weights are streamed once; NULL disables it)
 * Round 4: workspaces of at least 4096 floats also lend their LAST 512 words to the persistent 256x256 kernel as 64 eight-word
 * ticket slots (dynamic per-XCD tile schedule; taken round robin, zeroed by the launcher).  A workspace is this call's scratch: one
 * workspace must not serve GEMMs of two streams at the same time (the split-K slabs at its head would collide). */
/* Unquantised bf16/f16 linear for M > 64 (UnquantizedLinearMethod.apply, layers/quantization/unquant.py);
 * also the matmul half of AWQLinearMethod.apply (layers/quantization/awq.py:401-418). */
int sgl_mi355_dense_gemm(const void* x, int64_t x_stride_elems, const void* w, int64_t w_stride_elems, void* y,
                         int64_t y_stride_elems, const void* bias, int M, int N, int K, int in_dtype, int out_dtype,
                         float* workspace, int64_t workspace_floats, void* stream);
/* awq_dequantize, sgl-kernel/csrc/gemm/awq_kernel.cu:127-221 (python gemm.py:7-10); HIP reference path
 * awq_dequantize_triton, layers/quantization/awq_triton.py:14-108.  out [K, 8*num_packed_cols] in the scale dtype. */
int sgl_mi355_awq_dequantize(const void* qweight, const void* scales, const void* qzeros, void* out, int K,
                             int num_packed_cols, int group_size, int scale_dtype, void* stream);

/* ---- elementwise neighbours on the step path -------------------------------- */
/* RMSNorm.forward_native, python/sglang/srt/layers/layernorm.py:135-171; sgl_kernel rmsnorm / fused_add_rmsnorm
 * (sgl-kernel/csrc/elementwise/fused_add_rms_norm_kernel.cu).  residual != NULL: residual += x, out = norm(sum). */
int sgl_mi355_rmsnorm(void* out, const void* x, void* residual, const void* weight, float eps, int64_t tokens,
                      int hidden, int64_t x_stride, int64_t out_stride, int dtype, void* stream);
/* SiluAndMul.forward_native, layers/activation.py:60-63; sgl_kernel silu_and_mul (csrc/elementwise/activation.cu) */
int sgl_mi355_silu_and_mul(void* out, const void* x, int64_t tokens, int d, int dtype, void* stream);
/* RotaryEmbedding.forward_native, layers/rotary_embedding.py:49-72,138-165 (in place on query/key) */
int sgl_mi355_rotary_embedding(const int64_t* positions, void* query, void* key, const float* cos_sin_cache,
                               int64_t tokens, int num_q_heads, int num_k_heads, int head_size, int rot_dim,
                               int64_t q_stride, int64_t k_stride, int is_neox, int dtype, void* stream);
/* VocabParallelEmbedding forward (F.embedding), layers/vocab_parallel_embedding.py; vocab = rows of the table: an id outside
 * [0, vocab) yields a row of NaNs (F.embedding asserts on the device) -- never an out-of-range read */
int sgl_mi355_embedding(void* out, const int64_t* ids, const void* table, int64_t tokens, int hidden, int64_t vocab, int dtype,
                        void* stream);
/* greedy sampling: torch.argmax(logits, -1), layers/sampler.py */
int sgl_mi355_argmax(int64_t* out, const void* logits, int64_t rows, int64_t vocab, int64_t row_stride, int dtype,
                     void* stream);

/* ---- fused decode-step kernels: bit-identical to the op sequences they replace ------------------------------ */
/* [split-K combine ->] fused_add_rmsnorm (layernorm.py:135-171) -> sgl_per_token_quant_fp8.  Exactly one of x / slabs. */
int sgl_mi355_fused_add_rmsnorm_quant_fp8(const void* x, const float* slabs, int nslabs, const float* slab_sx,
                                          const float* slab_sw, void* residual, const void* weight, float eps,
                                          void* out_norm, void* out_q, float* out_s, int tokens, int hidden, int dtype,
                                          void* stream);
/* gate_up_proj (fp8_scaled_mm) + SiluAndMul (models/llama.py:94-98) in one launch; weight rows interleaved per 16-row tile
 * (8 gate rows, 8 up rows).  act [M, N/2]. */
int sgl_mi355_fp8_gemm_silu_mul(const void* x, int64_t x_stride_elems, const void* w_interleaved, int64_t w_stride_elems,
                                void* act, int64_t act_stride_elems, const float* scales_x,
                                const float* scales_w_interleaved, int M, int N, int K, int out_dtype, int tile_rows, void* stream);
/* Per-device initialisation for the M > 64 form of the SiluAndMul fusion above (SiluAndMul.forward_native, activation.py:60-63,
 * evaluated through a table that is bit-identical to it by construction): runs the table fill on `stream` of the current device
 * and waits for it -- when it returns the table is ready for every stream of the device.  Call once per device, outside stream
 * capture (REQUIRED call order since round 4: the M > 64 form returns SGL_MI355_EINVAL until it has been called on the device;
 * round 3 filled the table inside the first launch).  Idempotent. */
int sgl_mi355_silu_table_init(void* stream);
/* qkv_proj (fp8_scaled_mm) + neox RoPE (rotary_embedding.py:49-72) + set_kv_buffer (memory_pool.py:401-407) in one launch
 * (models/llama.py:180-191); rows interleaved inside every q/k head so rotation pairs share a 16-row tile. */
int sgl_mi355_fp8_qkv_rope_set_kv(const void* x, int64_t x_stride_elems, const void* w_interleaved, int64_t w_stride_elems,
                                  void* q_out, int64_t q_stride_elems, const float* scales_x,
                                  const float* scales_w_interleaved, const void* bias_interleaved,
                                  const int64_t* positions, const float* cos_sin_cache, const int64_t* loc, void* k_buffer,
                                  void* v_buffer, int64_t k_slot_stride, int64_t v_slot_stride, int M, int num_q_heads,
                                  int num_kv_heads, int head_dim, int K, int out_dtype, int tile_rows,
                                  void* stream);
/* The same two fusions for either operand type: in_dtype SGL_FP8_E4M3 (both scale vectors, out bf16 / f16) or SGL_BF16 /
 * SGL_F16 (UnquantizedLinearMethod.apply, quantization/unquant.py: scales NULL, out_dtype == in_dtype).  K must fit one
 * k-range of the weight-streaming kernel: 4096 bytes per row, 8192 at M <= 32.  kv_dtype of the RoPE form: the pool's dtype =
 * out_dtype, or SGL_FP8_E4M3 (kv_cache_dtype fp8_e4m3: set_kv_buffer's div_(scale) + e4m3 conversion, memory_pool.py:385-395;
 * k_scale / v_scale <= 0: no scale). */
int sgl_mi355_gemm_silu_mul(const void* x, int64_t x_stride_elems, const void* w_interleaved, int64_t w_stride_elems, void* act,
                            int64_t act_stride_elems, const float* scales_x, const float* scales_w_interleaved, int M, int N,
                            int K, int in_dtype, int out_dtype, int tile_rows, void* stream);
/* sgl_mi355_gemm_silu_mul with per-launch scratch: sched = 8 int32 words (32-byte aligned is best) of device memory that no other launch
 * in flight uses (one buffer per stream suffices; the Python wrapper rotates over 64 slots).  With it the M > 64 (prefill) form runs the persistent 256 x 256 kernel on a dynamic per-XCD tile
 * schedule -- the same bits; NULL: exactly sgl_mi355_gemm_silu_mul.  (models/llama.py:94-98 at prefill sizes.) */
int sgl_mi355_gemm_silu_mul_ws(const void* x, int64_t x_stride_elems, const void* w_interleaved, int64_t w_stride_elems, void* act,
                               int64_t act_stride_elems, const float* scales_x, const float* scales_w_interleaved, int M, int N,
                               int K, int in_dtype, int out_dtype, int tile_rows, void* sched, void* stream);
int sgl_mi355_qkv_rope_set_kv(const void* x, int64_t x_stride_elems, const void* w_interleaved, int64_t w_stride_elems,
                              void* q_out, int64_t q_stride_elems, const float* scales_x, const float* scales_w_interleaved,
                              const void* bias_interleaved, const int64_t* positions, const float* cos_sin_cache,
                              const int64_t* loc, void* k_buffer, void* v_buffer, int64_t k_slot_stride, int64_t v_slot_stride,
                              int M, int num_q_heads, int num_kv_heads, int head_dim, int K, int in_dtype, int out_dtype,
                              int tile_rows, int kv_dtype, float k_scale, float v_scale, void* stream);
/* ---- persistent MLP half of a w8a8-fp8 decode layer (round 3; csrc/mlp_block.hip) --------------------------------------
 * ONE launch for LlamaDecoderLayer's post_attention_layernorm (fused add + RMSNorm, models/llama.py:265, layernorm.py:135-171)
 * -> per-token fp8 quant (fp8_utils.py:706-713) -> gate_up_proj + SiluAndMul (models/llama.py:94-96) -> per-token fp8 quant ->
 * down_proj (models/llama.py:97, linear.py:1285-1309) at M <= 32: one resident workgroup per CU whose weight loads run ahead of
 * the three chip-wide hand-offs between the steps.  x [M, hidden] = o_proj's output, residual [M, hidden] in/out; weights as for
 * sgl_mi355_fp8_gemm_silu_mul (16-row interleaving) and fp8_scaled_mm ([hidden, inter] row-major); out_slabs f32
 * [ceil(inter / 4096), M, hidden] raw down_proj partial sums, act_scales [M] their per-token scale (the consumer, e.g.
 * sgl_mi355_fused_add_rmsnorm_quant_fp8 with slabs, applies act_scales[m] * down_scale[n]).  Scratch: xq [M, hidden] bytes,
 * xs [M] f32, actq [M, inter] bytes, pmax (..._pmax_words uint32, 128-byte aligned), sync (..._sync_words uint32, 128-byte
 * aligned, ZEROED by the caller before every launch; word 16 != 0 afterwards: a hand-off timed out).  timeline: NULL, or
 * [CUs][8][16] int64 s_memrealtime stamps.  Arithmetic = the stand-alone kernels' (same rounding points; the RMSNorm variance is
 * summed in another order).  ..._supported: 1 if the shape is taken (else keep the four-launch path). */
int sgl_mi355_fp8_mlp_block_sync_words(void);
int sgl_mi355_fp8_mlp_block_pmax_words(void);
int sgl_mi355_fp8_mlp_block_supported(int M, int hidden, int inter);
int sgl_mi355_fp8_mlp_block(const void* x, void* residual, const void* ln_weight, float eps, const void* w_gate_up_interleaved,
                            const float* scales_gate_up_interleaved, const void* w_down, float* out_slabs, float* act_scales,
                            void* xq_scratch, float* xs_scratch, void* actq_scratch, void* pmax_scratch, void* sync, int M,
                            int hidden, int inter, int dtype, long long* timeline, void* stream);
/* Measurement / test hook for sgl_mi355_silu_and_mul_quant_fp8: 1 (default) = prefill-sized bf16 launches (tokens >= 1024) take the
 * kernel that tabulates T(silu(a)) over a's 16 bits in LDS (bit-identical by construction), 0 = always the exact-expression kernel. */
int sgl_mi355_silu_and_mul_quant_set_mode(int table);
/* silu_and_mul (activation.py:60-63) -> sgl_per_token_quant_fp8 */
int sgl_mi355_silu_and_mul_quant_fp8(const void* x, void* out_q, float* out_s, int tokens, int d, int dtype, void* stream);
/* rotary_embedding (rotary_embedding.py:138-165) on q, k in place -> set_kv_buffer (memory_pool.py:369-407) of (k, v) */
int sgl_mi355_rope_set_kv(const int64_t* positions, void* query, void* key, const void* value, const float* cos_sin_cache,
                          void* k_buffer, void* v_buffer, const int64_t* loc, int64_t tokens, int num_q_heads,
                          int num_k_heads, int head_size, int rot_dim, int64_t q_stride, int64_t k_stride,
                          int64_t v_stride, int64_t k_slot_stride, int64_t v_slot_stride, int is_neox, int dtype,
                          void* stream);
/* decode stage-2 LSE merge (decode_attention.py:492-552) -> sgl_per_token_quant_fp8 of the [batch, Hq*Dv] output */
int sgl_mi355_decode_merge_quant_fp8(const float* attn_logits, const float* attn_lse, const int32_t* kv_indptr,
                                     const int64_t* seq_lens, const int32_t* num_kv_splits, int max_kv_splits, int batch,
                                     int num_q_heads, int v_head_dim, void* out_o, void* out_q, float* out_s, int dtype,
                                     void* stream);
/* torch.argmax(logits, -1) with 16-byte loads (bf16/f16 rows) */
int sgl_mi355_argmax_vec(int64_t* out, const void* logits, int64_t rows, int64_t vocab, int64_t row_stride, int dtype,
                         void* stream);

/* ---- native radix tree for RadixAttention prefix sharing (host side, bit-exact slot indices) ------------------ */
/* python/sglang/srt/mem_cache/radix_cache.py:43-555 (TreeNode, RadixCache._match_prefix_helper / _split_node /
 * _insert_helper / evict / inc_lock_ref / dec_lock_ref); keys are token ids, values KV slot indices, both int64.
 * page_size > 1 matches whole pages (:111-120).  Node handles are int64 ids. */
void* sgl_mi355_radix_create(int page_size);
int sgl_mi355_radix_destroy(void* tree);
int sgl_mi355_radix_reset(void* tree);
int64_t sgl_mi355_radix_root(void* tree);
int64_t sgl_mi355_radix_match_prefix(void* tree, const int64_t* key, int64_t key_len, int64_t* out_values, int64_t out_cap,
                                     int64_t* last_node);
int64_t sgl_mi355_radix_insert(void* tree, const int64_t* key, const int64_t* values, int64_t len);
int64_t sgl_mi355_radix_evict(void* tree, int64_t num_tokens, int64_t* out_values, int64_t out_cap, int64_t* out_node_lens,
                              int64_t lens_cap, int64_t* n_nodes);
int64_t sgl_mi355_radix_inc_lock_ref(void* tree, int64_t node);
int64_t sgl_mi355_radix_dec_lock_ref(void* tree, int64_t node);
int64_t sgl_mi355_radix_evictable_size(void* tree);
int64_t sgl_mi355_radix_protected_size(void* tree);
int64_t sgl_mi355_radix_total_size(void* tree);
int64_t sgl_mi355_radix_num_nodes(void* tree);
int64_t sgl_mi355_radix_node_info(void* tree, int64_t node, int64_t* parent, int64_t* lock_ref, int64_t* num_children);

/* Fused int4 dequant + GEMM for decode-sized M (<= 32): the product AWQLinearMethod.apply computes as awq_dequantize +
 * torch.matmul (python/sglang/srt/layers/quantization/awq.py:401-418), with the weight values bit-identical to
 * sgl_mi355_awq_dequantize's and nothing but int4 read from HBM.  The weight is re-laid once at load time
 * (process_weights_after_loading, awq.py:393-399; the role awq_marlin_repack plays for the reference's Marlin path,
 * sgl-kernel/csrc/gemm/marlin/awq_marlin_repack.cu) into MFMA-fragment order:
 *   qpacked int32 [N/16][K/128][64][4], sz int32 [K/G][N] = (zero << 16) | scale bits (bf16 scales) or ((0xE400 | zero) << 16) |
 *   scale bits (f16 scales: the upper half is the f16 constant -(1024 + zero), round 4).   K % 128 == 0, N % 16 == 0,
 *   group_size 32, 64 or a multiple of 128. */
int sgl_mi355_awq_repack(const void* qweight, const void* scales, const void* qzeros, void* qpacked, void* sz, int K, int N,
                         int group_size, int scale_dtype, void* stream);
/* number of f32 [M, N] slabs sgl_mi355_awq_gemm needs as workspace for this K (1: none) */
/* Dense W [N, K] (row-major, scale dtype) from the repacked image: awq_dequantize + transpose in one pass for the M > 64
 * (prefill) matmul of AWQLinearMethod.apply (awq.py:401-418); values are exactly awq_dequantize's. */
int sgl_mi355_awq_unpack_nk(const void* qpacked, const void* sz, void* out, int N, int K, int group_size, int dtype,
                            void* stream);
/* split-K ranges of sgl_mi355_awq_gemm (f32 [M, N] slabs of workspace it wants; 1: none): 4096 k per range for M <= 32, 2048 k
 * for 33..64 rows */
int sgl_mi355_awq_gemm_num_kranges(int M, int K);
/* Parity switch (process-wide): 1 = the int4 GEMM multiplies by awq_dequantize's weights rounded to the scale dtype for bf16 too,
 * bit-compatible with the reference's awq_dequantize -> torch.matmul (awq.py:401-418); 0 (default) = bf16 operands with one scale
 * group per 128-k block use the exact (q - z) * s (faster, closer to exact arithmetic, not bit-identical to dequantise + matmul). */
int sgl_mi355_awq_set_exact_weights(int on);
int sgl_mi355_awq_gemm(const void* x, int64_t x_stride_elems, const void* qpacked, const void* sz, void* y,
                       int64_t y_stride_elems, const void* bias, int M, int N, int K, int group_size, int dtype,
                       float* workspace, int64_t workspace_floats, void* stream);
/* Producer half of the launch-boundary split-K reduce for an int4 weight: raw f32 partial sums [kranges, M, N]
 * (kranges = sgl_mi355_awq_gemm_num_kranges(M, K)); consumed by sgl_mi355_fused_add_rmsnorm_quant_fp8 (slabs, no scales). */
int sgl_mi355_awq_gemm_slabs(const void* x, int64_t x_stride_elems, const void* qpacked, const void* sz, float* slabs, int M,
                             int N, int K, int group_size, int dtype, void* stream);
/* AWQLinearMethod.apply (awq.py:401-418) fused with the op that consumes it, bit-identical to awq_gemm followed by that op; the
 * packed columns (8 per int32) are interleaved before sgl_mi355_awq_repack.  K <= 4096 (one k-range), M <= 32.
 *   gate_up_proj + SiluAndMul (activation.py:60-63): 16-column tile t = [gate columns 8t..8t+7 | up columns 8t..8t+7] */
int sgl_mi355_awq_gemm_silu_mul(const void* x, int64_t x_stride_elems, const void* qpacked_interleaved, const void* sz_interleaved,
                                void* act, int64_t act_stride_elems, int M, int N, int K, int group_size, int dtype, void* stream);
/*   qkv_proj + neox RoPE (rotary_embedding.py:49-72) + set_kv_buffer (memory_pool.py:401-407): inside every q / k head tile
 *   u = [columns 8u..8u+7 | columns 64+8u..64+8u+7]; v heads in natural order; head_dim = rotary_dim = 128 */
int sgl_mi355_awq_qkv_rope_set_kv(const void* x, int64_t x_stride_elems, const void* qpacked_interleaved,
                                  const void* sz_interleaved, void* q_out, int64_t q_stride_elems, const void* bias_interleaved,
                                  const int64_t* positions, const float* cos_sin_cache, const int64_t* loc, void* k_buffer,
                                  void* v_buffer, int64_t k_slot_stride, int64_t v_slot_stride, int M, int num_q_heads,
                                  int num_kv_heads, int head_dim, int K, int group_size, int dtype, int kv_dtype, float k_scale,
                                  float v_scale, void* stream);

/* out[cols, rows] = in[rows, cols]^T for 16-bit elements (weight re-layout between awq_dequantize's [K, N] and
 * the [N, K] the GEMMs stream; AWQLinearMethod.apply, layers/quantization/awq.py:401-418) */
int sgl_mi355_transpose_2d(void* out, const void* in, int rows, int cols, void* stream);

/* ---- P2P all-reduce over IPC-mapped peer buffers (one-shot and two-stage) ----------------------
 * The tensor-parallel all-reduce of the decode path (RowParallelLinear, python/sglang/srt/layers/linear.py:1302-1303 ->
 * GroupCoordinator.all_reduce, distributed/parallel_state.py:480-500, which prefers the custom all-reduce for small
 * messages): sgl-kernel/csrc/allreduce/custom_all_reduce_hip.cuh:261,294,543-549 (one-stage cross-device reduce) and
 * python/sglang/srt/distributed/device_communicators/custom_all_reduce.py (buffer registration over IPC handles).
 * car_alloc: one uncached device allocation per rank (signal block + the data halves of the four kernel families, 16 x
 * max_bytes: each family -- one-shot plain / gather, one-shot fused, two-stage plain, two-stage fused -- keeps its own flags and
 * halves because the no-closing-barrier protocol needs ONE packet -> block map per pair of halves) and its 64-byte IPC handle; car_open / car_close: map / unmap a peer's allocation; car_all_reduce: in-place sum over `world` ranks, every
 * rank reads every peer once and adds in rank order with f32 accumulation (bit-identical on all ranks), HIP-graph capturable;
 * car_error: 1 if a peer failed to arrive within the spin bound since the last query (the call's output is then undefined). */
int sgl_mi355_car_alloc(int64_t max_bytes, void** ptr_out, void* handle_out);
int sgl_mi355_car_open(const void* handle, void** ptr_out);
int sgl_mi355_car_close(void* peer_ptr);
int sgl_mi355_car_free(void* own_ptr);
int sgl_mi355_car_error(void* own_ptr);
int sgl_mi355_car_all_reduce(void* inout, int64_t num_elements, int dtype, const void* const* peer_bufs, int rank, int world,
                             int64_t max_bytes, void* stream);
/* The same with the algorithm chosen by the caller: algo 0 = the reference's dispatch (custom_all_reduce_hip.cuh:543-549: one
 * stage at 2 ranks, below 512 KiB at <= 4 ranks, below 256 KiB at <= 8 ranks), 1 = one-shot (cross_device_reduce_1stage, :261),
 * 2 = two-stage (cross_device_reduce_2stage, :294: every rank sums the 8-KiB chunks it owns -- chunk c belongs to rank c % world --
 * then collects the other owners' sums; 2 x (world-1)/world of the message come in per rank instead of (world-1) x).  Every rank
 * must pass the same value.  sgl_mi355_car_all_reduce == algo 0. */
int sgl_mi355_car_all_reduce_algo(void* inout, int64_t num_elements, int dtype, const void* const* peer_bufs, int rank, int world,
                                  int64_t max_bytes, int algo, void* stream);
/* The all-reduce of a row-parallel linear fused with what follows it on the decode path (linear.py:1302-1303 ->
 * layernorm.py:135-171 -> per_token_quant_fp8.cu): x = all_reduce(partial); residual += x; y = rmsnorm(residual) * weight;
 * optional out_norm (T) and out_q / out_s.  Bit-identical to sgl_mi355_car_all_reduce + sgl_mi355_fused_add_rmsnorm_quant_fp8. */
int sgl_mi355_car_all_reduce_add_rmsnorm_quant(const void* partial, void* residual, const void* weight, float eps,
                                               void* out_norm, void* out_q, float* out_s, int rows, int hidden, int dtype,
                                               const void* const* peer_bufs, int rank, int world, int64_t max_bytes,
                                               void* stream);
/* algo as above; in the two-stage form the owner of a row (row % world) finishes it (sum, add, RMSNorm, quant) once and the
 * other ranks collect the finished row.  hidden % 16 == 0 for the two-stage form (else one-shot); one `hidden` per communicator. */
int sgl_mi355_car_all_reduce_add_rmsnorm_quant_algo(const void* partial, void* residual, const void* weight, float eps,
                                                    void* out_norm, void* out_q, float* out_s, int rows, int hidden, int dtype,
                                                    const void* const* peer_bufs, int rank, int world, int64_t max_bytes,
                                                    int algo, void* stream);
/* The same with this rank's operand given as the split-K slabs of its row-parallel GEMM (raw f32 accumulators [nslabs][rows][hidden]
 * from sgl_mi355_fp8_gemm_slabs / sgl_mi355_skinny_gemm_slabs_min) and the GEMM's scale vectors (NULL: none): the kernel forms
 * x = T((slab 0 + slab 1 + ...) * slab_sx[row] * slab_sw[col]) while it publishes the row, so the GEMM's reduce launch and its
 * [rows, hidden] round trip go away (round 4).  Bit-identical to that launch followed by the entry point above. */
int sgl_mi355_car_all_reduce_add_rmsnorm_quant_slabs(const float* slabs, int nslabs, const float* slab_sx, const float* slab_sw,
                                                     void* residual, const void* weight, float eps, void* out_norm, void* out_q,
                                                     float* out_s, int rows, int hidden, int dtype, const void* const* peer_bufs,
                                                     int rank, int world, int64_t max_bytes, int algo, void* stream);
/* all-gather along the last dimension with the same buffers and protocol (the logits all-gather of a vocab-sharded lm_head,
 * python/sglang/srt/layers/logits_processor.py:471-500): out [rows, world * row_bytes] <- rank r's in [rows, row_bytes] */
int sgl_mi355_car_all_gather(const void* in, void* out, int64_t rows, int64_t row_bytes, const void* const* peer_bufs, int rank,
                             int world, int64_t max_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGL_MI355_H */
