// Launch parameters shared by the extend-attention translation units (extend_attention.hip: the 16x16x32 kernels and the C-ABI
// entry; extend_attention_phased.hip: the phased 32x32x16 kernel).  Field meanings follow extend_attention_fwd
// (python/sglang/srt/layers/attention/triton_ops/extend_attention.py:41-438) and extend_attention_cpu
// (sgl-kernel/csrc/cpu/extend.cpp:579-723).
#pragma once
#include "common.h"

struct ExtendParams {
  const void* q;   // [T, Hq, D]
  const void* ke;  // [T, Hkv, D]
  const void* ve;  // [T, Hkv, D]
  void* o;         // [T, Hq, D]
  int64_t q_stride_t, ke_stride_t, ve_stride_t, o_stride_t;  // elements; heads contiguous (stride D)
  const void* k_buf;
  const void* v_buf;
  int64_t k_stride_t, k_stride_h, v_stride_t, v_stride_h;
  const int32_t* qo_indptr;   // [bs+1]
  const int32_t* kv_indptr;   // [bs+1] prefix lengths cumsum
  const int32_t* kv_indices;  // prefix slots
  // alternative addressing (CPU op schema): prefix = req_to_token[req_pool_indices[b], :seq_lens[b]-ext]
  const int32_t* req_to_token;
  int64_t req_to_token_stride;
  const int64_t* req_pool_indices;
  const int64_t* seq_lens;
  const int32_t* extend_seq_lens;
  const int32_t* extend_start_loc;
  int bs, hq, hkv, group, nqb, bq_log2, hchunks;
  float sm_scale, logit_cap;
  int is_causal;
  int kv_fp8;              // the pool (prefix) rows are e4m3fn bytes; k_extend / v_extend stay in the q dtype
  float k_scale, v_scale;  // K_true = K_fp8 * k_scale, V_true = V_fp8 * v_scale for the prefix part
  // custom mask / sliding window (extend_attention.py:131-203 stage 1, :205-284 stage 2), MASKED instantiations only:
  //   custom_mask u8, request b's block starts at mask_indptr[b] and is [ext_len, pre_len + ext_len] row-major; in the prefix
  //   phase it applies unless skip_prefix_mask; in the extend phase it REPLACES the causal rule (:253-268);
  //   sliding_window W > 0 masks prefix key j (index inside kv_indices) for query row i (index inside the extend part)
  //   unless i <= j + W (:191-196) -- the backend hands in the last W + 1 prefix slots (triton_backend.py:927-955).
  const uint8_t* custom_mask;
  const int64_t* mask_indptr;
  int skip_prefix_mask;
  int sliding_window;
  // Cascade (shared-prefix) decode, PREFIX pass (casc_bs > 0; sgl_mi355_decode_attention_cascade): the casc_bs decode queries
  // q [casc_bs, Hq, D] are the "extend tokens" of bs = prefix splits virtual sequences that all start at query row 0; sequence
  // b attends ONLY to the shared prefix rows kv_indices[b * casc_chunk, min((b + 1) * casc_chunk, casc_prefix_len)) (no extend
  // keys), and instead of o the kernel writes the split partial O = acc / l (f32) and its natural-log LSE to the decode
  // kernel's split slots: part_o [casc_bs][hq][max_kv_splits][D], part_lse [casc_bs][hq][max_kv_splits], slot casc_slot0 + b.
  int casc_bs = 0, casc_prefix_len = 0, casc_chunk = 0, casc_slot0 = 0, max_kv_splits = 0;
  float* part_o = nullptr;
  float* part_lse = nullptr;
  long long* tl = nullptr;   // SGL_EXT_TIMELINE builds only
};

constexpr int kKT = 64;  // kv tokens per tile
constexpr float kLog2e = 1.4426950408889634f;

// extend_attention_phased.hip: the phased 8-wave kernel (16-bit K / V, D = 128, head group <= 8, no mask / cap / cascade / fp8 pool);
// returns SGL_MI355_OK after a launch.  The caller has checked eligibility with extend_phased_eligible.
bool extend_phased_eligible(const ExtendParams& p);
int launch_extend_phased(ExtendParams& p, int max_len_extend, int dtype, hipStream_t st);
