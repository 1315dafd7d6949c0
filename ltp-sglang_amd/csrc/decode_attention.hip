// Split-KV (flash-decoding) GQA decode attention over the paged token_to_kv_pool, gfx950.
//
// Replaces, for the hot path, the reference's decode kernels:
//   python/sglang/srt/layers/attention/triton_ops/decode_attention.py:38-728
//     (_fwd_grouped_kernel_stage1 / _fwd_kernel_stage1 / _fwd_kernel_stage2)
//   sgl-kernel/csrc/cpu/decode.cpp:1375-1575 (decode_attention_cpu: same math, takes
//     req_to_token/req_pool_indices/seq_lens instead of kv_indices)
// and matches the semantics of the oracle torch_native_backend.py:112-180.
//
// Design (MI355X-first, see DESIGN.md "decode attention"):
//  * one workgroup = NW independent wavefronts working on one (request, kv-head, kv-split);
//    every wave owns whole 32-token tiles, a private 2 x 32 x D LDS image and its own
//    online-softmax state, so the main loop has no workgroup barrier at all;
//  * K/V rows are gathered through kv_indices with 16-byte loads in which 16 (D=128)
//    consecutive lanes cover one contiguous 256-byte row: every HBM request is a whole row;
//    the gather is register-staged and issued one tile ahead of the compute (loads of
//    tile i+1 are in flight while tile i is consumed from LDS);
//  * LDS images are XOR-swizzled so both the ds_read_b128 K-fragment reads and the
//    ds_read_b64_tr_b16 transposed V reads are bank-conflict free;
//  * QK^T is computed transposed (S^T = K Q^T) with v_mfma_f32_16x16x32, so that the
//    accumulator (head on lane&15, 4 consecutive tokens in registers) is, after the
//    in-register softmax (two DPP/bpermute xor-shuffles per row reduction), already the
//    B operand of O^T = V^T P^T; all q heads of the GQA group share every K/V byte;
//  * partial results use the reference's own scratch layout (attn_logits f32
//    [bs, Hq, max_splits, Dv] holding acc/l, attn_lse f32 [bs, Hq, max_splits] holding
//    m + log(l)) and stage 2 merges the splits by LSE exactly like _fwd_kernel_stage2.
#include "row_helpers.h"

namespace {

struct DecodeParams {
  const void* q;
  int64_t q_stride_t;  // elements between tokens of q; heads are contiguous [Hq, D]
  const void* k_buf;
  const void* v_buf;
  int64_t k_stride_t, k_stride_h, v_stride_t, v_stride_h;  // elements
  void* o;
  int64_t o_stride_t;
  // index mode A (Triton backend metadata): kv_indptr/kv_indices
  const int32_t* kv_indptr;
  const int32_t* kv_indices;
  // index mode B (CPU op schema): req_to_token[req_pool_indices[b], :seq_lens[b]]
  const int32_t* req_to_token;
  int64_t req_to_token_stride;
  const int64_t* req_pool_indices;
  const int64_t* seq_lens;
  float* attn_logits;
  float* attn_lse;
  const int32_t* num_kv_splits;
  int max_kv_splits;
  int bs, hq, hkv, group, dv;
  float sm_scale, logit_cap;
  int kv_fp8;             // K/V pool rows are e4m3fn bytes (kv_cache_dtype fp8_e4m3, memory_pool.py:385-395), else the q dtype
  float k_scale, v_scale;  // K_true = K_fp8 * k_scale, V_true = V_fp8 * v_scale (radix_attention.py:73-76); 1 for 16-bit pools
  // in-launch stage 2 (MODE 0): the last workgroup of a request to finish merges its splits and quantises the row
  int32_t* merge_cnt;  // [bs], zero on entry, left zero; NULL = off
  void* mq_o;          // optional T [bs, Hq * Dv]
  void* mq_q;          // optional e4m3 [bs, Hq * Dv] with mq_s f32 [bs]
  float* mq_s;
  // Cascade shared-prefix decode: all `bs` requests share the first prefix_len slots of their sequences.
  //   prefix pass  a launch of the EXTEND kernel (extend_attention.hip, ExtendParams::casc_*): the bs decode queries are the
  //                query block, the shared rows the keys, staged once per (kv head, 32 requests, split) in LDS for all 4 x 32
  //                (head, request) pairs of a workgroup; prefix_splits split partials (O = acc / l, LSE) per (request, head)
  //                go to the TOP split slots [max_kv_splits - prefix_splits, max_kv_splits).  (Round 2 first ran the prefix as
  //                16-column chunks of this kernel -- every chunk re-read the rows, 19.6 us at 64 x 1536 -- and then had the
  //                merging workgroup of every request read prefix + suffix partials, +2 us per prefix split.)
  //   suffix pass  (CASC = 2, prefix_splits > 0): the ordinary kernel over every request's private slots, except that split 0
  //                of every (request, kv head) CONTINUES the online softmax from the prefix state (the LSE-weighted sum of the
  //                prefix partials is its initial (m, l, acc) -- the math of merge_state, merge_attn_states.cu, applied before
  //                instead of after), so the in-launch merge sees suffix slots only.
  int prefix_len, prefix_splits;
  // Flat unit list (round 5, SCHED kernels; csrc/kv_index.hip decode_schedule_kernel): the (request, split) units of the launch as a
  // list sorted by length, longest first -- grid y walks the list, so the dispatcher starts the long units first and fills the
  // chip's slots with short ones as they free up (ragged batches), and the grid carries no never-live workgroups.
  // sched: {T, units, total tokens, capacity}, then per unit {request, split | splits << 16, kv_indptr[request], its length}.
  // sched_chunks = grid y = the capacity the host sized the list for.
  const int32_t* sched;
  int sched_chunks;
#ifdef SGL_DEC_TIMELINE
  long long* tl;   // tools/debug/dec_timeline.py: s_memrealtime stamps [workgroup (linear)][wave][8]
#endif
};

#ifdef SGL_DEC_TIMELINE
long long* g_dec_tl = nullptr;
#define DEC_STAMP(k)                                                                                                   \
  do {                                                                                                                 \
    const long long t_ = (long long)__builtin_amdgcn_s_memrealtime();                                                  \
    if (p.tl && lane == 0)                                                                                             \
      p.tl[((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + w) * 8 + (k)] = t_;        \
  } while (0)
#else
#define DEC_STAMP(k) do { } while (0)
#endif

constexpr int kTile = 32;  // tokens per wave tile == reference _MIN_BLOCK_KV (decode_attention.py:35)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ void request_range(const DecodeParams& p, int b, const int32_t*& idx_row, int& seq_len) {
  if (p.kv_indptr != nullptr) {
    const int s0 = p.kv_indptr[b];
    seq_len = p.kv_indptr[b + 1] - s0;
    idx_row = p.kv_indices + s0;
  } else {
    seq_len = (int)p.seq_lens[b];
    idx_row = p.req_to_token + p.req_pool_indices[b] * p.req_to_token_stride;
  }
}

__device__ __forceinline__ int split_len(int seq_len, int nsplit) {
  // decode_attention.py:90-94: ceil(ceil(seq/splits)/MIN_BLOCK_KV)*MIN_BLOCK_KV
  const int per = (seq_len + nsplit - 1) / nsplit;
  return (per + kTile - 1) / kTile * kTile;
}

__device__ __forceinline__ float softcap_log2(float s_scaled, float cap) {
  // cap * tanh(x / cap), tanh(y) = 2*sigmoid(2y) - 1 (decode_attention.py:38-41)
  const float y = s_scaled / cap;
  const float t = 2.0f / (1.0f + __expf(-2.0f * y)) - 1.0f;
  return cap * t * kLog2e;
}

// Memory-model note: the hand-off below is the first row of the "hand-offs measured with sc1 loads / one agent-scope atomic
// add per storing workgroup" table of the CDNA4 guide (MI355X_MICROARCH.md, Workgroup dispatch ... inter-workgroup visibility)
// with the consumer's acquire KEPT (two workgroups per CU are resident, outside that table's one-per-CU cell): every byte is
// stored sc1 (write-through), every storing wave drains vmcnt before the workgroup barrier, ONE lane then adds to the counter,
// and the workgroup whose add returned total - 1 runs ONE agent-scope acquire before any wave loads the partials.  The counters
// are re-zeroed by the host side once per decode step (HipAttnBackend._decode_metadata), so a stale count cannot persist.
// In-launch stage 2 (cross-workgroup hand-off in its counter form): every workgroup of request b publishes its split
// partial -- WRITE-THROUGH (sc1) stores, so no release fence (a release per workgroup writes back the XCD's L2 and made the
// launch 35 us slower); every wave drains vmcnt, barrier, then lane 0 draws a ticket -- and the workgroup that draws the
// last ticket acquires (agent scope: ONE cache invalidate per request), merges all splits of the request, rounds, optionally
// quantises per token (the op sequence stage 2 -> sgl_per_token_quant_fp8) and puts the counter back to zero.  Correct for
// any placement of the request's workgroups over CUs / XCDs; workgroup-scope fences or an L1-only invalidate would not be.
template <typename T, bool CASC = false>
__device__ __forceinline__ void arrive_and_merge(const DecodeParams& p, int b, int seq_len, int nsplit, int hchunks, char* smem) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* flag = (int*)smem;  // (the LDS of the cross-wave merge is free again)
  if (threadIdx.x == 0) {
    const int total = p.hkv * hchunks * nsplit;
    const int old = __hip_atomic_fetch_add(p.merge_cnt + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = (old == total - 1) ? 1 : 0;
  }
  __syncthreads();
  if (*flag == 0) return;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  float* red = (float*)smem + 4;
  if (p.hq * p.dv <= 256 * 8 * 4)
    merge_quant_row<T, 4>(b, p.attn_logits, p.attn_lse, seq_len, nsplit, p.max_kv_splits, p.hq, p.dv, (T*)p.mq_o, (uint8_t*)p.mq_q, p.mq_s, red, CASC);
  else
    merge_quant_row<T, 8>(b, p.attn_logits, p.attn_lse, seq_len, nsplit, p.max_kv_splits, p.hq, p.dv, (T*)p.mq_o, (uint8_t*)p.mq_q, p.mq_s, red, CASC);
  if (threadIdx.x == 0) __hip_atomic_store(p.merge_cnt + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 16 e4m3 bytes -> 16 T (exact: every e4m3 value is representable in bf16 and f16)
template <typename T>
__device__ __forceinline__ void cvt16_fp8(const u32x4_t& in, u32x4_t& lo, u32x4_t& hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  struct P2 { T a, b; };
  uint32_t out[8];
#pragma unroll
  for (int w4 = 0; w4 < 4; ++w4) {
    const f32x2 f01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)in[w4], false);
    const f32x2 f23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)in[w4], true);
    out[2 * w4] = __builtin_bit_cast(uint32_t, P2{(T)f01[0], (T)f01[1]});
    out[2 * w4 + 1] = __builtin_bit_cast(uint32_t, P2{(T)f23[0], (T)f23[1]});
  }
  lo = u32x4_t{out[0], out[1], out[2], out[3]};
  hi = u32x4_t{out[4], out[5], out[6], out[7]};
}

// MODE 0: the NW waves of a workgroup share one (request, kv head, split) and interleave its tiles; LDS merge.
// MODE 1: every wave is its own (request, kv head, split) unit; the NW waves of a workgroup are NW consecutive
//         units of one request with the kv head fastest, so they read ADJACENT 256-byte pieces of the same token rows
//         at about the same time (whole 2-KiB token rows per workgroup instead of scattered 256-byte pieces) and no
//         barrier or cross-wave merge exists at all.
// KVB: bytes per pool element (1 = e4m3 KV cache, converted on the way into LDS).  CASC = 2: the SUFFIX pass of the cascade
// (shared-prefix) decode, a form of MODE 0 whose split 0 starts from the prefix state (DecodeParams::prefix_*) -- its own
// instantiation, so the extra code stays out of the ordinary one.
// decode_piece: one (request, kv head x head chunk, split [start, end)) by the NW waves of a workgroup (MODE 0) or by one wave (MODE 1).
template <typename T, int D, int NW, int MODE, int KVB, int CASC, bool SCHED>
__device__ __forceinline__ void decode_piece(const DecodeParams& p, char* smem, const int b, const int split, const int nsplit,
                                             const int seq_len, const int32_t* idx_row, const int start, const int end,
                                             const bool continues_prefix, const int kh, const int hc, const int nh, const int hchunks,
                                             const int tid) {
  using Tr = ElemTraits<T>;
  using vec8 = typename Tr::vec8;
  constexpr int ROWB = D * 2;
  constexpr int LPR = ROWB / 16;   // lanes covering one K/V row with 16-byte loads
  constexpr int RPI = 64 / LPR;    // rows per wave-wide load instruction
  constexpr int NI = kTile / RPI;  // load instructions per tile
  constexpr int GLPR = D * KVB / 16;  // the same three for the rows as they lie in the pool
  constexpr int GRPI = 64 / GLPR;
  constexpr int GNI = kTile / GRPI;
  constexpr int KS = D / 32;       // QK^T k-steps
  constexpr int NT = D / 16;       // PV output column tiles
  constexpr int TILE_B = kTile * ROWB;
  constexpr int RPB = (ROWB >= 256) ? 1 : 256 / ROWB;  // rows per 256-B LDS bank row
  constexpr int KMASK = (LPR < 16 ? LPR : 16) - 1;
  constexpr int VCH = ROWB / 32;
  constexpr int VMASK = (VCH < 8 ? VCH : 8) - 1;

  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = lane & 15;  // MFMA n index: q head within the chunk
  const int g = lane >> 4;  // MFMA k/m group
  const int c16 = lane % GLPR;  // 16-byte chunk of the pool row / row within a load instruction, of this lane
  const int rsub = lane / GLPR;

  char* kl = smem + w * (2 * TILE_B);
  char* vl = kl + TILE_B;

  // Q fragments: B operand of S^T = K Q^T, lane (a,g) holds Q[h0+a][32*ks + 8*g .. +8]
  vec8 qf[KS];
  auto load_q = [&](int bb, vec8 (&dst)[KS]) {
    const int va = hc * 16 + min(a, nh - 1);
    const T* qrow = (const T*)p.q + (int64_t)bb * p.q_stride_t + (int64_t)(kh * p.group + va) * D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (a < nh) {
        dst[ks] = *(const vec8*)(qrow + 32 * ks + 8 * g);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[ks][j] = (T)0.0f;
      }
    }
  };
  load_q(b, qf);

  const char* kbase = (const char*)p.k_buf + ((int64_t)kh * p.k_stride_h) * KVB + c16 * 16;
  const char* vbase = (const char*)p.v_buf + ((int64_t)kh * p.v_stride_h) * KVB + c16 * 16;
  const int64_t kst = p.k_stride_t * KVB, vst = p.v_stride_t * KVB;

  const int ntiles = max(0, end - start + kTile - 1) / kTile;
  const float sm_scale = p.sm_scale * p.k_scale;
  const float scale_log2 = sm_scale * kLog2e;
  const bool use_cap = p.logit_cap > 0.0f;

  float m_i = -INFINITY, l_i = 0.0f;
  f32x4_t acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if constexpr (CASC == 2) {
    // Split 0 of the private part continues from the shared prefix: wave 0 starts with the LSE-weighted sum of the prefix
    // split partials (top slots of this (request, head)) as its (m, l, acc) -- m in log2 units, l on the g = 0 lane (the four
    // g-lanes of a head are summed at the end), acc before the v_scale the partials already carry.
    if (continues_prefix && w == 0 && a < nh) {
      const int64_t slot0 = ((int64_t)b * p.hq + (kh * p.group + hc * 16 + a)) * p.max_kv_splits + (p.max_kv_splits - p.prefix_splits);
      float M = -INFINITY;
      for (int sI = 0; sI < p.prefix_splits; ++sI) M = fmaxf(M, p.attn_lse[slot0 + sI]);
      float L = 0.0f;
      for (int sI = 0; sI < p.prefix_splits; ++sI) {
        const float wgt = __expf(p.attn_lse[slot0 + sI] - M);
        L += wgt;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const f32x4_t o = *(const f32x4_t*)(p.attn_logits + (slot0 + sI) * D + 16 * n + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[n][r] = fmaf(o[r], wgt, acc[n][r]);
        }
      }
      const float inv_vs = 1.0f / p.v_scale;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[n][r] *= inv_vs;
      m_i = M * kLog2e;
      l_i = g == 0 ? L : 0.0f;
    }
  }

  u32x4_t kreg[GNI], vreg[GNI];

  auto load_idx_of = [&](const int32_t* ir, int st, int en, int nt, int t) -> int {
    const int tok = st + t * kTile + (lane & 31);
    return (t < nt && tok < en) ? ir[tok] : 0;
  };
  auto load_idx = [&](int t) -> int { return load_idx_of(idx_row, start, end, ntiles, t); };
  auto issue_to = [&](u32x4_t (&kd)[GNI], u32x4_t (&vd)[GNI], int idxreg) {
#pragma unroll
    for (int i = 0; i < GNI; ++i) {
      const int id = __shfl(idxreg, i * GRPI + rsub, WAVE);
      // NON-TEMPORAL loads: every K/V row is read once per decode step, so keeping it in L2 / the Infinity Cache only
      // evicts what the step re-reads (activations, split partials) and lengthens the miss path.  Same-box A/B at the
      // BASELINE shape (round 2): stage 1 55.0 -> 50.3 us per launch (4.9 -> 5.4 TB/s), decode step 4.55 -> 4.39 ms.
      kd[i] = __builtin_nontemporal_load((const u32x4_t*)(kbase + (int64_t)id * kst));
      vd[i] = __builtin_nontemporal_load((const u32x4_t*)(vbase + (int64_t)id * vst));
    }
  };
  auto issue = [&](int idxreg) { issue_to(kreg, vreg, idxreg); };

  constexpr int TS = (MODE == 0) ? NW : 1;  // tile stride of this wave
  int tile = (MODE == 0) ? w : 0;
  int idx_next = 0;
  if (tile < ntiles) {
    // (Round 4 tried the first tile's 32 slot ids by SCALAR loads -- s_load_dwordx16 x 2, a select chain per load instruction -- on
    // the theory that the younger workgroup's index load waits behind the older one's 64 KiB of K / V loads in the CU's in-order
    // vector queue: the timeline moved by 0.3 us and the step not at all (profiles/round4_ab_attn_timeline.json).  Removed.)
    issue(load_idx(tile));
    idx_next = load_idx(tile + TS);
  }
  DEC_STAMP(2);
#ifdef SGL_DEC_TIMELINE
  bool first_tile = true;
#endif

  for (; tile < ntiles; tile += TS) {
    const int tok0 = start + tile * kTile;
    // ---- staged registers -> swizzled LDS images (wave private, no barrier) ----
#pragma unroll
    for (int i = 0; i < GNI; ++i) {
      const int row = i * GRPI + rsub;
      const int fk = (row / RPB) & KMASK;
      const int fv = (row / RPB) & VMASK;
      const bool dead = tok0 + row >= end;  // 0 * garbage must stay 0
      if constexpr (KVB == 2) {
        *(u32x4_t*)(kl + row * ROWB + ((c16 ^ fk) << 4)) = kreg[i];
        u32x4_t vv = vreg[i];
        if (dead) vv = u32x4_t{0u, 0u, 0u, 0u};
        *(u32x4_t*)(vl + row * ROWB + ((((c16 >> 1) ^ fv) << 5) | ((c16 & 1) << 4))) = vv;
      } else {  // 16 e4m3 bytes = the 16-bit chunks 2 c16 and 2 c16 + 1 of the row
        u32x4_t k0, k1, v0, v1;
        cvt16_fp8<T>(kreg[i], k0, k1);
        cvt16_fp8<T>(vreg[i], v0, v1);
        if (dead) v0 = v1 = u32x4_t{0u, 0u, 0u, 0u};
        *(u32x4_t*)(kl + row * ROWB + (((2 * c16) ^ fk) << 4)) = k0;
        *(u32x4_t*)(kl + row * ROWB + (((2 * c16 + 1) ^ fk) << 4)) = k1;
        *(u32x4_t*)(vl + row * ROWB + ((c16 ^ fv) << 5)) = v0;
        *(u32x4_t*)(vl + row * ROWB + (((c16 ^ fv) << 5) | 16)) = v1;
      }
    }
#ifdef SGL_DEC_TIMELINE
    if (first_tile) { DEC_STAMP(3); first_tile = false; }
#endif
    // ---- issue the gather of this wave's next tile, prefetch indices two tiles ahead ----
    if (tile + TS < ntiles) issue(idx_next);
    idx_next = load_idx(tile + 2 * TS);

    // ---- S^T = K Q^T for two 16-token halves ----
    f32x4_t s[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      s[tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const int row = 16 * tt + a;
      const int fk = (row / RPB) & KMASK;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int chunk = 4 * ks + g;
        const vec8 kf = *(const vec8*)(kl + row * ROWB + ((chunk ^ fk) << 4));
        s[tt] = Tr::mfma16(kf, qf[ks], s[tt]);
      }
    }
    // ---- online softmax; lane (a,g) holds head a, tokens 16*tt + 4*g + r.  One wave-uniform branch for the logit cap (tested
    // per score it cut the loop body into a dozen basic blocks); the score scale is folded into the exponent's fma. ----
    float x[2][4];
    if (use_cap) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[tt][r] = softcap_log2(s[tt][r] * sm_scale, p.logit_cap);
    } else {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[tt][r] = s[tt][r];
    }
    const float cs = use_cap ? 1.0f : scale_log2;  // the exponent is x * cs - m (cs > 0)
    float mt = -INFINITY;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tok = tok0 + 16 * tt + 4 * g + r;
        x[tt][r] = (tok < end) ? x[tt][r] : -INFINITY;
        mt = fmaxf(mt, x[tt][r]);
      }
    }
    mt = pair32_max(pair16_max(mt));   // over the four lane groups of a head (VALU lane swaps, no LDS round trip)
    const float m_new = fmaxf(m_i, mt * cs);  // finite: every tile holds >= 1 valid token
    const float alpha = __builtin_amdgcn_exp2f(m_i - m_new);
    float lsum = 0.0f;
    vec8 pf;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(x[tt][r], cs, -m_new));
        lsum += pv;
        pf[4 * tt + r] = Tr::from_f32(pv);  // P is rounded to the V dtype before PV (decode_attention.py:373)
      }
    }
    l_i = l_i * alpha + lsum;  // per-lane partial; the 4 g-lanes of a head are summed once at the end
    m_i = m_new;
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      acc[n] *= alpha;
      s16x4_t t0, t1;
      {
        const int row = 4 * g + (a >> 2);
        const int fv = (row / RPB) & VMASK;
        t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4_t __attribute__((address_space(3)))*)(vl + row * ROWB + ((n ^ fv) << 5) + ((a & 3) << 3)));
      }
      {
        const int row = 16 + 4 * g + (a >> 2);
        const int fv = (row / RPB) & VMASK;
        t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4_t __attribute__((address_space(3)))*)(vl + row * ROWB + ((n ^ fv) << 5) + ((a & 3) << 3)));
      }
      const s16x8_t t01 = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
      acc[n] = Tr::mfma16(__builtin_bit_cast(vec8, t01), pf, acc[n]);
    }
  }

  DEC_STAMP(4);
  // ---- merge the NW wave-private states, write the split partial ----
  l_i = pair32_sum(pair16_sum(l_i));
  if constexpr (MODE == 1) {
    // the wave IS the split: write acc / l and m + log(l) straight from the accumulator layout
    if (a < nh) {
      const int64_t slot = ((int64_t)b * p.hq + (kh * p.group + hc * 16 + a)) * p.max_kv_splits + split;
      const float inv = p.v_scale / l_i;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        f32x4_t o = acc[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] *= inv;
        *(f32x4_t*)(p.attn_logits + slot * D + 16 * n + 4 * g) = o;
      }
      if (g == 0) p.attn_lse[slot] = m_i * kLn2 + __logf(l_i);
    }
    return;
  }
  __syncthreads();
  float* red_m = (float*)smem;       // [NW][16]
  float* red_l = red_m + NW * 16;    // [NW][16]
  float* red_acc = red_l + NW * 16;  // [NW][16][D]
  if (a < nh) {
    if (g == 0) {
      red_m[w * 16 + a] = m_i;
      red_l[w * 16 + a] = l_i;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) *(f32x4_t*)(red_acc + (w * 16 + a) * D + 16 * n + 4 * g) = acc[n];
  }
  __syncthreads();
  for (int out = tid; out < nh * D; out += NW * 64) {
    const int h = out / D, d = out - h * D;
    float M = -INFINITY;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) M = fmaxf(M, red_m[ww * 16 + h]);
    float L = 0.f, val = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
      const float sc = __builtin_amdgcn_exp2f(red_m[ww * 16 + h] - M);
      L += red_l[ww * 16 + h] * sc;
      val += red_acc[(ww * 16 + h) * D + d] * sc;
    }
    const int vh = hc * 16 + h;
    const int64_t slot = ((int64_t)b * p.hq + (kh * p.group + vh)) * p.max_kv_splits + split;
    const float ov = val / L * p.v_scale, lv = M * kLn2 + __logf(L);
    if (p.merge_cnt) {  // in-launch merge: the partial is PUBLISHED -- write-through (sc1) stores, no release fence needed
      __hip_atomic_store(p.attn_logits + slot * D + d, ov, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == 0) __hip_atomic_store(p.attn_lse + slot, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      p.attn_logits[slot * D + d] = ov;
      if (d == 0) p.attn_lse[slot] = lv;
    }
  }
  DEC_STAMP(5);
  if (p.merge_cnt) arrive_and_merge<T, CASC == 2>(p, b, seq_len, nsplit, hchunks, smem);
  DEC_STAMP(6);
}


// SCHED (a MODE 0 form, round 5): the (request, split) unit of the workgroup comes from the sorted unit list (DecodeParams::sched)
// instead of (blockIdx.y, blockIdx.z) -- the same splits of the same requests, in another dispatch order and without the grid's
// never-live part; everything after the mapping is the unscheduled kernel.
template <typename T, int D, int NW, int MODE, int KVB = 2, int CASC = 0, bool SCHED = false>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 1 : 2) void decode_attn_stage1(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  static_assert(CASC == 0 || (CASC == 2 && MODE == 0), "the cascade suffix pass is a MODE 0 form");
  static_assert(!SCHED || (MODE == 0 && CASC == 0), "the scheduled form is a plain MODE 0 form");
  DEC_STAMP(0);
  const int hchunks = (p.group + 15) >> 4;
  if constexpr (SCHED) {
    const int khc = blockIdx.x;
    const int kh = khc / hchunks;
    const int hc = khc - kh * hchunks;
    const int nh = min(16, p.group - hc * 16);
    if ((int)blockIdx.y >= p.sched[1]) return;
    const int4 ent = *(const int4*)(p.sched + 4 + 4 * blockIdx.y);   // {request, split | splits << 16, kv_indptr[request], its length}
    const int b = ent.x, split = ent.y & 0xffff, nsplit = ent.y >> 16, seq_len = ent.w;
    DEC_STAMP(1);
    const int per = split_len(seq_len, nsplit);
    const int start = split * per;
    const int end = min(start + per, seq_len);
    if (start >= end) {   // (a request without keys, or a split past its end: counts as arrived)
      if (p.merge_cnt) arrive_and_merge<T, false>(p, b, seq_len, nsplit, hchunks, smem);
      return;
    }
    decode_piece<T, D, NW, MODE, KVB, CASC, true>(p, smem, b, split, nsplit, seq_len, p.kv_indices + ent.z, start, end, false, kh, hc, nh, hchunks, tid);
    return;
  } else {
  int khc, split;
  if constexpr (MODE == 0) {
    khc = blockIdx.x;
    split = blockIdx.z;  // slowest grid dimension: workgroups of splits a request does not use are dispatched LAST
  } else {
    const int unit = blockIdx.x * NW + w;  // wave-uniform
    khc = unit % (p.hkv * hchunks);
    split = unit / (p.hkv * hchunks);
  }
  const int kh = khc / hchunks;
  const int hc = khc - kh * hchunks;
  const int b = blockIdx.y;
  const int nh = min(16, p.group - hc * 16);

  const int32_t* idx_row;
  int seq_len;
  request_range(p, b, idx_row, seq_len);
  const int nsplit = max(1, min(p.num_kv_splits[b], p.max_kv_splits - (CASC == 2 ? p.prefix_splits : 0)));
  if (split >= nsplit) return;  // MODE 1: a whole-wave exit; the kernel has no barrier in that mode
  DEC_STAMP(1);
  const int per = split_len(seq_len, nsplit);
  const int start = split * per;
  const int end = min(start + per, seq_len);
  const bool continues_prefix = CASC == 2 && split == 0;   // carries the prefix state even if the private part is empty
  if (start >= end && !continues_prefix) {
    if constexpr (MODE == 0) {  // an empty split still counts as arrived for the in-launch merge (uniform per workgroup)
      if (p.merge_cnt) arrive_and_merge<T, CASC == 2>(p, b, seq_len, nsplit, hchunks, smem);
    }
    return;
  }
  decode_piece<T, D, NW, MODE, KVB, CASC, false>(p, smem, b, split, nsplit, seq_len, idx_row, start, end, continues_prefix, kh, hc, nh, hchunks, tid);
  }
}

// Any-head-dim fallback (D, Dv <= 256, not multiples of 32 allowed): one wave per
// (request, q head, split).  Correctness path for the reference's odd test shapes
// (test_triton_attention_kernels.py: D in {96, 80, 13}); not a performance path.
template <typename T>
__global__ __launch_bounds__(64) void decode_attn_stage1_generic(const DecodeParams p, int d_qk) {
  __shared__ float p_lds[64];
  __shared__ int id_lds[64];
  const int h = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
  const int kh = h / p.group;
  const int lane = threadIdx.x;
  const int32_t* idx_row;
  int seq_len;
  request_range(p, b, idx_row, seq_len);
  int nsplit = max(1, min(p.num_kv_splits[b], p.max_kv_splits));
  if (split >= nsplit) return;
  const int per = split_len(seq_len, nsplit);
  const int start = split * per, end = min(start + per, seq_len);
  if (start >= end) return;
  const T* qrow = (const T*)p.q + (int64_t)b * p.q_stride_t + (int64_t)h * d_qk;
  const int dv = p.dv;
  float accv[4] = {0.f, 0.f, 0.f, 0.f};  // dv <= 256: lane owns columns lane, lane+64, ...
  float m_i = -INFINITY, l_i = 0.f;
  for (int t0 = start; t0 < end; t0 += 64) {
    const int tok = t0 + lane;
    const bool valid = tok < end;
    const int id = valid ? idx_row[tok] : 0;
    float sdot = 0.f;
    if (valid) {
      const T* krow = (const T*)p.k_buf + (int64_t)id * p.k_stride_t + (int64_t)kh * p.k_stride_h;
      for (int d = 0; d < d_qk; ++d) sdot += (float)qrow[d] * (float)krow[d];
    }
    float xv = sdot * p.sm_scale;
    xv = (p.logit_cap > 0.f) ? softcap_log2(xv, p.logit_cap) : xv * kLog2e;
    xv = valid ? xv : -INFINITY;
    const float m_new = fmaxf(m_i, wave_reduce_max(xv));
    const float alpha = __builtin_amdgcn_exp2f(m_i - m_new);
    const float pv = __builtin_amdgcn_exp2f(xv - m_new);
    l_i = l_i * alpha + wave_reduce_sum(pv);
    m_i = m_new;
    __syncthreads();
    p_lds[lane] = (float)(T)pv;
    id_lds[lane] = id;
    __syncthreads();
    const int nt = min(64, end - t0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int d = lane + 64 * c;
      float av = accv[c] * alpha;
      if (d < dv) {
        for (int t = 0; t < nt; ++t) {
          const T* vrow = (const T*)p.v_buf + (int64_t)id_lds[t] * p.v_stride_t + (int64_t)kh * p.v_stride_h;
          av += p_lds[t] * (float)vrow[d];
        }
      }
      accv[c] = av;
    }
  }
  const int64_t slot = ((int64_t)b * p.hq + h) * p.max_kv_splits + split;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int d = lane + 64 * c;
    if (d < dv) p.attn_logits[slot * dv + d] = accv[c] / l_i;
  }
  if (lane == 0) p.attn_lse[slot] = m_i * kLn2 + __logf(l_i);
}

// Stage 2: LSE merge of the split partials (decode_attention.py:492-552).
template <typename T>
__global__ __launch_bounds__(128) void decode_attn_stage2(const DecodeParams p) {
  const int b = blockIdx.x, h = blockIdx.y;
  const int32_t* idx_row;
  int seq_len;
  request_range(p, b, idx_row, seq_len);
  const int nsplit = max(1, min(p.num_kv_splits[b], p.max_kv_splits));
  const int per = split_len(seq_len, nsplit);
  const int dv = p.dv;
  const int64_t slot0 = ((int64_t)b * p.hq + h) * p.max_kv_splits;
  for (int d = threadIdx.x; d < dv; d += blockDim.x) {
    LseMerge mg;
    float accv = 0.f;
    for (int s = 0; s < nsplit; ++s) {
      if (s * per < seq_len) {
        mg.begin(p.attn_lse[slot0 + s]);
        accv = mg.acc(accv, p.attn_logits[(slot0 + s) * dv + d]);
      }
    }
    const float r = mg.finish(accv);
    ((T*)p.o)[(int64_t)b * p.o_stride_t + (int64_t)h * dv + d] = ElemTraits<T>::from_f32(r);
  }
}

// 0 = workgroup-shared split, four waves per workgroup -- or EIGHT (one 512-thread workgroup per CU, round 4) when the launch has
//     at most one (request, kv head, head chunk) unit per TWO CUs: half as many splits fill the chip, so the in-launch merge reads
//     half as many partials (sgl_mi355_decode_metadata's balance rule makes the same test and sizes the splits for one workgroup
//     per CU).  Measured on whole decode steps (profiles/round4_ab_attn_eight_waves.json): batch 8 / 16 -0.6 / -0.8 %; with one
//     unit per CU (batch 32: every request ONE split) +1.6 %, at batch 24 (1.5 rounds of 512-thread workgroups) +1.4 % -- hence
//     the factor two;
// 1 = wave-per-unit (faster for 1 kv head per rank); 2 / 3 = measurement hooks: always four / always eight waves
int g_decode_mode = 0;
int decode_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return cus;
}

// The unit list's geometry, shared with csrc/kv_index.hip (sgl_mi355_decode_schedule): eight waves (one workgroup per CU) when the
// launch has at most one (request, kv head x head chunk) unit per two CUs, as the unscheduled form decides; target = the units per
// (kv head x head chunk) that fill the chip's resident workgroups `rounds_pct` / 100 times; the list holds at most target + batch
// units (every request's last split is a remainder), the grid's y.
bool decode_sched_eight_waves(int bs, int khc) { return g_decode_mode == 3 || (g_decode_mode != 2 && 2ll * khc * bs <= decode_cus()); }
int decode_sched_target(int bs, int khc, int rounds_pct) {
  const long long slots = (decode_sched_eight_waves(bs, khc) ? 1 : 2) * (long long)decode_cus();
  const long long t = (slots * rounds_pct / 100 + khc - 1) / khc;
  return (int)(t < 1 ? 1 : t);
}

template <typename T, int D, int KVB>
int launch_mfma(const DecodeParams& p, hipStream_t st) {
  const int hchunks = (p.group + 15) / 16;
  if (p.prefix_splits > 0) {   // cascade suffix pass: MODE 0 only (checked by the entry point)
    constexpr int NW = 4;
    constexpr int smem = NW * 2 * kTile * D * 2;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)decode_attn_stage1<T, D, NW, 0, KVB, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      attr_set = true;
    }
    dim3 grid(p.hkv * hchunks, p.bs, p.max_kv_splits - p.prefix_splits);
    hipLaunchKernelGGL((decode_attn_stage1<T, D, NW, 0, KVB, 2>), grid, dim3(NW * 64), smem, st, p);
  } else if (p.sched != nullptr) {   // equal-chunk schedule: the same four / eight wave rule, grid y = the schedule's chunk capacity
    const dim3 grid(p.hkv * hchunks, p.sched_chunks, 1);
    if (decode_sched_eight_waves(p.bs, p.hkv * hchunks)) {
      constexpr int NW = 8;
      constexpr int smem = NW * 2 * kTile * D * 2;
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)decode_attn_stage1<T, D, NW, 0, KVB, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
      }
      hipLaunchKernelGGL((decode_attn_stage1<T, D, NW, 0, KVB, 0, true>), grid, dim3(NW * 64), smem, st, p);
    } else {
      constexpr int NW = 4;
      constexpr int smem = NW * 2 * kTile * D * 2;
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)decode_attn_stage1<T, D, NW, 0, KVB, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
      }
      hipLaunchKernelGGL((decode_attn_stage1<T, D, NW, 0, KVB, 0, true>), grid, dim3(NW * 64), smem, st, p);
    }
  } else if (g_decode_mode == 3 || (g_decode_mode == 0 && 2ll * p.hkv * hchunks * p.bs <= decode_cus())) {
    constexpr int NW = 8;
    constexpr int smem = NW * 2 * kTile * D * 2;   // 128 KiB at D = 128: one workgroup per CU
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)decode_attn_stage1<T, D, NW, 0, KVB>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      attr_set = true;
    }
    dim3 grid(p.hkv * hchunks, p.bs, p.max_kv_splits);
    hipLaunchKernelGGL((decode_attn_stage1<T, D, NW, 0, KVB>), grid, dim3(NW * 64), smem, st, p);
  } else if (g_decode_mode != 1) {
    constexpr int NW = 4;
    constexpr int smem = NW * 2 * kTile * D * 2;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)decode_attn_stage1<T, D, NW, 0, KVB>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      attr_set = true;
    }
    // (kv head, request, split): with the split index slowest, the empty workgroups of unused splits sit at the end of
    // the dispatch order instead of being interleaved with the real ones (bs 128 x 1 split: 417 us -> 201 us)
    dim3 grid(p.hkv * hchunks, p.bs, p.max_kv_splits);
    hipLaunchKernelGGL((decode_attn_stage1<T, D, NW, 0, KVB>), grid, dim3(NW * 64), smem, st, p);
  } else {
    constexpr int NW = 8;
    constexpr int smem = NW * 2 * kTile * D * 2;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)decode_attn_stage1<T, D, NW, 1, KVB>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      attr_set = true;
    }
    const int units = p.hkv * hchunks * p.max_kv_splits;
    dim3 grid((units + NW - 1) / NW, p.bs, 1);
    hipLaunchKernelGGL((decode_attn_stage1<T, D, NW, 1, KVB>), grid, dim3(NW * 64), smem, st, p);
  }
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

template <typename T>
int launch_all(const DecodeParams& p, int d_qk, hipStream_t st) {
  int rc;
  if (d_qk == p.dv && d_qk == 128) {
    rc = p.kv_fp8 ? launch_mfma<T, 128, 1>(p, st) : launch_mfma<T, 128, 2>(p, st);
  } else if (d_qk == p.dv && d_qk == 64) {
    rc = p.kv_fp8 ? launch_mfma<T, 64, 1>(p, st) : launch_mfma<T, 64, 2>(p, st);
  } else {
    dim3 grid(p.hq, p.max_kv_splits, p.bs);
    hipLaunchKernelGGL((decode_attn_stage1_generic<T>), grid, dim3(64), 0, st, p, d_qk);
    SGL_HIP_LAUNCH_CHECK();
    rc = SGL_MI355_OK;
  }
  if (rc != SGL_MI355_OK) return rc;
  if (p.o == nullptr) return SGL_MI355_OK;  // partials only: the caller merges (sgl_mi355_decode_merge_quant_fp8)
  hipLaunchKernelGGL((decode_attn_stage2<T>), dim3(p.bs, p.hq), dim3(128), 0, st, p);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

}  // namespace

#ifdef SGL_DEC_TIMELINE
extern "C" int sgl_mi355_decode_attention_debug_timeline(long long* buf) {
  g_dec_tl = buf;
  return SGL_MI355_OK;
}
#endif

extern "C" int sgl_mi355_decode_attention_set_mode(int mode) {
  g_decode_mode = (mode >= 0 && mode <= 3) ? mode : 0;
  return SGL_MI355_OK;
}

static int decode_entry(
    const void* q, int64_t q_stride_t, const void* k_buffer, const void* v_buffer, int64_t k_stride_t,
    int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h, void* o, int64_t o_stride_t, const int32_t* kv_indptr,
    const int32_t* kv_indices, const int32_t* req_to_token, int64_t req_to_token_stride, const int64_t* req_pool_indices,
    const int64_t* seq_lens, float* attn_logits, float* attn_lse, const int32_t* num_kv_splits, int max_kv_splits,
    int batch, int num_q_heads, int num_kv_heads, int head_dim, int v_head_dim, float sm_scale, float logit_cap,
    int dtype, int kv_dtype, float k_scale, float v_scale, int32_t* merge_cnt, void* mq_o, void* mq_q, float* mq_s,
    void* stream, int prefix_len = 0, int prefix_splits = 0, const int32_t* sched = nullptr, int sched_capacity = 0) {
  SGL_CHECK(batch >= 0, "decode_attention: negative batch %d", batch);
  if (batch == 0) return SGL_MI355_OK;
  SGL_CHECK(q && k_buffer && v_buffer, "decode_attention: null tensor pointer");  // o == NULL: split partials only
  SGL_CHECK(attn_logits && attn_lse && num_kv_splits, "decode_attention: null split scratch pointer");
  SGL_CHECK(kv_indptr || (req_to_token && req_pool_indices && seq_lens),
            "decode_attention: need either (kv_indptr, kv_indices) or (req_to_token, req_pool_indices, seq_lens)");
  SGL_CHECK(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0, "decode_attention: Hq=%d not a multiple of Hkv=%d",
            num_q_heads, num_kv_heads);
  SGL_CHECK(head_dim > 0 && head_dim <= 256 && v_head_dim > 0 && v_head_dim <= 256,
            "decode_attention: head dims (%d, %d) outside (0, 256]", head_dim, v_head_dim);
  SGL_CHECK(max_kv_splits >= 1 && max_kv_splits <= 65535, "decode_attention: max_kv_splits=%d out of range",
            max_kv_splits);
  SGL_CHECK(batch <= 65535, "decode_attention: batch %d exceeds grid.z limit", batch);
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "decode_attention: dtype code %d unsupported (bf16=0, f16=1)", dtype);
  SGL_CHECK(kv_dtype == dtype || kv_dtype == SGL_FP8_E4M3, "decode_attention: kv_dtype %d must be the q dtype or fp8_e4m3", kv_dtype);
  const bool kv8 = kv_dtype == SGL_FP8_E4M3;
  SGL_CHECK(!kv8 || (head_dim == v_head_dim && (head_dim == 128 || head_dim == 64)),
            "decode_attention: the fp8 KV cache needs head_dim == v_head_dim in {64, 128} (got %d, %d)", head_dim, v_head_dim);
  SGL_CHECK(!kv8 || (k_stride_t % 16 == 0 && v_stride_t % 16 == 0 && k_stride_h % 16 == 0 && v_stride_h % 16 == 0),
            "decode_attention: fp8 K/V rows must be 16-byte aligned");
  if (head_dim == v_head_dim && (head_dim == 128 || head_dim == 64)) {
    SGL_CHECK(k_stride_t % 8 == 0 && v_stride_t % 8 == 0 && k_stride_h % 8 == 0 && v_stride_h % 8 == 0 &&
                  q_stride_t % 8 == 0 && ((uintptr_t)k_buffer % 16) == 0 && ((uintptr_t)v_buffer % 16) == 0 &&
                  ((uintptr_t)q % 16) == 0,
              "decode_attention: q/k/v rows must be 16-byte aligned for the MFMA path");
  }
  DecodeParams p;
#ifdef SGL_DEC_TIMELINE
  p.tl = g_dec_tl;
#endif
  p.q = q; p.q_stride_t = q_stride_t;
  p.k_buf = k_buffer; p.v_buf = v_buffer;
  p.k_stride_t = k_stride_t; p.k_stride_h = k_stride_h; p.v_stride_t = v_stride_t; p.v_stride_h = v_stride_h;
  p.o = o; p.o_stride_t = o_stride_t;
  p.kv_indptr = kv_indptr; p.kv_indices = kv_indices;
  p.req_to_token = req_to_token; p.req_to_token_stride = req_to_token_stride;
  p.req_pool_indices = req_pool_indices; p.seq_lens = seq_lens;
  p.attn_logits = attn_logits; p.attn_lse = attn_lse; p.num_kv_splits = num_kv_splits; p.max_kv_splits = max_kv_splits;
  p.bs = batch; p.hq = num_q_heads; p.hkv = num_kv_heads; p.group = num_q_heads / num_kv_heads; p.dv = v_head_dim;
  p.sm_scale = sm_scale; p.logit_cap = logit_cap;
  p.kv_fp8 = kv8 ? 1 : 0;
  p.k_scale = kv8 ? k_scale : 1.0f; p.v_scale = kv8 ? v_scale : 1.0f;
  p.merge_cnt = merge_cnt; p.mq_o = mq_o; p.mq_q = mq_q; p.mq_s = mq_s;
  p.prefix_len = prefix_len; p.prefix_splits = prefix_splits;
  p.sched = sched; p.sched_chunks = 0;
  if (sched != nullptr) {
    SGL_CHECK(merge_cnt != nullptr && kv_indptr != nullptr && kv_indices != nullptr && prefix_splits == 0 &&
                  ((uintptr_t)sched % 16) == 0 && sched_capacity >= 1 && sched_capacity <= 65535,
              "decode_attention_scheduled: needs kv_indptr / kv_indices, merge counters, a 16-byte aligned unit list and 1 <= capacity <= 65535");
    p.sched_chunks = sched_capacity;
  }
  if (prefix_splits > 0) {
    SGL_CHECK(head_dim == v_head_dim && (head_dim == 128 || head_dim == 64) && g_decode_mode != 1,
              "decode_attention_cascade: needs the MFMA kernel (head_dim 64 / 128) in its workgroup-per-split mode");
    SGL_CHECK(prefix_len > 0 && prefix_splits < max_kv_splits,
              "decode_attention_cascade: prefix_len=%d, prefix_splits=%d must be positive and leave split slots for the suffix (max_kv_splits=%d)",
              prefix_len, prefix_splits, max_kv_splits);
  }
  if (merge_cnt != nullptr) {
    SGL_CHECK(head_dim == v_head_dim && (head_dim == 128 || head_dim == 64) && g_decode_mode != 1,
              "decode_attention_merge_quant: needs the MFMA kernel (head_dim 64 / 128) in its workgroup-per-split mode");
    SGL_CHECK((mq_o || mq_q) && (!mq_q || mq_s) && (num_q_heads * v_head_dim) % 8 == 0 && num_q_heads * v_head_dim <= 16384,
              "decode_attention_merge_quant: bad outputs or Hq*Dv=%d", num_q_heads * v_head_dim);
  }
  hipStream_t st = (hipStream_t)stream;
  return dtype == SGL_BF16 ? launch_all<__bf16>(p, head_dim, st) : launch_all<_Float16>(p, head_dim, st);
}

extern "C" int sgl_mi355_decode_attention(
    const void* q, int64_t q_stride_t, const void* k_buffer, const void* v_buffer, int64_t k_stride_t,
    int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h, void* o, int64_t o_stride_t, const int32_t* kv_indptr,
    const int32_t* kv_indices, const int32_t* req_to_token, int64_t req_to_token_stride, const int64_t* req_pool_indices,
    const int64_t* seq_lens, float* attn_logits, float* attn_lse, const int32_t* num_kv_splits, int max_kv_splits,
    int batch, int num_q_heads, int num_kv_heads, int head_dim, int v_head_dim, float sm_scale, float logit_cap,
    int dtype, int kv_dtype, float k_scale, float v_scale, void* stream) {
  return decode_entry(q, q_stride_t, k_buffer, v_buffer, k_stride_t, k_stride_h, v_stride_t, v_stride_h, o, o_stride_t, kv_indptr,
                      kv_indices, req_to_token, req_to_token_stride, req_pool_indices, seq_lens, attn_logits, attn_lse,
                      num_kv_splits, max_kv_splits, batch, num_q_heads, num_kv_heads, head_dim, v_head_dim, sm_scale, logit_cap,
                      dtype, kv_dtype, k_scale, v_scale, nullptr, nullptr, nullptr, nullptr, stream);
}

// Stage 1 with the in-launch stage 2: the last workgroup of each request merges its splits (out_o, optional) and quantises
// the merged row per token (out_q / out_s, optional) -- decode_attention_fwd + sgl_per_token_quant_fp8 in ONE launch.
// merge_counters: int32 [batch], zero on entry (they are left zero).
extern "C" int sgl_mi355_decode_attention_merge_quant(
    const void* q, int64_t q_stride_t, const void* k_buffer, const void* v_buffer, int64_t k_stride_t,
    int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h, const int32_t* kv_indptr, const int32_t* kv_indices,
    float* attn_logits, float* attn_lse, const int32_t* num_kv_splits, int max_kv_splits, int batch, int num_q_heads,
    int num_kv_heads, int head_dim, int v_head_dim, float sm_scale, float logit_cap, int dtype, int kv_dtype, float k_scale,
    float v_scale, int32_t* merge_counters, void* out_o, void* out_q, float* out_s, void* stream) {
  SGL_CHECK(merge_counters != nullptr && kv_indptr != nullptr, "decode_attention_merge_quant: null pointer");
  return decode_entry(q, q_stride_t, k_buffer, v_buffer, k_stride_t, k_stride_h, v_stride_t, v_stride_h, nullptr, 0, kv_indptr,
                      kv_indices, nullptr, 0, nullptr, nullptr, attn_logits, attn_lse, num_kv_splits, max_kv_splits, batch,
                      num_q_heads, num_kv_heads, head_dim, v_head_dim, sm_scale, logit_cap, dtype, kv_dtype, k_scale, v_scale,
                      merge_counters, out_o, out_q, out_s, stream);
}

// sgl_mi355_decode_attention_merge_quant with the workgroups' (request, split) units taken from the sorted list
// sgl_mi355_decode_schedule wrote (sched; csrc/kv_index.hip) -- num_kv_splits from the same call.  Same splits, same arithmetic, same
// outputs bit for bit; another dispatch order (longest unit first) and a grid of (kv heads x head chunks, sched_units).
extern "C" int sgl_mi355_decode_attention_scheduled(
    const void* q, int64_t q_stride_t, const void* k_buffer, const void* v_buffer, int64_t k_stride_t,
    int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h, const int32_t* kv_indptr, const int32_t* kv_indices,
    float* attn_logits, float* attn_lse, const int32_t* num_kv_splits, int max_kv_splits, const int32_t* sched, int sched_units,
    int batch, int num_q_heads, int num_kv_heads, int head_dim, int v_head_dim, float sm_scale, float logit_cap, int dtype,
    int kv_dtype, float k_scale, float v_scale, int32_t* merge_counters, void* out_o, void* out_q, float* out_s, void* stream) {
  SGL_CHECK(merge_counters != nullptr && kv_indptr != nullptr && sched != nullptr, "decode_attention_scheduled: null pointer");
  return decode_entry(q, q_stride_t, k_buffer, v_buffer, k_stride_t, k_stride_h, v_stride_t, v_stride_h, nullptr, 0, kv_indptr,
                      kv_indices, nullptr, 0, nullptr, nullptr, attn_logits, attn_lse, num_kv_splits, max_kv_splits, batch,
                      num_q_heads, num_kv_heads, head_dim, v_head_dim, sm_scale, logit_cap, dtype, kv_dtype, k_scale, v_scale,
                      merge_counters, out_o, out_q, out_s, stream, 0, 0, sched, sched_units);
}

// host side of the unit list's geometry for csrc/kv_index.hip
int sgl_mi355_internal_decode_sched_target(int batch, int num_q_heads, int num_kv_heads, int rounds_pct) {
  return decode_sched_target(batch, num_kv_heads * ((num_q_heads / num_kv_heads + 15) / 16), rounds_pct);
}

// Cascade shared-prefix decode: every request of the batch shares its first prefix_len slots (one radix-tree
// node, radix_cache.py:370-412).  Two launches: (1) the prefix is attended ONCE for all requests -- a launch of the extend kernel
// with the batch's decode queries as its query block and at most prefix_splits splits of the shared rows (prefix_indices int32
// [prefix_len]), partials into the top split slots; (2) every request's private part (kv_indptr / kv_indices over the slots after
// the prefix, num_kv_splits[b] splits, at least the new token) whose first split continues the online softmax from the prefix
// state -- the math of merge_state (sgl-kernel/csrc/attention/merge_attn_states.cu) -- with the in-launch stage 2 producing
// out_o (T, optional) and / or the per-token fp8 row out_q / out_s (optional).  attn_logits / attn_lse hold max_kv_splits
// split slots per (request, head): prefix_splits + max(num_kv_splits) <= max_kv_splits.  merge_counters: int32 [batch], zero
// on entry, left zero.
int sgl_mi355_internal_cascade_prefix(const void* q, int64_t q_stride_t, const void* k_buffer, const void* v_buffer, int64_t k_stride_t,
                                      int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h, const int32_t* prefix_indices,
                                      int prefix_len, int chunk, int splits, int slot0, float* attn_logits, float* attn_lse,
                                      int max_kv_splits, int batch, int num_q_heads, int num_kv_heads, int head_dim, float sm_scale,
                                      float logit_cap, int dtype, int kv_dtype, float k_scale, float v_scale, hipStream_t st);  // extend_attention.hip

extern "C" int sgl_mi355_decode_attention_cascade(
    const void* q, int64_t q_stride_t, const void* k_buffer, const void* v_buffer, int64_t k_stride_t,
    int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h, const int32_t* prefix_indices, int prefix_len,
    int prefix_splits, const int32_t* kv_indptr, const int32_t* kv_indices, float* attn_logits, float* attn_lse,
    const int32_t* num_kv_splits, int max_kv_splits, int batch, int num_q_heads, int num_kv_heads, int head_dim,
    int v_head_dim, float sm_scale, float logit_cap, int dtype, int kv_dtype, float k_scale, float v_scale,
    int32_t* merge_counters, void* out_o, void* out_q, float* out_s, void* stream) {
  SGL_CHECK(merge_counters != nullptr && kv_indptr != nullptr && prefix_indices != nullptr, "decode_attention_cascade: null pointer");
  if (batch == 0) return SGL_MI355_OK;
  SGL_CHECK(q && k_buffer && v_buffer && attn_logits && attn_lse, "decode_attention_cascade: null tensor pointer");
  SGL_CHECK(head_dim == v_head_dim && (head_dim == 128 || head_dim == 64), "decode_attention_cascade: head_dim 64 / 128 only (got %d, %d)",
            head_dim, v_head_dim);
  SGL_CHECK(prefix_len > 0 && prefix_splits >= 1 && prefix_splits < max_kv_splits && num_kv_heads > 0 && num_q_heads % num_kv_heads == 0,
            "decode_attention_cascade: prefix_len=%d prefix_splits=%d max_kv_splits=%d", prefix_len, prefix_splits, max_kv_splits);
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "decode_attention_cascade: dtype code %d unsupported", dtype);
  SGL_CHECK(q_stride_t % 8 == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)k_buffer % 16) == 0 && ((uintptr_t)v_buffer % 16) == 0,
            "decode_attention_cascade: q/k/v rows must be 16-byte aligned");
  // whole 64-row tiles per split; fewer splits than asked for when the prefix is short (every split holds at least one row)
  const int chunk = ((prefix_len + prefix_splits - 1) / prefix_splits + 63) / 64 * 64;
  const int splits = (prefix_len + chunk - 1) / chunk;
  int rc = sgl_mi355_internal_cascade_prefix(q, q_stride_t, k_buffer, v_buffer, k_stride_t, k_stride_h, v_stride_t, v_stride_h,
                                             prefix_indices, prefix_len, chunk, splits, max_kv_splits - splits, attn_logits, attn_lse,
                                             max_kv_splits, batch, num_q_heads, num_kv_heads, head_dim, sm_scale, logit_cap, dtype,
                                             kv_dtype, k_scale, v_scale, (hipStream_t)stream);
  if (rc != SGL_MI355_OK) return rc;
  return decode_entry(q, q_stride_t, k_buffer, v_buffer, k_stride_t, k_stride_h, v_stride_t, v_stride_h, nullptr, 0, kv_indptr,
                      kv_indices, nullptr, 0, nullptr, nullptr, attn_logits, attn_lse, num_kv_splits, max_kv_splits, batch,
                      num_q_heads, num_kv_heads, head_dim, v_head_dim, sm_scale, logit_cap, dtype, kv_dtype, k_scale, v_scale,
                      merge_counters, out_o, out_q, out_s, stream, prefix_len, splits);
}
