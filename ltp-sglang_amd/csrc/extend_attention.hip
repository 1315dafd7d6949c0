// Extend (prefill / chunked-prefill / RadixAttention-hit) attention, gfx950.
//
// Replaces extend_attention_fwd / _fwd_kernel
//   (python/sglang/srt/layers/attention/triton_ops/extend_attention.py:41-438): stage 1 scores the new query
//   tokens against the cached PREFIX gathered from the paged pool through kv_indices, stage 2 runs the causal
//   triangle against the contiguous k_extend / v_extend of the same request; o = acc / l.
// and, through the req_to_token addressing mode, extend_attention_cpu (sgl-kernel/csrc/cpu/extend.cpp:579-723).
// Semantics match the oracle torch_native_backend.py:27-110 (causal, top-left aligned with the prefix offset).
//
// Structure: one 256-thread workgroup = one (request, kv head [8-head chunk], block of BQ query positions):
// 128 query "rows" = (q head of the GQA group) x (position), so every K/V tile staged in LDS is shared by all
// heads of the group.  K/V tiles of 64 tokens are gathered (prefix: 256-B pool rows through kv_indices;
// extend: contiguous rows) with 16-B loads, register-staged one tile ahead and written to double-buffered,
// XOR-swizzled LDS images (conflict-free ds_read_b128 K fragments and ds_read_b64_tr_b16 V^T fragments).
// Math per wave (2 x 16 query rows): S^T = K Q^T and O^T += V^T P^T with v_mfma_f32_16x16x32, online softmax in
// registers with two xor-shuffles per row reduction -- the decode kernel's formulation with (head, position)
// pairs in place of heads.  Workgroups of the same (request, kv head) are mapped to the same XCD so the K/V they
// re-read stay in that XCD's L2.
#include "extend_params.h"
#include <type_traits>

namespace {

__device__ __forceinline__ float softcap2(float s_scaled, float cap) {
  const float y = s_scaled / cap;
  const float t = 2.0f / (1.0f + __expf(-2.0f * y)) - 1.0f;
  return cap * t * kLog2e;
}

// 8 e4m3 bytes -> 8 T (exact)
template <typename T>
__device__ __forceinline__ u32x4_t cvt8_fp8(const u32x2_t& in) {
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  struct P2 { T a, b; };
  u32x4_t out;
#pragma unroll
  for (int w2 = 0; w2 < 2; ++w2) {
    const f32x2 f01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)in[w2], false);
    const f32x2 f23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)in[w2], true);
    out[2 * w2] = __builtin_bit_cast(uint32_t, P2{(T)f01[0], (T)f01[1]});
    out[2 * w2 + 1] = __builtin_bit_cast(uint32_t, P2{(T)f23[0], (T)f23[1]});
  }
  return out;
}

// CAP: logit soft-capping (its own instantiation).  MASKED: custom mask and / or sliding window (speculative decoding, window
// models): every tile takes the masking pass, the per-score mask bytes are plain global loads -- a correctness path whose
// extra work stays out of the unmasked instantiations; the logit cap is a run-time branch there.
// QT: 16-row query tiles per wave (2: 128 (head, position) pairs per workgroup, the prefill form; 1: 64 pairs, half the per-tile
// latency of a wave -- the cascade prefix pass, whose few workgroups are latency-bound).
template <typename T, int D, bool KV8 = false, bool CAP = false, bool MASKED = false, int QT = 2>
__global__ __launch_bounds__(256, 2) void extend_attn_kernel(const ExtendParams p) {
  using Tr = ElemTraits<T>;
  using vec8 = typename Tr::vec8;
  constexpr int ROWB = D * 2;
  constexpr int LPR = ROWB / 16;
  constexpr int RPI = 256 / LPR;   // rows per workgroup-wide load instruction
  constexpr int NI = kKT / RPI;    // load instructions per operand per tile
  constexpr int KS = D / 32;
  constexpr int NT = D / 16;
  constexpr int TILE_B = kKT * ROWB;
  constexpr int RPB = (ROWB >= 256) ? 1 : 256 / ROWB;
  constexpr int KMASK = (LPR < 16 ? LPR : 16) - 1;
  constexpr int VCH = ROWB / 32;
  constexpr int VMASK = (VCH < 8 ? VCH : 8) - 1;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- block -> (request, kv head chunk, q block); same (request, kv head) => same blockIdx % 8 (one XCD) ----
  const int bid = blockIdx.x;
  const int lo = bid & 7, rest = bid >> 3;
  const int qb = p.nqb - 1 - rest % p.nqb;  // longest key ranges first (causal): the launch ends with the short blocks
  const int pair = (rest / p.nqb) * 8 + lo;
  const int npairs = p.bs * p.hkv * p.hchunks;
  if (pair >= npairs) return;
  const int b = pair / (p.hkv * p.hchunks);
  const int khc = pair - b * (p.hkv * p.hchunks);
  const int kh = khc / p.hchunks, hc = khc - kh * p.hchunks;

  const int bq = 1 << p.bq_log2;          // query positions per workgroup
  const int gslots = (64 * QT) >> p.bq_log2;    // head slots per workgroup (<= 4 QT)
  const bool cascade = p.casc_bs > 0;
  const int q0 = cascade ? 0 : (p.qo_indptr ? p.qo_indptr[b] : p.extend_start_loc[b]);
  const int ext_len = cascade ? p.casc_bs : (p.qo_indptr ? p.qo_indptr[b + 1] - q0 : p.extend_seq_lens[b]);
  const int qpos0 = qb * bq;
  if (qpos0 >= ext_len) return;
  int pre_len;
  const int32_t* idx_row;
  if (cascade) {
    pre_len = min(p.casc_chunk, p.casc_prefix_len - b * p.casc_chunk);   // > 0: the launcher sizes bs = ceil(len / chunk)
    idx_row = p.kv_indices + b * p.casc_chunk;
  } else if (p.kv_indptr) {  // mode by indptr: kv_indices may legitimately be NULL when no request has a prefix
    const int s0 = p.kv_indptr[b];
    pre_len = p.kv_indptr[b + 1] - s0;
    idx_row = p.kv_indices + s0;
  } else {
    pre_len = (int)p.seq_lens[b] - ext_len;
    idx_row = p.req_to_token + p.req_pool_indices[b] * p.req_to_token_stride;
  }

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = lane & 15, g = lane >> 4;
  const uint8_t* mrow = nullptr;        // MASKED: this request's mask block
  const int seq_total = pre_len + ext_len;
  if constexpr (MASKED) {
    if (p.custom_mask) mrow = p.custom_mask + p.mask_indptr[b];
  }
  const bool capped = CAP || (MASKED && p.logit_cap > 0.0f);

  // ---- this wave's two 16-row query tiles: tile t covers head slot (16 t) / bq, positions (16 t) % bq ... ----
  int hq_idx[QT], qpos[QT];
  bool head_ok[QT];
  vec8 qf[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int t16 = (QT * w + qt) * 16;
    const int hslot = t16 >> p.bq_log2;
    const int hl = hc * (4 * QT) + hslot;  // head within the GQA group (at most 4 QT head slots per workgroup: >= 16 positions each)
    head_ok[qt] = hl < p.group && hslot < gslots;
    hq_idx[qt] = kh * p.group + min(hl, p.group - 1);
    qpos[qt] = qpos0 + (t16 & (bq - 1)) + a;  // position of this lane's query row inside the extend part
    const bool row_ok = head_ok[qt] && qpos[qt] < ext_len;
    const T* qrow = (const T*)p.q + (int64_t)(q0 + min(qpos[qt], ext_len - 1)) * p.q_stride_t + (int64_t)hq_idx[qt] * D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (row_ok) {
        qf[qt][ks] = *(const vec8*)(qrow + 32 * ks + 8 * g);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[qt][ks][j] = (T)0.0f;
      }
    }
  }

  // ---- cooperative staging: thread loads chunk c16 of rows rsub + RPI*i ----
  const int c16 = tid % LPR, rsub = tid / LPR;
  constexpr int PB = KV8 ? 1 : 2;  // bytes per pool element; a thread's 8 elements are 8 * PB bytes of the pool row
  const char* kpool = (const char*)p.k_buf + ((int64_t)kh * p.k_stride_h) * PB + c16 * 8 * PB;
  const char* vpool = (const char*)p.v_buf + ((int64_t)kh * p.v_stride_h) * PB + c16 * 8 * PB;
  const char* kext = (const char*)p.ke + ((int64_t)q0 * p.ke_stride_t + (int64_t)kh * D) * 2 + c16 * 16;
  const char* vext = (const char*)p.ve + ((int64_t)q0 * p.ve_stride_t + (int64_t)kh * D) * 2 + c16 * 16;
  const int64_t kpst = p.k_stride_t * PB, vpst = p.v_stride_t * PB, kest = p.ke_stride_t * 2, vest = p.ve_stride_t * 2;

  const int npre_tiles = (pre_len + kKT - 1) / kKT;
  const int ext_end = p.is_causal ? min(ext_len, qpos0 + bq) : ext_len;  // keys any row of this block may see
  const int next_tiles = cascade ? 0 : (ext_end + kKT - 1) / kKT;
  const int ntiles = npre_tiles + next_tiles;

  // Staging registers hold RAW loaded data: nothing consumes a load at issue time (zeroing the V rows past the end and the
  // fp8 -> T conversion happen when the tile is written to LDS, one iteration later), and the prefix slots of tile t + 1 are
  // fetched one iteration before its rows are issued -- a consumer at issue time puts a full global-memory round trip in
  // front of every tile's first MFMA.
  constexpr int RW = KV8 ? 2 : 4;  // dwords per lane and row piece as loaded
  typedef uint32_t raw_t __attribute__((ext_vector_type(RW)));
  u32x4_t kreg[NI], vreg[NI];   // extend rows (always 16-bit elements), and 16-bit prefix rows
  raw_t kraw[NI], vraw[NI];     // fp8 prefix rows (KV8)
  int idn[NI];                  // pool slots of the next prefix tile's rows
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};
  // (16-bit pools: ONE unconditional load site per operand with selected addresses -- loads under a branch make the
  // wait-count insertion put vmcnt(0) in front of the next tile's first MFMA)
  const int32_t* idx_dummy = cascade ? p.kv_indices : (p.qo_indptr ? p.qo_indptr : p.extend_start_loc);  // any readable int32 when there is no prefix row
  auto load_idx = [&](int t) {
    const bool pre = t < npre_tiles;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = t * kKT + rsub + RPI * i;
      const int32_t* src = pre ? idx_row + min(r, pre_len - 1) : idx_dummy;
      idn[i] = *src;
    }
  };
  auto issue = [&](int t) {
    const bool pre = t < npre_tiles;
    if constexpr (KV8) {
      if (pre) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          kraw[i] = *(const raw_t*)(kpool + (int64_t)idn[i] * kpst);
          vraw[i] = *(const raw_t*)(vpool + (int64_t)idn[i] * vpst);
        }
      } else {
        const int base = (t - npre_tiles) * kKT;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int rr = min(base + rsub + RPI * i, ext_len - 1);
          kreg[i] = *(const u32x4_t*)(kext + (int64_t)rr * kest);
          vreg[i] = *(const u32x4_t*)(vext + (int64_t)rr * vest);
        }
      }
    } else {
      const int base = (t - npre_tiles) * kKT;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int rr = min(base + rsub + RPI * i, ext_len - 1);
        const char* ka = pre ? kpool + (int64_t)idn[i] * kpst : kext + (int64_t)rr * kest;
        const char* va = pre ? vpool + (int64_t)idn[i] * vpst : vext + (int64_t)rr * vest;
        kreg[i] = *(const u32x4_t*)ka;
        vreg[i] = *(const u32x4_t*)va;
      }
    }
  };
  auto lstore = [&](int buf, int t) {  // t: the tile held in the staging registers
    char* kl = smem + buf * 2 * TILE_B;
    char* vl = kl + TILE_B;
    const bool pre = t < npre_tiles;
    const int base = pre ? t * kKT : (t - npre_tiles) * kKT;
    const int limit = pre ? pre_len : ext_len;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = rsub + RPI * i;
      const int fk = (row / RPB) & KMASK, fv = (row / RPB) & VMASK;
      u32x4_t kk = kreg[i], vv = vreg[i];
      if constexpr (KV8) {
        if (pre) {
          kk = cvt8_fp8<T>(__builtin_bit_cast(u32x2_t, kraw[i]));
          vv = cvt8_fp8<T>(__builtin_bit_cast(u32x2_t, vraw[i]));
        }
      }
      if (base + row >= limit) vv = zero4;  // 0 * garbage must stay 0
      *(u32x4_t*)(kl + row * ROWB + ((c16 ^ fk) << 4)) = kk;
      *(u32x4_t*)(vl + row * ROWB + ((((c16 >> 1) ^ fv) << 5) | ((c16 & 1) << 4))) = vv;
    }
  };

  float m_i[QT], l_i[QT];
  f32x4_t acc[QT][NT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_i[qt] = -INFINITY;
    l_i[qt] = 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[qt][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  const float scale_log2 = p.sm_scale * kLog2e;

  if (ntiles > 0) {
    load_idx(0);
    issue(0);
    load_idx(1);
    lstore(0, 0);
  }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    issue(min(t + 1, ntiles - 1));  // unconditional (after the last tile: a re-read that is never stored)
    load_idx(t + 2);
    const char* kl = smem + buf * 2 * TILE_B;
    const char* vl = kl + TILE_B;
    const bool in_prefix = t < npre_tiles;
    const int kbase = in_prefix ? t * kKT : (t - npre_tiles) * kKT;  // index of the tile's first key in its phase
    const int klimit = in_prefix ? pre_len : ext_len;
    float tsm = p.sm_scale, tlog2 = scale_log2, pvs = 1.0f;  // the pool's k_scale / v_scale apply to prefix tiles only
    if constexpr (KV8) {
      if (in_prefix) {
        tsm *= p.k_scale;
        tlog2 *= p.k_scale;
        pvs = p.v_scale;
      }
    }

    // a wave whose rows all lie in the causal past of this tile skips it (wave-uniform)
    const int wave_qmax = qpos0 + (((QT * w + QT - 1) * 16) & (bq - 1)) + 15;
    const int wave_qmax0 = qpos0 + (((QT * w) * 16) & (bq - 1)) + 15;
    // (a custom mask replaces the causal rule in the extend phase: no causal skip then)
    const bool skip = !in_prefix && p.is_causal && !(MASKED && mrow) && kbase > max(wave_qmax, wave_qmax0);
    const int wave_qmin = qpos0 + (bq >= 16 * QT ? ((16 * QT * w) & (bq - 1)) : 0);
    const bool need_mask = MASKED || (kbase + kKT > klimit) || (!in_prefix && p.is_causal && kbase + kKT - 1 > wave_qmin);
    if (!skip) {
      // ---- S^T tiles: s[qt][tt][r] = score(query row a of tile qt, key 16 tt + 4 g + r) ----
      f32x4_t s[QT][4];
      // K fragments two key groups ahead of their MFMAs (double-buffered registers): a fragment set requested right before its
      // MFMAs made every 8 of them wait for an LDS round trip
      vec8 kfb[2][KS];
      auto ld_k = [&](int tt, vec8 (&kf)[KS]) {
        const int row = 16 * tt + a;
        const int fk = (row / RPB) & KMASK;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *(const vec8*)(kl + row * ROWB + (((4 * ks + g) ^ fk) << 4));
      };
      ld_k(0, kfb[0]);
      ld_k(1, kfb[1]);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          f32x4_t c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) c = Tr::mfma16(kfb[tt & 1][ks], qf[qt][ks], c);
          s[qt][tt] = c;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tt + 2 < 4) ld_k(tt + 2, kfb[tt & 1]);
      }
      // ---- soft-cap every score (CAP; otherwise the scale is folded into the exponent below); then ONE wave-uniform branch
      // masks the ragged last tile of a phase and the causal diagonal (tested per score, the flags cut this loop into ~70
      // basic blocks and nothing was scheduled across them) ----
      if (capped) {   // compile-time true / false in the unmasked instantiations
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[qt][tt][r] = softcap2(s[qt][tt][r] * tsm, p.logit_cap);
      }
      if constexpr (MASKED) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const int qp = qpos[qt];
          const bool qrow_ok = qp < ext_len;   // padding rows are never stored: keep their mask reads in bounds
          const uint8_t* mq = mrow ? mrow + (int64_t)min(qp, ext_len - 1) * seq_total + (in_prefix ? 0 : pre_len) : nullptr;
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kbase + 16 * tt + 4 * g + r;
              bool ok = key < klimit;
              if (in_prefix) {
                if (p.sliding_window > 0) ok = ok && (qp <= key + p.sliding_window);
                if (mq && !p.skip_prefix_mask && ok && qrow_ok) ok = mq[key] != 0;
              } else if (mq) {
                if (ok && qrow_ok) ok = mq[key] != 0;
              } else if (p.is_causal) {
                ok = ok && (key <= qp);
              }
              s[qt][tt][r] = ok ? s[qt][tt][r] : -INFINITY;
            }
        }
      } else if (need_mask) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kbase + 16 * tt + 4 * g + r;
              bool ok = key < klimit;
              if (!in_prefix && p.is_causal) ok = ok && (key <= qpos[qt]);
              s[qt][tt][r] = ok ? s[qt][tt][r] : -INFINITY;
            }
      }
      // ---- online softmax per query tile, branch-free (m_i in log2 units; the score scale is folded into the exponent's fma;
      // a row that has seen no key yet keeps m = -inf, l = 0, acc = 0 through the clamped maximum) ----
      const float cs = capped ? 1.0f : tlog2;  // x = s * cs (sm_scale * log2 e > 0; soft-capped scores are already in log2 units)
      vec8 pf[QT][2];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float m = -INFINITY;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) m = fmaxf(m, s[qt][tt][r]);
        m = pair32_max(pair16_max(m));
        const float m_new = fmaxf(m_i[qt], m * cs);
        const float m_safe = fmaxf(m_new, -1e30f);
        const float alpha = __builtin_amdgcn_exp2f(m_i[qt] - m_safe);
        float lsum = 0.f;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][tt][r], cs, -m_safe));
            lsum += pv;
            // PV k-step u = tt / 2 takes keys 32 u + {4 g + j, 16 + 4 g + j}: element index 4 (tt & 1) + r
            pf[qt][tt >> 1][4 * (tt & 1) + r] = Tr::from_f32(KV8 ? pv * pvs : pv);
          }
        }
        l_i[qt] = l_i[qt] * alpha + lsum;
        m_i[qt] = m_new;
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[qt][n] *= alpha;
      }
      // ---- O^T += V^T P^T: the V^T fragments of two column tiles (8 transposed reads) are requested while the previous two are
      // multiplied -- left to itself the compiler kept ONE fragment in flight, i.e. an LDS round trip per pair of MFMAs ----
      auto ld_v = [&](int n, int u) -> vec8 {
        s16x4_t t0, t1;
        {
          const int row = 32 * u + 4 * g + (a >> 2);
          const int fv = (row / RPB) & VMASK;
          t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4_t __attribute__((address_space(3)))*)(vl + row * ROWB + ((n ^ fv) << 5) + ((a & 3) << 3)));
        }
        {
          const int row = 32 * u + 16 + 4 * g + (a >> 2);
          const int fv = (row / RPB) & VMASK;
          t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4_t __attribute__((address_space(3)))*)(vl + row * ROWB + ((n ^ fv) << 5) + ((a & 3) << 3)));
        }
        return __builtin_bit_cast(vec8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
      };
      constexpr int NB = 2;  // column tiles per batch
      vec8 vfb[2][NB][2];
#pragma unroll
      for (int nn = 0; nn < NB; ++nn)
#pragma unroll
        for (int u = 0; u < 2; ++u) vfb[0][nn][u] = ld_v(nn, u);
#pragma unroll
      for (int nb = 0; nb < NT / NB; ++nb) {
        __builtin_amdgcn_sched_barrier(0);
        if (nb + 1 < NT / NB) {
#pragma unroll
          for (int nn = 0; nn < NB; ++nn)
#pragma unroll
            for (int u = 0; u < 2; ++u) vfb[(nb + 1) & 1][nn][u] = ld_v((nb + 1) * NB + nn, u);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nn = 0; nn < NB; ++nn)
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
              acc[qt][nb * NB + nn] = Tr::mfma16(vfb[nb & 1][nn][u], pf[qt][u], acc[qt][nb * NB + nn]);
      }
    }
    if (t + 1 < ntiles) lstore(buf ^ 1, t + 1);
    __syncthreads();
  }

  // ---- o = acc / l ; lane (a, g) owns query row a, columns 16 n + 4 g + [0, 4) ----
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = l_i[qt];
    l = pair32_sum(pair16_sum(l));
    if (cascade) {   // split partial of (request qpos, head hq_idx): O = acc / l in f32 and LSE = m ln 2 + ln l, the decode kernel's slot layout
      if (head_ok[qt] && qpos[qt] < ext_len) {
        const int64_t slot = ((int64_t)qpos[qt] * p.hq + hq_idx[qt]) * p.max_kv_splits + p.casc_slot0 + b;
        const float inv = 1.0f / l;   // l > 0: every split holds at least one key
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          f32x4_t o = acc[qt][n];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] *= inv;
          *(f32x4_t*)(p.part_o + slot * D + 16 * n + 4 * g) = o;
        }
        if (g == 0) p.part_lse[slot] = m_i[qt] * 0.6931471805599453f + __logf(l);
      }
      continue;
    }
    if (head_ok[qt] && qpos[qt] < ext_len) {
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      T* orow = (T*)p.o + (int64_t)(q0 + qpos[qt]) * p.o_stride_t + (int64_t)hq_idx[qt] * D;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        typename Tr::vec4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = Tr::from_f32(acc[qt][n][r] * inv);
        *(typename Tr::vec4*)(orow + 16 * n + 4 * g) = ov;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA form (round 3) of the unmasked 16-bit D = 128 kernel: extend_attn_kernel's workgroup shape, LDS images, fragment maps,
// 64-key tiles and arithmetic ORDER (the output is bit-identical, which the tests check), with two changes that only move data:
//   * K / V tiles travel HBM -> LDS by LDS-DMA (global_load_lds, 16 B per lane; the swizzle is applied on the source side: the lane
//     that writes position `pos` of row r fetches the chunk the image keeps there), one tile ahead into the other half of the
//     double buffer: no staging registers and no LDS-store pass (600 of the 4 600 cycles a lone wave spent per tile);
//   * the 36 registers this frees hold a whole phase's fragments: all 16 K fragments of a tile are requested before the first
//     S^T MFMA and all 16 V^T fragments before the first PV MFMA, so an LDS round trip is paid once per phase instead of once per
//     pair of key groups (QK^T took 1 700 cycles for 512 cycles of MFMA).
// The loop's global traffic is issued from asm statements and waited for by hand: seen by the compiler, an LDS-DMA makes
// SIInsertWaitcnts put vmcnt(0) in front of every ds_read_b64_tr_b16 (it cannot tell the two buffers apart), i.e. the next tile's
// flight would be waited for before this tile's PV.  Rows past the end of a phase are fetched from the phase's last row (finite
// data; their probabilities are exactly 0).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ext_dma16(const char* src, unsigned lds_addr) {   // 64 lanes x 16 B -> LDS [lds_addr, + 1 KiB)
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ int ext_load_i32(const int32_t* src) {
  int v;
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(src) : "memory");
  return v;
}

#ifdef SGL_EXT_TIMELINE
// tools/debug/ext_timeline.py: s_memtime stamps of workgroup SGL_EXT_TIMELINE's four waves, first 24 tiles, 6 stamps per tile
long long* g_ext_tl = nullptr;
#define EXT_STAMP(k) do { if (tl && t < 24) tl[((w * 24 + t) * 6) + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define EXT_STAMP(k) do { } while (0)
#endif
int g_extend_dma = 1;  // sgl_mi355_extend_attention_set_mode: 0 = always the register-staged kernel, 1 = default rule (the 8-wave 32x32x16 kernel of extend_attention_phased.hip where it applies and the grid is large enough, else the LDS-DMA kernel with 4 or 8 waves by the rule in launch_mfma), 2 / 3 = always the LDS-DMA kernel with 8 / 4 waves, 5 = always the 32x32x16 kernel where it applies
int g_extend_kv_hint = 0;  // sgl_mi355_extend_attention_set_kv_hint: mean keys a query block attends to (0: unknown)

// NW: waves per workgroup -- 4 (128 rows, two workgroups per CU) or 8 (256 rows = twice the query positions per K / V tile, one
// workgroup per CU: half the tile traffic)
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void extend_attn_dma_kernel(const ExtendParams p) {
  using Tr = ElemTraits<T>;
  using vec8 = typename Tr::vec8;
  constexpr int D = 128, QT = 2, ROWB = 256, KS = 4, NT = 8;
  constexpr int TILE_B = kKT * ROWB;  // 16 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][K 16 KiB | V 16 KiB]

  const int bid = blockIdx.x;
  const int lo = bid & 7, rest = bid >> 3;
  const int qb = p.nqb - 1 - rest % p.nqb;
  const int pair = (rest / p.nqb) * 8 + lo;
  const int npairs = p.bs * p.hkv * p.hchunks;
  if (pair >= npairs) return;
  const int b = pair / (p.hkv * p.hchunks);
  const int khc = pair - b * (p.hkv * p.hchunks);
  const int kh = khc / p.hchunks, hc = khc - kh * p.hchunks;

  const int bq = 1 << p.bq_log2;
  const int gslots = (16 * NW * QT) >> p.bq_log2;
  const int q0 = p.qo_indptr ? p.qo_indptr[b] : p.extend_start_loc[b];
  const int ext_len = p.qo_indptr ? p.qo_indptr[b + 1] - q0 : p.extend_seq_lens[b];
  const int qpos0 = qb * bq;
  if (qpos0 >= ext_len) return;
  int pre_len;
  const int32_t* idx_row;
  if (p.kv_indptr) {
    const int s0 = p.kv_indptr[b];
    pre_len = p.kv_indptr[b + 1] - s0;
    idx_row = p.kv_indices + s0;
  } else {
    pre_len = (int)p.seq_lens[b] - ext_len;
    idx_row = p.req_to_token + p.req_pool_indices[b] * p.req_to_token_stride;
  }

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = lane & 15, g = lane >> 4;
#ifdef SGL_EXT_TIMELINE
  long long* tl = (blockIdx.x == SGL_EXT_TIMELINE && lane == 0) ? p.tl : nullptr;
#endif

  vec8 qf[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int t16 = (QT * w + qt) * 16;
    const int hslot = t16 >> p.bq_log2;
    const int hl = hc * (4 * QT) + hslot;   // (hchunks > 1 only when group > 8: NW = 4 there)
    const int qp = qpos0 + (t16 & (bq - 1)) + a;
    const bool row_ok = hl < p.group && hslot < gslots && qp < ext_len;
    const T* qrow = (const T*)p.q + (int64_t)(q0 + min(qp, ext_len - 1)) * p.q_stride_t + (int64_t)(kh * p.group + min(hl, p.group - 1)) * D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (row_ok) {
        qf[qt][ks] = *(const vec8*)(qrow + 32 * ks + 8 * g);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[qt][ks][j] = (T)0.0f;
      }
    }
  }

  const int npre_tiles = (pre_len + kKT - 1) / kKT;
  const int ext_end = p.is_causal ? min(ext_len, qpos0 + bq) : ext_len;
  const int ntiles = npre_tiles + (ext_end + kKT - 1) / kKT;   // >= 1: the block has at least one extend key

  // ---- LDS-DMA staging: wave w fills rows RPWV w .. of a tile (64 / NW rows), 4 rows (1 KiB) per instruction; lane = (row l >> 4, position l & 15) ----
  constexpr int RPWV = kKT / NW, NIW = RPWV / 4;
  const int srow = lane >> 4, spos = lane & 15;
  const char* kpool = (const char*)p.k_buf + ((int64_t)kh * p.k_stride_h) * 2;
  const char* vpool = (const char*)p.v_buf + ((int64_t)kh * p.v_stride_h) * 2;
  const char* kext = (const char*)p.ke + ((int64_t)q0 * p.ke_stride_t + (int64_t)kh * D) * 2;
  const char* vext = (const char*)p.ve + ((int64_t)q0 * p.ve_stride_t + (int64_t)kh * D) * 2;
  const int64_t kpst = p.k_stride_t * 2, vpst = p.v_stride_t * 2, kest = p.ke_stride_t * 2, vest = p.ve_stride_t * 2;
  const int32_t* idx_dummy = p.qo_indptr ? p.qo_indptr : p.extend_start_loc;   // any readable int32 when there is no prefix row
  typedef __attribute__((address_space(3))) void* lptr_t;

  auto load_idx = [&](int t, int (&idn)[NIW]) {   // pool slots of tile t's rows (valid only after the caller's wait)
    const bool pre = t < npre_tiles;
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
      const int r = t * kKT + RPWV * w + 4 * i + srow;
      idn[i] = ext_load_i32(pre ? idx_row + min(r, pre_len - 1) : idx_dummy);
    }
  };
  auto stage = [&](int t, const int (&idn)[NIW]) {   // tile t -> buffer t & 1 (2 NIW LDS-DMA instructions per wave)
    const bool pre = t < npre_tiles;
    const int base = (t - npre_tiles) * kKT + RPWV * w + srow;
    const unsigned kdst = (unsigned)(uintptr_t)(lptr_t)(smem + (t & 1) * 2 * TILE_B + (RPWV * w) * ROWB);   // LDS byte address
    const unsigned vdst = kdst + TILE_B;
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
      const int rr = min(base + 4 * i, ext_len - 1);
      const int r = (RPWV * w + 4 * i + srow) & 15;   // the chunk the image keeps at this lane's position depends on row & 15 (K), row & 7 (V)
      const char* ks = (pre ? kpool + (int64_t)idn[i] * kpst : kext + (int64_t)rr * kest) + ((spos ^ r) << 4);
      const char* vs = (pre ? vpool + (int64_t)idn[i] * vpst : vext + (int64_t)rr * vest) + (((((spos >> 1) ^ (r & 7)) << 1) | (spos & 1)) << 4);
      ext_dma16(ks, kdst + i * 4 * ROWB);
      ext_dma16(vs, vdst + i * 4 * ROWB);
    }
  };

  auto stage_ext = [&](int t) {   // the same for a tile of new tokens only (npre_tiles == 0)
    const int base = t * kKT + RPWV * w + srow;
    const unsigned kdst = (unsigned)(uintptr_t)(lptr_t)(smem + (t & 1) * 2 * TILE_B + (RPWV * w) * ROWB);
    const unsigned vdst = kdst + TILE_B;
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
      const int rr = min(base + 4 * i, ext_len - 1);
      const int r = (RPWV * w + 4 * i + srow) & 15;
      ext_dma16(kext + (int64_t)rr * kest + ((spos ^ r) << 4), kdst + i * 4 * ROWB);
      ext_dma16(vext + (int64_t)rr * vest + (((((spos >> 1) ^ (r & 7)) << 1) | (spos & 1)) << 4), vdst + i * 4 * ROWB);
    }
  };

  float m_i[QT], l_i[QT];
  f32x4_t acc[QT][NT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_i[qt] = -INFINITY;
    l_i[qt] = 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[qt][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const float cs = p.sm_scale * kLog2e;

  // ---- prologue: tile 0 landed, slots of tile 1 known ----
  int idn[NIW];
  {
    int i0[NIW];
    load_idx(0, i0);
    load_idx(1, idn);
#pragma unroll
    for (int i = 0; i < NIW; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(i0[i]), "+v"(idn[i])::"memory");
    stage(0, i0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    // slots of tile t + 2 first, then tile t + 1's DMA (buffer (t + 1) & 1 held tile t - 1: its last readers passed the barrier)
    EXT_STAMP(0);
    int idn2[NIW];
    if (npre_tiles > 0) {   // (kernel-uniform: a batch without cached prefix needs neither pool slots nor address selects)
      load_idx(t + 2, idn2);
      if (t + 1 < ntiles) stage(t + 1, idn);
    } else {
#pragma unroll
      for (int i = 0; i < NIW; ++i) idn2[i] = 0;
      if (t + 1 < ntiles) stage_ext(t + 1);
    }
    EXT_STAMP(1);
    const char* kl = smem + (t & 1) * 2 * TILE_B;
    const char* vl = kl + TILE_B;
    const bool in_prefix = t < npre_tiles;
    const int kbase = in_prefix ? t * kKT : (t - npre_tiles) * kKT;
    const int klimit = in_prefix ? pre_len : ext_len;

    // ---- S^T = K Q^T: every K fragment of the tile in flight before the first MFMA; per (tt, qt) the k-steps in the order 0..3 ----
    f32x4_t s[QT][4];
    {
      vec8 kf[16];
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        const int row = 16 * (f >> 2) + a;
        kf[f] = *(const vec8*)(kl + row * ROWB + (((4 * (f & 3) + g) ^ (row & 15)) << 4));
      }
      __builtin_amdgcn_sched_barrier(0);   // (left free, the scheduler sinks each read to just before its MFMA: a round trip per pair)
      // k-steps outermost: eight independent accumulators between two MFMAs on the same one (per accumulator still ks = 0..3 in order)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt)
            s[qt][tt] = Tr::mfma16(kf[4 * tt + ks], qf[qt][ks], ks == 0 ? f32x4_t{0.f, 0.f, 0.f, 0.f} : s[qt][tt]);
    }
    EXT_STAMP(2);
    // ragged last tile of a phase / causal diagonal: ONE wave-uniform branch, branch-free inside
    const int wave_qmin = qpos0 + (bq >= 16 * QT ? ((16 * QT * w) & (bq - 1)) : 0);
    if ((kbase + kKT > klimit) || (!in_prefix && p.is_causal && kbase + kKT - 1 > wave_qmin)) {
      const bool causal = !in_prefix && p.is_causal;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const int qp = qpos0 + (((QT * w + qt) * 16) & (bq - 1)) + a;
        const int lim = min(klimit - 1, causal ? qp : 0x7fffffff) - kbase - 4 * g;   // key index relative to 16 tt + r
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[qt][tt][r] = (16 * tt + r <= lim) ? s[qt][tt][r] : -INFINITY;
      }
    }
    // ---- online softmax per query tile (extend_attn_kernel's arithmetic) ----
    vec8 pf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float m = -INFINITY;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, s[qt][tt][r]);
      m = pair32_max(pair16_max(m));   // two VALU lane swaps instead of two LDS round trips in the middle of the softmax's chain
      const float m_new = fmaxf(m_i[qt], m * cs);
      const float m_safe = fmaxf(m_new, -1e30f);
      const float alpha = __builtin_amdgcn_exp2f(m_i[qt] - m_safe);
      float lsum = 0.f;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][tt][r], cs, -m_safe));
          lsum += pv;
          pf[qt][tt >> 1][4 * (tt & 1) + r] = Tr::from_f32(pv);
        }
      }
      l_i[qt] = l_i[qt] * alpha + lsum;
      m_i[qt] = m_new;
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[qt][n] *= alpha;
    }
    EXT_STAMP(3);
    // ---- O^T += V^T P^T: every V^T fragment of the tile in flight before the first MFMA; per (qt, n) u = 0, 1 in order ----
    {
      vec8 vf[16];   // fragment 2 n + u
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        const int n = f >> 1, u = f & 1;
        s16x4_t t0, t1;
        {
          const int row = 32 * u + 4 * g + (a >> 2);
          t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4_t __attribute__((address_space(3)))*)(vl + row * ROWB + ((n ^ (row & 7)) << 5) + ((a & 3) << 3)));
        }
        {
          const int row = 32 * u + 16 + 4 * g + (a >> 2);
          t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4_t __attribute__((address_space(3)))*)(vl + row * ROWB + ((n ^ (row & 7)) << 5) + ((a & 3) << 3)));
        }
        vf[f] = __builtin_bit_cast(vec8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
      __builtin_amdgcn_sched_barrier(0);
      // k-steps outermost: sixteen independent accumulators between the two MFMAs of one (per accumulator still u = 0, 1 in order)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) acc[qt][n] = Tr::mfma16(vf[2 * n + u], pf[qt][u], acc[qt][n]);
    }
    EXT_STAMP(4);
    // tile t + 1 has landed and the slots of tile t + 2 are known (handed on THROUGH the wait so that nothing reads them above it)
#pragma unroll
    for (int i = 0; i < NIW; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(idn2[i])::"memory");
#pragma unroll
    for (int i = 0; i < NIW; ++i) idn[i] = idn2[i];
    __syncthreads();
    EXT_STAMP(5);
  }

  // ---- o = acc / l ----
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = l_i[qt];
    l = pair32_sum(pair16_sum(l));
    const int t16 = (QT * w + qt) * 16;
    const int hslot = t16 >> p.bq_log2;
    const int hl = hc * (4 * QT) + hslot;
    const int qp = qpos0 + (t16 & (bq - 1)) + a;
    if (hl < p.group && hslot < gslots && qp < ext_len) {
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      T* orow = (T*)p.o + (int64_t)(q0 + qp) * p.o_stride_t + (int64_t)(kh * p.group + hl) * D;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        typename Tr::vec4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = Tr::from_f32(acc[qt][n][r] * inv);
        *(typename Tr::vec4*)(orow + 16 * n + 4 * g) = ov;
      }
    }
  }
}

// Any-head-dim fallback: one wave per (query token, q head); correctness path for odd head sizes.
template <typename T>
__global__ __launch_bounds__(64) void extend_attn_generic(const ExtendParams p, int d_qk, int dv, int total_q) {
  __shared__ float p_lds[64];
  __shared__ int id_lds[64];
  const int tq = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  if (tq >= total_q) return;
  // locate the request of this query token
  int b = 0;
  if (p.qo_indptr) {
    while (b + 1 < p.bs && p.qo_indptr[b + 1] <= tq) ++b;
  } else {
    while (b + 1 < p.bs && p.extend_start_loc[b + 1] <= tq) ++b;
  }
  const int q0 = p.qo_indptr ? p.qo_indptr[b] : p.extend_start_loc[b];
  const int ext_len = p.qo_indptr ? p.qo_indptr[b + 1] - q0 : p.extend_seq_lens[b];
  const int qi = tq - q0;
  if (qi >= ext_len) return;
  int pre_len;
  const int32_t* idx_row;
  if (p.kv_indptr) {  // mode by indptr: kv_indices may legitimately be NULL when no request has a prefix
    const int s0 = p.kv_indptr[b];
    pre_len = p.kv_indptr[b + 1] - s0;
    idx_row = p.kv_indices + s0;
  } else {
    pre_len = (int)p.seq_lens[b] - ext_len;
    idx_row = p.req_to_token + p.req_pool_indices[b] * p.req_to_token_stride;
  }
  const int kh = h / p.group;
  const int nvis = p.is_causal ? qi + 1 : ext_len;
  const int total = pre_len + nvis;
  // custom mask (subsets of the causal triangle: a row's own key range is [0, qi]; the reference bounds the extend keys per
  // 64-row block instead, which only differs for mask bits above the diagonal) and sliding window, as in the MFMA kernel
  const uint8_t* mq = p.custom_mask ? p.custom_mask + p.mask_indptr[b] + (int64_t)qi * (pre_len + ext_len) : nullptr;
  const T* qrow = (const T*)p.q + (int64_t)tq * p.q_stride_t + (int64_t)h * d_qk;
  float accv[4] = {0.f, 0.f, 0.f, 0.f};
  float m_i = -INFINITY, l_i = 0.f;
  for (int t0 = 0; t0 < total; t0 += 64) {
    const int key = t0 + lane;
    const bool valid = key < total;
    const bool in_pre = key < pre_len;
    bool seen = valid;
    if (valid && in_pre && p.sliding_window > 0) seen = qi <= key + p.sliding_window;
    if (seen && mq && !(in_pre && p.skip_prefix_mask)) seen = mq[key] != 0;
    const int id = valid ? (in_pre ? idx_row[key] : key - pre_len) : 0;
    float sdot = 0.f;
    if (valid) {
      const T* krow = in_pre ? (const T*)p.k_buf + (int64_t)id * p.k_stride_t + (int64_t)kh * p.k_stride_h
                             : (const T*)p.ke + (int64_t)(q0 + id) * p.ke_stride_t + (int64_t)kh * d_qk;
      for (int d = 0; d < d_qk; ++d) sdot += (float)qrow[d] * (float)krow[d];
    }
    float xv = sdot * p.sm_scale;
    xv = (p.logit_cap > 0.f) ? softcap2(xv, p.logit_cap) : xv * kLog2e;
    xv = seen ? xv : -INFINITY;
    const float m_new = fmaxf(fmaxf(m_i, wave_reduce_max(xv)), -1e30f);   // (a tile may be masked out entirely)
    const float alpha = __builtin_amdgcn_exp2f(m_i - m_new);
    const float pv = __builtin_amdgcn_exp2f(xv - m_new);
    l_i = l_i * alpha + wave_reduce_sum(pv);
    m_i = m_new;
    __syncthreads();
    p_lds[lane] = (float)(T)pv;
    id_lds[lane] = valid ? (in_pre ? id : -(id + 1)) : 0;  // negative => row of v_extend
    __syncthreads();
    const int nt = min(64, total - t0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int d = lane + 64 * c;
      float av = accv[c] * alpha;
      if (d < dv) {
        for (int t = 0; t < nt; ++t) {
          const int e = id_lds[t];
          const T* vrow = e >= 0 && (t0 + t) < pre_len
                              ? (const T*)p.v_buf + (int64_t)e * p.v_stride_t + (int64_t)kh * p.v_stride_h
                              : (const T*)p.ve + (int64_t)(q0 + (-e - 1)) * p.ve_stride_t + (int64_t)kh * dv;
          av += p_lds[t] * (float)vrow[d];
        }
      }
      accv[c] = av;
    }
  }
  T* orow = (T*)p.o + (int64_t)tq * p.o_stride_t + (int64_t)h * dv;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int d = lane + 64 * c;
    if (d < dv) orow[d] = (T)(accv[c] / l_i);
  }
}

template <typename T, int D, bool KV8, bool CAP = false, bool MASKED = false, int QT = 2>
int launch_mfma(ExtendParams& p, int max_len_extend, hipStream_t st) {
  if constexpr (!CAP && !MASKED && QT == 2) {
    if (p.custom_mask != nullptr || p.sliding_window > 0) return launch_mfma<T, D, KV8, false, true>(p, max_len_extend, st);
    if (p.casc_bs > 0) return p.logit_cap > 0.0f ? launch_mfma<T, D, KV8, true, false, 1>(p, max_len_extend, st)
                                                 : launch_mfma<T, D, KV8, false, false, 1>(p, max_len_extend, st);
    if (p.logit_cap > 0.0f) return launch_mfma<T, D, KV8, true>(p, max_len_extend, st);
  }
  constexpr int smem = 2 * 2 * kKT * D * 2;
  constexpr bool kDma = !KV8 && !CAP && !MASKED && QT == 2 && D == 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)extend_attn_kernel<T, D, KV8, CAP, MASKED, QT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if constexpr (kDma) {
      (void)hipFuncSetAttribute((const void*)extend_attn_dma_kernel<T, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      (void)hipFuncSetAttribute((const void*)extend_attn_dma_kernel<T, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    attr_set = true;
  }
  // head slots per workgroup: smallest power of two >= min(group, 4 QT); positions per workgroup = 64 QT / slots (>= 16)
  int slots = 1;
  while (slots < p.group && slots < 4 * QT) slots <<= 1;
  int bq = 64 * QT / slots, lg = 0;
  while ((1 << lg) < bq) ++lg;
  p.bq_log2 = lg;
  p.hchunks = (p.group + 4 * QT - 1) / (4 * QT);
  p.nqb = (max_len_extend + bq - 1) / bq;
  const int npairs = p.bs * p.hkv * p.hchunks;
  const int64_t nblocks = (int64_t)((npairs + 7) / 8) * p.nqb * 8;
  if (nblocks <= 0) return SGL_MI355_OK;
  if (nblocks >= (1ll << 31)) {
    snprintf(g_sgl_mi355_err, sizeof(g_sgl_mi355_err), "extend_attention: grid too large");
    return SGL_MI355_EINVAL;
  }
  if constexpr (kDma) {
    // The 8-wave 32x32x16 kernel (extend_attention_phased.hip; 256-row workgroups, one per CU) where it applies and the launch still
    // has a workgroup for most CUs: same box, 4-wave LDS-DMA kernel -> this one: 32 x 2048 1 493 -> 1 384 us, 8 x 2048 441 -> 426,
    // radix hit 16 x (1536 + 512) 333 -> 318, 2 x 8192 1 219 -> 1 148 (profiles/round5_ab_extend_phased.json)
    if ((g_extend_dma == 1 || g_extend_dma == 5) && extend_phased_eligible(p)) {
      int slots = 1;
      while (slots < p.group) slots <<= 1;
      const int64_t nbp = (int64_t)((p.bs * p.hkv + 7) / 8) * ((max_len_extend + 256 / slots - 1) / (256 / slots)) * 8;
      if (g_extend_dma == 5 || nbp >= 192)
        return launch_extend_phased(p, max_len_extend, std::is_same<T, __bf16>::value ? SGL_BF16 : SGL_F16, st);
    }
    if (g_extend_dma && p.casc_bs == 0 && !p.kv_fp8) {
      // 8 waves (twice the query positions per K / V tile, half the tile traffic) pay where a query block walks many keys: same
      // box, 4 -> 8 waves: radix hit 16 x (1536 + 512) 367 -> 334 us, 2 x 8192 1 333 -> 1 238 us, 32 x 2048 without prefix
      // 1 600 -> 1 614 us.  Mean keys per query block: the caller's hint (prefix + extend / 2 per request), else extend / 2.
      const int mean_keys = g_extend_kv_hint > 0 ? g_extend_kv_hint : max_len_extend / 2;
      const bool wide = g_extend_dma == 2 || (g_extend_dma == 1 && mean_keys >= 1536);
      if (wide && p.group <= 8 && max_len_extend > 2 * bq) {
        p.bq_log2 = lg + 1;
        p.nqb = (max_len_extend + 2 * bq - 1) / (2 * bq);
        const int64_t nb8 = (int64_t)((npairs + 7) / 8) * p.nqb * 8;
        hipLaunchKernelGGL((extend_attn_dma_kernel<T, 8>), dim3((unsigned)nb8), dim3(512), smem, st, p);
      } else {
        hipLaunchKernelGGL((extend_attn_dma_kernel<T, 4>), dim3((unsigned)nblocks), dim3(256), smem, st, p);
      }
      SGL_HIP_LAUNCH_CHECK();
      return SGL_MI355_OK;
    }
  }
  hipLaunchKernelGGL((extend_attn_kernel<T, D, KV8, CAP, MASKED, QT>), dim3((unsigned)nblocks), dim3(256), smem, st, p);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

template <typename T>
int launch_all(ExtendParams& p, int d_qk, int dv, int total_q, int max_len_extend, hipStream_t st) {
  if (d_qk == dv && d_qk == 128) return p.kv_fp8 ? launch_mfma<T, 128, true>(p, max_len_extend, st) : launch_mfma<T, 128, false>(p, max_len_extend, st);
  if (d_qk == dv && d_qk == 64) return p.kv_fp8 ? launch_mfma<T, 64, true>(p, max_len_extend, st) : launch_mfma<T, 64, false>(p, max_len_extend, st);
  hipLaunchKernelGGL((extend_attn_generic<T>), dim3(total_q, p.hq), dim3(64), 0, st, p, d_qk, dv, total_q);
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}

}  // namespace

// Cascade prefix pass (called by sgl_mi355_decode_attention_cascade, decode_attention.hip): see ExtendParams::casc_*.
// Returns the number of prefix split slots written through *splits_out (slots [slot0, slot0 + splits)).
int sgl_mi355_internal_cascade_prefix(const void* q, int64_t q_stride_t, const void* k_buffer, const void* v_buffer, int64_t k_stride_t,
                                      int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h, const int32_t* prefix_indices,
                                      int prefix_len, int chunk, int splits, int slot0, float* attn_logits, float* attn_lse,
                                      int max_kv_splits, int batch, int num_q_heads, int num_kv_heads, int head_dim, float sm_scale,
                                      float logit_cap, int dtype, int kv_dtype, float k_scale, float v_scale, hipStream_t st) {
  ExtendParams p{};
  p.q = q; p.ke = nullptr; p.ve = nullptr; p.o = nullptr;
  p.q_stride_t = q_stride_t; p.ke_stride_t = 0; p.ve_stride_t = 0; p.o_stride_t = 0;
  p.k_buf = k_buffer; p.v_buf = v_buffer;
  p.k_stride_t = k_stride_t; p.k_stride_h = k_stride_h; p.v_stride_t = v_stride_t; p.v_stride_h = v_stride_h;
  p.kv_indices = prefix_indices;
  p.bs = splits; p.hq = num_q_heads; p.hkv = num_kv_heads; p.group = num_q_heads / num_kv_heads;
  p.sm_scale = sm_scale; p.logit_cap = logit_cap; p.is_causal = 0;
  const bool kv8 = kv_dtype == SGL_FP8_E4M3;
  p.kv_fp8 = kv8 ? 1 : 0; p.k_scale = kv8 ? k_scale : 1.0f; p.v_scale = kv8 ? v_scale : 1.0f;
  p.nqb = 0; p.bq_log2 = 0; p.hchunks = 1;
  p.casc_bs = batch; p.casc_prefix_len = prefix_len; p.casc_chunk = chunk; p.casc_slot0 = slot0; p.max_kv_splits = max_kv_splits;
  p.part_o = attn_logits; p.part_lse = attn_lse;
  return dtype == SGL_BF16 ? launch_all<__bf16>(p, head_dim, head_dim, batch, batch, st) : launch_all<_Float16>(p, head_dim, head_dim, batch, batch, st);
}

// measurement / test hook: 0 = always the register-staged kernel, 1 (default) = the 8-wave 32x32x16 kernel where it applies and the grid
// has >= 192 workgroups, else the LDS-DMA kernel with 4 or 8 waves per workgroup by the mean-keys rule, 2 / 3 = always the LDS-DMA kernel
// with 8 / 4 waves, 5 = always the 32x32x16 kernel where it applies (4 was round 4's 64-rows-per-wave kernel: removed, 0.6 x the default)
extern "C" int sgl_mi355_extend_attention_set_mode(int mode) {
  g_extend_dma = mode < 0 ? 0 : (mode > 5 ? 5 : (mode == 4 ? 1 : mode));
  return SGL_MI355_OK;
}
// host-side knowledge the kernel arguments do not carry: the mean number of keys a query block of the NEXT calls attends to
// (per request: prefix + extend / 2 when causal); 0 = unknown (the launcher then assumes no prefix)
extern "C" int sgl_mi355_extend_attention_set_kv_hint(int mean_keys_per_query_block) {
  g_extend_kv_hint = mean_keys_per_query_block > 0 ? mean_keys_per_query_block : 0;
  return SGL_MI355_OK;
}

#ifdef SGL_EXT_TIMELINE
extern "C" int sgl_mi355_extend_attention_debug_timeline(long long* buf) {
  g_ext_tl = buf;
  return SGL_MI355_OK;
}
#endif

extern "C" int sgl_mi355_extend_attention(
    const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend, int64_t q_stride_t,
    int64_t k_stride_t_ext, int64_t v_stride_t_ext, int64_t o_stride_t, const void* k_buffer, const void* v_buffer,
    int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h, const int32_t* qo_indptr,
    const int32_t* kv_indptr, const int32_t* kv_indices, const int32_t* req_to_token, int64_t req_to_token_stride,
    const int64_t* req_pool_indices, const int64_t* seq_lens, const int32_t* extend_seq_lens,
    const int32_t* extend_start_loc, int batch, int total_q_tokens, int max_len_extend, int num_q_heads,
    int num_kv_heads, int head_dim, int v_head_dim, float sm_scale, float logit_cap, int is_causal, int dtype,
    int kv_dtype, float k_scale, float v_scale, const uint8_t* custom_mask, const int64_t* mask_indptr,
    int skip_prefix_custom_mask, int sliding_window_size, void* stream) {
  SGL_CHECK(batch >= 0 && total_q_tokens >= 0, "extend_attention: negative sizes");
  SGL_CHECK(custom_mask == nullptr || (mask_indptr != nullptr && qo_indptr != nullptr),
            "extend_attention: custom_mask needs mask_indptr (and the qo_indptr / kv_indptr addressing mode)");
  if (batch == 0 || total_q_tokens == 0 || max_len_extend <= 0) return SGL_MI355_OK;
  SGL_CHECK(q_extend && k_extend && v_extend && o_extend, "extend_attention: null tensor pointer");
  SGL_CHECK((qo_indptr && kv_indptr) || (req_to_token && req_pool_indices && seq_lens && extend_seq_lens && extend_start_loc),
            "extend_attention: need (qo_indptr, kv_indptr, kv_indices) or (req_to_token, req_pool_indices, seq_lens, "
            "extend_seq_lens, extend_start_loc)");
  SGL_CHECK(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0, "extend_attention: Hq=%d not a multiple of Hkv=%d",
            num_q_heads, num_kv_heads);
  SGL_CHECK(head_dim > 0 && head_dim <= 256 && v_head_dim > 0 && v_head_dim <= 256,
            "extend_attention: head dims (%d, %d) outside (0, 256]", head_dim, v_head_dim);
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "extend_attention: dtype code %d unsupported (bf16=0, f16=1)", dtype);
  SGL_CHECK(kv_dtype == dtype || kv_dtype == SGL_FP8_E4M3, "extend_attention: kv_dtype %d must be the q dtype or fp8_e4m3", kv_dtype);
  const bool kv8 = kv_dtype == SGL_FP8_E4M3 && k_buffer != nullptr;
  SGL_CHECK(!kv8 || (head_dim == v_head_dim && (head_dim == 128 || head_dim == 64)),
            "extend_attention: the fp8 KV cache needs head_dim == v_head_dim in {64, 128} (got %d, %d)", head_dim, v_head_dim);
  if (head_dim == v_head_dim && (head_dim == 128 || head_dim == 64)) {
    SGL_CHECK(q_stride_t % 8 == 0 && k_stride_t_ext % 8 == 0 && v_stride_t_ext % 8 == 0 && o_stride_t % 4 == 0 &&
                  k_stride_t % 8 == 0 && v_stride_t % 8 == 0 && k_stride_h % 8 == 0 && v_stride_h % 8 == 0 &&
                  ((uintptr_t)q_extend % 16) == 0 && ((uintptr_t)k_extend % 16) == 0 && ((uintptr_t)v_extend % 16) == 0 &&
                  ((uintptr_t)o_extend % 8) == 0 && (!k_buffer || ((uintptr_t)k_buffer % 16) == 0) &&
                  (!v_buffer || ((uintptr_t)v_buffer % 16) == 0),
              "extend_attention: rows must be 16-byte aligned for the MFMA path");
  }
  ExtendParams p;
  p.q = q_extend; p.ke = k_extend; p.ve = v_extend; p.o = o_extend;
  p.q_stride_t = q_stride_t; p.ke_stride_t = k_stride_t_ext; p.ve_stride_t = v_stride_t_ext; p.o_stride_t = o_stride_t;
  // with no prefix anywhere the pool may be absent: point at the extend tensors (never dereferenced for real rows)
  p.k_buf = k_buffer ? k_buffer : k_extend; p.v_buf = v_buffer ? v_buffer : v_extend;
  p.k_stride_t = k_stride_t; p.k_stride_h = k_stride_h; p.v_stride_t = v_stride_t; p.v_stride_h = v_stride_h;
  p.qo_indptr = qo_indptr; p.kv_indptr = kv_indptr; p.kv_indices = kv_indices;
  p.req_to_token = req_to_token; p.req_to_token_stride = req_to_token_stride; p.req_pool_indices = req_pool_indices;
  p.seq_lens = seq_lens; p.extend_seq_lens = extend_seq_lens; p.extend_start_loc = extend_start_loc;
  p.bs = batch; p.hq = num_q_heads; p.hkv = num_kv_heads; p.group = num_q_heads / num_kv_heads;
  p.sm_scale = sm_scale; p.logit_cap = logit_cap; p.is_causal = is_causal;
  p.kv_fp8 = kv8 ? 1 : 0; p.k_scale = kv8 ? k_scale : 1.0f; p.v_scale = kv8 ? v_scale : 1.0f;
  p.nqb = 0; p.bq_log2 = 0; p.hchunks = 1;
  p.custom_mask = custom_mask; p.mask_indptr = mask_indptr; p.skip_prefix_mask = skip_prefix_custom_mask ? 1 : 0;
  p.sliding_window = sliding_window_size > 0 ? sliding_window_size : 0;
#ifdef SGL_EXT_TIMELINE
  p.tl = g_ext_tl;
#endif
  hipStream_t st = (hipStream_t)stream;
  return dtype == SGL_BF16 ? launch_all<__bf16>(p, head_dim, v_head_dim, total_q_tokens, max_len_extend, st)
                           : launch_all<_Float16>(p, head_dim, v_head_dim, total_q_tokens, max_len_extend, st);
}
