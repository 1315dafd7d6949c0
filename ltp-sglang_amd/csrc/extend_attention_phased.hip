// Extend (prefill / chunked-prefill / RadixAttention-hit) attention, gfx950: the PHASED 8-wave kernel (round 5).
//
// Same contract as extend_attn_kernel / extend_attn_dma_kernel (extend_attention.hip): replaces extend_attention_fwd
// (python/sglang/srt/layers/attention/triton_ops/extend_attention.py:41-438) for 16-bit K / V, D = 128, head group <= 8, no custom
// mask / window / logit cap / cascade / fp8 pool; semantics of torch_native_backend.py:27-110.
//
// Why another structure.  The 16x16x32 kernels run two waves per SIMD that execute the SAME phase at the same time (round-4 counters:
// vector and matrix instructions co-execute in 16 % of the MFMA-busy cycles, MFMA busy 0.32): both in their MFMA block, then both in
// their softmax chain, and 128-row workgroups ingest 64 KiB of K / V per CU and tile step.  This kernel:
//   * 256 (head, position) rows per workgroup = 8 waves x 32 rows, ONE workgroup per CU: 32 KiB of K / V per CU and tile step;
//   * v_mfma_f32_32x32x16 with the QUERY on the lane: S^T = K Q^T leaves 32 scores of one query row in a lane's registers (row
//     maximum = a register chain + ONE v_permlane32_swap; the running maximum, the row sum and the rescale factor are per-lane
//     scalars), and registers 8 s .. 8 s + 7 of the S^T accumulator, converted pairwise, ARE the B operand of k-step s of
//     O^T += V^T P^T in the permuted key order 16 s + 8 (j >> 2) + 4 h + (j & 3) (cdna_hip_programming.md, "An accumulator tile as
//     the next MFMA's operand"); V^T fragments take the same order by ds_read_b64_tr_b16;
//   * FOUR CLUSTERS per 64-key tile, separated by s_barrier, with waves 4-7 (the SIMD partners of waves 0-3) one cluster behind:
//       C1  K fragments LDS -> registers, LDS-DMA of the next tile's K
//       C2  16 MFMAs  S^T = K Q^T                       (operands in registers: no LDS access)
//       C3  V^T fragments LDS -> registers (they take over the K fragments' registers), LDS-DMA of the next tile's V, online softmax
//       C4  16 MFMAs  O^T += V^T P^T
//     so whenever a wave is in a matrix cluster its SIMD partner is in a load / softmax cluster (MI355X_MICROARCH.md, "Two waves per
//     SIMD"): the matrix pipe always has exactly one client per SIMD and the VALU work of the softmax runs beside the partner's MFMAs;
//   * deferred rescale: exponentials are taken against a reference maximum that follows the true one only after it moved by 2^6
//     (o = acc / l does not depend on the reference point; power-of-two scaling is exact in bf16 / f16 / f32), so the 64 accumulator
//     registers are touched by the VALU once per few dozen tiles;
//   * K / V tiles by LDS-DMA one tile ahead (double buffer), issued from asm and waited for by counted vmcnt; new-token tiles from a
//     scalar base + constant lane offsets (no per-lane address arithmetic in the loop);
//   * O leaves through LDS as whole 256-byte rows (16-byte stores of full lines instead of 8-byte stores at a row stride).
// LDS images, source-side swizzles and fragment maps are those of the round-4 64-rows-per-wave kernel (verified against the oracle
// there): K rows of 256 B with 16-byte chunk ^ (row & 15); V rows with 32-byte chunk ^ 2 (row & 3).
#include "extend_params.h"

namespace {

typedef __attribute__((address_space(3))) void* ph_lptr_t;

// 64 lanes x 16 B -> LDS [lds, lds + 1 KiB); source = scalar base + per-lane 32-bit offset
__device__ __forceinline__ void ph_dma_s(unsigned voff, const char* sbase, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}
// the same with a per-lane 64-bit source address (pool rows)
__device__ __forceinline__ void ph_dma_v(const char* vaddr, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vaddr), "s"(lds) : "memory");
}
__device__ __forceinline__ int ph_load_i32(const int32_t* src) {
  int v;
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(src) : "memory");
  return v;
}
#define PH_BARRIER()                        \
  do {                                      \
    __builtin_amdgcn_sched_barrier(0);      \
    __builtin_amdgcn_s_barrier();           \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)
#define PH_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

#ifdef SGL_EXT_TIMELINE
// tools/debug/ext_timeline.py: s_memtime stamps of workgroup SGL_EXT_TIMELINE's eight waves, first 24 tiles, 8 stamps per tile
#define PH_STAMP(k)                                                                                                          \
  do {                                                                                                                       \
    if (tl && t < 24) {                                                                                                      \
      unsigned long long ts_;                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                                     \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_)::"memory");                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                                     \
      tl[((w * 24 + t) * 8) + (k)] = (long long)ts_;                                                                         \
    }                                                                                                                        \
  } while (0)
#else
#define PH_STAMP(k) do { } while (0)
#endif

constexpr float kPhDefer = 6.0f;   // log2 units: the reference maximum is moved when the true one exceeds it by more than this

template <typename T>
__global__ __launch_bounds__(512, 2) void extend_attn_phased_kernel(const ExtendParams p) {
  using Tr = ElemTraits<T>;
  using vec8 = typename Tr::vec8;
  constexpr int ROWB = 256, KS = 8, DB = 4, KB = 2, D = 128;
  constexpr int TILE_B = kKT * ROWB;  // 16 KiB
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [2 buffers][K 16 KiB | V 16 KiB]; the O rows at the end (1 KiB aligned: the fragment addresses XOR into bits 5-7)

  // ---- block -> (request, kv head, q block); same (request, kv head) => same blockIdx % 8 (one XCD's L2 serves the K / V re-reads) ----
  const int bid = blockIdx.x;
  const int lo = bid & 7, rest = bid >> 3;
  const int qb_idx = p.nqb - 1 - rest % p.nqb;   // longest key ranges first
  const int pair = (rest / p.nqb) * 8 + lo;
  if (pair >= p.bs * p.hkv) return;
  const int b = pair / p.hkv;
  const int kh = pair - b * p.hkv;

  const int bq = 1 << p.bq_log2;              // query positions per workgroup = 256 / head slots
  const int q0 = p.qo_indptr ? p.qo_indptr[b] : p.extend_start_loc[b];
  const int ext_len = p.qo_indptr ? p.qo_indptr[b + 1] - q0 : p.extend_seq_lens[b];
  const int qpos0 = qb_idx * bq;
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
  const bool grpB = w >= 4;                   // the SIMD partners of waves 0-3: one cluster behind
  const int c = lane & 31, h = lane >> 5;     // MFMA column (query row of the wave) and lane half
  const int gq = lane >> 4, a = lane & 15;    // 16-lane group / lane in group (transposed reads, DMA rows)
#ifdef SGL_EXT_TIMELINE
  long long* tl = (blockIdx.x == SGL_EXT_TIMELINE && lane == 0) ? p.tl : nullptr;
#endif

  // ---- this wave's 32 rows: head slot hl of the group, positions wpos0 .. wpos0 + 31 of the extend part ----
  const int t32 = 32 * w;
  const int hl = t32 >> p.bq_log2;
  const int wpos0 = qpos0 + (t32 & (bq - 1));
  const int qpos = wpos0 + c;
  const bool wave_on = hl < p.group && wpos0 < ext_len;   // wave-uniform: an idle head slot / a block past the end only moves data
  const bool qok = wave_on && qpos < ext_len;

  // Q^T fragments (B operand): lane (c, h) holds Q[row c][16 ks + 8 h .. + 7].  Rows past the end / idle head slots read a clamped
  // (valid) row: a query row is one MFMA column from S^T to O^T, so what they compute touches no other row, and they are not stored.
  vec8 qf[KS];
  {
    const T* qrow = (const T*)p.q + (int64_t)(q0 + min(qpos, ext_len - 1)) * p.q_stride_t + (int64_t)(kh * p.group + min(hl, p.group - 1)) * D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const vec8*)(qrow + 16 * ks + 8 * h);
  }

  const int npre_tiles = (pre_len + kKT - 1) / kKT;
  const int ext_end = p.is_causal ? min(ext_len, qpos0 + bq) : ext_len;
  const int ntiles = npre_tiles + (ext_end + kKT - 1) / kKT;   // >= 1

  // ---- LDS-DMA staging: wave w fills rows 8 w .. 8 w + 7 of a tile, 4 rows (1 KiB) per instruction; lane = (row l >> 4, position l & 15).
  // Only the two offsets per operand of the common case (a whole tile of new tokens: scalar base + constant lane offset) stay in
  // registers across the loop; the pool-row and ragged-tile paths rebuild theirs from an opaque copy of the lane id (hoisted, such
  // tables cost a dozen registers the loop does not have).
  const char* kpool = (const char*)p.k_buf + ((int64_t)kh * p.k_stride_h) * 2;
  const char* vpool = (const char*)p.v_buf + ((int64_t)kh * p.v_stride_h) * 2;
  const char* kext = (const char*)p.ke + ((int64_t)q0 * p.ke_stride_t + (int64_t)kh * D) * 2;
  const char* vext = (const char*)p.ve + ((int64_t)q0 * p.ve_stride_t + (int64_t)kh * D) * 2;
  const int64_t kpst = p.k_stride_t * 2, vpst = p.v_stride_t * 2;
  const unsigned kest = (unsigned)(p.ke_stride_t * 2), vest = (unsigned)(p.ve_stride_t * 2);
  const int32_t* idx_dummy = p.qo_indptr ? p.qo_indptr : p.extend_start_loc;   // any readable int32 when a tile has no pool rows
  const unsigned lds_base = (unsigned)(uintptr_t)(ph_lptr_t)smem;
  // row of the tile lane l fetches a piece of (instruction i), the chunk the K / V image keeps at its position
  auto dma_row = [&](int l, int i) { return 8 * w + 4 * i + (l >> 4); };
  auto dma_kch = [&](int l, int i) { return (unsigned)(((l & 15) ^ (dma_row(l, i) & 15)) << 4); };
  auto dma_vch = [&](int l) { return (unsigned)((((((l & 15) >> 1) ^ (((l >> 4) & 3) << 1)) << 1) | (l & 1)) << 4); };
  unsigned kvo[2], vvo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    kvo[i] = (unsigned)dma_row(lane, i) * kest + dma_kch(lane, i);
    vvo[i] = (unsigned)dma_row(lane, i) * vest + dma_vch(lane);
  }

  auto load_idx = [&](int tn, int (&idn)[2]) {   // pool slots of tile tn's rows (valid only after the caller's wait); always 2 loads
    if (tn < npre_tiles) {
      int lo_ = lane;
      asm volatile("" : "+v"(lo_));
#pragma unroll
      for (int i = 0; i < 2; ++i) idn[i] = ph_load_i32(idx_row + min(tn * kKT + dma_row(lo_, i), pre_len - 1));
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) idn[i] = ph_load_i32(idx_dummy);
    }
  };
  auto stage = [&](int tn, const int (&idn)[2], bool isv) {   // K or V of tile tn -> buffer tn & 1; always 2 LDS-DMA instructions
    const unsigned dst = lds_base + (tn & 1) * 2 * TILE_B + (isv ? TILE_B : 0) + (8 * w) * ROWB;
    const int te = tn - npre_tiles;
    if (te >= 0 && te * kKT + kKT <= ext_len) {   // a whole tile of new tokens
      const unsigned est = isv ? vest : kest;
      const char* sb = (isv ? vext : kext) + (int64_t)te * kKT * est;
#pragma unroll
      for (int i = 0; i < 2; ++i) ph_dma_s(isv ? vvo[i] : kvo[i], sb, dst + i * 4 * ROWB);
    } else {
      int lo_ = lane;
      asm volatile("" : "+v"(lo_));
      if (te < 0) {   // pool rows through their slots
        const char* pool = isv ? vpool : kpool;
        const int64_t pst = isv ? vpst : kpst;
#pragma unroll
        for (int i = 0; i < 2; ++i) ph_dma_v(pool + (int64_t)idn[i] * pst + (isv ? dma_vch(lo_) : dma_kch(lo_, i)), dst + i * 4 * ROWB);
      } else {   // ragged last tile: rows past the end re-read the last row (finite data; their probabilities are exactly 0)
        const unsigned est = isv ? vest : kest;
        const char* sb = (isv ? vext : kext) + (int64_t)te * kKT * est;
        const int last = ext_len - 1 - te * kKT;
#pragma unroll
        for (int i = 0; i < 2; ++i)
          ph_dma_s((unsigned)min(dma_row(lo_, i), last) * est + (isv ? dma_vch(lo_) : dma_kch(lo_, i)), sb, dst + i * 4 * ROWB);
      }
    }
  };

  // fragment read addresses: ONE lane-constant register per operand; the buffer's base is added per tile into an opaque copy so that
  // the eight / four XOR variants are rebuilt per tile (one VALU each) instead of living in a dozen registers:
  //   K fragment (kk, ks): row 32 kk + c, chunk (2 ks + h) ^ (c & 15)           = (klane ^ 32 ks) + 8192 kk
  //   V^T read (n, sI, half): row 16 sI + 8 half + 4 h + (a >> 2), 32-byte chunk (2 n + (gq & 1)) ^ 2 ((a >> 2) & 3), piece a & 3
  //                                                                               = (vlane ^ 64 n) + 4096 sI + 2048 half
  const unsigned klane = (unsigned)(c * ROWB + ((c & 14) << 4) + ((h ^ (c & 1)) << 4));
  const unsigned vlane = (unsigned)((4 * h + (a >> 2)) * ROWB + (((a >> 2) & 3) << 6) + ((gq & 1) << 5) + ((a & 3) << 3));
  typedef const vec8 __attribute__((address_space(3)))* kptr_t;
  typedef s16x4_t __attribute__((address_space(3)))* vptr_t;

  float m_i = -INFINITY, l_i = 0.f;   // reference maximum (log2 units); this lane half's share of the row sum
  f32x16_t acc[DB];
#pragma unroll
  for (int n = 0; n < DB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const float cs = p.sm_scale * kLog2e;

  // ---- prologue: tile 0 landed, slots of tile 1 known ----
  int idc[2];
  {
    int i0[2];
    load_idx(0, i0);
    load_idx(min(1, ntiles - 1), idc);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(i0[0]), "+v"(i0[1]), "+v"(idc[0]), "+v"(idc[1])::"memory");
    stage(0, i0, false);
    stage(0, i0, true);
  }
  PH_VMCNT(0);
  __syncthreads();
  if (grpB) PH_BARRIER();   // waves 4-7 run one cluster behind waves 0-3 from here on

  for (int t = 0; t < ntiles; ++t) {
    const int tn = min(t + 1, ntiles - 1);
    unsigned kb = lds_base + (t & 1) * 2 * TILE_B + klane, vb = lds_base + (t & 1) * 2 * TILE_B + TILE_B + vlane;
    asm volatile("" : "+v"(kb), "+v"(vb));
    const bool in_prefix = t < npre_tiles;
    const int kbase = in_prefix ? t * kKT : (t - npre_tiles) * kKT;
    const int klimit = in_prefix ? pre_len : ext_len;
    const bool causal = !in_prefix && p.is_causal;
    const bool active = wave_on && !(causal && kbase > wpos0 + 31);   // wave-uniform: tiles in the causal future of every row are skipped
    const bool need_mask = (kbase + kKT > klimit) || (causal && kbase + kKT - 1 > wpos0);

    // ================= C1: the first K fragments -> registers; DMA of the next tile's K =================
    PH_STAMP(0);
    auto kread = [&](int kk, int ks) -> vec8 { return *(kptr_t)(uintptr_t)((kb ^ (unsigned)(32 * ks)) + 8192u * kk); };
    auto vread = [&](int n, int sI) -> vec8 {   // V^T fragment of d block n, k-step sI: keys 16 sI + 4 h .. + 3 and 16 sI + 8 + 4 h .. + 3
      const unsigned ad = (vb ^ (unsigned)(64 * n)) + 4096u * sI;
      const s16x4_t t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((vptr_t)(uintptr_t)ad);
      const s16x4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((vptr_t)(uintptr_t)(ad + 2048u));
      return __builtin_bit_cast(vec8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // fragment f of S^T = K Q^T: k-step f >> 1, key block f & 1 (the two MFMAs of a k-step share their Q^T fragment); fragment g of
    // O^T += V^T P^T: k-step g >> 2, d block g & 3.  A load cluster requests the first LEAD fragments, the matrix cluster requests
    // fragment f + LEAD right behind MFMA f: LEAD x 4 registers instead of 64 hold the fragments, and an LDS round trip has LEAD - 1
    // MFMAs (32 cycles each) to complete.
    constexpr int LEAD = 6;
    vec8 kf[16];
    if (active) {
#pragma unroll
      for (int f = 0; f < LEAD; ++f) kf[f] = kread(f & 1, f >> 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    stage(tn, idc, false);
    if (grpB) PH_VMCNT(2);
    PH_STAMP(1);
    PH_BARRIER();
    PH_STAMP(2);

    // ================= C2: S^T = K Q^T; the later fragments stream in behind the MFMAs =================
    f32x16_t sq[KB];
    if (active) {
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        const int kk = f & 1, ks = f >> 1;
        if (ks == 0) {
          f32x16_t z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          sq[kk] = Tr::mfma32(kf[f], qf[ks], z);
        } else {
          sq[kk] = Tr::mfma32(kf[f], qf[ks], sq[kk]);
        }
        if (f + LEAD < 16) kf[f + LEAD] = kread((f + LEAD) & 1, (f + LEAD) >> 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!grpB) PH_VMCNT(2);
    PH_STAMP(3);
    PH_BARRIER();
    PH_STAMP(4);

    // ================= C3: the first V^T fragments -> registers; slots of tile t + 2, DMA of the next tile's V; online softmax =================
    vec8 vf[16];
    if (active) {
#pragma unroll
      for (int g = 0; g < LEAD; ++g) vf[g] = vread(g & 3, g >> 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    int idn[2];
    load_idx(min(t + 2, ntiles - 1), idn);
    stage(tn, idc, true);
    __builtin_amdgcn_sched_barrier(0);
    vec8 pq[4];
    if (active) {
      if (need_mask) {
        const int lim = min(klimit - 1, causal ? qpos : 0x7fffffff) - kbase - 4 * h;   // key index relative to 32 kk + (r & 3) + 8 (r >> 2)
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
#pragma unroll
          for (int r = 0; r < 16; ++r) sq[kk][r] = (32 * kk + (r & 3) + 8 * (r >> 2) <= lim) ? sq[kk][r] : -INFINITY;
      }
      float m = -INFINITY;
#pragma unroll
      for (int kk = 0; kk < KB; ++kk)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, sq[kk][r]);
      m = pair32_max(m) * cs;
      // (version 0: the reference maximum follows the true one every tile and O^T is rescaled in straight line -- as a rare
      // wave-uniform branch the rescale made the compiler keep a second 64-register copy of O^T across it)
      {
        const float m_new = fmaxf(m_i, m);
        const float alpha = __builtin_amdgcn_exp2f(m_i - fmaxf(m_new, -1e30f));
        l_i *= alpha;
        m_i = m_new;
#pragma unroll
        for (int n = 0; n < DB; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[n][r] *= alpha;
      }
      const float m_safe = fmaxf(m_i, -1e30f);
      float lsum = 0.f;
#pragma unroll
      for (int kk = 0; kk < KB; ++kk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sq[kk][r], cs, -m_safe));
          lsum += pv;
          pq[2 * kk + (r >> 3)][r & 7] = Tr::from_f32(pv);   // registers 8 sI .. 8 sI + 7 = the B operand of PV k-step 2 kk + sI
        }
      l_i += lsum;
    }
    if (grpB) PH_VMCNT(4);
    PH_STAMP(5);
    PH_BARRIER();
    PH_STAMP(6);

    // ================= C4: O^T += V^T P^T; the later fragments stream in behind the MFMAs =================
    if (active) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        acc[g & 3] = Tr::mfma32(vf[g], pq[g >> 2], acc[g & 3]);
        if (g + LEAD < 16) vf[g + LEAD] = vread((g + LEAD) & 3, (g + LEAD) >> 2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the next tile's K has landed (waves 0-3; waves 4-7 waited at the end of C3) and the slots of tile t + 2 are known; they are
    // handed on THROUGH the wait so that nothing reads them above it
    asm volatile("s_waitcnt vmcnt(2)" : "+v"(idn[0]), "+v"(idn[1])::"memory");
    idc[0] = idn[0];
    idc[1] = idn[1];
    PH_STAMP(7);
    PH_BARRIER();
  }
  if (!grpB) PH_BARRIER();   // waves 0-3: the barrier waves 4-7 spent on their lag
  PH_VMCNT(0);               // (the clamped re-fetches of the last tile)
  PH_BARRIER();

  // ---- o = acc / l through LDS: wave w owns bytes [8 KiB w, 8 KiB (w + 1)) = its 32 rows x 256 B, 16-byte chunk ^ (row & 15) ----
  {
    const float l = pair32_sum(l_i);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    char* orw = smem + w * (32 * ROWB);
#pragma unroll
    for (int n = 0; n < DB; ++n)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {   // d = 32 n + 8 g4 + 4 h + r: chunk 4 n + g4, byte 8 h of it
        typename Tr::vec4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = Tr::from_f32(acc[n][4 * g4 + r] * inv);
        *(typename Tr::vec4*)(orw + c * ROWB + (((4 * n + g4) ^ (c & 15)) << 4) + 8 * h) = ov;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (LDS executes a wave's accesses in order; the clobber keeps the compiler's order)
    if (wave_on) {
      T* obase = (T*)p.o + (int64_t)(q0 + wpos0) * p.o_stride_t + (int64_t)(kh * p.group + hl) * D;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = 4 * j + gq;
        const u32x4_t v = *(const u32x4_t*)(orw + row * ROWB + (a << 4));
        if (wpos0 + row < ext_len) *(u32x4_t*)(obase + (int64_t)row * p.o_stride_t + ((a ^ (row & 15)) << 3)) = v;
      }
    }
  }
}

}  // namespace

bool extend_phased_eligible(const ExtendParams& p) {
  return p.casc_bs == 0 && !p.kv_fp8 && p.custom_mask == nullptr && p.sliding_window <= 0 && !(p.logit_cap > 0.0f) && p.group >= 1 &&
         p.group <= 8 && p.o_stride_t % 8 == 0 && ((uintptr_t)p.o % 16) == 0 && p.ke_stride_t * 2 * 64 < (1ll << 31) &&
         p.ve_stride_t * 2 * 64 < (1ll << 31);
}

int launch_extend_phased(ExtendParams& p, int max_len_extend, int dtype, hipStream_t st) {
  constexpr int smem = 2 * 2 * kKT * 128 * 2;   // 64 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)extend_attn_phased_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)extend_attn_phased_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  int slots = 1;
  while (slots < p.group) slots <<= 1;   // head slots per workgroup (<= 8)
  const int bq = 256 / slots;
  int lg = 0;
  while ((1 << lg) < bq) ++lg;
  p.bq_log2 = lg;
  p.hchunks = 1;
  p.nqb = (max_len_extend + bq - 1) / bq;
  const int64_t nb = (int64_t)((p.bs * p.hkv + 7) / 8) * p.nqb * 8;
  if (nb <= 0) return SGL_MI355_OK;
  if (nb >= (1ll << 31)) {
    snprintf(g_sgl_mi355_err, sizeof(g_sgl_mi355_err), "extend_attention: grid too large");
    return SGL_MI355_EINVAL;
  }
  if (dtype == SGL_BF16) {
    hipLaunchKernelGGL((extend_attn_phased_kernel<__bf16>), dim3((unsigned)nb), dim3(512), smem, st, p);
  } else {
    hipLaunchKernelGGL((extend_attn_phased_kernel<_Float16>), dim3((unsigned)nb), dim3(512), smem, st, p);
  }
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}
