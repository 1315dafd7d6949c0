// Extend (prefill / chunked-prefill / RadixAttention-hit) attention, gfx950: the 8-wave 32x32x16 kernel (round 5).
//
// Same contract as extend_attn_kernel / extend_attn_dma_kernel (extend_attention.hip): replaces extend_attention_fwd
// (python/sglang/srt/layers/attention/triton_ops/extend_attention.py:41-438) for 16-bit K / V, D = 128, head group <= 8, no custom
// mask / window / logit cap / cascade / fp8 pool; semantics of torch_native_backend.py:27-110.
//
// Why another structure.  The 16x16x32 kernels run two waves per SIMD that execute the SAME phase at the same time (round-4 counters:
// vector and matrix instructions co-execute in 16 % of the MFMA-busy cycles, MFMA busy 0.32): both in their MFMA block, then both in
// their softmax chain -- a wave alone issues one VALU instruction per 4 cycles (8 for v_exp), so a tile's ~170 softmax instructions are
// ~1 000 cycles in which the matrix pipe idles -- and their 128-row workgroups ingest 64 KiB of K / V per CU and tile step.
// This kernel:
//   * 256 (head, position) rows per workgroup = 8 waves x 32 rows, ONE workgroup per CU: 32 KiB of K / V per CU and tile step;
//   * v_mfma_f32_32x32x16 with the QUERY on the lane: S^T = K Q^T leaves 32 scores of one query row in a lane's registers, the
//     reference maximum, the row sum and the rescale factor are per-lane scalars, and registers 8 s .. 8 s + 7 of the S^T accumulator,
//     converted pairwise, ARE the B operand of k-step s of O^T += V^T P^T in the permuted key order 16 s + 8 (j >> 2) + 4 h + (j & 3)
//     (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"); V^T fragments take that order by ds_read_b64_tr_b16;
//   * DEFERRED maximum: exponentials are taken against a reference maximum that moves only when a score exceeds it by 2^THR (bf16
//     has f32's exponent range, f16 reaches 2^15; power-of-two scaling is exact, and o = acc / l does not depend on the reference
//     point).  The per-tile work is then exp2(fma) / add / convert per score, with NO dependence on the tile's own maximum, so it is
//     cut into pair steps that sit in the shadow of the MFMAs: the second key block's exponentials beside the first PV MFMAs;
//   * the (rare) rescale of O^T is a PER-ROW predicated update, which the compiler performs in place; as a wave-uniform branch the
//     scaled copy lived beside the old one across the branch (154 spilled registers in the first build of this file; an intermediate
//     build kept O^T in asm-owned AGPRs, which halves the compiler's VGPR budget to 128);
//   * K / V fragments stream LDS -> registers LEAD MFMAs ahead of their use through the whole tile (K ring, then V ring): an LDS
//     round trip is always covered by LEAD - 1 MFMAs, and 28 registers hold fragments instead of 64;
//   * ONE barrier per tile, taken by waves 0-3 in front of QK^T and by waves 4-7 (their SIMD partners) between the softmax and PV: the
//     partners are a piece apart, so one's exponentials meet the other's MFMAs (three tile buffers);
//   * K / V tiles are register-staged, T14 "async-STAGE split": plain global loads right behind a barrier, ds_write_b128 into the
//     swizzled image in front of the next one (LDS-DMA from computing waves cost 550-780 cycles of ISSUE per wave and tile);
//   * O leaves through LDS as whole 256-byte rows (16-byte stores of full lines instead of 8-byte stores at a row stride).
// LDS images, source-side swizzles and fragment maps are those of the round-4 64-rows-per-wave kernel (verified against the oracle
// there): K rows of 256 B with 16-byte chunk ^ (row & 15); V rows with 32-byte chunk ^ 2 (row & 3).
// AUDIT after every edit (tools/debug/isa_audit_phased.sh): no spills, no scratch.
#include <type_traits>

#include "extend_params.h"

namespace {

typedef __attribute__((address_space(3))) void* ph_lptr_t;

#define PH_SB() __builtin_amdgcn_sched_barrier(0)

#ifdef SGL_EXT_TIMELINE
// tools/debug/ext_timeline_phased.py: s_memtime stamps of workgroup SGL_EXT_TIMELINE's eight waves, first 24 tiles, 8 stamps per tile
#define PH_STAMP(kk_)                                                                          \
  do {                                                                                         \
    if (tl && k < 24) {                                                                        \
      unsigned long long ts_;                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_)::"memory");             \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      tl[((w * 24 + k) * 8) + (kk_)] = (long long)ts_;                                           \
    }                                                                                          \
  } while (0)
#else
#define PH_STAMP(k) do { } while (0)
#endif

// log2 units: the reference maximum moves when a score exceeds it by more than this (P <= 2^THR: inside f16's 65504 / bf16's range)
template <typename T> struct PhDefer { static constexpr float thr = 30.0f; };
template <> struct PhDefer<_Float16> { static constexpr float thr = 12.0f; };

template <typename T>
__global__ __launch_bounds__(512, 2) void extend_attn_phased_kernel(const ExtendParams p) {
  using Tr = ElemTraits<T>;
  using vec8 = typename Tr::vec8;
  constexpr int ROWB = 256, KS = 8, D = 128;
  constexpr int TILE_B = kKT * ROWB;  // 16 KiB
  constexpr float THR = PhDefer<T>::thr;
  // [3 buffers][K 16 KiB | V 16 KiB]; the O rows at the end (1 KiB aligned: the fragment addresses XOR into bits 5-7)
  extern __shared__ __attribute__((aligned(1024))) char smem[];

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

  // ---- staging, T14 "async-STAGE split" (cdna_hip_programming.md): wave w fills rows 8 w .. 8 w + 7 of a tile; lane l fetches the 16-byte
  // chunk l & 15 of row 8 w + 4 i + (l >> 4) (i = 0, 1: two loads per operand, whole 256-byte rows per 16 lanes) right after a barrier,
  // and writes it to the swizzled position of the LDS image just before the next barrier.  (The first two builds of this file used
  // LDS-DMA, as extend_attn_dma_kernel does: the in-kernel timeline showed the four global_load_lds of a wave taking 550-780 cycles to
  // ISSUE per tile from a wave that also computes -- a fifth of the tile; plain loads issue in a few cycles and the four ds_write_b128
  // cost ~60.)
  const char* kpool = (const char*)p.k_buf + ((int64_t)kh * p.k_stride_h) * 2;
  const char* vpool = (const char*)p.v_buf + ((int64_t)kh * p.v_stride_h) * 2;
  const char* kext = (const char*)p.ke + ((int64_t)q0 * p.ke_stride_t + (int64_t)kh * D) * 2;
  const char* vext = (const char*)p.ve + ((int64_t)q0 * p.ve_stride_t + (int64_t)kh * D) * 2;
  const int64_t kpst = p.k_stride_t * 2, vpst = p.v_stride_t * 2;
  const unsigned kest = (unsigned)(p.ke_stride_t * 2), vest = (unsigned)(p.ve_stride_t * 2);
  // lane constants of the staging: source offsets of a whole tile of new tokens (instruction i adds 4 rows: a scalar), LDS offsets
  const unsigned ksrc0 = (unsigned)(8 * w + gq) * kest + 16u * a, vsrc0 = (unsigned)(8 * w + gq) * vest + 16u * a;
  // LDS image positions: K row r keeps chunk j at position j ^ (r & 15); V row r keeps 32-byte chunk J at J ^ 2 (r & 3)
  const unsigned kdst0 = (unsigned)((8 * w + gq) * ROWB + ((a ^ ((8 * w + gq) & 15)) << 4));
  const unsigned vdst0 = (unsigned)((8 * w + gq) * ROWB + (((((a >> 1) ^ ((gq & 3) << 1)) << 1) | (a & 1)) << 4));
  u32x4_t kst[2], vst[2];   // the staged tile: in flight through a whole interval
  int idc[2] = {0, 0};      // pool slots of the rows the NEXT staging call fetches (prefix tiles)
  auto load_idx = [&](int tn) {   // slots of tile tn's rows
    if (tn < npre_tiles) {
#pragma unroll
      for (int i = 0; i < 2; ++i) idc[i] = idx_row[min(tn * kKT + 8 * w + 4 * i + gq, pre_len - 1)];
    }
  };
  auto stage_load = [&](int tn) {   // tile tn -> registers (tn < ntiles)
    const int te = tn - npre_tiles;
    if (te >= 0) {
      const char* kb_ = kext + (int64_t)te * kKT * kest;
      const char* vb_ = vext + (int64_t)te * kKT * vest;
      if (te * kKT + kKT <= ext_len) {   // a whole tile of new tokens
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          kst[i] = *(const u32x4_t*)(kb_ + 4 * i * kest + ksrc0);
          vst[i] = *(const u32x4_t*)(vb_ + 4 * i * vest + vsrc0);
        }
      } else {   // ragged last tile: rows past the end re-read the last row (finite data; their probabilities are exactly 0)
        const int last = ext_len - 1 - te * kKT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned rr = (unsigned)min(8 * w + 4 * i + gq, last);
          kst[i] = *(const u32x4_t*)(kb_ + rr * kest + 16u * a);
          vst[i] = *(const u32x4_t*)(vb_ + rr * vest + 16u * a);
        }
      }
    } else {   // pool rows through their slots
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        kst[i] = *(const u32x4_t*)(kpool + (int64_t)idc[i] * kpst + 16u * a);
        vst[i] = *(const u32x4_t*)(vpool + (int64_t)idc[i] * vpst + 16u * a);
      }
    }
  };
  auto stage_write = [&](int buf) {   // registers -> buffer buf (instruction i: rows + 4: the K chunk position flips bit 2)
#ifdef SGL_EXT_NOSTAGE   // timing-only build (wrong results): no staging stores
    if (buf >= 0) return;
#endif
    char* kd = smem + buf * 2 * TILE_B;
    char* vd = kd + TILE_B;
    *(u32x4_t*)(kd + kdst0) = kst[0];
    *(u32x4_t*)(kd + 4 * ROWB + (kdst0 ^ 64u)) = kst[1];
    *(u32x4_t*)(vd + vdst0) = vst[0];
    *(u32x4_t*)(vd + 4 * ROWB + vdst0) = vst[1];
  };

  // fragment read addresses: ONE lane-constant register per operand; the buffer's base is added per tile into an opaque copy so that
  // the eight / four XOR variants are rebuilt per tile (one VALU each) instead of living in a dozen registers:
  //   K fragment (kk, ks): row 32 kk + c, chunk (2 ks + h) ^ (c & 15)           = (klane ^ 32 ks) + 8192 kk
  //   V^T read (n, sI, half): row 16 sI + 8 half + 4 h + (a >> 2), 32-byte chunk (2 n + (gq & 1)) ^ 2 ((a >> 2) & 3), piece a & 3
  //                                                                               = (vlane ^ 64 n) + 4096 sI + 2048 half
  const unsigned lds_base = (unsigned)(uintptr_t)(ph_lptr_t)smem;
  const unsigned klane = (unsigned)(c * ROWB + ((c & 14) << 4) + ((h ^ (c & 1)) << 4));
  const unsigned vlane = (unsigned)((4 * h + (a >> 2)) * ROWB + (((a >> 2) & 3) << 6) + ((gq & 1) << 5) + ((a & 3) << 3));
  typedef const vec8 __attribute__((address_space(3)))* kptr_t;
  typedef s16x4_t __attribute__((address_space(3)))* vptr_t;

  // The reference maximum of a row enters the scores as the C input of each key block's first MFMA (minit = 16 x -m_ref in units of
  // q . k; it changes on the rare path only), so a probability is TWO instructions away from the accumulator: p = exp2(cs * S^T register).
  // (Folding cs = sm_scale log2 e into Q as well -- one instruction -- rounds Q a second time to 8 bits: an error of ~1e-3 |score|
  // in the exponent, 0.86 absolute in the outputs of test_extend_deferred_rescale_branch_is_taken_and_right.)
  const float cs = p.sm_scale * kLog2e;
  float l_i = 0.f;   // this lane half's share of the row sum
  f32x16_t minit;
#pragma unroll
  for (int r = 0; r < 16; ++r) minit[r] = 0.f;
  f32x16_t acc[4];   // O^T: acc[n] = d block n of the wave's 32 query rows (32x32 accumulator tile)
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const bool grpB = w >= 4;   // the SIMD partners of waves 0-3 (a workgroup's waves w and w + 4 share a SIMD)

  // ---- prologue: tile 0 in LDS, tile 1 in the staging registers, the slots of tile 2 known ----
  load_idx(0);
  stage_load(0);
  stage_write(0);
  if (ntiles > 1) {
    load_idx(1);
    stage_load(1);
  }
  load_idx(min(2, ntiles - 1));
  __syncthreads();

  // ---- the loop.  Per wave and 64-key tile: M1 16 MFMAs S^T = K Q^T | N2 decision + exponentials of key block 0 | M2 16 MFMAs
  // O^T += V^T P^T with the exponentials of key block 1 in their shadow.  ONE workgroup barrier per tile, taken by waves 0-3 in front of
  // M1 and by waves 4-7 between N2 and M2:
  //      waves 0-3:  | M1(k) N2(k) M2(k)   |            waves 4-7:  | M2(k) M1(k+1) N2(k+1) |
  // so the vector piece N2 of one wave of a SIMD meets a matrix piece of its partner.  With the barrier at the SAME point of every wave's
  // tile (the first version of this loop) the partners ran their matrix pieces together (16 MFMAs took 700-1 000 cycles instead of
  // 512) and their vector pieces together (matrix pipe idle): 5 000 cycles per tile step against 2 048 of MFMA -- the figure of the
  // 16x16x32 kernels.  Barrier #k makes tile k + 1 visible (every wave writes its staged share in front of it) and frees the buffer of
  // tile k - 1 (waves 0-3 are through tile k - 1, waves 4-7 through M2(k - 1)); behind it every wave fetches its share of tile k + 2.
  // Tile t lives in buffer t % 3.
  constexpr int LEAD = 6;   // fragments requested ahead of the MFMA that consumes them
  // fragment f of S^T = K Q^T: k-step f >> 1, key block f & 1 (the two MFMAs of a k-step share their Q^T fragment); fragment g of
  // O^T += V^T P^T: k-step g >> 2, d block g & 3
#ifdef SGL_EXT_NOREAD   // timing-only build (wrong results): no fragment reads
  auto kread = [&](unsigned kb, int f) -> vec8 { return qf[f & 7]; };
  auto vread = [&](unsigned vb, int g) -> vec8 { return qf[g & 7]; };
  auto vread_unused = [&](unsigned vb, int g) -> vec8 {
#else
  auto kread = [&](unsigned kb, int f) -> vec8 { return *(kptr_t)(uintptr_t)((kb ^ (unsigned)(32 * (f >> 1))) + 8192u * (f & 1)); };
  auto vread = [&](unsigned vb, int g) -> vec8 {   // keys 16 sI + 4 h .. + 3 and 16 sI + 8 + 4 h .. + 3 of d block n
#endif
    const unsigned ad = (vb ^ (unsigned)(64 * (g & 3))) + 4096u * (g >> 2);
    const s16x4_t t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((vptr_t)(uintptr_t)ad);
    const s16x4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((vptr_t)(uintptr_t)(ad + 2048u));
    return __builtin_bit_cast(vec8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  auto sync_point = [&](int k, int bufn) __attribute__((always_inline)) {   // barrier #k of this wave
    if (k + 1 < ntiles) stage_write(bufn);   // tile k + 1 (fetched behind barrier #(k - 1)) -> buffer (k + 1) % 3
    PH_SB();
    __syncthreads();                         // (waits for the LDS stores, then s_barrier)
    PH_SB();
  };
  auto fetch_next = [&](int k) __attribute__((always_inline)) {   // tile k + 2 -> registers, the slots of tile k + 3
    if (k + 2 < ntiles) stage_load(k + 2);
    load_idx(min(k + 3, ntiles - 1));
    PH_SB();
  };

#ifndef SGL_EXT_NOPRIO
  // waves 4-7 are the younger half of the workgroup and lose the arbitration for the SIMD's vector port to their partners on every
  // piece (MI355X_MICROARCH.md, "Two waves per SIMD", items 2 and 4): in the timeline their pieces took 1.3-1.8 x as long and waves
  // 0-3 then idled at the barrier while their partners finished alone (a lone wave uses the port and the matrix pipe worse than two).
  if (grpB) __builtin_amdgcn_s_setprio(1);
#endif
  int bufK = 0;   // k % 3
#ifdef SGL_EXT_NOLOOP   // timing-only build (wrong results): prologue and epilogue only
  for (int k = 0; k < 0; ++k) {
#else
  for (int k = 0; k < ntiles; ++k) {
#endif
    PH_STAMP(0);
    const int bufN = bufK == 2 ? 0 : bufK + 1;   // (k + 1) % 3
    if (!grpB) {
      sync_point(k, bufN);
      fetch_next(k);
    }
    PH_STAMP(1);
    const bool in_prefix = k < npre_tiles;
    const int kbase = in_prefix ? k * kKT : (k - npre_tiles) * kKT;
    const int klimit = in_prefix ? pre_len : ext_len;
    const bool causal = !in_prefix && p.is_causal;
#ifdef SGL_EXT_NOCOMPUTE   // timing-only build (wrong results): staging and barriers only
    const bool active = false;
#else
    const bool active = wave_on && !(causal && kbase > wpos0 + 31);   // wave-uniform: tiles in the causal future of every row are skipped
#endif
    // key of register r of block kk = kbase + 32 kk + (r & 3) + 8 (r >> 2) + 4 h; seen iff 32 kk + (r & 3) + 8 (r >> 2) <= lim
    const int lim = min(klimit - 1, causal ? qpos : 0x7fffffff) - kbase - 4 * h;
    unsigned kb = lds_base + bufK * 2 * TILE_B + klane, vb = lds_base + bufK * 2 * TILE_B + TILE_B + vlane;
    asm volatile("" : "+v"(kb), "+v"(vb));   // opaque per tile: the XOR variants are rebuilt, not kept in registers

    // ONE instance of the matrix pieces (every further copy -- a masked / unmasked pair, one per wave group -- is another set of
    // 64 + 32 accumulator registers joined at the end of a branch: 17 - 64 spilled registers per copy in the builds that had them)
    vec8 kf[16], vf[16], pq[4];
    f32x16_t sq[2];
    float lsum = 0.f;
    // P^T pair step j of block kk: registers 2 j, 2 j + 1 -> two elements of pq[2 kk + (j >> 2)] (the B operand of PV k-step 2 kk + (j >> 2)):
    // per score one v_mul_f32, one v_exp_f32, one v_add_f32 and half a v_cvt_pk.  The SIMD's vector port is what bounds this kernel (tools/microbench/
    // coissue.hip: ~4 cycles per VALU instruction, 8.5 per v_exp, for BOTH waves of the SIMD together, 8 of every MFMA's 32 cycles taken
    // by its issue): 32 v_exp + ~120 other vector instructions per wave and tile is what fits beside 32 MFMAs.
    // Two stages, software-pipelined by one step: stage A (scale, exponential) of step j + 1 is issued in front of stage B (row sum,
    // convert) of step j, so that no instruction waits for the transcendental unit's result of the instruction before it (in order,
    // alone on its SIMD, a wave lost ~30 cycles per pair step to those waits).
    float pe0[2], pe1[2];   // the two probabilities of the step in flight (index: step parity)
    float lsum2 = 0.f;      // second partial sum: two independent add chains
    auto pstepA = [&](int kk, int j) {
      const int r0 = 2 * j, r1 = 2 * j + 1;
      const float x0 = sq[kk][r0] * cs, x1 = sq[kk][r1] * cs;
#ifdef SGL_EXT_NOEXP   // timing-only build (wrong results)
      pe0[j & 1] = x0; pe1[j & 1] = x1;
#else
      pe0[j & 1] = __builtin_amdgcn_exp2f(x0);
      pe1[j & 1] = __builtin_amdgcn_exp2f(x1);
#endif
    };
    auto pstepB = [&](int kk, int j) {
      const int r0 = 2 * j, r1 = 2 * j + 1;
      const float p0 = pe0[j & 1], p1 = pe1[j & 1];
      lsum += p0;
      lsum2 += p1;
      pq[2 * kk + (r0 >> 3)][r0 & 7] = Tr::from_f32(p0);
      pq[2 * kk + (r0 >> 3)][r1 & 7] = Tr::from_f32(p1);
    };
    if (active) {
      // ---- M1: S^T = K Q^T; the first V^T fragments are requested behind the last MFMAs ----
#pragma unroll
      for (int f = 0; f < LEAD; ++f) kf[f] = kread(kb, f);
      PH_SB();
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        const int kk = f & 1, ks = f >> 1;
        if (ks == 0) {
          sq[kk] = Tr::mfma32(kf[f], qf[ks], minit);
        } else {
          sq[kk] = Tr::mfma32(kf[f], qf[ks], sq[kk]);
        }
        if (f + LEAD < 16) kf[f + LEAD] = kread(kb, f + LEAD);
        else vf[f + LEAD - 16] = vread(vb, f + LEAD - 16);
        PH_SB();
      }
      PH_STAMP(2);

      // ---- N2: decision -- does a row's maximum exceed its reference maximum by more than 2^THR?  (rare: a jump; the first tile) ----
      // Keys a row must not see (ragged end of a phase, causal diagonal): a PER-LANE condition, so that the update is predicated code
      // that rewrites the scores in place (skipped by every wave of an unmasked tile); as a wave-uniform branch it is a second copy of
      // the 32 score registers joined behind the branch.
      if (lim < 63) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int r = 0; r < 16; ++r) sq[kk][r] = (32 * kk + (r & 3) + 8 * (r >> 2) <= lim) ? sq[kk][r] : -INFINITY;
      }
      float mc[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // four independent chains (one chain of 16 v_max3 waits on itself)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int r = 0; r < 16; ++r) mc[(r >> 1) & 3] = fmaxf(mc[(r >> 1) & 3], sq[kk][r]);
      const float mloc = fmaxf(fmaxf(mc[0], mc[1]), fmaxf(mc[2], mc[3]));
      // A PER-ROW (divergent) condition, the same in both lane halves of a row: as predicated code the scaled O^T takes the registers
      // of the old one (lanes that skip the update keep their values in place).  As a wave-uniform branch the scaled copy lived beside
      // the old one across the branch: +64 registers on the loop (154 spilled registers in the first build of this file).
      const float mrow = pair32_max(mloc);   // the row's maximum in this tile RELATIVE to its reference maximum, units of q . k
      if (k == 0 || mrow * cs > THR) {       // the first tile fixes the reference; later it moves when a score exceeds it by 2^THR
        const float delta = fmaxf(mrow, -1e30f);   // (a row without a visible key: keeps finite arithmetic; it is not stored)
        if (k != 0) {
          const float alpha = __builtin_amdgcn_exp2f(-delta * cs);
          l_i *= alpha;
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          minit[r] -= delta;
          sq[0][r] -= delta;
          sq[1][r] -= delta;
        }
      }
      pstepA(0, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j + 1 < 8) pstepA(0, j + 1);
        else pstepA(1, 0);            // the first step of key block 1: its stage B sits behind the first PV MFMA
        PH_SB();
        pstepB(0, j);
        PH_SB();
      }
      PH_STAMP(3);
    }
    if (grpB) {
      sync_point(k, bufN);
      fetch_next(k);   // (issued here, the loads have M2, M1 and N2 to land; issued behind M2 -- to keep the two groups' fetches apart --
    }                  //  waves 4-7 stalled ~700 cycles per tile at their staging stores: a global load takes 2 000+ cycles under this load)
    PH_STAMP(4);
    if (active) {
      // ---- M2: O^T += V^T P^T; the second key block's exponentials sit behind the MFMAs of k-steps 0, 1 ----
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        acc[g & 3] = Tr::mfma32(vf[g], pq[g >> 2], acc[g & 3]);
        if (g + LEAD < 16) vf[g + LEAD] = vread(vb, g + LEAD);
        if (g < 8) {
          if (g + 1 < 8) pstepA(1, g + 1);
          PH_SB();
          pstepB(1, g);
        }
        PH_SB();
      }
      l_i += lsum + lsum2;
    }
    PH_STAMP(5);
    bufK = bufN;
    PH_STAMP(6);
  }
  __syncthreads();   // every wave is through its last tile: the tile buffers become the O rows

  // ---- o = acc / l through LDS: wave w owns bytes [8 KiB w, 8 KiB (w + 1)) = its 32 rows x 256 B, 16-byte chunk ^ (row & 15) ----
  {
    const float l = pair32_sum(l_i);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    char* orw = smem + w * (32 * ROWB);
#pragma unroll
    for (int n = 0; n < 4; ++n)
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
         p.group <= 8 && p.o_stride_t % 8 == 0 && ((uintptr_t)p.o % 16) == 0 && p.ke_stride_t % 128 == 0 && p.ve_stride_t % 128 == 0 && p.ke_stride_t * 2 * 64 < (1ll << 31) &&
         p.ve_stride_t * 2 * 64 < (1ll << 31);
}

int launch_extend_phased(ExtendParams& p, int max_len_extend, int dtype, hipStream_t st) {
  constexpr int smem = 3 * 2 * kKT * 128 * 2;   // 96 KiB: three tile buffers
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
