// Persistent MLP half of a w8a8-fp8 decode layer (M <= 32 tokens): ONE launch for
//
//     post_attention_layernorm (fused add + RMSNorm)            models/llama.py:283-285, layernorm.py:135-171
//  -> per-token fp8 quantisation of the normed row               fp8_utils.py:706-713, per_token_quant_fp8.cu:15-228
//  -> gate_up_proj (fp8_scaled_mm) + SiluAndMul                  models/llama.py:94-98, activation.py:60-63
//  -> per-token fp8 quantisation of the MLP activation           w8a8_fp8.py:177-190
//  -> down_proj (fp8_scaled_mm), raw split-K partial sums        linear.py:1285-1309
//
// instead of four launches (add_rmsnorm_quant, gemm_silu_mul, per_token_quant, skinny_gemm_slabs).  Why one launch: at
// batch 32 each of those launches pays ~4-5 us of skeleton (dependent-launch gap, the first bytes of a cold weight stream,
// the drain of the last tile) around 6 TB/s of streaming; the three row-shaped steps between the two GEMMs move almost no
// bytes.  Here one 512-thread workgroup per CU stays resident for the whole block and its weight loads RUN AHEAD of every
// dependency: the first gate_up tiles are requested before the normed rows exist, the first down_proj tiles before the
// activation is quantised, so HBM keeps streaming while the chip-wide hand-offs (MI355X_MICROARCH.md, price list:
// allgather / fanin rows) complete.
//
// Structure (G = one workgroup per CU, 8 waves; the GEMM core is skinny_gemm.hip's X-stationary weight stream):
//   * every wave owns a 512-byte K slice; its X fragments are built once per GEMM; weight tiles (16 rows) travel
//     HBM -> registers (two tiles in flight per wave, nt policy) -> wave-private swizzled LDS image -> MFMA 16x16x32 fp8;
//     the 8 K-slice partials of two tiles meet in LDS between two barriers;
//   * wave 7 is the UTILITY wave: it alone touches the hand-off state, so the other seven keep their run-ahead loads in
//     flight across a hand-off (vmcnt is in-order: a wave that polls or drains must not have a deep prefetch outstanding);
//     its own first tiles after a hand-off are requested late, everything else is identical;
//   * hand-off 1 (normed rows): wave 7 of workgroup r < M computes row r (add, RMSNorm, quant), publishes 4 KiB of fp8 +
//     one scale with write-through (sc1) stores, drains, adds to a counter; every workgroup's wave 7 polls the counter,
//     the workgroup barrier releases the sc1 loads of the X staging;
//   * hand-off 2 (row maxima of the activation): per workgroup 32 partial maxima -> pmax[b][32] (one 128-byte line),
//     8 sharded arrival counters; every workgroup reduces the G lines itself (32 KiB of L2 reads);
//   * hand-off 3 (quantised activation): wave 7 quantises the workgroup's own columns (still in LDS) with the global row
//     scale and publishes them (8-byte sc1 stores), sharded arrival counters again; then the down_proj X staging.
// Every spin is bounded (a timeout sets the error word and the block runs on to completion with wrong data instead of
// hanging the GPU); the caller zeroes the sync block before every launch and checks the error word when it syncs anyway.
//
// Numerics: the arithmetic of every step is the stand-alone kernels' (same rounding points); the RMSNorm sum and the row
// maxima are reduced in a different order (one wave per row; max is order-independent), so results agree with the
// four-launch path to f32 rounding of the variance, not bit for bit.
#include <type_traits>

#include "gemm_epilogue.h"
#include "row_helpers.h"

namespace {

constexpr int kMlpWaves = 8;
constexpr int kMlpDS = 8;                  // 16-byte pieces of K per lane and tile row group
constexpr int kMlpKW = kMlpDS * 64;        // bytes of K per wave (512)
constexpr int kMlpLPR = kMlpKW / 16;       // lanes per weight row in one load instruction (32)
constexpr int kMlpRPI = 64 / kMlpLPR;      // rows per load instruction (2)
constexpr int kMlpIMG = 16 * kMlpKW;       // one wave's LDS image
constexpr int kMlpRange = kMlpWaves * kMlpKW;  // K bytes one workgroup covers (4096)
constexpr int kMlpMaxT1 = 16;              // gate_up tiles per workgroup (act columns kept in LDS: 8 per tile)
constexpr int kMlpActLd = kMlpMaxT1 * 8 + 1;
constexpr int kMlpSpin = 400000;           // bounded spins: ~0.3 s
// sync block (uint32 words; the caller zeroes all of it before every launch)
constexpr int kSyncA = 0, kSyncErr = 16, kSyncB = 32, kSyncC = 32 * 9, kSyncWords = 32 * 18;

struct MlpParams {
  const void* x;         // [M, H] T: o_proj output
  void* residual;        // [M, H] T, in/out
  const void* ln_w;      // [H] T
  float eps;
  const char* wbase;     // min(gate_up, down_proj) address: both weights are read through one buffer descriptor
  unsigned wbytes, w1off, w2off;  // descriptor size; byte offsets of gate_up (row-interleaved [2I, H] e4m3) and down_proj ([H, I] e4m3)
  const float* sw1;      // [2I] gate_up scales, interleaved the same way
  float* slabs;          // out: [kranges, M, H] raw f32 partial sums
  float* act_scales;     // out: [M] per-token scale of the activation (the consumer's sx)
  uint8_t* xq;           // scratch [M, H]
  float* xs;             // scratch [M]
  uint8_t* actq;         // scratch [M, I]
  uint32_t* pmax;        // scratch [G, 32]
  uint32_t* sync;
  long long* tl;         // optional timeline [G][8][16] (s_memrealtime ticks, 10 ns)
  int M, H, I;
  int ntiles1, ntiles2, kranges, gpr;
};

typedef __attribute__((address_space(1))) uint32_t gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1_64(void* p, unsigned long long v) {
  __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void arrive(uint32_t* c) { __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one wave: wait until *c >= target (relaxed sc1 poll + sleep)
__device__ __forceinline__ void wait_count(const uint32_t* c, uint32_t target, uint32_t* err, uint32_t code, int lane) {
  for (int spin = 0; spin < kMlpSpin; ++spin) {
    if (ld_sc1(c) >= target) return;
    __builtin_amdgcn_s_sleep(2);
  }
  if (lane == 0) st_sc1(err, code);
}
// one wave: lanes 0..7 each watch one shard (shard s counts the workgroups with blockIdx % 8 == s)
__device__ __forceinline__ void wait_shards(const uint32_t* base, int G, uint32_t* err, uint32_t code, int lane) {
  const int s = lane & 7;
  const uint32_t target = (uint32_t)((G - s + 7) >> 3);
  for (int spin = 0; spin < kMlpSpin; ++spin) {
    const bool ok = ld_sc1(base + 32 * s) >= target;
    if (__all(ok)) return;
    __builtin_amdgcn_s_sleep(2);
  }
  if (lane == 0) st_sc1(err, code);
}

__device__ __forceinline__ void mfma_fp8(const u32x4_t& xa, const u32x4_t& wb, f32x4_t& acc) {
  const long a0 = ((long)xa[1] << 32) | (long)xa[0], a1 = ((long)xa[3] << 32) | (long)xa[2];
  const long b0 = ((long)wb[1] << 32) | (long)wb[0], b1 = ((long)wb[3] << 32) | (long)wb[2];
  acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, acc, 0, 0, 0);
}

template <typename T, int MT, bool TL>
__global__ __launch_bounds__(kMlpWaves * 64, 1) void fp8_mlp_block_kernel(const MlpParams p) {
  constexpr int NWV = kMlpWaves, DS = kMlpDS, KW = kMlpKW, LPR = kMlpLPR, RPI = kMlpRPI, IMG = kMlpIMG;
  constexpr int MR = MT * 16;
  constexpr int NS = 2;  // weight tiles in flight per wave (register sets; tile u of the stream sits in set u % 2)
  constexpr unsigned OOB = 0xFFFFFFF0u;
  __shared__ __attribute__((aligned(16))) char wimg[NWV * IMG];  // 64 KiB
  __shared__ float red[2][NWV][MR][16];                          // 32 KiB at MT = 2
  __shared__ float act_tile[MR][kMlpActLd];                      // this workgroup's act columns, rounded to T
  __shared__ float redw[2][NWV];
  __shared__ float sw_l[kMlpMaxT1 * 16];                         // gate_up weight scales of this workgroup's tiles
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = lane & 15, g = lane >> 4;
  const int lc = lane % LPR, lr = lane / LPR;
  const int b = blockIdx.x, G = gridDim.x;
  char* wl = wimg + w * IMG;
  const bool util = w == NWV - 1;
  uint32_t* const err = p.sync + kSyncErr;
#define MLP_STAMP(i)                                                                                        \
  do {                                                                                                      \
    if constexpr (TL) {                                                                                     \
      const long long t_ = (long long)__builtin_amdgcn_s_memrealtime();                                     \
      if (lane == 0) p.tl[((int64_t)b * NWV + w) * 16 + (i)] = t_;                                          \
    }                                                                                                       \
  } while (0)
  MLP_STAMP(0);

  // ---- geometry: ONE stream of weight tiles per workgroup, gate_up tiles 0 .. cnt1p - 1 then down_proj tiles ----
  const int cnt1 = (p.ntiles1 - b + G - 1) / G;  // gate_up tiles b, b + G, ...  (host: ntiles1 >= G, so >= 1)
  const int cnt1p = (cnt1 + 1) & ~1;
  const int kr = b / p.gpr, bx = b - kr * p.gpr;  // down_proj: k-range and slot inside it
  const int cnt2 = kr < p.kranges ? max(0, (p.ntiles2 - bx + p.gpr - 1) / p.gpr) : 0;
  const int cnt2p = (cnt2 + 1) & ~1;
  const int utot = cnt1p + cnt2p;
  const int koff1 = w * KW + lc * 16;
  const bool kok1 = koff1 < p.H;
  const int koff2 = kr * kMlpRange + w * KW + lc * 16;
  const bool kok2 = kr < p.kranges && koff2 < p.I;
  // both weights through ONE descriptor (the host checks that they lie within one 4 GiB window): the run-ahead from the
  // gate_up stream into the down_proj stream is then the same load instructions with other offsets, no branch
  const auto rsrcw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wbase, 0, p.wbytes, 0x00020000);
  const auto rsrcx1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.xq, 0, (unsigned)(p.M * p.H), 0x00020000);
  const auto rsrcx2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.actq, 0, (unsigned)((int64_t)p.M * p.I), 0x00020000);

  u32x4_t wreg[NS][DS];
  // request tile u of the stream into register set `slot`.  hold2: the utility wave holds back down_proj tiles while it is
  // still in the gate_up phase (its vmcnt must stay clear for the polls and drains of the hand-offs).  Out-of-range offsets
  // are answered with zeros without touching memory: no branch around a load.  Cache policy 2 = nt: every weight byte is
  // read by one CU, once per step (MI355X_MICROARCH.md "nt-weights").
  // Everything that differs between the two streams is held as a pair of plain values and mixed with a bit mask: a `ph2 ? a : b`
  // between two captured variables became a select of their ADDRESSES (scratch round trip + vmcnt(0) inside the loop), and a
  // `live ? off : OOB` became a branch around each load, again with `s_waitcnt vmcnt(0)` in front of it.
  const unsigned lm1 = kok1 ? 0xFFFFFFFFu : 0u, lm2 = kok2 ? 0xFFFFFFFFu : 0u;  // lanes past the end of a K row request nothing
  const unsigned st1 = (unsigned)p.H, st2 = (unsigned)p.I;
  const unsigned fb1 = p.w1off + (unsigned)koff1 + (unsigned)(b * 16 + lr) * st1;        // gate_up tile 0, this lane's first row
  const unsigned fb2 = p.w2off + (unsigned)koff2 + (unsigned)(bx * 16 + lr) * st2;       // down_proj tile 0
  const unsigned ts1 = (unsigned)(G * 16) * st1, ts2 = (unsigned)(p.gpr * 16) * st2;    // tile j -> j + 1
  auto issue = [&](auto slot_c, int u, bool hold2) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
    const bool ph2 = u >= cnt1p;  // (wave-uniform, like everything derived from it)
    const unsigned pm = ph2 ? 0xFFFFFFFFu : 0u;
    const int j = u - (cnt1p & (int)pm);
    const int cnt = (cnt1 & ~(int)pm) | (cnt2 & (int)pm);
    const bool tlive = j < cnt && !(ph2 && hold2);
    const unsigned m = ((lm1 & ~pm) | (lm2 & pm)) & (tlive ? 0xFFFFFFFFu : 0u);
    const unsigned stride = (st1 & ~pm) | (st2 & pm);
    const unsigned first = ((fb1 & ~pm) | (fb2 & pm)) + (unsigned)j * ((ts1 & ~pm) | (ts2 & pm));
#pragma unroll
    for (int i = 0; i < DS; ++i) {
      const unsigned off = ((first + (unsigned)(RPI * i) * stride) & m) | (OOB & ~m);
      wreg[slot][i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrcw, off, 0, 2));
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // X rows of this wave's K slice (sc1 loads: the rows were published inside this launch) -> swizzled image -> A fragments.
  // Rows past M come back as zeros from the buffer bounds check.  Every workgroup starts at a different row (skinny_gemm.hip).
  u32x4_t xf[MT][DS];
  const int xrot = b;
  auto stage_x = [&](auto rsrc, int64_t xstride, int koff, bool kok) __attribute__((always_inline)) {
    u32x4_t xr[MT][DS];
#pragma unroll
    for (int q = 0; q < MT; ++q)
#pragma unroll
      for (int i = 0; i < DS; ++i) {
        const int m = q * 16 + lr + RPI * ((i + xrot) & (DS - 1));
        const unsigned off = (m < p.M && kok) ? (unsigned)(m * xstride) + koff : OOB;
        xr[q][i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16));  // aux 16 = sc1
      }
#pragma unroll
    for (int q = 0; q < MT; ++q) {
#pragma unroll
      for (int i = 0; i < DS; ++i) {
        const int row = lr + RPI * ((i + xrot) & (DS - 1));
        *(u32x4_t*)(wl + row * KW + (((lc ^ row) & (LPR - 1)) << 4)) = xr[q][i];
      }
#pragma unroll
      for (int sI = 0; sI < DS; ++sI) xf[q][sI] = *(const u32x4_t*)(wl + a * KW + ((((4 * sI + g) ^ a) & (LPR - 1)) << 4));
    }
  };

  // one staged tile: registers -> image (wave private: same-wave LDS ops are ordered) -> MFMA -> partial sums in LDS; the
  // register set is requested again (tile u + 2) as soon as it has been written out
  auto run_tile = [&](auto slot_c, int jj, int u, bool hold2) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
#pragma unroll
    for (int i = 0; i < DS; ++i) {
      const int row = lr + RPI * i;
      *(u32x4_t*)(wl + row * KW + (((lc ^ row) & (LPR - 1)) << 4)) = wreg[slot][i];
    }
    issue(slot_c, u + NS, hold2);
    f32x4_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sI = 0; sI < DS; ++sI) {
      const u32x4_t wf = *(const u32x4_t*)(wl + a * KW + ((((4 * sI + g) ^ a) & (LPR - 1)) << 4));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) mfma_fp8(xf[mt][sI], wf, acc[mt]);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[jj][w][mt * 16 + 4 * g + r][a] = acc[mt][r];
  };

  // The weight scales of ALL this workgroup's gate_up tiles are fetched now, ahead of every weight load: requested at the start
  // of each pair (as the stand-alone kernel does with one tile in flight) they queued behind the two weight tiles already in
  // flight, and each pair's epilogue waited ~5 us for them -- the stream ran at 3.8 us per tile instead of 3.0.
  if (tid < cnt1 * 16) sw_l[tid] = p.sw1[(b + (tid >> 4) * G) * 16 + (tid & 15)];

  // ---- phase A: the normed + quantised rows.  Workgroup r < M computes row r with all its waves BEFORE it starts its
  // weight stream (a dependent round trip of a streaming CU costs 2-3 us: the first version, one wave beside seven streaming
  // ones, published its row after 9 us), publishes it write-through and adds to the counter. ----
  if (b < p.M) {
    const int nvec = p.H / 8;
    const bool on = tid < nvec;
    const int i = on ? tid : 0;
    const T* xrow = (const T*)p.x + (int64_t)b * p.H;
    T* rrow = (T*)p.residual + (int64_t)b * p.H;
    const V8<T> xv = ld8(xrow + i * 8), rv = ld8(rrow + i * 8), wv = ld8((const T*)p.ln_w + i * 8);
    float f[8];
    float ss = 0.f;
    V8<T> ro;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] = (float)xv.v[j] + (float)rv.v[j];
      ro.v[j] = (T)f[j];
      if (on) ss = fmaf(f[j], f[j], ss);
    }
    if (on) st8(rrow + i * 8, ro);
    ss = wave_reduce_sum(ss);
    if (lane == 0) redw[0][w] = ss;
    __syncthreads();
    ss = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWV; ++ww) ss += redw[0][ww];
    const float rs = 1.0f / sqrtf(ss / (float)p.H + p.eps);
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] = round_via<T>((f[j] * rs) * (float)wv.v[j]);
      if (on) amax = fmaxf(amax, fabsf(f[j]));
    }
    amax = wave_reduce_max(amax);
    if (lane == 0) redw[1][w] = amax;
    __syncthreads();
    amax = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWV; ++ww) amax = fmaxf(amax, redw[1][ww]);
    const float scale = amax / kFp8Max;
    const float inv = (scale == 0.f) ? 0.f : 1.0f / scale;
    if (on) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = clamp448(f[j] * inv);
      const u32x2_t q = pack8_fp8(f);
      st_sc1_64(p.xq + (int64_t)b * p.H + i * 8, ((unsigned long long)q[1] << 32) | q[0]);
    }
    if (tid == 0) st_sc1((uint32_t*)p.xs + b, __builtin_bit_cast(uint32_t, scale));
    drain_vm();
    __syncthreads();
    if (tid == 0) arrive(p.sync + kSyncA);
    MLP_STAMP(9);
  }
  // ---- the first tiles of the stream are requested before the rows exist (they depend on nothing); the utility wave
  // first waits for the rows (hand-off 1) ----
  // A workgroup's memory pipeline is in-order: the utility wave's poll returns only after everything the other seven waves have
  // in flight (two tiles = 112 KiB = 4.5 us; measured: rows published at 2.9 us, hand-off passed at 7.4).  So the second tile is
  // requested when the first has landed: still two tiles on chip by the time the rows arrive, but at most one in front of a poll.
  if (util) wait_count(p.sync + kSyncA, (uint32_t)p.M, err, 1u, lane);
  issue(S0{}, 0, util);
  if (!util && b >= p.M) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  issue(S1{}, 1, util);
  __syncthreads();
  MLP_STAMP(1);

  // ---- gate_up: X fragments ----
  stage_x(rsrcx1, (int64_t)p.H, koff1, kok1);
  const int em = tid >> 4, en = tid & 15;
  const bool erow = em < MR;  // (MT = 1: waves 4..7 have no output element)
  float sxv = 0.f;
  if (erow && em < p.M) sxv = __builtin_bit_cast(float, ld_sc1((const uint32_t*)p.xs + em));
  MLP_STAMP(2);

  // ---- hand-offs 2 + 3 between the two GEMMs ----
  auto between = [&]() __attribute__((always_inline)) {
    MLP_STAMP(3);
    if (util) {
      const int r = lane & 31, half = lane >> 5;
      float m = 0.f;
      if (r < MR && r < p.M)
        for (int c = half; c < cnt1 * 8; c += 2) m = fmaxf(m, fabsf(act_tile[r][c]));
      m = fmaxf(m, __shfl_xor(m, 32, WAVE));
      if (lane < 32) st_sc1(p.pmax + (int64_t)b * 32 + lane, __builtin_bit_cast(uint32_t, m));  // one 128-byte line
      drain_vm();
      if (lane == 0) arrive(p.sync + kSyncB + 32 * (b & 7));
      wait_shards(p.sync + kSyncB, G, err, 2u, lane);
      MLP_STAMP(4);
      // every workgroup reduces the G lines itself: lane l reads 16 bytes (rows 4 (l % 8) ..) of workgroup 8 i + l / 8
      const auto rsrcp = __builtin_amdgcn_make_buffer_rsrc((void*)p.pmax, 0, (unsigned)(G * 128), 0x00020000);
      float mx[4] = {0.f, 0.f, 0.f, 0.f};
      for (int i0 = 0; i0 < G; i0 += 64) {
        u32x4_t pv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int wg = i0 + 8 * q + (lane >> 3);
          pv[q] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrcp, wg < G ? (unsigned)(wg * 128 + (lane & 7) * 16) : OOB, 0, 16));
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const uint32_t bits = pv[q][c];  // (a bit_cast of the vector ELEMENT lvalue read element 0 every time: hipcc 7.2)
            mx[c] = fmaxf(mx[c], __uint_as_float(bits));
          }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], 8, WAVE));
        mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], 16, WAVE));
        mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], 32, WAVE));
      }
      // lane l now holds the maxima of rows 4 (l % 8) + c; row r's lives in lane r / 4, component r % 4
      float rm = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float t = __shfl(mx[c], r >> 2, WAVE);
        if ((r & 3) == c) rm = t;
      }
      const float scale = rm / kFp8Max;  // sgl_per_token_quant_fp8's scale of row r
      const float inv = (scale == 0.f) ? 0.f : 1.0f / scale;
      if (b == 0 && lane < p.M) p.act_scales[lane] = scale;
      if (r < MR && r < p.M) {
        for (int t = half; t < cnt1; t += 2) {
          float f[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = clamp448(act_tile[r][t * 8 + j] * inv);
          const u32x2_t q = pack8_fp8(f);
          st_sc1_64(p.actq + (int64_t)r * p.I + (int64_t)(b + t * G) * 8, ((unsigned long long)q[1] << 32) | q[0]);
        }
      }
      drain_vm();
      if (lane == 0) arrive(p.sync + kSyncC + 32 * (b & 7));
      MLP_STAMP(5);
      wait_shards(p.sync + kSyncC, G, err, 3u, lane);
      issue(S0{}, cnt1p, false);  // the tiles this wave held back
      issue(S1{}, cnt1p + 1, false);
    }
    __syncthreads();
    MLP_STAMP(6);
    stage_x(rsrcx2, (int64_t)p.I, koff2, kok2);  // X fragments of this workgroup's down_proj k-range
    MLP_STAMP(7);
  };

  // ---- the tile stream: pairs of tiles between two barriers ----
  auto pair = [&](int ua) __attribute__((always_inline)) {
    if (ua == cnt1p) between();  // (workgroup-uniform)
    const bool ph2 = ua >= cnt1p;
    float swv[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) swv[jj] = sw_l[min(ua + jj, cnt1 - 1) * 16 + en];  // (phase 2: unused)
    const bool hold2 = util && !ph2;
    run_tile(S0{}, 0, ua, hold2);
    run_tile(S1{}, 1, ua + 1, hold2);
    __syncthreads();
    if (erow) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < NWV; ++ww) v += red[jj][ww][em][en];
        if (!ph2) {  // gate_up: scales, SiluAndMul (interleaved tile: gate 0..7 | up 0..7), the result stays in LDS
          const int j = ua + jj;
          v = v * sxv * swv[jj];
          const float vr = rnd_to<T>(v);             // the GEMM's own output rounding
          const float pr = __shfl_xor(vr, 8, WAVE);  // up column of the same row
          if ((en & 8) == 0 && j < cnt1 && em < p.M) {
            const float sg = rnd_to<T>(vr / (1.0f + expf(-vr)));
            act_tile[em][j * 8 + en] = rnd_to<T>(sg * pr);
          }
        } else {     // down_proj: raw partial sums of this k-range
          const int j = ua - cnt1p + jj;
          if (j < cnt2 && em < p.M) p.slabs[((int64_t)kr * p.M + em) * p.H + (bx + j * p.gpr) * 16 + en] = v;
        }
      }
    }
    __syncthreads();
  };
  for (int ua = 0; ua < utot; ua += 2) pair(ua);
  if (cnt2p == 0) between();  // a workgroup without down_proj tiles (kranges does not divide the grid) still publishes its
                              // act columns and takes part in the hand-offs
  MLP_STAMP(8);
#undef MLP_STAMP
}

inline int mlp_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) cus = 256;
    else cus = prop.multiProcessorCount;
  }
  return cus;
}

}  // namespace

// uint32 words of the sync block the caller zeroes before every launch / floats of the pmax scratch
extern "C" int sgl_mi355_fp8_mlp_block_sync_words(void) { return kSyncWords; }
extern "C" int sgl_mi355_fp8_mlp_block_pmax_words(void) { return mlp_cus() * 32; }

// 1 if sgl_mi355_fp8_mlp_block takes this shape (else the caller keeps the four-launch path)
extern "C" int sgl_mi355_fp8_mlp_block_supported(int M, int hidden, int inter) {
  const int G = mlp_cus();
  if (M < 1 || M > 32 || hidden < 512 || hidden > kMlpRange || hidden % 16 != 0) return 0;
  if (inter < 64 || inter % 16 != 0) return 0;
  const int ntiles1 = inter / 8;
  // (round 5) fewer gate_up tiles than CUs -- a tensor-parallel shard: Llama-3-8B at TP 8 has inter 1 792 = 224 tiles -- run on that
  // many workgroups (one tile each); at least 64 so that the down_proj phase keeps a quarter of the chip streaming
  if (ntiles1 < 64 || ntiles1 > kMlpMaxT1 * G) return 0;
  const int kranges = (inter + kMlpRange - 1) / kMlpRange;
  if (kranges > (ntiles1 < G ? ntiles1 : G)) return 0;
  if ((int64_t)2 * inter * hidden >= 0xFFFFFFF0ll) return 0;
  return 1;
}

// See the header of this file.  w_gate_up_interleaved / scales_gate_up_interleaved: sgl_mi355_gemm_silu_mul's 16-row interleaving
// ([gate rows 8t..8t+7 | up rows 8t..8t+7]).  out_slabs f32 [ceil(inter / 4096), M, hidden]: raw partial sums of down_proj (the
// consumer applies act_scales[m] * down_proj scale[n], e.g. sgl_mi355_fused_add_rmsnorm_quant_fp8 with slabs).  Scratch:
// xq [M, hidden] bytes, xs [M] f32, actq [M, inter] bytes, pmax (sgl_mi355_fp8_mlp_block_pmax_words uint32), sync
// (sgl_mi355_fp8_mlp_block_sync_words uint32, ZEROED by the caller before every launch; word 16 != 0 afterwards = a hand-off
// timed out).  timeline: optional [CUs][8][16] int64 of s_memrealtime stamps (10 ns ticks), NULL in production.
extern "C" int sgl_mi355_fp8_mlp_block(const void* x, void* residual, const void* ln_weight, float eps,
                                       const void* w_gate_up_interleaved, const float* scales_gate_up_interleaved,
                                       const void* w_down, float* out_slabs, float* act_scales, void* xq_scratch,
                                       float* xs_scratch, void* actq_scratch, void* pmax_scratch, void* sync, int M,
                                       int hidden, int inter, int dtype, long long* timeline, void* stream) {
  SGL_CHECK(x && residual && ln_weight && w_gate_up_interleaved && scales_gate_up_interleaved && w_down && out_slabs && act_scales &&
                xq_scratch && xs_scratch && actq_scratch && pmax_scratch && sync,
            "fp8_mlp_block: null pointer");
  SGL_CHECK(dtype == SGL_BF16 || dtype == SGL_F16, "fp8_mlp_block: dtype must be bf16 or f16");
  SGL_CHECK(sgl_mi355_fp8_mlp_block_supported(M, hidden, inter),
            "fp8_mlp_block: unsupported shape M=%d hidden=%d inter=%d (needs M <= 32, hidden a multiple of 16 and <= 4096, "
            "64 <= inter / 8 <= 16 CUs)", M, hidden, inter);
  SGL_CHECK(((uintptr_t)x % 16) == 0 && ((uintptr_t)residual % 16) == 0 && ((uintptr_t)ln_weight % 16) == 0 &&
                ((uintptr_t)w_gate_up_interleaved % 16) == 0 && ((uintptr_t)w_down % 16) == 0 && ((uintptr_t)xq_scratch % 16) == 0 &&
                ((uintptr_t)actq_scratch % 16) == 0 && ((uintptr_t)pmax_scratch % 128) == 0 && ((uintptr_t)sync % 128) == 0,
            "fp8_mlp_block: misaligned pointer");
  MlpParams p;
  p.x = x; p.residual = residual; p.ln_w = ln_weight; p.eps = eps;
  {  // one buffer descriptor over both weights (see the kernel): they must lie within one 4 GiB window
    const uintptr_t a1 = (uintptr_t)w_gate_up_interleaved, a2 = (uintptr_t)w_down, lo = a1 < a2 ? a1 : a2;
    const uint64_t end1 = (a1 - lo) + (uint64_t)2 * inter * hidden, end2 = (a2 - lo) + (uint64_t)inter * hidden;
    const uint64_t span = end1 > end2 ? end1 : end2;
    SGL_CHECK(span < 0xFFFFFFF0ull, "fp8_mlp_block: gate_up and down_proj weights must lie within one 4 GiB window (allocate them "
                                   "from one buffer); they are %llu bytes apart", (unsigned long long)span);
    p.wbase = (const char*)lo; p.wbytes = (unsigned)span; p.w1off = (unsigned)(a1 - lo); p.w2off = (unsigned)(a2 - lo);
  }
  p.sw1 = scales_gate_up_interleaved;
  p.slabs = out_slabs; p.act_scales = act_scales; p.xq = (uint8_t*)xq_scratch; p.xs = xs_scratch; p.actq = (uint8_t*)actq_scratch;
  p.pmax = (uint32_t*)pmax_scratch; p.sync = (uint32_t*)sync; p.tl = timeline;
  p.M = M; p.H = hidden; p.I = inter;
  const int G = inter / 8 < mlp_cus() ? inter / 8 : mlp_cus();   // one workgroup per CU, or per gate_up tile where those are fewer
  p.ntiles1 = inter / 8;
  p.ntiles2 = hidden / 16;
  p.kranges = (inter + kMlpRange - 1) / kMlpRange;
  p.gpr = G / p.kranges;
  hipStream_t st = (hipStream_t)stream;
#define SGL_MLP_LAUNCH(T, MTv, TLv) \
  hipLaunchKernelGGL((fp8_mlp_block_kernel<T, MTv, TLv>), dim3(G), dim3(kMlpWaves * 64), 0, st, p)
#define SGL_MLP_BY_M(T)                         \
  do {                                          \
    if (timeline) {                             \
      if (M <= 16) SGL_MLP_LAUNCH(T, 1, true);  \
      else SGL_MLP_LAUNCH(T, 2, true);          \
    } else {                                    \
      if (M <= 16) SGL_MLP_LAUNCH(T, 1, false); \
      else SGL_MLP_LAUNCH(T, 2, false);         \
    }                                           \
  } while (0)
  if (dtype == SGL_BF16) SGL_MLP_BY_M(__bf16);
  else SGL_MLP_BY_M(_Float16);
#undef SGL_MLP_BY_M
#undef SGL_MLP_LAUNCH
  SGL_HIP_LAUNCH_CHECK();
  return SGL_MI355_OK;
}
