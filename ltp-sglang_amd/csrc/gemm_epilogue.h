// Fused epilogues of the decode-sized GEMMs (skinny_gemm.hip: fp8 / bf16 / f16 operands; awq.hip: int4 weights), shared so that
// both produce the same bits from the same f32 sums.
#pragma once
#include "common.h"

namespace {

// Fused epilogues (single k-range only).  Both rely on an INTERLEAVED weight row order so that the two values an output
// needs sit in one 16-row tile, 8 columns apart (thread en and en ^ 8 of the same row m exchange them with one shuffle):
//   EPI_SILU: tile t = [gate rows 8t..8t+7 | up rows 8t..8t+7]  -> act[m][8t+j] = T(T(silu(gate)) * up)
//             (gate_up_proj + SiluAndMul, models/llama.py:94-98, activation.py:60-63)
//   EPI_ROPE: inside every q/k head, tile u = [rows 8u..8u+7 | rows 64+8u..64+8u+7] (the neox rotation pairs); q is
//             written rotated to q_out, k rotated and v straight into the KV pool rows loc[m] (16-bit pool, or float8_e4m3fn
//             with set_kv_buffer's conversion)
//             (qkv_proj -> rotary_emb -> set_kv_buffer, models/llama.py:180-191, rotary_embedding.py:49-72,
//             memory_pool.py:401-407).  head_dim = rot_dim = 128.
// Every rounding point of the unfused op sequence is kept (GEMM output -> T, each product -> T), so results are
// bit-identical to running the separate kernels.
enum { EPI_NONE = 0, EPI_SILU = 1, EPI_ROPE = 2 };
struct EpiParams {
  const int64_t* positions;
  const float* cos_sin;  // [max_pos, 128]: cos | sin
  const int64_t* loc;
  void* k_buf;
  void* v_buf;
  int64_t k_slot_stride, v_slot_stride;  // elements
  int hq, hkv;
  int kv_fp8 = 0;                        // the pool is float8_e4m3fn (kv_cache_dtype fp8_e4m3): convert as set_kv_buffer does
  float k_scale = -1.f, v_scale = -1.f;  // RadixAttention.k_scale / v_scale; <= 0: none
};

template <typename T>
__device__ __forceinline__ float rnd_to(float x) {
  asm volatile("" : "+v"(x));  // materialise the f32 first: no single-rounding v_fma_mix shortcut
  const T t = (T)x;
  uint16_t u = __builtin_bit_cast(uint16_t, t), v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(u));
  return (float)__builtin_bit_cast(T, v);
}

// One output element of an interleaved tile: thread (em, en) holds v = the finished f32 value (scales and bias applied) of row
// em, column n0 + en; its partner column (en ^ H, H = rpt / 2) sits in the same wave.  Must be called by all lanes of the wave
// (`live` masks the stores, not the shuffle).  cos_v / sin_v: the cos_sin row of this row's position at index H * u + (en & (H - 1));
// loc: this row's pool slot.
template <typename OutT, int EPI>
__device__ __forceinline__ void epi_store(float v, bool live, int em, int en, int n0, int rpt, const EpiParams& ep, void* y,
                                          int64_t y_stride, float cos_v, float sin_v, int64_t loc) {
  static_assert(EPI == EPI_SILU || EPI == EPI_ROPE, "fused epilogues only");
  const float vr = rnd_to<OutT>(v);             // the GEMM's own output rounding
  const int H = rpt >> 1;                       // rows of each half of an interleaved tile (8, or 4 for 8-row tiles)
  const float pr = H == 8 ? lane_xor8(vr) : lane_xor4(vr);   // the partner column of the same row m (DPP: no LDS round trip)
  const bool lo = (en & H) == 0;                // first half of the tile (gate / rotation-pair index i)
  if constexpr (EPI == EPI_SILU) {
    if (live && lo) {
      const float sg = rnd_to<OutT>(vr / (1.0f + expf(-vr)));
      ((OutT*)y)[(int64_t)em * y_stride + (n0 >> 1) + en] = (OutT)rnd_to<OutT>(sg * pr);
    }
  } else {
    const int head = n0 >> 7, u = (n0 & 127) / rpt;
    if (live) {
      if (head < ep.hq + ep.hkv) {
        const int i = H * u + (en & (H - 1));
        const float c = rnd_to<OutT>(cos_v), sn = rnd_to<OutT>(sin_v);
        const float x1 = lo ? vr : pr, x2 = lo ? pr : vr;
        const float o = lo ? rnd_to<OutT>(x1 * c) - rnd_to<OutT>(x2 * sn) : rnd_to<OutT>(x2 * c) + rnd_to<OutT>(x1 * sn);
        const int col = i + (lo ? 0 : 64);
        if (head < ep.hq)
          ((OutT*)y)[(int64_t)em * y_stride + head * 128 + col] = (OutT)o;
        else if (ep.kv_fp8)
          ((uint8_t*)ep.k_buf)[loc * ep.k_slot_stride + (head - ep.hq) * 128 + col] = kv_fp8_byte<OutT>(rnd_to<OutT>(o), ep.k_scale);
        else
          ((OutT*)ep.k_buf)[loc * ep.k_slot_stride + (head - ep.hq) * 128 + col] = (OutT)o;
      } else if (ep.kv_fp8) {
        ((uint8_t*)ep.v_buf)[loc * ep.v_slot_stride + (head - ep.hq - ep.hkv) * 128 + (n0 & 127) + en] = kv_fp8_byte<OutT>(vr, ep.v_scale);
      } else {
        ((OutT*)ep.v_buf)[loc * ep.v_slot_stride + (head - ep.hq - ep.hkv) * 128 + (n0 & 127) + en] = (OutT)vr;
      }
    }
  }
}

}  // namespace
