"""Host-side mirror of the reference's plugin surface for the hot path (python/sglang/srt/...).

Module paths below ``srt`` follow the reference tree so that a maintainer can map each file:
  model_executor/forward_batch_info.py  ForwardMode / ForwardBatch (the per-batch contract)
  layers/attention/base_attn_backend.py AttentionBackend ABC
  layers/attention/hip_backend.py       the MI355X backend (drop-in for --attention-backend)
  layers/radix_attention.py             RadixAttention layer (carrier of per-layer constants)
  mem_cache/                            ReqToTokenPool, MHATokenToKVPool, allocators, radix cache
  layers/quantization/                  W8A8Fp8 / Fp8 / AWQ / unquantised linear methods
"""
