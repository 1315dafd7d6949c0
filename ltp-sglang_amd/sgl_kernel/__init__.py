"""MI355X implementation of the ``sgl_kernel`` op API for the attention / dequant-GEMM hot path.

Export list follows sgl-kernel/python/sgl_kernel/__init__.py:12-96 restricted to the ops on
the path, plus the attention and KV-index ops the reference only has as
Triton / CPU kernels.
"""
from .attention import decode_attention, decode_attention_fwd, extend_attention, extend_attention_fwd
from .elementwise import (
    apply_rope_with_cos_sin_cache_inplace,
    argmax,
    embedding,
    fused_add_rmsnorm,
    rmsnorm,
    silu_and_mul,
)
from .fused import decode_merge_quant_fp8, fused_add_rmsnorm_quant_fp8, rope_set_kv, silu_and_mul_quant_fp8
from .gemm import (
    awq_dequantize,
    dense_linear,
    fp8_linear_slabs,
    fp8_scaled_mm,
    sgl_per_tensor_quant_fp8,
    sgl_per_token_group_quant_fp8,
    sgl_per_token_quant_fp8,
)
from .kvcache import (
    compute_position,
    create_kv_indices,
    decode_metadata,
    get_last_loc,
    set_kv_buffer,
    write_req_to_token,
)
