"""MI355X implementation of the ``sgl_kernel`` op API for the attention / dequant-GEMM hot path.

Export list follows sgl-kernel/python/sgl_kernel/__init__.py:12-96 restricted to the ops on
the path, plus the attention and KV-index ops the reference only has as
Triton / CPU kernels.
"""
from .attention import (decode_attention, decode_attention_cascade, decode_attention_fwd, decode_attention_merge_quant, extend_attention,
                        extend_attention_fwd, merge_state, merge_state_v2)
from .elementwise import (
    apply_rope_with_cos_sin_cache_inplace,
    argmax,
    embedding,
    fused_add_rmsnorm,
    rmsnorm,
    silu_and_mul,
)
from .fused import (
    decode_merge_quant_fp8,
    balanced_tile_rows,
    awq_gate_up_col_order,
    awq_rope_col_order,
    awq_permute_cols,
    awq_gemm_silu_mul,
    awq_qkv_rope_set_kv,
    awq_gemm_slabs,
    fp8_gemm_silu_mul,
    fp8_mlp_block,
    fp8_mlp_block_supported,
    fp8_mlp_block_pack_weights,
    Fp8MlpBlockScratch,
    fp8_qkv_rope_set_kv,
    gemm_silu_mul,
    qkv_rope_set_kv,
    fused_add_rmsnorm_quant_fp8,
    interleave_gate_up_rows,
    interleave_rope_rows,
    rope_set_kv,
    silu_and_mul_quant_fp8,
    silu_table_init,
)
from .gemm import (
    awq_dequantize,
    awq_gemm,
    awq_gemm_num_kranges,
    awq_repack,
    awq_set_exact_weights,
    awq_unpack_nk,
    dense_linear,
    dense_linear_kranges,
    fp8_gemm_num_slabs,
    fp8_gemm_slabs,
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
    decode_prepare,
    decode_schedule,
    decode_schedule_units,
    get_last_loc,
    move_kv_cache,
    set_kv_buffer,
    write_req_to_token,
)
from .torch_ops import register as register_torch_ops

register_torch_ops()   # torch.ops.sgl_kernel.<op>: what the reference's Python wrappers call (gemm.py:34-42)


def sglang_per_token_quant_fp8(x):
    """x [M, K] -> (x_q e4m3fn, x_s f32 [M, 1]); fp8_kernel.py:375-391."""
    import torch

    x_q = torch.empty_like(x, dtype=torch.float8_e4m3fn)
    x_s = torch.empty(x.shape[0], 1, device=x.device, dtype=torch.float32)
    sgl_per_token_quant_fp8(x, x_q, x_s)
    return x_q, x_s
