"""Registers this build's ops in the ``sgl_kernel`` torch-library namespace with the reference's schemas, so that the
reference's own Python wrappers -- which call ``torch.ops.sgl_kernel.<op>.default(...)``
(sgl-kernel/python/sgl_kernel/gemm.py:34-42,100-146, elementwise.py, attention.py:12-52) -- resolve to the C-ABI kernels.

Schemas are the reference's ``m.def`` strings (sgl-kernel/csrc/common_extension.cc:56-59,69-82,98-130 and, for the two native
attention ops, csrc/cpu/torch_extension_cpu.cpp:264-275); kernels are registered for the CUDA dispatch key, which is the key
HIP tensors dispatch on (torch_extension_rocm.cc:21-27).  Importing ``ltp_sglang_amd.sgl_kernel`` runs register() once; if a
library that already defines an op is loaded in the same process the definition is left alone and only this build's kernel
is installed for the CUDA key.
"""
import torch

from . import attention as _attn
from . import elementwise as _ew
from . import gemm as _gemm

_SCHEMAS = {
    "fp8_scaled_mm": "(Tensor mat_a, Tensor mat_b, Tensor scales_a, Tensor scales_b, ScalarType out_dtype, Tensor? bias) -> Tensor",
    "sgl_per_token_group_quant_fp8": "(Tensor input, Tensor output_q, Tensor output_s, int group_size, float eps, float fp8_min, float fp8_max, bool scale_ue8m0) -> ()",
    "sgl_per_tensor_quant_fp8": "(Tensor input, Tensor output_q, Tensor output_s, bool is_static) -> ()",
    "sgl_per_token_quant_fp8": "(Tensor input, Tensor output_q, Tensor output_s) -> ()",
    "awq_dequantize": "(Tensor qweight, Tensor scales, Tensor qzeros) -> Tensor",
    "merge_state": "(Tensor v_a, Tensor s_a, Tensor v_b, Tensor s_b, Tensor! v_merged, Tensor! s_merged) -> ()",
    "merge_state_v2": "(Tensor v_a, Tensor s_a, Tensor v_b, Tensor s_b, Tensor! v_merged, Tensor! s_merged) -> ()",
    "rmsnorm": "(Tensor! output, Tensor input, Tensor weight, float eps, bool enable_pdl) -> ()",
    "fused_add_rmsnorm": "(Tensor! input, Tensor! residual, Tensor weight, float eps, bool enable_pdl) -> ()",
    "silu_and_mul": "(Tensor! out, Tensor input) -> ()",
    "decode_attention_cpu": "(Tensor query, Tensor k_cache, Tensor v_cahce, Tensor output, Tensor key, Tensor value, Tensor loc, "
                            "Tensor attn_logits, Tensor req_to_token, Tensor req_pool_indices, Tensor seq_lens, float sm_scale, "
                            "float logit_cap) -> ()",
    "extend_attention_cpu": "(Tensor q_extend, Tensor k_extend, Tensor v_extend, Tensor o_extend, Tensor k_buffer, Tensor v_buffer, "
                            "Tensor req_to_token, Tensor req_pool_indices, Tensor seq_lens, Tensor extend_seq_lens, "
                            "Tensor extend_start_loc, int max_len_extend, float sm_scale, float logit_cap) -> ()",
}


def _merge_state(v_a, s_a, v_b, s_b, v_merged, s_merged):
    _attn.merge_state(v_a, s_a, v_b, s_b, v_merged, s_merged)


def _rmsnorm(output, input, weight, eps, enable_pdl):
    _ew.rmsnorm(input, weight, eps, out=output)


def _fused_add_rmsnorm(input, residual, weight, eps, enable_pdl):
    _ew.fused_add_rmsnorm(input, residual, weight, eps)


def _silu_and_mul(out, input):
    _ew.silu_and_mul(input, out)


_IMPLS = {
    "fp8_scaled_mm": _gemm.fp8_scaled_mm,
    "sgl_per_token_group_quant_fp8": _gemm.sgl_per_token_group_quant_fp8,
    "sgl_per_tensor_quant_fp8": _gemm.sgl_per_tensor_quant_fp8,
    "sgl_per_token_quant_fp8": _gemm.sgl_per_token_quant_fp8,
    "awq_dequantize": _gemm.awq_dequantize,
    "merge_state": _merge_state,
    "merge_state_v2": _merge_state,
    "rmsnorm": _rmsnorm,
    "fused_add_rmsnorm": _fused_add_rmsnorm,
    "silu_and_mul": _silu_and_mul,
    "decode_attention_cpu": _attn.decode_attention,
    "extend_attention_cpu": _attn.extend_attention,
}

_LIB = None


def register():
    """Idempotent; returns the names registered."""
    global _LIB
    if _LIB is not None:
        return sorted(_IMPLS)
    _LIB = torch.library.Library("sgl_kernel", "FRAGMENT")
    for name, schema in _SCHEMAS.items():
        try:
            _LIB.define(name + schema)
        except RuntimeError as e:   # defined by another loaded library: keep its schema, install our CUDA kernel
            if "already" not in str(e) and "multiple times" not in str(e):
                raise
        _LIB.impl(name, _IMPLS[name], "CUDA")
    return sorted(_IMPLS)
