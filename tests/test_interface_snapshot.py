"""Drop-in proof at the signature level: this build's plugin surfaces against a snapshot of
the reference's, captured from its source by tests/golden/make_golden.py::gen_interface (tests/golden/interface.json).

Rule for every callable: the reference's parameters appear in this build with the same names, in the same order and with
the same defaults (so every call the reference's callers make binds identically); this build may add trailing parameters
only if they have defaults.  ForwardBatch: every field this build keeps exists in the reference with the same default, and
every field the attention path reads is kept."""
import dataclasses
import inspect
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SNAP = json.load(open(os.path.join(ROOT, "tests", "golden", "interface.json")))

gpu_free = pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "ltp-sglang_amd", "lib", "libsgl_mi355.so")),
                              reason="the package imports the C-ABI library; build it first")


def _params(fn):
    out = []
    for p in inspect.signature(fn).parameters.values():
        name = ("*" if p.kind is p.VAR_POSITIONAL else "**" if p.kind is p.VAR_KEYWORD else "") + p.name
        out.append([name, None if p.default is p.empty else repr(p.default)])
    return out


def _norm_default(d):
    if d is None:
        return None
    d = d.replace('"', "'")
    # (fp8_dtype: the module constant of fp8_kernel.py:72-85 -- float8_e4m3fn on every target but gfx94x, and on gfx950)
    return {"AttentionType.DECODER": "<AttentionType.DECODER: 'decoder'>", "fp8_dtype": "torch.float8_e4m3fn"}.get(d, d)


def _check(ref_params, fn, where):
    mine = _params(fn)
    ref = [[n, _norm_default(d)] for n, d in ref_params]
    if ref and ref[0][0] == "cls" and (not mine or mine[0][0] != "cls"):
        ref = ref[1:]   # a classmethod looked up on the class is already bound
    variadic_ref = [p for p in ref if p[0].startswith("*")]
    fixed_ref = [p for p in ref if not p[0].startswith("*")]
    fixed_mine = [p for p in mine if not p[0].startswith("*")]
    assert len(fixed_mine) >= len(fixed_ref), f"{where}: missing parameters {fixed_ref[len(fixed_mine):]}"
    for i, (rp, mp) in enumerate(zip(fixed_ref, fixed_mine)):
        assert rp[0] == mp[0], f"{where}: parameter {i} is '{mp[0]}', the reference has '{rp[0]}'"
        if rp[1] is not None and rp[1].endswith("()"):   # a default computed at import by a call: any default binds the same
            assert mp[1] is not None, f"{where}: '{rp[0]}' needs a default"
            continue
        assert rp[1] == mp[1], f"{where}: default of '{rp[0]}' is {mp[1]}, the reference has {rp[1]}"
    for extra in fixed_mine[len(fixed_ref):]:
        assert extra[1] is not None, f"{where}: extra parameter '{extra[0]}' has no default"
    for v in variadic_ref:
        assert v in [[p[0], None] for p in mine if p[0].startswith("*")], f"{where}: missing {v[0]}"


@gpu_free
@pytest.mark.parametrize("cls_name,module,only", [
    ("AttentionBackend", "srt.layers.attention.base_attn_backend", None),
    ("QuantizeMethodBase", "srt.layers.quantization.base_config", None),
    ("LinearMethodBase", "srt.layers.quantization.base_config", None),
    ("QuantizationConfig", "srt.layers.quantization.base_config",
     {"__init__", "get_name", "get_supported_act_dtypes", "from_config", "get_quant_method", "get_from_keys"}),
    ("RadixAttention", "srt.layers.radix_attention", None),
    ("ReqToTokenPool", "srt.mem_cache.memory_pool", None),
    ("MHATokenToKVPool", "srt.mem_cache.memory_pool", None),
    ("TokenToKVPoolAllocator", "srt.mem_cache.allocator", {"__init__", "alloc", "free", "clear", "available_size"}),
    ("RadixCache", "srt.mem_cache.radix_cache", None),
])
def test_class_surface_matches_reference(cls_name, module, only, pkg):
    import importlib

    cls = getattr(importlib.import_module("ltp_sglang_amd." + module), cls_name)
    for meth, ref_params in SNAP[cls_name].items():
        if only is not None and meth not in only:
            continue
        assert hasattr(cls, meth), f"{cls_name}.{meth} is missing"
        _check(ref_params, getattr(cls, meth), f"{cls_name}.{meth}")   # bound class/static methods drop cls themselves


@gpu_free
def test_hip_backend_constructs_like_the_triton_backend(pkg):
    from ltp_sglang_amd.srt.layers.attention.hip_backend import HipAttnBackend

    for meth, ref_params in SNAP["TritonAttnBackend"].items():
        _check(ref_params, getattr(HipAttnBackend, meth), f"HipAttnBackend.{meth}")


@gpu_free
def test_sgl_kernel_op_signatures(pkg):
    from ltp_sglang_amd import sgl_kernel
    from ltp_sglang_amd.srt.layers.quantization.fp8_utils import apply_fp8_linear

    for group in ("sgl_kernel.gemm", "sgl_kernel.attention", "sgl_kernel.elementwise", "triton_ops"):
        for name, ref_params in SNAP[group].items():
            assert hasattr(sgl_kernel, name), f"sgl_kernel.{name} is missing"
            _check(ref_params, getattr(sgl_kernel, name), f"sgl_kernel.{name}")
    _check(SNAP["apply_fp8_linear"], apply_fp8_linear, "apply_fp8_linear")
    # a19's named helpers (fp8_kernel.py:437, fp8_utils.py:310; called by w8a8_fp8.py:129, fp8.py:375, fp8_utils.py:658)
    from ltp_sglang_amd.srt.layers.quantization import fp8_kernel, fp8_utils
    _check(SNAP["fp8_helpers"]["static_quant_fp8"], fp8_kernel.static_quant_fp8, "static_quant_fp8")
    _check(SNAP["fp8_helpers"]["input_to_float8"], fp8_utils.input_to_float8, "input_to_float8")
    assert fp8_utils.static_quant_fp8 is fp8_kernel.static_quant_fp8   # (fp8_utils.py:23 imports it)


@gpu_free
def test_native_attention_op_schemas(pkg):
    """decode_attention_cpu / extend_attention_cpu (torch_extension_cpu.cpp:264-275): same argument names, order and kinds."""
    from ltp_sglang_amd import sgl_kernel

    for op, fn in (("decode_attention_cpu", sgl_kernel.decode_attention), ("extend_attention_cpu", sgl_kernel.extend_attention)):
        schema = SNAP["cpu_op_schemas"][op]
        args = re.search(r"\((.*)\)\s*->", schema).group(1).split(",")
        names = [a.split()[-1] for a in args]
        names = ["v_cache" if n == "v_cahce" else n for n in names]   # (sic) the reference schema misspells it
        assert [p[0] for p in _params(fn)] == names, (op, [p[0] for p in _params(fn)], names)


@gpu_free
def test_forward_batch_and_mode(pkg):
    from ltp_sglang_amd.srt.model_executor.forward_batch_info import ForwardBatch, ForwardMode

    ref_fields = {n: (ann, dflt) for n, ann, dflt in SNAP["ForwardBatch.fields"]}
    mine = dataclasses.fields(ForwardBatch)
    for f in mine:
        assert f.name in ref_fields, f"ForwardBatch.{f.name} does not exist in the reference"
        ref_default = ref_fields[f.name][1]
        my_default = None if f.default is dataclasses.MISSING else repr(f.default)
        assert my_default == ref_default, f"ForwardBatch.{f.name}: default {my_default}, the reference has {ref_default}"
    # required (no-default) fields come first and in the reference's order, so positional construction binds identically
    ref_required = [n for n, _, d in SNAP["ForwardBatch.fields"] if d is None]
    my_required = [f.name for f in mine if f.default is dataclasses.MISSING]
    assert my_required == ref_required, (my_required, ref_required)
    for needed in ("forward_mode", "batch_size", "req_pool_indices", "seq_lens", "seq_lens_sum", "out_cache_loc", "extend_seq_lens",
                   "extend_prefix_lens", "spec_info", "token_to_kv_pool", "req_to_token_pool", "attn_backend", "positions",
                   "seq_lens_cpu", "extend_start_loc", "encoder_lens"):   # SURVEY 8b "Reads from forward_batch"
        assert needed in {f.name for f in mine}
    assert [m.name for m in ForwardMode] == SNAP["ForwardMode.members"]
    for meth in SNAP["ForwardMode.methods"]:
        assert hasattr(ForwardMode, meth), f"ForwardMode.{meth} is missing"


@gpu_free
def test_torch_ops_registered_with_the_reference_schemas(pkg):
    """torch.ops.sgl_kernel.<op>.default -- what the reference's Python wrappers call (gemm.py:34-42) -- exists with the
    reference's m.def schema (common_extension.cc / torch_extension_cpu.cpp) and has no CPU kernel (fails loudly)."""
    import torch

    from ltp_sglang_amd import sgl_kernel  # noqa: F401  (registers)

    def canon(schema):
        return ([(a.name, str(a.type), bool(a.alias_info and a.alias_info.is_write), a.has_default_value()) for a in schema.arguments],
                [str(r.type) for r in schema.returns])

    for name, ref_schema in {**SNAP["torch_op_schemas"], **SNAP["cpu_op_schemas"]}.items():
        op = getattr(torch.ops.sgl_kernel, name).default
        assert canon(op._schema) == canon(torch._C.parse_schema(ref_schema)), (name, str(op._schema), ref_schema)
    with pytest.raises(NotImplementedError):
        torch.ops.sgl_kernel.sgl_per_token_quant_fp8.default(torch.zeros(2, 8, dtype=torch.bfloat16),
                                                             torch.zeros(2, 8).to(torch.float8_e4m3fn), torch.zeros(2))
