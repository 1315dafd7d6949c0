"""Llama-shaped synthetic stack + a ModelRunner-shaped harness around the hot path.

This is the CALLER side of the path (python/sglang/srt/models/llama.py:94-98,118-191,245-268,308-340 and the
``bench_one_batch`` flow, python/sglang/bench_one_batch.py:214-269): random-weight layers of the Llama / Qwen2
architecture whose every device op is one of this build's HIP kernels, the reference's own pools and allocator
contracts, and the extend / decode drivers that build ForwardBatch objects the way ScheduleBatch.prepare_for_extend /
prepare_for_decode do (schedule_batch.py:1123-1310,1520-1590).  No checkpoint loading, tokenizer or scheduler: the
north star measures synthetic random-weight batches.
"""
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional

import torch
from torch import nn

from ... import sgl_kernel as K
from ..distributed.communication_op import get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size
from ..layers.activation import SiluAndMul
from ..layers.attention.hip_backend import HipAttnBackend
from ..layers.layernorm import RMSNorm
from ..layers.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from ..layers.quantization import get_quantization_config
from ..layers.quantization.w8a8_fp8 import per_channel_quant_fp8
from ..layers.radix_attention import RadixAttention
from ..layers.rotary_embedding import get_rope
from ..mem_cache.allocator import TokenToKVPoolAllocator
from ..mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool
from .forward_batch_info import ForwardBatch, ForwardMode


@dataclass
class LlamaShape:
    hidden_size: int = 4096
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    head_dim: int = 128
    num_hidden_layers: int = 32
    intermediate_size: int = 14336
    vocab_size: int = 128256
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    max_position_embeddings: int = 8192
    attention_bias: bool = False  # Qwen2 sets True (qkv bias)

    @staticmethod
    def llama3_8b():
        return LlamaShape()

    @staticmethod
    def llama3_70b():
        return LlamaShape(hidden_size=8192, num_attention_heads=64, num_key_value_heads=8, num_hidden_layers=80,
                          intermediate_size=28672)

    @staticmethod
    def qwen2_7b():
        return LlamaShape(hidden_size=3584, num_attention_heads=28, num_key_value_heads=4, num_hidden_layers=28,
                          intermediate_size=18944, vocab_size=152064, rms_norm_eps=1e-6, rope_theta=1e6,
                          max_position_embeddings=32768, attention_bias=True)

    @staticmethod
    def tiny(layers=2):
        return LlamaShape(hidden_size=512, num_attention_heads=8, num_key_value_heads=2, head_dim=64,
                          num_hidden_layers=layers, intermediate_size=1024, vocab_size=2048, max_position_embeddings=1024)


class LlamaMLP(nn.Module):
    def __init__(self, cfg: LlamaShape, quant_config, dtype, prefix=""):
        super().__init__()
        self.gate_up_proj = MergedColumnParallelLinear(cfg.hidden_size, [cfg.intermediate_size] * 2, bias=False,
                                                       quant_config=quant_config, params_dtype=dtype, prefix=f"{prefix}.gate_up_proj")
        self.down_proj = RowParallelLinear(cfg.intermediate_size, cfg.hidden_size, bias=False, quant_config=quant_config,
                                           params_dtype=dtype, prefix=f"{prefix}.down_proj")
        self.act_fn = SiluAndMul()

    def forward(self, x):
        gate_up, _ = self.gate_up_proj(x)
        x, _ = self.down_proj(self.act_fn(gate_up))
        return x


class LlamaAttention(nn.Module):
    def __init__(self, cfg: LlamaShape, layer_id: int, quant_config, dtype, prefix=""):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.num_heads = cfg.num_attention_heads // tp
        self.num_kv_heads = max(1, cfg.num_key_value_heads // tp)
        self.head_dim = cfg.head_dim
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.qkv_proj = QKVParallelLinear(cfg.hidden_size, self.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads,
                                          bias=cfg.attention_bias, quant_config=quant_config, params_dtype=dtype,
                                          prefix=f"{prefix}.qkv_proj")
        self.o_proj = RowParallelLinear(cfg.num_attention_heads * self.head_dim, cfg.hidden_size, bias=False,
                                        quant_config=quant_config, params_dtype=dtype, prefix=f"{prefix}.o_proj")
        self.rotary_emb = get_rope(self.head_dim, self.head_dim, cfg.max_position_embeddings, cfg.rope_theta, True, dtype=dtype)
        self.attn = RadixAttention(self.num_heads, self.head_dim, self.head_dim ** -0.5, self.num_kv_heads, layer_id)
        self.fused_rope_kv = True   # decode: rotary_emb + set_kv_buffer in one launch (bit-identical; 16-bit KV pools)

    def kv_scales(self):
        """(k_scale, v_scale) as set_kv_buffer takes them: the layer's scales or None."""
        a = self.attn
        return (None if a.k_scale is None else float(a.k_scale)), (None if a.v_scale is None else float(a.v_scale))

    def rope_and_write_kv(self, positions, q, k, v, forward_batch):
        """rotary_emb on q, k in place, then (k, v) into the pool rows out_cache_loc: one launch for 16-bit pools, the
        rotary kernel + the converting scatter for float8_e4m3fn pools."""
        pool, lid = forward_batch.token_to_kv_pool, self.attn.layer_id
        if pool.dtype == torch.float8_e4m3fn:
            self.rotary_emb(positions, q, k)
            pool.set_kv_buffer(self.attn, forward_batch.out_cache_loc, k, v, self.attn.k_scale, self.attn.v_scale)
        else:
            K.rope_set_kv(positions, q, k, v, self.head_dim, self.rotary_emb.cos_sin_cache, True, pool.get_key_buffer(lid),
                          pool.get_value_buffer(lid), forward_batch.out_cache_loc)

    def forward(self, positions, hidden_states, forward_batch):
        qkv, _ = self.qkv_proj(hidden_states)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        pool = forward_batch.token_to_kv_pool
        if self.fused_rope_kv and forward_batch.forward_mode.is_decode() and pool.dtype != torch.float8_e4m3fn:
            self.rope_and_write_kv(positions, q, k, v, forward_batch)
            attn_output = self.attn(q, k, v, forward_batch, save_kv_cache=False)
            output, _ = self.o_proj(attn_output)
            return output
        q, k = self.rotary_emb(positions, q, k)
        attn_output = self.attn(q, k, v, forward_batch)
        output, _ = self.o_proj(attn_output)
        return output


class LlamaDecoderLayer(nn.Module):
    def __init__(self, cfg: LlamaShape, layer_id: int, quant_config, dtype):
        super().__init__()
        self.self_attn = LlamaAttention(cfg, layer_id, quant_config, dtype, prefix=f"layers.{layer_id}.self_attn")
        self.mlp = LlamaMLP(cfg, quant_config, dtype, prefix=f"layers.{layer_id}.mlp")
        self.input_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype)

    def forward(self, positions, hidden_states, forward_batch, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states, forward_batch)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        hidden_states = self.mlp(hidden_states)
        return hidden_states, residual


class LlamaForCausalLM(nn.Module):
    """embed -> L x decoder layer -> norm -> lm_head (last token of each request) -> logits."""

    def __init__(self, cfg: LlamaShape, quantization: Optional[str], dtype=torch.bfloat16):
        super().__init__()
        self.cfg, self.dtype = cfg, dtype
        self.fused_decode = True
        self.fused_extend = True
        self.pad_qkv_rows = True   # forward_extend_fused: qkv output rows padded by 256 bytes (measurement hook: False)
        self.fused_attn_merge = True   # decode: stage-2 merge + quant by the last-arriving workgroup of each request
        self.fused_epilogues = True
        # decode, M <= 32: post-attention norm -> gate_up + SiluAndMul -> quant -> down_proj as ONE persistent launch
        # (csrc/mlp_block.hip) instead of four.  Measured 2.5 % slower than the four launches at the headline shape, so
        # off unless SGL_MI355_MLP_BLOCK=1
        self.fused_mlp_block = os.environ.get("SGL_MI355_MLP_BLOCK", "0") != "0"
        self.prefill_silu_epilogue = os.environ.get("SGL_MI355_PREFILL_SILU", "1") != "0"   # gate_up + SiluAndMul in the prefill GEMM's epilogue
        self._mlp_scratch = {}
        # decode, M <= 64: o_proj in this many k-ranges, summed by the add + RMSNorm that follows (1 = one launch with the plain epilogue)
        self.o_proj_kranges = int(os.environ.get("SGL_MI355_OPROJ_KRANGES", "1"))
        qc = None
        if quantization is not None and not isinstance(quantization, str):
            qc = quantization   # a QuantizationConfig instance (e.g. a serialized-checkpoint config)
        elif quantization is not None:
            cls = get_quantization_config(quantization)
            qc = cls(4, 128, True) if quantization == "awq" else cls()
        self.quant_config = qc
        self.embed_tokens = nn.Parameter(torch.empty(cfg.vocab_size, cfg.hidden_size, dtype=dtype), requires_grad=False)
        self.layers = nn.ModuleList([LlamaDecoderLayer(cfg, i, qc, dtype) for i in range(cfg.num_hidden_layers)])
        self.norm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype)
        # lm_head is vocab-sharded over TP ranks and the logits all-gathered (logits_processor.py:471-500);
        # the embedding table is replicated (a row gather needs no collective)
        self.tp_size, self.tp_rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        assert cfg.vocab_size % self.tp_size == 0
        self.lm_head = nn.Parameter(torch.empty(cfg.vocab_size // self.tp_size, cfg.hidden_size, dtype=dtype), requires_grad=False)

    @torch.no_grad()
    def init_random_weights(self, seed: int = 0, std: float = 0.02):
        """bf16 N(0, std) weights, then each linear's own process_weights_after_loading (fp8 per-channel quantiser:
        w8a8_fp8.py:119-126; AWQ: random int32 packs + U(0,1)*0.01 scales as test_awq_dequant.py:71-100)."""
        dev = self.embed_tokens.device
        g = torch.Generator(device=dev).manual_seed(seed)

        def randn(shape, s=std):
            return (torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * s).to(self.dtype)

        self.embed_tokens.copy_(randn(self.embed_tokens.shape, 1.0))
        vs = self.cfg.vocab_size // self.tp_size
        self.lm_head.copy_(randn((self.cfg.vocab_size, self.cfg.hidden_size))[self.tp_rank * vs : (self.tp_rank + 1) * vs])
        for mod in self.modules():
            if isinstance(mod, RMSNorm):
                mod.weight.copy_((1.0 + 0.1 * torch.randn(mod.weight.shape, generator=g, device=dev)).to(self.dtype))
            qm = getattr(mod, "quant_method", None)
            if qm is None or not hasattr(mod, "input_size"):
                continue
            if hasattr(mod, "qweight"):  # AWQ
                mod.qweight.copy_(torch.randint(0, 2**31 - 1, mod.qweight.shape, generator=g, device=dev, dtype=torch.int32))
                mod.qzeros.copy_(torch.randint(0, 2**31 - 1, mod.qzeros.shape, generator=g, device=dev, dtype=torch.int32))
                mod.scales.copy_((torch.rand(mod.scales.shape, generator=g, device=dev) * 0.01 * 0.3).to(self.dtype))
            elif hasattr(mod, "shard_cols"):  # row parallel: every rank draws the full matrix and keeps its K slice
                mod.weight.copy_(mod.shard_cols(randn((mod.output_size, mod.input_size))))
            else:                             # column parallel (incl. merged / qkv): keep this rank's rows
                full_rows = (mod.total_num_heads + 2 * mod.total_num_kv_heads) * mod.head_size if hasattr(mod, "total_num_heads") else mod.output_size
                mod.weight.copy_(mod.shard_rows(randn((full_rows, mod.input_size))))
            if getattr(mod, "bias", None) is not None:
                if hasattr(mod, "shard_rows"):
                    full_rows = (mod.total_num_heads + 2 * mod.total_num_kv_heads) * mod.head_size if hasattr(mod, "total_num_heads") else mod.output_size
                    mod.bias.copy_(mod.shard_rows(randn((full_rows,))))
                else:
                    mod.bias.copy_(randn(mod.bias.shape))
            qm.process_weights_after_loading(mod)

    @torch.no_grad()
    def load_checkpoint(self, ckpt: dict):
        """Loads an unsharded in-memory checkpoint (TP 1): ``embed``, ``lm_head``, ``norm`` and per layer ``ln1``, ``ln2`` and
        ``qkv`` / ``o`` / ``gate_up`` / ``down`` dicts whose keys are the parameter names each linear method's
        create_weights registered (weight, weight_scale, input_scale, qweight, qzeros, scales, bias), then runs
        process_weights_after_loading -- the two calls a model loader makes (model_loader/loader.py).  Used by the G7 test."""
        assert self.tp_size == 1, "load_checkpoint takes unsharded tensors"
        dev = self.embed_tokens.device
        self.embed_tokens.copy_(ckpt["embed"].to(dev))
        self.lm_head.copy_(ckpt["lm_head"].to(dev))
        self.norm.weight.copy_(ckpt["norm"].to(dev))
        for layer, L in zip(self.layers, ckpt["layers"]):
            layer.input_layernorm.weight.copy_(L["ln1"].to(dev))
            layer.post_attention_layernorm.weight.copy_(L["ln2"].to(dev))
            for mod, t in ((layer.self_attn.qkv_proj, L["qkv"]), (layer.self_attn.o_proj, L["o"]),
                           (layer.mlp.gate_up_proj, L["gate_up"]), (layer.mlp.down_proj, L["down"])):
                for name, value in t.items():
                    param = getattr(mod, name)
                    if param is None:
                        raise RuntimeError(f"checkpoint tensor '{name}' has no parameter in {type(mod.quant_method).__name__}")
                    param.data.copy_(value.to(dev).view(param.shape) if value.numel() == param.numel() else value.to(dev))
                mod.quant_method.process_weights_after_loading(mod)
                if hasattr(layer, "_fused_w"):
                    del layer._fused_w

    def _fused_decode_ok(self, forward_batch) -> bool:
        return (self.fused_decode and forward_batch.forward_mode.is_decode()
                and self.quant_config is not None and self.quant_config.get_name() == "w8a8_fp8"
                and self.cfg.hidden_size <= 8192)

    def _fused_weights(self, layer):
        """Row-interleaved copies of the qkv / gate_up weights for the fused GEMM epilogues (built once; K <= 4096,
        head_dim 128).  They replace the plain copies on the decode path; memory cost = one extra copy of those two."""
        if not self.fused_epilogues:
            return None
        cached = getattr(layer, "_fused_w", None)
        if cached is not None:
            return cached or None
        attn, mlp = layer.self_attn, layer.mlp
        kdim = attn.qkv_proj.weight.shape[0]
        if attn.head_dim != 128 or kdim > 4096 or kdim % 64 != 0:
            layer._fused_w = {}
            return None
        hq, hkv = attn.num_heads, attn.num_kv_heads
        qw = attn.qkv_proj.weight.t()        # [N, K] fp8 (the parameter is the [K, N] view)
        gw = mlp.gate_up_proj.weight.t()
        as_u8 = lambda w: w.contiguous().view(torch.uint8)
        tq, tg = K.balanced_tile_rows(qw.shape[0]), K.balanced_tile_rows(gw.shape[0])   # 16- or 8-row tiles (workgroup balance)
        layer._fused_w = dict(
            qkv_w=K.interleave_rope_rows(as_u8(qw), hq, hkv, 128, tq).view(torch.float8_e4m3fn),
            qkv_s=K.interleave_rope_rows(attn.qkv_proj.weight_scale.view(-1), hq, hkv, 128, tq),
            qkv_b=None if attn.qkv_proj.bias is None else K.interleave_rope_rows(attn.qkv_proj.bias.data, hq, hkv, 128, tq),
            gu_w=K.interleave_gate_up_rows(as_u8(gw), tg).view(torch.float8_e4m3fn),
            gu_s=K.interleave_gate_up_rows(mlp.gate_up_proj.weight_scale.view(-1), tg),
            qkv_tile=tq, gu_tile=tg,
        )
        if self.fused_mlp_block and tg == 16 and self.tp_size == 1:
            # the persistent MLP launch reads gate_up and down_proj through ONE buffer descriptor: both in one allocation
            # (the interleaved gate_up copy moves there; down_proj gets a second copy)
            g, d = K.fp8_mlp_block_pack_weights(layer._fused_w["gu_w"], mlp.down_proj.weight.t())
            layer._fused_w["gu_w"], layer._fused_w["mlp_down"] = g, d
        return layer._fused_w

    def forward_decode_fused(self, input_ids, positions, forward_batch: ForwardBatch):
        """The same decode step with the elementwise work fused into 4 kernels per layer (sgl_kernel/fused.py) and the
        down_proj split-K combine folded into the next layer's RMSNorm.  Every fused kernel is bit-identical to the op
        sequence it replaces, so this path and forward() give the same logits.  Under tensor parallelism the two
        row-parallel outputs are materialised and all-reduced at the reference's call sites (linear.py:1302-1303); the
        split-K slab hand-off is a single-rank shortcut."""
        from ..distributed.communication_op import (fused_all_reduce_takes_slabs, tensor_model_parallel_all_gather,
                                                    tensor_model_parallel_all_reduce_add_rmsnorm_quant as ar_norm_quant)

        tp = self.tp_size
        pool = forward_batch.token_to_kv_pool
        backend = forward_batch.attn_backend
        m = input_ids.numel()
        hidden = K.embedding(input_ids, self.embed_tokens)
        residual, slabs, slab_sx, slab_sw = None, None, None, None
        mlp_scratch = None
        if (self.fused_mlp_block and tp == 1 and m <= 32 and self.fused_epilogues
                and K.fp8_mlp_block_supported(m, self.cfg.hidden_size, self.cfg.intermediate_size)):
            mlp_scratch = self._mlp_scratch.get(m)
            if mlp_scratch is None:
                mlp_scratch = self._mlp_scratch[m] = K.Fp8MlpBlockScratch(m, self.cfg.hidden_size, self.cfg.intermediate_size,
                                                                         len(self.layers), hidden.device)
            mlp_scratch.reset()   # the sync blocks of all layers, once per step
        for li, layer in enumerate(self.layers):
            attn, mlp = layer.self_attn, layer.mlp
            ln1 = layer.input_layernorm
            if residual is None:
                _, xq, xs = K.fused_add_rmsnorm_quant_fp8(hidden, None, ln1.weight.data, ln1.variance_epsilon)
                residual = hidden
            elif slabs is None and tp > 1:
                # `hidden` is this rank's PARTIAL down_proj output: its all-reduce (linear.py:1302-1303) rides in the same
                # launch as the add + RMSNorm + quant that consumes it (one-shot P2P communicator; else the unfused pair)
                _, xq, xs = ar_norm_quant(hidden, residual, ln1.weight.data, ln1.variance_epsilon)
            elif tp > 1:   # ... handed over as the GEMM's split-K partial sums: the fused all-reduce forms the operand itself (r4)
                _, xq, xs = ar_norm_quant(None, residual, ln1.weight.data, ln1.variance_epsilon, slabs=slabs, slab_sx=slab_sx,
                                          slab_sw=slab_sw, dtype=self.dtype)
            elif slabs is None:
                _, xq, xs = K.fused_add_rmsnorm_quant_fp8(hidden, residual, ln1.weight.data, ln1.variance_epsilon)
            else:
                _, xq, xs = K.fused_add_rmsnorm_quant_fp8(None, residual, ln1.weight.data, ln1.variance_epsilon, slabs=slabs,
                                                          slab_sx=slab_sx, slab_sw=slab_sw, dtype=self.dtype)
            lid = attn.attn.layer_id
            fw = self._fused_weights(layer) if m <= 64 else None   # the GEMM epilogue fusions are weight-streaming (M <= 64) kernels
            ks, vs = attn.kv_scales()
            if fw is not None:   # qkv GEMM with the RoPE + KV-write epilogue (16-bit or fp8 pool)
                q = K.fp8_qkv_rope_set_kv(xq, xs.view(-1), fw["qkv_w"], fw["qkv_s"], fw["qkv_b"], positions,
                                          attn.rotary_emb.cos_sin_cache, forward_batch.out_cache_loc, pool.get_key_buffer(lid),
                                          pool.get_value_buffer(lid), attn.num_heads, attn.num_kv_heads, attn.head_dim, self.dtype,
                                          fw["qkv_tile"], ks, vs)
            else:
                qkv = K.fp8_scaled_mm(xq, attn.qkv_proj.weight, xs.view(-1), attn.qkv_proj.weight_scale.view(-1), self.dtype,
                                      attn.qkv_proj.bias)
                q, k, v = qkv.split([attn.q_size, attn.kv_size, attn.kv_size], dim=-1)
                attn.rope_and_write_kv(positions, q, k, v, forward_batch)
            if self.fused_attn_merge and attn.head_dim in (64, 128):
                _, oq, osc = backend.forward_decode_merged_quant(q, attn.attn, forward_batch)   # stage 2 inside the launch
            else:
                md = backend.forward_decode_partial(q, attn.attn, forward_batch)
                _, oq, osc = K.decode_merge_quant_fp8(md.attn_logits, md.attn_lse, md.kv_indptr, md.num_kv_splits,
                                                      backend.max_kv_splits, self.dtype)
            ln2 = layer.post_attention_layernorm
            wo = attn.o_proj.weight   # [K, N] column-major view of the [N, K] parameter
            if tp == 1 and m <= 64 and self.o_proj_kranges > 1 and not (mlp_scratch is not None and fw is not None and "mlp_down" in fw):
                # o_proj as split-K partial sums straight into the add + RMSNorm + quant (no bit-identity with the per-op path:
                # the accumulation order follows the K partition; the parity contract is a tolerance, tests/test_model_gpu.py)
                o_slabs = K.fp8_linear_slabs(oq, wo.t(), m, wo.shape[1], wo.shape[0], min_kranges=self.o_proj_kranges)
                _, hq2, hs2 = K.fused_add_rmsnorm_quant_fp8(None, residual, ln2.weight.data, ln2.variance_epsilon, slabs=o_slabs,
                                                            slab_sx=osc.view(-1), slab_sw=attn.o_proj.weight_scale.view(-1),
                                                            dtype=self.dtype)
            elif tp == 1 and K.fp8_gemm_num_slabs(m, wo.shape[1], wo.shape[0], wo.device) > 1:
                # 64 < M <= 256: o_proj's split-K partial sums go straight into the add + RMSNorm + quant (no reduce launch)
                o_slabs = K.fp8_gemm_slabs(oq, wo.t())
                _, hq2, hs2 = K.fused_add_rmsnorm_quant_fp8(None, residual, ln2.weight.data, ln2.variance_epsilon, slabs=o_slabs,
                                                            slab_sx=osc.view(-1), slab_sw=attn.o_proj.weight_scale.view(-1),
                                                            dtype=self.dtype)
            elif mlp_scratch is not None and fw is not None and "mlp_down" in fw:
                # the whole MLP half in one persistent launch: add + RMSNorm + quant -> gate_up + SiluAndMul -> quant -> down_proj
                attn_out = K.fp8_scaled_mm(oq, wo, osc.view(-1), attn.o_proj.weight_scale.view(-1), self.dtype)
                slabs, slab_sx = K.fp8_mlp_block(attn_out, residual, ln2.weight.data, ln2.variance_epsilon, fw["gu_w"], fw["gu_s"],
                                                 fw["mlp_down"], mlp_scratch, li)
                slab_sw = mlp.down_proj.weight_scale.view(-1)
                continue
            elif (tp > 1 and K.fp8_gemm_num_slabs(m, wo.shape[1], wo.shape[0], wo.device) > 1
                  and fused_all_reduce_takes_slabs(m, wo.shape[1], self.dtype)):
                # tensor parallel, 64 < M <= 256: this rank's o_proj shard as split-K partial sums; the fused all-reduce forms its
                # operand from them while it publishes the row (no reduce launch, no [M, hidden] round trip; bit-identical)
                o_slabs = K.fp8_gemm_slabs(oq, wo.t())
                _, hq2, hs2 = ar_norm_quant(None, residual, ln2.weight.data, ln2.variance_epsilon, slabs=o_slabs, slab_sx=osc.view(-1),
                                            slab_sw=attn.o_proj.weight_scale.view(-1), dtype=self.dtype)
            else:
                attn_out = K.fp8_scaled_mm(oq, wo, osc.view(-1), attn.o_proj.weight_scale.view(-1), self.dtype)
                if tp > 1:   # partial sums: all-reduce + add + RMSNorm + quant in one launch
                    _, hq2, hs2 = ar_norm_quant(attn_out, residual, ln2.weight.data, ln2.variance_epsilon)
                else:
                    _, hq2, hs2 = K.fused_add_rmsnorm_quant_fp8(attn_out, residual, ln2.weight.data, ln2.variance_epsilon)
            if fw is not None:   # gate_up GEMM with the SiluAndMul epilogue, then the per-token quantisation
                act = K.fp8_gemm_silu_mul(hq2, hs2.view(-1), fw["gu_w"], fw["gu_s"], self.dtype, fw["gu_tile"])
                aq, asc = K.sglang_per_token_quant_fp8(act)
            else:
                gate_up = K.fp8_scaled_mm(hq2, mlp.gate_up_proj.weight, hs2.view(-1), mlp.gate_up_proj.weight_scale.view(-1), self.dtype)
                aq, asc = K.silu_and_mul_quant_fp8(gate_up)
            wd = mlp.down_proj.weight  # [K, N] column-major view of the [N, K] parameter
            if tp == 1 and m <= 64:
                slabs = K.fp8_linear_slabs(aq, wd.t(), m, wd.shape[1], wd.shape[0])
                slab_sx, slab_sw = asc.view(-1), mlp.down_proj.weight_scale.view(-1)
            elif ((tp == 1 or fused_all_reduce_takes_slabs(m, wd.shape[1], self.dtype))
                  and K.fp8_gemm_num_slabs(m, wd.shape[1], wd.shape[0], wd.device) > 1):
                slabs = K.fp8_gemm_slabs(aq, wd.t())   # 64 < M <= 256: the streaming tile's partial sums, same hand-off
                slab_sx, slab_sw = asc.view(-1), mlp.down_proj.weight_scale.view(-1)
            else:
                slabs = None
                hidden = K.fp8_scaled_mm(aq, wd, asc.view(-1), mlp.down_proj.weight_scale.view(-1), self.dtype)   # (tp > 1: partial)
        if slabs is not None and tp > 1:
            hidden, _, _ = ar_norm_quant(None, residual, self.norm.weight.data, self.norm.variance_epsilon, want_norm=True,
                                         want_quant=False, slabs=slabs, slab_sx=slab_sx, slab_sw=slab_sw, dtype=self.dtype)
        elif slabs is not None:
            hidden, _, _ = K.fused_add_rmsnorm_quant_fp8(None, residual, self.norm.weight.data, self.norm.variance_epsilon,
                                                         slabs=slabs, slab_sx=slab_sx, slab_sw=slab_sw, want_norm=True,
                                                         want_quant=False, dtype=self.dtype)
        elif tp > 1:
            hidden, _, _ = ar_norm_quant(hidden, residual, self.norm.weight.data, self.norm.variance_epsilon, want_norm=True,
                                         want_quant=False)
        else:
            hidden, _, _ = K.fused_add_rmsnorm_quant_fp8(hidden, residual, self.norm.weight.data, self.norm.variance_epsilon,
                                                         want_norm=True, want_quant=False)
        logits = K.dense_linear(hidden, self.lm_head, None, out_dtype=self.dtype)
        return tensor_model_parallel_all_gather(logits) if tp > 1 else logits

    def _fused_dense_ok(self, forward_batch) -> bool:
        """Unquantised (bf16 / f16) weights, or int4 AWQ weights at M <= 64 (the fused dequant GEMM's range)."""
        if not (self.fused_decode and forward_batch.forward_mode.is_decode() and self.tp_size == 1
                and self.cfg.hidden_size <= 8192):
            return False
        if self.quant_config is None:
            return True
        if self.quant_config.get_name() != "awq" or forward_batch.batch_size > 64:
            return False
        lin = self.layers[0]
        return all(getattr(mod, "_awq_packed", None) is not None and mod.scales.dtype == self.dtype
                   for mod in (lin.self_attn.qkv_proj, lin.self_attn.o_proj, lin.mlp.gate_up_proj, lin.mlp.down_proj))

    def _fused_weights_dense(self, layer, m: int):
        """Row-interleaved copies of the unquantised qkv / gate_up weights for the GEMM epilogue fusions; None when the K
        dimension does not fit one k-range of the weight-streaming kernel at this M (4096 bytes; 8192 at M <= 32)."""
        if not self.fused_epilogues:
            return None
        attn, mlp = layer.self_attn, layer.mlp
        kbytes = attn.qkv_proj.weight.shape[1] * attn.qkv_proj.weight.element_size()
        if attn.head_dim != 128 or kbytes % 64 != 0 or not (kbytes <= 4096 or (kbytes <= 8192 and m <= 32)) or m > 64:
            return None
        cached = getattr(layer, "_fused_w", None)
        if cached is None:
            hq, hkv = attn.num_heads, attn.num_kv_heads
            qw, gw = attn.qkv_proj.weight.data, mlp.gate_up_proj.weight.data   # [N, K]
            tq, tg = K.balanced_tile_rows(qw.shape[0], qw.element_size()), K.balanced_tile_rows(gw.shape[0], gw.element_size())
            cached = layer._fused_w = dict(
                qkv_w=K.interleave_rope_rows(qw, hq, hkv, 128, tq),
                qkv_b=None if attn.qkv_proj.bias is None else K.interleave_rope_rows(attn.qkv_proj.bias.data, hq, hkv, 128, tq),
                gu_w=K.interleave_gate_up_rows(gw, tg), qkv_tile=tq, gu_tile=tg)
        return cached

    def _fused_weights_awq(self, layer, m: int):
        """Column-interleaved re-packs of the int4 qkv / gate_up weights for the fused dequant GEMM's epilogues (built once; one
        extra int4 copy of those two); None when M > 32, K exceeds one k-range (4096) or head_dim is not 128."""
        if not self.fused_epilogues or m > 32:
            return None
        attn, mlp = layer.self_attn, layer.mlp
        if attn.head_dim != 128 or attn.qkv_proj.qweight.shape[0] > 4096:
            return None
        cached = getattr(layer, "_fused_w", None)
        if cached is None:
            hq, hkv = attn.num_heads, attn.num_kv_heads
            dev = attn.qkv_proj.qweight.device

            def repack(mod, order, bias):
                qw, qz, sc, b = K.awq_permute_cols(order, mod.qweight.data, mod.qzeros.data, mod.scales.data, bias)
                return K.awq_repack(qw, sc, qz) + (b,)

            q = repack(attn.qkv_proj, K.awq_rope_col_order(hq, hkv, dev), None if attn.qkv_proj.bias is None else attn.qkv_proj.bias.data)
            g = repack(mlp.gate_up_proj, K.awq_gate_up_col_order(mlp.gate_up_proj.scales.shape[1], dev), None)
            cached = layer._fused_w = dict(qkv_qp=q[0], qkv_sz=q[1], qkv_b=q[2], gu_qp=g[0], gu_sz=g[1])
        return cached

    def forward_decode_fused_dense(self, input_ids, positions, forward_batch: ForwardBatch):
        """The decode step of the 16-bit-activation linears -- unquantised bf16 / f16 weights (UnquantizedLinearMethod) or int4
        AWQ weights (AWQLinearMethod at M <= 64; the GEMM epilogue fusions at M <= 32) -- on fused kernels: 6 launches per layer instead of 10: add + RMSNorm (consuming
        the previous down_proj's split-K partial sums), qkv GEMM with the RoPE + KV-write epilogue, attention with the in-launch
        merge, o_proj, add + RMSNorm, gate_up GEMM with the SiluAndMul epilogue, down_proj (raw split-K slabs).  Every fused
        kernel is bit-identical to the op sequence it replaces, so this path and forward() give the same logits."""
        pool = forward_batch.token_to_kv_pool
        m = input_ids.numel()
        awq = self.quant_config is not None
        gsz = self.quant_config.group_size if awq else 0
        hidden = K.embedding(input_ids, self.embed_tokens)
        residual, slabs = None, None
        norm = lambda x, res, ln, sl=None: K.fused_add_rmsnorm_quant_fp8(
            x, res, ln.weight.data, ln.variance_epsilon, slabs=sl, want_norm=True, want_quant=False, dtype=self.dtype)[0]

        def linear(mod, x):
            if awq:
                return K.awq_gemm(x, mod._awq_packed[0], mod._awq_packed[1], gsz, mod.bias)
            return K.dense_linear(x, mod.weight.data, mod.bias)

        for layer in self.layers:
            attn, mlp = layer.self_attn, layer.mlp
            if residual is None:
                x = norm(hidden, None, layer.input_layernorm)
                residual = hidden
            else:
                x = norm(hidden if slabs is None else None, residual, layer.input_layernorm, slabs)
            lid = attn.attn.layer_id
            fw = self._fused_weights_awq(layer, m) if awq else self._fused_weights_dense(layer, m)
            rope_args = (positions, attn.rotary_emb.cos_sin_cache, forward_batch.out_cache_loc, pool.get_key_buffer(lid),
                         pool.get_value_buffer(lid), attn.num_heads, attn.num_kv_heads, attn.head_dim)
            ks, vs = attn.kv_scales()
            if fw is not None and awq:
                q = K.awq_qkv_rope_set_kv(x, fw["qkv_qp"], fw["qkv_sz"], fw["qkv_b"], gsz, *rope_args, ks, vs)
            elif fw is not None:
                q = K.qkv_rope_set_kv(x, fw["qkv_w"], fw["qkv_b"], *rope_args, fw["qkv_tile"], ks, vs)
            else:
                qkv = linear(attn.qkv_proj, x)
                q, k, v = qkv.split([attn.q_size, attn.kv_size, attn.kv_size], dim=-1)
                attn.rope_and_write_kv(positions, q, k, v, forward_batch)
            o = attn.attn(q, None, None, forward_batch, save_kv_cache=False)
            h2 = norm(linear(attn.o_proj, o), residual, layer.post_attention_layernorm)
            if fw is not None and awq:
                act = K.awq_gemm_silu_mul(h2, fw["gu_qp"], fw["gu_sz"], gsz)
            elif fw is not None:
                act = K.gemm_silu_mul(h2, fw["gu_w"], fw["gu_tile"])
            else:
                act = K.silu_and_mul(linear(mlp.gate_up_proj, h2))
            down = mlp.down_proj
            if down.bias is not None:   # the slab hand-off into the next RMSNorm carries no bias term
                slabs, hidden = None, linear(down, act)
            elif awq and K.awq_gemm_num_kranges(m, act.shape[1]) > 1:
                slabs = K.awq_gemm_slabs(act, down._awq_packed[0], down._awq_packed[1], gsz)
            elif not awq and m <= 64 and K.dense_linear_kranges(m, down.weight.shape[0], down.weight.shape[1], down.weight.dtype) > 1:
                slabs = K.fp8_linear_slabs(act, down.weight.data, m, down.weight.shape[0], down.weight.shape[1])
            else:
                slabs, hidden = None, linear(down, act)
        hidden = norm(hidden if slabs is None else None, residual, self.norm, slabs)
        return K.dense_linear(hidden, self.lm_head, None, out_dtype=self.dtype)

    def _fused_extend_ok(self, forward_batch) -> bool:
        return (self.fused_extend and forward_batch.forward_mode.is_extend()
                and self.quant_config is not None and self.quant_config.get_name() == "w8a8_fp8"
                and self.cfg.hidden_size <= 8192 and 2 * self.cfg.intermediate_size <= 65536)

    def forward_extend_fused(self, input_ids, positions, forward_batch: ForwardBatch, last_index):
        """Prefill with the same fused elementwise kernels as the decode step (add+RMSNorm+quant, RoPE+KV write,
        SiluAndMul+quant); the linears are the tiled fp8 GEMM.  Bit-identical to the unfused op sequence of forward()."""
        from ..distributed.communication_op import tensor_model_parallel_all_gather, tensor_model_parallel_all_reduce

        tp = self.tp_size
        pool = forward_batch.token_to_kv_pool
        backend = forward_batch.attn_backend
        hidden = K.embedding(input_ids, self.embed_tokens)
        residual = None
        qkv_buf = None   # one padded [T, N + 128] buffer, reused by every layer
        for layer in self.layers:
            attn, mlp = layer.self_attn, layer.mlp
            ln1, ln2 = layer.input_layernorm, layer.post_attention_layernorm
            if residual is None:
                _, xq, xs = K.fused_add_rmsnorm_quant_fp8(hidden, None, ln1.weight.data, ln1.variance_epsilon)
                residual = hidden
            else:
                _, xq, xs = K.fused_add_rmsnorm_quant_fp8(hidden, residual, ln1.weight.data, ln1.variance_epsilon)
            lid = attn.attn.layer_id
            # qkv rows padded by 256 bytes: extend attention reads the new tokens' K / V rows out of this tensor, and at the natural
            # 12 KiB row stride (3 x 4 KiB) those rows fall into few HBM channels (tools/debug/ext_stride.py: 1 543 -> 1 411 us per launch; the 32x32x16 kernel takes
            # row strides that are multiples of 128 elements)
            n_qkv = attn.q_size + 2 * attn.kv_size
            if self.pad_qkv_rows and (qkv_buf is None or qkv_buf.shape[0] != xq.shape[0] or qkv_buf.shape[1] != n_qkv + 128):
                qkv_buf = torch.empty((xq.shape[0], n_qkv + 128), dtype=self.dtype, device=xq.device)
            qkv = K.fp8_scaled_mm(xq, attn.qkv_proj.weight, xs.view(-1), attn.qkv_proj.weight_scale.view(-1), self.dtype,
                                  attn.qkv_proj.bias, out=qkv_buf[:, :n_qkv] if self.pad_qkv_rows else None)
            q, k, v = qkv.split([attn.q_size, attn.kv_size, attn.kv_size], dim=-1)
            attn.rope_and_write_kv(positions, q, k, v, forward_batch)
            o = backend.forward_extend(q, k.view(-1, attn.num_kv_heads, attn.head_dim), v.view(-1, attn.num_kv_heads, attn.head_dim),
                                       attn.attn, forward_batch, save_kv_cache=False)
            oq, osc = K.sglang_per_token_quant_fp8(o)
            hidden = K.fp8_scaled_mm(oq, attn.o_proj.weight, osc.view(-1), attn.o_proj.weight_scale.view(-1), self.dtype)
            if tp > 1:
                hidden = tensor_model_parallel_all_reduce(hidden)
            _, hq2, hs2 = K.fused_add_rmsnorm_quant_fp8(hidden, residual, ln2.weight.data, ln2.variance_epsilon)
            fw = self._fused_weights(layer) if self.prefill_silu_epilogue else None
            if (fw is not None and fw["gu_tile"] == 16 and self.dtype == torch.bfloat16 and fw["gu_w"].shape[0] % 256 == 0
                    and fw["gu_w"].shape[1] % 128 == 0 and hq2.shape[0] > 64
                    and K.fp8_gemm_num_slabs(hq2.shape[0], fw["gu_w"].shape[0], fw["gu_w"].shape[1], hq2.device) == 1):
                # (one k-range: fp8_scaled_mm then sums the K slices in the same order as the 256x256 tile, so the fused form stays
                # bit-identical to the per-op path whichever kernel that path picks)
                # gate_up with the SiluAndMul epilogue on the decode path's interleaved weight copy: the [T, 2 I] intermediate is never
                # written; the per-token quant reads the [T, I] activation once (bit-identical to the two-kernel sequence)
                act = K.fp8_gemm_silu_mul(hq2, hs2.view(-1), fw["gu_w"], fw["gu_s"], self.dtype, 16)
                aq, asc = K.sglang_per_token_quant_fp8(act)
            else:
                gate_up = K.fp8_scaled_mm(hq2, mlp.gate_up_proj.weight, hs2.view(-1), mlp.gate_up_proj.weight_scale.view(-1), self.dtype)
                aq, asc = K.silu_and_mul_quant_fp8(gate_up)
            hidden = K.fp8_scaled_mm(aq, mlp.down_proj.weight, asc.view(-1), mlp.down_proj.weight_scale.view(-1), self.dtype)
            if tp > 1:
                hidden = tensor_model_parallel_all_reduce(hidden)
        if last_index is not None:  # only the last token of each request reaches the final norm and lm_head
            hidden = hidden.index_select(0, last_index)
            residual = residual.index_select(0, last_index)
        hidden, _, _ = K.fused_add_rmsnorm_quant_fp8(hidden, residual, self.norm.weight.data, self.norm.variance_epsilon,
                                                     want_norm=True, want_quant=False)
        logits = K.dense_linear(hidden, self.lm_head, None, out_dtype=self.dtype)
        return tensor_model_parallel_all_gather(logits) if tp > 1 else logits

    def forward(self, input_ids, positions, forward_batch: ForwardBatch, last_index: Optional[torch.Tensor] = None):
        if self._fused_decode_ok(forward_batch):
            return self.forward_decode_fused(input_ids, positions, forward_batch)
        if self._fused_dense_ok(forward_batch):
            return self.forward_decode_fused_dense(input_ids, positions, forward_batch)
        if self._fused_extend_ok(forward_batch):
            return self.forward_extend_fused(input_ids, positions, forward_batch, last_index)
        hidden_states = K.embedding(input_ids, self.embed_tokens)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, forward_batch, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        if last_index is not None:  # extend: logits of the last token of each request only (logits_processor.py)
            hidden_states = hidden_states.index_select(0, last_index)
        logits = K.dense_linear(hidden_states, self.lm_head, None, out_dtype=self.dtype)
        if self.tp_size > 1:
            from ..distributed.communication_op import tensor_model_parallel_all_gather

            logits = tensor_model_parallel_all_gather(logits)
        return logits


class SyntheticModelRunner:
    """Owns the pools, allocator, backend and model; exposes the attributes an attention backend reads from a
    ModelRunner and the extend()/decode() drivers of bench_one_batch."""

    def __init__(self, cfg: LlamaShape, quantization: Optional[str], max_running_requests: int, context_len: int,
                 max_total_tokens: int, device: str = "cuda:0", dtype=torch.bfloat16, seed: int = 0,
                 kv_cache_dtype: Optional[torch.dtype] = None, init_weights: bool = True, max_kv_splits: int = 16,
                 kv_split_rule: int = 3, kv_sched_rounds_pct: int = 150):
        self.device = device
        self.gpu_id = torch.device(device).index or 0
        self.dtype = dtype
        self.cfg = cfg
        tp = get_tensor_model_parallel_world_size()
        self.attention_tp_size = tp
        self.tp_rank = get_tensor_model_parallel_rank()
        kv_heads = max(1, cfg.num_key_value_heads // tp)
        self.model_config = SimpleNamespace(
            num_attention_heads=cfg.num_attention_heads, context_len=context_len, is_encoder_decoder=False,
            get_num_kv_heads=lambda tp_size: max(1, cfg.num_key_value_heads // tp_size), hidden_size=cfg.hidden_size,
            vocab_size=cfg.vocab_size, num_hidden_layers=cfg.num_hidden_layers, head_dim=cfg.head_dim,
        )
        self.sliding_window_size = None
        self.server_args = SimpleNamespace(triton_attention_num_kv_splits=int(max_kv_splits), kv_split_rule=int(kv_split_rule),
                                           kv_sched_rounds_pct=int(kv_sched_rounds_pct),
                                           speculative_num_draft_tokens=None,
                                           speculative_num_steps=None, page_size=1)
        self.page_size = 1
        self.req_to_token_pool = ReqToTokenPool(max_running_requests, context_len, device, False)
        self.token_to_kv_pool = MHATokenToKVPool(max_total_tokens, 1, kv_cache_dtype or dtype, kv_heads, cfg.head_dim,
                                                 cfg.num_hidden_layers, device, False)
        self.token_to_kv_pool_allocator = TokenToKVPoolAllocator(max_total_tokens, kv_cache_dtype or dtype, device, self.token_to_kv_pool)
        with torch.device(device):
            self.model = LlamaForCausalLM(cfg, quantization, dtype)
        if init_weights:
            self.model.init_random_weights(seed)   # else: the caller loads a checkpoint (model.load_checkpoint)
        self.attn_backend = HipAttnBackend(self)
        self._graphs = {}

    # ---- bench_one_batch-style drivers -------------------------------------------------------------------
    fused_decode_prepare = True   # decode_graph: one HIP launch for prepare_for_decode + replay_prepare

    def clear(self):
        self.req_to_token_pool.clear()
        self.token_to_kv_pool_allocator.clear()

    @torch.no_grad()
    def extend(self, input_ids: List[torch.Tensor], prefix_indices: Optional[List[torch.Tensor]] = None):
        """Prefill a batch: allocates request rows and KV slots, writes req_to_token, runs the model.
        Returns (next-token logits [bs, vocab], batch state for decode)."""
        bs = len(input_ids)
        dev = self.device
        req_pool_indices = torch.tensor(self.req_to_token_pool.alloc(bs), dtype=torch.int64, device=dev)
        pre = [0 if prefix_indices is None else int(prefix_indices[i].numel()) for i in range(bs)]
        ext = [int(x.numel()) for x in input_ids]
        seq = [p + e for p, e in zip(pre, ext)]
        out_cache_loc = self.token_to_kv_pool_allocator.alloc(sum(ext))
        if out_cache_loc is None:
            raise RuntimeError("Prefill out of memory. Try to lower your batch size.")
        pre_t = torch.tensor(pre, dtype=torch.int64, device=dev)
        ext_t = torch.tensor(ext, dtype=torch.int64, device=dev)
        seq_t = torch.tensor(seq, dtype=torch.int64, device=dev)
        if prefix_indices is not None:
            for i in range(bs):
                if pre[i]:
                    self.req_to_token_pool.write((req_pool_indices[i], slice(0, pre[i])), prefix_indices[i].to(torch.int32))
        K.write_req_to_token(self.req_to_token_pool.req_to_token, req_pool_indices, pre_t, seq_t, ext_t, out_cache_loc)
        ids = torch.cat(input_ids).to(dev)
        fb = ForwardBatch.init_new(ForwardMode.EXTEND, req_pool_indices, seq_t, out_cache_loc, ids, self.req_to_token_pool,
                                   self.token_to_kv_pool, self.attn_backend, extend_prefix_lens=pre_t, extend_seq_lens=ext_t,
                                   seq_lens_cpu=torch.tensor(seq, dtype=torch.int64))
        self.attn_backend.init_forward_metadata(fb)
        last_index = torch.cumsum(ext_t, 0) - 1
        logits = self.model(ids, fb.positions, fb, last_index)
        state = SimpleNamespace(req_pool_indices=req_pool_indices, seq_lens=seq_t, seq_lens_cpu=list(seq))
        return logits, state

    def _prepare_decode(self, state, next_ids):
        """seq_lens += 1, one new slot per request, req_to_token[idx, seq_len - 1] = slot (schedule_batch.py:1560-1590)."""
        bs = len(state.seq_lens_cpu)
        out_cache_loc = self.token_to_kv_pool_allocator.alloc(bs)
        if out_cache_loc is None:
            raise RuntimeError("Decode out of memory. Try to lower your batch size.")
        locs = state.seq_lens.clone()
        state.seq_lens = state.seq_lens + 1
        state.seq_lens_cpu = [s + 1 for s in state.seq_lens_cpu]
        self.req_to_token_pool.write((state.req_pool_indices, locs), out_cache_loc.to(torch.int32))
        return ForwardBatch.init_new(ForwardMode.DECODE, state.req_pool_indices, state.seq_lens, out_cache_loc, next_ids,
                                     self.req_to_token_pool, self.token_to_kv_pool, self.attn_backend,
                                     seq_lens_cpu=torch.tensor(state.seq_lens_cpu, dtype=torch.int64))

    def check_errors(self) -> None:
        """Raises if a bounded spin of the persistent MLP launch (SGL_MI355_MLP_BLOCK=1, off by default) ran out during an earlier
        step: the launch then ran on with incomplete hand-off data and the logits of that step are invalid (a captured graph
        would replay the condition unnoticed).  Reads the per-layer sync words (a device -> host copy: synchronises).  decode() /
        decode_graph() call it for the PREVIOUS step, whose logits the caller has consumed by then; call it after the last step.
        The launch also needs every CU for itself: nothing else may run on the device during a step."""
        for sc in getattr(self.model, "_mlp_scratch", {}).values():
            codes = sc.error_codes()
            if int(codes.abs().sum()) != 0:
                raise RuntimeError(f"fp8_mlp_block: a hand-off timed out in an earlier step (sync words {codes.tolist()}); "
                                   "that step's logits are invalid -- unset SGL_MI355_MLP_BLOCK (the four-launch path)")

    @torch.no_grad()
    def decode(self, state, next_ids: torch.Tensor, shared_prefix_len: int = 0):
        """shared_prefix_len > 0: every request of the batch shares its first slots (one radix node): cascade decode attention."""
        if self.model.fused_mlp_block:
            self.check_errors()
        fb = self._prepare_decode(state, next_ids)
        if shared_prefix_len > 0:
            self.attn_backend.init_forward_metadata_cascade(fb, shared_prefix_len)
        else:
            self.attn_backend.init_forward_metadata(fb)
        return self.model(next_ids, fb.positions, fb)

    # ---- HIP-graph decode (cuda_graph_runner.py:280,618,760 hooks) ------------------------------------------
    @torch.no_grad()
    def capture_decode_graph(self, bs: int, shared_prefix_len: int = 0):
        """shared_prefix_len > 0 captures the cascade (shared-prefix) decode step: the length is baked into the launches, so the
        graph serves exactly the batches whose requests share that many leading slots."""
        dev = self.device
        if self.attn_backend._graph is None:
            self.attn_backend.init_cuda_graph_state(bs, bs)
        self.attn_backend.cascade_shared_prefix_len = int(shared_prefix_len)
        buf = SimpleNamespace(
            input_ids=torch.zeros(bs, dtype=torch.int64, device=dev),
            req_pool_indices=torch.zeros(bs, dtype=torch.int64, device=dev),
            seq_lens=torch.full((bs,), 1 + int(shared_prefix_len), dtype=torch.int64, device=dev),
            out_cache_loc=torch.zeros(bs, dtype=torch.int64, device=dev),
            positions=torch.zeros(bs, dtype=torch.int64, device=dev),
        )
        fb = ForwardBatch(forward_mode=ForwardMode.DECODE, batch_size=bs, input_ids=buf.input_ids,
                          req_pool_indices=buf.req_pool_indices, seq_lens=buf.seq_lens, out_cache_loc=buf.out_cache_loc,
                          seq_lens_sum=bs, positions=buf.positions, req_to_token_pool=self.req_to_token_pool,
                          token_to_kv_pool=self.token_to_kv_pool, attn_backend=self.attn_backend)
        self.attn_backend.init_forward_metadata_capture_cuda_graph(bs, bs, buf.req_pool_indices, buf.seq_lens, None,
                                                                   ForwardMode.DECODE, None)
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            for _ in range(2):  # warm-up outside capture (first-launch attribute calls, allocator pools)
                self.model(buf.input_ids, buf.positions, fb)
        torch.cuda.current_stream().wait_stream(stream)
        graph = torch.cuda.CUDAGraph()
        # The step's metadata launches (merge-counter reset, kv_indptr + split counts, kv_indices) read and write only the graph's
        # static buffers, so they are captured WITH the step (round 4): three dependent launches inside the graph instead of three
        # host launches between two replays.  Not for the cascade form (its metadata bakes in host-side lengths) or window layers.
        buf.meta_in_graph = bool(getattr(self, "graph_metadata", True)) and int(shared_prefix_len) == 0 and not self.attn_backend._has_window()
        # thread_local: the RCCL watchdog thread of torch.distributed may touch the device while this thread captures (TP > 1)
        with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
            if buf.meta_in_graph:
                self.attn_backend.init_forward_metadata_replay_cuda_graph(bs, buf.req_pool_indices, buf.seq_lens, bs, None,
                                                                          ForwardMode.DECODE, None, None)
            buf.logits = self.model(buf.input_ids, buf.positions, fb)
        self._graphs[(bs, int(shared_prefix_len))] = (graph, buf)
        self.attn_backend.cascade_shared_prefix_len = 0

    @torch.no_grad()
    def decode_graph(self, state, next_ids: torch.Tensor, shared_prefix_len: int = 0):
        bs = len(state.seq_lens_cpu)
        graph, buf = self._graphs[(bs, int(shared_prefix_len))]
        if self.model.fused_mlp_block:
            self.check_errors()
        self.attn_backend.cascade_shared_prefix_len = int(shared_prefix_len)
        if not self.fused_decode_prepare:
            fb = self._prepare_decode(state, next_ids)
            buf.input_ids.copy_(next_ids)
            buf.req_pool_indices.copy_(fb.req_pool_indices)
            buf.seq_lens.copy_(fb.seq_lens)
            buf.out_cache_loc.copy_(fb.out_cache_loc)
            buf.positions.copy_(fb.positions)
            seq_sum, seq_cpu = fb.seq_lens_sum, fb.seq_lens_cpu
        else:
            # the same index work (slot allocation aside) and the copies into the graph's static buffers in ONE launch: a dozen
            # 5-17 us index / copy kernels per step otherwise sit between two graph replays
            out_cache_loc = self.token_to_kv_pool_allocator.alloc(bs)
            if out_cache_loc is None:
                raise RuntimeError("Decode out of memory. Try to lower your batch size.")
            K.decode_prepare(state.req_pool_indices, state.seq_lens, out_cache_loc, next_ids, self.req_to_token_pool.req_to_token,
                             buf.input_ids, buf.req_pool_indices, buf.seq_lens, buf.out_cache_loc, buf.positions)
            state.seq_lens_cpu = [s + 1 for s in state.seq_lens_cpu]
            seq_sum, seq_cpu = sum(state.seq_lens_cpu), None
        if not buf.meta_in_graph:
            self.attn_backend.init_forward_metadata_replay_cuda_graph(bs, buf.req_pool_indices, buf.seq_lens, seq_sum,
                                                                      None, ForwardMode.DECODE, None, seq_cpu)
        graph.replay()
        self.attn_backend.cascade_shared_prefix_len = 0
        return buf.logits
