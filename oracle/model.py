"""End-to-end oracle: a CPU restatement of the Llama decoder forward with the reference's torch-native building blocks.

  layer structure   python/sglang/srt/models/llama.py:94-98 (MLP), :180-191 (attention), :245-268 (decoder layer),
                    :308-340 (model): fused-add RMSNorm residual stream, qkv -> rope -> attention -> o_proj, gate_up ->
                    SiluAndMul -> down
  attention         torch_native_backend.py:27-180 through oracle/attention.py (KV written to the pool first)
  norm / rope / act oracle/elementwise.py (forward_native restatements)
  w8a8 fp8 linear   W8A8Fp8LinearMethod.apply (w8a8_fp8.py:177-190): per-token dynamic activation quant +
                    per-channel weights, product as the reference's torch_scaled_mm (oracle/quant.py)
  logits            last token of each request, lm_head in the model dtype (logits_processor.py)

TEST INFRASTRUCTURE: see oracle/__init__.py.  Weights are handed in as CPU tensors by the test.
"""
import torch
import torch.nn.functional as F

from . import attention as oa
from . import elementwise as oe
from . import quant as oq


class OracleLlama:
    def __init__(self, cfg, weights, dtype, quantized: bool, pool_slots: int):
        """weights: dict with embed, lm_head, norm and per layer ln1, ln2, qkv (w, s, b), o (w, s), gate_up (w, s),
        down (w, s); for quantized models w is e4m3fn [N, K] and s f32 [N]; else w is dtype [N, K] and s None."""
        self.cfg, self.w, self.dtype, self.quantized = cfg, weights, dtype, quantized
        hkv, d = cfg.num_key_value_heads, cfg.head_dim
        self.k_buf = [torch.zeros(pool_slots, hkv, d, dtype=dtype) for _ in range(cfg.num_hidden_layers)]
        self.v_buf = [torch.zeros(pool_slots, hkv, d, dtype=dtype) for _ in range(cfg.num_hidden_layers)]
        self.cos_sin = oe.rope_cache(d, d, cfg.max_position_embeddings, cfg.rope_theta)

    def _linear(self, x, lw, bias=None):
        w, s = lw
        if not self.quantized:
            return F.linear(x, w, bias)
        xq, sx = oq.per_token_quant_fp8(x)
        return oq.scaled_mm(xq, w.t(), sx.flatten(), s, self.dtype, bias)

    def forward(self, input_ids, positions, req_to_token, req_pool_indices, seq_lens, out_cache_loc, extend_prefix_lens=None,
                extend_seq_lens=None):
        """decode when extend_seq_lens is None; returns next-token logits [bs, vocab]."""
        cfg, W = self.cfg, self.w
        hq, hkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        h = W["embed"][input_ids]
        residual = None
        for l in range(cfg.num_hidden_layers):
            L = W["layers"][l]
            if residual is None:
                residual, h = h, oe.rmsnorm(h, L["ln1"], cfg.rms_norm_eps)
            else:
                h, residual = oe.rmsnorm(h, L["ln1"], cfg.rms_norm_eps, residual)
            qkv = self._linear(h, L["qkv"][:2], L["qkv"][2])
            q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
            q, k = oe.rope(positions, q.contiguous(), k.contiguous(), d, self.cos_sin, True)
            self.k_buf[l][out_cache_loc] = k.view(-1, hkv, d)
            self.v_buf[l][out_cache_loc] = v.reshape(-1, hkv, d)
            qh = q.view(-1, hq, d)
            if extend_seq_lens is None:
                o = oa.decode_attention_sdpa(qh, self.k_buf[l], self.v_buf[l], req_to_token, req_pool_indices, seq_lens, d ** -0.5)
            else:
                o = oa.extend_attention_sdpa(qh, self.k_buf[l], self.v_buf[l], req_to_token, req_pool_indices, seq_lens,
                                             extend_prefix_lens, extend_seq_lens, d ** -0.5)
            a = self._linear(o.reshape(-1, hq * d), L["o"])
            h, residual = oe.rmsnorm(a, L["ln2"], cfg.rms_norm_eps, residual)
            gu = self._linear(h, L["gate_up"])
            h = self._linear(oe.silu_and_mul(gu), L["down"])
        h, _ = oe.rmsnorm(h, W["norm"], cfg.rms_norm_eps, residual)
        if extend_seq_lens is not None:
            h = h[torch.cumsum(extend_seq_lens.long(), 0) - 1]
        return F.linear(h, W["lm_head"])
