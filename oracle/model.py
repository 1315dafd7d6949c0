"""End-to-end oracle: a CPU restatement of the Llama / Qwen2 decoder forward with the reference's torch-native building blocks.

  layer structure   python/sglang/srt/models/llama.py:94-98 (MLP), :180-191 (attention), :245-268 (decoder layer),
                    :308-340 (model): fused-add RMSNorm residual stream, qkv -> rope -> attention -> o_proj, gate_up ->
                    SiluAndMul -> down; Qwen2 adds the qkv bias (models/qwen2.py:118-125)
  attention         torch_native_backend.py:27-180 through oracle/attention.py (KV written to the pool first)
  norm / rope / act oracle/elementwise.py (forward_native restatements)
  w8a8 fp8 linear   W8A8Fp8LinearMethod.apply (w8a8_fp8.py:177-190): per-token dynamic activation quant +
                    per-channel weights, product as the reference's torch_scaled_mm (oracle/quant.py)
  fp8 linear        Fp8LinearMethod (fp8.py:336-501): per-shard weight scales requantised to the max scale
                    (quantization/utils.py:95-120), static (input_scale.max()) or dynamic per-tensor activation quant
  awq linear        AWQLinearMethod.apply (awq.py:401-418): awq_dequantize -> matmul in the activation dtype (+ bias)
  logits            last token of each request, lm_head in the model dtype (logits_processor.py)

Two arithmetic modes: ``dtype=torch.bfloat16`` reproduces the reference's rounding points (pinned bit for bit by the G7
fixture, tests/golden/model.npz); ``exact=True`` evaluates the SAME function in float64 with no intermediate rounding --
the stored parameters (bf16 / fp8 x scale / int4) and the fp8 activation quantisation steps (they define the quantised
model) are kept, every bf16 rounding of an intermediate is dropped.  The exact logits are what both the reference and the
HIP stack approximate; the GPU test compares their distances to it.

TEST INFRASTRUCTURE: see oracle/__init__.py.  Weights are handed in as CPU tensors by the test.
"""
import torch
import torch.nn.functional as F

from . import attention as oa
from . import elementwise as oe
from . import quant as oq


def process_checkpoint(ckpt, quant, qkv_widths=None):
    """Checkpoint tensors (tests/_cases.py::build_model_case) -> the per-linear tuples OracleLlama consumes; the oracle's
    restatement of each linear method's process_weights_after_loading."""
    def lin(t, widths=None):
        if quant is None:
            return ("dense", t["weight"], t.get("bias"))
        if quant == "w8a8_fp8":
            return ("w8a8", t["weight"], t["weight_scale"].flatten(), t.get("bias"))
        if quant == "fp8":
            w, ws = t["weight"], t["weight_scale"]
            smax = ws.max()
            if ws.numel() > 1:   # requantize_with_max_scale: dequantise each shard with its scale, quantise with the max
                parts, start = [], 0
                for i, width in enumerate(widths):
                    dq = w[start:start + width].to(torch.float16) * ws[i]   # per_tensor_dequantize goes through f16 (utils.py:59-64)
                    parts.append(oq.per_tensor_quant_fp8(dq, smax.reshape(1))[0])
                    start += width
                w = torch.cat(parts)
            ins = t.get("input_scale")
            return ("fp8", w, smax.reshape(1), None if ins is None else ins.max().reshape(1), t.get("bias"))
        if quant == "awq":
            return ("awq", t["qweight"], t["scales"], t["qzeros"], t.get("bias"))
        raise ValueError(quant)

    out = dict(embed=ckpt["embed"], lm_head=ckpt["lm_head"], norm=ckpt["norm"], layers=[])
    for L in ckpt["layers"]:
        n_qkv = L["qkv"]["weight"].shape[0] if "weight" in L["qkv"] else L["qkv"]["scales"].shape[1]
        n_gu = L["gate_up"]["weight"].shape[0] if "weight" in L["gate_up"] else L["gate_up"]["scales"].shape[1]
        out["layers"].append(dict(ln1=L["ln1"], ln2=L["ln2"], qkv=lin(L["qkv"], qkv_widths), o=lin(L["o"]),
                                  gate_up=lin(L["gate_up"], [n_gu // 2, n_gu // 2]), down=lin(L["down"]), n_qkv=n_qkv))
    return out


def _fp8_round(x64):
    """float64 -> nearest e4m3fn value (through float32: the double rounding can only matter at exact ties of a 3-bit
    mantissa, which a float64 product does not hit in practice), returned as float64."""
    return x64.clamp(-oq.FP8_MAX, oq.FP8_MAX).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64)


class OracleLlama:
    def __init__(self, cfg, weights, dtype, quantized=None, pool_slots: int = 0, exact: bool = False):
        """weights: either the legacy dict (per layer qkv = (w, s, b), o / gate_up / down = (w, s); quantized = bool) or
        the output of process_checkpoint()."""
        self.cfg, self.w, self.exact = cfg, weights, exact
        self.dtype = torch.float64 if exact else dtype
        self.quantized = quantized
        hkv, d = cfg.num_key_value_heads, cfg.head_dim
        self.k_buf = [torch.zeros(pool_slots, hkv, d, dtype=self.dtype) for _ in range(cfg.num_hidden_layers)]
        self.v_buf = [torch.zeros(pool_slots, hkv, d, dtype=self.dtype) for _ in range(cfg.num_hidden_layers)]
        if exact:
            inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
            fr = torch.einsum("i,j -> ij", torch.arange(cfg.max_position_embeddings, dtype=torch.float64), inv)
            self.cos_sin = torch.cat((fr.cos(), fr.sin()), dim=-1)
        else:
            self.cos_sin = oe.rope_cache(d, d, cfg.max_position_embeddings, cfg.rope_theta)

    # ---- linears ---------------------------------------------------------------------------------------------
    def _linear(self, x, lw, bias=None):
        if not isinstance(lw[0], str):   # legacy tuples
            w, s = lw[0], lw[1]
            lw = ("w8a8", w, s, bias) if self.quantized else ("dense", w, bias)
        kind = lw[0]
        return getattr(self, "_lin_" + kind)(x, *lw[1:])

    def _lin_dense(self, x, w, bias):
        if self.exact:
            y = x @ w.to(torch.float64).t()
            return y if bias is None else y + bias.to(torch.float64)
        return F.linear(x, w, bias)

    def _lin_w8a8(self, x, w, s, bias):
        if self.exact:
            sx = x.abs().amax(dim=1, keepdim=True) / oq.FP8_MAX
            xq = _fp8_round(torch.where(sx == 0, torch.zeros_like(x), x / sx))
            y = (xq @ w.to(torch.float64).t()) * sx * s.to(torch.float64).view(1, -1)
            return y if bias is None else y + bias.to(torch.float64)
        xq, sx = oq.per_token_quant_fp8(x)
        return oq.scaled_mm(xq, w.t(), sx.flatten(), s, self.dtype, bias)

    def _lin_fp8(self, x, w, ws, in_scale, bias):
        if self.exact:
            sx = (x.abs().amax() / oq.FP8_MAX) if in_scale is None else in_scale.to(torch.float64)
            xq = _fp8_round(x / sx)
            y = (xq @ w.to(torch.float64).t()) * sx * ws.to(torch.float64)
            return y if bias is None else y + bias.to(torch.float64)
        xq, sx = oq.per_tensor_quant_fp8(x, in_scale)
        m, n = x.shape[0], w.shape[0]
        return oq.scaled_mm(xq, w.t(), sx.expand(m), ws.expand(n), self.dtype, bias)

    def _lin_awq(self, x, qweight, scales, qzeros, bias):
        if self.exact:
            k = qweight.shape[0]
            g = k // scales.shape[0]
            wq = oq._awq_unpack(qweight).to(torch.float64)
            z = oq._awq_unpack(qzeros).to(torch.float64).repeat_interleave(g, dim=0)
            y = x @ ((wq - z) * scales.to(torch.float64).repeat_interleave(g, dim=0))
            return y if bias is None else y + bias.to(torch.float64)
        out = torch.matmul(x, oq.awq_dequantize(qweight, scales, qzeros).to(x.dtype))
        return out if bias is None else out + bias

    # ---- elementwise -----------------------------------------------------------------------------------------
    def _norm(self, x, w, residual=None):
        if not self.exact:
            return oe.rmsnorm(x, w, self.cfg.rms_norm_eps, residual)
        if residual is not None:
            x = x + residual
            residual = x
        y = x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + self.cfg.rms_norm_eps) * w.to(torch.float64)
        return y if residual is None else (y, residual)

    def forward(self, input_ids, positions, req_to_token, req_pool_indices, seq_lens, out_cache_loc, extend_prefix_lens=None,
                extend_seq_lens=None):
        """decode when extend_seq_lens is None; returns next-token logits [bs, vocab]."""
        cfg, W = self.cfg, self.w
        hq, hkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        h = W["embed"][input_ids].to(self.dtype)
        residual = None
        for l in range(cfg.num_hidden_layers):
            L = W["layers"][l]
            if residual is None:
                residual, h = h, self._norm(h, L["ln1"])
            else:
                h, residual = self._norm(h, L["ln1"], residual)
            if isinstance(L["qkv"][0], str):
                qkv = self._linear(h, L["qkv"])
            else:
                qkv = self._linear(h, L["qkv"][:2], L["qkv"][2])
            q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
            q, k = oe.rope(positions, q.contiguous(), k.contiguous(), d, self.cos_sin, True)
            self.k_buf[l][out_cache_loc] = k.view(-1, hkv, d)
            self.v_buf[l][out_cache_loc] = v.reshape(-1, hkv, d)
            qh = q.view(-1, hq, d)
            if self.exact:
                if extend_seq_lens is None:
                    o = oa.decode_attention_f64(qh, self.k_buf[l], self.v_buf[l], req_to_token, req_pool_indices, seq_lens, d ** -0.5)
                else:
                    o = oa.extend_attention_f64(qh, self.k_buf[l], self.v_buf[l], req_to_token, req_pool_indices, seq_lens,
                                                extend_prefix_lens, extend_seq_lens, d ** -0.5)
            elif extend_seq_lens is None:
                o = oa.decode_attention_sdpa(qh, self.k_buf[l], self.v_buf[l], req_to_token, req_pool_indices, seq_lens, d ** -0.5)
            else:
                o = oa.extend_attention_sdpa(qh, self.k_buf[l], self.v_buf[l], req_to_token, req_pool_indices, seq_lens,
                                             extend_prefix_lens, extend_seq_lens, d ** -0.5)
            a = self._linear(o.reshape(-1, hq * d), L["o"])
            h, residual = self._norm(a, L["ln2"], residual)
            gu = self._linear(h, L["gate_up"])
            h = self._linear(oe.silu_and_mul(gu), L["down"])
        h, _ = self._norm(h, W["norm"], residual)
        if extend_seq_lens is not None:
            h = h[torch.cumsum(extend_seq_lens.long(), 0) - 1]
        if self.exact:
            return h @ W["lm_head"].to(torch.float64).t()
        return F.linear(h, W["lm_head"])
