"""RadixAttention: the attention layer object the model holds; it only carries per-layer constants and forwards
to ``forward_batch.attn_backend`` (python/sglang/srt/layers/radix_attention.py:39-110)."""
from enum import Enum

from torch import nn


class AttentionType(Enum):
    DECODER = "decoder"
    ENCODER_ONLY = "encoder_only"


class RadixAttention(nn.Module):
    def __init__(self, num_heads, head_dim, scaling, num_kv_heads, layer_id, logit_cap=0.0, v_head_dim=-1,
                 sliding_window_size=-1, is_cross_attention=False, quant_config=None,
                 attn_type=AttentionType.DECODER, use_irope=False, prefix=""):
        super().__init__()
        self.tp_q_head_num = num_heads
        self.tp_k_head_num = self.tp_v_head_num = num_kv_heads
        self.head_dim = self.qk_head_dim = head_dim
        self.v_head_dim = head_dim if v_head_dim == -1 else v_head_dim
        self.scaling = scaling
        self.layer_id = layer_id
        self.logit_cap = logit_cap
        self.sliding_window_size = sliding_window_size or -1
        self.is_cross_attention = is_cross_attention
        self.use_irope = use_irope
        self.attn_type = attn_type
        self.k_scale = self.v_scale = None
        self.k_scale_float = self.v_scale_float = None
        self.quant_method = None
        if quant_config is not None:
            self.quant_method = quant_config.get_quant_method(self, prefix=prefix)
            if self.quant_method is not None:
                self.quant_method.create_weights(self)

    def forward(self, q, k, v, forward_batch, save_kv_cache=True, **kwargs):
        if k is not None:  # cross-layer KV sharing passes None
            assert v is not None
            k = k.view(-1, self.tp_k_head_num, self.qk_head_dim)
            v = v.view(-1, self.tp_v_head_num, self.v_head_dim)
        return forward_batch.attn_backend.forward(q, k, v, self, forward_batch, save_kv_cache, **kwargs)
