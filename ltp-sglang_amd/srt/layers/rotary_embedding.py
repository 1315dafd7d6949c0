"""RotaryEmbedding (python/sglang/srt/layers/rotary_embedding.py:75-236): f32 cos/sin cache + the HIP in-place kernel."""
from typing import Optional, Tuple

import torch
from torch import nn

from ...sgl_kernel import apply_rope_with_cos_sin_cache_inplace


class RotaryEmbedding(nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float, is_neox_style: bool,
                 dtype: torch.dtype):
        super().__init__()
        self.head_size, self.rotary_dim = head_size, rotary_dim
        self.max_position_embeddings, self.base = max_position_embeddings, base
        self.is_neox_style, self.dtype = is_neox_style, dtype
        inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
        freqs = torch.einsum("i,j -> ij", torch.arange(max_position_embeddings, dtype=torch.float), inv_freq)
        self.register_buffer("cos_sin_cache", torch.cat((freqs.cos(), freqs.sin()), dim=-1), persistent=False)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                offsets: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if offsets is not None:
            positions = positions + offsets
        apply_rope_with_cos_sin_cache_inplace(positions, query, key, self.head_size, self.cos_sin_cache, self.is_neox_style)
        return query, key


def get_rope(head_size, rotary_dim, max_position, base, is_neox_style=True, rope_scaling=None, dtype=None):
    if rope_scaling is not None:
        raise RuntimeError("scaled RoPE variants are outside this build's hot path")
    return RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, dtype or torch.bfloat16)
