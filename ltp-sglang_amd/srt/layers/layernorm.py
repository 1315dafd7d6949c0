"""RMSNorm module (python/sglang/srt/layers/layernorm.py:60-171) on the HIP rmsnorm / fused_add_rmsnorm kernels."""
from typing import Optional

import torch
from torch import nn

from ...sgl_kernel import fused_add_rmsnorm, rmsnorm


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6, dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=dtype), requires_grad=False)
        self.variance_epsilon = eps
        self.hidden_size = hidden_size

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None):
        if not x.is_contiguous():
            x = x.contiguous()
        if residual is not None:
            fused_add_rmsnorm(x, residual, self.weight.data, self.variance_epsilon)
            return x, residual
        return rmsnorm(x, self.weight.data, self.variance_epsilon)
