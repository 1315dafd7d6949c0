"""SiluAndMul (python/sglang/srt/layers/activation.py:60-70) on the HIP kernel."""
import torch
from torch import nn

from ...sgl_kernel import silu_and_mul


class SiluAndMul(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return silu_and_mul(x if x.is_contiguous() else x.contiguous())
