"""MI355X implementation of the ``sgl_kernel`` op API for the attention / dequant-GEMM hot path.

Export list follows sgl-kernel/python/sgl_kernel/__init__.py:12-96 restricted to the ops on
the path, plus the attention ops the reference only has as Triton / CPU
kernels.
"""
from .attention import decode_attention_fwd

__all__ = ["decode_attention_fwd"]
