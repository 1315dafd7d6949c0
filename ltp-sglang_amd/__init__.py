"""MI355X-native hot path for sglang-style serving (attention over the paged KV pool + fp8/AWQ GEMM).

The directory name carries a hyphen, so the package is registered under the import
name ``ltp_sglang_amd`` by ``__graft_entry__.load_package()`` (repo root).  Layout:

  csrc/        hand-written HIP kernels for gfx950 + the C-ABI (include/sgl_mi355.h)
  _cabi.py     ctypes binding of lib/libsgl_mi355.so; fails loudly when it is missing
  sgl_kernel/  the reference's ``sgl_kernel`` Python op API for this path
  srt/         host-side mirror of the reference's plugin surface (AttentionBackend,
               ForwardBatch, KV pools, allocator, radix cache, quantized linear methods)
"""

__version__ = "0.1.0"
