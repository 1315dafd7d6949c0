"""ctypes binding of the C-ABI shared library (include/sgl_mi355.h).

The product path has no CPU fallback: if the HIP library is missing or an entry
point is absent, importing this module raises.  Errors reported by the library are
re-raised as RuntimeError carrying the library's message, the same convention as the
reference's TORCH_CHECK (sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1078-1108).
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsgl_mi355.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with ltp-sglang_amd/csrc/build.sh "
        "(or __graft_entry__.build()); there is no fallback path"
    )

lib = ctypes.CDLL(LIB_PATH)

BF16, F16, F32, FP8_E4M3 = 0, 1, 2, 3

# name -> (restype, argtypes); kept in one table so tests can check it against the header
SIGNATURES = {
    "sgl_mi355_last_error": (ctypes.c_char_p, []),
    "sgl_mi355_abi_version": (c_int, []),
    "sgl_mi355_device_cu_count": (c_int, [c_int]),
    "sgl_mi355_decode_attention": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64,
         c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
         c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_void_p],
    ),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == library/header mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


def last_error() -> str:
    return lib.sgl_mi355_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(last_error() or f"sgl_mi355 call failed with status {rc}")


def dtype_code(dtype) -> int:
    import torch

    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float16:
        return F16
    if dtype == torch.float32:
        return F32
    if dtype == torch.float8_e4m3fn:
        return FP8_E4M3
    raise RuntimeError(f"unsupported dtype {dtype}")


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
