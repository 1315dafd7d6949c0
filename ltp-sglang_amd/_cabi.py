"""ctypes binding of the C-ABI shared library, generated from include/sgl_mi355.h.

The product path has no CPU fallback: if the HIP library is missing, or lacks an entry
point the header declares, importing this module raises.  Errors reported by the library
are re-raised as RuntimeError carrying the library's message -- the convention of the
reference's TORCH_CHECK (sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1078-1108).
"""
import ctypes
import os
import re
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

import torch  # noqa: F401  -- must be loaded BEFORE the library: torch brings its own libamdhip64 and the process
#                  must end up with that single HIP runtime (loading ours first makes torch fail with
#                  "no ROCm-capable device is detected")

_HERE = os.path.dirname(os.path.abspath(__file__))
# SGL_MI355_LIB selects another build of the same library (tools/ab_libs.sh: same-box A/B without touching the product file)
LIB_PATH = os.environ.get("SGL_MI355_LIB") or os.path.join(_HERE, "lib", "libsgl_mi355.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sgl_mi355.h")

BF16, F16, F32, FP8_E4M3 = 0, 1, 2, 3


def parse_header(path=HEADER_PATH):
    """[(name, restype, [argtypes])] for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = []
    for m in re.finditer(r"\n\s*(const char\*|void\*|int64_t|int)\s+(sgl_mi355_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argtypes.append(c_void_p)
                elif a.startswith("int64_t"):
                    argtypes.append(c_int64)
                elif a.startswith("int"):
                    argtypes.append(c_int)
                elif a.startswith("float"):
                    argtypes.append(c_float)
                else:
                    raise RuntimeError(f"{name}: cannot map C parameter '{a}'")
        restype = {"const char*": c_char_p, "void*": c_void_p, "int64_t": c_int64, "int": c_int}[ret]
        protos.append((name, restype, argtypes))
    return protos


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with ltp-sglang_amd/csrc/build.sh "
        "(or __graft_entry__.build()); there is no fallback path"
    )

lib = ctypes.CDLL(LIB_PATH)
SIGNATURES = {}
for _name, _res, _args in parse_header():
    _fn = getattr(lib, _name)  # AttributeError == library/header mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args
    SIGNATURES[_name] = (_res, _args)


def _install_trace(path: str) -> None:
    """SGL_MI355_TRACE=<file>: every C-ABI call is written to <file> BEFORE it runs (flushed to disk), the device is waited for
    after it, and 'ok' is appended -- the last line without 'ok' names the launch a GPU memory fault belongs to (faults are
    asynchronous: without the per-call wait the process dies at some later, unrelated call).  A debugging aid; never set in a
    timed run."""
    import functools

    log = open(path.replace("{rank}", os.environ.get("RANK", "0")), "a", buffering=1)   # "{rank}": one file per rank

    def wrap(name, fn):
        @functools.wraps(fn)
        def traced(*args):
            log.write(f"{name}({', '.join(str(a) for a in args)})")
            log.flush()
            os.fsync(log.fileno())
            rc = fn(*args)
            if name not in ("sgl_mi355_last_error",) and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize()
            log.write(f" -> {rc} ok\n")
            return rc

        return traced

    for name in SIGNATURES:
        setattr(lib, name, wrap(name, getattr(lib, name)))


if os.environ.get("SGL_MI355_TRACE"):
    _install_trace(os.environ["SGL_MI355_TRACE"])


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


def is64(t) -> int:
    import torch

    if t is None:
        return 0
    if t.dtype == torch.int64:
        return 1
    if t.dtype == torch.int32:
        return 0
    raise RuntimeError(f"index tensor must be int32 or int64, got {t.dtype}")


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
