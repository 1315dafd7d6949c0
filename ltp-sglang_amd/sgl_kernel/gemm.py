"""GEMM / quantisation ops with the reference's ``sgl_kernel`` signatures
(sgl-kernel/python/sgl_kernel/gemm.py:7-10,34-42,100-146)."""
from typing import Optional, Tuple

import os

import torch

from .._cabi import check, current_stream, dtype_code, lib, ptr

_WORKSPACES = {}
_RETIRED = []  # superseded workspaces: never freed (a captured HIP graph may still hold their address)
WORKSPACE_FLOATS = 1 << 25  # 128 MiB: 4 k-ranges x 64 rows x 128 256 columns (70B-shaped lm_head at M = 64) fits


def _workspace(device, need: int):
    """The per-device f32 scratch shared by the skinny, tiled and AWQ GEMMs: split-K slabs at its head, the persistent tile
    kernel's 64 ticket slots in its last 512 words.  ONE per device, not per stream (torch.cuda.graph captures on a fresh side
    stream each time: a per-stream buffer would be allocated -- and zero-filled -- inside every captured graph), so GEMMs of two
    streams of one process must not run side by side; the model runner and the bench use one stream per process.

    Its address must stay valid for every HIP graph that captured a launch using it, so it is allocated ONCE at
    ``WORKSPACE_FLOATS`` (covers every shape of the BASELINE configs); a larger request allocates a new buffer and the old
    one is parked in ``_RETIRED`` for the life of the process instead of being returned to the caching allocator (where
    another tensor would receive its memory while an old graph still writes split-K partials into it)."""
    buf = _WORKSPACES.get(device)
    need = int(need) + 512   # the ticket slots are never part of a slab set (a set that filled the buffer overlapped them)
    if buf is None or buf.numel() < need:
        if buf is not None:
            _RETIRED.append(buf)
        # (the last 512 words are the persistent tile kernel's ticket slots, zeroed by its launcher before every launch)
        buf = torch.zeros(max(need, WORKSPACE_FLOATS), dtype=torch.float32, device=device)
        _WORKSPACES[device] = buf
    return buf, buf.numel()


def _splitk_workspace(m, n, k, dtype, device):
    """f32 slab workspace for the split-K form of the weight-streaming GEMM."""
    kr = lib.sgl_mi355_skinny_gemm_num_kranges(m, n, k, dtype_code(dtype))
    if kr <= 1:
        return None, 0
    return _workspace(device, kr * m * n)


def _tiled_workspace(device):
    """f32 scratch for the tiled GEMM's split-K (few-tile launches: decode at 64 < M <= 256)."""
    return _workspace(device, 1 << 24)


def _cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("sgl_kernel (MI355X) ops need device tensors; there is no CPU path")


def awq_dequantize(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor) -> torch.Tensor:
    """qweight [K, N/8] int32, scales [K/G, N] f16/bf16, qzeros [K/G, N/8] int32 -> [K, N] (gemm.py:7-10)."""
    _cuda(qweight, scales, qzeros)
    if qweight.dtype != torch.int32 or qzeros.dtype != torch.int32:
        raise RuntimeError("qweight and qzeros must be int32")
    assert qweight.is_contiguous() and scales.is_contiguous() and qzeros.is_contiguous()
    k, nc = qweight.shape
    groups = scales.shape[0]
    if scales.shape[1] != nc * 8 or tuple(qzeros.shape) != (groups, nc) or k % groups != 0:
        raise RuntimeError("awq_dequantize: inconsistent qweight / scales / qzeros shapes")
    out = torch.empty((k, nc * 8), dtype=scales.dtype, device=qweight.device)
    check(lib.sgl_mi355_awq_dequantize(ptr(qweight), ptr(scales), ptr(qzeros), ptr(out), k, nc, k // groups,
                                       dtype_code(scales.dtype), current_stream()))
    return out


def awq_repack(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """One-off re-layout of an AWQ weight into MFMA-fragment order for :func:`awq_gemm`.
    Returns (qpacked int32 [N/16, K/128, 64, 4], sz int32 [K/G, N] = zero << 16 | scale bits; f16 scales:
    (0xE400 | zero) << 16 | scale bits, include/sgl_mi355.h)."""
    _cuda(qweight, scales, qzeros)
    if qweight.dtype != torch.int32 or qzeros.dtype != torch.int32:
        raise RuntimeError("qweight and qzeros must be int32")
    assert qweight.is_contiguous() and scales.is_contiguous() and qzeros.is_contiguous()
    k, nc = qweight.shape
    groups, n = scales.shape
    if n != nc * 8 or tuple(qzeros.shape) != (groups, nc) or k % groups != 0:
        raise RuntimeError("awq_repack: inconsistent qweight / scales / qzeros shapes")
    qpacked = torch.empty((n // 16, k // 128, 64, 4), dtype=torch.int32, device=qweight.device)
    sz = torch.empty((groups, n), dtype=torch.int32, device=qweight.device)
    check(lib.sgl_mi355_awq_repack(ptr(qweight), ptr(scales), ptr(qzeros), ptr(qpacked), ptr(sz), k, n, k // groups,
                                   dtype_code(scales.dtype), current_stream()))
    return qpacked, sz


def awq_unpack_nk(qpacked: torch.Tensor, sz: torch.Tensor, group_size: int, dtype) -> torch.Tensor:
    """Dense W [N, K] (row-major) from the image made by :func:`awq_repack`: awq_dequantize(...).t() in one pass."""
    _cuda(qpacked, sz)
    n, k = sz.shape[1], qpacked.shape[1] * 128
    out = torch.empty((n, k), dtype=dtype, device=qpacked.device)
    check(lib.sgl_mi355_awq_unpack_nk(ptr(qpacked), ptr(sz), ptr(out), n, k, int(group_size), dtype_code(dtype), current_stream()))
    return out


def awq_gemm(x: torch.Tensor, qpacked: torch.Tensor, sz: torch.Tensor, group_size: int,
             bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y [M, N] = x [M, K] @ dequant(W) (+ bias) for M <= 64 on a weight re-laid by :func:`awq_repack`; the weight values
    are exactly awq_dequantize's (awq.py:401-418 computes the same product as dequantize + matmul)."""
    _cuda(x, qpacked, sz, bias)
    assert x.dim() == 2 and x.stride(1) == 1 and qpacked.is_contiguous() and sz.is_contiguous()
    m, k = x.shape
    n = sz.shape[1]
    if qpacked.shape[0] * 16 != n or qpacked.shape[1] * 128 != k:
        raise RuntimeError("awq_gemm: x / qpacked / sz shapes do not match")
    out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    kr = lib.sgl_mi355_awq_gemm_num_kranges(m, k)
    ws, ws_n = None, 0
    if kr > 1:
        ws, ws_n = _workspace(x.device, kr * m * n)
    check(lib.sgl_mi355_awq_gemm(ptr(x), x.stride(0), ptr(qpacked), ptr(sz), ptr(out), out.stride(0), ptr(bias), m, n, k,
                                 int(group_size), dtype_code(x.dtype), ptr(ws), ws_n, current_stream()))
    return out


def awq_set_exact_weights(on: bool) -> None:
    """True: awq_gemm (and the fused int4 forms) multiply by awq_dequantize's rounded weights for bf16 as well -- bit-compatible
    with the reference's dequantise + matmul, for parity runs (also: SGL_MI355_AWQ_EXACT_WEIGHTS=1 at import).  False (default):
    the offset form for bf16 (exact (q - z) s)."""
    check(lib.sgl_mi355_awq_set_exact_weights(1 if on else 0))


if os.environ.get("SGL_MI355_AWQ_EXACT_WEIGHTS", "0") == "1":
    awq_set_exact_weights(True)


def awq_gemm_num_kranges(m: int, k: int) -> int:
    """Split-K ranges of awq_gemm for this M and K (1: no slabs)."""
    return int(lib.sgl_mi355_awq_gemm_num_kranges(int(m), int(k)))


def fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias=None, out=None):
    """out[M,N] = (mat_a[M,K] @ mat_b[K,N]) * scales_a[m] * scales_b[n] (+ bias[n]).

    ``out`` (not in the reference's signature): an [M, N] tensor of ``out_dtype`` to write into; its rows may be padded (row stride
    >= N, a multiple of 8 elements) -- the prefill path hands in a qkv buffer whose row stride is not a multiple of 4 KiB.

    Argument checks and messages follow fp8_scaled_mm in
    sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1078-1108.
    """
    if not mat_a.is_cuda:
        raise RuntimeError("mat_a must be a CUDA tensor")
    if not mat_b.is_cuda:
        raise RuntimeError("mat_b must be a CUDA tensor")
    if mat_a.dim() != 2:
        raise RuntimeError("mat_a must be a 2D tensor")
    if mat_b.dim() != 2:
        raise RuntimeError("mat_b must be a 2D tensor")
    if mat_a.stride(1) != 1:
        raise RuntimeError("mat_a must be a row major tensor")
    if mat_b.stride(0) != 1:
        raise RuntimeError("mat_a must be a column major tensor")  # (sic) reference wording
    if mat_a.size(1) != mat_b.size(0):
        raise RuntimeError("mat_a and mat_b shapes cannot be multiplied")
    if (mat_a.size(1) * mat_a.element_size()) % 16 != 0:
        raise RuntimeError("mat_a must be multiple of 16 bytes for memory alignment")
    if (mat_b.size(0) * mat_b.element_size()) % 16 != 0:
        raise RuntimeError("mat_b must be multiple of 16 bytes for memory alignment")
    if mat_a.dtype != torch.float8_e4m3fn:
        raise RuntimeError("mat_a must be Float8_e4m3fn")
    if mat_b.dtype != torch.float8_e4m3fn:
        raise RuntimeError("mat_b must be Float8_e4m3fn")
    if out_dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("out_dtype must be Half or BFloat16")
    if scales_a.numel() != mat_a.size(0):
        raise RuntimeError("size of scales_a is not matched")
    if scales_b.numel() != mat_b.size(1):
        raise RuntimeError("size of scales_b is not matched")
    if not scales_a.is_contiguous():
        raise RuntimeError("scales_a must be contiguous")
    if not scales_b.is_contiguous():
        raise RuntimeError("scales_b msut be contiguous")
    if scales_a.dtype != torch.float32:
        raise RuntimeError("scales_a must be Float32")
    if scales_b.dtype != torch.float32:
        raise RuntimeError("scales_b must be Float32")
    if bias is not None:
        if bias.numel() != mat_b.size(1):
            raise RuntimeError("size of bias is not matched")
        if not bias.is_contiguous():
            raise RuntimeError("bias must be contiguous")
        if bias.dtype != out_dtype:
            raise RuntimeError("bias dtype must match output dtype")
    m, k = mat_a.shape
    n = mat_b.size(1)
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=mat_a.device)
    elif (out.shape != (m, n) or out.dtype != out_dtype or out.device != mat_a.device or out.stride(1) != 1 or out.stride(0) < n
          or (out.stride(0) * out.element_size()) % 16 != 0 or out.data_ptr() % 16 != 0):
        raise RuntimeError("out must be an [M, N] tensor of out_dtype with contiguous, 16-byte aligned rows")
    if (out.size(1) * out.element_size()) % 16 != 0:
        raise RuntimeError("out must be multiple of 16 bytes for memory alignment")
    w_stride = mat_b.stride(1)  # W[n, :] = mat_b[:, n]
    if m <= 64:
        ws, ws_n = _splitk_workspace(m, n, k, mat_a.dtype, mat_a.device)
        check(lib.sgl_mi355_skinny_gemm(ptr(mat_a), mat_a.stride(0), ptr(mat_b), w_stride, ptr(out), out.stride(0),
                                        ptr(scales_a), ptr(scales_b), ptr(bias), m, n, k, dtype_code(mat_a.dtype),
                                        dtype_code(out_dtype), ptr(ws), ws_n, current_stream()))
    else:
        ws, ws_n = _tiled_workspace(mat_a.device)
        check(lib.sgl_mi355_fp8_gemm(ptr(mat_a), mat_a.stride(0), ptr(mat_b), w_stride, ptr(out), out.stride(0),
                                     ptr(scales_a), ptr(scales_b), ptr(bias), m, n, k, dtype_code(out_dtype), ptr(ws), ws_n,
                                     current_stream()))
    return out


def fp8_linear_slabs(x_q: torch.Tensor, weight_nk: torch.Tensor, m: int, n: int, k: int, out: Optional[torch.Tensor] = None,
                     min_kranges: int = 1) -> torch.Tensor:
    """Raw f32 split-K partial sums [S, M, N] of x_q[M,K] @ weight_nk[N,K]^T (no scales): the producer half of the
    launch-boundary split-K reduce; the consumer (fused_add_rmsnorm_quant_fp8 with ``slabs=``) applies the scales.
    Operands fp8 (fp8_scaled_mm) or bf16 / f16 (the unquantised linear: the consumer gets no scales)."""
    _cuda(x_q, weight_nk)
    assert x_q.dtype == weight_nk.dtype
    kr = lib.sgl_mi355_skinny_gemm_slabs_count_min(m, k * x_q.element_size(), int(min_kranges))
    if out is None:
        out = torch.empty((kr, m, n), dtype=torch.float32, device=x_q.device)
    check(lib.sgl_mi355_skinny_gemm_slabs_min(ptr(x_q), x_q.stride(0), ptr(weight_nk), weight_nk.stride(0), ptr(out), m, n, k,
                                              dtype_code(x_q.dtype), int(min_kranges), current_stream()))
    return out


def fp8_gemm_num_slabs(m: int, n: int, k: int, device) -> int:
    """How many f32 [M, N] split-K slabs fp8_scaled_mm forms for this shape where it runs the streaming tile (64 < M <= 256, and
    larger M with too few 256-wide tiles to fill the chip; 1: none)."""
    if m <= 64:
        return 1
    _, ws_n = _tiled_workspace(device)
    return int(lib.sgl_mi355_fp8_gemm_num_slabs(int(m), int(n), int(k), ws_n))


def fp8_gemm_slabs(x_q: torch.Tensor, weight_nk: torch.Tensor) -> torch.Tensor:
    """The M > 64 form of fp8_linear_slabs: raw f32 partial sums [S, M, N] of fp8_scaled_mm's streaming tile, with its
    own k-range partition (so the consumer's sum equals fp8_scaled_mm's); requires fp8_gemm_num_slabs(...) > 1."""
    _cuda(x_q, weight_nk)
    assert x_q.dtype == torch.float8_e4m3fn and weight_nk.dtype == torch.float8_e4m3fn
    m, k = x_q.shape
    n = weight_nk.shape[0]
    _, ws_n = _tiled_workspace(x_q.device)
    kr = int(lib.sgl_mi355_fp8_gemm_num_slabs(m, n, k, ws_n))
    if kr <= 1:
        raise RuntimeError(f"fp8_gemm_slabs: M={m} N={n} K={k} runs as one k-range (no slabs)")
    out = torch.empty((kr, m, n), dtype=torch.float32, device=x_q.device)
    check(lib.sgl_mi355_fp8_gemm_slabs(ptr(x_q), x_q.stride(0), ptr(weight_nk), weight_nk.stride(0), ptr(out), m, n, k, ws_n,
                                       current_stream()))
    return out


def dense_linear_kranges(m: int, n: int, k: int, dtype) -> int:
    """How many split-K ranges dense_linear / fp8_scaled_mm use for this shape at M <= 64 (1: no slabs; 0: generic kernel)."""
    return int(lib.sgl_mi355_skinny_gemm_num_kranges(m, n, k, dtype_code(dtype))) if m <= 64 else 0


def dense_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, out_dtype=None) -> torch.Tensor:
    """Unquantised F.linear(x, weight, bias) for bf16/f16 (UnquantizedLinearMethod.apply,
    python/sglang/srt/layers/quantization/unquant.py); weight [N, K] row-major."""
    _cuda(x, weight, bias)
    assert x.dim() == 2 and weight.dim() == 2 and x.stride(1) == 1 and weight.stride(1) == 1
    m, k = x.shape
    n = weight.shape[0]
    out_dtype = out_dtype or x.dtype
    out = torch.empty((m, n), dtype=out_dtype, device=x.device)
    if m <= 64:
        ws, ws_n = _splitk_workspace(m, n, k, x.dtype, x.device)
        check(lib.sgl_mi355_skinny_gemm(ptr(x), x.stride(0), ptr(weight), weight.stride(0), ptr(out), out.stride(0), None,
                                        None, ptr(bias), m, n, k, dtype_code(x.dtype), dtype_code(out_dtype), ptr(ws), ws_n,
                                        current_stream()))
    else:
        ws, ws_n = _tiled_workspace(x.device)
        check(lib.sgl_mi355_dense_gemm(ptr(x), x.stride(0), ptr(weight), weight.stride(0), ptr(out), out.stride(0),
                                       ptr(bias), m, n, k, dtype_code(x.dtype), dtype_code(out_dtype), ptr(ws), ws_n,
                                       current_stream()))
    return out


def sgl_per_token_quant_fp8(input: torch.Tensor, output_q: torch.Tensor, output_s: torch.Tensor) -> None:
    """gemm.py:140-145; in-place outputs: output_q e4m3fn [M,K], output_s f32 [M] or [M,1]."""
    _cuda(input, output_q, output_s)
    assert input.dim() == 2 and input.stride(1) == 1 and output_q.is_contiguous() and output_s.is_contiguous()
    assert output_q.dtype == torch.float8_e4m3fn and output_s.dtype == torch.float32
    check(lib.sgl_mi355_per_token_quant_fp8(ptr(input), input.stride(0), ptr(output_q), ptr(output_s), input.shape[0],
                                            input.shape[1], dtype_code(input.dtype), current_stream()))


def sgl_per_tensor_quant_fp8(input: torch.Tensor, output_q: torch.Tensor, output_s: torch.Tensor, is_static: bool) -> None:
    """gemm.py:129-137; dynamic mode expects output_s zero-initialised (fp8_kernel.py scaled_fp8_quant does torch.zeros)."""
    _cuda(input, output_q, output_s)
    assert input.is_contiguous() and output_q.is_contiguous()
    assert output_q.dtype == torch.float8_e4m3fn and output_s.dtype == torch.float32
    check(lib.sgl_mi355_per_tensor_quant_fp8(ptr(input), ptr(output_q), ptr(output_s), input.numel(), int(bool(is_static)),
                                             dtype_code(input.dtype), current_stream()))


def input_to_float8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp8_utils.py:310-326 on the device: (x_q e4m3fn like x, 1 / scale as a 0-dim f32 tensor); bf16 / f16 input (f32: cast first, as
    checkpoints hold 16-bit weights)."""
    _cuda(x)
    xc = x.contiguous()
    if xc.dtype == torch.float32:
        xc = xc.to(torch.bfloat16)
    x_q = torch.empty_like(xc, dtype=torch.float8_e4m3fn)
    s = torch.empty(2, device=x.device, dtype=torch.float32)   # [1 / scale, amax scratch]
    check(lib.sgl_mi355_input_to_float8(ptr(xc), ptr(x_q), ptr(s), ptr(s[1:]), xc.numel(), dtype_code(xc.dtype), current_stream()))
    return x_q, s[0]


def sgl_per_token_group_quant_fp8(input, output_q, output_s, group_size, eps, fp8_min, fp8_max, scale_ue8m0) -> None:
    """gemm.py:100-112 (row-major float scales only)."""
    _cuda(input, output_q, output_s)
    if scale_ue8m0:
        raise RuntimeError("scale_ue8m0 (DeepSeek block-fp8 layout) is out of scope of this build")
    assert input.is_contiguous() and output_q.is_contiguous() and output_s.is_contiguous()
    if input.numel() % group_size != 0:
        raise RuntimeError("input.numel() must be divisible by group_size")
    check(lib.sgl_mi355_per_token_group_quant_fp8(ptr(input), ptr(output_q), ptr(output_s), input.numel(), int(group_size),
                                                  float(eps), float(fp8_min), float(fp8_max), dtype_code(input.dtype),
                                                  current_stream()))
