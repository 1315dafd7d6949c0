"""fp8 / AWQ oracle: CPU restatements of the reference's own torch reference formulas.

  per_token_quant_fp8   scale = absmax/448 as computed by sgl-kernel/csrc/gemm/per_token_quant_fp8.cu:49-57;
                        q as torch_per_token_quant_fp8, sgl-kernel/tests/test_per_token_quant_fp8.py:14-23
  per_tensor_quant_fp8  sgl-kernel/csrc/gemm/per_tensor_quant_fp8.cu:10-52 + torch_scaled_fp8_quant,
                        sgl-kernel/tests/test_per_tensor_quant_fp8.py:30-37
  per_token_group_quant sgl-kernel/csrc/gemm/per_token_group_quant_8bit.cu (y_s = max(absmax, eps)/max; q = clamp(x / y_s))
  scaled_mm             torch_scaled_mm, sgl-kernel/tests/test_fp8_gemm.py:6-14
  awq_dequantize        awq_dequantize_torch + reverse_awq_order, sgl-kernel/tests/test_awq_dequant.py:9-58

TEST INFRASTRUCTURE: see oracle/__init__.py.
"""
import torch

FP8_MAX = 448.0
AWQ_ORDER = [0, 4, 1, 5, 2, 6, 3, 7]


def _to_fp8(x32):
    return x32.clamp(min=-FP8_MAX, max=FP8_MAX).to(torch.float8_e4m3fn)


def per_token_quant_fp8(x):
    """x [M,K] bf16/f16 -> (q e4m3fn [M,K], scale f32 [M,1])."""
    xf = x.to(torch.float32)
    scale = xf.abs().amax(dim=1, keepdim=True) / FP8_MAX
    inv = torch.where(scale == 0, torch.zeros_like(scale), scale.reciprocal())
    return _to_fp8(xf * inv), scale


def per_tensor_quant_fp8(x, scale=None):
    """Dynamic (scale=None) or static per-tensor quantisation -> (q, scale f32 [1])."""
    xf = x.to(torch.float32)
    if scale is None:
        scale = (xf.abs().amax() / FP8_MAX).reshape(1)
    return _to_fp8(xf * scale.reciprocal()), scale


def per_token_group_quant_fp8(x, group_size, eps=1e-10, fp8_min=-FP8_MAX, fp8_max=FP8_MAX):
    xf = x.to(torch.float32).reshape(-1, group_size)
    amax = xf.abs().amax(dim=1, keepdim=True).clamp(min=eps)
    ys = amax / fp8_max
    q = (xf / ys).clamp(min=fp8_min, max=fp8_max).to(torch.float8_e4m3fn)
    return q.reshape(x.shape), ys.reshape(*x.shape[:-1], x.shape[-1] // group_size)


def scaled_mm(a, b, scale_a, scale_b, out_dtype, bias=None):
    """a [M,K] fp8, b [K,N] fp8 -> out_dtype [M,N]; same op order as the reference test."""
    o = torch.matmul(a.to(torch.float32), b.to(torch.float32))
    o = (o * scale_a.view(-1, 1)) * scale_b.view(1, -1)
    o = o.to(out_dtype)
    if bias is not None:
        o = o + bias.view(1, -1)
    return o


def _awq_unpack(packed):
    """int32 [R, C] -> int [R, 8C] in AWQ column order (nibble order[j] holds column 8c + j)."""
    shifts = torch.tensor([4 * o for o in AWQ_ORDER], dtype=torch.int32)
    return ((packed.unsqueeze(-1) >> shifts) & 0xF).reshape(packed.shape[0], -1)


def awq_repack(qweight, scales, qzeros):
    """Layout oracle of sgl_mi355_awq_repack (include/sgl_mi355.h): qpacked int32 [N/16, K/128, 64, 4] whose word s of lane
    (a = n % 16, g) holds q[128 b + 32 s + 8 g + e][n] in nibble (e & 1) * 4 + e / 2; sz int32 [K/G, N] = zero << 16 | scale bits (bf16 scales),
    (0xE400 | zero) << 16 | scale bits (f16 scales: the upper half is the f16 bit pattern of -(1024 + zero))."""
    k, nc = qweight.shape
    n = nc * 8
    q = _awq_unpack(qweight).to(torch.int64)                                 # [K, N] natural column order
    z = _awq_unpack(qzeros).to(torch.int64)                                  # [K/G, N]
    qb = q.reshape(k // 128, 4, 4, 8, n // 16, 16)                           # [b, s, g, e, t, a]
    nib = torch.tensor([(e & 1) * 4 + (e >> 1) for e in range(8)])           # element e of the lane's k-run -> nibble
    words = (qb << (4 * nib.view(1, 1, 1, 8, 1, 1))).sum(dim=3)              # [b, s, g, t, a]
    words = words.permute(3, 0, 2, 4, 1).reshape(n // 16, k // 128, 64, 4)   # [t, b, (g, a), s]
    sbits = scales.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF
    sz = (((z | 0xE400) if scales.dtype == torch.float16 else z) << 16) | sbits
    to_i32 = lambda t: torch.where(t >= 2**31, t - 2**32, t).to(torch.int32)
    return to_i32(words), to_i32(sz)


def awq_dequantize(qweight, scales, qzeros, group_size=None):
    """qweight [K, N/8] i32, scales [K/G, N] f16/bf16, qzeros [K/G, N/8] i32 -> [K, N] in scales.dtype."""
    k = qweight.shape[0]
    g = group_size or k // scales.shape[0]
    w = _awq_unpack(qweight).to(torch.int8)
    z = _awq_unpack(qzeros).to(torch.int8).repeat_interleave(g, dim=0)
    return (w - z) * scales.repeat_interleave(g, dim=0)
