"""Oracle for the elementwise neighbours, restating the reference's forward_native methods.

  rmsnorm        RMSNorm.forward_native          python/sglang/srt/layers/layernorm.py:135-171
  silu_and_mul   SiluAndMul.forward_native       python/sglang/srt/layers/activation.py:60-63
  rope           RotaryEmbedding.forward_native  python/sglang/srt/layers/rotary_embedding.py:49-72,103-165

TEST INFRASTRUCTURE: see oracle/__init__.py.
"""
import torch
import torch.nn.functional as F


def rmsnorm(x, weight, eps, residual=None):
    dt = x.dtype
    xf = x.to(torch.float32)
    if residual is not None:
        xf = xf + residual.to(torch.float32)
        residual = xf.to(dt)
    var = xf.pow(2).mean(dim=-1, keepdim=True)
    y = ((xf * torch.rsqrt(var + eps)) * weight).to(dt)
    return y if residual is None else (y, residual)


def silu_and_mul(x):
    d = x.shape[-1] // 2
    return F.silu(x[..., :d]) * x[..., d:]


def rope_cache(head_size, rotary_dim, max_pos, base):
    inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
    t = torch.arange(max_pos, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1)


def rope(positions, q, k, head_size, cos_sin_cache, is_neox=True):
    """q [T, Hq*hs], k [T, Hk*hs] -> rotated copies (arithmetic in the tensors' dtype, like the reference)."""
    rot = cos_sin_cache.shape[1]
    cs = cos_sin_cache.index_select(0, positions.flatten())
    cos, sin = cs.chunk(2, dim=-1)

    def one(x):
        t = x.shape[0]
        xv = x.view(t, -1, head_size)
        xr, xp = xv[..., :rot], xv[..., rot:]
        c, s = cos.unsqueeze(-2).to(x.dtype), sin.unsqueeze(-2).to(x.dtype)
        if is_neox:
            x1, x2 = torch.chunk(xr, 2, dim=-1)
        else:
            x1, x2 = xr[..., ::2], xr[..., 1::2]
        o1 = x1 * c - x2 * s
        o2 = x2 * c + x1 * s
        o = torch.cat((o1, o2), dim=-1) if is_neox else torch.stack((o1, o2), dim=-1).flatten(-2)
        return torch.cat((o, xp), dim=-1).reshape(x.shape)

    return one(q), one(k)
