"""Fused decode-step ops (no counterpart in the reference's op list: each equals a sequence of reference ops and is
bit-identical to running this build's stand-alone kernels in that sequence)."""
from typing import Optional, Tuple

import torch

from .._cabi import check, current_stream, dtype_code, lib, ptr


def fused_add_rmsnorm_quant_fp8(x: Optional[torch.Tensor], residual: Optional[torch.Tensor], weight: torch.Tensor, eps: float,
                                slabs: Optional[torch.Tensor] = None, slab_sx: Optional[torch.Tensor] = None,
                                slab_sw: Optional[torch.Tensor] = None, want_norm: bool = False,
                                want_quant: bool = True, dtype=None):
    """residual += x (in place, if given); y = rmsnorm(sum) * weight; returns (y or None, y_q fp8, y_scale [M,1]).
    ``slabs`` [S, M, H] f32 replaces x by the fused split-K combine ((sum_s slabs) * slab_sx[m] * slab_sw[n])."""
    if slabs is not None:
        _, m, h = slabs.shape
        dt = dtype or weight.dtype
        dev = slabs.device
    else:
        m, h = x.shape
        dt, dev = x.dtype, x.device
        assert x.is_contiguous()
    out_norm = torch.empty((m, h), dtype=dt, device=dev) if want_norm else None
    out_q = torch.empty((m, h), dtype=torch.float8_e4m3fn, device=dev) if want_quant else None
    out_s = torch.empty((m, 1), dtype=torch.float32, device=dev) if want_quant else None
    check(lib.sgl_mi355_fused_add_rmsnorm_quant_fp8(ptr(x), ptr(slabs), 0 if slabs is None else slabs.shape[0], ptr(slab_sx),
                                                    ptr(slab_sw), ptr(residual), ptr(weight), float(eps), ptr(out_norm),
                                                    ptr(out_q), ptr(out_s), m, h, dtype_code(dt), current_stream()))
    return out_norm, out_q, out_s


def silu_and_mul_quant_fp8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    m, d2 = x.shape
    d = d2 // 2
    assert x.is_contiguous()
    out_q = torch.empty((m, d), dtype=torch.float8_e4m3fn, device=x.device)
    out_s = torch.empty((m, 1), dtype=torch.float32, device=x.device)
    check(lib.sgl_mi355_silu_and_mul_quant_fp8(ptr(x), ptr(out_q), ptr(out_s), m, d, dtype_code(x.dtype), current_stream()))
    return out_q, out_s


def rope_set_kv(positions, query, key, value, head_size, cos_sin_cache, is_neox, k_buffer, v_buffer, loc) -> None:
    """query [T, Hq*hs], key [T, Hk*hs] rotated in place; (rotated key, value) written to the pool rows ``loc``."""
    t = positions.numel()
    q2, k2, v2 = query.view(t, -1), key.view(t, -1), value.view(t, -1)
    assert q2.stride(1) == 1 and k2.stride(1) == 1 and v2.stride(1) == 1 and loc.dtype == torch.int64
    assert k_buffer[0].is_contiguous() and v_buffer[0].is_contiguous() and cos_sin_cache.dtype == torch.float32
    check(lib.sgl_mi355_rope_set_kv(ptr(positions), ptr(q2), ptr(k2), ptr(v2), ptr(cos_sin_cache), ptr(k_buffer), ptr(v_buffer),
                                    ptr(loc), t, q2.shape[1] // head_size, k2.shape[1] // head_size, head_size,
                                    cos_sin_cache.shape[1], q2.stride(0), k2.stride(0), v2.stride(0), k_buffer.stride(0),
                                    v_buffer.stride(0), int(bool(is_neox)), dtype_code(query.dtype), current_stream()))


def decode_merge_quant_fp8(attn_logits, attn_lse, kv_indptr, num_kv_splits, max_kv_splits, dtype, want_o=False):
    bs, hq, _, dv = attn_logits.shape
    dev = attn_logits.device
    out_o = torch.empty((bs, hq * dv), dtype=dtype, device=dev) if want_o else None
    out_q = torch.empty((bs, hq * dv), dtype=torch.float8_e4m3fn, device=dev)
    out_s = torch.empty((bs, 1), dtype=torch.float32, device=dev)
    check(lib.sgl_mi355_decode_merge_quant_fp8(ptr(attn_logits), ptr(attn_lse), ptr(kv_indptr), None, ptr(num_kv_splits),
                                               int(max_kv_splits), bs, hq, dv, ptr(out_o), ptr(out_q), ptr(out_s),
                                               dtype_code(dtype), current_stream()))
    return out_o, out_q, out_s


def interleave_gate_up_rows(t: torch.Tensor, tile_rows: int = 16) -> torch.Tensor:
    """[2I, ...] (gate rows then up rows) -> tiles of tile_rows = 2H rows: (H gate rows, H up rows); weights, scales, biases."""
    two_i, h = t.shape[0], tile_rows // 2
    assert two_i % tile_rows == 0 and tile_rows in (8, 16)
    rest = t.shape[1:]
    return t.reshape(2, two_i // tile_rows, h, *rest).transpose(0, 1).reshape(two_i, *rest).contiguous()


def interleave_rope_rows(t: torch.Tensor, num_q_heads: int, num_kv_heads: int, head_dim: int = 128, tile_rows: int = 16) -> torch.Tensor:
    """[(Hq + 2 Hkv) * 128, ...] -> inside every q/k head, tiles of tile_rows = 2H rows: (rows Hu..Hu+H-1, rows 64+Hu..)."""
    assert head_dim == 128 and tile_rows in (8, 16)
    rest = t.shape[1:]
    h = tile_rows // 2
    nrope = (num_q_heads + num_kv_heads) * head_dim
    qk = t[:nrope].reshape(num_q_heads + num_kv_heads, 2, 64 // h, h, *rest).transpose(1, 2).reshape(nrope, *rest)
    return torch.cat([qk, t[nrope:]], dim=0).contiguous()


_SILU_TABLE_READY = set()   # device indices whose silu table has been filled AND waited for
_TILE_TICKETS = {}          # device index -> [64, 8] int32: ticket-word slots of sgl_mi355_gemm_silu_mul_ws, taken round robin
_TILE_TICKET_NEXT = [0]


def silu_table_init(device) -> None:
    """Once per device: fills the table the M > 64 SiluAndMul epilogue reads (sgl_mi355_silu_table_init enqueues, this waits, so
    that any stream may use it afterwards).  Must happen outside stream capture: run one eager step before capturing."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx in _SILU_TABLE_READY:
        return
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("fp8_gemm_silu_mul: the silu table of this device is not initialised and the stream is capturing; "
                           "run the op (or sgl_kernel.silu_table_init) once eagerly first")
    with torch.cuda.device(idx):
        check(lib.sgl_mi355_silu_table_init(current_stream()))
        # (allocated here, i.e. never under capture: the ticket words of the prefill form's dynamic tile schedule)
        _TILE_TICKETS[idx] = torch.zeros((64, 8), dtype=torch.int32, device=f"cuda:{idx}")
        torch.cuda.synchronize(idx)
    _SILU_TABLE_READY.add(idx)


def fp8_gemm_silu_mul(x_q, x_s, w_interleaved_nk, w_s_interleaved, out_dtype, tile_rows: int = 16):
    """act [M, I] = SiluAndMul(fp8_scaled_mm(x_q, W)) with W's rows interleaved by interleave_gate_up_rows."""
    m, k = x_q.shape
    n = w_interleaved_nk.shape[0]
    act = torch.empty((m, n // 2), dtype=out_dtype, device=x_q.device)
    if m > 64:
        silu_table_init(x_q.device)
        # prefill sizes: the persistent kernel's ticket words -- one of 64 slots, round robin, so that launches overlapping on two
        # streams do not share them (the launcher zeroes the slot with a memset node; a captured launch keeps the slot it drew)
        idx = x_q.device.index if x_q.device.index is not None else torch.cuda.current_device()
        slot = _TILE_TICKET_NEXT[0] = (_TILE_TICKET_NEXT[0] + 1) & 63
        sched = _TILE_TICKETS[idx][slot]
        st = current_stream()
        check(lib.sgl_mi355_gemm_silu_mul_ws(ptr(x_q), x_q.stride(0), ptr(w_interleaved_nk), w_interleaved_nk.stride(0), ptr(act),
                                             act.stride(0), ptr(x_s), ptr(w_s_interleaved), m, n, k, dtype_code(x_q.dtype),
                                             dtype_code(out_dtype), int(tile_rows), ptr(sched), st))
        return act
    check(lib.sgl_mi355_fp8_gemm_silu_mul(ptr(x_q), x_q.stride(0), ptr(w_interleaved_nk), w_interleaved_nk.stride(0), ptr(act),
                                          act.stride(0), ptr(x_s), ptr(w_s_interleaved), m, n, k, dtype_code(out_dtype),
                                          int(tile_rows), current_stream()))
    return act


def _kv_args(k_buffer, v_buffer, k_scale, v_scale):
    """(kv_dtype code, k_scale, v_scale) of a pool write: the pool's dtype; scales only apply to float8_e4m3fn pools."""
    assert k_buffer.dtype == v_buffer.dtype and k_buffer[0].is_contiguous() and v_buffer[0].is_contiguous()
    return dtype_code(k_buffer.dtype), (-1.0 if k_scale is None else float(k_scale)), (-1.0 if v_scale is None else float(v_scale))


def fp8_qkv_rope_set_kv(x_q, x_s, w_interleaved_nk, w_s_interleaved, bias_interleaved, positions, cos_sin_cache, loc, k_buffer,
                        v_buffer, num_q_heads, num_kv_heads, head_dim, out_dtype, tile_rows: int = 16, k_scale=None, v_scale=None):
    """q [M, Hq*D] (rotated); rotated k and v are written to pool rows ``loc`` of k_buffer / v_buffer (16-bit pools, or
    float8_e4m3fn pools with set_kv_buffer's conversion and the layer's k_scale / v_scale)."""
    m, k = x_q.shape
    q = torch.empty((m, num_q_heads * head_dim), dtype=out_dtype, device=x_q.device)
    assert cos_sin_cache.shape[1] == head_dim
    check(lib.sgl_mi355_qkv_rope_set_kv(ptr(x_q), x_q.stride(0), ptr(w_interleaved_nk), w_interleaved_nk.stride(0), ptr(q),
                                        q.stride(0), ptr(x_s), ptr(w_s_interleaved), ptr(bias_interleaved), ptr(positions),
                                        ptr(cos_sin_cache), ptr(loc), ptr(k_buffer), ptr(v_buffer), k_buffer.stride(0),
                                        v_buffer.stride(0), m, num_q_heads, num_kv_heads, head_dim, k, dtype_code(x_q.dtype),
                                        dtype_code(out_dtype), int(tile_rows), *_kv_args(k_buffer, v_buffer, k_scale, v_scale),
                                        current_stream()))
    return q


def gemm_silu_mul(x, w_interleaved_nk, tile_rows: int = 16):
    """act [M, I] = SiluAndMul(F.linear(x, W)) for bf16 / f16 operands, W's rows interleaved by interleave_gate_up_rows."""
    m, k = x.shape
    n = w_interleaved_nk.shape[0]
    assert x.dtype == w_interleaved_nk.dtype and x.stride(1) == 1 and w_interleaved_nk.stride(1) == 1
    act = torch.empty((m, n // 2), dtype=x.dtype, device=x.device)
    check(lib.sgl_mi355_gemm_silu_mul(ptr(x), x.stride(0), ptr(w_interleaved_nk), w_interleaved_nk.stride(0), ptr(act),
                                      act.stride(0), None, None, m, n, k, dtype_code(x.dtype), dtype_code(x.dtype),
                                      int(tile_rows), current_stream()))
    return act


def qkv_rope_set_kv(x, w_interleaved_nk, bias_interleaved, positions, cos_sin_cache, loc, k_buffer, v_buffer, num_q_heads,
                    num_kv_heads, head_dim, tile_rows: int = 16, k_scale=None, v_scale=None):
    """The bf16 / f16 form of fp8_qkv_rope_set_kv (unquantised qkv_proj): q [M, Hq*D] rotated; k, v -> pool rows ``loc``."""
    m, k = x.shape
    assert x.dtype == w_interleaved_nk.dtype and x.stride(1) == 1 and w_interleaved_nk.stride(1) == 1
    q = torch.empty((m, num_q_heads * head_dim), dtype=x.dtype, device=x.device)
    assert cos_sin_cache.shape[1] == head_dim
    check(lib.sgl_mi355_qkv_rope_set_kv(ptr(x), x.stride(0), ptr(w_interleaved_nk), w_interleaved_nk.stride(0), ptr(q), q.stride(0),
                                        None, None, ptr(bias_interleaved), ptr(positions), ptr(cos_sin_cache), ptr(loc),
                                        ptr(k_buffer), ptr(v_buffer), k_buffer.stride(0), v_buffer.stride(0), m, num_q_heads,
                                        num_kv_heads, head_dim, k, dtype_code(x.dtype), dtype_code(x.dtype), int(tile_rows),
                                        *_kv_args(k_buffer, v_buffer, k_scale, v_scale), current_stream()))
    return q


def awq_gate_up_col_order(n: int, device=None) -> torch.Tensor:
    """Order of the N/8 packed columns (8 weights per int32) of an AWQ gate_up weight for awq_gemm_silu_mul: 16-column tile t =
    (gate columns 8t..8t+7, up columns 8t..8t+7)."""
    c = n // 16
    return torch.arange(2 * c, device=device).view(2, c).t().reshape(-1)


def awq_rope_col_order(num_q_heads: int, num_kv_heads: int, device=None) -> torch.Tensor:
    """Order of the packed columns of an AWQ qkv weight for awq_qkv_rope_set_kv: inside every q / k head (16 packed columns)
    tile u = (columns 8u..8u+7, columns 64+8u..64+8u+7); v heads unchanged."""
    head = torch.arange(16, device=device).view(2, 8).t().reshape(-1)
    nrope = num_q_heads + num_kv_heads
    qk = (torch.arange(nrope, device=device).view(-1, 1) * 16 + head.view(1, -1)).reshape(-1)
    return torch.cat([qk, torch.arange(nrope * 16, (nrope + num_kv_heads) * 16, device=device)])


def awq_permute_cols(order: torch.Tensor, qweight, qzeros, scales, bias=None):
    """Applies a packed-column order to an AWQ weight: qweight [K, N/8], qzeros [K/G, N/8], scales [K/G, N], bias [N]."""
    g, n = scales.shape
    out = (qweight[:, order].contiguous(), qzeros[:, order].contiguous(), scales.view(g, n // 8, 8)[:, order].reshape(g, n).contiguous())
    return out + ((None if bias is None else bias.view(n // 8, 8)[order].reshape(n).contiguous()),)


def awq_gemm_silu_mul(x, qpacked_interleaved, sz_interleaved, group_size: int):
    """act [M, I] = SiluAndMul(awq_gemm(x, W)); W's packed columns in awq_gate_up_col_order before awq_repack."""
    m, k = x.shape
    n = sz_interleaved.shape[1]
    assert x.stride(1) == 1 and qpacked_interleaved.is_contiguous() and sz_interleaved.is_contiguous()
    act = torch.empty((m, n // 2), dtype=x.dtype, device=x.device)
    check(lib.sgl_mi355_awq_gemm_silu_mul(ptr(x), x.stride(0), ptr(qpacked_interleaved), ptr(sz_interleaved), ptr(act), act.stride(0),
                                          m, n, k, int(group_size), dtype_code(x.dtype), current_stream()))
    return act


def awq_qkv_rope_set_kv(x, qpacked_interleaved, sz_interleaved, bias_interleaved, group_size, positions, cos_sin_cache, loc,
                        k_buffer, v_buffer, num_q_heads, num_kv_heads, head_dim, k_scale=None, v_scale=None):
    """The int4 form of qkv_rope_set_kv: q [M, Hq*D] rotated; rotated k and v -> pool rows ``loc``."""
    m, k = x.shape
    assert x.stride(1) == 1 and cos_sin_cache.shape[1] == head_dim
    q = torch.empty((m, num_q_heads * head_dim), dtype=x.dtype, device=x.device)
    check(lib.sgl_mi355_awq_qkv_rope_set_kv(ptr(x), x.stride(0), ptr(qpacked_interleaved), ptr(sz_interleaved), ptr(q), q.stride(0),
                                            ptr(bias_interleaved), ptr(positions), ptr(cos_sin_cache), ptr(loc), ptr(k_buffer),
                                            ptr(v_buffer), k_buffer.stride(0), v_buffer.stride(0), m, num_q_heads, num_kv_heads,
                                            head_dim, k, int(group_size), dtype_code(x.dtype),
                                            *_kv_args(k_buffer, v_buffer, k_scale, v_scale), current_stream()))
    return q


def awq_gemm_slabs(x, qpacked, sz, group_size: int):
    """Raw f32 split-K partial sums [S, M, N] of awq_gemm (no bias): consumed by fused_add_rmsnorm_quant_fp8(slabs=...)."""
    m, k = x.shape
    n = sz.shape[1]
    kr = lib.sgl_mi355_awq_gemm_num_kranges(m, k)
    if kr <= 1:   # one k-range: the kernel would take its plain-epilogue store path and write T-typed outputs into an f32 buffer
        raise RuntimeError(f"awq_gemm_slabs: M={m} K={k} runs as one k-range (no slabs); use awq_gemm")
    out = torch.empty((kr, m, n), dtype=torch.float32, device=x.device)
    check(lib.sgl_mi355_awq_gemm_slabs(ptr(x), x.stride(0), ptr(qpacked), ptr(sz), ptr(out), m, n, k, int(group_size),
                                       dtype_code(x.dtype), current_stream()))
    return out


def balanced_tile_rows(n_rows: int, elem_bytes: int = 1) -> int:
    """16-row tiles, or -- fp8 operands only -- 8-row tiles when 16-row tiles would leave the last round of the persistent
    workgroups mostly idle (the rule of the plain skinny GEMM's launcher): e.g. qkv_proj of Llama-3-8B, 384 tiles on 256 CUs.
    (16-bit operands: an 8-row tile costs a wave the same 16 load instructions as a 16-row tile; measured slower.)"""
    if elem_bytes != 1:
        return 16
    cus = lib.sgl_mi355_device_cu_count(torch.cuda.current_device())
    t16 = n_rows // 16
    rounds = -(-t16 // cus)
    return 8 if (rounds < 4 and t16 % cus != 0 and (t16 % cus) < (3 * cus) // 4 and n_rows % 8 == 0) else 16


class Fp8MlpBlockScratch:
    """Device buffers of ``fp8_mlp_block`` for one (M, hidden, inter) and ``layers`` layers: the hand-off scratch (shared by all
    layers: every launch writes it before it reads it), ONE sync block per layer (all zeroed by ``reset()`` once per step, before the
    first layer's launch) and the output slabs / activation scales (consumed by the next kernel, so shared as well)."""

    def __init__(self, m: int, hidden: int, inter: int, layers: int, device, timeline: bool = False):
        self.m, self.hidden, self.inter, self.layers = m, hidden, inter, layers
        words = int(lib.sgl_mi355_fp8_mlp_block_sync_words())
        self.sync = torch.zeros((layers, words), dtype=torch.int32, device=device)
        self.pmax = torch.zeros(int(lib.sgl_mi355_fp8_mlp_block_pmax_words()), dtype=torch.int32, device=device)
        self.xq = torch.zeros((m, hidden), dtype=torch.uint8, device=device)
        self.xs = torch.zeros(m, dtype=torch.float32, device=device)
        self.actq = torch.zeros((m, inter), dtype=torch.uint8, device=device)
        self.slabs = torch.zeros(((inter + 4095) // 4096, m, hidden), dtype=torch.float32, device=device)
        self.act_scales = torch.zeros(m, dtype=torch.float32, device=device)
        cus = self.pmax.numel() // 32
        self.timeline = torch.zeros((cus, 8, 16), dtype=torch.int64, device=device) if timeline else None

    def reset(self) -> None:
        self.sync.zero_()

    def error_codes(self) -> torch.Tensor:
        """Per layer: 0, or the hand-off (1: normed rows, 2: row maxima, 3: activation) whose wait timed out (device sync)."""
        return self.sync[:, 16].cpu()


def fp8_mlp_block_pack_weights(w_gate_up_interleaved_nk: torch.Tensor, w_down_nk: torch.Tensor):
    """Copies the two weights of one layer into ONE allocation and returns the two views: ``fp8_mlp_block`` reads both through a
    single buffer descriptor, so they must lie within one 4 GiB window (two separate torch allocations can be anywhere)."""
    assert w_gate_up_interleaved_nk.element_size() == 1 and w_down_nk.element_size() == 1
    n1, n2 = w_gate_up_interleaved_nk.numel(), w_down_nk.numel()
    pad = (-n1) % 256
    buf = torch.empty(n1 + pad + n2, dtype=torch.uint8, device=w_gate_up_interleaved_nk.device)
    g = buf[:n1].view(w_gate_up_interleaved_nk.shape)
    d = buf[n1 + pad:].view(w_down_nk.shape)
    g.copy_(w_gate_up_interleaved_nk.contiguous().view(torch.uint8))
    d.copy_(w_down_nk.contiguous().view(torch.uint8))
    return g.view(w_gate_up_interleaved_nk.dtype), d.view(w_down_nk.dtype)


def fp8_mlp_block_supported(m: int, hidden: int, inter: int) -> bool:
    return bool(lib.sgl_mi355_fp8_mlp_block_supported(int(m), int(hidden), int(inter)))


def fp8_mlp_block(x: torch.Tensor, residual: torch.Tensor, ln_weight: torch.Tensor, eps: float, w_gate_up_interleaved_nk: torch.Tensor,
                  scales_gate_up_interleaved: torch.Tensor, w_down_nk: torch.Tensor, scratch: Fp8MlpBlockScratch, layer: int):
    """post_attention_layernorm (fused add + RMSNorm) -> per-token fp8 quant -> gate_up_proj + SiluAndMul -> per-token fp8 quant ->
    down_proj in ONE persistent launch (M <= 32; csrc/mlp_block.hip).  ``residual`` is updated in place.  Returns
    (slabs f32 [S, M, hidden] raw down_proj partial sums, act_scales [M]): the consumer (``fused_add_rmsnorm_quant_fp8`` with
    ``slabs=``, ``slab_sx=act_scales``, ``slab_sw=`` the down_proj weight scale) finishes the GEMM.  ``scratch.reset()`` must have
    run since the last launch with the same ``layer``."""
    m, h = x.shape
    inter = w_down_nk.shape[1]
    assert (m, h, inter) == (scratch.m, scratch.hidden, scratch.inter) and 0 <= layer < scratch.layers
    assert x.is_contiguous() and residual.is_contiguous() and w_gate_up_interleaved_nk.is_contiguous() and w_down_nk.is_contiguous()
    assert w_gate_up_interleaved_nk.shape == (2 * inter, h) and w_down_nk.shape == (h, inter)
    assert w_gate_up_interleaved_nk.element_size() == 1 and w_down_nk.element_size() == 1
    assert scales_gate_up_interleaved.dtype == torch.float32 and scales_gate_up_interleaved.numel() == 2 * inter
    check(lib.sgl_mi355_fp8_mlp_block(ptr(x), ptr(residual), ptr(ln_weight), float(eps), ptr(w_gate_up_interleaved_nk),
                                      ptr(scales_gate_up_interleaved), ptr(w_down_nk), ptr(scratch.slabs), ptr(scratch.act_scales),
                                      ptr(scratch.xq), ptr(scratch.xs), ptr(scratch.actq), ptr(scratch.pmax),
                                      scratch.sync[layer].data_ptr(), m, h, inter, dtype_code(x.dtype), ptr(scratch.timeline),
                                      current_stream()))
    return scratch.slabs, scratch.act_scales
