"""Attention oracle: CPU restatement of the reference's torch-native backend.

Follows python/sglang/srt/layers/attention/torch_native_backend.py:
  * decode  -> _run_sdpa_forward_decode  (:112-180): per request, one query row against the
    K/V rows gathered through req_to_token[req_pool_idx, :seq_len], non-causal SDPA,
    GQA by contiguous head groups (enable_gqa);
  * extend  -> _run_sdpa_forward_extend  (:27-110): per request, the new query rows are
    placed behind ``prefix_len`` dummy rows so that a top-left causal mask of size
    seq_len x seq_len gives row i visibility of keys [0, prefix_len + i]; the dummy rows are
    zeros here (the reference leaves them uninitialised; causal rows are independent).

Two flavours:
  *_sdpa : same dtype policy as the reference (torch SDPA on the input dtype) -- this is
           the parity oracle;
  *_f64  : the same math in float64 with optional logit cap (reference formula
           cap*tanh(x/cap), triton_ops/decode_attention.py:38-41,122-123) -- the tight
           bound used to measure each kernel's own rounding error.

TEST INFRASTRUCTURE: see oracle/__init__.py.
"""
import math

import torch
import torch.nn.functional as F


def _gather_rows(buf, req_to_token, req_pool_idx, n):
    locs = req_to_token[req_pool_idx, :n].long()
    return buf[locs]  # [n, Hkv, D]


def decode_attention_sdpa(q, k_buffer, v_buffer, req_to_token, req_pool_indices, seq_lens, scaling):
    """q [bs, Hq, D] -> o [bs, Hq, Dv]; torch_native_backend.py:112-180."""
    bs, hq, _ = q.shape
    hkv = k_buffer.shape[1]
    out = torch.empty(bs, hq, v_buffer.shape[2], dtype=q.dtype)
    for i in range(bs):
        n = int(seq_lens[i])
        keys = _gather_rows(k_buffer, req_to_token, int(req_pool_indices[i]), n).transpose(0, 1)  # [Hkv, n, D]
        vals = _gather_rows(v_buffer, req_to_token, int(req_pool_indices[i]), n).transpose(0, 1)
        qi = q[i].unsqueeze(1)  # [Hq, 1, D]
        oi = F.scaled_dot_product_attention(
            qi.unsqueeze(0), keys.unsqueeze(0), vals.unsqueeze(0), enable_gqa=(hq != hkv), scale=scaling, is_causal=False
        )
        out[i] = oi[0, :, 0, :]
    return out


def extend_attention_sdpa(
    q, k_buffer, v_buffer, req_to_token, req_pool_indices, seq_lens, extend_prefix_lens, extend_seq_lens, scaling, causal=True
):
    """q [T, Hq, D] (new tokens of all requests, concatenated) -> o [T, Hq, Dv].

    The new tokens' K/V must already be in the pool at req_to_token[.., prefix:seq]
    (the backend writes them first, torch_native_backend.py:196-199).
    """
    t, hq, d = q.shape
    hkv = k_buffer.shape[1]
    out = torch.empty(t, hq, v_buffer.shape[2], dtype=q.dtype)
    pos = 0
    for i in range(len(seq_lens)):
        n, pre, ext = int(seq_lens[i]), int(extend_prefix_lens[i]), int(extend_seq_lens[i])
        keys = _gather_rows(k_buffer, req_to_token, int(req_pool_indices[i]), n).transpose(0, 1)
        vals = _gather_rows(v_buffer, req_to_token, int(req_pool_indices[i]), n).transpose(0, 1)
        padded = torch.zeros(hq, n, d, dtype=q.dtype)
        padded[:, pre:, :] = q[pos : pos + ext].transpose(0, 1)
        oi = F.scaled_dot_product_attention(
            padded.unsqueeze(0), keys.unsqueeze(0), vals.unsqueeze(0), enable_gqa=(hq != hkv), scale=scaling, is_causal=causal
        )
        out[pos : pos + ext] = oi[0, :, pre:, :].transpose(0, 1)
        pos += ext
    return out


def _softcap(x, cap):
    return cap * torch.tanh(x / cap) if cap and cap > 0 else x


def decode_attention_f64(q, k_buffer, v_buffer, req_to_token, req_pool_indices, seq_lens, scaling, logit_cap=0.0):
    bs, hq, _ = q.shape
    hkv = k_buffer.shape[1]
    grp = hq // hkv
    out = torch.empty(bs, hq, v_buffer.shape[2], dtype=torch.float64)
    for i in range(bs):
        n = int(seq_lens[i])
        keys = _gather_rows(k_buffer, req_to_token, int(req_pool_indices[i]), n).double()  # [n, Hkv, D]
        vals = _gather_rows(v_buffer, req_to_token, int(req_pool_indices[i]), n).double()
        keys = keys.repeat_interleave(grp, dim=1)
        vals = vals.repeat_interleave(grp, dim=1)
        s = torch.einsum("hd,nhd->hn", q[i].double(), keys) * scaling
        p = torch.softmax(_softcap(s, logit_cap), dim=-1)
        out[i] = torch.einsum("hn,nhd->hd", p, vals)
    return out


def extend_attention_f64(
    q, k_buffer, v_buffer, req_to_token, req_pool_indices, seq_lens, extend_prefix_lens, extend_seq_lens, scaling,
    causal=True, logit_cap=0.0,
):
    t, hq, _ = q.shape
    hkv = k_buffer.shape[1]
    grp = hq // hkv
    out = torch.empty(t, hq, v_buffer.shape[2], dtype=torch.float64)
    pos = 0
    for i in range(len(seq_lens)):
        n, pre, ext = int(seq_lens[i]), int(extend_prefix_lens[i]), int(extend_seq_lens[i])
        keys = _gather_rows(k_buffer, req_to_token, int(req_pool_indices[i]), n).double().repeat_interleave(grp, dim=1)
        vals = _gather_rows(v_buffer, req_to_token, int(req_pool_indices[i]), n).double().repeat_interleave(grp, dim=1)
        s = torch.einsum("qhd,nhd->hqn", q[pos : pos + ext].double(), keys) * scaling
        s = _softcap(s, logit_cap)
        if causal:
            qpos = pre + torch.arange(ext).view(1, ext, 1)
            kpos = torch.arange(n).view(1, 1, n)
            s = s.masked_fill(kpos > qpos, -math.inf)
        p = torch.softmax(s, dim=-1)
        out[pos : pos + ext] = torch.einsum("hqn,nhd->qhd", p, vals)
        pos += ext
    return out


def extend_attention_masked_f64(q, k_extend, v_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, custom_mask, mask_indptr,
                                scaling, is_causal=True, skip_prefix_custom_mask=True, sliding_window_size=-1, logit_cap=0.0):
    """The Triton extend kernel's visibility rules (extend_attention.py:131-203 prefix phase, :205-284 extend phase) in float64:
    prefix key j (position inside kv_indices) is seen by query row i (position inside the extend part) iff
    [custom_mask[i, j] unless skipped] and [i <= j + W when a window is set]; extend key j iff custom_mask[i, pre + j] when a
    mask is given, else j <= i when causal.  Pinned by tests/golden/extend_mask.npz (the reference kernel on the interpreter)."""
    t, hq, _ = q.shape
    hkv = k_extend.shape[1]
    grp = hq // hkv
    out = torch.zeros(t, hq, v_extend.shape[2], dtype=torch.float64)
    for b in range(len(qo_indptr) - 1):
        q0, q1 = int(qo_indptr[b]), int(qo_indptr[b + 1])
        ext = q1 - q0
        if ext == 0:
            continue
        idx = kv_indices[int(kv_indptr[b]): int(kv_indptr[b + 1])].long()
        pre = idx.numel()
        keys = torch.cat([k_buffer[idx], k_extend[q0:q1]]).double().repeat_interleave(grp, dim=1)   # [pre + ext, Hq, D]
        vals = torch.cat([v_buffer[idx], v_extend[q0:q1]]).double().repeat_interleave(grp, dim=1)
        s = torch.einsum("qhd,nhd->hqn", q[q0:q1].double(), keys) * scaling
        s = _softcap(s, logit_cap)
        qi = torch.arange(ext).view(ext, 1)
        vis = torch.ones(ext, pre + ext, dtype=torch.bool)
        if custom_mask is not None:
            m = custom_mask[int(mask_indptr[b]): int(mask_indptr[b]) + ext * (pre + ext)].view(ext, pre + ext).bool()
            if not skip_prefix_custom_mask:
                vis[:, :pre] &= m[:, :pre]
            vis[:, pre:] &= m[:, pre:]
        elif is_causal:
            vis[:, pre:] &= torch.arange(ext).view(1, ext) <= qi
        if sliding_window_size is not None and sliding_window_size > 0:
            vis[:, :pre] &= qi <= torch.arange(pre).view(1, pre) + sliding_window_size
        s = s.masked_fill(~vis.unsqueeze(0), -math.inf)
        p = torch.softmax(s, dim=-1)
        out[q0:q1] = torch.einsum("hqn,nhd->qhd", torch.nan_to_num(p), vals)
    return out


def merge_state(v_a, s_a, v_b, s_b):
    """merge_state_torch, sgl-kernel/tests/test_merge_state_v2.py:101-135 (the reference's own torch restatement)."""
    p, s = s_a.float().clone(), s_b.float().clone()
    p[p == torch.inf] = -torch.inf
    s[s == torch.inf] = -torch.inf
    m = torch.maximum(p, s)
    pe, se_ = torch.exp(p - m), torch.exp(s - m)
    tot = pe + se_
    out_lse = torch.log(tot) + m
    out = v_a * (pe / tot).unsqueeze(2) + v_b * (se_ / tot).unsqueeze(2)
    return out, out_lse
