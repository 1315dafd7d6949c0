"""HipAttnBackend: the MI355X attention backend behind the reference's AttentionBackend plugin surface.

Metadata semantics are those of the reference's GPU backend (TritonAttnBackend,
python/sglang/srt/layers/attention/triton_backend.py:40-336,640-732): per batch it builds
  decode : kv_indptr = cumsum(seq_lens), kv_indices = req_to_token rows, num_kv_splits, attn_logits / attn_lse scratch
  extend : kv_indptr / kv_indices over the cached PREFIX, qo_indptr = cumsum(extend_seq_lens), max_extend_len
and every forward writes the new K/V into the pool at out_cache_loc before attending (:647-650,706-709).
All device work is hand-written HIP behind the C-ABI; the object reads exactly the model_runner fields listed in
SURVEY.md 8b and is HIP-graph capturable (fixed-address metadata buffers, fill value 1: :338-630).
"""
from dataclasses import dataclass
from typing import Optional

import torch

from .... import sgl_kernel as K
from ...._cabi import lib
from .base_attn_backend import AttentionBackend


@dataclass
class ForwardMetadata:
    attn_logits: Optional[torch.Tensor]
    attn_lse: Optional[torch.Tensor]
    max_extend_len: Optional[int]
    num_kv_splits: Optional[torch.Tensor]
    kv_indptr: torch.Tensor
    kv_indices: torch.Tensor
    qo_indptr: Optional[torch.Tensor]
    custom_mask: Optional[torch.Tensor] = None
    mask_indptr: Optional[torch.Tensor] = None
    # sliding-window layers attend over the last min(len, W + 1) slots only (triton_backend.py:23-37,927-955)
    window_kv_indptr: Optional[torch.Tensor] = None
    window_kv_indices: Optional[torch.Tensor] = None
    window_num_kv_splits: Optional[torch.Tensor] = None
    # cascade shared-prefix decode: every request's first len(cascade_prefix_indices) slots are these (one radix node);
    # kv_indptr / kv_indices / num_kv_splits then describe the private suffixes only
    cascade_prefix_indices: Optional[torch.Tensor] = None
    cascade_prefix_splits: int = 0
    # sorted (request, split) unit list of the decode launch (kv_split_rule 3; sgl_kernel.decode_schedule)
    sched: Optional[torch.Tensor] = None


def _rows(t):
    """[T, H, D] view whose heads are contiguous (a qkv split view qualifies); copy only if it is not."""
    return t if (t.stride(2) == 1 and t.stride(1) == t.shape[2]) else t.contiguous()


def default_max_kv_splits() -> int:
    """The reference forces 16 on HIP (server_args.py:454-456; 8 elsewhere, :238)."""
    return 16


class HipAttnBackend(AttentionBackend):
    def __init__(self, model_runner, skip_prefill: bool = False, kv_indptr_buf: Optional[torch.Tensor] = None):
        super().__init__()
        self.device = model_runner.device
        max_bs = model_runner.req_to_token_pool.size
        self.req_to_token = model_runner.req_to_token_pool.req_to_token
        self.kv_indptr = (
            torch.zeros((max_bs + 1,), dtype=torch.int32, device=self.device) if kv_indptr_buf is None else kv_indptr_buf
        )
        self.skip_prefill = skip_prefill
        if not skip_prefill:
            self.qo_indptr = torch.zeros((max_bs + 1,), dtype=torch.int32, device=self.device)
        self.sliding_window_size = getattr(model_runner, "sliding_window_size", None)
        self.window_kv_indptr = torch.zeros_like(self.kv_indptr) if self._has_window() else None   # triton_backend.py:81-89
        cfg = model_runner.model_config
        tp = getattr(model_runner, "attention_tp_size", 1)
        self.num_head = cfg.num_attention_heads // tp
        self.num_kv_head = cfg.get_num_kv_heads(tp)
        args = getattr(model_runner, "server_args", None)
        self.num_draft_tokens = getattr(args, "speculative_num_draft_tokens", None)
        if not skip_prefill:
            self.mask_indptr = torch.zeros((max_bs + 1,), dtype=torch.int64, device=self.device)   # triton_backend.py:97-99
        self.max_kv_splits = getattr(args, "triton_attention_num_kv_splits", None) or default_max_kv_splits()
        # 0 = reference heuristic, 1 = static, 2 = MI355X balance rule (split counts are not parity-critical), 3 (default) = rule 2's
        # splits as a unit list sorted longest first, ragged batches cut into smaller units (where it does not apply -- window
        # layers, cascade, head dims outside {64, 128}, more than 4096 units -- rule 2)
        self.static_kv_splits = 1 if getattr(args, "static_kv_splits", False) else int(getattr(args, "kv_split_rule", 3))
        self.kv_split_rule = self.static_kv_splits
        if self.static_kv_splits == 3:
            self.static_kv_splits = 2
        self.kv_sched_rounds_pct = int(getattr(args, "kv_sched_rounds_pct", 150))
        self.v_head_dim = model_runner.token_to_kv_pool.get_value_buffer(0).shape[-1]
        self.qk_head_dim = model_runner.token_to_kv_pool.get_key_buffer(0).shape[-1]
        self.max_context_len = cfg.context_len
        gpu_id = getattr(model_runner, "gpu_id", 0)
        self.device_core_count = lib.sgl_mi355_device_cu_count(int(gpu_id))  # 256 on MI355X
        self.forward_metadata: Optional[ForwardMetadata] = None
        self.merge_in_launch = True   # forward_decode: stage 2 by the last workgroup of each request (False = two kernels)
        self.cascade_shared_prefix_len = 0   # > 0: the graph hooks build cascade (shared-prefix) decode metadata
        self._graph = None
        # counters of the in-launch stage-2 merge (one per request slot; allocated eagerly so graph capture never allocates)
        self.max_bs_hint = int(model_runner.req_to_token_pool.size)
        self._merge_counters = torch.zeros(self.max_bs_hint, dtype=torch.int32, device=self.device)

    # ------------------------------------------------------------------ metadata
    def _has_window(self) -> bool:
        return self.sliding_window_size is not None and self.sliding_window_size > 0

    def _window_metadata(self, bs, req_pool_indices, lens, want_splits, kv_indices=None, num_kv_splits=None):
        """update_sliding_window_buffer (triton_backend.py:927-983): the last min(len, W + 1) slots of every request."""
        window_lens = torch.clamp(lens, max=self.sliding_window_size + 1)
        indptr = self.window_kv_indptr[: bs + 1]
        if want_splits and num_kv_splits is None:
            num_kv_splits = torch.empty((bs,), dtype=torch.int32, device=self.device)
        K.decode_metadata(indptr, num_kv_splits if want_splits else None, window_lens, 1, self.num_head, self.num_kv_head,
                          self.max_kv_splits, self.device_core_count, self.static_kv_splits)
        if kv_indices is None:
            kv_indices = torch.empty(bs * (self.sliding_window_size + 1), dtype=torch.int32, device=self.device)
        K.create_kv_indices(self.req_to_token, req_pool_indices, window_lens, indptr, lens - window_lens, kv_indices)
        return indptr, kv_indices, (num_kv_splits if want_splits else None)

    def _use_schedule(self) -> bool:
        qk = self.qk_head_dim
        return (self.kv_split_rule == 3 and self.merge_in_launch and not self._has_window()
                and self.v_head_dim in (64, 128) and qk == self.v_head_dim and self.num_head * self.v_head_dim <= 16384)

    def _decode_metadata(self, bs, req_pool_indices, seq_lens, seq_lens_sum, kv_indices=None, scratch=None, window_bufs=None,
                         sched_buf=None):
        kv_indptr = self.kv_indptr[: bs + 1]
        if scratch is None:
            num_kv_splits = torch.empty((bs,), dtype=torch.int32, device=self.device)
            attn_logits = torch.empty((bs, self.num_head, self.max_kv_splits, self.v_head_dim), dtype=torch.float32, device=self.device)
            attn_lse = torch.empty((bs, self.num_head, self.max_kv_splits), dtype=torch.float32, device=self.device)
        else:
            num_kv_splits, attn_logits, attn_lse = scratch
        # the in-launch merge leaves its per-request tickets at zero; re-zeroing them once per step makes that self-healing
        # (an aborted launch or a foreign write would otherwise disable the merge of that request for good)
        if self._merge_counters is not None:
            self._merge_counters.zero_()
        sched = None
        units = K.decode_schedule_units(bs, self.num_head, self.num_kv_head, self.kv_sched_rounds_pct) if self._use_schedule() else 0
        if units > 0:
            # one launch: kv_indptr, the split counts and the (request, split) units sorted longest first
            words = 4 + 4 * units
            sched = sched_buf[:words] if sched_buf is not None else torch.empty((words,), dtype=torch.int32, device=self.device)
            K.decode_schedule(kv_indptr, num_kv_splits, sched, seq_lens, self.num_head, self.num_kv_head, self.max_kv_splits,
                              self.kv_sched_rounds_pct)
        else:
            # one launch: kv_indptr[1:bs+1] = cumsum(seq_lens) and the per-request split heuristic
            K.decode_metadata(kv_indptr, num_kv_splits, seq_lens, 1, self.num_head, self.num_kv_head, self.max_kv_splits,
                              self.device_core_count, self.static_kv_splits)
        if kv_indices is None:
            kv_indices = torch.empty(seq_lens_sum, dtype=torch.int32, device=self.device)
        K.create_kv_indices(self.req_to_token, req_pool_indices, seq_lens, kv_indptr, None, kv_indices)
        md = ForwardMetadata(attn_logits, attn_lse, None, num_kv_splits, kv_indptr, kv_indices, None, sched=sched)
        if self._has_window():
            wi, ws = window_bufs if window_bufs is not None else (None, None)
            md.window_kv_indptr, md.window_kv_indices, md.window_num_kv_splits = self._window_metadata(
                bs, req_pool_indices, seq_lens, True, wi, ws)
        return md

    def _cascade_prefix_splits(self, bs: int, shared_prefix_len: int) -> int:
        """Splits of the shared prefix: the prefix pass (the extend kernel over the batch's decode queries) runs one workgroup
        per (kv head, 64 (head, request) pairs, split); enough splits for about one workgroup per CU, at least 128 rows each,
        at most half of the split slots (the private parts need the rest)."""
        group = self.num_head // self.num_kv_head
        slots = 1
        while slots < group and slots < 4:
            slots *= 2
        per_split = self.num_kv_head * ((group + 3) // 4) * ((bs + 64 // slots - 1) // (64 // slots))
        return max(1, min(self.device_core_count // max(per_split, 1), shared_prefix_len // 128, self.max_kv_splits // 2))

    def _cascade_ok(self, bs: int, shared_prefix_len: int) -> bool:
        return shared_prefix_len >= 64 and bs >= 2 and self.v_head_dim in (64, 128)

    def _cascade_metadata(self, bs, req_pool_indices, seq_lens, seq_lens_sum, shared_prefix_len, prefix_splits, kv_indices=None,
                          scratch=None, prefix_buf=None):
        suffix_lens = seq_lens - shared_prefix_len
        kv_indptr = self.kv_indptr[: bs + 1]
        if scratch is None:
            num_kv_splits = torch.empty((bs,), dtype=torch.int32, device=self.device)
            attn_logits = torch.empty((bs, self.num_head, self.max_kv_splits, self.v_head_dim), dtype=torch.float32, device=self.device)
            attn_lse = torch.empty((bs, self.num_head, self.max_kv_splits), dtype=torch.float32, device=self.device)
        else:
            num_kv_splits, attn_logits, attn_lse = scratch
        if self._merge_counters is not None:
            self._merge_counters.zero_()
        K.decode_metadata(kv_indptr, num_kv_splits, suffix_lens, 1, self.num_head, self.num_kv_head,
                          self.max_kv_splits - prefix_splits, self.device_core_count, self.static_kv_splits)
        if kv_indices is None:
            kv_indices = torch.empty(max(int(seq_lens_sum) - bs * shared_prefix_len, 1), dtype=torch.int32, device=self.device)
        K.create_kv_indices(self.req_to_token, req_pool_indices, suffix_lens, kv_indptr,
                            torch.full_like(suffix_lens, shared_prefix_len), kv_indices)
        # (index_select with a one-element device index: `req_to_token[req_pool_indices[0], ...]` resolves the 0-dim CUDA index on
        # the host -- a device sync in every cascade step and graph replay)
        row0 = self.req_to_token.index_select(0, req_pool_indices[:1].long())[0, :shared_prefix_len]
        if prefix_buf is None:
            prefix = row0.contiguous()
        else:   # HIP graph: the kernel arguments hold this buffer's address
            prefix = prefix_buf[:shared_prefix_len]
            prefix.copy_(row0)
        return ForwardMetadata(attn_logits, attn_lse, None, num_kv_splits, kv_indptr, kv_indices, None,
                               cascade_prefix_indices=prefix, cascade_prefix_splits=int(prefix_splits))

    def init_forward_metadata_cascade(self, forward_batch, shared_prefix_len: int, prefix_splits: Optional[int] = None):
        """Decode metadata for a batch whose requests all share their first ``shared_prefix_len`` KV slots (they matched the
        same RadixCache node, radix_cache.py:370-412; the scheduler knows the length from match_prefix).  The shared rows are
        then streamed once per kv head for the whole batch instead of once per request.  Falls back to the
        ordinary metadata when the prefix is too short to pay for the second launch.  For the HIP-graph hooks set
        ``cascade_shared_prefix_len`` before capture / replay instead (the length is baked into the captured launches)."""
        bs = forward_batch.batch_size
        if not forward_batch.forward_mode.is_decode() or not self._cascade_ok(bs, shared_prefix_len):
            return self.init_forward_metadata(forward_batch)
        lens_cpu = getattr(forward_batch, "seq_lens_cpu", None)
        if lens_cpu is not None and len(lens_cpu) and min(int(x) for x in lens_cpu) <= shared_prefix_len:
            # a request that ends inside the "shared" prefix would hand the suffix pass a negative length (checked on host-side
            # lengths only: no device sync; that the slots really are shared is the caller's contract -- match_prefix)
            raise ValueError(f"cascade decode: every request must be longer than the shared prefix ({shared_prefix_len}); "
                             f"shortest is {min(int(x) for x in lens_cpu)}")
        if prefix_splits is None:
            prefix_splits = self._cascade_prefix_splits(bs, shared_prefix_len)
        self.forward_metadata = self._cascade_metadata(bs, forward_batch.req_pool_indices, forward_batch.seq_lens,
                                                       forward_batch.seq_lens_sum, shared_prefix_len, prefix_splits)

    def _target_verify_metadata(self, forward_batch):
        """Speculative-decoding verification (triton_backend.py:224-258): every request extends by num_draft_tokens over its
        whole cached sequence under the tree mask spec_info.custom_mask; only the attention inputs are built here (the
        draft / verify control flow around it is outside this build)."""
        bs = len(forward_batch.req_pool_indices)
        n = int(self.num_draft_tokens)
        spec = forward_batch.spec_info
        qo_indptr = torch.arange(0, (1 + bs) * n, step=n, dtype=torch.int32, device=self.device)
        kv_indptr = self.kv_indptr[: bs + 1]
        K.decode_metadata(kv_indptr, None, forward_batch.seq_lens, 1, self.num_head, self.num_kv_head, self.max_kv_splits,
                          self.device_core_count)
        kv_indices = torch.empty(int(forward_batch.seq_lens_sum), dtype=torch.int32, device=self.device)
        K.create_kv_indices(self.req_to_token, forward_batch.req_pool_indices, forward_batch.seq_lens, kv_indptr, None, kv_indices)
        mask_indptr = self.mask_indptr[: bs + 1]
        mask_indptr[1:] = torch.cumsum(n * (forward_batch.seq_lens[:bs] + n), dim=0)
        return ForwardMetadata(None, None, n, None, kv_indptr, kv_indices, qo_indptr, spec.custom_mask, mask_indptr)

    def init_forward_metadata(self, forward_batch):
        bs = forward_batch.batch_size
        if forward_batch.forward_mode.is_target_verify():
            self.forward_metadata = self._target_verify_metadata(forward_batch)
            return
        if forward_batch.spec_info is not None:
            raise RuntimeError("HipAttnBackend: draft-model speculative modes are outside this build's hot path")
        if forward_batch.forward_mode.is_decode_or_idle():
            self.forward_metadata = self._decode_metadata(bs, forward_batch.req_pool_indices, forward_batch.seq_lens,
                                                          forward_batch.seq_lens_sum)
            return
        # extend: indices over the cached prefix only; the new tokens are read from k/v directly
        kv_indptr = self.kv_indptr[: bs + 1]
        K.decode_metadata(kv_indptr, None, forward_batch.extend_prefix_lens, 1, self.num_head, self.num_kv_head,
                          self.max_kv_splits, self.device_core_count)
        pre_cpu = forward_batch.extend_prefix_lens_cpu
        prefix_sum = int(sum(pre_cpu)) if pre_cpu is not None else int(forward_batch.extend_prefix_lens.sum().item())
        kv_indices = torch.empty(prefix_sum, dtype=torch.int32, device=self.device)
        K.create_kv_indices(self.req_to_token, forward_batch.req_pool_indices, forward_batch.extend_prefix_lens, kv_indptr,
                            None, kv_indices)
        qo_indptr = self.qo_indptr[: bs + 1]
        K.decode_metadata(qo_indptr, None, forward_batch.extend_seq_lens, 1, self.num_head, self.num_kv_head,
                          self.max_kv_splits, self.device_core_count)
        ext_cpu = forward_batch.extend_seq_lens_cpu
        max_extend_len = int(max(ext_cpu)) if ext_cpu is not None else int(forward_batch.extend_seq_lens.max().item())
        self.forward_metadata = ForwardMetadata(None, None, max_extend_len, None, kv_indptr, kv_indices, qo_indptr)
        if pre_cpu is not None and ext_cpu is not None and bs > 0:
            # what the extend kernel's launcher cannot see in its arguments: how many keys a query block walks on average
            # (prefix + half the causal triangle) -- it picks 4 or 8 waves per workgroup by it
            lib.sgl_mi355_extend_attention_set_kv_hint(int(sum(p + e // 2 for p, e in zip(pre_cpu, ext_cpu)) // bs))
        if self._has_window():
            md = self.forward_metadata
            md.window_kv_indptr, md.window_kv_indices, _ = self._window_metadata(
                bs, forward_batch.req_pool_indices, forward_batch.extend_prefix_lens, False)

    # ------------------------------------------------------------------ HIP-graph hooks
    def init_cuda_graph_state(self, max_bs: int, max_num_tokens: int, kv_indices_buf: Optional[torch.Tensor] = None):
        self._graph = dict(
            kv_indices=(torch.zeros((max_num_tokens * self.max_context_len,), dtype=torch.int32, device=self.device)
                        if kv_indices_buf is None else kv_indices_buf),
            attn_logits=torch.zeros((max_num_tokens, self.num_head, self.max_kv_splits, self.v_head_dim), dtype=torch.float32, device=self.device),
            attn_lse=torch.zeros((max_num_tokens, self.num_head, self.max_kv_splits), dtype=torch.float32, device=self.device),
            num_kv_splits=torch.full((max_num_tokens,), self.max_kv_splits, dtype=torch.int32, device=self.device),
        )
        self._graph["cascade_prefix"] = torch.zeros((self.max_context_len,), dtype=torch.int32, device=self.device)
        self._graph["sched"] = torch.zeros((4 + 4 * 4096,), dtype=torch.int32, device=self.device)   # (the capacity's upper bound)
        if self._has_window():   # triton_backend.py:371-392
            self._graph["window_kv_indices"] = torch.zeros((max_num_tokens * (self.sliding_window_size + 1),), dtype=torch.int32, device=self.device)
            self._graph["window_num_kv_splits"] = torch.full((max_num_tokens,), self.max_kv_splits, dtype=torch.int32, device=self.device)

    def init_forward_metadata_capture_cuda_graph(self, bs, num_tokens, req_pool_indices, seq_lens, encoder_lens,
                                                 forward_mode, spec_info):
        if not forward_mode.is_decode_or_idle() or spec_info is not None:
            raise ValueError(f"Invalid forward mode: {forward_mode=} for HIP graph capture.")
        g = self._graph
        scratch = (g["num_kv_splits"][:bs], g["attn_logits"][:bs], g["attn_lse"][:bs])
        if self.cascade_shared_prefix_len > 0 and self._cascade_ok(bs, self.cascade_shared_prefix_len):
            p_len = self.cascade_shared_prefix_len   # (the capture buffers hold seq_lens = p_len + 1)
            self.forward_metadata = self._cascade_metadata(bs, req_pool_indices, seq_lens, 0, p_len, self._cascade_prefix_splits(bs, p_len),
                                                           g["kv_indices"], scratch, g["cascade_prefix"])
            return
        wb = (g["window_kv_indices"], g["window_num_kv_splits"][:bs]) if self._has_window() else None
        self.forward_metadata = self._decode_metadata(bs, req_pool_indices, seq_lens, 0, g["kv_indices"], scratch, wb, g["sched"])

    def init_forward_metadata_replay_cuda_graph(self, bs, req_pool_indices, seq_lens, seq_lens_sum, encoder_lens,
                                                forward_mode, spec_info, seq_lens_cpu):
        if not forward_mode.is_decode_or_idle() or spec_info is not None:
            raise ValueError(f"Invalid forward mode: {forward_mode=} for HIP graph replay.")
        g = self._graph
        scratch = (g["num_kv_splits"][:bs], g["attn_logits"][:bs], g["attn_lse"][:bs])
        if self.cascade_shared_prefix_len > 0 and self._cascade_ok(bs, self.cascade_shared_prefix_len):
            p_len = self.cascade_shared_prefix_len
            self.forward_metadata = self._cascade_metadata(bs, req_pool_indices[:bs], seq_lens[:bs], seq_lens_sum, p_len,
                                                           self._cascade_prefix_splits(bs, p_len), g["kv_indices"], scratch,
                                                           g["cascade_prefix"])
            return
        wb = (g["window_kv_indices"], g["window_num_kv_splits"][:bs]) if self._has_window() else None
        self.forward_metadata = self._decode_metadata(bs, req_pool_indices[:bs], seq_lens[:bs], seq_lens_sum,
                                                      g["kv_indices"], scratch, wb, g["sched"])

    def get_cuda_graph_seq_len_fill_value(self):
        return 1

    # ------------------------------------------------------------------ forwards
    def _layer_kv(self, layer):
        """(kv_indptr, kv_indices, num_kv_splits, window) a layer attends over: the window buffers for a sliding-window layer
        (triton_backend.py:655-666,713-719), the full ones otherwise.  A window layer's decode uses the split counts computed
        for the window lengths (the reference passes the full-length counts; split counts do not change the result)."""
        md = self.forward_metadata
        if layer.sliding_window_size is not None and layer.sliding_window_size > -1:
            if md.window_kv_indptr is None:
                raise RuntimeError("HipAttnBackend: a sliding-window layer needs model_runner.sliding_window_size to be set")
            return md.window_kv_indptr, md.window_kv_indices, md.window_num_kv_splits, int(layer.sliding_window_size)
        return md.kv_indptr, md.kv_indices, md.num_kv_splits, -1

    def forward_extend(self, q, k, v, layer, forward_batch, save_kv_cache=True):
        kv_indptr, kv_indices, _, window = self._layer_kv(layer)
        o = q.new_empty((q.shape[0], layer.tp_q_head_num * layer.v_head_dim))
        if save_kv_cache:
            forward_batch.token_to_kv_pool.set_kv_buffer(layer, forward_batch.out_cache_loc, k, v, layer.k_scale, layer.v_scale)
        md = self.forward_metadata
        causal = not (layer.is_cross_attention or getattr(layer.attn_type, "value", "decoder") == "encoder_only")
        # no cached prefix anywhere in the batch (the size of kv_indices is host knowledge): nothing is read from the pool, so the
        # kernel need not know its dtype -- an fp8 pool would otherwise select the fp8-prefix instantiation for rows that do not exist
        no_prefix = kv_indices.numel() == 0
        K.extend_attention_fwd(
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), _rows(k), _rows(v),
            o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            None if no_prefix else forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id),
            None if no_prefix else forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id),
            md.qo_indptr, kv_indptr, kv_indices, md.custom_mask, causal, md.mask_indptr, md.max_extend_len, layer.scaling,
            layer.logit_cap, sliding_window_size=window, k_scale=layer.k_scale_float or 1.0, v_scale=layer.v_scale_float or 1.0,
        )
        return o

    def forward_decode(self, q, k, v, layer, forward_batch, save_kv_cache=True):
        kv_indptr, kv_indices, num_kv_splits, _ = self._layer_kv(layer)
        q = q.reshape(-1, layer.tp_q_head_num * layer.qk_head_dim)
        if save_kv_cache:  # decode reads the new token from the pool, so this must precede the attention launch
            forward_batch.token_to_kv_pool.set_kv_buffer(layer, forward_batch.out_cache_loc, k, v, layer.k_scale, layer.v_scale)
        md = self.forward_metadata
        if md.cascade_prefix_indices is not None:
            return self._forward_decode_cascade(q, layer, forward_batch, want_o=True, want_quant=False)[0]
        if self.merge_in_launch and layer.qk_head_dim == layer.v_head_dim and layer.v_head_dim in (64, 128) \
                and layer.tp_q_head_num * layer.v_head_dim <= 16384:
            # stage 2 inside the stage-1 launch (the last workgroup of each request merges its splits): one launch instead of
            # two, bit-identical to the two-kernel sequence (tests/test_decode_attention_gpu.py)
            return K.decode_attention_merge_quant(
                q.view(-1, layer.tp_q_head_num, layer.qk_head_dim),
                forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id),
                forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id),
                kv_indptr, kv_indices, md.attn_logits, md.attn_lse, num_kv_splits, self.max_kv_splits, layer.scaling,
                self._merge_counter_buf(q), layer.logit_cap, layer.k_scale_float or 1.0, layer.v_scale_float or 1.0,
                want_o=True, want_quant=False, sched=md.sched)[0]
        if md.sched is not None:
            raise RuntimeError("HipAttnBackend: the sorted unit list (kv_split_rule 3) needs the in-launch merge path")
        o = q.new_empty((q.shape[0], layer.tp_q_head_num * layer.v_head_dim))
        K.decode_attention_fwd(
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim),
            forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id),
            forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id),
            o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            kv_indptr, kv_indices, md.attn_logits, md.attn_lse, num_kv_splits, self.max_kv_splits,
            layer.scaling, layer.logit_cap, layer.k_scale_float or 1.0, layer.v_scale_float or 1.0,
        )
        return o

    def _merge_counter_buf(self, q):
        if self._merge_counters is None or self._merge_counters.device != q.device or self._merge_counters.numel() < q.shape[0]:
            self._merge_counters = torch.zeros(max(self.max_bs_hint, q.shape[0]), dtype=torch.int32, device=q.device)
        return self._merge_counters

    def _forward_decode_cascade(self, q, layer, forward_batch, want_o, want_quant):
        md = self.forward_metadata
        if layer.sliding_window_size is not None and layer.sliding_window_size > -1:
            raise RuntimeError("HipAttnBackend: cascade decode metadata cannot serve a sliding-window layer")
        return K.decode_attention_cascade(
            q.reshape(-1, layer.tp_q_head_num, layer.qk_head_dim),
            forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id),
            forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id),
            md.cascade_prefix_indices, md.cascade_prefix_splits, md.kv_indptr, md.kv_indices, md.attn_logits, md.attn_lse,
            md.num_kv_splits, self.max_kv_splits, layer.scaling, self._merge_counter_buf(q), layer.logit_cap,
            layer.k_scale_float or 1.0, layer.v_scale_float or 1.0, want_o=want_o, want_quant=want_quant)

    def forward_decode_partial(self, q, layer, forward_batch):
        """Stage 1 only (the caller has already written K/V and will merge the split partials itself, e.g. fused with
        the next op's quantisation): returns the ForwardMetadata holding attn_logits / attn_lse / num_kv_splits."""
        kv_indptr, kv_indices, num_kv_splits, _ = self._layer_kv(layer)
        md = self.forward_metadata
        if md.sched is not None:
            raise RuntimeError("HipAttnBackend: the sorted unit list (kv_split_rule 3) has no stage-1-only form")
        K.decode_attention_fwd(
            q.reshape(-1, layer.tp_q_head_num, layer.qk_head_dim),
            forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id),
            forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id),
            None, kv_indptr, kv_indices, md.attn_logits, md.attn_lse, num_kv_splits, self.max_kv_splits,
            layer.scaling, layer.logit_cap, layer.k_scale_float or 1.0, layer.v_scale_float or 1.0,
        )
        return md

    def forward_decode_merged_quant(self, q, layer, forward_batch, want_o=False):
        """Stage 1 + in-launch stage 2 + per-token fp8 quantisation of the merged rows (the attention -> o_proj hand-off of the
        w8a8 decode step in one kernel).  Returns (o or None, o_q, o_scale)."""
        kv_indptr, kv_indices, num_kv_splits, _ = self._layer_kv(layer)
        md = self.forward_metadata
        if md.cascade_prefix_indices is not None:
            return self._forward_decode_cascade(q, layer, forward_batch, want_o=want_o, want_quant=True)
        self._merge_counter_buf(q)
        return K.decode_attention_merge_quant(
            q.reshape(-1, layer.tp_q_head_num, layer.qk_head_dim),
            forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id),
            forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id),
            kv_indptr, kv_indices, md.attn_logits, md.attn_lse, num_kv_splits, self.max_kv_splits, layer.scaling,
            self._merge_counters, layer.logit_cap, layer.k_scale_float or 1.0, layer.v_scale_float or 1.0, want_o=want_o,
            sched=md.sched)

    def support_triton(self):
        return False  # the host helpers use this build's HIP index kernels, never Triton
