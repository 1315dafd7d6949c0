"""ForwardMode / ForwardBatch: the per-batch contract between the scheduler side and the attention backend.

Field names, dtypes and meanings follow python/sglang/srt/model_executor/forward_batch_info.py:68-138
(ForwardMode) and :163-300 (ForwardBatch); only the fields the attention / KV-pool hot path reads are kept
(SURVEY.md 8b "Reads from forward_batch").  Index dtypes: req_pool_indices / seq_lens / out_cache_loc /
positions int64, extend_* int32 (SURVEY.md 8a notes).
"""
from dataclasses import dataclass
from enum import IntEnum, auto
from typing import Any, List, Optional

import torch


class ForwardMode(IntEnum):
    EXTEND = auto()          # prefill / extend a sequence whose prefix may already be cached
    DECODE = auto()          # one new token per request
    MIXED = auto()           # chunked prefill batch holding both
    IDLE = auto()            # nothing to run on this rank
    TARGET_VERIFY = auto()   # speculative decoding (kept for enum-value compatibility; not implemented here)
    DRAFT_EXTEND = auto()
    DUMMY_FIRST = auto()
    SPLIT_PREFILL = auto()

    def is_extend(self):
        return self in (ForwardMode.EXTEND, ForwardMode.MIXED, ForwardMode.DRAFT_EXTEND, ForwardMode.TARGET_VERIFY)

    def is_prefill(self):
        return self.is_extend()

    def is_decode(self):
        return self == ForwardMode.DECODE

    def is_mixed(self):
        return self == ForwardMode.MIXED

    def is_idle(self):
        return self == ForwardMode.IDLE

    def is_decode_or_idle(self):
        return self in (ForwardMode.DECODE, ForwardMode.IDLE)

    def is_target_verify(self):
        return self == ForwardMode.TARGET_VERIFY

    def is_draft_extend(self):
        return self == ForwardMode.DRAFT_EXTEND

    def is_extend_or_draft_extend_or_mixed(self):
        return self in (ForwardMode.EXTEND, ForwardMode.DRAFT_EXTEND, ForwardMode.MIXED)

    def is_cuda_graph(self):
        return self in (ForwardMode.DECODE, ForwardMode.TARGET_VERIFY, ForwardMode.IDLE)

    def is_dummy_first(self):
        return self == ForwardMode.DUMMY_FIRST

    def is_split_prefill(self):
        return self == ForwardMode.SPLIT_PREFILL


@dataclass
class ForwardBatch:
    forward_mode: ForwardMode
    batch_size: int
    input_ids: torch.Tensor
    req_pool_indices: torch.Tensor   # int64 [bs] rows of req_to_token
    seq_lens: torch.Tensor           # int64 [bs] (already includes the token being decoded)
    out_cache_loc: torch.Tensor      # int64 [num new tokens] pool slots the new K/V go to
    seq_lens_sum: int
    seq_lens_cpu: Optional[torch.Tensor] = None
    positions: Optional[torch.Tensor] = None          # int64 [num tokens]
    extend_num_tokens: Optional[int] = None
    extend_seq_lens: Optional[torch.Tensor] = None    # int32 [bs]
    extend_prefix_lens: Optional[torch.Tensor] = None  # int32 [bs]
    extend_start_loc: Optional[torch.Tensor] = None   # int32 [bs]
    extend_prefix_lens_cpu: Optional[List[int]] = None
    extend_seq_lens_cpu: Optional[List[int]] = None
    encoder_lens: Optional[torch.Tensor] = None
    spec_info: Any = None
    req_to_token_pool: Any = None
    token_to_kv_pool: Any = None
    attn_backend: Any = None

    @classmethod
    def init_new(cls, forward_mode, req_pool_indices, seq_lens, out_cache_loc, input_ids, req_to_token_pool,
                 token_to_kv_pool, attn_backend, extend_prefix_lens=None, extend_seq_lens=None, seq_lens_cpu=None):
        """Builds the batch the way ForwardBatch.init_new does for the fields above
        (forward_batch_info.py:302-450): decode positions = clamp(seq_lens - 1, 0) (:958-960); extend positions
        and extend_start_loc from the prefix/extend lengths (:885-955) -- here with the HIP index kernel."""
        from ...sgl_kernel import compute_position

        ret = cls(
            forward_mode=forward_mode, batch_size=len(seq_lens), input_ids=input_ids,
            req_pool_indices=req_pool_indices, seq_lens=seq_lens, out_cache_loc=out_cache_loc,
            seq_lens_sum=int(seq_lens_cpu.sum()) if seq_lens_cpu is not None else int(seq_lens.sum().item()),
            seq_lens_cpu=seq_lens_cpu, req_to_token_pool=req_to_token_pool, token_to_kv_pool=token_to_kv_pool,
            attn_backend=attn_backend,
        )
        if forward_mode.is_decode():
            ret.positions = torch.clamp(seq_lens - 1, min=0).to(torch.int64)
        elif forward_mode.is_extend():
            ret.extend_seq_lens = extend_seq_lens.to(torch.int32)
            ret.extend_prefix_lens = extend_prefix_lens.to(torch.int32)
            ret.extend_num_tokens = int(out_cache_loc.numel())
            ret.positions, ret.extend_start_loc = compute_position(
                ret.extend_prefix_lens, ret.extend_seq_lens, ret.extend_num_tokens
            )
            ret.extend_prefix_lens_cpu = ret.extend_prefix_lens.tolist()
            ret.extend_seq_lens_cpu = ret.extend_seq_lens.tolist()
        return ret
