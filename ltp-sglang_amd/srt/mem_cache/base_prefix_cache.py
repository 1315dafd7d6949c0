"""Prefix-cache interface (python/sglang/srt/mem_cache/base_prefix_cache.py:12-108)."""
from abc import ABC, abstractmethod
from typing import Any, List, NamedTuple

import torch


class MatchResult(NamedTuple):
    device_indices: torch.Tensor   # KV slots of the matched prefix (int64)
    last_device_node: Any          # deepest matched node
    last_host_node: Any            # == last_device_node (no host tier in this build)
    host_hit_length: int = 0


class BasePrefixCache(ABC):
    @abstractmethod
    def reset(self): ...

    @abstractmethod
    def match_prefix(self, key: List[int], **kwargs) -> MatchResult: ...

    @abstractmethod
    def cache_finished_req(self, req, **kwargs): ...

    @abstractmethod
    def cache_unfinished_req(self, req, **kwargs): ...

    @abstractmethod
    def evict(self, num_tokens: int): ...

    @abstractmethod
    def inc_lock_ref(self, node: Any): ...

    @abstractmethod
    def dec_lock_ref(self, node: Any): ...

    def evictable_size(self):
        return 0

    def protected_size(self):
        return 0

    def total_size(self):
        raise NotImplementedError()

    def take_events(self):
        return []
