"""RadixCache: RadixAttention's prefix tree, on the native C++ tree (csrc/radix_tree.hip).

Interface and request bookkeeping follow python/sglang/srt/mem_cache/radix_cache.py:101-348: match_prefix returns the
KV slots of the longest cached prefix (splitting a node on a partial match), cache_finished_req / cache_unfinished_req
insert a request's tokens and free the slots that duplicated an existing prefix, evict releases LRU unlocked leaves to
the allocator, inc/dec_lock_ref pin a request's path.  Node handles are small proxy objects over int64 ids.
"""
import ctypes
from typing import List, Optional

import numpy as np
import torch

from ..._cabi import lib
from .base_prefix_cache import BasePrefixCache, MatchResult


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(ctypes.c_void_p)


class TreeNode:
    """Handle of a native node (identity == id)."""

    __slots__ = ("_tree", "id")

    def __init__(self, tree, node_id: int):
        self._tree, self.id = tree, int(node_id)

    def _info(self):
        parent, lock, nch = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        klen = lib.sgl_mi355_radix_node_info(self._tree, self.id, ctypes.byref(parent), ctypes.byref(lock), ctypes.byref(nch))
        return klen, parent.value, lock.value, nch.value

    @property
    def lock_ref(self) -> int:
        return self._info()[2]

    @property
    def parent(self) -> Optional["TreeNode"]:
        p = self._info()[1]
        return None if p < 0 else TreeNode(self._tree, p)

    def __eq__(self, other):
        return isinstance(other, TreeNode) and other.id == self.id

    def __hash__(self):
        return hash(self.id)

    def __repr__(self):
        return f"TreeNode(id={self.id})"


class RadixCache(BasePrefixCache):
    def __init__(self, req_to_token_pool, token_to_kv_pool_allocator, page_size: int, disable: bool = False,
                 enable_kv_cache_events: bool = False):
        self.req_to_token_pool = req_to_token_pool
        self.token_to_kv_pool_allocator = token_to_kv_pool_allocator
        self.page_size = page_size
        self.disable = disable
        self.device = token_to_kv_pool_allocator.device if token_to_kv_pool_allocator else torch.device("cpu")
        self._tree = ctypes.c_void_p(lib.sgl_mi355_radix_create(int(page_size)))
        if not self._tree:
            raise RuntimeError("sgl_mi355_radix_create failed")
        self.root_node = TreeNode(self._tree, lib.sgl_mi355_radix_root(self._tree))

    def __del__(self):
        tree = getattr(self, "_tree", None)
        if tree:
            lib.sgl_mi355_radix_destroy(tree)

    def reset(self):
        lib.sgl_mi355_radix_reset(self._tree)
        self.root_node = TreeNode(self._tree, lib.sgl_mi355_radix_root(self._tree))

    # ------------------------------------------------------------------ tree primitives
    def match_prefix(self, key: List[int], **kwargs) -> MatchResult:
        if self.disable or len(key) == 0:
            empty = torch.empty((0,), dtype=torch.int64, device=self.device)
            return MatchResult(empty, self.root_node, self.root_node)
        karr, kptr = _i64(key)
        out = np.empty(len(karr), dtype=np.int64)
        last = ctypes.c_int64()
        n = lib.sgl_mi355_radix_match_prefix(self._tree, kptr, len(karr), out.ctypes.data_as(ctypes.c_void_p), len(out),
                                             ctypes.byref(last))
        value = torch.from_numpy(out[:n].copy()).to(self.device)
        node = TreeNode(self._tree, last.value)
        return MatchResult(value, node, node)

    def insert(self, key: List[int], value=None) -> int:
        if self.disable:
            return 0
        if value is None:
            value = list(key)
        karr, kptr = _i64(key)
        varr, vptr = _i64(value.cpu().numpy() if isinstance(value, torch.Tensor) else value)
        assert len(karr) == len(varr)
        return int(lib.sgl_mi355_radix_insert(self._tree, kptr, vptr, len(karr)))

    def evict(self, num_tokens: int):
        if self.disable:
            return
        cap = max(int(lib.sgl_mi355_radix_total_size(self._tree)), 1)
        vals = np.empty(cap, dtype=np.int64)
        lens = np.empty(max(int(lib.sgl_mi355_radix_num_nodes(self._tree)), 1), dtype=np.int64)
        nn = ctypes.c_int64()
        total = lib.sgl_mi355_radix_evict(self._tree, int(num_tokens), vals.ctypes.data_as(ctypes.c_void_p), cap,
                                          lens.ctypes.data_as(ctypes.c_void_p), len(lens), ctypes.byref(nn))
        off = 0
        for i in range(nn.value):  # one free() per evicted node, in eviction order, like the reference
            n = int(lens[i])
            self.token_to_kv_pool_allocator.free(torch.from_numpy(vals[off : off + n].copy()).to(self.device))
            off += n
        assert off == total

    def inc_lock_ref(self, node: TreeNode) -> int:
        return 0 if self.disable else int(lib.sgl_mi355_radix_inc_lock_ref(self._tree, node.id))

    def dec_lock_ref(self, node: TreeNode) -> int:
        return 0 if self.disable else int(lib.sgl_mi355_radix_dec_lock_ref(self._tree, node.id))

    def evictable_size(self) -> int:
        return int(lib.sgl_mi355_radix_evictable_size(self._tree))

    def protected_size(self) -> int:
        return int(lib.sgl_mi355_radix_protected_size(self._tree))

    def total_size(self) -> int:
        return int(lib.sgl_mi355_radix_total_size(self._tree))

    # ------------------------------------------------------------------ request bookkeeping
    def cache_finished_req(self, req):
        """radix_cache.py:208-243: insert the finished request's tokens, free duplicated slots and its request row."""
        if self.disable:
            n = len(req.origin_input_ids) + len(req.output_ids) - 1
            self.token_to_kv_pool_allocator.free(self.req_to_token_pool.req_to_token[req.req_pool_idx, :n])
            self.req_to_token_pool.free(req.req_pool_idx)
            return
        token_ids = (req.origin_input_ids + req.output_ids)[:-1]
        kv_indices = self.req_to_token_pool.req_to_token[req.req_pool_idx, : len(token_ids)]
        aligned = len(kv_indices) // self.page_size * self.page_size if self.page_size != 1 else len(kv_indices)
        if aligned < len(kv_indices):
            self.token_to_kv_pool_allocator.free(kv_indices[aligned:])
        new_prefix_len = self.insert(token_ids[:aligned], kv_indices[:aligned].to(dtype=torch.int64, copy=True))
        self.token_to_kv_pool_allocator.free(kv_indices[len(req.prefix_indices) : new_prefix_len])
        self.req_to_token_pool.free(req.req_pool_idx)
        self.dec_lock_ref(req.last_node)

    def cache_unfinished_req(self, req):
        """radix_cache.py:245-288: insert the tokens computed so far, re-point the request at the tree's own slots."""
        if self.disable:
            return
        token_ids = req.fill_ids
        kv_indices = self.req_to_token_pool.req_to_token[req.req_pool_idx, : len(token_ids)]
        aligned = len(kv_indices) // self.page_size * self.page_size if self.page_size != 1 else len(kv_indices)
        aligned_ids = token_ids[:aligned]
        new_prefix_len = self.insert(aligned_ids, kv_indices[:aligned].to(dtype=torch.int64, copy=True))
        self.token_to_kv_pool_allocator.free(kv_indices[len(req.prefix_indices) : new_prefix_len])
        new_indices, new_last_node, _, _ = self.match_prefix(aligned_ids)
        self.req_to_token_pool.write((req.req_pool_idx, slice(len(req.prefix_indices), len(new_indices))),
                                     new_indices[len(req.prefix_indices) :].to(torch.int32))
        self.dec_lock_ref(req.last_node)
        self.inc_lock_ref(new_last_node)
        req.prefix_indices = new_indices if self.page_size == 1 else torch.cat([new_indices, kv_indices[len(new_indices) :]])
        req.last_node = new_last_node
