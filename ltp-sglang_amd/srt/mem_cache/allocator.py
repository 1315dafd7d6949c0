"""Slot allocators over the KV pool (bit-exact index order).

Follows python/sglang/srt/mem_cache/allocator.py: BaseTokenToKVPoolAllocator (:37-114),
TokenToKVPoolAllocator (page_size == 1, :117-160) and PagedTokenToKVPoolAllocator (:396-560, with the
alloc_extend / alloc_decode index kernels :275-395 run as HIP kernels here).

Order contract: free slots start as arange(1, size + 1) (slot 0 is the padding sink); ``alloc(n)`` takes the first
n of the free list; ``free`` appends to a release list that is merged and SORTED into the free list only when an
allocation would otherwise fail (lazy merge_and_sort_free, :82-88); inside a free group frees are batched.
"""
import torch


class BaseTokenToKVPoolAllocator:
    def __init__(self, size: int, page_size: int, dtype: torch.dtype, device: str, kvcache):
        self.size = size
        self.page_size = page_size
        self.dtype = dtype
        self.device = device
        self._kvcache = kvcache
        self.free_pages = None
        self.release_pages = None
        self.is_not_in_free_group = True
        self.free_group = []

    def debug_print(self) -> str:
        return ""

    def available_size(self):
        return (len(self.free_pages) + len(self.release_pages)) * self.page_size

    def get_kvcache(self):
        return self._kvcache

    def backup_state(self):
        return (self.free_pages, self.release_pages)

    def restore_state(self, state):
        self.free_pages, self.release_pages = state

    def free_group_begin(self):
        self.is_not_in_free_group = False
        self.free_group = []

    def free_group_end(self):
        self.is_not_in_free_group = True
        if self.free_group:
            self.free(torch.cat(self.free_group))

    def merge_and_sort_free(self):
        if len(self.release_pages) > 0:
            merged = torch.cat((self.free_pages, self.release_pages))
            self.free_pages = torch.sort(merged).values
            self.release_pages = torch.empty((0,), dtype=self.release_pages.dtype, device=self.device)

    def alloc_extend(self, *args, **kwargs):
        raise NotImplementedError("alloc_extend is only for paged allocator")

    def alloc_decode(self, *args, **kwargs):
        raise NotImplementedError("alloc_decode is only for paged allocator")


class TokenToKVPoolAllocator(BaseTokenToKVPoolAllocator):
    def __init__(self, size: int, dtype: torch.dtype, device: str, kvcache):
        super().__init__(size, 1, dtype, device, kvcache)
        self.clear()

    def clear(self):
        self.free_pages = torch.arange(1, self.size + 1, dtype=torch.int64, device=self.device)
        self.release_pages = torch.empty((0,), dtype=torch.int64, device=self.device)
        self.is_not_in_free_group = True
        self.free_group = []

    def available_size(self):
        return len(self.free_pages) + len(self.release_pages)

    def alloc(self, need_size: int):
        if need_size > len(self.free_pages):
            self.merge_and_sort_free()
            if need_size > len(self.free_pages):
                return None
        out = self.free_pages[:need_size]
        self.free_pages = self.free_pages[need_size:]
        return out

    def free(self, free_index: torch.Tensor):
        if free_index.numel() == 0:
            return
        if self.is_not_in_free_group:
            self.release_pages = torch.cat((self.release_pages, free_index))
        else:
            self.free_group.append(free_index)


class PagedTokenToKVPoolAllocator(BaseTokenToKVPoolAllocator):
    """Page-aligned allocator (allocator.py:396-560).  ``alloc`` hands out whole pages; ``alloc_extend`` /
    ``alloc_decode`` assign slots per request so that every request's KV stays page-aligned, with the index arithmetic
    done by the HIP kernels sgl_mi355_alloc_extend / sgl_mi355_alloc_decode (bit-exact with the reference's Triton
    kernels, :275-395).  """

    def __init__(self, size: int, page_size: int, dtype: torch.dtype, device: str, kvcache):
        super().__init__(size, page_size, dtype, device, kvcache)
        self.num_pages = size // page_size
        self.ret_values = torch.empty((1,), dtype=torch.int64, device=self.device)
        self.clear()

    def clear(self):
        self.free_pages = torch.arange(1, self.num_pages + 1, dtype=torch.int64, device=self.device)
        self.release_pages = torch.empty((0,), dtype=torch.int64, device=self.device)
        self.is_not_in_free_group = True
        self.free_group = []

    def alloc(self, need_size: int):
        num_pages = need_size // self.page_size
        if num_pages > len(self.free_pages):
            self.merge_and_sort_free()
            if num_pages > len(self.free_pages):
                return None
        pages = self.free_pages[:num_pages]
        self.free_pages = self.free_pages[num_pages:]
        return (pages[:, None] * self.page_size + torch.arange(self.page_size, device=self.device)).reshape(-1)

    def _new_pages(self, before, after):
        ps = self.page_size
        return int((((after + ps - 1) // ps) - ((before + ps - 1) // ps)).sum().item())

    def alloc_extend(self, prefix_lens: torch.Tensor, seq_lens: torch.Tensor, last_loc: torch.Tensor, extend_num_tokens: int):
        from ..._cabi import check, current_stream, is64, lib, ptr

        if self._new_pages(prefix_lens, seq_lens) > len(self.free_pages):
            self.merge_and_sort_free()
        bs = len(prefix_lens)
        out_indices = torch.empty((extend_num_tokens,), dtype=torch.int64, device=self.device)
        check(lib.sgl_mi355_alloc_extend(ptr(prefix_lens), is64(prefix_lens), ptr(seq_lens), is64(seq_lens), ptr(last_loc),
                                         is64(last_loc), ptr(self.free_pages), ptr(out_indices), ptr(self.ret_values),
                                         self.page_size, bs, current_stream()))
        num_new_pages = int(self.ret_values.item()) >> 32
        if num_new_pages > len(self.free_pages):
            return None
        self.free_pages = self.free_pages[num_new_pages:]
        return out_indices

    def alloc_decode(self, seq_lens: torch.Tensor, last_loc: torch.Tensor):
        from ..._cabi import check, current_stream, is64, lib, ptr

        if self._new_pages(seq_lens - 1, seq_lens) > len(self.free_pages):
            self.merge_and_sort_free()
        bs = len(seq_lens)
        out_indices = torch.empty((bs,), dtype=torch.int64, device=self.device)
        check(lib.sgl_mi355_alloc_decode(ptr(seq_lens), is64(seq_lens), ptr(last_loc), is64(last_loc), ptr(self.free_pages),
                                         ptr(out_indices), ptr(self.ret_values), self.page_size, bs, current_stream()))
        num_new_pages = int(self.ret_values.item())
        if num_new_pages > len(self.free_pages):
            return None
        self.free_pages = self.free_pages[num_new_pages:]
        return out_indices

    def free(self, free_index: torch.Tensor):
        if free_index.numel() == 0:
            return
        if self.is_not_in_free_group:
            pages = torch.unique(free_index // self.page_size)
            self.release_pages = torch.cat((pages, self.release_pages))
        else:
            self.free_group.append(free_index)
