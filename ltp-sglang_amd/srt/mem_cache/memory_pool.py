"""Request->token table and the per-layer MHA KV pool.

Follows python/sglang/srt/mem_cache/memory_pool.py: ReqToTokenPool (:47-94), KVCache (:97-150) and
MHATokenToKVPool (:162-417).  Layout contract kept bit-for-bit:
  * req_to_token is int32 [size, max_context_len], zero-initialised, with a FIFO free list of request rows;
  * every layer has k_buffer/v_buffer of shape [size + page_size, head_num, head_dim]; slot 0 is the padding
    sink that padded tokens write to and the allocator never hands out (:222-227).
MI355X layout: all layers live in ONE [2, layer_num, slots, head_num, head_dim] allocation (K plane, V plane) so
that 288 GB of HBM is carved once; ``k_buffer[l]`` / ``v_buffer[l]`` are views with the reference's shapes.
``set_kv_buffer`` is a single HIP scatter launch for K and V (the reference issues two index_puts, :401-407).
"""
from typing import List, Optional, Tuple, Union

import torch


class ReqToTokenPool:
    def __init__(self, size: int, max_context_len: int, device: str, enable_memory_saver: bool):
        self.size = size
        self.max_context_len = max_context_len
        self.device = device
        self.req_to_token = torch.zeros((size, max_context_len), dtype=torch.int32, device=device)
        self.free_slots = list(range(size))

    def write(self, indices, values):
        self.req_to_token[indices] = values

    def available_size(self) -> int:
        return len(self.free_slots)

    def alloc(self, need_size: int) -> Optional[List[int]]:
        if need_size > len(self.free_slots):
            return None
        taken, self.free_slots = self.free_slots[:need_size], self.free_slots[need_size:]
        return taken

    def free(self, free_index: Union[int, List[int]]):
        if isinstance(free_index, int):
            self.free_slots.append(free_index)
        else:
            self.free_slots.extend(free_index)

    def clear(self):
        self.free_slots = list(range(self.size))


class KVCache:
    def __init__(self, size, page_size, dtype, layer_num, device, start_layer=None, end_layer=None):
        self.size = size
        self.page_size = page_size
        self.dtype = dtype
        self.device = device
        # fp8 pools are stored as bytes (the reference does the same because index_put lacks fp8, :114-118)
        self.store_dtype = torch.uint8 if dtype in (torch.float8_e5m2, torch.float8_e4m3fn) else dtype
        self.layer_num = layer_num
        self.start_layer = start_layer or 0
        self.end_layer = end_layer or layer_num - 1
        self.mem_usage = 0


class MHATokenToKVPool(KVCache):
    def __init__(self, size: int, page_size: int, dtype: torch.dtype, head_num: int, head_dim: int, layer_num: int,
                 device: str, enable_memory_saver: bool, start_layer: Optional[int] = None,
                 end_layer: Optional[int] = None):
        super().__init__(size, page_size, dtype, layer_num, device, start_layer, end_layer)
        self.head_num = head_num
        self.head_dim = head_dim
        slots = size + page_size
        self._kv = torch.zeros((2, layer_num, slots, head_num, head_dim), dtype=self.store_dtype, device=device)
        self.k_buffer = [self._kv[0, l] for l in range(layer_num)]
        self.v_buffer = [self._kv[1, l] for l in range(layer_num)]
        self.layer_transfer_counter = None
        # base address and bytes per row of every K then every V buffer (memory_pool.py:241-256): move_kv_cache's operands
        self.k_data_ptrs = torch.tensor([x.data_ptr() for x in self.k_buffer], dtype=torch.uint64, device=device)
        self.v_data_ptrs = torch.tensor([x.data_ptr() for x in self.v_buffer], dtype=torch.uint64, device=device)
        self.data_ptrs = torch.cat([self.k_data_ptrs, self.v_data_ptrs], dim=0)
        self.data_strides = torch.tensor([x[0].numel() * x.element_size() for x in self.k_buffer + self.v_buffer],
                                         dtype=torch.int64, device=device)
        k_bytes, v_bytes = self.get_kv_size_bytes()
        self.mem_usage = (k_bytes + v_bytes) / (1 << 30)

    def get_kv_size_bytes(self) -> Tuple[int, int]:
        half = self._kv[0].numel() * self._kv.element_size()
        return half, half

    def _typed(self, buf):
        return buf.view(self.dtype) if self.store_dtype != self.dtype else buf

    def get_key_buffer(self, layer_id: int):
        if self.layer_transfer_counter is not None:
            self.layer_transfer_counter.wait_until(layer_id - self.start_layer)
        return self._typed(self.k_buffer[layer_id - self.start_layer])

    def get_value_buffer(self, layer_id: int):
        if self.layer_transfer_counter is not None:
            self.layer_transfer_counter.wait_until(layer_id - self.start_layer)
        return self._typed(self.v_buffer[layer_id - self.start_layer])

    def get_kv_buffer(self, layer_id: int):
        return self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)

    def set_kv_buffer(self, layer, loc: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor,
                      k_scale: Optional[float] = None, v_scale: Optional[float] = None,
                      layer_id_override: Optional[int] = None):
        """pool[layer][loc] = (cache_k, cache_v); memory_pool.py:369-407."""
        from ...sgl_kernel import set_kv_buffer

        layer_id = layer.layer_id if layer_id_override is None else layer_id_override
        if self.dtype == torch.float8_e4m3fn and cache_k.dtype in (torch.bfloat16, torch.float16) and cache_k.is_cuda:
            # kv_cache_dtype fp8_e4m3: division by the scale (if any) and the e4m3 conversion happen inside the scatter
            # kernel (the reference's div_ / .to() / index_put sequence, :385-407; the caller's tensors stay untouched)
            i = layer_id - self.start_layer
            set_kv_buffer(self._typed(self.k_buffer[i]), self._typed(self.v_buffer[i]), loc, cache_k, cache_v,
                          None if k_scale is None else float(k_scale), None if v_scale is None else float(v_scale))
            return
        if cache_k.dtype != self.dtype:
            if k_scale is not None:
                cache_k.div_(k_scale)
            if v_scale is not None:
                cache_v.div_(v_scale)
            cache_k, cache_v = cache_k.to(self.dtype), cache_v.to(self.dtype)
        if self.store_dtype != self.dtype:
            cache_k, cache_v = cache_k.view(self.store_dtype), cache_v.view(self.store_dtype)
        i = layer_id - self.start_layer
        set_kv_buffer(self.k_buffer[i], self.v_buffer[i], loc, cache_k, cache_v)

    def move_kv_cache(self, tgt_loc: torch.Tensor, src_loc: torch.Tensor):
        """pool[:, tgt_loc] = pool[:, src_loc] for the K and V buffers of every layer, in place, byte for byte
        (memory_pool.py:409-417 -> copy_all_layer_kv_cache :1046-1081): one HIP launch."""
        from ...sgl_kernel import move_kv_cache

        move_kv_cache(self.data_ptrs, self.data_strides, self.head_num * self.head_dim * self._kv.element_size(), tgt_loc, src_loc)
