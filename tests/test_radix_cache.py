"""CPU: RadixAttention host logic.  The reference's own RadixCache / allocator produced tests/golden/radix.json from
seeded scripts (tests/_cases.py); the oracle restatement and the PRODUCT classes (native C++ tree behind the C-ABI,
allocator, ReqToTokenPool) must reproduce every recorded index, length, freed slot and counter bit-exactly."""
import json
import os

import pytest
import torch

import _cases
from oracle.radix import RadixOracle

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "radix.json")))


def _norm(trace):
    return json.loads(json.dumps(trace))


class _OracleAdapter:
    def __init__(self, page_size, free_log):
        self.c, self.free_log = RadixOracle(page_size), free_log

    def match_prefix(self, key):
        return self.c.match_prefix(key)

    def insert(self, key, vals):
        return self.c.insert(key, vals)

    def evict(self, n):
        self.free_log.extend(self.c.evict(n))

    def inc_lock_ref(self, node):
        return self.c.inc_lock_ref(node)

    def dec_lock_ref(self, node):
        return self.c.dec_lock_ref(node)

    def evictable_size(self):
        return self.c.evictable

    def protected_size(self):
        return self.c.protected

    def total_size(self):
        return self.c.total_size()


@pytest.mark.parametrize("page_size", [1, 4])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_radix_matches_reference_trace(page_size, seed):
    free_log = []
    trace = _cases.radix_primitive_script(lambda: _OracleAdapter(page_size, free_log), free_log, seed=seed, page_size=page_size)
    assert _norm(trace) == GOLD[f"prim_p{page_size}_s{seed}"]


@pytest.mark.parametrize("page_size", [1, 4])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_native_radix_matches_reference_trace(page_size, seed, pkg):
    from ltp_sglang_amd.srt.mem_cache.radix_cache import RadixCache

    free_log = []
    fake_alloc = type("A", (), {"device": "cpu", "free": lambda self, idx: free_log.append([int(x) for x in idx])})()

    class Adapter:
        def __init__(self):
            self.c = RadixCache(None, fake_alloc, page_size=page_size)

        def match_prefix(self, key):
            r = self.c.match_prefix(key)
            return r.device_indices.tolist(), r.last_device_node

        def insert(self, key, vals):
            return self.c.insert(key, torch.tensor(vals, dtype=torch.int64))

        def __getattr__(self, n):
            return getattr(self.c, n)

    trace = _cases.radix_primitive_script(Adapter, free_log, seed=seed, page_size=page_size)
    assert _norm(trace) == GOLD[f"prim_p{page_size}_s{seed}"]


@pytest.mark.parametrize("seed", [0, 1])
def test_request_level_flow_matches_reference(seed, pkg):
    """match_prefix -> alloc -> req_to_token writes -> lock -> cache_unfinished_req -> decode -> cache_finished_req ->
    evict, with the product's RadixCache + TokenToKVPoolAllocator + ReqToTokenPool: slot-for-slot identical."""
    from ltp_sglang_amd.srt.mem_cache.allocator import TokenToKVPoolAllocator
    from ltp_sglang_amd.srt.mem_cache.memory_pool import ReqToTokenPool
    from ltp_sglang_amd.srt.mem_cache.radix_cache import RadixCache

    def env():
        pool = ReqToTokenPool(32, 256, "cpu", False)
        alloc = TokenToKVPoolAllocator(600, torch.bfloat16, "cpu", None)
        return RadixCache(pool, alloc, page_size=1), pool, alloc

    assert _norm(_cases.radix_request_script(env, seed=seed)) == GOLD[f"req_s{seed}"]


def test_shared_prefix_aliases_slots(pkg):
    """64 requests with a common 96-token prefix: every prefix_indices aliases the same slots and the tree holds the
    prefix once (the RadixAttention hit path of BASELINE configs[2])."""
    from ltp_sglang_amd.srt.mem_cache.radix_cache import RadixCache

    freed = []
    alloc = type("A", (), {"device": "cpu", "free": lambda self, idx: freed.append(idx)})()
    cache = RadixCache(None, alloc, page_size=1)
    prefix = list(range(1000, 1096))
    cache.insert(prefix + [1], torch.arange(1, 98))
    base = cache.match_prefix(prefix).device_indices
    assert base.tolist() == list(range(1, 97))
    for r in range(64):
        got = cache.match_prefix(prefix + [5000 + r]).device_indices
        assert torch.equal(got, base)
    assert cache.total_size() == 97 and cache.evictable_size() == 97


def test_allocator_order_and_lazy_merge(pkg):
    from ltp_sglang_amd.srt.mem_cache.allocator import TokenToKVPoolAllocator

    a = TokenToKVPoolAllocator(12, torch.bfloat16, "cpu", None)
    assert a.alloc(5).tolist() == [1, 2, 3, 4, 5]          # slot 0 is the padding sink and is never handed out
    x = a.alloc(4)
    a.free(torch.tensor([3, 1]))
    assert a.available_size() == 5
    assert a.alloc(3).tolist() == [10, 11, 12]             # released slots are not reused before the free list runs dry
    assert a.alloc(3) is None                              # lazy merge happened (sorted: [1, 3]) but only 2 are free
    assert a.alloc(2).tolist() == [1, 3]
    a.free_group_begin()
    a.free(x[:2]); a.free(x[2:])
    assert a.available_size() == 0
    a.free_group_end()
    assert sorted(a.alloc(4).tolist()) == x.tolist()
    assert a.alloc(1) is None


class _OraclePagedAlloc:
    """numpy restatement of PagedTokenToKVPoolAllocator (oracle/kv_index.py alloc_extend / alloc_decode + the host logic)."""

    def __init__(self, size, page_size):
        self.ps = page_size
        self.free_pages = torch.arange(1, size // page_size + 1)
        self.release = torch.empty(0, dtype=torch.int64)

    def merge_and_sort_free(self):
        if len(self.release):
            self.free_pages = torch.sort(torch.cat((self.free_pages, self.release))).values
            self.release = torch.empty(0, dtype=torch.int64)

    def available_size(self):
        return (len(self.free_pages) + len(self.release)) * self.ps

    def _need(self, before, after):
        return int((-(-after // self.ps) - (-(-before // self.ps))).sum())

    def alloc_extend(self, pre, seq, last, n):
        from oracle import kv_index as oi

        if self._need(pre, seq) > len(self.free_pages):
            self.merge_and_sort_free()
        out, used = oi.alloc_extend(pre.numpy(), seq.numpy(), last.numpy(), self.free_pages.numpy(), self.ps)
        self.free_pages = self.free_pages[used:]
        return torch.from_numpy(out)

    def alloc_decode(self, seq, last):
        from oracle import kv_index as oi

        if self._need(seq - 1, seq) > len(self.free_pages):
            self.merge_and_sort_free()
        out, used = oi.alloc_decode(seq.numpy(), last.numpy(), self.free_pages.numpy(), self.ps)
        self.free_pages = self.free_pages[used:]
        return torch.from_numpy(out)

    def free(self, idx):
        self.release = torch.cat((torch.unique(idx // self.ps), self.release))


@pytest.mark.parametrize("page_size", [4, 16])
@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_paged_allocator_matches_reference(page_size, seed):
    trace = _cases.paged_alloc_script(_OraclePagedAlloc, page_size, seed=seed)
    assert _norm(trace) == GOLD[f"paged_p{page_size}_s{seed}"]


@pytest.mark.gpu
@pytest.mark.parametrize("page_size", [4, 16])
@pytest.mark.parametrize("seed", [0, 1])
def test_hip_paged_allocator_matches_reference(page_size, seed, pkg):
    from ltp_sglang_amd.srt.mem_cache.allocator import PagedTokenToKVPoolAllocator

    trace = _cases.paged_alloc_script(
        lambda size, page: PagedTokenToKVPoolAllocator(size, page, torch.bfloat16, "cuda:0", None), page_size, seed=seed,
        device="cuda:0")
    assert _norm(trace) == GOLD[f"paged_p{page_size}_s{seed}"]
