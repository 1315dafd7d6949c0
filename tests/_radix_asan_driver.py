"""Child process of tests/test_radix_sanitizer.py: replays the reference's radix-tree traces (tests/golden/radix.json) through the
product's RadixCache with the native tree (csrc/radix_tree.hip, host code) swapped for an AddressSanitizer + UBSan build of the
same source.  argv: repo root, sanitized .so.  Exit code 0 = traces identical and no sanitizer report (a report aborts)."""
import ctypes, json, os, sys
ROOT = sys.argv[1]; SO = sys.argv[2]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
load_package()
import torch
import _cases
from ltp_sglang_amd import _cabi
from ltp_sglang_amd.srt.mem_cache import radix_cache as rc
asan = ctypes.CDLL(SO)
class Shim:
    def __getattr__(self, name):
        real = getattr(_cabi.lib, name)
        if not name.startswith("sgl_mi355_radix_"):
            return real
        f = getattr(asan, name)
        f.argtypes, f.restype = real.argtypes, real.restype
        return f
rc.lib = Shim()
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "radix.json")))
def norm(t): return json.loads(json.dumps(t))
bad = 0
for page_size in (1, 4):
    for seed in (0, 1, 2):
        free_log = []
        fake_alloc = type("A", (), {"device": "cpu", "free": lambda self, idx: free_log.append([int(x) for x in idx])})()
        class Adapter:
            def __init__(self): self.c = rc.RadixCache(None, fake_alloc, page_size=page_size)
            def match_prefix(self, key):
                r = self.c.match_prefix(key); return r.device_indices.tolist(), r.last_device_node
            def insert(self, key, vals): return self.c.insert(key, torch.tensor(vals, dtype=torch.int64))
            def __getattr__(self, n): return getattr(self.c, n)
        trace = _cases.radix_primitive_script(Adapter, free_log, seed=seed, page_size=page_size)
        bad += norm(trace) != GOLD[f"prim_p{page_size}_s{seed}"]
from ltp_sglang_amd.srt.mem_cache.allocator import TokenToKVPoolAllocator
from ltp_sglang_amd.srt.mem_cache.memory_pool import ReqToTokenPool


def env():
    pool = ReqToTokenPool(32, 256, "cpu", False)
    alloc = TokenToKVPoolAllocator(600, torch.bfloat16, "cpu", None)
    return rc.RadixCache(pool, alloc, page_size=1), pool, alloc


for seed in (0, 1):   # request-level flow: match -> alloc -> lock -> cache_unfinished_req -> decode -> cache_finished_req -> evict
    bad += norm(_cases.radix_request_script(env, seed=seed)) != GOLD[f"req_s{seed}"]
print("mismatches", bad)
sys.exit(1 if bad else 0)
