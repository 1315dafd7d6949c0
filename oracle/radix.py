"""Radix-tree oracle: a small pure-Python restatement of the reference's prefix tree semantics.

Follows python/sglang/srt/mem_cache/radix_cache.py: _match_prefix_helper (:370-395), _split_node (:397-412),
_insert_helper (:414-445), evict (:297-320), inc_lock_ref / dec_lock_ref (:322-348).  Access order is a logical clock
(the reference uses time.monotonic(): the same order, without ties).  page_size == 1 keys by the first token, larger
pages by the first page tuple and match whole pages only.

TEST INFRASTRUCTURE: see oracle/__init__.py.
"""
import heapq
import itertools


class _N:
    def __init__(self, clock):
        self.kids, self.parent, self.key, self.val, self.lock, self.t = {}, None, [], [], 0, next(clock)

    def __lt__(self, other):
        return self.t < other.t


class RadixOracle:
    def __init__(self, page_size=1):
        self.ps = page_size
        self.reset()

    def reset(self):
        self.clock = itertools.count(1)
        self.root = _N(self.clock)
        self.root.lock = 1
        self.evictable = self.protected = 0

    def _ck(self, key):
        return key[0] if self.ps == 1 else tuple(key[: self.ps])

    def _common(self, a, b):
        n, i = min(len(a), len(b)), 0
        step = self.ps
        while i + step <= n and a[i : i + step] == b[i : i + step]:
            i += step
        return i

    def _split(self, child, n):
        top = _N(self.clock)
        top.kids = {self._ck(child.key[n:]): child}
        top.parent, top.lock = child.parent, child.lock
        top.key, top.val = child.key[:n], child.val[:n]
        child.parent, child.key, child.val = top, child.key[n:], child.val[n:]
        top.parent.kids[self._ck(top.key)] = top
        return top

    def match_prefix(self, key):
        key = list(key)
        if self.ps != 1:
            key = key[: len(key) // self.ps * self.ps]
        node, out = self.root, []
        if not key:
            return out, node
        node.t = next(self.clock)
        while key and self._ck(key) in node.kids:
            child = node.kids[self._ck(key)]
            child.t = next(self.clock)
            n = self._common(child.key, key)
            if n < len(child.key):
                node = self._split(child, n)
                out += node.val
                break
            out += child.val
            node, key = child, key[n:]
        return out, node

    def insert(self, key, val):
        key, val = list(key), list(val)
        node, total = self.root, 0
        node.t = next(self.clock)
        while key and self._ck(key) in node.kids:
            node = node.kids[self._ck(key)]
            node.t = next(self.clock)
            n = self._common(node.key, key)
            total += n
            key, val = key[n:], val[n:]
            if n < len(node.key):
                node = self._split(node, n)
        if key:
            leaf = _N(self.clock)
            leaf.parent, leaf.key, leaf.val = node, key, val
            node.kids[self._ck(key)] = leaf
            self.evictable += len(val)
        return total

    def evict(self, num_tokens):
        """-> list of freed value lists, in eviction order."""
        leaves, stack = [], [self.root]
        while stack:
            n = stack.pop()
            (leaves.append(n) if not n.kids else stack.extend(n.kids.values()))
        heapq.heapify(leaves)
        freed, count = [], 0
        while count < num_tokens and leaves:
            x = heapq.heappop(leaves)
            if x is self.root:
                break
            if x.lock > 0:
                continue
            freed.append(list(x.val))
            count += len(x.val)
            del x.parent.kids[self._ck(x.key)]
            self.evictable -= len(x.key)
            if not x.parent.kids:
                heapq.heappush(leaves, x.parent)
        return freed

    def _walk_lock(self, node, d):
        delta = 0
        while node is not self.root:
            if node.lock == (0 if d > 0 else 1):
                self.evictable -= d * len(node.val)
                self.protected += d * len(node.val)
                delta -= d * len(node.val)
            node.lock += d
            node = node.parent
        return delta

    def inc_lock_ref(self, node):
        return self._walk_lock(node, +1)

    def dec_lock_ref(self, node):
        return self._walk_lock(node, -1)

    def total_size(self):
        tot, stack = 0, [self.root]
        while stack:
            n = stack.pop()
            tot += len(n.val)
            stack.extend(n.kids.values())
        return tot
