// Native radix tree over token ids for RadixAttention prefix sharing (host code; no device work).
//
// Semantics follow python/sglang/srt/mem_cache/radix_cache.py:43-555 exactly -- the KV slot indices a request gets
// back from match_prefix / insert, and the order in which evict() releases them, are the bit-exact part of the
// contract:
//   match_prefix  walks children by the first page of the remaining key, SPLITS a node on a partial match (:370-412)
//   insert        same walk, splitting on partial matches, then hangs the unmatched tail as a new leaf (:414-445)
//   evict         LRU over leaves by last access time, skipping locked nodes, re-queuing parents that become
//                 leaves (:297-320)
//   inc/dec_lock_ref walk to the root moving sizes between the evictable and protected counters (:322-348)
// The reference orders nodes by time.monotonic(); here a logical clock ticks on every touch, which gives the same
// order without ties.  (The reference also ships an experimental C++ tree, mem_cache/cpp_radix_tree; this is an
// independent implementation of the Python semantics.)
#include <stdint.h>

#include <algorithm>
#include <map>
#include <memory>
#include <queue>
#include <unordered_map>
#include <vector>

namespace {

struct Node {
  int64_t id;
  Node* parent = nullptr;
  std::map<std::vector<int64_t>, Node*> children;  // keyed by the first page of the child's key
  std::vector<int64_t> key, value;
  int64_t lock_ref = 0;
  uint64_t last_access = 0;
};

struct Tree {
  int page_size;
  uint64_t clock = 0;
  int64_t next_id = 0;
  int64_t evictable = 0, protected_ = 0;
  Node* root = nullptr;
  std::unordered_map<int64_t, std::unique_ptr<Node>> nodes;

  explicit Tree(int ps) : page_size(ps) { reset(); }

  Node* make_node() {
    auto n = std::make_unique<Node>();
    n->id = next_id++;
    n->last_access = ++clock;
    Node* raw = n.get();
    nodes[raw->id] = std::move(n);
    return raw;
  }
  void reset() {
    nodes.clear();
    evictable = protected_ = 0;
    root = make_node();
    root->lock_ref = 1;
  }
  std::vector<int64_t> child_key(const int64_t* key, int64_t len) const {
    return std::vector<int64_t>(key, key + std::min<int64_t>(len, page_size));
  }
  int64_t match_len(const std::vector<int64_t>& a, const int64_t* b, int64_t blen) const {
    const int64_t n = std::min<int64_t>((int64_t)a.size(), blen);
    int64_t i = 0;
    if (page_size == 1) {
      while (i < n && a[i] == b[i]) ++i;
      return i;
    }
    while (i < n) {  // whole pages only; a trailing partial page never matches a full one
      const int64_t e = i + page_size;
      if (e > (int64_t)a.size() || e > blen) break;
      if (!std::equal(a.begin() + i, a.begin() + e, b + i)) break;
      i = e;
    }
    return i;
  }
  Node* split(Node* child, int64_t split_len) {
    Node* nn = make_node();
    nn->children[child_key(child->key.data() + split_len, (int64_t)child->key.size() - split_len)] = child;
    nn->parent = child->parent;
    nn->lock_ref = child->lock_ref;
    nn->key.assign(child->key.begin(), child->key.begin() + split_len);
    nn->value.assign(child->value.begin(), child->value.begin() + split_len);
    child->parent = nn;
    child->key.erase(child->key.begin(), child->key.begin() + split_len);
    child->value.erase(child->value.begin(), child->value.begin() + split_len);
    nn->parent->children[child_key(nn->key.data(), (int64_t)nn->key.size())] = nn;
    return nn;
  }
};

inline Tree* T(void* p) { return static_cast<Tree*>(p); }

}  // namespace

extern "C" void* sgl_mi355_radix_create(int page_size) { return page_size >= 1 ? new Tree(page_size) : nullptr; }
extern "C" int sgl_mi355_radix_destroy(void* t) {
  delete T(t);
  return 0;
}
extern "C" int sgl_mi355_radix_reset(void* t) {
  T(t)->reset();
  return 0;
}
extern "C" int64_t sgl_mi355_radix_root(void* t) { return T(t)->root->id; }

// Returns the number of matched tokens; out_values receives min(match, out_cap) slot indices; *last_node the id of
// the deepest matched node (the root when nothing matches).  Mutates the tree (access times, splits).
extern "C" int64_t sgl_mi355_radix_match_prefix(void* tp, const int64_t* key, int64_t key_len, int64_t* out_values,
                                                int64_t out_cap, int64_t* last_node) {
  Tree* t = T(tp);
  if (t->page_size != 1) key_len = key_len / t->page_size * t->page_size;
  Node* node = t->root;
  int64_t got = 0;
  auto emit = [&](const std::vector<int64_t>& v) {
    for (int64_t x : v) {
      if (got < out_cap) out_values[got] = x;
      ++got;
    }
  };
  if (key_len > 0) {
    node->last_access = ++t->clock;
    const int64_t* k = key;
    int64_t rem = key_len;
    while (rem > 0) {
      auto it = node->children.find(t->child_key(k, rem));
      if (it == node->children.end()) break;
      Node* child = it->second;
      child->last_access = ++t->clock;
      const int64_t pl = t->match_len(child->key, k, rem);
      if (pl < (int64_t)child->key.size()) {
        Node* nn = t->split(child, pl);
        emit(nn->value);
        node = nn;
        break;
      }
      emit(child->value);
      node = child;
      k += pl;
      rem -= pl;
    }
  }
  *last_node = node->id;
  return got;
}

// Returns the length of the prefix that was already present.
extern "C" int64_t sgl_mi355_radix_insert(void* tp, const int64_t* key, const int64_t* values, int64_t len) {
  Tree* t = T(tp);
  Node* node = t->root;
  node->last_access = ++t->clock;
  if (len == 0) return 0;
  const int64_t* k = key;
  const int64_t* v = values;
  int64_t rem = len, total = 0;
  while (rem > 0) {
    auto it = node->children.find(t->child_key(k, rem));
    if (it == node->children.end()) break;
    node = it->second;
    node->last_access = ++t->clock;
    const int64_t pl = t->match_len(node->key, k, rem);
    total += pl;
    k += pl;
    v += pl;
    rem -= pl;
    if (pl < (int64_t)node->key.size()) node = t->split(node, pl);
  }
  if (rem > 0) {
    Node* nn = t->make_node();
    nn->parent = node;
    nn->key.assign(k, k + rem);
    nn->value.assign(v, v + rem);
    node->children[t->child_key(k, rem)] = nn;
    t->evictable += rem;
  }
  return total;
}

// Evicts least-recently-used unlocked leaves until at least num_tokens slots are released.  The released slot indices
// are written node by node (out_values, up to out_cap) with the per-node counts in out_node_lens; returns the total
// number of released slots and sets *n_nodes.
extern "C" int64_t sgl_mi355_radix_evict(void* tp, int64_t num_tokens, int64_t* out_values, int64_t out_cap,
                                         int64_t* out_node_lens, int64_t lens_cap, int64_t* n_nodes) {
  Tree* t = T(tp);
  auto later = [](const Node* a, const Node* b) { return a->last_access > b->last_access; };
  std::priority_queue<Node*, std::vector<Node*>, decltype(later)> heap(later);
  for (auto& kv : t->nodes)
    if (kv.second->children.empty()) heap.push(kv.second.get());
  int64_t evicted = 0, nn = 0;
  while (evicted < num_tokens && !heap.empty()) {
    Node* x = heap.top();
    heap.pop();
    if (x == t->root) break;
    if (x->lock_ref > 0) continue;
    for (int64_t s : x->value) {
      if (evicted < out_cap) out_values[evicted] = s;
      ++evicted;
    }
    if (nn < lens_cap) out_node_lens[nn] = (int64_t)x->value.size();
    ++nn;
    Node* parent = x->parent;
    for (auto it = parent->children.begin(); it != parent->children.end(); ++it)
      if (it->second == x) {
        parent->children.erase(it);
        break;
      }
    t->evictable -= (int64_t)x->key.size();
    const int64_t dead = x->id;
    t->nodes.erase(dead);
    if (parent->children.empty()) heap.push(parent);
  }
  *n_nodes = nn;
  return evicted;
}

extern "C" int64_t sgl_mi355_radix_inc_lock_ref(void* tp, int64_t node_id) {
  Tree* t = T(tp);
  auto it = t->nodes.find(node_id);
  if (it == t->nodes.end()) return 0;
  int64_t delta = 0;
  for (Node* n = it->second.get(); n != t->root; n = n->parent) {
    if (n->lock_ref == 0) {
      t->evictable -= (int64_t)n->value.size();
      t->protected_ += (int64_t)n->value.size();
      delta -= (int64_t)n->value.size();
    }
    n->lock_ref += 1;
  }
  return delta;
}

extern "C" int64_t sgl_mi355_radix_dec_lock_ref(void* tp, int64_t node_id) {
  Tree* t = T(tp);
  auto it = t->nodes.find(node_id);
  if (it == t->nodes.end()) return 0;
  int64_t delta = 0;
  for (Node* n = it->second.get(); n != t->root; n = n->parent) {
    if (n->lock_ref == 1) {
      t->evictable += (int64_t)n->value.size();
      t->protected_ -= (int64_t)n->value.size();
      delta += (int64_t)n->value.size();
    }
    n->lock_ref -= 1;
  }
  return delta;
}

extern "C" int64_t sgl_mi355_radix_evictable_size(void* t) { return T(t)->evictable; }
extern "C" int64_t sgl_mi355_radix_protected_size(void* t) { return T(t)->protected_; }
extern "C" int64_t sgl_mi355_radix_total_size(void* tp) {
  int64_t s = 0;
  for (auto& kv : T(tp)->nodes) s += (int64_t)kv.second->value.size();
  return s;
}
extern "C" int64_t sgl_mi355_radix_num_nodes(void* tp) { return (int64_t)T(tp)->nodes.size(); }

// Introspection for tests / debugging: returns key length (-1 for an unknown id).
extern "C" int64_t sgl_mi355_radix_node_info(void* tp, int64_t node_id, int64_t* parent, int64_t* lock_ref,
                                             int64_t* num_children) {
  Tree* t = T(tp);
  auto it = t->nodes.find(node_id);
  if (it == t->nodes.end()) return -1;
  Node* n = it->second.get();
  *parent = n->parent ? n->parent->id : -1;
  *lock_ref = n->lock_ref;
  *num_children = (int64_t)n->children.size();
  return (int64_t)n->key.size();
}
