"""BASELINE config 3: Llama-3-8B fp8, batch 64, shared-prefix RadixAttention hit path (common prefix 1536 + unique 512).
Prefill with the radix cache (first request in full, the other 63 extend 512 tokens over the cached 1536-token prefix)
vs prefill of all 64 prompts from scratch; then graph-captured decode is not measured here (bench.py covers decode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd.srt.mem_cache.radix_cache import RadixCache
from ltp_sglang_amd.srt.model_executor.synthetic_llama import LlamaShape, SyntheticModelRunner

def main():
    dev = "cuda:0"
    bs, pre, uniq = 64, 1536, 512
    cfg = LlamaShape.llama3_8b()
    runner = SyntheticModelRunner(cfg, "w8a8_fp8", max_running_requests=bs, context_len=pre + uniq + 16,
                                  max_total_tokens=bs * (pre + uniq) + 8192, device=dev, seed=0)
    g = torch.Generator().manual_seed(0)
    prefix = torch.randint(0, 10000, (pre,), generator=g)
    prompts = [torch.cat([prefix, torch.randint(0, 10000, (uniq,), generator=g)]).to(dev) for _ in range(bs)]
    # warm-up (kernel code objects), undone
    runner.extend([prompts[0][:512]]); runner.clear()

    def scratch():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for c0 in range(0, bs, 8):
            runner.extend(prompts[c0:c0 + 8])
        torch.cuda.synchronize(); return time.perf_counter() - t0

    def with_cache():
        cache = RadixCache(runner.req_to_token_pool, runner.token_to_kv_pool_allocator, page_size=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, st0 = runner.extend([prompts[0]])
        slots0 = runner.req_to_token_pool.req_to_token[st0.req_pool_indices[0], : pre + uniq].to(torch.int64)
        cache.insert(prompts[0].tolist(), slots0)
        rest = prompts[1:]
        for c0 in range(0, len(rest), 21):
            chunk = rest[c0:c0 + 21]
            hits = [cache.match_prefix(p.tolist()).device_indices for p in chunk]
            assert all(int(h.numel()) == pre for h in hits)
            runner.extend([p[pre:] for p in chunk], prefix_indices=[h.to(dev) for h in hits])
        torch.cuda.synchronize(); return time.perf_counter() - t0

    t_s = scratch(); runner.clear()
    t_c = with_cache(); runner.clear()
    toks = bs * (pre + uniq)
    print(f"batch {bs}, prefix {pre} + unique {uniq}: from scratch {t_s*1e3:.1f} ms ({toks/t_s:,.0f} tok/s)   "
          f"radix hit path {t_c*1e3:.1f} ms ({toks/t_c:,.0f} prompt tok/s, x{t_s/t_c:.2f})")
    # ---- decode over the shared prefix: the whole graph-replayed step, plain vs cascade attention ----
    from types import SimpleNamespace
    from ltp_sglang_amd import sgl_kernel as K
    cache = RadixCache(runner.req_to_token_pool, runner.token_to_kv_pool_allocator, page_size=1)
    l0, st0 = runner.extend([prompts[0]])
    slots0 = runner.req_to_token_pool.req_to_token[st0.req_pool_indices[0], : pre + uniq].to(torch.int64)
    cache.insert(prompts[0].tolist(), slots0)
    states, logits = [st0], [l0]
    rest = prompts[1:]
    for c0 in range(0, len(rest), 21):
        chunk = rest[c0:c0 + 21]
        hits = [cache.match_prefix(p.tolist()).device_indices for p in chunk]
        l, st = runner.extend([p[pre:] for p in chunk], prefix_indices=[h.to(dev) for h in hits])
        states.append(st); logits.append(l)
    state = SimpleNamespace(req_pool_indices=torch.cat([s.req_pool_indices for s in states]),
                            seq_lens=torch.cat([s.seq_lens for s in states]), seq_lens_cpu=sum([s.seq_lens_cpu for s in states], []))
    nxt = K.argmax(torch.cat(logits))
    for p_len in (0, pre):
        runner.capture_decode_graph(bs, shared_prefix_len=p_len)
        for _ in range(3):
            nxt = K.argmax(runner.decode_graph(state, nxt, shared_prefix_len=p_len))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        steps = 16
        for _ in range(steps):
            nxt = K.argmax(runner.decode_graph(state, nxt, shared_prefix_len=p_len))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        print(f"decode step, batch {bs} x ({pre} shared + {uniq}+ private), {'cascade' if p_len else 'plain'} attention: "
              f"{dt*1e3:.3f} ms/step = {bs/dt:,.0f} tok/s")

if __name__ == "__main__":
    main()
