"""Generates tests/golden/*.npz by running the *reference itself* in the build container.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [attention|index|quant|radix|all]

The reference tree (/root/reference) is imported in-process with the recipe of
tests/golden/_ref_import.py; it is never copied, and it does not exist on the GPU box:
only the vectors written here travel.  Inputs come from tests/_cases.py (seeded numpy),
so each fixture stores the expected outputs and the small integer tensors only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import _ref_import  # noqa: E402

_ref_import.install()

import _cases  # noqa: E402


def gen_attention():
    from types import SimpleNamespace

    from sglang.srt.layers.attention.torch_native_backend import TorchNativeAttnBackend
    from sglang.srt.layers.radix_attention import RadixAttention
    from sglang.srt.mem_cache.memory_pool import MHATokenToKVPool
    from sglang.srt.model_executor.forward_batch_info import ForwardBatch, ForwardMode

    out = {}
    for case in _cases.ATTN_CASES:
        c = _cases.build_attn_case(case)
        backend = TorchNativeAttnBackend(SimpleNamespace(device="cpu"))
        pool = MHATokenToKVPool(
            size=c["pool_size"], page_size=1, dtype=c["dtype"], head_num=c["hkv"], head_dim=c["d"],
            layer_num=1, device="cpu", enable_memory_saver=False,
        )
        # the pool holds every row except the ones this forward writes itself
        new = c["out_cache_loc"]
        pool.k_buffer[0].copy_(c["k_buffer"])
        pool.v_buffer[0].copy_(c["v_buffer"])
        pool.k_buffer[0][new] = 0
        pool.v_buffer[0][new] = 0
        layer = RadixAttention(num_heads=c["hq"], head_dim=c["d"], scaling=c["scaling"], num_kv_heads=c["hkv"], layer_id=0)
        decode = case["kind"] == "decode"
        fb = ForwardBatch(
            forward_mode=ForwardMode.DECODE if decode else ForwardMode.EXTEND,
            batch_size=c["bs"],
            input_ids=torch.zeros(len(new), dtype=torch.int64),
            req_pool_indices=c["req_pool_indices"],
            seq_lens=c["seq_lens"],
            out_cache_loc=new,
            seq_lens_sum=int(c["seq_lens"].sum()),
            extend_prefix_lens=None if decode else c["extend_prefix_lens"],
            extend_seq_lens=None if decode else c["extend_seq_lens"],
            attn_backend=backend,
        )
        fb.req_to_token_pool = SimpleNamespace(req_to_token=c["req_to_token"], size=c["req_to_token"].shape[0])
        fb.token_to_kv_pool = pool
        backend.init_forward_metadata(fb)
        q = c["q"].reshape(len(new), -1).clone()
        k_new = c["k_buffer"][new].clone()
        v_new = c["v_buffer"][new].clone()
        o = layer(q, k_new.reshape(len(new), -1), v_new.reshape(len(new), -1), fb)
        assert torch.equal(pool.k_buffer[0], c["k_buffer"]) and torch.equal(pool.v_buffer[0], c["v_buffer"]), \
            "set_kv_buffer did not restore the pool"
        rows = _cases.golden_rows(case, c)  # extend outputs are sub-sampled by row to keep fixtures small
        out[case["name"]] = _cases.bits16(o[rows])
        print(case["name"], tuple(o.shape), float(o.float().abs().max()))
    np.savez_compressed(os.path.join(HERE, "attention.npz"), **out)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.manual_seed(0)
    if what in ("attention", "all"):
        gen_attention()
    for extra in ("index", "quant", "radix"):
        fn = globals().get("gen_" + extra)
        if fn is not None and what in (extra, "all"):
            fn()
