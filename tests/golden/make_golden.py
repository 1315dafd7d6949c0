"""Generates tests/golden/*.npz by running the *reference itself* in the build container.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [attention|index|quant|elementwise|radix|model|all]

The reference tree (/root/reference) is imported in-process with the recipe of
tests/golden/_ref_import.py; it is never copied, and it does not exist on the GPU box:
only the vectors written here travel.  Inputs come from tests/_cases.py (seeded numpy),
so each fixture stores the expected outputs and the small integer tensors only.
"""
import os
import sys

os.environ.setdefault("TRITON_INTERPRET", "1")  # the reference's Triton index kernels run on the CPU interpreter

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root: oracle/ (the exact-arithmetic twin of G7)

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


def _ref_functions(path, names, extra_globals=None):
    """Executes selected top-level function definitions of a reference source file in a scratch
    namespace (for modules whose import drags in absent third-party packages).  Nothing is copied:
    the text is read from /root/reference at generation time only."""
    import ast

    src = open(path).read()
    tree = ast.parse(src)
    ns = dict(extra_globals or {})
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), ns)
    return ns


def gen_index():
    """Needs TRITON_INTERPRET=1 (set by the __main__ block before triton is imported)."""
    import triton
    import triton.language as tl

    from sglang.srt.layers.attention.triton_backend import get_num_kv_splits_triton
    from sglang.srt.layers.attention.utils import create_flashinfer_kv_indices_triton
    from sglang.srt.model_executor.forward_batch_info import compute_position_torch, compute_position_triton

    sb = _ref_functions(
        os.path.join(_ref_import.REF_ROOT, "python/sglang/srt/managers/schedule_batch.py"),
        {"get_last_loc_torch", "write_req_to_token_pool_triton"},
        {"torch": torch, "triton": triton, "tl": tl},
    )
    out = {}
    for case in _cases.INDEX_CASES:
        c = _cases.build_index_case(case)
        bs = c["bs"]
        r2t = torch.from_numpy(c["req_to_token"].copy())
        rpi = torch.from_numpy(c["req_pool_indices"])
        pre, seq, ext = (torch.from_numpy(c[k]) for k in ("pre", "seq", "ext"))
        loc = torch.from_numpy(c["out_cache_loc"])
        sb["write_req_to_token_pool_triton"][(bs,)](r2t, rpi, pre, seq, ext, loc, r2t.shape[1])
        out[case["name"] + ".req_to_token"] = r2t.numpy().copy()
        # decode-style kv indices over the full sequences, and extend-style over the prefixes
        for tag, lens in (("seq", seq), ("pre", pre)):
            kv_indptr = torch.zeros(bs + 1, dtype=torch.int32)
            kv_indptr[1:] = torch.cumsum(lens, dim=0)
            kv_indices = torch.zeros(int(lens.sum()), dtype=torch.int32)
            create_flashinfer_kv_indices_triton[(bs,)](r2t, rpi, lens, kv_indptr, None, kv_indices, r2t.stride(0))
            out[f"{case['name']}.kv_indptr_{tag}"] = kv_indptr.numpy().copy()
            out[f"{case['name']}.kv_indices_{tag}"] = kv_indices.numpy().copy()
        # sliding-window style start offsets
        win = torch.minimum(seq, torch.tensor(9))
        kv_indptr = torch.zeros(bs + 1, dtype=torch.int32)
        kv_indptr[1:] = torch.cumsum(win, dim=0)
        kv_indices = torch.zeros(int(win.sum()), dtype=torch.int32)
        create_flashinfer_kv_indices_triton[(bs,)](r2t, rpi, win, kv_indptr, seq - win, kv_indices, r2t.stride(0))
        out[case["name"] + ".kv_indices_win"] = kv_indices.numpy().copy()
        pos_t, start_t = compute_position_torch(pre.int(), ext.int())
        pos_k, start_k = compute_position_triton(pre.int(), ext.int(), int(ext.sum()))
        assert torch.equal(pos_t, pos_k) and torch.equal(start_t.int(), start_k)
        out[case["name"] + ".positions"] = pos_k.numpy().copy()
        out[case["name"] + ".extend_start_loc"] = start_k.numpy().copy()
        out[case["name"] + ".last_loc"] = sb["get_last_loc_torch"](r2t, rpi, pre).numpy().copy()
    for case in _cases.SPLIT_CASES:
        seq = torch.tensor(case["seq"], dtype=torch.int64)
        n = len(case["seq"])
        res = torch.zeros(n, dtype=torch.int32)
        sched = 256 if n < 256 else triton.next_power_of_2(n)
        get_num_kv_splits_triton[(1,)](res, seq, n, 1, case["num_head"], case["num_kv_head"], case["max_splits"],
                                       case["cores"], MAX_NUM_SEQ=sched)
        out[case["name"]] = res.numpy().copy()
        print(case["name"], res[:8].tolist())
    np.savez_compressed(os.path.join(HERE, "index.npz"), **out)


def _load_ref_test_module(relpath, name):
    import importlib.util

    spec = importlib.util.spec_from_file_location(name, os.path.join(_ref_import.REF_ROOT, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_quant():
    t_tok = _load_ref_test_module("sgl-kernel/tests/test_per_token_quant_fp8.py", "ref_t_tok")
    t_ten = _load_ref_test_module("sgl-kernel/tests/test_per_tensor_quant_fp8.py", "ref_t_ten")
    t_mm = _load_ref_test_module("sgl-kernel/tests/test_fp8_gemm.py", "ref_t_mm")
    t_awq = _load_ref_test_module("sgl-kernel/tests/test_awq_dequant.py", "ref_t_awq")
    out = {}
    for case in _cases.QUANT_CASES:
        x = _cases.build_quant_case(case)
        # the scale is what the kernel computes (per_token_quant_fp8.cu:49-57); q is the reference's restatement
        scale = x.float().abs().amax(dim=1, keepdim=True) / 448.0
        q = t_tok.torch_per_token_quant_fp8(x, scale) if not case.get("edge") else None
        if q is None:
            # the test restatement divides by zero for the all-zero row; the kernel defines scale_inv = 0 there
            safe = torch.where(scale == 0, torch.ones_like(scale), scale)
            q = t_tok.torch_per_token_quant_fp8(x, safe)
        out[case["name"] + ".tok_q"] = q.view(torch.uint8).numpy()
        out[case["name"] + ".tok_s"] = scale.numpy()
        ts = (x.float().abs().amax() / 448.0).reshape(1)
        out[case["name"] + ".ten_q"] = t_ten.torch_scaled_fp8_quant(x, ts).view(torch.uint8).numpy()
        out[case["name"] + ".ten_s"] = ts.numpy()
        static = torch.tensor([0.37], dtype=torch.float32)
        out[case["name"] + ".ten_static_q"] = t_ten.torch_scaled_fp8_quant(x, static).view(torch.uint8).numpy()
    for case in _cases.GEMM_CASES:
        c = _cases.build_gemm_case(case)
        o = t_mm.torch_scaled_mm(c["a"], c["w"].t(), c["sa"], c["sb"], c["out_dtype"], c["bias"])
        out[case["name"] + ".mm"] = _cases.bits16(o)
    for case in _cases.AWQ_CASES:
        c = _cases.build_awq_case(case)
        o = t_awq.awq_dequantize_torch(c["qweight"], c["scales"], c["qzeros"], c["g"])
        out[case["name"] + ".deq"] = _cases.bits16(o)
    np.savez_compressed(os.path.join(HERE, "quant.npz"), **out)
    print("quant.npz", len(out), "arrays")


def gen_elementwise():
    from types import SimpleNamespace

    from sglang.srt.layers.rotary_embedding import RotaryEmbedding

    ln = _ref_functions  # layernorm.py imports vllm at module import on non-CUDA hosts; take the class method text
    import ast

    path = os.path.join(_ref_import.REF_ROOT, "python/sglang/srt/layers/layernorm.py")
    tree = ast.parse(open(path).read())
    fn = None
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == "RMSNorm":
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == "forward_native":
                    fn = item
    ns = {"torch": torch, "Optional": __import__("typing").Optional, "Union": __import__("typing").Union,
          "Tuple": __import__("typing").Tuple}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    rms_forward_native = ns["forward_native"]
    out = {}
    for case in _cases.NORM_CASES:
        c = _cases.build_norm_case(case)
        self = SimpleNamespace(weight=c["w"], variance_epsilon=c["eps"], hidden_size=case["h"], variance_size_override=None)
        y = rms_forward_native(self, c["x"].clone())
        y2, r2 = rms_forward_native(self, c["x"].clone(), c["res"].clone())
        out[case["name"] + ".y"] = _cases.bits16(y)
        out[case["name"] + ".y_res"] = _cases.bits16(y2)
        out[case["name"] + ".res"] = _cases.bits16(r2)
    for case in _cases.ROPE_CASES:
        c = _cases.build_rope_case(case)
        self = SimpleNamespace(head_size=case["hs"], rotary_dim=case["rot"], max_position_embeddings=4096,
                               base=case["base"], is_neox_style=case["neox"])
        self._compute_inv_freq = lambda base, s=self: RotaryEmbedding._compute_inv_freq(s, base)
        cache = RotaryEmbedding._compute_cos_sin_cache(self)
        self.cos_sin_cache = cache
        q, k = RotaryEmbedding.forward_native(self, c["positions"], c["q"].clone(), c["k"].clone())
        out[case["name"] + ".q"] = _cases.bits16(q)
        out[case["name"] + ".k"] = _cases.bits16(k)
        out[case["name"] + ".cache_sum"] = np.array([cache.double().sum().item(), cache.double().abs().sum().item()])
        # cos/sin differ in the last ulp between host libm code paths, so the rows used travel with the fixture
        out[case["name"] + ".cache_rows"] = cache[c["positions"]].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "elementwise.npz"), **out)
    print("elementwise.npz", len(out), "arrays")


def gen_radix():
    import json

    from sglang.srt.mem_cache.allocator import TokenToKVPoolAllocator
    from sglang.srt.mem_cache.memory_pool import ReqToTokenPool
    from sglang.srt.mem_cache.radix_cache import RadixCache

    out = {}
    for page_size in (1, 4):
        for seed in (0, 1, 2):
            free_log = []
            fake_alloc = type("A", (), {"device": "cpu", "free": lambda self, idx: free_log.append([int(x) for x in idx])})()

            class Adapter:
                def __init__(self):
                    self.c = RadixCache(None, fake_alloc, page_size=page_size)

                def match_prefix(self, key):
                    r = self.c.match_prefix(key)
                    return (torch.cat([torch.as_tensor(v) for v in [r.device_indices]]).tolist(), r.last_device_node)

                def insert(self, key, vals):
                    return self.c.insert(key, torch.tensor(vals, dtype=torch.int64))

                def __getattr__(self, n):
                    return getattr(self.c, n)

            out[f"prim_p{page_size}_s{seed}"] = _cases.radix_primitive_script(Adapter, free_log, seed=seed, page_size=page_size)

    def env():
        pool = ReqToTokenPool(32, 256, "cpu", False)
        alloc = TokenToKVPoolAllocator(600, torch.bfloat16, "cpu", None)
        return RadixCache(pool, alloc, page_size=1), pool, alloc

    for seed in (0, 1):
        out[f"req_s{seed}"] = _cases.radix_request_script(env, seed=seed)
    from sglang.srt.mem_cache.allocator import PagedTokenToKVPoolAllocator

    for ps in (4, 16):
        for seed in (0, 1):
            out[f"paged_p{ps}_s{seed}"] = _cases.paged_alloc_script(
                lambda size, page: PagedTokenToKVPoolAllocator(size, page, torch.bfloat16, "cpu", None), ps, seed=seed)
    with open(os.path.join(HERE, "radix.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("radix.json", {k: len(v) for k, v in out.items()})


def _ref_method(relpath, cls_name, fn_name, ns):
    """A single method of a reference class, executed from its source text (for modules whose import needs absent
    third-party packages); nothing is copied into the repo."""
    import ast

    path = os.path.join(_ref_import.REF_ROOT, relpath)
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls_name:
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == fn_name:
                    exec(compile(ast.Module(body=[item], type_ignores=[]), path, "exec"), ns)
                    return ns[fn_name]
    raise KeyError(f"{cls_name}.{fn_name} not found in {relpath}")


def gen_model():
    """G7: a 2-layer Llama / Qwen2-shaped stack composed from the REFERENCE's own blocks --
    RMSNorm.forward_native (layernorm.py), RotaryEmbedding.forward_native (rotary_embedding.py), SiluAndMul.forward_native
    (activation.py), TorchNativeAttnBackend + RadixAttention + MHATokenToKVPool + ForwardBatch (the attention path exactly
    as gen_attention runs it), and the linear methods in the form of the reference's own torch restatements
    (sgl-kernel/tests: torch_per_token_quant_fp8, torch_scaled_fp8_quant, torch_scaled_mm, awq_dequantize_torch) wired as
    W8A8Fp8LinearMethod.apply / Fp8LinearMethod (+ requantize_with_max_scale) / AWQLinearMethod.apply do.  Layer order:
    models/llama.py:94-98,180-191,245-268,308-340.  Stores bf16 logits of one extend + MODEL_DECODE_STEPS greedy decode
    steps, the tokens chosen, and the float64 'exact' logits of the same function (oracle/model.py, exact=True)."""
    import typing
    from types import SimpleNamespace

    import torch.nn.functional as F
    from sglang.srt.layers.attention.torch_native_backend import TorchNativeAttnBackend
    from sglang.srt.layers.radix_attention import RadixAttention
    from sglang.srt.layers.rotary_embedding import RotaryEmbedding
    from sglang.srt.mem_cache.memory_pool import MHATokenToKVPool
    from sglang.srt.model_executor.forward_batch_info import ForwardBatch, ForwardMode

    from oracle.model import OracleLlama, process_checkpoint

    t_tok = _load_ref_test_module("sgl-kernel/tests/test_per_token_quant_fp8.py", "ref_t_tok")
    t_ten = _load_ref_test_module("sgl-kernel/tests/test_per_tensor_quant_fp8.py", "ref_t_ten")
    t_mm = _load_ref_test_module("sgl-kernel/tests/test_fp8_gemm.py", "ref_t_mm")
    t_awq = _load_ref_test_module("sgl-kernel/tests/test_awq_dequant.py", "ref_t_awq")
    tns = {"torch": torch, "F": F, "Optional": typing.Optional, "Union": typing.Union, "Tuple": typing.Tuple}
    rms_native = _ref_method("python/sglang/srt/layers/layernorm.py", "RMSNorm", "forward_native", dict(tns))
    silu_native = _ref_method("python/sglang/srt/layers/activation.py", "SiluAndMul", "forward_native", dict(tns))

    out = {}
    for case in _cases.MODEL_CASES:
        m = _cases.build_model_case(case)
        ck, quant = m["ckpt"], case["quant"]
        H, hq, hkv, d, I = case["hidden"], case["hq"], case["hkv"], case["d"], case["inter"]
        L = case["layers"]
        dt = torch.bfloat16

        def rms(x, w, residual=None):
            self = SimpleNamespace(weight=w, variance_epsilon=m["eps"], hidden_size=H, variance_size_override=None)
            return rms_native(self, x) if residual is None else rms_native(self, x, residual)

        rope_self = SimpleNamespace(head_size=d, rotary_dim=d, max_position_embeddings=_cases.MODEL_MAX_POS, base=m["theta"],
                                    is_neox_style=True)
        rope_self._compute_inv_freq = lambda base, s=rope_self: RotaryEmbedding._compute_inv_freq(s, base)
        rope_self.cos_sin_cache = RotaryEmbedding._compute_cos_sin_cache(rope_self)

        def linear(x, t, widths):
            bias = t.get("bias")
            if quant is None:                    # UnquantizedLinearMethod.apply
                return F.linear(x, t["weight"], bias)
            if quant == "w8a8_fp8":              # W8A8Fp8LinearMethod.apply -> apply_fp8_linear (per-token, per-channel)
                scale = x.float().abs().amax(dim=1, keepdim=True) / 448.0      # per_token_quant_fp8.cu:49-57
                safe = torch.where(scale == 0, torch.ones_like(scale), scale)
                xq = t_tok.torch_per_token_quant_fp8(x, safe)
                return t_mm.torch_scaled_mm(xq, t["weight"].t(), scale.flatten(), t["weight_scale"].flatten(), dt, bias)
            if quant == "fp8":                   # Fp8LinearMethod: requantize_with_max_scale + static per-tensor activation
                w, ws = t["weight"], t["weight_scale"]
                smax = ws.max()
                if ws.numel() > 1:
                    parts, start = [], 0
                    for i, width in enumerate(widths):
                        dq = w[start:start + width].to(torch.float16) * ws[i]          # per_tensor_dequantize (utils.py:59-64)
                        parts.append(t_ten.torch_scaled_fp8_quant(dq, smax.reshape(1)))  # scaled_fp8_quant(w_dq, max_scale)
                        start += width
                    w = torch.cat(parts)
                sx = t["input_scale"].max().reshape(1)
                xq = t_ten.torch_scaled_fp8_quant(x, sx)
                return t_mm.torch_scaled_mm(xq, w.t(), sx.expand(x.shape[0]), smax.expand(w.shape[0]), dt, bias)
            if quant == "awq":                   # AWQLinearMethod.apply (awq.py:401-418)
                wd = t_awq.awq_dequantize_torch(t["qweight"], t["scales"], t["qzeros"], case["group"])
                o = torch.matmul(x, wd.to(x.dtype))
                if bias is not None:
                    o.add_(bias)
                return o
            raise ValueError(quant)

        backend = TorchNativeAttnBackend(SimpleNamespace(device="cpu"))
        pool = MHATokenToKVPool(size=m["pool_size"], page_size=1, dtype=dt, head_num=hkv, head_dim=d, layer_num=L,
                                device="cpu", enable_memory_saver=False)
        attn_layers = [RadixAttention(num_heads=hq, head_dim=d, scaling=d ** -0.5, num_kv_heads=hkv, layer_id=i) for i in range(L)]
        r2t = torch.zeros(m["max_reqs"], m["max_ctx"], dtype=torch.int32)
        rpi = m["req_pool_indices"]

        def forward(ids, positions, fb, last_index):
            h = ck["embed"][ids]
            residual = None
            for li in range(L):
                W = ck["layers"][li]
                if residual is None:
                    residual, h = h, rms(h, W["ln1"])
                else:
                    h, residual = rms(h, W["ln1"], residual)
                qkv = linear(h, W["qkv"], [hq * d, hkv * d, hkv * d])
                q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
                q, k = RotaryEmbedding.forward_native(rope_self, positions, q, k)
                a = attn_layers[li](q, k, v, fb)
                a = linear(a, W["o"], [H])
                h, residual = rms(a, W["ln2"], residual)
                gu = linear(h, W["gate_up"], [I, I])
                h = linear(silu_native(None, gu), W["down"], [H])
            h, _ = rms(h, ck["norm"], residual)
            if last_index is not None:
                h = h[last_index]
            return F.linear(h, ck["lm_head"])

        lens = list(_cases.MODEL_LENS)
        bs = len(lens)
        slots = m["slots"]
        cur = 0
        for i, n in enumerate(lens):
            r2t[rpi[i], :n] = slots[cur:cur + n].int()
            cur += n
        ids = torch.cat(m["input_ids"])
        seq = torch.tensor(lens, dtype=torch.int64)
        loc = slots[: sum(lens)].clone()
        fb = ForwardBatch(forward_mode=ForwardMode.EXTEND, batch_size=bs, input_ids=ids, req_pool_indices=rpi, seq_lens=seq,
                          out_cache_loc=loc, seq_lens_sum=int(seq.sum()), extend_prefix_lens=torch.zeros(bs, dtype=torch.int32),
                          extend_seq_lens=torch.tensor(lens, dtype=torch.int32), attn_backend=backend)
        fb.req_to_token_pool = SimpleNamespace(req_to_token=r2t, size=r2t.shape[0])
        fb.token_to_kv_pool = pool
        backend.init_forward_metadata(fb)
        pos = torch.cat([torch.arange(n) for n in lens])
        logits = [forward(ids, pos, fb, torch.cumsum(seq, 0) - 1)]
        tokens = []
        for step in range(_cases.MODEL_DECODE_STEPS):
            nxt = torch.argmax(logits[-1].float(), dim=-1)
            tokens.append(nxt)
            loc = slots[cur:cur + bs].clone()
            cur += bs
            for i in range(bs):
                r2t[rpi[i], seq[i]] = int(loc[i])
            seq = seq + 1
            fb = ForwardBatch(forward_mode=ForwardMode.DECODE, batch_size=bs, input_ids=nxt, req_pool_indices=rpi, seq_lens=seq,
                              out_cache_loc=loc, seq_lens_sum=int(seq.sum()), attn_backend=backend)
            fb.req_to_token_pool = SimpleNamespace(req_to_token=r2t, size=r2t.shape[0])
            fb.token_to_kv_pool = pool
            backend.init_forward_metadata(fb)
            logits.append(forward(nxt, seq - 1, fb, None))
        ref = torch.stack(logits)                                     # [1 + steps, bs, V] bf16
        out[case["name"] + ".logits"] = _cases.bits16(ref)
        out[case["name"] + ".tokens"] = torch.stack(tokens).numpy()
        # the same function in exact arithmetic, teacher-forced with the reference's tokens
        cfg = _cases.model_cfg(case, m)
        ex = OracleLlama(cfg, process_checkpoint(ck, quant, [hq * d, hkv * d, hkv * d]), torch.float64, pool_slots=m["pool_size"] + 1, exact=True)
        exact = _cases.run_model_script(ex, m, torch.stack(tokens))
        out[case["name"] + ".exact"] = exact.to(torch.float32).numpy()
        err = (ref.double() - exact).abs()
        print(case["name"], "max|logit|", float(exact.abs().max()), "ref-vs-exact max", float(err.max()), "mean", float(err.mean()))
    np.savez_compressed(os.path.join(HERE, "model.npz"), **out)
    print("model.npz", len(out), "arrays")


def gen_extend_mask():
    """Custom-mask / sliding-window extend attention from the reference's own Triton kernel (extend_attention.py:41-438) on the
    CPU interpreter.  The module cannot be imported here (it pulls prefill_attention.py, which queries the GPU at import), so
    the three definitions it needs -- tanh, _fwd_kernel, extend_attention_fwd -- are executed from its source text with the
    module's platform switches set to the generic branch (64 x 64 tiles)."""
    import ast

    import triton
    import triton.language as tl

    path = os.path.join(_ref_import.REF_ROOT, "python/sglang/srt/layers/attention/triton_ops/extend_attention.py")
    tree = ast.parse(open(path).read())
    ns = {"torch": torch, "triton": triton, "tl": tl, "_is_hip": False, "_is_cuda": False, "CUDA_CAPABILITY": (0, 0)}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("tanh", "_fwd_kernel", "extend_attention_fwd"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    fwd = ns["extend_attention_fwd"]
    out = {}
    for case in _cases.MASK_CASES:
        c = _cases.build_mask_case(case)
        q = c["q"]
        o = torch.zeros_like(q)
        fwd(q, c["k_extend"], c["v_extend"], o, c["k_buffer"], c["v_buffer"], c["qo_indptr"], c["kv_indptr"], c["kv_indices"],
            c["custom_mask"], True, c["mask_indptr"], c["max_len_extend"], c["scaling"], 0.0, c["skip_prefix"], c["window"])
        assert torch.isfinite(o.float()).all()
        out[case["name"]] = _cases.bits16(o[_cases.mask_rows(c)])
        print(case["name"], tuple(o.shape), float(o.float().abs().max()))
    np.savez_compressed(os.path.join(HERE, "extend_mask.npz"), **out)


def gen_interface():
    """Interface snapshot: names, parameter order and defaults of the plugin surfaces this build mirrors,
    read from the reference's source with ``ast`` (signatures only -- data, no code), plus the two native attention op
    schemas (string literals of torch_extension_cpu.cpp).  tests/test_interface_snapshot.py compares this build with it."""
    import ast
    import json
    import re

    ref = _ref_import.REF_ROOT

    def sig(fn):
        a = fn.args
        pos = a.posonlyargs + a.args
        defaults = [None] * (len(pos) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
        params = [[p.arg, d] for p, d in zip(pos, defaults)]
        if a.vararg:
            params.append(["*" + a.vararg.arg, None])
        for p, d in zip(a.kwonlyargs, a.kw_defaults):
            params.append([p.arg, None if d is None else ast.unparse(d)])
        if a.kwarg:
            params.append(["**" + a.kwarg.arg, None])
        return params

    def module(rel):
        return ast.parse(open(os.path.join(ref, rel)).read())

    def class_methods(rel, cls, only=None):
        for node in module(rel).body:
            if isinstance(node, ast.ClassDef) and node.name == cls:
                return {f.name: sig(f) for f in node.body if isinstance(f, ast.FunctionDef) and (only is None or f.name in only)}
        raise KeyError(cls)

    def functions(rel, names):
        return {f.name: sig(f) for f in module(rel).body if isinstance(f, ast.FunctionDef) and f.name in names}

    def dataclass_fields(rel, cls):
        for node in module(rel).body:
            if isinstance(node, ast.ClassDef) and node.name == cls:
                return [[st.target.id, ast.unparse(st.annotation), None if st.value is None else ast.unparse(st.value)]
                        for st in node.body if isinstance(st, ast.AnnAssign) and isinstance(st.target, ast.Name)]
        raise KeyError(cls)

    def enum_members(rel, cls):
        for node in module(rel).body:
            if isinstance(node, ast.ClassDef) and node.name == cls:
                return [st.targets[0].id for st in node.body if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name)]
        raise KeyError(cls)

    cpu_ext = open(os.path.join(ref, "sgl-kernel/csrc/cpu/torch_extension_cpu.cpp")).read()
    schemas = {}
    for op in ("decode_attention_cpu", "extend_attention_cpu"):
        mm = re.search(r'"(' + op + r'\([^;]*?->\s*\(\))"', cpu_ext.replace('"\n', '"').replace('\n', ' '), flags=re.S)
        text = re.sub(r'"\s*"', "", mm.group(1))
        schemas[op] = " ".join(text.split())
    ext = open(os.path.join(ref, "sgl-kernel/csrc/common_extension.cc")).read()
    ext = re.sub(r'"\s*\n\s*"', "", ext)   # join adjacent string literals
    torch_ops = {}
    for op in ("fp8_scaled_mm", "sgl_per_token_group_quant_fp8", "sgl_per_tensor_quant_fp8", "sgl_per_token_quant_fp8",
               "awq_dequantize", "merge_state", "merge_state_v2", "rmsnorm", "fused_add_rmsnorm", "silu_and_mul"):
        mm = re.search(r'm\.def\(\s*"(' + op + r'\([^"]*)"', ext)
        torch_ops[op] = " ".join(mm.group(1).split())
    snap = {
        "torch_op_schemas": torch_ops,
        "AttentionBackend": class_methods("python/sglang/srt/layers/attention/base_attn_backend.py", "AttentionBackend"),
        "TritonAttnBackend": class_methods("python/sglang/srt/layers/attention/triton_backend.py", "TritonAttnBackend",
                                           {"__init__", "init_forward_metadata", "init_cuda_graph_state",
                                            "init_forward_metadata_capture_cuda_graph", "init_forward_metadata_replay_cuda_graph",
                                            "get_cuda_graph_seq_len_fill_value", "forward_extend", "forward_decode"}),
        "ForwardBatch.fields": dataclass_fields("python/sglang/srt/model_executor/forward_batch_info.py", "ForwardBatch"),
        "ForwardMode.members": enum_members("python/sglang/srt/model_executor/forward_batch_info.py", "ForwardMode"),
        "ForwardMode.methods": sorted(class_methods("python/sglang/srt/model_executor/forward_batch_info.py", "ForwardMode")),
        "QuantizeMethodBase": class_methods("python/sglang/srt/layers/quantization/base_config.py", "QuantizeMethodBase"),
        "LinearMethodBase": class_methods("python/sglang/srt/layers/quantization/base_config.py", "LinearMethodBase"),
        "QuantizationConfig": class_methods("python/sglang/srt/layers/quantization/base_config.py", "QuantizationConfig"),
        "RadixAttention": class_methods("python/sglang/srt/layers/radix_attention.py", "RadixAttention", {"__init__", "forward"}),
        "ReqToTokenPool": class_methods("python/sglang/srt/mem_cache/memory_pool.py", "ReqToTokenPool"),
        "MHATokenToKVPool": class_methods("python/sglang/srt/mem_cache/memory_pool.py", "MHATokenToKVPool",
                                          {"__init__", "get_key_buffer", "get_value_buffer", "get_kv_buffer", "set_kv_buffer"}),
        "TokenToKVPoolAllocator": class_methods("python/sglang/srt/mem_cache/allocator.py", "TokenToKVPoolAllocator"),
        "RadixCache": class_methods("python/sglang/srt/mem_cache/radix_cache.py", "RadixCache",
                                    {"__init__", "reset", "match_prefix", "insert", "cache_finished_req", "cache_unfinished_req",
                                     "evict", "inc_lock_ref", "dec_lock_ref", "evictable_size", "protected_size"}),
        "sgl_kernel.gemm": functions("sgl-kernel/python/sgl_kernel/gemm.py",
                                     {"awq_dequantize", "fp8_scaled_mm", "sgl_per_token_group_quant_fp8", "sgl_per_tensor_quant_fp8",
                                      "sgl_per_token_quant_fp8"}),
        "sgl_kernel.attention": functions("sgl-kernel/python/sgl_kernel/attention.py", {"merge_state", "merge_state_v2"}),
        "sgl_kernel.elementwise": functions("sgl-kernel/python/sgl_kernel/elementwise.py",
                                            {"rmsnorm", "fused_add_rmsnorm", "silu_and_mul", "apply_rope_with_cos_sin_cache_inplace"}),
        "triton_ops": {**functions("python/sglang/srt/layers/attention/triton_ops/decode_attention.py", {"decode_attention_fwd"}),
                       **functions("python/sglang/srt/layers/attention/triton_ops/extend_attention.py", {"extend_attention_fwd"})},
        "apply_fp8_linear": functions("python/sglang/srt/layers/quantization/fp8_utils.py", {"apply_fp8_linear"})["apply_fp8_linear"],
        "fp8_helpers": {**functions("python/sglang/srt/layers/quantization/fp8_kernel.py", {"static_quant_fp8"}),
                        **functions("python/sglang/srt/layers/quantization/fp8_utils.py", {"input_to_float8"})},
        "cpu_op_schemas": schemas,
    }
    with open(os.path.join(HERE, "interface.json"), "w") as f:
        json.dump(snap, f, indent=1, sort_keys=True)
    print("interface.json", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in snap.items()})


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.manual_seed(0)
    if what in ("attention", "all"):
        gen_attention()
    for extra in ("index", "quant", "elementwise", "radix", "model", "interface", "extend_mask"):
        fn = globals().get("gen_" + extra)
        if fn is not None and what in (extra, "all"):
            fn()
