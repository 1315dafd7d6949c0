"""CPU: the C-ABI shared library loads, exports every entry point include/sgl_mi355.h declares, and reports argument
errors through the status + thread-local message convention (no kernel is launched here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    from ltp_sglang_amd import _cabi

    protos = _cabi.parse_header()
    names = [p[0] for p in protos]
    text = open(_cabi.HEADER_PATH).read()
    declared = set(re.findall(r"\b(sgl_mi355_\w+)\s*\(", text))
    assert declared == set(names) and len(names) >= 40
    raw = ctypes.CDLL(_cabi.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in the header but not exported by the library"
    assert _cabi.lib.sgl_mi355_abi_version() == 1


def test_every_entry_point_cites_the_reference():
    text = open(os.path.join(ROOT, "include", "sgl_mi355.h")).read()
    assert text.count("sgl-kernel/") + text.count("python/sglang/") + text.count(".py:") >= 25


def test_error_convention_status_and_message(pkg):
    from ltp_sglang_amd import _cabi

    lib = _cabi.lib
    rc = lib.sgl_mi355_per_token_quant_fp8(ctypes.c_void_p(16), 24, ctypes.c_void_p(16), ctypes.c_void_p(16), 4, 20, 0, None)
    assert rc == 1 and "divisible by 8" in _cabi.last_error()          # reference message, per_token_quant_fp8.cu:173
    with pytest.raises(RuntimeError, match="divisible by 8"):
        _cabi.check(rc)
    rc = lib.sgl_mi355_decode_attention(None, 0, None, None, 0, 0, 0, 0, None, 0, None, None, None, 0, None, None, None, None,
                                        None, 4, 2, 8, 2, 128, 128, 1.0, 0.0, 0, 0, 1.0, 1.0, None)
    assert rc == 1 and "null" in _cabi.last_error()
    rc = lib.sgl_mi355_skinny_gemm(ctypes.c_void_p(16), 64, ctypes.c_void_p(16), 64, ctypes.c_void_p(16), 64, None, None, None,
                                   128, 8, 64, 3, 0, None, 0, None)
    assert rc == 1 and "exceeds 64" in _cabi.last_error()
    assert lib.sgl_mi355_decode_attention(None, 0, None, None, 0, 0, 0, 0, None, 0, None, None, None, 0, None, None, None,
                                          None, None, 4, 0, 8, 2, 128, 128, 1.0, 0.0, 0, 0, 1.0, 1.0, None) == 0  # empty batch: no-op


def test_product_has_no_oracle_or_cpu_fallback():
    """The product package never imports oracle/ and fails loudly without the HIP library."""
    pkg_dir = os.path.join(ROOT, "ltp-sglang_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"
    src = open(os.path.join(pkg_dir, "_cabi.py")).read()
    assert "raise ImportError" in src and "no fallback" in src


def test_forward_mode_and_backend_dispatch(pkg):
    import torch

    from ltp_sglang_amd.srt.layers.attention.base_attn_backend import AttentionBackend
    from ltp_sglang_amd.srt.model_executor.forward_batch_info import ForwardMode

    assert [m.value for m in (ForwardMode.EXTEND, ForwardMode.DECODE, ForwardMode.MIXED, ForwardMode.IDLE)] == [1, 2, 3, 4]
    assert ForwardMode.MIXED.is_extend() and ForwardMode.TARGET_VERIFY.is_extend() and not ForwardMode.DECODE.is_extend()
    assert ForwardMode.IDLE.is_decode_or_idle() and ForwardMode.DECODE.is_cuda_graph()

    calls = []

    class B(AttentionBackend):
        def init_forward_metadata(self, fb):
            pass

        def forward_decode(self, q, k, v, layer, fb, save_kv_cache=True):
            calls.append("decode")

        def forward_extend(self, q, k, v, layer, fb, save_kv_cache=True):
            calls.append("extend")

    from types import SimpleNamespace

    layer = SimpleNamespace(tp_q_head_num=4, v_head_dim=8)
    q = torch.zeros(3, 32)
    b = B()
    for mode in (ForwardMode.DECODE, ForwardMode.EXTEND, ForwardMode.MIXED):
        b.forward(q, None, None, layer, SimpleNamespace(forward_mode=mode))
    assert calls == ["decode", "extend", "extend"]
    assert b.forward(q, None, None, layer, SimpleNamespace(forward_mode=ForwardMode.IDLE)).shape == (3, 32)
    assert b.support_triton() is True


def test_fp8_gemm_kernel_choice_and_k_partition_are_host_logic(pkg):
    """Which tile fp8_scaled_mm takes above M = 64 and into how many k-ranges it splits is decided on the host (csrc/tiled_gemm.hip
    takes128s / splits128s: two measured cost lines); sgl_mi355_fp8_gemm_num_slabs reports it without touching a GPU (256 CUs assumed
    when no device is present).  Pinned here: the Llama-3-70B TP-8 shard shapes at M = 128, the 256 < M <= 1024 band, prefill."""
    from ltp_sglang_amd import _cabi

    f = _cabi.lib.sgl_mi355_fp8_gemm_num_slabs
    ws = 1 << 24
    want = {(128, 1280, 8192): 8, (128, 8192, 1024): 1, (128, 7168, 8192): 4, (128, 8192, 3584): 3,   # streaming tile, split-K by tiles < CUs
            (512, 1024, 4096): 4, (320, 4096, 2048): 2,                                             # the band where the 256x256 tile would idle CUs
            (1024, 4096, 2048): 1, (512, 28672, 4096): 1, (1024, 4096, 14336): 1,                   # one k-range (either tile)
            (65536, 28672, 4096): 1, (16384, 6144, 4096): 1}                                        # prefill: the 256x256 tile
    got = {k: int(f(k[0], k[1], k[2], ws)) for k in want}
    assert got == want
    assert int(f(128, 1280, 8192, 0)) == 1          # no workspace: no split
    assert int(f(0, 128, 128, ws)) == 1 and int(f(128, 128, 100, ws)) == 1   # degenerate / K not whole 128-byte slices
    # the tile itself: 0 = 256x256, 1 = streaming 128x128, 2 = 256x128 (chunked-prefill band: 256x256 tiles fewer than CUs, half-size tiles fill them)
    t = _cabi.lib.sgl_mi355_fp8_gemm_tile_choice
    want_t = {(64 + 1, 4096, 4096): 1, (256, 28672, 4096): 1, (1024, 4096, 14336): 1, (1024, 4096, 4096): 1,
              (2048, 4096, 4096): 2, (2048, 4096, 14336): 2, (1536, 4096, 4096): 2, (1024, 6144, 4096): 2, (8192, 1024, 8192): 2,
              (4096, 6144, 4096): 2, (520, 8192, 8192): 2,
              (2048, 6144, 4096): 0, (2048, 28672, 4096): 0, (4096, 4096, 4096): 0, (8192, 6144, 4096): 0,
              (65536, 28672, 4096): 0, (65536, 4096, 14336): 0, (2048, 4096, 4000): 0}
    got_t = {k: int(t(k[0], k[1], k[2], ws)) for k in want_t}
    assert got_t == want_t
