"""Kernel-level sweep of fp8_scaled_mm over the reference's own shape tables and of awq_dequantize over its dequant table, with a
roofline fraction per row (SURVEY.md 8d "GEMM shape table"):
  * fp8: WEIGHT_SHAPES of sgl-kernel/benchmark/bench_fp8_gemm.py:19-52 (Llama-3.1-8B, Llama-3.3-70B, Qwen2.5-7B; [K, N]) x
    M in {1, 16, 64, 128, 256, 512, 1024, 2048} (:72-76);
  * awq_dequantize: qweight rows x cols of bench_awq_dequant.py:59-62.
Each row: graph-captured launches over distinct weight copies (cold weights), median of 10 replays; bound = the larger of
weight-bytes / 8 TB/s (HBM) and 2 M N K / 5 PFLOP/s (dense fp8 MFMA); frac = bound time / measured time.
    python tools/gemm_sweep.py > profiles/roundN_gemm_sweep.csv"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package

load_package()
from ltp_sglang_amd import sgl_kernel as K

DEV = "cuda:0"
FP8_SHAPES = {
    "Llama-3.1-8B": [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)],
    "Llama-3.3-70B": [(8192, 10240), (8192, 8192), (8192, 57344), (28672, 8192)],
    "Llama-3.3-70B TP8 shard": [(8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192)],
    "Qwen2.5-7B": [(3584, 4608), (3584, 3584), (3584, 37888), (18944, 3584)],
}
MS = [1, 16, 64, 128, 256, 512, 1024, 2048]
AWQ_ROWS, AWQ_COLS = [3584, 18944, 128, 256, 512, 1024], [448, 576, 4736, 16, 32, 64, 128]


def timed(fns, iters=10):
    for f in fns:
        f()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for f in fns:
            f()
    gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / len(fns))
    ts.sort()
    return ts[len(ts) // 2] * 1e3  # us


def main():
    print("op,model,M,N,K,us,weight_TBps,TFLOPs,bound,frac_of_roofline")
    for model, shapes in FP8_SHAPES.items():
        for k, n in shapes:
            copies = max(2, min(8, int(1.2e9 // (n * k))))
            ws = [torch.randn(n, k, device=DEV).clamp(-3, 3).to(torch.float8_e4m3fn) for _ in range(copies)]
            sb = torch.rand(n, device=DEV)
            for m in MS:
                x = torch.randn(m, k, device=DEV).to(torch.float8_e4m3fn)
                sa = torch.rand(m, device=DEV)
                us = timed([(lambda w=w: K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)) for w in ws])
                t_hbm, t_mfma = n * k / 8e12 * 1e6, 2.0 * m * n * k / 5e15 * 1e6
                bound, tb = ("hbm", t_hbm) if t_hbm >= t_mfma else ("mfma", t_mfma)
                print(f"fp8_scaled_mm,{model},{m},{n},{k},{us:.1f},{n * k / us / 1e6:.2f},{2.0 * m * n * k / us / 1e6:.1f},{bound},{tb / us:.3f}", flush=True)
            del ws
            torch.cuda.empty_cache()
    for rows in AWQ_ROWS:
        for cols in AWQ_COLS:
            g = 128 if rows % 128 == 0 else rows
            copies = max(2, min(8, int(4e8 // (rows * cols * 4))))
            cases = []
            for _ in range(copies):
                qw = torch.randint(-2**31, 2**31 - 1, (rows, cols), dtype=torch.int32, device=DEV)
                qz = torch.randint(-2**31, 2**31 - 1, (rows // g, cols), dtype=torch.int32, device=DEV)
                sc = (torch.rand(rows // g, cols * 8, device=DEV) * 0.01).to(torch.float16)
                cases.append((qw, sc, qz))
            us = timed([(lambda c=c: K.awq_dequantize(*c)) for c in cases])
            byts = rows * cols * 4 + rows * cols * 8 * 2   # int4 read + f16 written
            print(f"awq_dequantize,-,{rows},{cols * 8},-,{us:.1f},{byts / us / 1e6:.2f},-,hbm,{byts / 8e12 * 1e6 / us:.3f}", flush=True)


if __name__ == "__main__":
    main()
