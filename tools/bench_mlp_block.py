"""MLP half of a Llama-3-8B w8a8 decode layer at M = 32: the persistent one-launch block (csrc/mlp_block.hip) against the
four-launch sequence it replaces, graph-captured over COPIES distinct weight sets (cold weights, as in the model), plus
the block's in-kernel timeline (--timeline): where each phase starts on the slowest / median workgroup."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package

load_package()
from ltp_sglang_amd import sgl_kernel as K

DEV = "cuda:0"


def graph_time(fn, iters=20):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.current_stream().wait_stream(st)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        fn()
    gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2] * 1e3


def main():
    m, hidden, inter, copies = 32, 4096, 14336, 8
    if "--m" in sys.argv:
        m = int(sys.argv[sys.argv.index("--m") + 1])
    if "--inter" in sys.argv:   # e.g. 1792: the Llama-3-8B shard at TP 8
        inter = int(sys.argv[sys.argv.index("--inter") + 1])
        copies = 32 if inter <= 4096 else copies   # keep the weight sets beyond the caches: 32 x 22 MB
    dt = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(0)
    x = (torch.randn(m, hidden, device=DEV, generator=g) * 0.7).to(dt)
    res0 = torch.randn(m, hidden, device=DEV, generator=g).to(dt)
    lnw = torch.ones(hidden, device=DEV, dtype=dt)
    ws = []
    for _ in range(copies):
        wgu = (torch.randn(2 * inter, hidden, device=DEV, generator=g) * 0.6).clamp(-3, 3).to(torch.float8_e4m3fn)
        wd = (torch.randn(hidden, inter, device=DEV, generator=g) * 0.6).clamp(-3, 3).to(torch.float8_e4m3fn)
        sgu = torch.rand(2 * inter, device=DEV, generator=g) * 0.02 + 0.005
        wi, wdd = K.fp8_mlp_block_pack_weights(K.interleave_gate_up_rows(wgu.view(torch.uint8), 16).view(torch.float8_e4m3fn), wd)
        del wgu, wd
        ws.append((wi, K.interleave_gate_up_rows(sgu, 16), wdd))
    scratch = K.Fp8MlpBlockScratch(m, hidden, inter, copies, DEV)
    res = res0.clone()

    def four_launch():
        for wi, si, wd in ws:
            _, hq, hs = K.fused_add_rmsnorm_quant_fp8(x, res, lnw, 1e-5)
            act = K.fp8_gemm_silu_mul(hq, hs.view(-1), wi, si, dt, 16)
            aq, asc = K.sglang_per_token_quant_fp8(act) if hasattr(K, "sglang_per_token_quant_fp8") else quant(act)
            K.fp8_linear_slabs(aq, wd, m, hidden, inter)

    def quant(act):
        aq = torch.empty(act.shape, dtype=torch.float8_e4m3fn, device=DEV)
        a_s = torch.empty((act.shape[0], 1), dtype=torch.float32, device=DEV)
        K.sgl_per_token_quant_fp8(act, aq, a_s)
        return aq, a_s

    def block():
        scratch.reset()
        for layer, (wi, si, wd) in enumerate(ws):
            K.fp8_mlp_block(x, res, lnw, 1e-5, wi, si, wd, scratch, layer)

    t4 = graph_time(four_launch) / copies
    tb = graph_time(block) / copies
    assert int(scratch.error_codes().abs().sum()) == 0, scratch.error_codes().tolist()
    mb = (2 * inter * hidden + hidden * inter) / 1e6
    print(f"M={m} hidden={hidden} inter={inter}: four launches {t4:6.1f} us   persistent block {tb:6.1f} us   ({mb:.0f} MB of weights: {mb / tb / 1e3 * 1e3:.0f} GB/s -> "
          f"{mb / tb * 1e6 / 1e12:.2f} TB/s; at 8 TB/s {mb / 8e6 * 1e6 / 1e6:.1f} us)")

    if "--timeline" in sys.argv:
        sc = K.Fp8MlpBlockScratch(m, hidden, inter, 1, DEV, timeline=True)
        wi, si, wd = ws[0]
        for rep in range(3):  # the last repetition is the one read (weights of copy 0 are cold again after the copies in between)
            for w2 in ws[1:]:
                K.fp8_linear_slabs(torch.zeros(m, inter, dtype=torch.float8_e4m3fn, device=DEV), w2[2], m, hidden, inter)
            sc.reset()
            sc.timeline.zero_()
            K.fp8_mlp_block(x, res, lnw, 1e-5, wi, si, wd, sc, 0)
        torch.cuda.synchronize()
        tl = sc.timeline.cpu().double() * 0.01  # us
        t0 = tl[:, :, 0][tl[:, :, 0] > 0].min()
        names = {0: "entry", 9: "row published (row workgroups, wave 7)", 1: "hand-off 1 passed", 2: "X1 fragments built",
                 3: "gate_up done", 4: "row maxima gathered (wave 7)", 5: "act published (wave 7)", 6: "hand-off 3 passed",
                 7: "X2 fragments built", 8: "down_proj done"}
        print("in-kernel timeline, us after the first workgroup's entry: min / median / max over workgroups (waves that stamp)")
        for i in (0, 9, 1, 2, 3, 4, 5, 6, 7, 8):
            v = tl[:, :, i]
            v = v[v > 0] - t0
            if v.numel():
                print(f"  {names[i]:45s} {v.min():7.2f} {v.median():7.2f} {v.max():7.2f}")


if __name__ == "__main__":
    main()
