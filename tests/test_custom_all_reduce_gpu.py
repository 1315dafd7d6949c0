"""One-shot P2P all-reduce (csrc/custom_all_reduce.hip, srt/distributed/custom_all_reduce.py) validated with 2 and 4
PROCESSES ON ONE GPU: every rank maps its peers' buffers through hipIpcGetMemHandle / hipIpcOpenMemHandle -- the code path an
8-GPU node runs over xGMI -- and the result is compared BIT-EXACTLY with the host-staged sum in rank order (f32 accumulate,
one rounding), eagerly and replayed from a HIP graph.  (The box allows at most 6 processes on its GPU: no 8-rank case here.)
Reference: sgl-kernel/csrc/allreduce/custom_all_reduce_hip.cuh:261-294, python/sglang/srt/distributed/device_communicators/
custom_all_reduce.py; test model: sgl-kernel/tests/test_custom_allreduce.py (compares with NCCL all_reduce)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [((32, 4096), torch.bfloat16), ((128, 8192), torch.bfloat16), ((1, 8), torch.float16), ((5, 1000), torch.float32),
          ((32, 512), torch.bfloat16), ((4096, 1024), torch.bfloat16)]   # 256 KiB / 2 MiB = the two TP configs; 8 MiB = max_size


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package

    load_package()
    from ltp_sglang_amd.srt.distributed import communication_op as comm
    from ltp_sglang_amd.srt.distributed.custom_all_reduce import CustomAllreduce

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    # a failure on ONE rank (here: rank 1 passes a size the C-ABI rejects) must leave EVERY rank disabled, with the reason, and
    # nobody stuck in the handle exchange
    bad = CustomAllreduce(dist.group.WORLD, dev, max_size=-16 if rank == 1 else 8 * 1024 * 1024)
    assert bad.disabled and "rank 1" in bad.disabled_reason, bad.disabled_reason
    assert not bad.should_use(torch.zeros(64, device=dev))
    car = CustomAllreduce(dist.group.WORLD, dev)
    assert not car.disabled and car.disabled_reason == ""
    ok, worst = True, 0.0
    for it, (shape, dtype) in enumerate(SHAPES * 2):   # twice: both data halves of every size, epochs keep counting
        g = torch.Generator().manual_seed(1000 * it + rank)
        x = (torch.randn(shape, generator=g) * 3).to(dtype)
        # host-staged reference: every rank's operand gathered over gloo, summed in rank order in f32, rounded once
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x)
        ref = torch.zeros(shape, dtype=torch.float32)
        for part in parts:
            ref += part.float()
        ref = ref.to(dtype)
        xd = x.to(dev)
        assert car.should_use(xd)
        out = car.all_reduce(xd)
        car.check_error()
        assert out.data_ptr() == xd.data_ptr()
        same = torch.equal(out.cpu(), ref)
        if not same:
            bad = (out.cpu() != ref)
            print(f"[rank {rank}] call {it} {shape} {dtype}: {int(bad.sum())} of {bad.numel()} elements differ; first at {bad.flatten().nonzero()[:4].flatten().tolist()}",
                  flush=True)
        ok = ok and same
        worst = max(worst, float((out.cpu().float() - ref.float()).abs().max()))
    # back-to-back calls without host synchronisation in between (a rank may run one call ahead of its peers)
    xs = [torch.full((32, 4096), float(rank + 1 + k), dtype=torch.bfloat16, device=dev) for k in range(16)]
    for t in xs:
        car.all_reduce(t)
    car.check_error()
    for k, t in enumerate(xs):
        ok = ok and bool((t == float(sum(r + 1 + k for r in range(world)))).all())
    # through the reference's call site: tensor_model_parallel_all_reduce prefers the custom path for small messages
    comm.init_tensor_parallel()
    comm.set_custom_all_reduce(car)
    y = torch.full((32, 4096), float(rank + 1), dtype=torch.bfloat16, device=dev)
    y = comm.tensor_model_parallel_all_reduce(y)
    car.check_error()
    ok = ok and bool((y == float(world * (world + 1) // 2)).all())
    # all-reduce fused with add + RMSNorm + per-token quant == the unfused pair, bit for bit (both TP all-reduce sites of a layer)
    from ltp_sglang_amd import sgl_kernel as K

    for (m, h), dtype in (((32, 4096), torch.bfloat16), ((7, 1024), torch.float16), ((128, 8192), torch.bfloat16), ((70, 512), torch.bfloat16)):
        g = torch.Generator().manual_seed(31 * m + rank)
        part = torch.randn(m, h, generator=g).to(dtype).to(dev)
        g2 = torch.Generator().manual_seed(5)   # residual and weight are replicated across ranks
        res0 = torch.randn(m, h, generator=g2).to(dtype).to(dev)
        w = (1 + 0.1 * torch.randn(h, generator=g2)).to(dtype).to(dev)
        summed = car.all_reduce(part.clone())
        res_a = res0.clone()
        ya, qa, sa = K.fused_add_rmsnorm_quant_fp8(summed, res_a, w, 1e-5, want_norm=True, want_quant=True)
        res_b = res0.clone()
        yb, qb, sb = comm.tensor_model_parallel_all_reduce_add_rmsnorm_quant(part.clone(), res_b, w, 1e-5, want_norm=True, want_quant=True)
        car.check_error()
        same = (torch.equal(ya, yb) and torch.equal(qa.view(torch.uint8), qb.view(torch.uint8)) and torch.equal(sa, sb)
                and torch.equal(res_a, res_b))
        if not same:
            print(f"[rank {rank}] fused all-reduce + norm differs at {(m, h)} {dtype}", flush=True)
        ok = ok and same
        yc, _, _ = comm.tensor_model_parallel_all_reduce_add_rmsnorm_quant(part.clone(), None, w, 1e-5, want_norm=True, want_quant=False)
        yd, _, _ = K.fused_add_rmsnorm_quant_fp8(summed, None, w, 1e-5, want_norm=True, want_quant=False)
        ok = ok and torch.equal(yc, yd)
    # all-gather along the last dimension (the vocab-sharded logits), through the call site as well
    for shape in ((32, 16032), (3, 8), (128, 2048)):
        part = torch.randn(shape, generator=torch.Generator().manual_seed(7 + rank)).to(torch.bfloat16)
        parts = [torch.empty_like(part) for _ in range(world)]
        dist.all_gather(parts, part)
        got = comm.tensor_model_parallel_all_gather(part.to(dev))
        car.check_error()
        ok = ok and torch.equal(got.cpu(), torch.cat(parts, dim=-1))
    # HIP graph: capture one all-reduce, replay it on fresh operands
    z = torch.zeros((32, 4096), dtype=torch.bfloat16, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        z.fill_(float(rank))
        car.all_reduce(z)          # warm-up launch on the side stream (same count on every rank)
        torch.cuda.synchronize()
        dist.barrier()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            car.all_reduce(z)
    for k in range(5):
        z.fill_(float(rank + k))
        torch.cuda.synchronize()
        dist.barrier()
        graph.replay()
        car.check_error()
        ok = ok and bool((z == float(sum(r + k for r in range(world)))).all())
    comm.set_custom_all_reduce(None)
    car.close()
    if rank == 0:
        q.put((ok, worst))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_one_shot_all_reduce_bit_exact_over_ipc(world, pkg):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, worst = q.get(timeout=150)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok, f"custom all-reduce differs from the host-staged rank-order sum (max |diff| {worst})"
