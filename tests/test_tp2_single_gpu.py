"""Tensor-parallel rehearsal of the device path on ONE GPU: two ranks (two processes on cuda:0) whose collectives run over
gloo through host copies (communication_op._HOST_STAGED).  Everything except the RCCL transport is the N > 1 code path of
bench.py: sharded weights, per-rank KV heads, the fused decode / prefill path with its two all-reduce sites per layer
(linear.py:1302-1303) and the logits all-gather."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_model(fused, feed=None):
    """feed: the token ids to decode with (so that every run follows the same sequence); None = own greedy choices."""
    from ltp_sglang_amd.srt.model_executor.synthetic_llama import LlamaShape, SyntheticModelRunner

    cfg = LlamaShape(hidden_size=1024, num_attention_heads=8, num_key_value_heads=2, head_dim=128, num_hidden_layers=3,
                     intermediate_size=3584, vocab_size=4096, max_position_embeddings=512)
    runner = SyntheticModelRunner(cfg, "w8a8_fp8", max_running_requests=8, context_len=256, max_total_tokens=2048,
                                  device="cuda:0", seed=5)
    runner.model.fused_decode = runner.model.fused_extend = fused
    g = torch.Generator().manual_seed(1)
    ids = [torch.randint(0, cfg.vocab_size, (n,), generator=g).to("cuda:0") for n in (50, 7, 33, 1)]
    logits, state = runner.extend(ids)
    out, fed = [logits.clone()], []
    for step in range(3):
        nxt = torch.argmax(logits.float(), dim=-1) if feed is None else torch.tensor(feed[step], device="cuda:0")
        fed.append(nxt.tolist())
        logits = runner.decode(state, nxt)
        out.append(logits.clone())
    return torch.stack(out).float().cpu(), fed


def _worker(rank, world, port, q, feed):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package

    load_package()
    from ltp_sglang_amd.srt.distributed import communication_op as comm

    comm.init_tensor_parallel()
    assert comm.get_tensor_model_parallel_world_size() == world
    plain, _ = _run_model(False, feed)
    fused, _ = _run_model(True, feed)
    if rank == 0:
        q.put((plain.numpy(), fused.numpy()))  # by value: the parent may read after this process has exited
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])   # 4: the 2 kv heads are replicated across rank pairs (llama.py:118-133)
def test_tp_fused_path_matches_plain_and_tp1(world, pkg):
    tp1, feed = _run_model(True)          # this process: no process group -> tp 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, feed)) for r in range(world)]
    for p in procs:
        p.start()
    plain, fused = (torch.from_numpy(x) for x in q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # same sharding, same all-reduce sites, bit-identical fused kernels -> identical logits
    assert torch.equal(plain, fused)
    # TP 2 vs TP 1: the row-parallel weights are quantised per shard (another per-channel grid) and the partial sums are
    # rounded to bf16 before the all-reduce, so agreement is at the fp8-stack tolerance of test_model_gpu.py, not bitwise
    err = (fused - tp1).abs()
    assert err.max().item() <= 2e-1 and err.mean().item() <= 2.5e-2, (err.max().item(), err.mean().item())  # measured 0.10 / 0.016
