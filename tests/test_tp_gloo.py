"""CPU, world_size 2, gloo: the tensor-parallel wiring of the path -- column/row/QKV sharding (the reference's
Megatron split, linear.py / llama.py:118-133) and the ONE collective site (all-reduce after a row-parallel linear,
linear.py:1302-1303) plus the logits all-gather.  The device kernels cannot run here, so the linear method used is a
test-only F.linear stand-in; what is under test is the host logic that the RCCL run at N > 1 relies on."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _TorchLinearMethod:
    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size, params_dtype, **kw):
        layer.register_parameter("weight", torch.nn.Parameter(torch.empty(sum(output_partition_sizes), input_size_per_partition, dtype=params_dtype), requires_grad=False))

    def apply(self, layer, x, bias=None):
        return F.linear(x, layer.weight, bias)


class _TorchQuantConfig:
    def get_quant_method(self, layer, prefix=""):
        return _TorchLinearMethod()


def _worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from __graft_entry__ import load_package

    load_package()
    from ltp_sglang_amd.srt.distributed import communication_op as comm
    from ltp_sglang_amd.srt.layers.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear

    comm.init_tensor_parallel()
    assert comm.get_tensor_model_parallel_world_size() == world and comm.get_tensor_model_parallel_rank() == rank
    g = torch.Generator().manual_seed(0)  # same full weights on every rank
    hidden, hq, hkv, hs, inter, m = 64, 8, 2, 16, 96, 5
    x = torch.randn(m, hidden, generator=g)
    w_qkv = torch.randn((hq + 2 * hkv) * hs, hidden, generator=g)
    w_o = torch.randn(hidden, hq * hs, generator=g)
    w_gu = torch.randn(2 * inter, hidden, generator=g)
    w_down = torch.randn(hidden, inter, generator=g)
    qc = _TorchQuantConfig()
    qkv = QKVParallelLinear(hidden, hs, hq, hkv, quant_config=qc, params_dtype=torch.float32)
    o = RowParallelLinear(hq * hs, hidden, quant_config=qc, params_dtype=torch.float32)
    gu = MergedColumnParallelLinear(hidden, [inter, inter], quant_config=qc, params_dtype=torch.float32)
    down = RowParallelLinear(inter, hidden, bias=True, quant_config=qc, params_dtype=torch.float32)
    qkv.weight.copy_(qkv.shard_rows(w_qkv))
    o.weight.copy_(o.shard_cols(w_o))
    gu.weight.copy_(gu.shard_rows(w_gu))
    down.weight.copy_(down.shard_cols(w_down))
    b_down = torch.randn(hidden, generator=g)
    down.bias.copy_(b_down)

    # attention block: heads shard with no exchange; o_proj partial sums meet in the all-reduce
    y, _ = qkv(x)
    qs, ks = hq // world * hs, max(1, hkv // world) * hs
    q, k, v = y.split([qs, ks, ks], dim=-1)
    full = F.linear(x, w_qkv)
    fq, fk, fv = full.split([hq * hs, hkv * hs, hkv * hs], dim=-1)
    assert torch.allclose(q, fq[:, rank * qs : (rank + 1) * qs], atol=1e-5)
    assert torch.allclose(k, fk[:, rank * ks : (rank + 1) * ks], atol=1e-5)
    assert torch.allclose(v, fv[:, rank * ks : (rank + 1) * ks], atol=1e-5)
    attn_like = fq[:, rank * qs : (rank + 1) * qs].contiguous()  # per-rank head slice stands in for the attention output
    out, _ = o(attn_like)
    assert torch.allclose(out, F.linear(fq, w_o), atol=1e-4)
    # MLP block
    h, _ = gu(x)
    gate, up = h.chunk(2, dim=-1)
    fg, fu = F.linear(x, w_gu).chunk(2, dim=-1)
    ip = inter // world
    assert torch.allclose(gate, fg[:, rank * ip : (rank + 1) * ip], atol=1e-5)
    act = F.silu(gate) * up
    out, _ = down(act)
    assert torch.allclose(out, F.linear(F.silu(fg) * fu, w_down, b_down), atol=1e-4)  # bias added once, not per rank
    # logits all-gather (vocab-sharded lm_head)
    vocab = 40
    w_head = torch.randn(vocab, hidden, generator=g)
    vs = vocab // world
    logits = comm.tensor_model_parallel_all_gather(F.linear(x, w_head[rank * vs : (rank + 1) * vs]))
    assert torch.allclose(logits, F.linear(x, w_head), atol=1e-5)
    results[rank] = True
    dist.destroy_process_group()


def test_tensor_parallel_sharding_and_all_reduce_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
    assert all(results.get(r) for r in range(world))


def test_kv_head_replication_when_tp_exceeds_kv_heads(pkg):
    """Llama-3-70B at TP=8 has 8 kv heads -> 1 per rank; with 4 kv heads at TP=8 each kv head is replicated on 2 ranks
    (llama.py:118-133).  shard_rows must pick kv head rank // replicas."""
    from ltp_sglang_amd.srt.layers.linear import QKVParallelLinear

    hs, hq, hkv, hidden = 4, 16, 4, 8
    full = torch.arange((hq + 2 * hkv) * hs * hidden, dtype=torch.float32).view(-1, hidden)
    for rank in range(8):
        lin = QKVParallelLinear(hidden, hs, hq, hkv, quant_config=_TorchQuantConfig(), params_dtype=torch.float32, tp_rank=rank, tp_size=8)
        assert lin.num_heads == 2 and lin.num_kv_heads == 1 and lin.num_kv_head_replicas == 2
        shard = lin.shard_rows(full)
        q, k, v = shard.split([2 * hs, hs, hs], dim=0)
        assert torch.equal(q, full[rank * 2 * hs : (rank + 1) * 2 * hs])
        kv = rank // 2
        assert torch.equal(k, full[hq * hs + kv * hs : hq * hs + (kv + 1) * hs])
        assert torch.equal(v, full[(hq + hkv) * hs + kv * hs : (hq + hkv) * hs + (kv + 1) * hs])
