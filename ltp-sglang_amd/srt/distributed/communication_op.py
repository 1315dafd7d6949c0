"""Tensor-parallel collectives at the reference's call sites.

``tensor_model_parallel_all_reduce`` is called by RowParallelLinear.forward exactly where the reference calls it
(python/sglang/srt/layers/linear.py:1302-1303 -> distributed/communication_op.py:11-13 ->
GroupCoordinator.all_reduce, parallel_state.py:459-538).  The MI355X build routes it to RCCL over xGMI through
torch.distributed (backend "nccl" is RCCL on ROCm); CPU tests use gloo.  One process per GPU; no other collective is
added to the path: attention shards by head with no exchange.
"""
from typing import Optional

import torch
import torch.distributed as dist

_TP_GROUP: Optional[dist.ProcessGroup] = None
_TP_SIZE = 1
_TP_RANK = 0
_HOST_STAGED = False  # gloo group + device tensors (single-GPU rehearsal of the TP path): collectives go through host copies
_EMULATED = False     # one process builds rank 0's shard of a tp-N model; every collective is a same-size device copy
_CUSTOM_AR = None     # optional one-shot P2P all-reduce over IPC-mapped peer buffers (custom_all_reduce.py)


def init_tensor_parallel(group: Optional[dist.ProcessGroup] = None) -> None:
    """Adopts an initialised torch.distributed group (default: WORLD) as the tensor-parallel group."""
    global _TP_GROUP, _TP_SIZE, _TP_RANK, _HOST_STAGED, _EMULATED
    _EMULATED = False
    if not dist.is_initialized():
        _TP_GROUP, _TP_SIZE, _TP_RANK, _HOST_STAGED = None, 1, 0, False
        return
    _TP_GROUP = group if group is not None else dist.group.WORLD
    _HOST_STAGED = dist.get_backend(_TP_GROUP) == "gloo"
    _TP_SIZE = dist.get_world_size(_TP_GROUP)
    _TP_RANK = dist.get_rank(_TP_GROUP)


def init_emulated_tensor_parallel(world_size: int) -> None:
    """Measurement mode (bench.py --emulate-tp N): this process is rank 0 of a tensor-parallel group of ``world_size`` whose
    other ranks do not exist.  Layers build rank 0's shard (Hq/N heads, N/N columns, K/N rows, vocab/N); each all-reduce /
    all-gather is replaced by a device copy of the same byte count, so the step time is ONE rank's compute + launch cost
    with zero communication latency.  Results are NOT the model's (partial sums are never summed): timing only."""
    global _TP_GROUP, _TP_SIZE, _TP_RANK, _HOST_STAGED, _EMULATED
    _TP_GROUP, _TP_SIZE, _TP_RANK, _HOST_STAGED, _EMULATED = None, int(world_size), 0, False, world_size > 1


def set_custom_all_reduce(obj) -> None:
    """Installs (or removes, with None) a custom all-reduce object with ``should_use(t)`` / ``all_reduce(t)``
    (the reference's GroupCoordinator.ca_comm, parallel_state.py:480-500)."""
    global _CUSTOM_AR
    _CUSTOM_AR = obj


def get_tensor_model_parallel_world_size() -> int:
    return _TP_SIZE


def get_tensor_model_parallel_rank() -> int:
    return _TP_RANK


def tensor_model_parallel_all_reduce(input_: torch.Tensor) -> torch.Tensor:
    """In-place sum over the TP group (bypassed when tp == 1, parallel_state.py:466-468)."""
    if _TP_SIZE == 1:
        return input_
    if _EMULATED:
        return input_.clone()
    if _CUSTOM_AR is not None and _CUSTOM_AR.should_use(input_):
        return _CUSTOM_AR.all_reduce(input_)
    if _HOST_STAGED and input_.is_cuda:
        host = input_.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=_TP_GROUP)
        input_.copy_(host)
        return input_
    dist.all_reduce(input_, op=dist.ReduceOp.SUM, group=_TP_GROUP)
    return input_


def fused_all_reduce_takes_slabs(rows: int, hidden: int, dtype) -> bool:
    """True when tensor_model_parallel_all_reduce_add_rmsnorm_quant can take the row-parallel GEMM's split-K slabs instead of its
    reduced output (the P2P communicator's fused kernels, or the emulated-TP stand-in): the GEMM's reduce launch is then skipped."""
    if _TP_SIZE <= 1:
        return False
    if _EMULATED:
        return True
    return _CUSTOM_AR is not None and hasattr(_CUSTOM_AR, "fused_norm_takes") and _CUSTOM_AR.fused_norm_takes(rows, hidden, dtype)


def tensor_model_parallel_all_reduce_add_rmsnorm_quant(partial, residual, weight, eps, want_norm=False, want_quant=True,
                                                       slabs=None, slab_sx=None, slab_sw=None, dtype=None):
    """The all-reduce of a row-parallel linear together with the fused add + RMSNorm (+ per-token fp8 quant) that consumes it
    on the decode path: one launch when the one-shot P2P communicator can take the message, the unfused pair otherwise (same
    bits either way).  ``residual`` is updated in place.  Returns (y or None, y_q or None, y_scale or None).
    ``partial=None, slabs=...``: the operand as the GEMM's split-K partial sums (only where fused_all_reduce_takes_slabs says so)."""
    from ...sgl_kernel import fused_add_rmsnorm_quant_fp8

    if partial is None:
        if _EMULATED:   # stand-in: the consumer kernel sums the slabs, as the fused all-reduce kernel does on real ranks
            return fused_add_rmsnorm_quant_fp8(None, residual, weight, eps, want_norm=want_norm, want_quant=want_quant, slabs=slabs,
                                               slab_sx=slab_sx, slab_sw=slab_sw, dtype=dtype)
        return _CUSTOM_AR.all_reduce_add_rmsnorm_quant(None, residual, weight, eps, want_norm, want_quant, slabs=slabs,
                                                       slab_sx=slab_sx, slab_sw=slab_sw, dtype=dtype)
    if _TP_SIZE > 1 and not _EMULATED and _CUSTOM_AR is not None and _CUSTOM_AR.should_use_fused_norm(partial):
        return _CUSTOM_AR.all_reduce_add_rmsnorm_quant(partial, residual, weight, eps, want_norm, want_quant)
    if not _EMULATED:   # (emulated TP: the fused kernel's cost is the norm kernel's -- the stand-in copy is dropped with the launch)
        partial = tensor_model_parallel_all_reduce(partial)
    return fused_add_rmsnorm_quant_fp8(partial, residual, weight, eps, want_norm=want_norm, want_quant=want_quant)


def tensor_model_parallel_all_gather(input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """Concatenate shards along ``dim`` (logits all-gather, logits_processor.py:471-500)."""
    if _TP_SIZE == 1:
        return input_
    if dim < 0:
        dim += input_.dim()
    if _EMULATED:
        shape = list(input_.shape)
        shape[dim] *= _TP_SIZE
        # the bytes one rank contributes; the peers' slices are ZEROS (a same-size device write, as the gather is): left unwritten
        # (round 4) they were whatever the allocator handed back, and bench.py's finite-logits check failed on that garbage -- 8B
        # --emulate-tp 8 -- while every kernel was right
        out = torch.zeros(shape, dtype=input_.dtype, device=input_.device)
        out.narrow(dim, 0, input_.shape[dim]).copy_(input_)
        return out
    if _CUSTOM_AR is not None and dim == input_.dim() - 1 and _CUSTOM_AR.should_use_gather(input_):
        return _CUSTOM_AR.all_gather_last_dim(input_)
    if _HOST_STAGED and input_.is_cuda:
        host = input_.contiguous().cpu()
        parts = [torch.empty_like(host) for _ in range(_TP_SIZE)]
        dist.all_gather(parts, host, group=_TP_GROUP)
        return torch.cat(parts, dim=dim).to(input_.device)
    parts = [torch.empty_like(input_) for _ in range(_TP_SIZE)]
    dist.all_gather(parts, input_.contiguous(), group=_TP_GROUP)
    return torch.cat(parts, dim=dim)
