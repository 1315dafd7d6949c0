"""CustomAllreduce: one-shot P2P all-reduce over IPC-mapped peer buffers (xGMI), the small-message path of the
tensor-parallel all-reduce (python/sglang/srt/distributed/device_communicators/custom_all_reduce.py:46-430 over
sgl-kernel/csrc/allreduce/custom_all_reduce_hip.cuh:261-549; chosen before RCCL by GroupCoordinator.all_reduce,
parallel_state.py:480-500).

One process per GPU.  At construction every rank allocates its uncached buffer through the C-ABI (``sgl_mi355_car_alloc``),
the 64-byte IPC handles travel over a CPU (gloo) group exactly as the reference exchanges them
(custom_all_reduce.py:_gather_ipc_meta), and each rank maps its peers' buffers.  ``all_reduce`` is then ONE kernel: publish,
flag every peer, wait for every peer, sum in rank order (see csrc/custom_all_reduce.hip).  The validation rig runs 2 / 4
processes on ONE GPU (tests/test_custom_all_reduce_gpu.py); on a node the same code path reads over xGMI.
"""
import ctypes
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from ..._cabi import check, current_stream, dtype_code, lib

_SUPPORTED_WORLD_SIZES = (2, 4, 6, 8)   # custom_all_reduce.py:52


class CustomAllreduce:
    def __init__(self, group: "dist.ProcessGroup", device: torch.device, max_size: int = 8 * 1024 * 1024):
        """``group``: a CPU-capable (gloo) group used only to exchange the IPC handles; ``max_size``: largest message in bytes
        (the reference's default is 8 MiB, custom_all_reduce.py:58)."""
        self.disabled = True
        self.disabled_reason = ""
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.device = torch.device(device)
        self.max_size = int(max_size) // 16 * 16
        self._own = ctypes.c_void_p()
        self._peers: List[Optional[int]] = []
        # The two-stage kernels (algo 2) have run with several processes on ONE GPU only; over xGMI they are unvalidated.  Until a
        # caller has checked them on the devices of this group (``validate_two_stage``; bench.py does at start-up) or opts in with
        # SGL_MI355_CAR_TWO_STAGE=1, the default dispatch (algo 0) stays on the one-shot kernel at every message size.
        self.two_stage_enabled = os.environ.get("SGL_MI355_CAR_TWO_STAGE", "0") == "1"
        if self.world_size == 1 or self.world_size not in _SUPPORTED_WORLD_SIZES:
            self.disabled_reason = f"world size {self.world_size} not in {_SUPPORTED_WORLD_SIZES}"
            return
        torch.cuda.set_device(self.device)
        # Every step that can fail on ONE rank (allocation, IPC export, mapping a peer) is followed by an exchange of the outcome,
        # so all ranks end up enabled or all disabled and nobody waits in a collective a failed peer never enters.
        handle = (ctypes.c_ubyte * 64)()
        err = self._try(lambda: check(lib.sgl_mi355_car_alloc(self.max_size, ctypes.byref(self._own), handle)))
        metas = [None] * self.world_size
        dist.all_gather_object(metas, (err, bytes(handle)), group=group)
        if any(e for e, _ in metas):
            self.disabled_reason = "; ".join(f"rank {r}: {e}" for r, (e, _) in enumerate(metas) if e)
            if not err:
                lib.sgl_mi355_car_free(self._own)
            return
        ptrs = (ctypes.c_void_p * self.world_size)()

        def open_peers():
            for r, (_, h) in enumerate(metas):
                if r == self.rank:
                    ptrs[r] = self._own.value
                    self._peers.append(None)
                else:
                    peer = ctypes.c_void_p()
                    buf = (ctypes.c_ubyte * 64).from_buffer_copy(h)
                    check(lib.sgl_mi355_car_open(buf, ctypes.byref(peer)))
                    ptrs[r] = peer.value
                    self._peers.append(peer.value)

        err = self._try(open_peers)
        outcomes = [None] * self.world_size
        dist.all_gather_object(outcomes, err, group=group)   # also the barrier: every rank has mapped every buffer before anyone launches
        if any(outcomes):
            self.disabled_reason = "; ".join(f"rank {r}: {e}" for r, e in enumerate(outcomes) if e)
            for p in self._peers:
                if p is not None:
                    lib.sgl_mi355_car_close(ctypes.c_void_p(p))
            self._peers = []
            lib.sgl_mi355_car_free(self._own)
            return
        self._ptrs = ptrs
        self.disabled = False

    @staticmethod
    def _try(fn) -> str:
        try:
            fn()
            return ""
        except Exception as e:   # reported to every rank; the communicator stays disabled everywhere
            return f"{type(e).__name__}: {e}"

    def should_use(self, inp: torch.Tensor) -> bool:
        """should_custom_ar (custom_all_reduce.py:345-360): small, 16-byte multiples, contiguous."""
        if self.disabled or not inp.is_cuda or not inp.is_contiguous():
            return False
        nbytes = inp.numel() * inp.element_size()
        return (inp.dtype in (torch.bfloat16, torch.float16, torch.float32) and nbytes % 16 == 0 and 0 < nbytes <= self.max_size
                and inp.data_ptr() % 16 == 0)

    should_custom_ar = should_use

    def _algo(self, algo: int) -> int:
        """The default (0 = the reference's size rule, custom_all_reduce_hip.cuh:543-549) may pick the two-stage kernels only
        once they are enabled on every rank alike; an explicit 1 / 2 is passed through."""
        return 1 if int(algo) == 0 and not self.two_stage_enabled else int(algo)

    def validate_two_stage(self, reference_all_reduce, rows: int = 128, hidden: int = 8192) -> bool:
        """Runs the plain and the fused two-stage kernels once on a message above the reference's 8-rank threshold (rows x hidden
        bf16 = 2 MiB by default) and compares them with ``reference_all_reduce`` (an f32 sum over RCCL / the host) and with the
        unfused row arithmetic.  Every rank must call it; the outcome is agreed over the CPU group, and only a pass on ALL ranks
        sets ``two_stage_enabled`` (so the dispatch stays the same everywhere).  Returns the agreed outcome."""
        from ...sgl_kernel import fused_add_rmsnorm_quant_fp8

        ok = False
        if not self.disabled and rows * hidden * 2 <= self.max_size:
            try:
                dev = self.device
                gen = torch.Generator(device=dev).manual_seed(1000 + self.rank)
                probe = torch.randn(rows, hidden, device=dev, generator=gen).to(torch.bfloat16)
                want = reference_all_reduce(probe.float())
                got = self.all_reduce(probe.clone(), algo=2)
                self.check_error()
                ok = bool(((got.float() - want).abs() <= 0.02 * want.abs() + 0.05).all())
                g7 = torch.Generator(device=dev).manual_seed(7)
                w = (1 + 0.1 * torch.randn(hidden, device=dev, generator=g7)).to(torch.bfloat16)
                res0 = torch.randn(rows, hidden, device=dev, generator=torch.Generator(device=dev).manual_seed(11)).to(torch.bfloat16)
                r1, r2 = res0.clone(), res0.clone()
                n1, _, s1 = fused_add_rmsnorm_quant_fp8(want.to(torch.bfloat16), r1, w, 1e-5, want_norm=True)
                n2, _, s2 = self.all_reduce_add_rmsnorm_quant(probe.clone(), r2, w, 1e-5, want_norm=True, algo=2)
                self.check_error()
                ok = (ok and bool(((n2.float() - n1.float()).abs() <= 0.03 * n1.float().abs() + 0.06).all())
                      and bool(((r2.float() - r1.float()).abs() <= 0.03 * r1.float().abs() + 0.06).all())
                      and bool(((s2 - s1).abs() <= 0.03 * s1.abs() + 1e-6).all()))
            except Exception:   # a spin bound reached, a launch error: the one-shot kernel stays the default
                ok = False
        outcomes = [None] * self.world_size
        dist.all_gather_object(outcomes, bool(ok), group=self.group)
        self.two_stage_enabled = all(outcomes)
        return self.two_stage_enabled

    def all_reduce(self, inp: torch.Tensor, algo: int = 0) -> torch.Tensor:
        """In-place sum over the group; returns ``inp``.  Capturable in a HIP graph (epochs live in device memory).
        ``algo``: 0 = the reference's dispatch rule (custom_all_reduce_hip.cuh:543-549) once the two-stage kernels are enabled
        (``two_stage_enabled``; one-shot at every size before), 1 = one-shot, 2 = two-stage (reduce-scatter + all-gather); the
        same value on every rank."""
        check(lib.sgl_mi355_car_all_reduce_algo(inp.data_ptr(), inp.numel(), dtype_code(inp.dtype), self._ptrs, self.rank,
                                                self.world_size, self.max_size, self._algo(algo), current_stream()))
        return inp

    custom_all_reduce = all_reduce

    def should_use_fused_norm(self, partial: torch.Tensor) -> bool:
        return (self.should_use(partial) and partial.dim() == 2 and partial.dtype in (torch.bfloat16, torch.float16)
                and partial.shape[1] % 8 == 0 and partial.shape[1] <= 8192)

    def fused_norm_takes(self, rows: int, hidden: int, dtype) -> bool:
        """should_use_fused_norm for an operand that does not exist as a tensor yet (the slab form)."""
        return (not self.disabled and dtype in (torch.bfloat16, torch.float16) and hidden % 8 == 0 and hidden <= 8192
                and 0 < rows * hidden * 2 <= self.max_size)

    def all_reduce_add_rmsnorm_quant(self, partial, residual, weight, eps, want_norm=False, want_quant=True, algo: int = 0,
                                     slabs=None, slab_sx=None, slab_sw=None, dtype=None):
        """all_reduce(partial) -> residual += . -> rmsnorm * weight -> per-token fp8 quant, ONE launch, bit-identical to the
        unfused pair.  Returns (y or None, y_q or None, y_scale [M, 1] or None); ``residual`` is updated in place.  ``algo`` as
        for ``all_reduce`` (two-stage: the owner of a row finishes it once, the other ranks collect it).
        ``partial=None, slabs=[S, M, H] f32`` (+ the GEMM's ``slab_sx`` [M] / ``slab_sw`` [H], ``dtype``): the operand is formed from
        the split-K partial sums of this rank's GEMM inside the launch (sgl_mi355_car_all_reduce_add_rmsnorm_quant_slabs)."""
        if partial is None:
            assert slabs is not None and slabs.dtype == torch.float32 and slabs.dim() == 3 and slabs.is_contiguous() and dtype is not None
            _, m, h = slabs.shape
            out_dtype, dev = dtype, slabs.device
        else:
            m, h = partial.shape
            out_dtype, dev = partial.dtype, partial.device
        if getattr(self, "_fused_hidden", h) != h:
            # The protocol has no closing barrier: block b may re-enter a data half because the peers' blocks b have left it,
            # which needs the row -> byte-range map of the fused family to stay the same from call to call.  It depends on
            # `hidden`; a model keeps one hidden size, a test that changes it pays a host-level barrier here (never inside a
            # captured step).
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)
        self._fused_hidden = h
        out_norm = torch.empty((m, h), dtype=out_dtype, device=dev) if want_norm else None
        out_q = torch.empty((m, h), dtype=torch.float8_e4m3fn, device=dev) if want_quant else None
        out_s = torch.empty((m, 1), dtype=torch.float32, device=dev) if want_quant else None
        if partial is None:
            check(lib.sgl_mi355_car_all_reduce_add_rmsnorm_quant_slabs(
                slabs.data_ptr(), int(slabs.shape[0]), None if slab_sx is None else slab_sx.data_ptr(),
                None if slab_sw is None else slab_sw.data_ptr(), None if residual is None else residual.data_ptr(), weight.data_ptr(),
                float(eps), None if out_norm is None else out_norm.data_ptr(), None if out_q is None else out_q.data_ptr(),
                None if out_s is None else out_s.data_ptr(), m, h, dtype_code(out_dtype), self._ptrs, self.rank, self.world_size,
                self.max_size, self._algo(algo), current_stream()))
            return out_norm, out_q, out_s
        check(lib.sgl_mi355_car_all_reduce_add_rmsnorm_quant_algo(
            partial.data_ptr(), None if residual is None else residual.data_ptr(), weight.data_ptr(), float(eps),
            None if out_norm is None else out_norm.data_ptr(), None if out_q is None else out_q.data_ptr(),
            None if out_s is None else out_s.data_ptr(), m, h, dtype_code(partial.dtype), self._ptrs, self.rank, self.world_size,
            self.max_size, self._algo(algo), current_stream()))
        return out_norm, out_q, out_s

    def should_use_gather(self, inp: torch.Tensor) -> bool:
        if self.disabled or not inp.is_cuda or not inp.is_contiguous() or inp.dim() < 1:
            return False
        row_bytes = inp.shape[-1] * inp.element_size()
        return row_bytes % 16 == 0 and 0 < inp.numel() * inp.element_size() <= self.max_size and inp.data_ptr() % 16 == 0

    def all_gather_last_dim(self, inp: torch.Tensor) -> torch.Tensor:
        """Concatenation of every rank's ``inp`` along the last dimension (one kernel, same buffers and flags)."""
        rows = inp.numel() // inp.shape[-1]
        out = torch.empty(inp.shape[:-1] + (inp.shape[-1] * self.world_size,), dtype=inp.dtype, device=inp.device)
        check(lib.sgl_mi355_car_all_gather(inp.data_ptr(), out.data_ptr(), rows, inp.shape[-1] * inp.element_size(), self._ptrs,
                                           self.rank, self.world_size, self.max_size, current_stream()))
        return out

    def check_error(self) -> None:
        """Raises if a peer failed to arrive within the kernel's spin bound since the last check (synchronises)."""
        if self.disabled:
            return
        torch.cuda.synchronize(self.device)
        rc = lib.sgl_mi355_car_error(self._own)
        if rc != 0:
            raise RuntimeError("custom all-reduce: a peer rank did not arrive (spin bound reached); results are invalid")

    def close(self) -> None:
        if self.disabled:
            return
        self.disabled = True
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)   # nobody unmaps while a peer may still read
        for p in self._peers:
            if p is not None:
                lib.sgl_mi355_car_close(ctypes.c_void_p(p))
        lib.sgl_mi355_car_free(self._own)

    def __del__(self):
        # No collective here: at interpreter shutdown the peers may already be gone and a barrier would hang the process.
        # The mappings and the allocation die with the process; call close() explicitly for an orderly teardown.
        self.disabled = True
