"""pagerank_dist.py — row-sharded PageRank: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests) for the one
exchange step the path has.

The reference has no multi-GPU code; its single-GPU host loop is
src/pagerank.cu:50-153.  Here rank p owns the contiguous row block
[p*shard_len, (p+1)*shard_len) of the n x n matrix and keeps a full-length rank
vector.  Per iteration:

    engine.step(r_old, r_new)   fused HIP kernel over the local rows: SpMV, damping /
                                teleport update, partial residual^2 and dangling mass
    all_reduce(sums)            2 doubles (RCCL)               — only when world > 1
    engine.commit(sums)         device-side residual / convergence flag / next dangling mass
    all_gather(r_new)           shard_len floats per rank, in place — only when world > 1

Nothing else crosses ranks; the SpMV itself needs no collective (replicated x,
sharded A).  The compute engine is the C ABI of libspmv_amd.so (HipEngine); the
loop itself is backend-agnostic so the world_size-2 gloo tests drive it with a
test double on CPU tensors.
"""
from __future__ import annotations

import ctypes
from ctypes import byref, c_void_p

import numpy as np
import torch
import torch.distributed as dist

from . import PrStatus, csr_destroy, csr_wrap_device, lib


def shard_bounds(n: int, world: int, rank: int):
    """Equal contiguous row blocks: (shard_len, row_begin, row_end); the last shards may be short or empty."""
    shard_len = (n + world - 1) // world
    begin = min(rank * shard_len, n)
    end = min(begin + shard_len, n)
    return shard_len, begin, end


def initial_dangling_mass(num_dangling: int, n: int) -> float:
    """Left-to-right fp32 sum of `num_dangling` copies of 1/n (src/pagerank.cu:94-99 on the start vector)."""
    start = np.float32(1.0) / np.float32(n)
    if num_dangling > 1_000_000:
        return float(np.float32(num_dangling) * start)
    acc = np.float32(0.0)
    for _ in range(num_dangling):
        acc = np.float32(acc + start)
    return float(acc)


class HipEngine:
    """The shard engine behind include/spmv_c.h (spmv_c_pr_*), on torch CUDA(HIP) tensors.

    row_ptrs (rebased to 0), col_indices, values: this rank's rows as device tensors.
    Kernels are enqueued on torch's current stream, so they order with the RCCL calls.
    """

    def __init__(self, row_ptrs, col_indices, values, row_begin, n):
        assert row_ptrs.is_cuda and row_ptrs.dtype == torch.int32
        self.device = row_ptrs.device
        self.n = n
        self.row_begin = row_begin
        self.local_rows = row_ptrs.numel() - 1
        self._keep = (row_ptrs, col_indices, values)
        self._A = csr_wrap_device(self.local_rows, n, int(col_indices.numel()), row_ptrs.data_ptr(),
                                  col_indices.data_ptr() if col_indices.numel() else 0,
                                  values.data_ptr() if values.numel() else 0)
        if self._A is None:
            raise RuntimeError("csr_wrap_device failed")
        self._shard = None
        self._mask = None
        self._sums = torch.zeros(2, dtype=torch.float64, device=self.device)

    @staticmethod
    def _stream():
        return c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _check(status, what):
        if status != 0:
            raise RuntimeError(f"{what}: {lib().spmv_c_error_string(status).decode()}")

    def column_sums(self) -> torch.Tensor:
        sums = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        self._check(lib().spmv_c_pr_column_sums(self._A, c_void_p(sums.data_ptr()), self._stream()), "pr_column_sums")
        return sums

    def set_dangling_mask(self, mask: torch.Tensor) -> None:
        assert mask.dtype == torch.uint8 and mask.numel() >= self.n
        self._mask = mask
        if self._shard:
            lib().spmv_c_pr_shard_destroy(self._shard)
        self._shard = lib().spmv_c_pr_shard_create(self._A, self.row_begin, self.n, c_void_p(mask.data_ptr()))
        if not self._shard:
            raise RuntimeError("spmv_c_pr_shard_create failed")

    def reset(self, dangling_sum: float) -> None:
        self._check(lib().spmv_c_pr_reset(self._shard, dangling_sum, self._stream()), "pr_reset")

    def step(self, r_old: torch.Tensor, r_new: torch.Tensor, damping: float) -> torch.Tensor:
        self._check(lib().spmv_c_pr_step(self._shard, c_void_p(r_old.data_ptr()), c_void_p(r_new.data_ptr()),
                                         damping, self._stream()), "pr_step")
        self._check(lib().spmv_c_pr_reduce(self._shard, c_void_p(self._sums.data_ptr()), self._stream()), "pr_reduce")
        return self._sums

    def commit(self, sums: torch.Tensor, tolerance: float) -> None:
        self._check(lib().spmv_c_pr_commit(self._shard, c_void_p(sums.data_ptr()), tolerance, self._stream()),
                    "pr_commit")

    def status(self):
        out = PrStatus()
        self._check(lib().spmv_c_pr_status_get(self._shard, byref(out), self._stream()), "pr_status_get")
        return out.iterations, float(out.final_residual), bool(out.converged), bool(out.done)

    def close(self):
        if self._shard:
            lib().spmv_c_pr_shard_destroy(self._shard)
            self._shard = None
        if self._A is not None:
            csr_destroy(self._A)
            self._A = None


class ShardedPageRank:
    """The host loop.  `engine` is a HipEngine (product) or any object with the same
    methods (the CPU test double in tests/test_distributed_gloo.py)."""

    def __init__(self, engine, n, rank=0, world=1, group=None, device=None):
        self.engine, self.n, self.rank, self.world, self.group = engine, n, rank, world, group
        self.shard_len, self.row_begin, self.row_end = shard_bounds(n, world, rank)
        self.padded = self.shard_len * world
        self.device = device if device is not None else getattr(engine, "device", torch.device("cpu"))
        self.r = [torch.zeros(self.padded, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.num_dangling = None

    # -- one-time setup: dangling mask from the globally summed column sums ------------
    def prepare(self):
        sums = self.engine.column_sums()
        if self.world > 1:
            dist.all_reduce(sums, group=self.group)
        mask = torch.zeros(self.padded, dtype=torch.uint8, device=self.device)
        mask[: self.n] = (sums[: self.n] == 0).to(torch.uint8)
        self.num_dangling = int(mask.sum().item())
        self.engine.set_dangling_mask(mask)
        return self

    def reset(self):
        start = np.float32(1.0) / np.float32(self.n)
        for buf in self.r:
            buf.zero_()
            buf[: self.n] = float(start)
        self.engine.reset(initial_dangling_mass(self.num_dangling, self.n))

    def iterate(self, k, damping, tolerance):
        """Enqueue `step` iterations k (0-based); r[k & 1] -> r[(k + 1) & 1]."""
        r_old, r_new = self.r[k & 1], self.r[(k + 1) & 1]
        sums = self.engine.step(r_old, r_new, damping)
        if self.world > 1:
            dist.all_reduce(sums, group=self.group)
        self.engine.commit(sums, tolerance)
        if self.world > 1:
            mine = r_new[self.rank * self.shard_len:(self.rank + 1) * self.shard_len]
            dist.all_gather_into_tensor(r_new, mine, group=self.group)

    def run(self, damping=0.85, tolerance=1e-6, max_iterations=100, check_every=1):
        """Full PageRank; returns (ranks[n] float32 numpy, iterations, final_residual, converged).
        Steps enqueued after convergence are no-ops on every rank (device-side `done` flag),
        so `check_every` > 1 only trades host syncs for a few empty launches."""
        self.reset()
        for k in range(max_iterations):
            self.iterate(k, damping, tolerance)
            if (k + 1) % check_every == 0 and self.engine.status()[3]:
                break
        iterations, residual, converged, _ = self.engine.status()
        last = self.r[iterations & 1][: self.n].to("cpu").numpy().copy()
        total = np.float32(last.sum(dtype=np.float64))
        if total > 0:
            last = (last / total).astype(np.float32)
        return last, iterations, residual, converged
