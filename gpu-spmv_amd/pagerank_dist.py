"""pagerank_dist.py — row-sharded PageRank: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests) for the one
exchange step the path has.

The reference has no multi-GPU code; its single-GPU host loop is
src/pagerank.cu:50-153.  Here rank p owns the contiguous row block
[p*shard_len, (p+1)*shard_len) of the n x n matrix and keeps a full-length rank
vector.  Per iteration (world > 1):

    engine.step(r_old, r_new)   HIP kernels over the local rows: SpMV, damping / teleport
                                update, partial residual^2 and dangling mass; the two partial
                                sums (doubles) are written into the 16-byte TAIL of this
                                rank's slice of r_new
    all_gather(r_new)           ONE collective per iteration (RCCL): shard_len + 4 floats per
                                rank, in place — delivers every slice and every rank's partials
    engine.commit_gathered()    every rank folds the P partial pairs in rank order (identical
                                result everywhere): residual, convergence flag, next dangling mass

so the vector lives in a padded layout: node g sits at (g // shard_len) * stride + g % shard_len
with stride = shard_len + 4; the shard's column indices are remapped to that layout once
(`Layout.remap_columns`), which costs nothing per iteration.  With world == 1 there is no
padding, no collective, and commit() consumes the local sums directly.

Overlapped form (Layout(chunks=C), C > 1; SURVEY.md H3 "multiply what has arrived while the rest is in
flight"): the vector is laid out as C blocks, block c = piece c of every rank, and the exchange is C in-place
all-gathers issued back to back on a side stream; as soon as block c is complete the tiled engine's phase 1
runs for the strips inside it (engine.expand), so only the last block's multiply and phase 2 are exposed.
Same arithmetic, same bits as the one-collective form.

Optional "push" mode (enable_push): the step kernels store every new value straight into the
peers' vectors as well (IPC-mapped buffers; xGMI is point-to-point, so all 7 links carry
traffic at once, under the step's own epilogue).  What remains per iteration is a 16-byte
all-reduce of the partial sums, which doubles as the cross-GPU barrier: it cannot complete
on any rank before every rank's step kernels have finished, i.e. before every push has landed
and before every peer has finished reading the buffer the next step will overwrite.

Nothing else crosses ranks; the SpMV itself needs no collective (replicated x, sharded A).
The compute engine is the C ABI of libspmv_amd.so (HipEngine); the loop itself is
backend-agnostic so the world_size-2 gloo tests drive it with a test double on CPU tensors.
"""
from __future__ import annotations

from ctypes import byref, c_void_p

import numpy as np
import torch
import torch.distributed as dist

from . import PrStatus, csr_destroy, csr_wrap_device, lib

TAIL = 4   # floats appended to every slice when world > 1 (two doubles)
EXIT_PEER_FAILED = 70      # exit code of a rank that had to leave because ANOTHER rank failed while it was blocked


class PeerFailure(RuntimeError):
    """Another rank of the job reported a failure (or the rendezvous store went away): this rank stops too."""


class _WatchPoller:
    """ONE daemon thread per process that polls the rendezvous store for every live FailureWatch (a bench run keeps
    several ShardedPageRank objects alive: each used to add its own polling thread beside the timed loop)."""
    _lock = None
    _thread = None
    _watches = []
    _wake = None

    @classmethod
    def add(cls, watch):
        import threading
        if cls._lock is None:
            cls._lock = threading.Lock()
            cls._wake = threading.Event()
        with cls._lock:
            cls._watches.append(watch)
            if cls._thread is None or not cls._thread.is_alive():
                cls._thread = threading.Thread(target=cls._loop, name="spmv-failure-watch", daemon=True)
                cls._thread.start()

    @classmethod
    def remove(cls, watch):
        if cls._lock is None:
            return
        with cls._lock:
            if watch in cls._watches:
                cls._watches.remove(watch)

    @classmethod
    def _loop(cls):
        import time
        while True:
            with cls._lock:
                live = list(cls._watches)
                if not live:
                    cls._thread = None
                    return
            for watch in live:
                watch._poll_once()
            time.sleep(min(w.poll for w in live))


class FailureWatch:
    """What one failing rank costs the others (failure detection).

    A collective has no way out when a peer never joins it: RCCL waits on the device, gloo in the host call.  So
    every rank polls a few keys of the job's rendezvous store a few times per second (one thread per process,
    _WatchPoller).  A rank whose loop raises writes its story under ITS OWN key (`report`) and re-raises — the process
    ends non-zero by the ordinary route, or the caller handles the exception and carries on: a rank never reads its own
    key, so its own report cannot take it down later.  A rank that sees ANOTHER rank's key gives its main thread `grace`
    seconds to notice (`check()` is called at the top of every iteration and raises PeerFailure, an ordinary exception
    the caller may handle); a main thread that does not — because it sits inside the collective the failed rank will
    never join — is taken down with the process (os._exit(EXIT_PEER_FAILED)), which is what releases the GPU side too.
    Nobody hangs, every exit is non-zero.

    The keys belong to ONE watch: they carry the watch's generation (the count of watches this process has created for
    this job — the ranks create their loops in lockstep, so the number agrees across ranks), and a failure reported
    against one loop (an exchange trial whose exception the caller swallowed) does not fail the loops created after it.
    """
    KEY = "spmv_amd/rank_failed"
    _generation = 0

    def __init__(self, rank, world, store=None, poll=0.25, grace=5.0):
        self.rank, self.world, self.poll, self.grace = rank, world, poll, grace
        self.store = store
        self.peer_failed = None          # the failed rank's story, once seen
        self.acknowledged = False        # the main thread has seen it (it is on its way out by itself)
        self._seen_at = None
        self._stopped = False
        FailureWatch._generation += 1
        self.generation = FailureWatch._generation
        if self.store is None and world > 1 and dist.is_available() and dist.is_initialized():
            try:
                from torch.distributed.distributed_c10d import _get_default_store
                self.store = dist.PrefixStore("spmv_amd_watch", _get_default_store())
            except Exception:           # noqa: BLE001 - no store, no watch
                self.store = None
        self._peer_keys = ["%s/%d/%d" % (self.KEY, self.generation, r) for r in range(world) if r != rank]
        if self.store is not None and world > 1:
            _WatchPoller.add(self)

    def _key(self, rank):
        return "%s/%d/%d" % (self.KEY, self.generation, rank)

    def _poll_once(self):
        """Called by the poller thread.  Other ranks' keys only: this rank's own report is not news to it."""
        import os
        import sys
        import time
        if self._stopped:
            return
        if self.peer_failed is None:
            try:
                for key in self._peer_keys:
                    if self.store.check([key]):
                        self.peer_failed = self.store.get(key).decode(errors="replace")
                        self._seen_at = time.time()
                        break
            except Exception:           # noqa: BLE001
                # The store lives in rank 0's process.  Losing it is what an orderly end of rank 0 looks like as well,
                # so it is not read as a failure: a rank that dies without a word is the launcher's business
                # (torch.distributed.run and mp.spawn both end the other workers when one exits non-zero).
                self.stop()
                return
        if self.peer_failed is None or self.acknowledged:
            return
        if time.time() - self._seen_at >= self.grace:
            print("[spmv] rank %d: another rank failed (%s) and this rank is blocked in the exchange: leaving with exit "
                  "code %d" % (self.rank, self.peer_failed, EXIT_PEER_FAILED), file=sys.stderr, flush=True)
            os._exit(EXIT_PEER_FAILED)

    def check(self):
        """Top of every iteration: cheap (one attribute), raises once a peer has failed."""
        if self.peer_failed is not None:
            self.acknowledged = True
            raise PeerFailure("rank %d stops: %s" % (self.rank, self.peer_failed))

    def report(self, exc):
        """This rank is failing with `exc`: tell the others before going down."""
        if self.store is None or isinstance(exc, PeerFailure):
            return
        self.acknowledged = True         # whatever the peers report from here on, this rank is already on its way out
        try:
            self.store.set(self._key(self.rank), "rank %d: %r" % (self.rank, exc))
        except Exception:               # noqa: BLE001 - best effort on the way out
            pass

    def stop(self):
        """The loop this watch belongs to is over (ShardedPageRank.close): no more polling for it, and its own key goes."""
        self._stopped = True
        _WatchPoller.remove(self)
        if self.store is not None:
            try:
                self.store.delete_key(self._key(self.rank))
            except Exception:           # noqa: BLE001 - a store without delete, or already gone
                pass


class Layout:
    """Where node g lives in the (padded) rank vector, and who owns which rows.

    Rank p owns the contiguous row block [bounds[p], bounds[p + 1]).  Default: equal ROWS
    (ceil(n / world), what a uniform matrix wants).  `bounds` (world + 1 ascending row indices, e.g. from
    `equal_nnz_bounds`) cuts the rows anywhere — SURVEY.md §8(e): "boundaries chosen by binary search
    on row_ptrs for equal nnz" — so that a power-law graph does not leave one rank with most of the
    entries.

    The vector is a sequence of `chunks` BLOCKS; block c holds piece c (`piece` floats) of every rank,
    back to back, so one in-place all-gather per block delivers it:

        position of rank p's local row i = (i // piece) * block + p * piece + i % piece,   block = world * piece

    chunks == 1 (default): one block, piece = stride = the longest row block (rounded up to even) + TAIL — every
    rank's slice is contiguous and ONE all-gather per iteration moves everything.  chunks > 1: the overlapped
    exchange — the all-gather of block c + 1 runs while the products of the columns in block c are computed;
    pieces are rounded up to `align` columns so that block boundaries are strip boundaries of the tiled engine.
    The last TAIL floats of a rank's last piece carry its two partial sums (as doubles) in either form."""

    def __init__(self, n: int, world: int = 1, rank: int = 0, bounds=None, chunks: int = 1, align: int = None,
                 exchange: bool = None):
        self.n, self.world, self.rank = n, world, rank
        # exchange=True with world == 1 keeps tails, blocks and collectives in the loop although there is nobody to
        # exchange with: the whole multi-rank code path on one device (how the RCCL calls are tested on a 1-GPU box)
        self.exchange = world > 1 if exchange is None else bool(exchange)
        if bounds is None:
            shard_len = (n + world - 1) // world
            if world > 1 and shard_len % 2:
                shard_len += 1                      # keeps every tail 8-byte aligned
            self.bounds = np.minimum(np.arange(world + 1, dtype=np.int64) * shard_len, n)
        else:
            self.bounds = np.asarray(bounds, dtype=np.int64)
            assert self.bounds.shape == (world + 1,) and self.bounds[0] == 0 and self.bounds[-1] == n
            assert (np.diff(self.bounds) >= 0).all()
            shard_len = int(np.diff(self.bounds).max()) if world > 0 else n
            if world > 1 and shard_len % 2:
                shard_len += 1
        if self.exchange and shard_len % 2:
            shard_len += 1
        tail = TAIL if self.exchange else 0
        self.chunks = max(1, int(chunks)) if self.exchange else 1
        if self.chunks == 1:
            self.piece = shard_len + tail
        else:
            if align is None:                   # the tiled engine's widest strip, where it can be in play at all
                align = 32768 if shard_len >= (1 << 18) else 4
            assert align >= 2 and align % 2 == 0
            per_chunk = (shard_len + tail + self.chunks - 1) // self.chunks
            self.piece = (per_chunk + align - 1) // align * align
            while self.piece < TAIL:            # (tiny shards, many blocks) the tail must fit inside the last piece
                self.piece += align
        self.block = self.piece * world
        self.padded = self.block * self.chunks
        self.shard_len = self.piece * self.chunks - tail     # rows a rank's pieces can hold
        self.stride = self.piece                               # distance between two ranks' pieces inside a block
        self.row_begin = int(self.bounds[rank])
        self.row_end = int(self.bounds[rank + 1])
        self.local_rows = self.row_end - self.row_begin
        self.row_offset = rank * self.piece     # position of this rank's first node in the vector

    @staticmethod
    def equal_nnz_bounds(row_ptrs, world: int) -> np.ndarray:
        """Row boundaries that give every rank about nnz / world entries: binary search on the
        (global) row_ptrs for the targets p * nnz / world."""
        row_ptrs = np.asarray(row_ptrs, dtype=np.int64)
        n, nnz = len(row_ptrs) - 1, int(row_ptrs[-1])
        targets = (np.arange(1, world, dtype=np.int64) * nnz + world // 2) // world
        inner = np.clip(np.searchsorted(row_ptrs, targets, side="left"), 0, n)
        # the search lands on the first boundary at or past the target; the one before may be nearer
        before = np.maximum(inner - 1, 0)
        nearer = np.abs(row_ptrs[before] - targets) < np.abs(row_ptrs[inner] - targets)
        inner = np.maximum.accumulate(np.where(nearer, before, inner))
        return np.concatenate([[0], inner, [n]]).astype(np.int64)

    def owner_of(self, nodes):
        """Owning rank of every node index (numpy or torch)."""
        if isinstance(nodes, torch.Tensor):
            edges = torch.as_tensor(self.bounds[1:-1], dtype=nodes.dtype, device=nodes.device)
            return torch.bucketize(nodes, edges, right=True)
        return np.searchsorted(self.bounds[1:-1], nodes, side="right")

    def _place(self, owner, local):
        """(owning rank, row inside its block of rows) -> position; numpy or torch, elementwise."""
        if self.chunks == 1:
            return owner * self.piece + local
        return (local // self.piece) * self.block + owner * self.piece + local % self.piece

    def remap_columns(self, cols):
        """Column (= node) indices -> positions in the padded vector (numpy or torch int32)."""
        if not self.exchange:
            return cols
        owner = self.owner_of(cols)
        if isinstance(cols, torch.Tensor):
            begin = torch.as_tensor(self.bounds[:-1], dtype=cols.dtype, device=cols.device)[owner]
            return self._place(owner.to(cols.dtype), cols - begin).to(cols.dtype)
        return self._place(owner, cols - self.bounds[:-1][owner]).astype(cols.dtype)

    def positions(self) -> np.ndarray:
        """Padded position of every node 0..n-1."""
        g = np.arange(self.n, dtype=np.int64)
        if not self.exchange:
            return g
        owner = np.searchsorted(self.bounds[1:-1], g, side="right")
        return self._place(owner, g - self.bounds[:-1][owner])

    def local_positions(self) -> np.ndarray:
        """Padded positions of this rank's rows, in row order."""
        return self._place(self.rank, np.arange(self.local_rows, dtype=np.int64))

    def tail_slice(self, rank=None):
        """Where rank's two partial sums travel: the last TAIL floats of its last piece."""
        rank = self.rank if rank is None else rank
        start = (self.chunks - 1) * self.block + (rank + 1) * self.piece - TAIL
        return slice(start, start + TAIL)

    def block_slice(self, c):
        """Block c of the vector = the output of all-gather c."""
        return slice(c * self.block, (c + 1) * self.block)

    def piece_slice(self, c, rank=None):
        """This rank's piece of block c = the input of all-gather c."""
        rank = self.rank if rank is None else rank
        start = c * self.block + rank * self.piece
        return slice(start, start + self.piece)

    def row_map(self):
        """(base, piece, block) for spmv_c_pr_shard_create_chunked."""
        if self.chunks == 1:
            return self.row_offset, 0x7FFFFFFF, 0
        return self.rank * self.piece, self.piece, self.block


def shard_bounds(n: int, world: int, rank: int):
    """(shard_len, row_begin, row_end) of Layout(n, world, rank)."""
    lay = Layout(n, world, rank)
    return lay.shard_len, lay.row_begin, lay.row_end


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

    row_ptrs (rebased to 0), col_indices (already remapped by Layout.remap_columns), values:
    this rank's rows as device tensors.  Kernels are enqueued on torch's current stream, so
    they order with the RCCL calls.
    """

    def __init__(self, row_ptrs, col_indices, values, layout: Layout):
        assert row_ptrs.is_cuda and row_ptrs.dtype == torch.int32
        assert row_ptrs.numel() - 1 == layout.local_rows
        self.device = row_ptrs.device
        self.layout = layout
        self._keep = (row_ptrs, col_indices, values)
        self._A = csr_wrap_device(layout.local_rows, layout.padded, int(col_indices.numel()), row_ptrs.data_ptr(),
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
        """Column sums of this shard's stored values, in padded coordinates."""
        sums = torch.zeros(self.layout.padded, dtype=torch.float32, device=self.device)
        self._check(lib().spmv_c_pr_column_sums(self._A, c_void_p(sums.data_ptr()), self._stream()), "pr_column_sums")
        return sums

    def set_dangling_mask(self, mask: torch.Tensor) -> None:
        assert mask.dtype == torch.uint8 and mask.numel() >= self.layout.padded
        self._mask = mask
        if self._shard:
            lib().spmv_c_pr_shard_destroy(self._shard)
        # the engine divides by the TRUE node count; the matrix header carries the padded width.
        # Creating the shard may build the matrix's tiled plan on the library's stream: whatever
        # filled the device arrays on torch's stream must be done first.
        torch.cuda.current_stream(self.device).synchronize()
        base, piece, block = self.layout.row_map()
        self._shard = lib().spmv_c_pr_shard_create_chunked(self._A, base, piece, block, self.layout.n,
                                                           c_void_p(mask.data_ptr()))
        if not self._shard:
            raise RuntimeError("spmv_c_pr_shard_create_chunked failed")

    def reset(self, dangling_sum: float) -> None:
        self._check(lib().spmv_c_pr_reset(self._shard, dangling_sum, self._stream()), "pr_reset")

    def step(self, r_old: torch.Tensor, r_new: torch.Tensor, damping: float, sums_out: torch.Tensor = None,
             push_to=None):
        """Enqueue one step; the two partial sums land in `sums_out` (2 doubles: a view into the
        tail of this rank's slice when world > 1) or in the engine's own buffer.  `push_to`: ctypes
        array of the peers' r_new device pointers — the step also stores its new slice there."""
        target = self._sums if sums_out is None else sums_out
        if push_to is None:
            self._check(lib().spmv_c_pr_step(self._shard, c_void_p(r_old.data_ptr()), c_void_p(r_new.data_ptr()),
                                             damping, self._stream()), "pr_step")
        else:
            self._check(lib().spmv_c_pr_step_push(self._shard, c_void_p(r_old.data_ptr()), c_void_p(r_new.data_ptr()),
                                                  damping, push_to, len(push_to), self._stream()), "pr_step_push")
        self._check(lib().spmv_c_pr_reduce(self._shard, c_void_p(target.data_ptr()), self._stream()), "pr_reduce")
        return target

    def expand(self, r_old: torch.Tensor, cols_ready: int) -> None:
        """Head start on the next step(r_old, ...): columns [0, cols_ready) of r_old are final."""
        self._check(lib().spmv_c_pr_expand(self._shard, c_void_p(r_old.data_ptr()), cols_ready, self._stream()),
                    "pr_expand")

    def step_and_commit(self, r_old: torch.Tensor, r_new: torch.Tensor, damping: float, tolerance: float) -> None:
        """Single-rank iteration: the step, then reduce + commit in one launch."""
        self._check(lib().spmv_c_pr_step(self._shard, c_void_p(r_old.data_ptr()), c_void_p(r_new.data_ptr()),
                                         damping, self._stream()), "pr_step")
        self._check(lib().spmv_c_pr_reduce_commit(self._shard, tolerance, self._stream()), "pr_reduce_commit")

    def commit(self, sums: torch.Tensor, tolerance: float) -> None:
        self._check(lib().spmv_c_pr_commit(self._shard, c_void_p(sums.data_ptr()), tolerance, self._stream()),
                    "pr_commit")

    def commit_gathered(self, gathered: torch.Tensor, tolerance: float) -> None:
        """`gathered`: the whole vector; the ranks' partial sums sit in the tails of its last block."""
        lay = self.layout
        last = gathered.data_ptr() + 4 * lay.block_slice(lay.chunks - 1).start      # (no tensor view: this runs every step)
        self._check(lib().spmv_c_pr_commit_gathered(self._shard, c_void_p(last), lay.world,
                                                    lay.piece, lay.piece - TAIL, tolerance, self._stream()),
                    "pr_commit_gathered")

    def status(self):
        out = PrStatus()
        self._check(lib().spmv_c_pr_status_get(self._shard, byref(out), self._stream()), "pr_status_get")
        return out.iterations, float(out.final_residual), bool(out.converged), bool(out.done)

    def close(self):
        """Destroys the shard and the matrix handle — and with it the side table (tiled plan, merge tables)
        keyed by the device arrays: torch's caching allocator may hand the same addresses to the next matrix."""
        if self._shard:
            lib().spmv_c_pr_shard_destroy(self._shard)
            self._shard = None
        if self._A is not None:
            csr_destroy(self._A)
            self._A = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:           # interpreter teardown: the library may be gone
            pass


class ShardedPageRank:
    """The host loop.  `engine` is a HipEngine (product) or any object with the same
    methods (the CPU test double in tests/test_distributed_gloo.py)."""

    def __init__(self, engine, layout: Layout, group=None, device=None, watch=None):
        self.engine, self.layout, self.group = engine, layout, group
        self.n, self.rank, self.world = layout.n, layout.rank, layout.world
        # one failing rank must not leave the others inside a collective for ever (FailureWatch above)
        self.watch = watch if watch is not None else FailureWatch(self.rank, self.world)
        self.device = device if device is not None else getattr(engine, "device", torch.device("cpu"))
        # The two rank vectors.  On a GPU they are plain hipMalloc allocations made through the C ABI
        # (base pointers, so they can be exported with hipIpcGetMemHandle for the push exchange) and
        # viewed by torch through __cuda_array_interface__; on CPU (gloo tests) ordinary tensors.
        self._owned = []
        if self.device.type == "cuda":
            self.r = [self._device_vector(layout.padded) for _ in range(2)]
        else:
            self.r = [torch.zeros(layout.padded, dtype=torch.float32, device=self.device) for _ in range(2)]
        self._pos = torch.from_numpy(layout.positions()).to(self.device)
        self.num_dangling = None
        self.mode = "gather"
        self._peer_ptrs = None
        self._peer_keepalive = []
        self._view_cache = {}
        # overlapped exchange: the collectives are issued from a side stream so that the compute stream is
        # free to multiply block c while block c + 1 is still on the links
        self._comm = torch.cuda.Stream(self.device) if self.device.type == "cuda" and layout.chunks > 1 else None
        assert layout.exchange or layout.world == 1

    def _device_vector(self, count):
        ptr = c_void_p(None)
        status = lib().spmv_c_device_malloc(byref(ptr), max(count, 1) * 4)
        if status != 0:
            raise RuntimeError("device allocation of the rank vector failed")
        self._owned.append(ptr)

        class _Raw:                      # zero-copy view for torch
            __cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr.value, False),
                                        "version": 2, "strides": None}
        t = torch.as_tensor(_Raw(), device=self.device)
        t.zero_()
        return t

    def close_peers(self):
        """Unmap the peers' vectors (every rank must do this before anybody frees its own)."""
        for p in self._peer_keepalive:
            lib().spmv_c_ipc_close(c_void_p(p))
        self._peer_keepalive = []
        self._peer_ptrs = None
        if self.mode == "push":
            self.mode = "gather"

    def close(self):
        """Unmap the peers' vectors, then (after a barrier when ranks share mappings) free the rank
        vectors.  The engine is closed by its owner."""
        had_peers = bool(self._peer_keepalive)
        self.watch.stop()
        self.close_peers()
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)      # nobody frees a vector a peer still has mapped
        del had_peers
        self.r = []
        self._view_cache = {}
        for ptr in self._owned:
            lib().spmv_c_device_free(ptr)
        self._owned = []

    # The views an iteration needs are made once per rank vector (a slice + view costs a few microseconds each,
    # and at eight ranks the whole step is ~200 us).
    def _views(self, buf):
        key = buf.data_ptr()
        cached = self._view_cache.get(key)
        if cached is None:
            lay = self.layout
            cached = {"tail": buf[lay.tail_slice()].view(torch.float64) if lay.exchange else None,
                      "pieces": [buf[lay.piece_slice(c)] for c in range(lay.chunks)],
                      "blocks": [buf[lay.block_slice(c)] for c in range(lay.chunks)]}
            self._view_cache[key] = cached
        return cached

    def _my_slice(self, buf, c=0):
        return self._views(buf)["pieces"][c]

    def _my_tail(self, buf):
        return self._views(buf)["tail"]

    # -- one-time setup: dangling mask from the globally summed column sums ------------
    def prepare(self):
        sums = self.engine.column_sums()
        if self.world > 1:
            dist.all_reduce(sums, group=self.group)
        mask = torch.zeros(self.layout.padded, dtype=torch.uint8, device=self.device)
        mask[self._pos] = (sums[self._pos] == 0).to(torch.uint8)
        self.num_dangling = int(mask.sum().item())
        self.engine.set_dangling_mask(mask)
        return self

    # -- optional: map the peers' vectors for the push-style exchange -------------------
    def enable_push(self, peer_devices=None) -> bool:
        """Exchange IPC handles of both rank vectors and map every peer's pair.  Returns True when
        all ranks succeeded (then mode == "push"); on any failure every rank stays in gather mode."""
        if self.world == 1:
            return False
        import ctypes
        ok = 1
        ptrs = None
        mine = []
        try:                                                            # local: export my two vectors
            for ptr in self._owned[:2]:
                raw = ctypes.create_string_buffer(64)
                if lib().spmv_c_ipc_get_handle(ptr, raw) != 0:
                    raise RuntimeError("hipIpcGetMemHandle refused")
                mine.append(raw.raw)
            if len(mine) != 2:
                raise RuntimeError("rank vectors are not device allocations")
        except Exception as exc:                                        # noqa: BLE001
            ok = 0
            self._push_error = repr(exc)
        agreed = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN, group=self.group)
        if int(agreed.item()) == 0:                                     # every rank skips the exchange together
            self.mode = "gather"
            return False
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.group)
        try:                                                            # local: map every peer's pair
            ptrs = [(c_void_p * (self.world - 1))() for _ in range(2)]
            for which in range(2):
                slot = 0
                for p in range(self.world):
                    if p == self.rank:
                        continue
                    if peer_devices is not None and which == 0 and peer_devices[p] != peer_devices[self.rank]:
                        # stores into a peer's memory need peer access from THIS device; without it the
                        # mapping below could still open and the first store would fault, so be strict
                        if lib().spmv_c_enable_peer_access(int(peer_devices[p])) != 0:
                            raise RuntimeError("no peer access from device %d to device %d"
                                               % (peer_devices[self.rank], peer_devices[p]))
                    opened = c_void_p(None)
                    if lib().spmv_c_ipc_open_handle(everyone[p][which], byref(opened)) != 0 or not opened.value:
                        raise RuntimeError("hipIpcOpenMemHandle refused for rank %d" % p)
                    self._peer_keepalive.append(opened.value)
                    # read one word through the mapping with a runtime copy: a mapping this device cannot
                    # reach comes back as an error here instead of as a fault inside a step kernel
                    probe = ctypes.c_float(0.0)
                    if lib().spmv_c_memcpy_d2h(ctypes.byref(probe), opened, 4) != 0:
                        raise RuntimeError("the mapping of rank %d's vector is not readable from here" % p)
                    ptrs[which][slot] = opened.value
                    slot += 1
        except Exception as exc:                                        # noqa: BLE001 - any failure => gather mode
            ok = 0
            self._push_error = repr(exc)
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 1:
            self._peer_ptrs = ptrs
            self.mode = "push"
            return True
        for p in self._peer_keepalive:
            lib().spmv_c_ipc_close(c_void_p(p))
        self._peer_ptrs = None
        self._peer_keepalive = []
        self.mode = "gather"
        return False

    def reset(self):
        start = float(np.float32(1.0) / np.float32(self.n))
        for buf in self.r:
            buf.zero_()
            buf[self._pos] = start
        self.engine.reset(initial_dangling_mass(self.num_dangling, self.n))
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            # peers may store into these vectors (push mode): nobody proceeds before every rank's
            # vectors hold the start state
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)

    def iterate(self, k, damping, tolerance):
        """Enqueue iteration k (0-based): r[k & 1] -> r[(k + 1) & 1]."""
        self.watch.check()
        try:
            self._iterate(k, damping, tolerance)
        except Exception as exc:            # noqa: BLE001 - told to the peers, then raised as it is
            self.watch.report(exc)
            raise

    def _iterate(self, k, damping, tolerance):
        r_old, r_new = self.r[k & 1], self.r[(k + 1) & 1]
        if not self.layout.exchange:
            if hasattr(self.engine, "step_and_commit"):
                self.engine.step_and_commit(r_old, r_new, damping, tolerance)
            else:
                self.engine.commit(self.engine.step(r_old, r_new, damping), tolerance)
            return
        if self.mode == "push":
            sums = self.engine.step(r_old, r_new, damping, push_to=self._peer_ptrs[(k + 1) & 1])
            dist.all_reduce(sums, group=self.group)        # 16 bytes: the sums and the cross-GPU barrier
            self.engine.commit(sums, tolerance)
            return
        lay = self.layout
        self.engine.step(r_old, r_new, damping, self._my_tail(r_new))
        if lay.chunks == 1:
            dist.all_gather_into_tensor(r_new, self._my_slice(r_new), group=self.group)
            self.engine.commit_gathered(r_new, tolerance)
            return
        # overlapped: one all-gather per block, issued back to back; as each completes, the next step's
        # products for the columns of that block are computed (the engine skips them when the step comes)
        works = []
        views = self._views(r_new)
        if self._comm is not None:
            stepped = torch.cuda.Event()
            stepped.record()
            self._comm.wait_event(stepped)
            with torch.cuda.stream(self._comm):
                for c in range(lay.chunks):
                    works.append(dist.all_gather_into_tensor(views["blocks"][c], views["pieces"][c],
                                                             group=self.group, async_op=True))
        else:
            for c in range(lay.chunks):
                works.append(dist.all_gather_into_tensor(views["blocks"][c], views["pieces"][c],
                                                         group=self.group, async_op=True))
        for c, work in enumerate(works):
            work.wait()                         # the compute stream waits for block c (no host block on a GPU)
            if c + 1 < lay.chunks and hasattr(self.engine, "expand"):
                self.engine.expand(r_new, (c + 1) * lay.block)
        self.engine.commit_gathered(r_new, tolerance)

    def run(self, damping=0.85, tolerance=1e-6, max_iterations=100, check_every=1):
        """Full PageRank; returns (ranks[n] float32 numpy, iterations, final_residual, converged).
        Steps enqueued after convergence are no-ops on every rank (device-side `done` flag),
        so `check_every` > 1 only trades host syncs for a few empty launches."""
        try:
            self.reset()
            for k in range(max_iterations):
                self.iterate(k, damping, tolerance)
                if (k + 1) % check_every == 0 and self.engine.status()[3]:
                    break
            iterations, residual, converged, _ = self.engine.status()
        except Exception as exc:            # noqa: BLE001 - an engine error (SpMVError from the C ABI) or a failed collective
            self.watch.report(exc)
            raise
        last = self.r[iterations & 1][self._pos].to("cpu").numpy().copy()
        total = np.float32(last.sum(dtype=np.float64))
        if total > 0:
            last = (last / total).astype(np.float32)
        return last, iterations, residual, converged
