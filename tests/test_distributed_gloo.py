"""world_size-2 (and 3) `gloo` tests of the row-sharded PageRank host loop
(gpu-spmv_amd/pagerank_dist.py) on CPU: partitioning, the all-reduce of the two
partial sums, the in-place all-gather of the rank slices, convergence handling and
the post-convergence no-op steps.  The compute engine is a CPU test double built on
the oracle (explicitly injected here; the product's engine is HipEngine and has no
CPU route).  Sharded result must equal the unsharded oracle PageRank."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class OracleEngine:
    """CPU stand-in with HipEngine's interface: same per-step mathematics as
    csrc/pagerank.hip (fp32 update, double partial sums, device-side `done` flag),
    same padded vector layout (columns already remapped)."""

    def __init__(self, oracle, row_ptrs, cols, vals, layout):
        self.oracle, self.layout = oracle, layout
        self.row_ptrs, self.cols, self.vals = row_ptrs, cols, vals
        self.device = torch.device("cpu")
        self.state = dict(dangling_sum=np.float32(0), residual=0.0, iterations=0, converged=False, done=False)
        self.mask = None
        self._sums = torch.zeros(2, dtype=torch.float64)
        self.promised = []              # (cols_ready, copy of r_old[:cols_ready]) from expand(): must still hold at step()
        self.head_starts = 0

    def expand(self, r_old, cols_ready):
        """The product's engine multiplies the columns [0, cols_ready) now; the double checks the promise that
        they are final (step() compares them with what it finally sees)."""
        assert 0 < cols_ready <= self.layout.padded and cols_ready % self.layout.block == 0
        self.promised.append((cols_ready, r_old[:cols_ready].clone()))
        self.head_starts += 1

    def column_sums(self):
        sums = np.zeros(self.layout.padded, np.float32)
        np.add.at(sums, self.cols, self.vals)
        return torch.from_numpy(sums)

    def set_dangling_mask(self, mask):
        self.mask = mask.numpy()

    def reset(self, dangling_sum):
        self.state = dict(dangling_sum=np.float32(dangling_sum), residual=0.0, iterations=0, converged=False, done=False)

    def step(self, r_old, r_new, damping, sums_out=None):
        target = self._sums if sums_out is None else sums_out
        for cols_ready, seen in self.promised:
            assert torch.equal(r_old[:cols_ready], seen), "expand() was told about columns that changed afterwards"
        self.promised = []
        if self.state["done"]:
            return target
        lay = self.layout
        old = r_old.numpy()
        y = self.oracle.spmv_csr(self.row_ptrs, self.cols, self.vals, old)
        d = np.float32(damping)
        teleport = (np.float32(1.0) - d) / np.float32(lay.n)
        dterm = d * self.state["dangling_sum"] / np.float32(lay.n)
        fresh = (d * y + dterm + teleport).astype(np.float32)
        sl = lay.local_positions()
        diff = fresh - old[sl]
        r_new.numpy()[sl] = fresh
        target[0] = float(np.sum((diff * diff).astype(np.float32), dtype=np.float64))
        target[1] = float(np.sum(fresh[self.mask[sl] != 0], dtype=np.float64))
        return target

    def _apply(self, res2, mass, tolerance):
        res = np.float32(np.sqrt(res2))
        self.state["iterations"] += 1
        self.state["residual"] = float(res)
        self.state["dangling_sum"] = np.float32(mass)
        if res < np.float32(tolerance):
            self.state["converged"] = True
            self.state["done"] = True

    def commit(self, sums, tolerance):
        if not self.state["done"]:
            self._apply(float(sums[0]), float(sums[1]), tolerance)

    def commit_gathered(self, gathered, tolerance):
        if self.state["done"]:
            return
        lay = self.layout
        res2 = mass = 0.0
        for p in range(lay.world):
            tail = gathered[lay.tail_slice(p)].view(torch.float64)
            res2 += float(tail[0])
            mass += float(tail[1])
        self._apply(res2, mass, tolerance)

    def status(self):
        s = self.state
        return s["iterations"], s["residual"], s["converged"], s["done"]


def make_graph(spmv, n, k, seed, dangling_cols=()):
    rp, ci, _ = spmv.synth.uniform_csr(seed, 0, n, n, k)
    keep = ~np.isin(ci, np.array(list(dangling_cols), dtype=np.int32))
    counts = np.add.reduceat(keep.astype(np.int64), rp[:-1]) if n else np.zeros(0, np.int64)
    ci = ci[keep]
    rp = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    va = spmv.synth.column_stochastic_values(ci, n)
    return rp, ci, va


def make_power_law_graph(spmv, n, seed):
    """Rows sorted by length, longest first: equal-ROW shards would hand rank 0 most of the entries."""
    lens = np.sort(spmv.synth.power_law_lengths(seed, n, max_len=n // 2, n_cols=n))[::-1]
    rp, ci, _ = spmv.synth.stratified_csr(seed, 0, lens, n)
    return rp, ci, spmv.synth.column_stochastic_values(ci, n)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, k, seed, dangling, tol, max_iter, check_every, out_dir, power_law=False, chunks=1):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spmv = importlib.import_module("gpu-spmv_amd")
        prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
        oracle = importlib.import_module("oracle")
        if power_law:                       # unequal-nnz rows: boundaries by binary search on row_ptrs
            rp, ci, va = make_power_law_graph(spmv, n, seed)
            lay = prd.Layout(n, world, rank, bounds=prd.Layout.equal_nnz_bounds(rp, world), chunks=chunks)
        else:
            rp, ci, va = make_graph(spmv, n, k, seed, dangling)
            lay = prd.Layout(n, world, rank, chunks=chunks)
        b, e = lay.row_begin, lay.row_end
        lrp = (rp[b:e + 1] - rp[b]).astype(np.int32)
        lci, lva = lay.remap_columns(ci[rp[b]:rp[e]]).astype(np.int32), va[rp[b]:rp[e]]
        engine = OracleEngine(oracle, lrp, lci, lva, lay)
        pr = prd.ShardedPageRank(engine, lay).prepare()
        ranks, iters, res, conv = pr.run(0.85, tol, max_iter, check_every)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ranks=ranks, iters=iters, res=res, conv=conv,
                 num_dangling=pr.num_dangling, local_rows=lay.local_rows, local_nnz=int(lrp[-1]),
                 head_starts=engine.head_starts)
    finally:
        dist.destroy_process_group()


def _run(world, tmp_path, n=600, k=6, seed=5, dangling=(3, 77, 401), tol=1e-6, max_iter=100, check_every=1,
         power_law=False, chunks=1):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, k, seed, dangling, tol, max_iter, check_every, str(tmp_path), power_law,
                            chunks),
             nprocs=world, join=True)
    return [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pagerank_equals_unsharded_oracle(spmv, oracle, tmp_path, world):
    n, k, seed, dangling = 601, 6, 5, (3, 77, 401)        # 601: shards of unequal length (padding path)
    outs = _run(world, tmp_path, n=n, k=k, seed=seed, dangling=dangling)
    rp, ci, va = make_graph(spmv, n, k, seed, dangling)
    want, iters, res, conv = oracle.pagerank(rp, ci, va, num_cols=n, wide_sums=True)
    assert conv
    for o in outs:                                          # every rank holds the same full answer
        np.testing.assert_array_equal(o["ranks"], outs[0]["ranks"])
        assert int(o["num_dangling"]) == int(oracle.dangling_mask(rp, ci, va, n).sum()) >= len(dangling)
        assert bool(o["conv"]) and abs(int(o["iters"]) - iters) <= 1
        assert abs(float(o["ranks"].sum()) - 1.0) < 1e-4 and (o["ranks"] >= 0).all()
    # every rank at 1e-5 RELATIVE, at equal iteration counts (a run that stops one step apart is re-run
    # on the oracle with the sharded loop's count: tolerance 0 returns the last computed vector)
    got_iters = int(outs[0]["iters"])
    if got_iters != iters:
        want, *_ = oracle.pagerank(rp, ci, va, num_cols=n, tolerance=0.0, max_iterations=got_iters, wide_sums=True)
    assert np.max(np.abs(outs[0]["ranks"].astype(np.float64) - want) / want) <= 1e-5


@pytest.mark.parametrize("world", [2, 3])
def test_equal_nnz_shards_on_a_power_law_graph(spmv, oracle, tmp_path, world):
    """shard boundaries by binary search on row_ptrs for equal nnz.  Rows sorted longest
    first, so equal-row shards would be badly unbalanced; the nnz partition gives shards of very different
    ROW counts (the padded layout's stride = the longest), and the answer must still be the oracle's."""
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
    n, seed = 900, 11
    rp, ci, va = make_power_law_graph(spmv, n, seed)
    nnz = int(rp[-1])
    rows_equal = [int(rp[min((r + 1) * ((n + world - 1) // world), n)] - rp[min(r * ((n + world - 1) // world), n)])
                  for r in range(world)]
    assert max(rows_equal) > 1.5 * nnz / world                      # the equal-row cut IS unbalanced here
    bounds = prd.Layout.equal_nnz_bounds(rp, world)
    shares = np.diff(rp[bounds])
    assert shares.sum() == nnz and shares.max() <= nnz / world + int(np.diff(rp).max())    # within one row of even
    outs = _run(world, tmp_path, n=n, seed=seed, power_law=True, max_iter=200)
    assert [int(o["local_nnz"]) for o in outs] == shares.tolist()
    assert len({int(o["local_rows"]) for o in outs}) > 1            # unequal row counts went through the layout
    want, iters, res, conv = oracle.pagerank(rp, ci, va, num_cols=n, max_iterations=200, wide_sums=True)
    got_iters = int(outs[0]["iters"])
    assert bool(outs[0]["conv"]) == conv and abs(got_iters - iters) <= 1
    if got_iters != iters:
        want, *_ = oracle.pagerank(rp, ci, va, num_cols=n, tolerance=0.0, max_iterations=got_iters, wide_sums=True)
    for o in outs:
        np.testing.assert_array_equal(o["ranks"], outs[0]["ranks"])
    assert np.max(np.abs(outs[0]["ranks"].astype(np.float64) - want) / want) <= 1e-5


@pytest.mark.parametrize("world,chunks,power_law", [(2, 3, False), (3, 2, False), (2, 4, True)])
def test_overlapped_exchange_gives_the_same_bits(tmp_path, world, chunks, power_law):
    """Layout(chunks=C): C in-place all-gathers (one per block of the chunk-major vector) with the next step's
    head start after each.  The arithmetic per node does not change, so the ranks must equal the one-collective
    form's bit for bit; the double also checks that every column range declared ready really was final."""
    kw = dict(n=900, seed=11, power_law=True, max_iter=200) if power_law else dict(n=601)
    (tmp_path / "plain").mkdir()
    (tmp_path / "chunked").mkdir()
    plain = _run(world, tmp_path / "plain", **kw)
    chunked = _run(world, tmp_path / "chunked", chunks=chunks, **kw)
    for a, b in zip(plain, chunked):
        np.testing.assert_array_equal(a["ranks"], b["ranks"])
        assert int(a["iters"]) == int(b["iters"]) and float(a["res"]) == float(b["res"])
        assert int(a["head_starts"]) == 0 and int(b["head_starts"]) == (chunks - 1) * int(b["iters"])


def test_running_ahead_of_the_convergence_check_changes_nothing(tmp_path):
    a = _run(2, tmp_path / "a", check_every=1) if (tmp_path / "a").mkdir() is None else None
    b = _run(2, tmp_path / "b", check_every=7) if (tmp_path / "b").mkdir() is None else None
    np.testing.assert_array_equal(a[0]["ranks"], b[0]["ranks"])
    assert int(a[0]["iters"]) == int(b[0]["iters"])


def test_max_iterations_without_convergence(tmp_path):
    outs = _run(2, tmp_path, tol=0.0, max_iter=5)
    assert int(outs[0]["iters"]) == 5 and not bool(outs[0]["conv"])


def test_layout(spmv):
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
    for n, world in [(10, 1), (10, 2), (10, 3), (10, 8), (7, 8), (601, 3), (1_000_000, 8)]:
        covered = []
        for r in range(world):
            lay = prd.Layout(n, world, r)
            assert 0 <= lay.row_begin <= lay.row_end <= n and lay.local_rows <= lay.shard_len
            assert lay.stride == lay.shard_len + (4 if world > 1 else 0) and lay.padded == lay.stride * world
            assert world == 1 or (lay.shard_len % 2 == 0 and lay.stride % 2 == 0)      # 8-byte aligned tails
            covered += list(range(lay.row_begin, lay.row_end)) if n < 1000 else []
        if n < 1000:
            assert covered == list(range(n))
        lay = prd.Layout(n, world, 0)
        pos = lay.positions()
        assert len(set(pos.tolist())) == n and pos.max() < lay.padded
        cols = np.arange(n, dtype=np.int32)
        np.testing.assert_array_equal(lay.remap_columns(cols), pos)                    # columns address nodes
        tails = set()
        for r in range(world if world > 1 else 0):
            t = lay.tail_slice(r)
            tails |= set(range(t.start, t.stop))
        assert not (tails & set(pos.tolist()))
    # arbitrary row bounds (equal-nnz partition): positions unique, remap == positions, tails disjoint
    for bounds in ([0, 1, 10], [0, 7, 7, 10], [0, 0, 3, 10], [0, 601, 601]):
        n, world = bounds[-1], len(bounds) - 1
        lay = prd.Layout(n, world, 0, bounds=bounds)
        pos = lay.positions()
        assert len(set(pos.tolist())) == n and pos.max() < lay.padded and lay.stride % 2 == 0
        np.testing.assert_array_equal(lay.remap_columns(np.arange(n, dtype=np.int32)), pos)
        np.testing.assert_array_equal(lay.remap_columns(torch.arange(n, dtype=torch.int32)).numpy(), pos)
        tails = set()
        for r in range(world):
            lr = prd.Layout(n, world, r, bounds=bounds)
            assert (lr.row_begin, lr.row_end) == (bounds[r], bounds[r + 1]) and lr.local_rows <= lr.shard_len
            t = lay.tail_slice(r)
            tails |= set(range(t.start, t.stop))
        assert not (tails & set(pos.tolist()))
    # chunk-major layouts (overlapped exchange): positions unique and inside the vector, a rank's piece of block c
    # holds exactly its rows [c * piece, (c + 1) * piece), tails in the last block and disjoint from every node
    for n, world, chunks, align, bounds in [(10, 2, 2, None, None), (601, 3, 4, None, None), (601, 3, 4, 64, None),
                                            (10, 3, 3, None, [0, 0, 3, 10]), (1_000_000, 8, 4, 4096, None),
                                            (10, 1, 4, None, None)]:
        lay = prd.Layout(n, world, 0, bounds=bounds, chunks=chunks, align=align)
        pos = lay.positions()
        assert len(set(pos.tolist())) == n and pos.max() < lay.padded == lay.chunks * lay.block
        assert lay.block == lay.piece * world and lay.piece % 2 == 0
        assert world == 1 or align is None or lay.piece % align == 0
        np.testing.assert_array_equal(lay.remap_columns(np.arange(n, dtype=np.int32)), pos)
        np.testing.assert_array_equal(lay.remap_columns(torch.arange(n, dtype=torch.int32)).numpy(), pos)
        tails = set()
        for r in range(world):
            lr = prd.Layout(n, world, r, bounds=bounds, chunks=chunks, align=align)
            mine = lr.local_positions()
            np.testing.assert_array_equal(mine, pos[lr.row_begin:lr.row_end])
            base, piece, block = lr.row_map()
            i = np.arange(lr.local_rows)
            want = base + i if piece == 0x7FFFFFFF else base + (i // piece) * block + i % piece
            np.testing.assert_array_equal(mine, want)                                  # the engine's RowMap formula
            for c in range(lr.chunks):
                sl = lr.piece_slice(c)
                inside = mine[(mine >= sl.start) & (mine < sl.stop)]
                np.testing.assert_array_equal(inside, mine[c * lr.piece:(c + 1) * lr.piece])
                assert lr.block_slice(c).start <= sl.start and sl.stop <= lr.block_slice(c).stop
            if world > 1:
                t = lr.tail_slice()
                assert t.start % 2 == 0 and lr.block_slice(lr.chunks - 1).start <= t.start and t.stop == lr.piece_slice(lr.chunks - 1).stop
                tails |= set(range(t.start, t.stop))
        assert not (tails & set(pos.tolist())) and len(tails) == (4 * world if world > 1 else 0)
    rp = np.array([0, 10, 10, 11, 12, 40, 41], dtype=np.int32)
    np.testing.assert_array_equal(prd.Layout.equal_nnz_bounds(rp, 2), [0, 4, 6])
    assert prd.initial_dangling_mass(0, 10) == 0.0
    assert prd.initial_dangling_mass(3, 10) == float(np.float32(np.float32(np.float32(0.1) + np.float32(0.1)) + np.float32(0.1)))


# ---------------------------------------------------------------------------------------------------------------
# failure behaviour (interface /root/reference/include/spmv/pagerank.h:29-43:
# the single-GPU call reports failure through its result — a sharded loop must not turn it into a hang)
class _FailingEngine(OracleEngine):
    """Raises inside step() of iteration `fail_at` — what an SpMVError from the C ABI looks like to the loop."""

    def __init__(self, *args, fail_at=None):
        super().__init__(*args)
        self.fail_at, self.steps = fail_at, 0

    def step(self, *args, **kwargs):
        if self.fail_at is not None and self.steps == self.fail_at:
            raise RuntimeError("injected engine failure (CUDA_KERNEL_LAUNCH)")
        self.steps += 1
        return super().step(*args, **kwargs)


def _failing_worker(rank, world, port, failing_rank, out_dir, linger=0.0):
    """Started as a plain process (not mp.spawn: its join would end the survivors itself, which is exactly what is
    under test here).  The surviving ranks block in the all-gather of the iteration the failing rank never finishes."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
    spmv = importlib.import_module("gpu-spmv_amd")
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
    oracle = importlib.import_module("oracle")
    rp, ci, va = make_graph(spmv, 400, 5, 9, ())
    lay = prd.Layout(400, world, rank)
    b, e = lay.row_begin, lay.row_end
    lrp = (rp[b:e + 1] - rp[b]).astype(np.int32)
    lci, lva = lay.remap_columns(ci[rp[b]:rp[e]]).astype(np.int32), va[rp[b]:rp[e]]
    engine = _FailingEngine(oracle, lrp, lci, lva, lay, fail_at=3 if rank == failing_rank else None)
    watch = prd.FailureWatch(rank, world, poll=0.1, grace=1.0)
    pr = prd.ShardedPageRank(engine, lay, watch=watch).prepare()
    open(os.path.join(out_dir, f"started{rank}"), "w").close()
    try:
        pr.run(0.85, 0.0, 50)      # tolerance 0: never converges, so rank `failing_rank` does reach step 3
    except Exception:
        if linger:                 # a failed rank that stays around (its sockets stay open: gloo on the peer sees nothing)
            import time
            time.sleep(linger)
        raise
    open(os.path.join(out_dir, f"finished{rank}"), "w").close()


@pytest.mark.parametrize("failing_rank", [1, 0])
def test_one_rank_failing_mid_loop_ends_every_rank_non_zero(tmp_path, failing_rank):
    import time
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, failing_rank, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    deadline = time.time() + 120
    for p in procs:
        p.join(timeout=max(1.0, deadline - time.time()))
    alive = [p.is_alive() for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert not any(alive), "a rank was still blocked two minutes after its peer failed"
    assert all(os.path.exists(os.path.join(tmp_path, f"started{r}")) for r in range(2))
    assert not any(os.path.exists(os.path.join(tmp_path, f"finished{r}")) for r in range(2))
    codes = [p.exitcode for p in procs]
    assert all(c not in (0, None) for c in codes), codes       # every process says it failed


def test_a_blocked_rank_is_released_while_the_failed_peer_is_still_alive(tmp_path):
    """The failed rank lingers for a minute with its connections open, so nothing but the failure watch can tell the
    other rank, which sits in the all-gather: it must leave with EXIT_PEER_FAILED within seconds, not after the minute."""
    import time
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, 1, str(tmp_path), 60.0)) for r in range(2)]
    for p in procs:
        p.start()
    t0 = time.time()
    procs[0].join(timeout=45)
    waited = time.time() - t0
    survivor_alive = procs[0].is_alive()
    for p in procs:
        if p.is_alive():
            p.kill()
    assert not survivor_alive and waited < 45, "the blocked rank was not released"
    assert procs[0].exitcode == prd.EXIT_PEER_FAILED, procs[0].exitcode
