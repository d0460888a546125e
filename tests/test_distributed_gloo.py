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
    csrc/pagerank.hip (fp32 update, double partial sums, device-side `done` flag)."""

    def __init__(self, oracle, row_ptrs, cols, vals, row_begin, n):
        self.oracle, self.n, self.row_begin = oracle, n, row_begin
        self.row_ptrs, self.cols, self.vals = row_ptrs, cols, vals
        self.local_rows = len(row_ptrs) - 1
        self.device = torch.device("cpu")
        self.state = dict(dangling_sum=np.float32(0), residual=0.0, iterations=0, converged=False, done=False)
        self.mask = None

    def column_sums(self):
        sums = np.zeros(self.n, np.float32)
        np.add.at(sums, self.cols, self.vals)
        return torch.from_numpy(sums)

    def set_dangling_mask(self, mask):
        self.mask = mask.numpy()

    def reset(self, dangling_sum):
        self.state = dict(dangling_sum=np.float32(dangling_sum), residual=0.0, iterations=0, converged=False, done=False)

    def step(self, r_old, r_new, damping):
        sums = torch.zeros(2, dtype=torch.float64)
        if self.state["done"]:
            return self._last_sums
        old = r_old.numpy()
        y = self.oracle.spmv_csr(self.row_ptrs, self.cols, self.vals, old[: self.n])
        d = np.float32(damping)
        teleport = (np.float32(1.0) - d) / np.float32(self.n)
        dterm = d * self.state["dangling_sum"] / np.float32(self.n)
        fresh = (d * y + dterm + teleport).astype(np.float32)
        sl = slice(self.row_begin, self.row_begin + self.local_rows)
        diff = fresh - old[sl]
        sums[0] = float(np.sum((diff * diff).astype(np.float32), dtype=np.float64))
        sums[1] = float(np.sum(fresh[self.mask[sl] != 0], dtype=np.float64))
        r_new.numpy()[sl] = fresh
        self._last_sums = sums
        return sums

    def commit(self, sums, tolerance):
        if self.state["done"]:
            return
        res = np.float32(np.sqrt(float(sums[0])))
        self.state["iterations"] += 1
        self.state["residual"] = float(res)
        self.state["dangling_sum"] = np.float32(float(sums[1]))
        if res < np.float32(tolerance):
            self.state["converged"] = True
            self.state["done"] = True

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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, k, seed, dangling, tol, max_iter, check_every, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spmv = importlib.import_module("gpu-spmv_amd")
        prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
        oracle = importlib.import_module("oracle")
        rp, ci, va = make_graph(spmv, n, k, seed, dangling)
        shard_len, b, e = prd.shard_bounds(n, world, rank)
        lrp = (rp[b:e + 1] - rp[b]).astype(np.int32)
        lci, lva = ci[rp[b]:rp[e]], va[rp[b]:rp[e]]
        engine = OracleEngine(oracle, lrp, lci, lva, b, n)
        pr = prd.ShardedPageRank(engine, n, rank, world).prepare()
        ranks, iters, res, conv = pr.run(0.85, tol, max_iter, check_every)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ranks=ranks, iters=iters, res=res, conv=conv,
                 num_dangling=pr.num_dangling)
    finally:
        dist.destroy_process_group()


def _run(world, tmp_path, n=600, k=6, seed=5, dangling=(3, 77, 401), tol=1e-6, max_iter=100, check_every=1):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, k, seed, dangling, tol, max_iter, check_every, str(tmp_path)),
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
        assert np.max(np.abs(o["ranks"] - want)) < 1e-6
        assert abs(float(o["ranks"].sum()) - 1.0) < 1e-4 and (o["ranks"] >= 0).all()


def test_running_ahead_of_the_convergence_check_changes_nothing(tmp_path):
    a = _run(2, tmp_path / "a", check_every=1) if (tmp_path / "a").mkdir() is None else None
    b = _run(2, tmp_path / "b", check_every=7) if (tmp_path / "b").mkdir() is None else None
    np.testing.assert_array_equal(a[0]["ranks"], b[0]["ranks"])
    assert int(a[0]["iters"]) == int(b[0]["iters"])


def test_max_iterations_without_convergence(tmp_path):
    outs = _run(2, tmp_path, tol=0.0, max_iter=5)
    assert int(outs[0]["iters"]) == 5 and not bool(outs[0]["conv"])


def test_shard_bounds(spmv):
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
    for n, world in [(10, 1), (10, 2), (10, 3), (10, 8), (7, 8), (1_000_000, 8)]:
        covered, shard_len = [], None
        for r in range(world):
            sl, b, e = prd.shard_bounds(n, world, r)
            shard_len = sl
            assert 0 <= b <= e <= n and e - b <= sl
            covered += list(range(b, e)) if n < 100 else []
        if n < 100:
            assert covered == list(range(n))
        assert shard_len * world >= n
    assert prd.initial_dangling_mass(0, 10) == 0.0
    assert prd.initial_dangling_mass(3, 10) == float(np.float32(np.float32(np.float32(0.1) + np.float32(0.1)) + np.float32(0.1)))
