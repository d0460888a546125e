"""GPU tests of the device-resident PageRank (C ABI spmv_c_pagerank + the shard engine)
against the CPU oracle (restatement of reference src/pagerank.cu:50-153).

Tolerance: EVERY rank is held to 1e-5 RELATIVE error against the oracle (every rank is at least
0.15 / n > 0, so the relative error is defined everywhere).  The device accumulates the residual /
dangling mass / final sum in double and reorders each row's SpMV sum, so the iteration at which the
residual crosses `tolerance` may differ by one from the oracle's; the vectors are then compared at
EQUAL iteration counts (both re-run with tolerance = 0 and max_iterations = the smaller count —
the result of such a run is the last computed vector, exactly what a converged run returns).

Parity caveat: src/pagerank.cu needs the CUDA runtime and cannot be built in this
image, so `oracle_pagerank` is a restatement pinned by the reference tests' known answers (3-cycle,
invariants) only — the numeric trajectory is "parity unpinned" beyond those."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def upload(spmv, rp, ci, va, n):
    A = spmv.csr_from_arrays(n, n, rp, ci, va)
    assert spmv.csr_to_gpu(A) == 0
    return A


def graph(spmv, n, k, seed, dangling=()):
    rp, ci, _ = spmv.synth.uniform_csr(seed, 0, n, n, k)
    keep = ~np.isin(ci, np.array(list(dangling), dtype=np.int32))
    counts = np.add.reduceat(keep.astype(np.int64), rp[:-1])
    ci = ci[keep]
    rp = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return rp, ci, spmv.synth.column_stochastic_values(ci, n)


RTOL = 1e-5        # north-star tolerance, relative, on every element


def worst_rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape and (want > 0).all()
    return float(np.max(np.abs(got - want) / want))


def compare(got, want, rtol=RTOL):
    worst = worst_rel(got, want)
    assert worst <= rtol, "worst relative rank error %.3g > %.1g" % (worst, rtol)


def assert_parity(gpu, oracle, A, rp, ci, va, n, result, wide_sums=True, damping=0.85, tolerance=1e-6,
                  max_iterations=100):
    """`result` = gpu.pagerank(A, (damping, tolerance, max_iterations)).  Flags and iteration count
    against the oracle (count +-1), then every rank at 1e-5 relative at equal iteration counts."""
    want, iters, res, conv = oracle.pagerank(rp, ci, va, num_cols=n, damping=damping, tolerance=tolerance,
                                             max_iterations=max_iterations, wide_sums=wide_sums)
    assert result.converged == conv and abs(result.iterations - iters) <= 1
    if result.iterations == iters:
        compare(result.ranks, want)
        return
    k = min(result.iterations, iters)
    again = gpu.pagerank(A, gpu.PageRankConfig(damping, 0.0, k))
    assert again.iterations == k and not again.converged
    want_k, iters_k, _, _ = oracle.pagerank(rp, ci, va, num_cols=n, damping=damping, tolerance=0.0,
                                            max_iterations=k, wide_sums=wide_sums)
    assert iters_k == k
    compare(again.ranks, want_k)


def test_three_cycle_equal_ranks(gpu):
    """reference tests/test_pagerank.cu:140-164"""
    dense = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], np.float32)
    A = gpu.csr_create(0, 0, 0)
    gpu.csr_from_dense(A, dense, 3, 3)
    gpu.csr_to_gpu(A)
    r = gpu.pagerank(A)
    assert r.converged and np.allclose(r.ranks, 1 / 3, atol=1e-4)
    top = gpu.pagerank_top_k(r, 3, 2)                       # tests/test_pagerank.cu:166-189
    assert len(top) == 2 and top[0][1] >= top[1][1]
    gpu.csr_destroy(A)


def test_score_invariants_and_oracle_parity_small_graphs(gpu, oracle):
    """reference tests/test_pagerank.cu:18-77 (P15) + parity with the oracle at test sizes"""
    rng = np.random.default_rng(42)
    with_dangling = 0
    for _ in range(25):
        n = int(rng.integers(5, 50))
        adj = (rng.random((n, n)) < 0.2).astype(np.float32)
        col = adj.sum(axis=0)
        with_dangling += int((col == 0).any())
        adj = np.where(col > 0, adj / np.maximum(col, 1), 0).astype(np.float32)
        if not adj.any():
            continue
        A = gpu.csr_create(0, 0, 0)
        gpu.csr_from_dense(A, adj, n, n)
        gpu.csr_to_gpu(A)
        r = gpu.pagerank(A, gpu.PageRankConfig(0.85, 1e-6, 100))
        rp, ci, va = gpu.csr_host_arrays(A)
        assert (r.ranks >= 0).all() and abs(float(r.ranks.sum()) - 1.0) < 1e-4
        assert (not r.converged) or r.final_residual < 1e-6
        assert_parity(gpu, oracle, A, rp, ci, va, n, r, wide_sums=False)      # the reference's fp32 sequential sums
        gpu.csr_destroy(A)
    assert with_dangling >= 3


def test_medium_graph_with_dangling_nodes(gpu, oracle):
    n = 200_000
    rp, ci, va = graph(gpu, n, 12, 9, dangling=(5, 1000, 150_000))
    A = upload(gpu, rp, ci, va, n)
    r = gpu.pagerank(A, gpu.PageRankConfig(0.85, 1e-6, 100))
    assert_parity(gpu, oracle, A, rp, ci, va, n, r)
    top = gpu.pagerank_top_k(r, n, 10)                      # tests/test_pagerank.cu:81-137 (P16)
    assert all(top[i][1] >= top[i + 1][1] for i in range(9))
    assert top[0][1] == r.ranks.max()
    gpu.csr_destroy(A)


def test_max_iterations_and_tolerance_are_honoured(gpu, oracle):
    n = 5000
    rp, ci, va = graph(gpu, n, 8, 2)
    A = upload(gpu, rp, ci, va, n)
    r = gpu.pagerank(A, gpu.PageRankConfig(0.85, 0.0, 7))  # tol 0: never converges
    assert r.iterations == 7 and not r.converged
    want, *_ = oracle.pagerank(rp, ci, va, num_cols=n, tolerance=0.0, max_iterations=7, wide_sums=True)
    compare(r.ranks, want)
    r0 = gpu.pagerank(A, gpu.PageRankConfig(0.85, 1e-6, 0))
    assert r0.iterations == 0 and np.allclose(r0.ranks, 1.0 / n)
    gpu.csr_destroy(A)


def test_null_matrix_and_not_uploaded(gpu):
    """pagerank(nullptr) returns a default result (reference src/pagerank.cu:56-58)"""
    r = gpu.pagerank(None)
    assert r.ranks is None and r.iterations == 0 and not r.converged


def test_device_only_matrix_uses_device_dangling_scan(gpu, oracle):
    """Matrix generated straight into HBM (no host arrays): dangling columns come from the
    device column-sum kernel and must agree with the oracle's host scan."""
    wl = importlib.import_module("gpu-spmv_amd.workloads")
    n = 100_000
    A = wl.uniform_csr_device(42, n, n, 4)                  # k=4: e^-4 of the columns are empty
    wl.make_column_stochastic(A)
    rp, ci, va = A.to_host()
    assert oracle.dangling_mask(rp, ci, va, n).sum() > 100
    r = gpu.pagerank(A.handle, gpu.PageRankConfig(0.85, 1e-6, 100))
    assert_parity(gpu, oracle, A.handle, rp, ci, va, n, r)
    A.close()


def test_shard_engine_single_rank_equals_pagerank(gpu, oracle):
    """The multi-GPU host loop with world = 1 (HipEngine on torch tensors) must reproduce
    spmv_c_pagerank — same kernels, two hosts."""
    torch = pytest.importorskip("torch")
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
    n = 50_000
    rp, ci, va = graph(gpu, n, 10, 4, dangling=(7, 9))
    A = upload(gpu, rp, ci, va, n)
    direct = gpu.pagerank(A, gpu.PageRankConfig(0.85, 1e-6, 100))
    assert_parity(gpu, oracle, A, rp, ci, va, n, direct)
    dev = torch.device("cuda:0")
    lay = prd.Layout(n)
    eng = prd.HipEngine(torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev), lay)
    pr = prd.ShardedPageRank(eng, lay).prepare()
    ranks, iters, res, conv = pr.run(0.85, 1e-6, 100, check_every=3)
    assert pr.num_dangling == int(oracle.dangling_mask(rp, ci, va, n).sum()) >= 2
    assert conv == direct.converged and iters == direct.iterations
    np.testing.assert_allclose(ranks, direct.ranks, rtol=0, atol=1e-9)
    eng.close()
    gpu.csr_destroy(A)


def test_two_shards_on_one_gpu_equal_one_shard(gpu, oracle):
    """Row-shard simulator on one device: two HipEngines own the halves of the matrix in the
    padded world-2 layout; the host plays the all-gather (copies each engine's slice, tail
    included, into the other's vector), then both commit from the gathered tails.  Must match
    the unsharded run — not bit for bit: a shard's rebased row_ptrs change which 16-byte group
    an entry falls in, hence the order of a row's fp32 sum (last-ulp differences)."""
    torch = pytest.importorskip("torch")
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
    n = 40_001                                   # odd: shard_len is rounded up to even, last shard short
    rp, ci, va = graph(gpu, n, 10, 8, dangling=(11,))
    dev = torch.device("cuda:0")

    def engine(lay):
        b, e = lay.row_begin, lay.row_end
        lrp = torch.from_numpy((rp[b:e + 1] - rp[b]).astype(np.int32)).to(dev)
        lci = torch.from_numpy(lay.remap_columns(ci[rp[b]:rp[e]]).astype(np.int32)).to(dev)
        return prd.HipEngine(lrp, lci, torch.from_numpy(va[rp[b]:rp[e]]).to(dev), lay)

    whole_lay = prd.Layout(n)
    lays = [prd.Layout(n, 2, r) for r in range(2)]
    whole = prd.ShardedPageRank(engine(whole_lay), whole_lay).prepare()
    shards = [prd.ShardedPageRank(engine(l), l) for l in lays]
    sums = shards[0].engine.column_sums() + shards[1].engine.column_sums()      # the all-reduce of prepare()
    num_dangling = int(oracle.dangling_mask(rp, ci, va, n).sum())
    for sp in shards:
        mask = torch.zeros(sp.layout.padded, dtype=torch.uint8, device=dev)
        mask[sp._pos] = (sums[sp._pos] == 0).to(torch.uint8)
        sp.num_dangling = int(mask.sum().item())
        assert sp.num_dangling == whole.num_dangling == num_dangling >= 1
        sp.engine.set_dangling_mask(mask)
        sp.reset()
    whole.reset()
    for k in range(6):
        whole.iterate(k, 0.85, 0.0)
        news = [sp.r[(k + 1) & 1] for sp in shards]
        for sp in shards:
            sp.engine.step(sp.r[k & 1], sp.r[(k + 1) & 1], 0.85, sp._my_tail(sp.r[(k + 1) & 1]))
        for src in (0, 1):                                                      # the all-gather
            sl = slice(lays[src].row_offset, lays[src].row_offset + lays[src].stride)
            news[1 - src][sl] = news[src][sl]
        for sp in shards:
            sp.engine.commit_gathered(sp.r[(k + 1) & 1], 0.0)
        torch.testing.assert_close(news[0], news[1], rtol=0, atol=0)           # both ranks hold the same vector
        torch.testing.assert_close(news[0][shards[0]._pos], whole.r[(k + 1) & 1][whole._pos], rtol=2e-6, atol=0)
    # ... and the oracle after the same 6 steps: every rank, 1e-5 relative (the shard vectors are not normalised)
    want, *_ = oracle.pagerank(rp, ci, va, num_cols=n, tolerance=0.0, max_iterations=6, wide_sums=True)
    for vec in (news[0][shards[0]._pos], whole.r[0][whole._pos]):
        v = vec.double().cpu().numpy()
        compare(v / v.sum(), want)
    st = [sp.engine.status() for sp in shards] + [whole.engine.status()]
    assert [x[0] for x in st] == [6, 6, 6]
    assert st[0][1] == st[1][1] and abs(st[0][1] - st[2][1]) <= 1e-12 + 1e-6 * st[2][1]
    for sp in shards + [whole]:
        sp.engine.close()


@pytest.mark.parametrize("chunks,align,skewed", [(3, None, False), (2, 32768, False), (3, 8192, True)])
def test_head_start_on_a_chunk_major_vector(gpu, oracle, chunks, align, skewed):
    """The overlapped exchange's engine side (Layout(chunks=C) + spmv_c_pr_expand) on one device: two shards
    large enough for the tiled engine; the host plays the per-block all-gathers and declares each block ready
    as it "arrives".  Blocks that have not arrived hold NaN when the head start runs, and once everything has
    been declared ready the foreign pieces are poisoned again before the step — so a head start that reads
    ahead, or a step that multiplies a strip twice (or falls back to the direct kernel), shows up as NaN."""
    torch = pytest.importorskip("torch")
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
    n = 400_000
    if skewed:      # power-law rows: long rows (direct path, needs ALL of x) in both shards, shards cut for equal nnz
        lens = gpu.synth.power_law_lengths(31, n, max_len=20000, n_cols=n)
        rp, ci, _ = gpu.synth.stratified_csr(31, 0, lens, n)
        va = gpu.synth.column_stochastic_values(ci, n)
        bounds = prd.Layout.equal_nnz_bounds(rp, 2)
    else:
        rp, ci, va = graph(gpu, n, 8, 31, dangling=(5, 250_000))
        bounds = None
    dev = torch.device("cuda:0")

    def sharded(lays):
        out = []
        for lay in lays:
            b, e = lay.row_begin, lay.row_end
            lrp = torch.from_numpy((rp[b:e + 1] - rp[b]).astype(np.int32)).to(dev)
            lci = torch.from_numpy(lay.remap_columns(ci[rp[b]:rp[e]]).astype(np.int32)).to(dev)
            out.append(prd.ShardedPageRank(prd.HipEngine(lrp, lci, torch.from_numpy(va[rp[b]:rp[e]]).to(dev), lay), lay))
        sums = out[0].engine.column_sums() + out[1].engine.column_sums()
        for sp in out:
            mask = torch.zeros(sp.layout.padded, dtype=torch.uint8, device=dev)
            mask[sp._pos] = (sums[sp._pos] == 0).to(torch.uint8)
            sp.num_dangling = int(mask.sum().item())
            sp.engine.set_dangling_mask(mask)
            sp.reset()
        return out

    lays = [prd.Layout(n, 2, r, bounds=bounds, chunks=chunks, align=align) for r in range(2)]
    assert lays[0].chunks == chunks and (align is None or lays[0].block % align == 0)
    shards = sharded(lays)
    for sp in shards:
        assert gpu.csr_has_tiled_plan(sp.engine._A)          # else expand() is a no-op and the test says nothing
        assert not skewed or gpu.csr_tiled_info(sp.engine._A)["long_rows"] > 0
    plain = sharded([prd.Layout(n, 2, r, bounds=bounds) for r in range(2)])
    steps = 5
    for k in range(steps):
        olds = [sp.r[k & 1] for sp in shards]
        news = [sp.r[(k + 1) & 1] for sp in shards]
        for me, sp in enumerate(shards):
            if k > 0:                                          # everything was declared ready: the step must not
                for c in range(chunks):                        # look at the other rank's pieces again
                    olds[me][lays[me].piece_slice(c, 1 - me)] = float("nan")
            sp.engine.step(olds[me], news[me], 0.85, sp._my_tail(news[me]))
            for c in range(chunks):                            # nothing of the new vector has arrived yet
                news[me][lays[me].piece_slice(c, 1 - me)] = float("nan")
        for c in range(chunks):
            for src in (0, 1):                                 # all-gather of block c
                sl = lays[src].piece_slice(c, src)
                news[1 - src][sl] = news[src][sl]
            for me, sp in enumerate(shards):
                sp.engine.expand(news[me], (c + 1) * lays[me].block)
        for sp in shards:
            sp.engine.commit_gathered(sp.r[(k + 1) & 1], 0.0)
        # the one-block form, same shards
        for sp in plain:
            sp.engine.step(sp.r[k & 1], sp.r[(k + 1) & 1], 0.85, sp._my_tail(sp.r[(k + 1) & 1]))
        for src in (0, 1):
            sl = plain[src].layout.piece_slice(0, src)
            plain[1 - src].r[(k + 1) & 1][sl] = plain[src].r[(k + 1) & 1][sl]
        for sp in plain:
            sp.engine.commit_gathered(sp.r[(k + 1) & 1], 0.0)
        got = [news[me][shards[me]._pos] for me in (0, 1)]
        assert bool(torch.isfinite(got[0]).all())
        torch.testing.assert_close(got[0], got[1], rtol=0, atol=0)
        # (not bit for bit: the chunk-major numbering moves entries between strips, i.e. changes a row's summation order)
        torch.testing.assert_close(got[0], plain[0].r[(k + 1) & 1][plain[0]._pos], rtol=2e-6, atol=0)
    want, *_ = oracle.pagerank(rp, ci, va, num_cols=n, tolerance=0.0, max_iterations=steps, wide_sums=True)
    v = shards[0].r[steps & 1][shards[0]._pos].double().cpu().numpy()
    # The skewed graph has rows of up to 20 000 entries, which the oracle sums left to right in fp32 (the
    # reference's order): that sum alone wanders by ~sqrt(20000) * 6e-8 = 8e-6 per step, so five fixed steps
    # on two differently cut shards sit at 1.3e-5 of it.  (pagerank() on this same graph is held to 1e-5 in
    # test_repeated_calls_on_a_power_law_graph_with_long_rows; what THIS test pins is the head start, above.)
    compare(v / v.sum(), want, rtol=3e-5 if skewed else RTOL)
    st = [sp.engine.status() for sp in shards + plain]
    assert [x[0] for x in st] == [steps] * 4 and st[0][1] == st[1][1]
    assert abs(st[0][1] - st[2][1]) <= 1e-12 + 1e-5 * st[2][1]
    for sp in shards + plain:
        sp.engine.close()


@pytest.mark.parametrize("chunks", [1, 3])
def test_exchange_loop_through_rccl_with_one_rank(gpu, oracle, chunks):
    """The one-process-per-GPU loop on the REAL backend (torch.distributed "nccl" = RCCL) with a single rank:
    Layout(exchange=True) keeps tails, blocks and collectives in the loop although there is nobody to exchange
    with, so the in-place all_gather_into_tensor on the library's own device memory, the async collectives on
    the side stream, Work.wait() ordering and the head starts all run exactly as they do with eight ranks —
    only the data movement between devices is missing.  Result = the plain single-rank loop's."""
    torch = pytest.importorskip("torch")
    import socket
    import torch.distributed as dist
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
    n = 400_000
    rp, ci, va = graph(gpu, n, 8, 41, dangling=(9, 123_456))
    dev = torch.device("cuda:0")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    try:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    except Exception as exc:                                   # noqa: BLE001 - no RCCL in this environment
        pytest.skip("RCCL process group unavailable: %r" % (exc,))
    try:
        def loop(lay):
            d_ci = torch.from_numpy(ci).to(dev)
            d_ci = lay.remap_columns(d_ci).to(torch.int32).contiguous()
            eng = prd.HipEngine(torch.from_numpy(rp).to(dev), d_ci, torch.from_numpy(va).to(dev), lay)
            return prd.ShardedPageRank(eng, lay).prepare()
        plain = loop(prd.Layout(n))
        want_ranks, want_iters, _, want_conv = plain.run(0.85, 1e-6, 100, check_every=2)
        lay = prd.Layout(n, 1, 0, chunks=chunks, exchange=True)
        assert lay.exchange and lay.chunks == chunks and lay.padded >= n + 4
        through = loop(lay)
        assert gpu.csr_has_tiled_plan(through.engine._A)
        got_ranks, got_iters, _, got_conv = through.run(0.85, 1e-6, 100, check_every=2)
        assert got_conv == want_conv and got_iters == want_iters
        assert np.max(np.abs(got_ranks.astype(np.float64) - want_ranks) / want_ranks) <= 2e-6
        want, iters, _, conv = oracle.pagerank(rp, ci, va, num_cols=n, wide_sums=True)
        if iters != got_iters:
            want, *_ = oracle.pagerank(rp, ci, va, num_cols=n, tolerance=0.0, max_iterations=got_iters, wide_sums=True)
        compare(got_ranks, want)
        for sp in (plain, through):
            sp.engine.close()
            sp.close()
    finally:
        dist.destroy_process_group()


def test_a_shard_keeps_its_plan_when_the_side_table_replaces_it(gpu, oracle):
    """Two engines over the SAME row-pointer array but different column arrays (bench.py builds the plain and the
    chunk-major numbering of one shard that way): the side table is keyed by the row pointers, so the second
    engine's plan replaces the first's there, and closing either engine drops the entry.  A shard must go on
    working on the plan it started with (shared ownership), not on freed memory."""
    torch = pytest.importorskip("torch")
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
    n = 300_000
    rp, ci, va = graph(gpu, n, 8, 77)
    dev = torch.device("cuda:0")
    lay = prd.Layout(n)
    d_rp, d_va = torch.from_numpy(rp).to(dev), torch.from_numpy(va).to(dev)
    d_ci = torch.from_numpy(ci).to(dev)
    mirrored = (n - 1 - d_ci).contiguous()            # another numbering of the same graph's columns

    def run(loop, steps=4):
        loop.reset()
        for k in range(steps):
            loop.iterate(k, 0.85, 0.0)
        torch.cuda.synchronize()
        return loop.r[steps & 1].clone()

    first = prd.ShardedPageRank(prd.HipEngine(d_rp, d_ci, d_va, lay), lay).prepare()
    assert gpu.csr_has_tiled_plan(first.engine._A)
    before = run(first)
    second = prd.ShardedPageRank(prd.HipEngine(d_rp, mirrored, d_va, lay), lay).prepare()   # replaces the table's plan
    torch.testing.assert_close(run(first), before, rtol=0, atol=0)
    other = run(second)
    second.engine.close()                               # drops the table entry both matrices share
    torch.testing.assert_close(run(first), before, rtol=0, atol=0)
    assert bool(torch.isfinite(other).all())
    first.engine.close()


def test_pagerank_through_the_tiled_engine(gpu, oracle):
    """A graph large enough (n >= 262144, >= 1 M entries) that pagerank() runs its steps through
    the LDS-tiled engine; same answer as the oracle."""
    n = 300_000
    rp, ci, va = graph(gpu, n, 8, 21, dangling=(1, 77_777, 299_999))
    A = upload(gpu, rp, ci, va, n)
    r = gpu.pagerank(A, gpu.PageRankConfig(0.85, 1e-6, 100))
    assert gpu.csr_has_tiled_plan(A)
    assert_parity(gpu, oracle, A, rp, ci, va, n, r)
    # and a fixed number of steps (no convergence test in the way): every rank, 1e-5 relative
    fixed = gpu.pagerank(A, gpu.PageRankConfig(0.85, 0.0, 4))
    want4, *_ = oracle.pagerank(rp, ci, va, num_cols=n, tolerance=0.0, max_iterations=4, wide_sums=True)
    compare(fixed.ranks, want4)
    gpu.csr_destroy(A)


def test_pagerank_switches_to_the_tiled_engine_mid_run(gpu, oracle):
    """A matrix without a cached plan starts on the direct-gather step and switches to the tiled engine once
    it has spent about a plan's worth of time (SPMV_DEBUG=pr_plan_after=N steps; default 4): same answer as the
    oracle whether the switch comes after 0, 2 or 4 steps, or never (converged / ran out of steps first)."""
    import os
    n = 300_000
    rp, ci, va = graph(gpu, n, 8, 33, dangling=(2, 150_000))
    try:
        for after, steps, expect_plan in (("2", 100, True), ("0", 100, True), ("4", 3, False), ("4", 100, True)):
            os.environ["SPMV_DEBUG"] = "pr_plan_after=" + after
            A = upload(gpu, rp, ci, va, n)
            tol = 1e-6 if steps == 100 else 0.0
            r = gpu.pagerank(A, gpu.PageRankConfig(0.85, tol, steps))
            assert bool(gpu.csr_has_tiled_plan(A)) == expect_plan, (after, steps)
            assert_parity(gpu, oracle, A, rp, ci, va, n, r, tolerance=tol, max_iterations=steps)
            gpu.csr_destroy(A)
    finally:
        del os.environ["SPMV_DEBUG"]


def test_repeated_calls_on_a_power_law_graph_with_long_rows(gpu, oracle):
    """In-degrees follow a power law, so some rows are too long for the cells of the tiled engine: they are
    summed by the direct path (extra wavefronts of the phase-1 grid) into the seed vector.  A second pagerank() on the same matrix, and an SpMV after it, must not
    see anything the first call's run-ahead step left behind."""
    n = 400_000
    lens = gpu.synth.power_law_lengths(31, n, max_len=20000, n_cols=n)
    rp, ci, _ = gpu.synth.stratified_csr(31, 0, lens, n)
    va = gpu.synth.column_stochastic_values(ci, n)
    A = upload(gpu, rp, ci, va, n)
    first = gpu.pagerank(A, gpu.PageRankConfig(0.85, 1e-6, 100))
    info = gpu.csr_tiled_info(A)
    assert info is not None and info["long_rows"] > 100                    # the long-row (direct, seeded) path is in use
    second = gpu.pagerank(A, gpu.PageRankConfig(0.85, 1e-6, 100))
    for r in (first, second):
        assert_parity(gpu, oracle, A, rp, ci, va, n, r)
    x = np.abs(gpu.synth.vector(31, 2, n)) + np.float32(0.01)
    d_x, d_y = gpu.CudaBuffer(n), gpu.CudaBuffer(n)
    d_x.copyFromHost(x, n)
    assert gpu.spmv_csr(A, d_x, d_y, gpu.SpMVConfig(2, 256, True), n).error_code == 0
    got = d_y.copyToHost(n)
    ref = oracle.spmv_csr(rp, ci, va, x)
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)) <= 1e-5
    gpu.csr_destroy(A)


def test_config5_pagerank_full_size_against_the_oracle(gpu, oracle):
    """BASELINE config 5: PageRank (d = 0.85, tol = 1e-6) on the 10 M-node / 160 M-edge column-stochastic
    matrix built in HBM.  (1) the reference's size-independent properties (tests/test_pagerank.cu:18-77):
    ranks >= 0, sum = 1 (1e-4), converged => residual < tol; (2) the ORACLE on the full matrix for a fixed
    number of steps (tolerance 0, k = 2: no convergence test in the way), every one of the 10 M ranks at
    1e-5 relative — once through the value-folded plan pagerank() picks for this matrix, once through the
    general (value stream) plan (SPMV_TILED_FOLD=0), which is the path bench.py quotes."""
    import os
    wl = importlib.import_module("gpu-spmv_amd.workloads")
    n = 10_000_000
    A = wl.uniform_csr_device(42, n, n, 16)
    wl.make_column_stochastic(A)
    r = gpu.pagerank(A.handle, gpu.PageRankConfig(0.85, 1e-6, 100))
    assert r.converged and r.final_residual < 1e-6 and 2 <= r.iterations <= 100
    assert (r.ranks >= 0).all() and abs(float(r.ranks.sum(dtype=np.float64)) - 1.0) < 1e-4
    # this graph converges before the loop has spent a plan's worth of time on direct steps, so the first call
    # never builds one; the fixed-k runs below force the tiled engine from step 0
    assert not gpu.csr_has_tiled_plan(A.handle)
    os.environ["SPMV_DEBUG"] = "pr_plan_after=0"

    rp, ci, va = A.to_host()
    k = 2
    want, iters, _, _ = oracle.pagerank(rp, ci, va, num_cols=n, tolerance=0.0, max_iterations=k, wide_sums=True)
    assert iters == k
    fixed = gpu.pagerank(A.handle, gpu.PageRankConfig(0.85, 0.0, k))
    assert fixed.iterations == k
    info = gpu.csr_tiled_info(A.handle)
    assert info is not None and info.get("values_folded")
    compare(fixed.ranks, want)                                        # folded plan
    # the converged run, too: the oracle converges on this matrix within a handful of steps
    want_c, iters_c, _, conv_c = oracle.pagerank(rp, ci, va, num_cols=n, wide_sums=True)
    assert conv_c and abs(iters_c - r.iterations) <= 1
    if iters_c == r.iterations:
        compare(r.ranks, want_c)

    previous = os.environ.get("SPMV_TILED_FOLD")
    os.environ["SPMV_TILED_FOLD"] = "0"
    try:
        gpu.csr_invalidate_gpu_cache(A.handle)                        # drop the folded plan; the next call rebuilds
        general = gpu.pagerank(A.handle, gpu.PageRankConfig(0.85, 0.0, k))
        info = gpu.csr_tiled_info(A.handle)
        assert info is not None and not info.get("values_folded")
        compare(general.ranks, want)                                  # general plan
    finally:
        del os.environ["SPMV_DEBUG"]
        if previous is None:
            del os.environ["SPMV_TILED_FOLD"]
        else:
            os.environ["SPMV_TILED_FOLD"] = previous
    A.close()


def test_push_exchange_two_shards_on_one_gpu(gpu, oracle):
    """The push-style exchange (spmv_c_pr_step_push): each shard's step stores its new slice into
    the other shard's vector as well; the host only sums the two partial pairs (the all-reduce).
    Both vectors must stay identical and track the unsharded run.  (Across processes the peers'
    pointers come from IPC handles — rehearsed by `SPMV_BENCH_BACKEND=gloo bench.py --gpus 2`.)"""
    torch = pytest.importorskip("torch")
    from ctypes import c_void_p
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
    n = 300_000                                   # large enough for the tiled engine on both shards
    rp, ci, va = graph(gpu, n, 10, 12, dangling=(5,))
    dev = torch.device("cuda:0")

    def engine(lay):
        b, e = lay.row_begin, lay.row_end
        lrp = torch.from_numpy((rp[b:e + 1] - rp[b]).astype(np.int32)).to(dev)
        lci = torch.from_numpy(lay.remap_columns(ci[rp[b]:rp[e]]).astype(np.int32)).to(dev)
        return prd.HipEngine(lrp, lci, torch.from_numpy(va[rp[b]:rp[e]]).to(dev), lay)

    whole_lay = prd.Layout(n)
    whole = prd.ShardedPageRank(engine(whole_lay), whole_lay).prepare()
    lays = [prd.Layout(n, 2, r) for r in range(2)]
    shards = [prd.ShardedPageRank(engine(l), l) for l in lays]
    sums = shards[0].engine.column_sums() + shards[1].engine.column_sums()
    for sp in shards:
        mask = torch.zeros(sp.layout.padded, dtype=torch.uint8, device=dev)
        mask[sp._pos] = (sums[sp._pos] == 0).to(torch.uint8)
        sp.num_dangling = int(mask.sum().item())
        sp.engine.set_dangling_mask(mask)
        sp.reset()
    whole.reset()
    for k in range(5):
        whole.iterate(k, 0.85, 0.0)
        partial = []
        for me, other in ((0, 1), (1, 0)):
            target = (c_void_p * 1)(shards[other].r[(k + 1) & 1].data_ptr())
            partial.append(shards[me].engine.step(shards[me].r[k & 1], shards[me].r[(k + 1) & 1], 0.85,
                                                  push_to=target).clone())
        total = partial[0] + partial[1]
        for sp in shards:
            sp.engine.commit(total, 0.0)
        a, b = shards[0].r[(k + 1) & 1][shards[0]._pos], shards[1].r[(k + 1) & 1][shards[1]._pos]
        torch.testing.assert_close(a, b, rtol=0, atol=0)
        torch.testing.assert_close(a, whole.r[(k + 1) & 1][whole._pos], rtol=2e-6, atol=0)
    want, *_ = oracle.pagerank(rp, ci, va, num_cols=n, tolerance=0.0, max_iterations=5, wide_sums=True)
    v = a.double().cpu().numpy()
    compare(v / v.sum(), want)                                   # the pushed vector against the oracle, every rank
    assert shards[0].engine.status()[0] == shards[1].engine.status()[0] == whole.engine.status()[0] == 5
    for sp in shards + [whole]:
        sp.engine.close()


def test_top_k_on_the_device_for_large_vectors(gpu):
    """for >= 2^20 nodes pagerank_top_k selects on the device (radix select on
    the float bits); same answer as a full sort, descending, ties broken arbitrarily among equals."""
    rng = np.random.default_rng(3)
    n = 3_000_000
    ranks = rng.random(n, dtype=np.float32)
    ranks[rng.integers(0, n, 1000)] = np.float32(0.99999)            # a block of ties inside the top
    ranks /= ranks.sum(dtype=np.float64)
    result = gpu.PageRankResult(ranks.astype(np.float32), 1, 0.0, True)
    for k in (1, 10, 777, 5000):
        top = gpu.pagerank_top_k(result, n, k)
        got = np.array([r for _, r in top], np.float32)
        want = np.sort(result.ranks)[::-1][:k]
        np.testing.assert_array_equal(got, want)                     # the k largest values, in order
        ids = [i for i, _ in top]
        assert len(set(ids)) == k and all(result.ranks[i] == r for i, r in top)
    # negative values are not orderable by bit pattern: falls back to the host path, still right
    result.ranks[5] = -1.0
    top = gpu.pagerank_top_k(result, n, 3)
    np.testing.assert_array_equal(np.array([r for _, r in top], np.float32), np.sort(result.ranks)[::-1][:3])


def test_two_matrices_over_one_row_pointer_array_keep_their_own_dangling_masks(gpu, oracle):
    """Every k-per-row graph has the same row-pointer array, and the per-matrix workspace (with the cached dangling
    mask) is keyed by it: the mask must be recomputed when the column / value arrays under the handle are not the
    ones it was computed for (the reference recomputes it every call, src/pagerank.cu:20-48)."""
    torch = pytest.importorskip("torch")
    n, k = 20_000, 8
    rp, ci, _ = gpu.synth.uniform_csr(91, 0, n, n, k)
    lonely = np.array([5, 777, 19_999], dtype=np.int32)            # nobody links to these in the second graph
    ci2 = ci.copy()
    hit = np.isin(ci2, lonely)
    ci2[hit] = (ci2[hit] + 1) % n
    assert not np.isin(ci2, lonely).any()
    va1, va2 = gpu.synth.column_stochastic_values(ci, n), gpu.synth.column_stochastic_values(ci2, n)
    dev = torch.device("cuda:0")
    d_rp = torch.from_numpy(rp).to(dev)
    t1 = (torch.from_numpy(ci).to(dev), torch.from_numpy(va1).to(dev))
    t2 = (torch.from_numpy(ci2).to(dev), torch.from_numpy(va2).to(dev))
    A1 = gpu.csr_wrap_device(n, n, ci.size, d_rp.data_ptr(), t1[0].data_ptr(), t1[1].data_ptr())
    A2 = gpu.csr_wrap_device(n, n, ci2.size, d_rp.data_ptr(), t2[0].data_ptr(), t2[1].data_ptr())
    cfg = gpu.PageRankConfig(0.85, 0.0, 6)
    want1, *_ = oracle.pagerank(rp, ci, va1, num_cols=n, tolerance=0.0, max_iterations=6, wide_sums=True)
    want2, *_ = oracle.pagerank(rp, ci2, va2, num_cols=n, tolerance=0.0, max_iterations=6, wide_sums=True)
    assert worst_rel(want1, want2) > 1e-3                           # the two graphs do differ
    for _ in range(2):                                              # back and forth: neither inherits the other's mask
        compare(gpu.pagerank(A1, cfg).ranks, want1)
        compare(gpu.pagerank(A2, cfg).ranks, want2)
    gpu.csr_destroy(A1)
    gpu.csr_destroy(A2)


def test_spmv_num_gpus_on_a_box_with_fewer_devices_still_returns_ranks(gpu, oracle, monkeypatch):
    """`SPMV_NUM_GPUS=8 ./existing_binary` on a machine with one device: the sharded run cannot start
    and returns nothing; pagerank() must not pass that on — the reference's pagerank() always hands back an allocated
    ranks array (include/spmv/pagerank.h:29-36) and its callers index it unchecked."""
    n = 5000
    rp, ci, va = graph(gpu, n, 6, 3, dangling=(2, 4000))
    A = upload(gpu, rp, ci, va, n)
    monkeypatch.setenv("SPMV_NUM_GPUS", "8")
    r = gpu.pagerank(A, gpu.PageRankConfig(0.85, 1e-6, 100))
    monkeypatch.delenv("SPMV_NUM_GPUS")
    assert r.ranks is not None and len(r.ranks) == n
    assert_parity(gpu, oracle, A, rp, ci, va, n, r)
    gpu.csr_destroy(A)
