"""GPU tests of the boundary plumbing: CudaBuffer, host<->HBM transfer of the
containers, device generators == numpy twins, bandwidth metrics, benchmark-style runs."""
import importlib

import numpy as np
import pytest

from conftest import reorder_err

pytestmark = pytest.mark.gpu


def test_cuda_buffer_semantics(gpu):
    """reference tests/test_common.cpp:21-98"""
    b = gpu.CudaBuffer(100)
    assert b.get() is not None and b.size() == 100 and not b.empty()
    data = np.arange(100, dtype=np.float32)
    b.copyFromHost(data, 100)
    np.testing.assert_array_equal(b.copyToHost(100), data)
    with pytest.raises(RuntimeError, match="Copy size exceeds buffer size"):
        b.copyFromHost(np.zeros(101, np.float32), 101)
    moved = b.move()                                         # move construction
    assert b.get() is None and b.size() == 0 and moved.size() == 100
    np.testing.assert_array_equal(moved.copyToHost(100), data)
    moved.resize(200)                                        # resize discards contents
    assert moved.size() == 200 and moved.get() is not None
    moved.resize(200)
    moved.release()
    assert moved.get() is None and moved.size() == 0
    ib = gpu.CudaBuffer(8, "int32")
    ib.copyFromHost(np.arange(8, dtype=np.int32), 8)
    np.testing.assert_array_equal(ib.copyToHost(), np.arange(8, dtype=np.int32))


def test_csr_and_ell_gpu_round_trip(gpu):
    """reference tests/test_csr.cpp:153-200 and tests/test_ell.cpp GPUTransfer"""
    rp, ci, va = gpu.synth.uniform_csr(3, 0, 300, 400, 6)
    A = gpu.csr_from_arrays(300, 400, rp, ci, va)
    assert gpu.csr_to_gpu(A) == 0
    m = A.contents
    assert m.d_values and m.d_col_indices and m.d_row_ptrs and m.owns_device_memory
    import ctypes
    ctypes.memset(m.values, 0, va.nbytes)                    # wipe the host copy, pull it back
    ctypes.memset(m.col_indices, 0, ci.nbytes)
    assert gpu.csr_from_gpu(A) == 0
    rp2, ci2, va2 = gpu.csr_host_arrays(A)
    np.testing.assert_array_equal(ci2, ci)
    np.testing.assert_array_equal(va2, va)
    assert gpu.csr_to_gpu(A) == 0                            # re-upload frees the old arrays first
    gpu.csr_free_gpu(A)
    assert not A.contents.d_values and not A.contents.owns_device_memory

    E = gpu.ell_create(0, 0, 0)
    assert gpu.ell_from_csr(E, A) == 0 and gpu.ell_to_gpu(E) == 0
    ecols, evals = gpu.ell_host_arrays(E)
    ctypes.memset(E.contents.values, 0, evals.nbytes)
    assert gpu.ell_from_gpu(E) == 0
    np.testing.assert_array_equal(gpu.ell_host_arrays(E)[1], evals)
    gpu.ell_destroy(E)
    gpu.csr_destroy(A)


def test_device_generators_equal_numpy_twins(gpu):
    wl = importlib.import_module("gpu-spmv_amd.workloads")
    A = wl.uniform_csr_device(42, 5000, 77777, 16, row_begin=123)
    rp, ci, va = A.to_host()
    hrp, hci, hva = gpu.synth.uniform_csr(42, 123, 5000, 77777, 16)
    np.testing.assert_array_equal(rp, hrp)
    np.testing.assert_array_equal(ci, hci)
    np.testing.assert_array_equal(va.view(np.uint32), hva.view(np.uint32))
    A.close()

    B = wl.power_law_csr_device(42, 20000, 50000, max_len=3000)
    rp, ci, va = B.to_host()
    lens = gpu.synth.power_law_lengths(42, 20000, max_len=3000, n_cols=50000)
    hrp, hci, hva = gpu.synth.stratified_csr(42, 0, lens, 50000)
    np.testing.assert_array_equal(rp, hrp)
    np.testing.assert_array_equal(ci, hci)
    np.testing.assert_array_equal(va.view(np.uint32), hva.view(np.uint32))
    wl.make_column_stochastic(B)
    np.testing.assert_array_equal(B.values.copyToHost(B.nnz).view(np.uint32),
                                  gpu.synth.column_stochastic_values(hci, 50000).view(np.uint32))
    B.close()

    x = wl.vector_device(42, 9, 10001)
    np.testing.assert_array_equal(x.copyToHost(10001).view(np.uint32), gpu.synth.vector(42, 9, 10001).view(np.uint32))


def test_device_only_matrix_stats_and_selector(gpu, oracle):
    """csr_compute_stats / spmv_auto_config on a wrapped device matrix (no host row_ptrs)"""
    wl = importlib.import_module("gpu-spmv_amd.workloads")
    B = wl.power_law_csr_device(7, 30000, 40000, max_len=2000)
    rp, _, _ = B.to_host()
    st = gpu.csr_compute_stats(B.handle)
    avg, mx, mn, skew = oracle.csr_stats(rp)
    assert (st.max_nnz_per_row, st.min_nnz_per_row) == (mx, mn) and st.skewness == pytest.approx(skew)
    cfg = gpu.spmv_auto_config(B.handle)
    assert cfg.kernel_type == gpu.SpMVConfig.MERGE_PATH and cfg.use_texture == 1
    B.close()


def test_bandwidth_metrics_properties(gpu):
    """reference tests/test_bandwidth.cu:19-64 (P12, peak range)"""
    assert gpu.get_gpu_peak_bandwidth() == 8000.0            # gfx950 table entry
    rp, ci, va = gpu.synth.uniform_csr(1, 0, 2000, 2000, 8)
    A = gpu.csr_from_arrays(2000, 2000, rp, ci, va)
    gpu.csr_to_gpu(A)
    d_x, d_y = gpu.CudaBuffer(2000), gpu.CudaBuffer(2000)
    d_x.copyFromHost(np.ones(2000, np.float32), 2000)
    for kt in (0, 1, 2):
        res = gpu.spmv_csr(A, d_x, d_y, gpu.SpMVConfig(kernel_type=kt), 2000)
        m = gpu.compute_bandwidth_csr(A, res.elapsed_ms)
        assert m.achieved_bandwidth_gb_s >= 0 and m.theoretical_bandwidth_gb_s > 0 and 0 <= m.efficiency <= 1
        assert res.bandwidth_gb_s == pytest.approx(m.achieved_bandwidth_gb_s, rel=1e-5)
        assert res.gflops == pytest.approx(2.0 * 16000 / (res.elapsed_ms * 1e6), rel=1e-4)
    gpu.csr_destroy(A)


def test_async_entry_point_on_a_stream(gpu, oracle):
    torch = pytest.importorskip("torch")
    rp, ci, va = gpu.synth.uniform_csr(1, 0, 3000, 3000, 9)
    x = gpu.synth.vector(1, 1, 3000)
    A = gpu.csr_from_arrays(3000, 3000, rp, ci, va)
    gpu.csr_to_gpu(A)
    stream = torch.cuda.Stream()
    tx = torch.from_numpy(x).cuda()
    ty = torch.empty(3000, device="cuda")
    with torch.cuda.stream(stream):
        for kt in (1, 2, 0):                                 # merge-path builds its tile table on first use
            assert gpu.spmv_csr_async(A, tx.data_ptr(), ty.data_ptr(), gpu.SpMVConfig(kernel_type=kt), 3000,
                                      stream.cuda_stream) == 0
    stream.synchronize()
    np.testing.assert_array_equal(ty.cpu().numpy(), oracle.spmv_csr(rp, ci, va, x))   # last launch: scalar
    gpu.csr_destroy(A)


def test_concurrent_streams_on_one_matrix_do_not_share_scratch(gpu, oracle):
    """The reference's kernels are stateless, so its callers may run SpMVs on ONE matrix from several streams
    at once.  Here the tiled engine (product stream, long-row sums) and merge-path (carry-out slots) write
    per-matrix scratch — one set per stream.  Four streams, four different x, many rounds in flight together:
    every y must be its own product (with shared scratch the kernels of different streams overwrite each
    other's products)."""
    torch = pytest.importorskip("torch")
    rows, cols = 300_000, 400_000
    lens = gpu.synth.power_law_lengths(5, rows, max_len=30000, n_cols=cols)      # long rows: the chunk sums are scratch too
    rp, ci, va = gpu.synth.stratified_csr(5, 0, lens, cols)
    A = gpu.csr_from_arrays(rows, cols, rp, ci, va)
    assert gpu.csr_to_gpu(A) == 0
    streams = [torch.cuda.Stream() for _ in range(4)]
    xs = [gpu.synth.vector(5, 10 + i, cols) * np.float32(1 + i) for i in range(4)]
    wants = [oracle.spmv_csr(rp, ci, va, x) for x in xs]
    txs = [torch.from_numpy(x).cuda() for x in xs]
    for cfg in (gpu.SpMVConfig(kernel_type=1, use_texture=True),        # tiled engine
                gpu.SpMVConfig(kernel_type=2, use_texture=False)):      # merge-path on the CSR arrays
        tys = [torch.zeros(rows, device="cuda") for _ in range(4)]
        # first call (builds the plan / the tile table) on stream 0, then everybody at once
        assert gpu.spmv_csr_async(A, txs[0].data_ptr(), tys[0].data_ptr(), cfg, cols, streams[0].cuda_stream) == 0
        torch.cuda.synchronize()
        for _ in range(12):
            for i, st in enumerate(streams):
                assert gpu.spmv_csr_async(A, txs[i].data_ptr(), tys[i].data_ptr(), cfg, cols, st.cuda_stream) == 0
        torch.cuda.synchronize()
        if cfg.use_texture:
            assert gpu.csr_has_tiled_plan(A) and gpu.csr_tiled_info(A)["long_rows"] > 0
        for i in range(4):
            got = tys[i].cpu().numpy()
            assert reorder_err(rp, ci, va, xs[i], wants[i], got) <= 1e-5, (cfg.kernel_type, i)
    gpu.csr_destroy(A)


def test_ell_from_csr_on_the_device_matches_the_host_conversion(gpu, oracle):
    """device-side ell_from_csr equals the reference's host conversion
    (src/ell_matrix.cpp:111-159) slab for slab, and feeds spmv_ell directly."""
    rng = np.random.default_rng(11)
    lens = rng.integers(0, 20, size=30001)
    lens[123] = 37
    rp, ci, va = gpu.synth.stratified_csr(3, 0, lens, 50000)
    A = gpu.csr_from_arrays(30001, 50000, rp, ci, va)
    assert gpu.csr_to_gpu(A) == 0
    E = gpu.ell_create(0, 0, 0)
    assert gpu.ell_from_csr_gpu(E, A) == 0
    e = E.contents
    assert (e.num_rows, e.num_cols, e.max_nnz_per_row) == (30001, 50000, 37) and e.d_values and e.owns_device_memory
    assert gpu.ell_from_gpu(E) == 0
    k, want_cols, want_vals = oracle.ell_from_csr(rp, ci, va)
    got_cols, got_vals = gpu.ell_host_arrays(E)
    np.testing.assert_array_equal(got_cols, want_cols)
    np.testing.assert_array_equal(got_vals.view(np.uint32), want_vals.view(np.uint32))
    x = gpu.synth.vector(3, 3, 50000)
    d_x, d_y = gpu.CudaBuffer(50000), gpu.CudaBuffer(30001)
    d_x.copyFromHost(x, 50000)
    assert gpu.spmv_ell(E, d_x, d_y, None, 50000).error_code == 0
    np.testing.assert_array_equal(d_y.copyToHost(30001), oracle.spmv_ell(30001, k, want_cols, want_vals, x))
    gpu.ell_destroy(E)
    B = gpu.csr_create(5, 5, 0)                              # not uploaded -> INVALID_FORMAT
    E2 = gpu.ell_create(0, 0, 0)
    assert gpu.ell_from_csr_gpu(E2, B) == gpu.SpMVError.INVALID_FORMAT
    gpu.ell_destroy(E2)
    gpu.csr_destroy(B)
    gpu.csr_destroy(A)


def test_async_spmv_and_pagerank_steps_replay_from_a_hip_graph(gpu, oracle):
    """The enqueue-only entry points are graph-safe once a matrix's auxiliary data exists:
    capture (a) one LDS-tiled SpMV and (b) two ping-pong PageRank steps into hipGraphs (through
    torch's capture API, which records whatever lands on its capture stream) and replay them."""
    torch = pytest.importorskip("torch")
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
    n = 200_000
    rp, ci, _ = gpu.synth.uniform_csr(6, 0, n, n, 8)
    va = gpu.synth.column_stochastic_values(ci, n)
    A = gpu.csr_from_arrays(n, n, rp, ci, va)
    gpu.csr_to_gpu(A)
    x = gpu.synth.vector(6, 1, n)
    tx, ty = torch.from_numpy(x).cuda(), torch.zeros(n, device="cuda")
    cfg = gpu.SpMVConfig(kernel_type=1, use_texture=True)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                            # warm-up outside capture: builds the plan
        assert gpu.spmv_csr_async(A, tx.data_ptr(), ty.data_ptr(), cfg, n, side.cuda_stream) == 0
    side.synchronize()
    assert gpu.csr_has_tiled_plan(A)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        assert gpu.spmv_csr_async(A, tx.data_ptr(), ty.data_ptr(), cfg, n,
                                  torch.cuda.current_stream().cuda_stream) == 0
    ty.zero_()
    tx.copy_(torch.from_numpy((2 * x).astype(np.float32)))   # new input, same graph
    graph.replay()
    torch.cuda.synchronize()
    want = oracle.spmv_csr(rp, ci, va, (2 * x).astype(np.float32))
    assert np.max(np.abs(ty.cpu().numpy() - want)) <= 1e-5 * max(1.0, float(np.abs(want).max()))

    # (b) PageRank: graph of two steps (A -> B -> A), replayed; equals the eager loop
    lay = prd.Layout(n)
    dev = torch.device("cuda:0")
    eng = prd.HipEngine(torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev), lay)
    pr = prd.ShardedPageRank(eng, lay).prepare()
    pr.reset()
    for k in range(6):
        pr.iterate(k, 0.85, 0.0)
    torch.cuda.synchronize()
    eager = pr.r[0].clone()
    pr.reset()
    pr.iterate(0, 0.85, 0.0)                                  # warm-up pair outside capture
    pr.iterate(1, 0.85, 0.0)
    torch.cuda.synchronize()
    pr.reset()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        pr.iterate(0, 0.85, 0.0)
        pr.iterate(1, 0.85, 0.0)
    pr.reset()
    for _ in range(3):
        g2.replay()
    torch.cuda.synchronize()
    assert eng.status()[0] == 6
    torch.testing.assert_close(pr.r[0], eager, rtol=2e-6, atol=0)
    eng.close()
    gpu.csr_destroy(A)


def test_host_threads_on_their_own_matrices(gpu, oracle):
    """Four host threads, each with its own matrix (two of them large enough for the tiled engine, so plan
    builds meet on the build lock), calling spmv_csr concurrently: per-thread event pairs, locked side tables."""
    import threading
    shapes = [(120_000, 150_000, 9), (3_000, 4_000, 11), (90_000, 200_000, 12), (500, 70_000, 30)]
    failures = []

    def worker(index, rows, cols, k):
        try:
            rp, ci, va = gpu.synth.uniform_csr(100 + index, 0, rows, cols, k)
            x = gpu.synth.vector(100 + index, 1, cols)
            want = oracle.spmv_csr(rp, ci, va, x)
            A = gpu.csr_from_arrays(rows, cols, rp, ci, va)
            assert gpu.csr_to_gpu(A) == 0
            d_x, d_y = gpu.CudaBuffer(cols), gpu.CudaBuffer(rows)
            d_x.copyFromHost(x, cols)
            for call in range(12):
                cfg = gpu.SpMVConfig(kernel_type=1 + call % 2, use_texture=True)
                assert gpu.spmv_csr(A, d_x, d_y, cfg, cols).error_code == 0
                assert reorder_err(rp, ci, va, x, want, d_y.copyToHost(rows)) <= 1e-5
            gpu.csr_destroy(A)
        except Exception as exc:            # noqa: BLE001 - reported by the main thread
            failures.append((index, repr(exc)))

    threads = [threading.Thread(target=worker, args=(i, *shape)) for i, shape in enumerate(shapes)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not failures, failures


def test_host_threads_sharing_one_matrix(gpu, oracle):
    """Four host threads on ONE matrix and the same (default) stream, each with its own x and y, racing to the
    first call (plan / tile-table build) and then calling spmv_csr in a loop through the tiled engine, merge-path
    and vector-CSR: the two launches of a call must stay together and the builds must happen once."""
    import threading
    rows, cols = 200_000, 300_000
    lens = gpu.synth.power_law_lengths(8, rows, max_len=20000, n_cols=cols)
    rp, ci, va = gpu.synth.stratified_csr(8, 0, lens, cols)
    A = gpu.csr_from_arrays(rows, cols, rp, ci, va)
    assert gpu.csr_to_gpu(A) == 0
    xs = [gpu.synth.vector(8, 20 + i, cols) * np.float32(1 + i) for i in range(4)]
    wants = [oracle.spmv_csr(rp, ci, va, x) for x in xs]
    failures = []
    gate = threading.Barrier(4)

    def worker(index):
        try:
            d_x, d_y = gpu.CudaBuffer(cols), gpu.CudaBuffer(rows)
            d_x.copyFromHost(xs[index], cols)
            gate.wait(timeout=60)
            for call in range(15):
                kind = call % 3
                cfg = gpu.SpMVConfig(kernel_type=(1, 2, 2)[kind], use_texture=kind != 2)      # tiled, tiled, direct merge-path
                assert gpu.spmv_csr(A, d_x, d_y, cfg, cols).error_code == 0
                assert reorder_err(rp, ci, va, xs[index], wants[index], d_y.copyToHost(rows)) <= 1e-5, (index, call)
        except Exception as exc:            # noqa: BLE001 - reported by the main thread
            failures.append((index, repr(exc)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not failures, failures
    assert gpu.csr_has_tiled_plan(A)
    gpu.csr_destroy(A)


def test_a_pagerank_loop_and_spmv_calls_share_a_matrix_and_a_stream(gpu, oracle):
    """A host thread inside pagerank(A) (steps through the tiled engine: two launches around the stream's product
    scratch) while others call spmv_csr(A, use_texture) on the same (default) stream: a step's pair of launches must
    not be split by another pair's phase 1 (the plan's launch lock covers both, csrc/pagerank.hip pr_step)."""
    import threading
    n, k = 300_000, 8
    rp, ci, _ = gpu.synth.uniform_csr(31, 0, n, n, k)
    va = gpu.synth.column_stochastic_values(ci, n)
    A = gpu.csr_from_arrays(n, n, rp, ci, va)
    assert gpu.csr_to_gpu(A) == 0
    cfg = gpu.PageRankConfig(0.85, 0.0, 12)
    alone = gpu.pagerank(A, cfg)                      # builds the plan after four direct steps
    assert gpu.csr_has_tiled_plan(A)
    alone = np.array(gpu.pagerank(A, cfg).ranks, copy=True)       # all twelve steps through the plan
    xs = [gpu.synth.vector(31, 40 + i, n) * np.float32(3 + i) for i in range(2)]
    wants = [oracle.spmv_csr(rp, ci, va, x) for x in xs]
    failures = []
    gate = threading.Barrier(3)

    def ranker():
        try:
            gate.wait(timeout=60)
            for _ in range(6):
                got = gpu.pagerank(A, cfg)
                assert np.array_equal(np.asarray(got.ranks), alone), "pagerank() changed beside concurrent spmv_csr calls"
        except Exception as exc:            # noqa: BLE001
            failures.append(("pagerank", repr(exc)))

    def multiplier(index):
        try:
            d_x, d_y = gpu.CudaBuffer(n), gpu.CudaBuffer(n)
            d_x.copyFromHost(xs[index], n)
            gate.wait(timeout=60)
            for call in range(40):
                assert gpu.spmv_csr(A, d_x, d_y, gpu.SpMVConfig(kernel_type=1, use_texture=True), n).error_code == 0
                assert reorder_err(rp, ci, va, xs[index], wants[index], d_y.copyToHost(n)) <= 1e-5, (index, call)
        except Exception as exc:            # noqa: BLE001
            failures.append((index, repr(exc)))

    threads = [threading.Thread(target=ranker)] + [threading.Thread(target=multiplier, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not failures, failures
    gpu.csr_destroy(A)


def test_merge_path_on_more_streams_than_the_matrix_keeps_scratch_for(gpu, oracle):
    """Every new stream used to add a carry pair to the matrix for good; past eight streams a call now borrows its
    pair from the stream-ordered allocator and returns it behind its kernels.  Twelve streams, results checked."""
    torch = pytest.importorskip("torch")
    rows, cols = 60_000, 50_000
    lens = gpu.synth.power_law_lengths(5, rows, max_len=4000, n_cols=cols)
    rp, ci, va = gpu.synth.stratified_csr(5, 0, lens, cols)
    x = gpu.synth.vector(5, 1, cols)
    want = oracle.spmv_csr(rp, ci, va, x)
    A = gpu.csr_from_arrays(rows, cols, rp, ci, va)
    assert gpu.csr_to_gpu(A) == 0
    d_x = torch.from_numpy(x).cuda()
    streams = [torch.cuda.Stream() for _ in range(12)]
    outs = [torch.empty(rows, dtype=torch.float32, device="cuda") for _ in streams]
    torch.cuda.synchronize()
    for _ in range(2):
        for s, y in zip(streams, outs):
            assert gpu.spmv_csr_async(A, d_x.data_ptr(), y.data_ptr(), gpu.SpMVConfig(kernel_type=2), cols, s.cuda_stream) == 0
    torch.cuda.synchronize()
    for y in outs:
        assert reorder_err(rp, ci, va, x, want, y.cpu().numpy()) <= 1e-5
    gpu.csr_destroy(A)
