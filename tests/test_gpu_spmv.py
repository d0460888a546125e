"""GPU parity tests: every SpMV kernel, called through the C ABI, against the CPU
oracle on the same inputs.  Tolerances: SCALAR_CSR and ELL reproduce the CPU's
summation order with unfused multiply/add and are held to BIT EQUALITY (stricter
than the reference's 1e-6, tests/test_spmv.cu:18-35).  VECTOR_CSR and MERGE_PATH
reorder a row's sum and are held to 1e-5 (BASELINE.json north_star) — relative to
|y_i| when the row has no cancellation (non-negative data: test_nonnegative_*),
and relative to max(|y_i|, sum_j |a_ij x_j|) on signed data (conftest.reorder_err)."""
import importlib

import numpy as np
import pytest

from conftest import max_rel_err, random_dense, reorder_err

pytestmark = pytest.mark.gpu

KERNELS = {"scalar": 0, "vector": 1, "merge": 2}
REORDER_TOL = 1e-5


def run_csr(spmv, row_ptrs, cols, vals, num_cols, x, kernel, vec_size=None):
    A = spmv.csr_from_arrays(len(row_ptrs) - 1, num_cols, row_ptrs, cols, vals)
    try:
        assert spmv.csr_to_gpu(A) == 0
        d_x = spmv.CudaBuffer(max(num_cols, 1))
        d_y = spmv.CudaBuffer(max(len(row_ptrs) - 1, 1))
        d_x.copyFromHost(x, len(x))
        cfg = spmv.SpMVConfig(kernel_type=kernel)
        res = spmv.spmv_csr(A, d_x, d_y, cfg, num_cols if vec_size is None else vec_size)
        assert res.error_code == 0, spmv.spmv_error_string(res.error_code)
        return d_y.copyToHost(len(row_ptrs) - 1), res
    finally:
        spmv.csr_destroy(A)


def check(spmv, oracle, row_ptrs, cols, vals, num_cols, x, kernels=KERNELS):
    want = oracle.spmv_csr(row_ptrs, cols, vals, x)
    for name, kt in kernels.items():
        got, _ = run_csr(spmv, row_ptrs, cols, vals, num_cols, x, kt)
        if name == "scalar":
            np.testing.assert_array_equal(got, want, err_msg=name)
        else:
            assert reorder_err(row_ptrs, cols, vals, x, want, got) <= REORDER_TOL, name


def test_nonnegative_data_is_within_1e5_relative(gpu, oracle):
    """No cancellation => plain relative error, the north-star's 1e-5, for the reordering kernels."""
    rp, ci, va = gpu.synth.uniform_csr(7, 0, 50000, 60000, 16)
    va = np.abs(va) + np.float32(0.01)
    x = np.abs(gpu.synth.vector(7, 3, 60000)) + np.float32(0.01)
    want = oracle.spmv_csr(rp, ci, va, x)
    for kt in (KERNELS["vector"], KERNELS["merge"]):
        got, _ = run_csr(gpu, rp, ci, va, 60000, x, kt)
        assert max_rel_err(want, got) <= REORDER_TOL


def test_property_random_dense_matrices(gpu, oracle):
    """reference tests/test_spmv.cu:40-78 (P8), widened to all three CSR kernels."""
    rng = np.random.default_rng(42)
    for _ in range(40):
        rows, cols = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        dense = random_dense(rng, rows, cols, rng.uniform(0.01, 0.3))
        x = rng.uniform(-10, 10, cols).astype(np.float32)
        rp, ci, va = oracle.csr_from_dense(dense)
        if ci.size == 0:
            continue
        check(gpu, oracle, rp, ci, va, cols, x)


@pytest.mark.parametrize("rows,cols,k", [(1000, 1000, 8), (4097, 5000, 16), (3000, 70000, 3),
                                         (777, 900, 33), (2048, 2048, 64), (5, 40, 1)])
def test_uniform_rows(gpu, oracle, rows, cols, k):
    rp, ci, va = gpu.synth.uniform_csr(42, 0, rows, cols, k)
    x = gpu.synth.vector(42, 7, cols)
    check(gpu, oracle, rp, ci, va, cols, x)


def test_power_law_rows_with_empty_rows(gpu, oracle):
    lens = gpu.synth.power_law_lengths(42, 20000, max_len=5000, n_cols=30000)
    lens[::7] = 0                       # sprinkle empty rows
    lens[0] = 0
    lens[-1] = 0
    rp, ci, va = gpu.synth.stratified_csr(42, 0, lens, 30000)
    x = gpu.synth.vector(42, 1, 30000)
    check(gpu, oracle, rp, ci, va, 30000, x)


def test_ragged_lengths_not_multiple_of_four(gpu, oracle):
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 23, size=5001)
    rp, ci, va = gpu.synth.stratified_csr(5, 0, lens, 4000)
    x = gpu.synth.vector(5, 2, 4000)
    check(gpu, oracle, rp, ci, va, 4000, x)


def test_single_long_row_and_single_row(gpu, oracle):
    lens = np.array([50000], dtype=np.int64)
    rp, ci, va = gpu.synth.stratified_csr(9, 0, lens, 60000)
    x = gpu.synth.vector(9, 3, 60000)
    check(gpu, oracle, rp, ci, va, 60000, x)
    lens = np.array([3, 0, 40000, 1, 0, 0, 9], dtype=np.int64)
    rp, ci, va = gpu.synth.stratified_csr(9, 0, lens, 60000)
    check(gpu, oracle, rp, ci, va, 60000, x)


def test_known_answers_from_reference_tests(gpu, oracle):
    """tests/test_spmv.cu:161-186 (5*2 = 10) and :188-218 ({3,0,7}); README.md:75-99 ({3,7,5})."""
    for dense, x, want in [
        (np.array([[5.0]], np.float32), np.array([2.0], np.float32), [10.0]),
        (np.array([[1, 2, 0], [0, 0, 0], [3, 0, 4]], np.float32), np.ones(3, np.float32), [3.0, 0.0, 7.0]),
        (np.array([[1, 0, 2], [0, 3, 4], [5, 0, 0]], np.float32), np.ones(3, np.float32), [3.0, 7.0, 5.0]),
    ]:
        rp, ci, va = oracle.csr_from_dense(dense)
        for kt in KERNELS.values():
            got, _ = run_csr(gpu, rp, ci, va, dense.shape[1], x, kt)
            np.testing.assert_array_equal(got, np.array(want, np.float32))


def test_default_config_and_ell_kernel_enum_fall_back_to_scalar(gpu, oracle):
    """nullptr config => SCALAR_CSR; ELL_KERNEL passed to spmv_csr behaves as scalar
    (reference src/spmv_kernels.cu:234-237, :287-288)."""
    rp, ci, va = gpu.synth.uniform_csr(1, 0, 300, 300, 5)
    x = gpu.synth.vector(1, 0, 300)
    want = oracle.spmv_csr(rp, ci, va, x)
    A = gpu.csr_from_arrays(300, 300, rp, ci, va)
    gpu.csr_to_gpu(A)
    d_x, d_y = gpu.CudaBuffer(300), gpu.CudaBuffer(300)
    d_x.copyFromHost(x, 300)
    res = gpu.spmv_csr(A, d_x, d_y, None, -1)          # vec_size = -1 skips the size check
    assert res.error_code == 0 and res.y == d_y.get()
    np.testing.assert_array_equal(d_y.copyToHost(300), want)
    res = gpu.spmv_csr(A, d_x, d_y, gpu.SpMVConfig(kernel_type=gpu.SpMVConfig.ELL_KERNEL), 300)
    assert res.error_code == 0
    np.testing.assert_array_equal(d_y.copyToHost(300), want)
    assert res.elapsed_ms > 0 and res.gflops > 0 and res.bandwidth_gb_s > 0
    gpu.csr_destroy(A)


def test_error_codes_in_reference_order(gpu):
    """null args -> -8; size mismatch -> -1; missing device arrays -> -5
    (reference src/spmv_kernels.cu:219-232)."""
    rp, ci, va = gpu.synth.uniform_csr(1, 0, 10, 10, 2)
    A = gpu.csr_from_arrays(10, 10, rp, ci, va)
    d = gpu.CudaBuffer(10)
    assert gpu.spmv_csr(None, d, d, None, 10).error_code == gpu.SpMVError.INVALID_ARGUMENT
    assert gpu.spmv_csr(A, None, d, None, 10).error_code == gpu.SpMVError.INVALID_ARGUMENT
    assert gpu.spmv_csr(A, d, d, None, 11).error_code == gpu.SpMVError.INVALID_DIMENSION
    assert gpu.spmv_csr(A, d, d, None, 10).error_code == gpu.SpMVError.INVALID_FORMAT   # not uploaded
    gpu.csr_destroy(A)


def test_empty_and_all_zero_matrices(gpu):
    """0x0 matrix is a successful no-op (reference tests/test_spmv.cu:148-159 expects it);
    nnz == 0 with rows > 0 succeeds with y = 0 (SURVEY.md D2/D3 deviations)."""
    A = gpu.csr_create(0, 0, 0)
    gpu.csr_to_gpu(A)
    d = gpu.CudaBuffer(1)
    assert gpu.spmv_csr(A, d, d, None, 1).error_code == 0
    gpu.csr_destroy(A)

    A = gpu.csr_create(3, 3, 0)
    assert gpu.csr_to_gpu(A) == 0
    d_x, d_y = gpu.CudaBuffer(3), gpu.CudaBuffer(3)
    d_x.copyFromHost(np.ones(3, np.float32), 3)
    d_y.copyFromHost(np.full(3, 9.0, np.float32), 3)
    for kt in KERNELS.values():
        assert gpu.spmv_csr(A, d_x, d_y, gpu.SpMVConfig(kernel_type=kt), 3).error_code == 0
        np.testing.assert_array_equal(d_y.copyToHost(3), np.zeros(3, np.float32))
    gpu.csr_destroy(A)


def test_nonfinite_x_outside_row_support_does_not_leak(gpu, oracle):
    """The 16-byte aligned loads read neighbours' entries; they must be masked by select."""
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 9, size=999)
    rp, ci, va = gpu.synth.stratified_csr(11, 0, lens, 5000)
    x = gpu.synth.vector(11, 0, 5000)
    x[ci[rp[500]:rp[501]]] = np.inf          # only row 500 (and rows sharing its columns) may see inf
    want = oracle.spmv_csr(rp, ci, va, x)
    got, _ = run_csr(gpu, rp, ci, va, 5000, x, KERNELS["vector"])
    finite = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), finite)
    x0 = np.where(np.isfinite(x), x, 0).astype(np.float32)
    assert reorder_err(rp, ci, va, x0, np.where(finite, want, 0), np.where(finite, got, 0)) <= REORDER_TOL


# ---------------------------------------------------------------- ELL ----------------
def run_ell(spmv, rows, cols, k, ell_cols, ell_vals, x):
    E = spmv.ell_create(rows, cols, k)
    import ctypes
    m = E.contents
    if rows * k:
        ctypes.memmove(m.col_indices, ell_cols.ctypes.data, ell_cols.nbytes)
        ctypes.memmove(m.values, ell_vals.ctypes.data, ell_vals.nbytes)
    assert spmv.ell_to_gpu(E) == 0
    d_x, d_y = spmv.CudaBuffer(max(cols, 1)), spmv.CudaBuffer(max(rows, 1))
    d_x.copyFromHost(x, cols)
    res = spmv.spmv_ell(E, d_x, d_y, None, cols)
    assert res.error_code == 0
    y = d_y.copyToHost(rows)
    spmv.ell_destroy(E)
    return y, res


@pytest.mark.parametrize("rows,cols,k", [(1000, 1000, 32), (1001, 1200, 7), (4096, 300, 5), (3, 9, 2)])
def test_ell_matches_oracle_bit_for_bit(gpu, oracle, rows, cols, k):
    """reference tests/test_spmv.cu:82-118 (P9); ragged rows so padding is exercised."""
    rng = np.random.default_rng(rows)
    lens = rng.integers(0, k + 1, size=rows)
    lens[rng.integers(0, rows)] = k
    rp, ci, va = gpu.synth.stratified_csr(4, 0, lens, cols)
    kk, ecols, evals = oracle.ell_from_csr(rp, ci, va)
    assert kk == k
    x = gpu.synth.vector(4, 9, cols)
    want = oracle.spmv_ell(rows, k, ecols, evals, x)
    got, res = run_ell(gpu, rows, cols, k, ecols, evals, x)
    np.testing.assert_array_equal(got, want)
    # gflops counts stored entries only (reference src/spmv_kernels.cu:398-407)
    assert res.gflops == pytest.approx(2.0 * ci.size / (res.elapsed_ms * 1e6), rel=1e-3)


# ------------------------------------------------------- BASELINE full-size properties --
def test_config2_full_size_linearity_and_oracle(gpu, oracle):
    """BASELINE config 2 (1M x 1M, 16/row): oracle comparison on the full matrix plus
    linearity A(ax + by) = aAx + bAy, a size-independent property."""
    n, k = 1_000_000, 16
    rp, ci, va = gpu.synth.uniform_csr(42, 0, n, n, k)
    x1, x2 = gpu.synth.vector(42, 1, n), gpu.synth.vector(42, 2, n)
    y1, _ = run_csr(gpu, rp, ci, va, n, x1, KERNELS["vector"])
    assert reorder_err(rp, ci, va, x1, oracle.spmv_csr(rp, ci, va, x1), y1) <= REORDER_TOL
    y2, _ = run_csr(gpu, rp, ci, va, n, x2, KERNELS["vector"])
    y3, _ = run_csr(gpu, rp, ci, va, n, (2.0 * x1 + 0.5 * x2).astype(np.float32), KERNELS["vector"])
    lin = 2.0 * y1.astype(np.float64) + 0.5 * y2.astype(np.float64)
    assert np.max(np.abs(lin - y3)) <= 1e-4 * max(1.0, np.max(np.abs(lin)))
    ym, _ = run_csr(gpu, rp, ci, va, n, x1, KERNELS["merge"])
    assert reorder_err(rp, ci, va, x1, y1, ym) <= 2 * REORDER_TOL


# ------------------------------------------------------------ LDS-tiled engine (use_texture) --
def run_tiled(spmv, rp, ci, va, cols, x, kernel=1, x_offset=0):
    A = spmv.csr_from_arrays(len(rp) - 1, cols, rp, ci, va)
    try:
        assert spmv.csr_to_gpu(A) == 0
        d_x = spmv.CudaBuffer(cols + 8)
        d_y = spmv.CudaBuffer(len(rp) - 1)
        shifted = np.concatenate([np.zeros(x_offset, np.float32), x])
        d_x.copyFromHost(shifted, shifted.size)
        cfg = spmv.SpMVConfig(kernel_type=kernel, use_texture=True)
        for _ in range(2):                                   # second call reuses the cached plan
            res = spmv.spmv_csr(A, d_x.get() + 4 * x_offset, d_y, cfg, cols)
            assert res.error_code == 0
        assert spmv.csr_has_tiled_plan(A), "expected the tiled engine to take this matrix"
        return d_y.copyToHost(len(rp) - 1)
    finally:
        spmv.csr_destroy(A)


@pytest.mark.parametrize("rows,cols,k", [(300_000, 400_000, 8), (131_073, 1_000_003, 9), (2_500_000, 270_000, 2),
                                         (150_000, 2_000_000, 16), (100_000, 3_000_000, 12),    # wide shards (16K / 32K strips)
                                         (200_000, 60_000_000, 12),         # 1832 strips: the builder's one-workgroup-per-CU LDS shape
                                         (100_000, 3072 * 32768, 12)])      # 3072 strips: the most the engine takes
def test_tiled_engine_uniform(gpu, oracle, rows, cols, k):
    """x through LDS strips (use_texture): shapes not multiples of the strip / tile sizes."""
    rp, ci, va = gpu.synth.uniform_csr(42, 0, rows, cols, k)
    x = gpu.synth.vector(42, 5, cols)
    want = oracle.spmv_csr(rp, ci, va, x)
    for kernel in (1, 2):
        got = run_tiled(gpu, rp, ci, va, cols, x, kernel)
        assert reorder_err(rp, ci, va, x, want, got) <= REORDER_TOL


def test_tiled_engine_power_law_empty_rows_and_unaligned_x(gpu, oracle):
    cols = 600_000
    lens = gpu.synth.power_law_lengths(42, 250_000, max_len=10000, n_cols=cols)
    lens[::5] = 0
    lens[-3:] = 0
    rp, ci, va = gpu.synth.stratified_csr(42, 0, lens, cols)
    x = gpu.synth.vector(42, 6, cols)
    want = oracle.spmv_csr(rp, ci, va, x)
    got = run_tiled(gpu, rp, ci, va, cols, x, 2, x_offset=1)     # x pointer only 4-byte aligned
    assert reorder_err(rp, ci, va, x, want, got) <= REORDER_TOL
    assert (got[lens == 0] == 0).all()


def test_tiled_engine_keeps_nan_and_inf_where_they_belong(gpu, oracle):
    """A NaN and an Inf in x reach exactly the rows that have an entry in those columns — the skip markers and the
    padding slots of the bucketed layout multiply 0 by x[first column of the strip] and must not carry the result
    into any row (here: the poisoned columns ARE strip starts)."""
    rows, cols = 200_000, 300_000
    rp, ci, va = gpu.synth.uniform_csr(9, 0, rows, cols, 6)
    x = gpu.synth.vector(9, 2, cols)
    x[0] = np.nan                 # column 0 = first column of strip 0: what every marker of that strip reads
    x[16384] = np.inf             # (a strip start for 4096-, 8192- and 16384-column strips)
    x[8192] = -np.inf
    want = oracle.spmv_csr(rp, ci, va, x)
    touched = np.zeros(rows, dtype=bool)
    row_of = np.repeat(np.arange(rows), np.diff(rp))
    touched[row_of[np.isin(ci, [0, 8192, 16384])]] = True
    assert 0 < touched.sum() < 100
    for kernel in (1, 2):
        got = run_tiled(gpu, rp, ci, va, cols, x, kernel)
        assert np.isfinite(got[~touched]).all()
        clean = ~touched
        # (6 entries per row, |a|, |x| < 1: the reordering bound 1e-5 * sum |a x| is at most 6e-5)
        np.testing.assert_allclose(got[clean], want[clean], rtol=1e-5, atol=6e-5)
        np.testing.assert_array_equal(np.isnan(got[touched]), np.isnan(want[touched]))
        both_inf = np.isinf(want[touched])
        np.testing.assert_array_equal(got[touched][both_inf], want[touched][both_inf])


def test_tiled_engine_nonnegative_relative_error(gpu, oracle):
    rp, ci, va = gpu.synth.uniform_csr(3, 0, 200_000, 500_000, 16)
    va = np.abs(va) + np.float32(0.01)
    x = np.abs(gpu.synth.vector(3, 3, 500_000)) + np.float32(0.01)
    got = run_tiled(gpu, rp, ci, va, 500_000, x)
    assert max_rel_err(oracle.spmv_csr(rp, ci, va, x), got) <= REORDER_TOL


def _tiled_with_info(spmv, rp, ci, va, cols, x):
    A = spmv.csr_from_arrays(len(rp) - 1, cols, rp, ci, va)
    try:
        assert spmv.csr_to_gpu(A) == 0
        d_x, d_y = spmv.CudaBuffer(cols), spmv.CudaBuffer(len(rp) - 1)
        d_x.copyFromHost(x, cols)
        res = spmv.spmv_csr(A, d_x, d_y, spmv.SpMVConfig(kernel_type=1, use_texture=True), cols)
        assert res.error_code == 0
        return d_y.copyToHost(len(rp) - 1), spmv.csr_tiled_info(A)
    finally:
        spmv.csr_destroy(A)


def test_tiled_engine_is_reproducible_run_to_run_and_across_plan_rebuilds(gpu):
    """SURVEY §5 / H4 (determinism).  The plan layout is a pure function of the matrix (no global atomics in
    the build), phase 2 accumulates every row in fp64 in LDS (a sum of fp32 products is then independent of
    the order in which the wavefronts' adds arrive, as long as the products of a row span < 29 binades), and
    the long rows' chunk sums are added in chunk order: y must come out bit for bit the same on every run and
    after a rebuild of the plan — on a uniform matrix and on a power-law one with long rows."""
    wl = importlib.import_module("gpu-spmv_amd.workloads")
    for A, kernel in ((wl.uniform_csr_device(5, 600_000, 700_000, 12), 1),
                      (wl.power_law_csr_device(5, 500_000, 500_000), 2)):
        x = wl.vector_device(5, 1, A.cols)
        y = gpu.CudaBuffer(A.rows)
        cfg = gpu.SpMVConfig(kernel, 256, True)
        outs = []
        for rebuild in (False, False, True, False):
            if rebuild:
                gpu.csr_invalidate_gpu_cache(A.handle)
            assert gpu.spmv_csr(A.handle, x, y, cfg, A.cols).error_code == 0
            assert gpu.csr_has_tiled_plan(A.handle)
            outs.append(y.copyToHost(A.rows).view(np.uint32).copy())
        if kernel == 2:
            assert gpu.csr_tiled_info(A.handle)["long_rows"] > 0
        for other in outs[1:]:
            assert np.array_equal(outs[0], other)
        x.release()
        y.release()
        A.close()


def test_rows_whose_products_span_more_than_fp64_can_hold_stay_within_one_ulp(gpu, oracle):
    """The limit of the reproducibility claim.  A tile adds fp32 products into fp64 accumulators
    in arrival order; that is order-free only while a row's products span < 29 binades (53 - 24 bits).  Rows built
    to break it: 2^40 and 2^16 in one strip (2^16 is exactly half an fp32 ulp of 2^40) and eight times 2^-14 in eight
    other strips (each below half an fp64 ulp of 2^40: absorbed one by one if they arrive after the big terms, 2^-11
    together if they meet first).  The exact sum rounds UP to 2^40 + 2^17, the reference's left-to-right fp32 sum
    gives 2^40; the engine may return either — one fp32 ulp apart, far inside the 1e-5 parity bound — and which one
    may change from run to run.  Every other row (narrow range) must still be bit-equal across runs."""
    rows, cols, special = 200_000, 262_144, 20_000
    base_rp, base_ci, base_va = gpu.synth.uniform_csr(17, 0, rows, cols, 6)
    strip = 16_384
    extra_cols = np.concatenate([[3, 7], strip * np.arange(2, 10) + 11]).astype(np.int32)        # ascending, ten strips
    extra_vals = np.concatenate([[2.0 ** 40, 2.0 ** 16], np.full(8, 2.0 ** -14)]).astype(np.float32)
    lens = np.full(rows, 6, dtype=np.int64)
    lens[:special] = extra_cols.size
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate([np.tile(extra_cols, special), base_ci[6 * special:]]).astype(np.int32)
    va = np.concatenate([np.tile(extra_vals, special), base_va[6 * special:]]).astype(np.float32)
    x = np.ones(cols, dtype=np.float32)
    want = oracle.spmv_csr(rp, ci, va, x)
    assert (want[:special] == np.float32(2.0 ** 40)).all()
    A = gpu.csr_from_arrays(rows, cols, rp, ci, va)
    assert gpu.csr_to_gpu(A) == 0
    d_x, d_y = gpu.CudaBuffer(cols), gpu.CudaBuffer(rows)
    d_x.copyFromHost(x, cols)
    allowed = {np.float32(2.0 ** 40), np.float32(2.0 ** 40 + 2.0 ** 17)}
    outs = []
    for _ in range(6):
        assert gpu.spmv_csr(A, d_x, d_y, gpu.SpMVConfig(1, 256, True), cols).error_code == 0
        got = d_y.copyToHost(rows)
        assert set(np.unique(got[:special]).tolist()) <= {float(v) for v in allowed}, np.unique(got[:special])
        assert reorder_err(rp, ci, va, x, want, got) <= 1e-5
        outs.append(got.view(np.uint32).copy())
    assert gpu.csr_has_tiled_plan(A)
    for other in outs[1:]:
        assert np.array_equal(outs[0][special:], other[special:])          # narrow rows: bit-equal, as claimed
    flips = int(sum(np.count_nonzero(outs[0][:special] != other[:special]) for other in outs[1:]))
    rounded_up = int(np.count_nonzero(outs[0][:special].view(np.float32) != np.float32(2.0 ** 40)))
    print("wide-range rows: %d of %d rounded up in run 0, %d row results changed between runs" % (rounded_up, special, flips))
    gpu.csr_destroy(A)


def test_both_placing_passes_build_the_same_plan(gpu, monkeypatch):
    """The builder's placing pass has two forms — slots assembled in LDS and stored as contiguous segments, or one
    scattered store per entry (what batches with long rows inside, or with more skip markers than the staging
    area holds, fall back to).  The layout is a pure function of the matrix, so both must produce the same bytes:
    equal checksums of the value / column / row-delta arrays and of the cell table, and bit-equal y — on a
    uniform matrix, a power-law one (long rows: mixed batches) and one whose rows are so far apart that almost
    every slot needs skip markers."""
    wl = importlib.import_module("gpu-spmv_amd.workloads")
    sparse_rows = 9_000_000
    lens = np.zeros(sparse_rows, dtype=np.int64)
    lens[::300] = 40                                   # 30000 rows with entries, 300 rows apart: every row change inside a cell needs a skip marker
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)

    def far_apart():
        A = wl.DeviceCSR(sparse_rows, 400_000, int(rp[-1]))
        A.row_ptrs.copyFromHost(rp.astype(np.int32), sparse_rows + 1)
        assert gpu.lib().spmv_c_gen_stratified_rows(5, 0, sparse_rows, 400_000, A.row_ptrs.get(), A.col_indices.get(),
                                                    A.values.get(), None) == 0
        gpu.device_synchronize()
        return A

    for make, kernel in ((lambda: wl.uniform_csr_device(6, 500_000, 800_000, 10), 1),
                         (lambda: wl.power_law_csr_device(6, 400_000, 600_000), 2),
                         (far_apart, 1)):
        A = make()
        assert gpu.tiled_shape(A.rows, A.cols, A.nnz)[0]
        x = wl.vector_device(6, 1, A.cols)
        y = gpu.CudaBuffer(A.rows)
        cfg = gpu.SpMVConfig(kernel, 256, True)
        seen = {}
        for form in ("staged", "scattered", "staged"):
            monkeypatch.setenv("SPMV_DEBUG", "place=" + form)
            gpu.csr_invalidate_gpu_cache(A.handle)
            assert gpu.spmv_csr(A.handle, x, y, cfg, A.cols).error_code == 0
            sums = gpu.csr_tiled_checksum(A.handle)
            assert sums is not None
            info = gpu.csr_tiled_info(A.handle)
            if make is far_apart:
                assert info["slots_in_cells"] > info["entries_in_cells"] * 1.2      # markers really are everywhere
            bits = y.copyToHost(A.rows).view(np.uint32).copy()
            if seen:
                assert sums == seen["sums"], (form, sums, seen["sums"])
                assert np.array_equal(bits, seen["bits"]), form
            else:
                seen = {"sums": sums, "bits": bits}
        monkeypatch.delenv("SPMV_DEBUG")
        x.release()
        y.release()
        A.close()


def test_stable_binning_and_ranking_by_comparison_build_the_same_plan(gpu, oracle, monkeypatch):
    """The builder's ranking pass sends every entry straight to its final place in its strip's bin (stable binning: the
    entries of a batch arrive in (row, column) order) and falls back to ranking by comparison when a bin holds more than
    255 entries or comes out unordered (rows whose columns are not ascending).  SPMV_DEBUG=rank=plain forces the
    comparison everywhere.  Same layout either way: equal checksums, bit-equal y — on a uniform matrix, a power-law one
    (long rows: mixed batches), one with few strips and fat bins (> 255 entries per bin: the fallback inside the default
    build), and one whose rows hold their columns in DESCENDING order (the order check must catch it)."""
    wl = importlib.import_module("gpu-spmv_amd.workloads")

    def descending_columns():
        rp, ci, va = gpu.synth.uniform_csr(9, 0, 300_000, 500_000, 9)
        ci = ci.reshape(-1, 9)[:, ::-1].copy().reshape(-1)              # every row back to front
        va = va.reshape(-1, 9)[:, ::-1].copy().reshape(-1)
        A = wl.DeviceCSR(300_000, 500_000, int(ci.size))
        A.row_ptrs.copyFromHost(rp.astype(np.int32), rp.size)
        A.col_indices.copyFromHost(ci.astype(np.int32), ci.size)
        A.values.copyFromHost(va.astype(np.float32), va.size)
        return A

    for name, make, kernel in (("uniform", lambda: wl.uniform_csr_device(6, 500_000, 800_000, 10), 1),
                               ("power_law", lambda: wl.power_law_csr_device(6, 400_000, 600_000), 2),
                               ("fat_bins", lambda: wl.uniform_csr_device(6, 400_000, 70_000, 24), 1),
                               ("descending", descending_columns, 1)):
        A = make()
        assert gpu.tiled_shape(A.rows, A.cols, A.nnz)[0], name
        x = wl.vector_device(6, 1, A.cols)
        y = gpu.CudaBuffer(A.rows)
        cfg = gpu.SpMVConfig(kernel, 256, True)
        seen = {}
        for form in ("stable", "plain", "stable"):
            monkeypatch.setenv("SPMV_DEBUG", "rank=" + form)
            gpu.csr_invalidate_gpu_cache(A.handle)
            assert gpu.spmv_csr(A.handle, x, y, cfg, A.cols).error_code == 0
            sums = gpu.csr_tiled_checksum(A.handle)
            assert sums is not None
            bits = y.copyToHost(A.rows).view(np.uint32).copy()
            if seen:
                assert sums == seen["sums"], (name, form, sums, seen["sums"])
                assert np.array_equal(bits, seen["bits"]), (name, form)
            else:
                seen = {"sums": sums, "bits": bits}
        monkeypatch.delenv("SPMV_DEBUG")
        if name == "descending":                # and the result is right, not just the same twice
            rp, ci, va = A.to_host()
            xh = x.copyToHost(A.cols)
            assert reorder_err(rp, ci, va, xh, oracle.spmv_csr(rp, ci, va, xh), y.copyToHost(A.rows)) <= 1e-5
        x.release()
        y.release()
        A.close()


def test_tiled_engine_folds_column_uniform_values(gpu, oracle, monkeypatch):
    """Every stored entry of a column equal (a_ij = 1 / outdeg(j), adjacency matrices): the plan keeps one
    weight per column and streams no values; one differing entry, or SPMV_TILED_FOLD=0, keeps the value
    stream.  All three agree with the oracle (the products are the same rounded numbers)."""
    rows = cols = 400_000
    lens = gpu.synth.power_law_lengths(9, rows, max_len=20000, n_cols=cols)       # long rows keep their CSR values
    rp, ci, _ = gpu.synth.stratified_csr(9, 0, lens, cols)
    ci = np.where(ci == 70_000, 70_001, ci).astype(np.int32)                      # column 70000 has no entry at all
    outdeg = np.bincount(ci, minlength=cols)
    va = (np.float32(1.0) / np.maximum(outdeg, 1).astype(np.float32))[ci]
    x = np.abs(gpu.synth.vector(9, 1, cols)) + np.float32(0.01)
    x[70_000] = np.inf                                                            # 0 * inf must not leak out of the unused column
    want = oracle.spmv_csr(rp, ci, va, x)
    assert np.isfinite(want).all()

    got, info = _tiled_with_info(gpu, rp, ci, va, cols, x)
    assert info["values_folded"] and info["long_rows"] > 0
    assert max_rel_err(want, got) <= REORDER_TOL

    monkeypatch.setenv("SPMV_TILED_FOLD", "0")
    got_plain, info = _tiled_with_info(gpu, rp, ci, va, cols, x)
    assert not info["values_folded"]
    assert max_rel_err(want, got_plain) <= REORDER_TOL
    monkeypatch.delenv("SPMV_TILED_FOLD")

    vb = va.copy()
    vb[len(vb) // 2] = np.nextafter(vb[len(vb) // 2], np.float32(2.0))           # one entry one ulp off
    got_b, info = _tiled_with_info(gpu, rp, ci, vb, cols, x)
    assert not info["values_folded"]
    assert max_rel_err(oracle.spmv_csr(rp, ci, vb, x), got_b) <= REORDER_TOL

    ones = np.ones_like(va)                                                       # plain adjacency matrix
    got_1, info = _tiled_with_info(gpu, rp, ci, ones, cols, x)
    assert info["values_folded"]
    assert max_rel_err(oracle.spmv_csr(rp, ci, ones, x), got_1) <= REORDER_TOL


def test_values_rewritten_in_place_need_a_cache_invalidation(gpu, oracle):
    """The tiled plan keeps its own copy of the entries: after the device values are overwritten in place
    the caller drops the cached data (csr_invalidate_gpu_cache) and the next call sees the new values."""
    rows, cols, k = 200_000, 300_000, 8
    rp, ci, va = gpu.synth.uniform_csr(5, 0, rows, cols, k)
    x = gpu.synth.vector(5, 1, cols)
    A = gpu.csr_from_arrays(rows, cols, rp, ci, va)
    assert gpu.csr_to_gpu(A) == 0
    d_x, d_y = gpu.CudaBuffer(cols), gpu.CudaBuffer(rows)
    d_x.copyFromHost(x, cols)
    cfg = gpu.SpMVConfig(kernel_type=1, use_texture=True)
    assert gpu.spmv_csr(A, d_x, d_y, cfg, cols).error_code == 0 and gpu.csr_has_tiled_plan(A)
    assert reorder_err(rp, ci, va, x, oracle.spmv_csr(rp, ci, va, x), d_y.copyToHost(rows)) <= REORDER_TOL
    doubled = (va * np.float32(2.0)).astype(np.float32)
    assert gpu.lib().spmv_c_memcpy_h2d(A.contents.d_values, doubled.ctypes.data, doubled.nbytes) == 0
    gpu.csr_invalidate_gpu_cache(A)
    assert not gpu.csr_has_tiled_plan(A)
    assert gpu.spmv_csr(A, d_x, d_y, cfg, cols).error_code == 0
    assert reorder_err(rp, ci, doubled, x, oracle.spmv_csr(rp, ci, doubled, x), d_y.copyToHost(rows)) <= REORDER_TOL
    gpu.csr_destroy(A)


def test_use_texture_on_small_matrix_keeps_direct_kernels(gpu, oracle):
    """Below the size where x leaves L2 the hint is ignored (no plan is built)."""
    rp, ci, va = gpu.synth.uniform_csr(1, 0, 5000, 20000, 8)
    x = gpu.synth.vector(1, 1, 20000)
    A = gpu.csr_from_arrays(5000, 20000, rp, ci, va)
    gpu.csr_to_gpu(A)
    d_x, d_y = gpu.CudaBuffer(20000), gpu.CudaBuffer(5000)
    d_x.copyFromHost(x, 20000)
    res = gpu.spmv_csr(A, d_x, d_y, gpu.SpMVConfig(kernel_type=1, use_texture=True), 20000)
    assert res.error_code == 0 and not gpu.csr_has_tiled_plan(A)
    assert reorder_err(rp, ci, va, x, oracle.spmv_csr(rp, ci, va, x), d_y.copyToHost(5000)) <= REORDER_TOL
    # SCALAR_CSR keeps its CPU-order contract even with the hint
    res = gpu.spmv_csr(A, d_x, d_y, gpu.SpMVConfig(kernel_type=0, use_texture=True), 20000)
    np.testing.assert_array_equal(d_y.copyToHost(5000), oracle.spmv_csr(rp, ci, va, x))
    gpu.csr_destroy(A)


def test_ell_through_the_tiled_engine(gpu, oracle):
    """spmv_ell with use_texture: the plan is built from the column-major slabs (padding skipped)."""
    import ctypes
    rows, cols, k = 300_000, 400_000, 12
    rng = np.random.default_rng(5)
    lens = rng.integers(0, k + 1, size=rows)
    lens[7] = k
    rp, ci, va = gpu.synth.stratified_csr(8, 0, lens, cols)
    kk, ecols, evals = oracle.ell_from_csr(rp, ci, va)
    x = gpu.synth.vector(8, 1, cols)
    want = oracle.spmv_ell(rows, kk, ecols, evals, x)
    E = gpu.ell_create(rows, cols, kk)
    ctypes.memmove(E.contents.col_indices, ecols.ctypes.data, ecols.nbytes)
    ctypes.memmove(E.contents.values, evals.ctypes.data, evals.nbytes)
    assert gpu.ell_to_gpu(E) == 0
    d_x, d_y = gpu.CudaBuffer(cols), gpu.CudaBuffer(rows)
    d_x.copyFromHost(x, cols)
    cfg = gpu.SpMVConfig(kernel_type=gpu.SpMVConfig.ELL_KERNEL, use_texture=True)
    for _ in range(2):
        res = gpu.spmv_ell(E, d_x, d_y, cfg, cols)
        assert res.error_code == 0
    assert reorder_err(rp, ci, va, x, want, d_y.copyToHost(rows)) <= REORDER_TOL
    res = gpu.spmv_ell(E, d_x, d_y, None, cols)              # default config: CPU order, bit-exact
    np.testing.assert_array_equal(d_y.copyToHost(rows), want)
    gpu.ell_destroy(E)


def test_ell_with_column_uniform_values_through_the_tiled_engine(gpu, oracle):
    """An ELL adjacency matrix (all stored values 1/outdeg of their column): the plan built from the slabs
    folds the values (padding slots are skipped by the probe) and still matches the oracle."""
    import ctypes
    rows, cols, k = 300_000, 400_000, 10
    rng = np.random.default_rng(11)
    lens = rng.integers(1, k + 1, size=rows)
    rp, ci, _ = gpu.synth.stratified_csr(12, 0, lens, cols)
    outdeg = np.bincount(ci, minlength=cols)
    va = (np.float32(1.0) / np.maximum(outdeg, 1).astype(np.float32))[ci]
    kk, ecols, evals = oracle.ell_from_csr(rp, ci, va)
    x = np.abs(gpu.synth.vector(12, 1, cols)) + np.float32(0.01)
    want = oracle.spmv_ell(rows, kk, ecols, evals, x)
    E = gpu.ell_create(rows, cols, kk)
    ctypes.memmove(E.contents.col_indices, ecols.ctypes.data, ecols.nbytes)
    ctypes.memmove(E.contents.values, evals.ctypes.data, evals.nbytes)
    assert gpu.ell_to_gpu(E) == 0
    d_x, d_y = gpu.CudaBuffer(cols), gpu.CudaBuffer(rows)
    d_x.copyFromHost(x, cols)
    cfg = gpu.SpMVConfig(kernel_type=gpu.SpMVConfig.ELL_KERNEL, use_texture=True)
    for _ in range(2):
        assert gpu.spmv_ell(E, d_x, d_y, cfg, cols).error_code == 0
    assert max_rel_err(want, d_y.copyToHost(rows)) <= REORDER_TOL
    gpu.ell_destroy(E)


# ------------------------------------------------ BASELINE configs 3, 4, 5 at full size --------
def _device_inputs(gpu, A, cols, tag):
    wl = __import__("importlib").import_module("gpu-spmv_amd.workloads")
    x = wl.vector_device(42, tag, cols)
    return x, x.copyToHost(cols)


def test_config5_full_size_oracle_parity_and_linearity(gpu, oracle):
    """BASELINE config 5's matrix (10 M x 10 M, 160 M entries, built in HBM): the LDS-tiled engine
    and the direct vector kernel against the CPU oracle on the full matrix, plus linearity."""
    import importlib
    wl = importlib.import_module("gpu-spmv_amd.workloads")
    n, k = 10_000_000, 16
    A = wl.uniform_csr_device(42, n, n, k)
    rp, ci, va = A.to_host()
    assert (np.diff(rp) == k).all() and ci.min() >= 0 and ci.max() < n
    x1d, x1 = _device_inputs(gpu, A, n, 1)
    x2d, x2 = _device_inputs(gpu, A, n, 2)
    want = oracle.spmv_csr(rp, ci, va, x1)
    y = gpu.CudaBuffer(n)
    got = {}
    for name, cfg in (("tiled", gpu.SpMVConfig(1, 256, True)), ("vector", gpu.SpMVConfig(1, 256, False))):
        res = gpu.spmv_csr(A.handle, x1d, y, cfg, n)
        assert res.error_code == 0
        got[name] = y.copyToHost(n)
        assert reorder_err(rp, ci, va, x1, want, got[name]) <= REORDER_TOL, name
    assert gpu.csr_has_tiled_plan(A.handle)
    # linearity through the tiled engine: A(2 x1 + 0.5 x2) = 2 A x1 + 0.5 A x2
    res = gpu.spmv_csr(A.handle, x2d, y, gpu.SpMVConfig(1, 256, True), n)
    y2 = y.copyToHost(n)
    x3d = gpu.CudaBuffer(n)
    x3d.copyFromHost((2.0 * x1 + 0.5 * x2).astype(np.float32), n)
    res = gpu.spmv_csr(A.handle, x3d, y, gpu.SpMVConfig(1, 256, True), n)
    lin = 2.0 * got["tiled"].astype(np.float64) + 0.5 * y2.astype(np.float64)
    assert np.max(np.abs(lin - y.copyToHost(n))) <= 1e-4 * max(1.0, float(np.max(np.abs(lin))))
    A.close()


def test_config4_full_size_power_law(gpu, oracle):
    """BASELINE config 4 (1 M rows, Pareto lengths, one row of 10 000): selector picks MERGE_PATH;
    merge-path, the tiled engine and vector-CSR all match the oracle."""
    import importlib
    wl = importlib.import_module("gpu-spmv_amd.workloads")
    n = 1_000_000
    A = wl.power_law_csr_device(42, n, n)
    rp, ci, va = A.to_host()
    cfg = gpu.spmv_auto_config(A.handle)
    assert cfg.kernel_type == gpu.SpMVConfig.MERGE_PATH and cfg.use_texture == 1
    st = gpu.csr_compute_stats(A.handle)
    assert st.max_nnz_per_row == 10000 and st.skewness >= 10
    xd, x = _device_inputs(gpu, A, n, 4)
    want = oracle.spmv_csr(rp, ci, va, x)
    y = gpu.CudaBuffer(n)
    for c in (cfg, gpu.SpMVConfig(2, 256, False), gpu.SpMVConfig(1, 256, False)):
        assert gpu.spmv_csr(A.handle, xd, y, c, n).error_code == 0
        assert reorder_err(rp, ci, va, x, want, y.copyToHost(n)) <= REORDER_TOL
    A.close()


def test_config3_full_size_ell(gpu, oracle):
    """BASELINE config 3 (ELL 1 M x 1 M, K = 32, no padding): default path bit-exact against the
    oracle, use_texture path within the reorder tolerance."""
    import importlib
    wl = importlib.import_module("gpu-spmv_amd.workloads")
    n, k = 1_000_000, 32
    E = wl.uniform_ell_device(42, n, n, k)
    ecols = E.col_indices.copyToHost(n * k)
    evals = E.values.copyToHost(n * k)
    rp, ci, va = gpu.synth.uniform_csr(42, 0, 4096, n, k)          # the generator's CSR twin, first rows
    np.testing.assert_array_equal(ecols.reshape(k, n)[:, :4096].T.reshape(-1), ci)
    xd, x = _device_inputs(gpu, E, n, 3)
    want = oracle.spmv_ell(n, k, ecols, evals, x)
    y = gpu.CudaBuffer(n)
    assert gpu.spmv_ell(E.handle, xd, y, None, n).error_code == 0
    np.testing.assert_array_equal(y.copyToHost(n), want)
    cfg = gpu.SpMVConfig(gpu.SpMVConfig.ELL_KERNEL, 256, True)
    assert gpu.spmv_ell(E.handle, xd, y, cfg, n).error_code == 0
    got = y.copyToHost(n)
    scale = np.abs(evals.reshape(k, n).astype(np.float64) * x[ecols.reshape(k, n)]).sum(axis=0)
    assert np.max(np.abs(got - want) / np.maximum(np.maximum(np.abs(want), scale), 1e-30)) <= REORDER_TOL
    E.close()


@pytest.mark.parametrize("rows,cols,k", [(200_000, 30_000, 12), (65_536, 32_768, 40), (50_000, 1_003, 7)])
def test_vector_kernel_with_x_resident_in_lds(gpu, oracle, rows, cols, k):
    """use_texture on a matrix whose x fits one CU's LDS (<= 32 K columns): the vector kernel
    gathers from an LDS copy of x; also with an x pointer that is only 4-byte aligned."""
    rp, ci, va = gpu.synth.uniform_csr(9, 0, rows, cols, k)
    x = gpu.synth.vector(9, 2, cols)
    want = oracle.spmv_csr(rp, ci, va, x)
    A = gpu.csr_from_arrays(rows, cols, rp, ci, va)
    gpu.csr_to_gpu(A)
    d_x, d_y = gpu.CudaBuffer(cols + 4), gpu.CudaBuffer(rows)
    cfg = gpu.SpMVConfig(kernel_type=1, use_texture=True)
    for offset in (0, 1):
        d_x.copyFromHost(np.concatenate([np.zeros(offset, np.float32), x]), cols + offset)
        assert gpu.spmv_csr(A, d_x.get() + 4 * offset, d_y, cfg, cols).error_code == 0
        assert reorder_err(rp, ci, va, x, want, d_y.copyToHost(rows)) <= REORDER_TOL
    assert not gpu.csr_has_tiled_plan(A)
    gpu.csr_destroy(A)


def _csr_from_lengths_and_cols(lens, col_fn, rng):
    rp = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    row_of = np.repeat(np.arange(len(lens), dtype=np.int64), lens)
    slot = np.arange(rp[-1], dtype=np.int64) - rp[row_of]
    ci = col_fn(row_of, slot).astype(np.int32)
    va = rng.uniform(-1, 1, ci.size).astype(np.float32)
    return rp.astype(np.int32), ci, va


@pytest.mark.parametrize("shape", ["banded", "banded_dense", "one_strip", "few_long_rows", "duplicate_columns", "dense_column",
                                   "unsorted_columns", "explicit_zeros"])
def test_tiled_engine_adversarial_structure(gpu, oracle, shape):
    """Structures that stress the cells of the tiled engine: every entry of a row in one strip
    (banded; banded_dense: 48 per row, so that ONE wavefront's run holds far more than the 64 passes whose descriptors it
    fetches at a time), every entry of the matrix in one strip, rows far longer than the long-row limit,
    repeated (row, column) pairs, one column referenced by every row, rows whose columns are stored in descending
    order (legal CSR; the reference never sorts), stored zeros (values and a whole column of them)."""
    rng = np.random.default_rng(17)
    if shape == "banded":
        rows = cols = 300_000
        lens = np.full(rows, 16)
        rp, ci, va = _csr_from_lengths_and_cols(lens, lambda r, s: np.clip(r - 8 + s, 0, cols - 1), rng)
    elif shape == "banded_dense":
        rows = cols = 200_000
        lens = np.full(rows, 48)
        rp, ci, va = _csr_from_lengths_and_cols(lens, lambda r, s: np.clip(r - 24 + s, 0, cols - 1), rng)
    elif shape == "one_strip":
        rows, cols = 400_000, 200_000
        lens = np.full(rows, 8)
        rp, ci, va = _csr_from_lengths_and_cols(lens, lambda r, s: (r * 7 + s * 13) % 3000, rng)
    elif shape == "few_long_rows":
        rows, cols = 3_000, 1_000_000
        lens = np.full(rows, 700)
        lens[::97] = 0
        rp, ci, va = _csr_from_lengths_and_cols(lens, lambda r, s: (s * 1427 + r * 31) % cols, rng)
    elif shape == "duplicate_columns":
        rows, cols = 200_000, 150_000
        lens = np.full(rows, 10)
        rp, ci, va = _csr_from_lengths_and_cols(lens, lambda r, s: (r * 3 + (s // 2) * 50_021) % cols, rng)
    elif shape == "unsorted_columns":
        rows, cols = 250_000, 400_000
        lens = np.full(rows, 12)
        rp, ci, va = _csr_from_lengths_and_cols(lens, lambda r, s: (r * 11 + (11 - s) * 30_011) % cols, rng)
    elif shape == "explicit_zeros":
        rows, cols = 250_000, 400_000
        lens = np.full(rows, 8)
        rp, ci, va = _csr_from_lengths_and_cols(lens, lambda r, s: np.where(s == 3, 123_456, (r * 13 + s * 40_009) % cols), rng)
        va[::3] = 0.0
        va[ci == 123_456] = 0.0                       # a column that holds nothing but stored zeros
    else:
        rows, cols = 500_000, 100_000
        lens = np.full(rows, 4)
        rp, ci, va = _csr_from_lengths_and_cols(lens, lambda r, s: np.where(s == 0, 77, (r * 5 + s * 33_331) % cols), rng)
    x = gpu.synth.vector(17, 1, cols)
    want = oracle.spmv_csr(rp, ci, va, x)
    got = run_tiled(gpu, rp, ci, va, cols, x, kernel=2)
    assert reorder_err(rp, ci, va, x, want, got) <= REORDER_TOL


@pytest.mark.gpu
def test_callers_that_spell_the_kernel_are_promoted_to_the_tiled_engine(gpu, oracle):
    """The reference's own callers pass {VECTOR_CSR, 256, false} (benchmarks/main.cu:52-56, src/pagerank.cu:89-90).  On a
    matrix the tiled engine would take, the first `spmv_get_tiled_promotion()` calls run the direct kernel, the next one
    builds the plan (in front of its timed region) and every later VECTOR_CSR / MERGE_PATH call runs on the engine; all
    of them match the oracle.  SCALAR_CSR never promotes and stays bit-exact; invalidating the matrix's cache starts the
    count again; promotion 0 keeps even a cached plan out of a call that did not ask for it."""
    import importlib
    wl = importlib.import_module("gpu-spmv_amd.workloads")
    n, k = 1_000_000, 16
    A = wl.uniform_csr_device(7, n, n, k)
    rp, ci, va = A.to_host()
    xd, x = _device_inputs(gpu, A, n, 3)
    want = oracle.spmv_csr(rp, ci, va, x)
    y = gpu.CudaBuffer(n)
    vector, merge, scalar = gpu.SpMVConfig(1, 256, False), gpu.SpMVConfig(2, 256, False), gpu.SpMVConfig(0, 256, False)
    gpu.set_tiled_promotion(4)
    try:
        assert gpu.get_tiled_promotion() == 4
        outputs = []
        for call in range(8):
            res = gpu.spmv_csr(A.handle, xd, y, vector, n)
            assert res.error_code == 0
            got = y.copyToHost(n)
            assert reorder_err(rp, ci, va, x, want, got) <= REORDER_TOL, call
            outputs.append((bool(gpu.csr_has_tiled_plan(A.handle)), got, res.elapsed_ms))
        assert [o[0] for o in outputs] == [False] * 4 + [True] * 4
        # the direct calls agree with each other bit for bit, and so do the promoted ones (same kernel, same inputs)
        assert all(np.array_equal(outputs[0][1], o[1]) for o in outputs[1:4])
        assert all(np.array_equal(outputs[4][1], o[1]) for o in outputs[5:])
        # the call that built the plan did not time the build (about 1 ms on this matrix; the kernel takes ~50 us)
        assert outputs[4][2] < 0.5 * outputs[0][2] + 0.05, [o[2] for o in outputs]
        info = gpu.csr_tiled_info(A.handle)
        assert info and info["plan_bytes"] > 0
        # MERGE_PATH callers ride on the same plan; SCALAR_CSR keeps its kernel and its bits
        assert gpu.spmv_csr(A.handle, xd, y, merge, n).error_code == 0
        assert np.array_equal(y.copyToHost(n), outputs[4][1])
        assert gpu.spmv_csr(A.handle, xd, y, scalar, n).error_code == 0
        assert np.array_equal(y.copyToHost(n).view(np.uint32), want.view(np.uint32))
        # promotion off: the cached plan is not used by a caller that did not ask for it
        gpu.set_tiled_promotion(0)
        assert gpu.spmv_csr(A.handle, xd, y, vector, n).error_code == 0
        assert np.array_equal(y.copyToHost(n), outputs[0][1])
        gpu.set_tiled_promotion(4)
        # in-place rewrite of the device arrays: the cache goes, and with it the count
        gpu.csr_invalidate_gpu_cache(A.handle)
        assert not gpu.csr_has_tiled_plan(A.handle)
        for call in range(4):
            assert gpu.spmv_csr(A.handle, xd, y, vector, n).error_code == 0
            assert not gpu.csr_has_tiled_plan(A.handle)
        assert gpu.spmv_csr(A.handle, xd, y, vector, n).error_code == 0 and gpu.csr_has_tiled_plan(A.handle)
        # two matrices taking turns over ONE row-pointer array (the side table's key; every k-per-row graph has the same
        # one): the plan is replaced a few times, then promotion gives up on the key instead of rebuilding on every call
        B = wl.uniform_csr_device(8, n, n, k)
        shared = gpu.csr_wrap_device(n, n, n * k, A.row_ptrs.get(), B.col_indices.get(), B.values.get())
        rpB, ciB, vaB = B.to_host()
        wantB = oracle.spmv_csr(rpB, ciB, vaB, x)
        builds = 0
        for turn in range(12):
            handle, ref = (A.handle, (rp, ci, va, want)) if turn % 2 == 0 else (shared, (rpB, ciB, vaB, wantB))
            before = gpu.csr_tiled_info(handle)
            assert gpu.spmv_csr(handle, xd, y, vector, n).error_code == 0
            assert reorder_err(ref[0], ref[1], ref[2], x, ref[3], y.copyToHost(n)) <= REORDER_TOL, turn
            after = gpu.csr_tiled_info(handle)
            builds += 1 if after and (not before or after["build_ms"] != before["build_ms"]) else 0
        assert builds <= 4, builds
        gpu.csr_destroy(shared)
        B.close()
        # a small matrix (below the engine's thresholds) is never promoted
        S = wl.uniform_csr_device(7, 20_000, 20_000, 8)
        xs = wl.vector_device(7, 1, 20_000)
        ys = gpu.CudaBuffer(20_000)
        for call in range(8):
            assert gpu.spmv_csr(S.handle, xs, ys, vector, 20_000).error_code == 0
        assert not gpu.csr_has_tiled_plan(S.handle)
        S.close(); xs.release(); ys.release()
    finally:
        gpu.set_tiled_promotion(0)
        A.close()
