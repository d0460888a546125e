"""Pins the CPU oracle (oracle/spmv_oracle.c) to the reference:
  * tests/golden/ref_cases.npz — outputs of the reference's own CPU sources
    (compiled from /root/reference by oracle/Makefile; see tests/golden/make_golden.py);
  * known-answer vectors the reference's README / design doc / tests hold.
Everything here is bit-exact: the oracle follows the reference's operation order."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR


@pytest.fixture(scope="module")
def golden():
    data = np.load(os.path.join(GOLDEN_DIR, "ref_cases.npz"), allow_pickle=False)
    return data, [str(n) for n in data["case_names"]]


def test_golden_has_all_cases(golden):
    data, names = golden
    assert len(names) == 13
    for n in names:
        assert f"{n}/y_csr" in data.files and f"{n}/csr_row_ptrs" in data.files


def test_csr_from_dense_bit_exact(oracle, golden):
    """reference src/csr_matrix.cpp:50-95"""
    data, names = golden
    for n in names:
        rp, ci, va = oracle.csr_from_dense(data[f"{n}/dense"])
        np.testing.assert_array_equal(rp, data[f"{n}/csr_row_ptrs"], err_msg=n)
        np.testing.assert_array_equal(ci, data[f"{n}/csr_col_indices"], err_msg=n)
        np.testing.assert_array_equal(va.view(np.uint32), data[f"{n}/csr_values"].view(np.uint32), err_msg=n)


def test_spmv_csr_bit_exact(oracle, golden):
    """reference src/spmv_cpu.cpp:6-16"""
    data, names = golden
    for n in names:
        y = oracle.spmv_csr(data[f"{n}/csr_row_ptrs"], data[f"{n}/csr_col_indices"], data[f"{n}/csr_values"],
                            data[f"{n}/x"])
        np.testing.assert_array_equal(y.view(np.uint32), data[f"{n}/y_csr"].view(np.uint32), err_msg=n)


def test_ell_from_csr_and_spmv_ell_bit_exact(oracle, golden):
    """reference src/ell_matrix.cpp:111-159 and src/spmv_cpu.cpp:18-32"""
    data, names = golden
    for n in names:
        k, ecols, evals = oracle.ell_from_csr(data[f"{n}/csr_row_ptrs"], data[f"{n}/csr_col_indices"],
                                              data[f"{n}/csr_values"])
        rows, cols, kk = data[f"{n}/ell_shape"]
        assert k == kk, n
        np.testing.assert_array_equal(ecols, data[f"{n}/ell_col_indices"], err_msg=n)
        np.testing.assert_array_equal(evals.view(np.uint32), data[f"{n}/ell_values"].view(np.uint32), err_msg=n)
        # ell_from_dense gives the same slabs (src/ell_matrix.cpp:53-109)
        np.testing.assert_array_equal(ecols, data[f"{n}/ell_dense_col_indices"], err_msg=n)
        y = oracle.spmv_ell(int(rows), int(k), ecols, evals, data[f"{n}/x"])
        np.testing.assert_array_equal(y.view(np.uint32), data[f"{n}/y_ell"].view(np.uint32), err_msg=n)


def test_stats_and_selector(oracle, golden):
    """reference src/csr_matrix.cpp:281-300 and src/spmv_cpu.cpp:34-50"""
    data, names = golden
    seen = set()
    for n in names:
        rp = data[f"{n}/csr_row_ptrs"]
        avg, mx, mn, skew = oracle.csr_stats(rp)
        ref = data[f"{n}/csr_stats"]
        assert np.float32(avg) == ref[0] and mx == int(ref[1]) and mn == int(ref[2]) and np.float32(skew) == ref[3], n
        kt, tex = oracle.auto_config(rp, int(data[f"{n}/csr_shape"][1]))
        assert [kt, 256, int(tex)] == list(data[f"{n}/auto_config"]), n
        seen.add(kt)
    assert seen == {0, 1, 2}      # the fixture exercises all three selector outcomes


def test_known_answers(oracle):
    """README.md:75-99 ({3,7,5}, row_ptrs {0,2,4,5}); design.md:372-385 (CSR/ELL layout);
    tests/test_spmv.cu:161-186 (10.0) and :188-218 ({3,0,7})."""
    rp, ci, va = oracle.csr_from_dense(np.array([[1, 0, 2], [0, 3, 4], [5, 0, 0]], np.float32))
    assert list(rp) == [0, 2, 4, 5]
    assert list(oracle.spmv_csr(rp, ci, va, np.ones(3, np.float32))) == [3.0, 7.0, 5.0]

    rp, ci, va = oracle.csr_from_dense(np.array([[1, 0, 2, 0], [0, 3, 4, 0], [0, 0, 0, 5]], np.float32))
    assert list(va) == [1, 2, 3, 4, 5] and list(ci) == [0, 2, 1, 2, 3] and list(rp) == [0, 2, 4, 5]
    k, ecols, evals = oracle.ell_from_csr(rp, ci, va)
    assert k == 2 and list(evals) == [1, 3, 5, 2, 4, 0] and list(ecols) == [0, 1, 3, 2, 2, -1]

    rp, ci, va = oracle.csr_from_dense(np.array([[5.0]], np.float32))
    assert list(oracle.spmv_csr(rp, ci, va, np.array([2.0], np.float32))) == [10.0]
    rp, ci, va = oracle.csr_from_dense(np.array([[1, 2, 0], [0, 0, 0], [3, 0, 4]], np.float32))
    assert list(oracle.spmv_csr(rp, ci, va, np.ones(3, np.float32))) == [3.0, 0.0, 7.0]


def test_byte_model(oracle):
    """reference src/bandwidth.cpp:34-42, :66-75; BASELINE.md §3 figures."""
    assert oracle.bytes_csr(1_000_000, 1_000_000, 16_000_000) == pytest.approx(140.0e6, rel=1e-4)
    assert oracle.bytes_ell(1_000_000, 1_000_000, 32) == 264.0e6
    assert oracle.bytes_csr(10_000_000, 10_000_000, 160_000_000) == pytest.approx(1.4e9, rel=1e-4)


def test_pagerank_reference_test_invariants(oracle):
    """reference tests/test_pagerank.cu:140-164: a 3-cycle converges to equal ranks (1e-4);
    :18-77: ranks >= 0, sum 1 (1e-4), converged => residual < tol."""
    dense = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], np.float32)
    rp, ci, va = oracle.csr_from_dense(dense)
    ranks, iters, res, conv = oracle.pagerank(rp, ci, va)
    assert conv and np.allclose(ranks, 1 / 3, atol=1e-4) and res < 1e-6

    rng = np.random.default_rng(42)
    for _ in range(30):
        n = int(rng.integers(5, 50))
        adj = (rng.random((n, n)) < 0.2).astype(np.float32)
        col = adj.sum(axis=0)
        adj = np.where(col > 0, adj / np.maximum(col, 1), 0).astype(np.float32)
        rp, ci, va = oracle.csr_from_dense(adj) if adj.any() else (np.zeros(n + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
        for wide in (False, True):
            ranks, iters, res, conv = oracle.pagerank(rp, ci, va, num_cols=n, wide_sums=wide)
            assert (ranks >= 0).all() and abs(ranks.sum() - 1.0) < 1e-4
            assert (not conv) or res < 1e-6


def test_oracle_matches_compiled_reference_live(oracle):
    """When oracle/_ref/ref_cpu is present (build container, or shipped prebuilt to the
    GPU box), re-check on fresh random inputs; skipped where the binary is absent."""
    if not oracle.have_reference_binary():
        pytest.skip("oracle/_ref/ref_cpu not built here")
    rng = np.random.default_rng(7)
    for _ in range(5):
        rows, cols = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        dense = np.where(rng.random((rows, cols)) < 0.1, rng.uniform(-10, 10, (rows, cols)), 0).astype(np.float32)
        x = rng.uniform(-10, 10, cols).astype(np.float32)
        ref = oracle.reference_case(dense, x)
        rp, ci, va = oracle.csr_from_dense(dense)
        np.testing.assert_array_equal(rp, ref["csr_row_ptrs"])
        np.testing.assert_array_equal(oracle.spmv_csr(rp, ci, va, x).view(np.uint32), ref["y_csr"].view(np.uint32))


def test_parallel_cpu_baseline_is_bit_identical(oracle, golden):
    """The OpenMP row-block loop used for the all-cores CPU baseline computes the oracle's y exactly."""
    data, names = golden
    for n in names:
        args = (data[f"{n}/csr_row_ptrs"], data[f"{n}/csr_col_indices"], data[f"{n}/csr_values"], data[f"{n}/x"])
        np.testing.assert_array_equal(oracle.spmv_csr_parallel(*args, threads=3).view(np.uint32),
                                      oracle.spmv_csr(*args).view(np.uint32), err_msg=n)
