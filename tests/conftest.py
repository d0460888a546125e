"""pytest configuration: markers, import path, shared fixtures."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def spmv():
    """The Python host mirror of the library (binds the C ABI of include/spmv_c.h)."""
    return importlib.import_module("gpu-spmv_amd")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/spmv_oracle.c) — the checker, never the thing under test."""
    mod = importlib.import_module("oracle")
    mod.lib()
    return mod


@pytest.fixture(scope="session")
def gpu(spmv):
    """Fails loudly (no silent CPU route) when a gpu-marked test runs without a device.
    The parity tests name the kernel they test: promotion of VECTOR_CSR / MERGE_PATH callers to the tiled engine
    (spmv_set_tiled_promotion) is off for the session; tests/test_gpu_spmv.py's promotion test turns it on itself."""
    spmv.require_gpu()
    spmv.set_tiled_promotion(0)
    return spmv


def max_rel_err(expected, actual, floor=1e-6):
    """Largest per-element error with the reference's comparator shape
    (tests/test_spmv.cu:18-35): relative to max(|a|,|b|), absolute below `floor`."""
    expected = np.asarray(expected, dtype=np.float64)
    actual = np.asarray(actual, dtype=np.float64)
    diff = np.abs(expected - actual)
    scale = np.maximum(np.abs(expected), np.abs(actual))
    rel = np.where(scale < 1e-10, np.where(diff > floor, np.inf, 0.0), diff / np.maximum(scale, 1e-300))
    return float(rel.max()) if rel.size else 0.0


def random_dense(rng, rows, cols, density, lo=-10.0, hi=10.0):
    mask = rng.random((rows, cols)) < density
    vals = rng.uniform(lo, hi, size=(rows, cols)).astype(np.float32)
    vals[vals == 0.0] = 1.0
    return np.where(mask, vals, np.float32(0.0)).astype(np.float32)


def reorder_err(row_ptrs, cols, vals, x, expected, actual):
    """Error measure for kernels that reorder a row's sum (VECTOR_CSR, MERGE_PATH, tiled):
    max over rows of |got - want| / max(|want|, sum_j |a_ij x_j|).  Bounding the error by
    the row's absolute sum is the standard backward-error scale for a reordered fp32 sum;
    a row whose terms cancel (|y_i| << sum |a x|) cannot be held to 1e-5 of |y_i| by ANY
    summation order other than the oracle's own.  For rows without cancellation this is
    exactly the relative error of the north-star (1e-5)."""
    row_ptrs = np.asarray(row_ptrs, dtype=np.int64)
    prod = np.abs(np.asarray(vals, np.float64) * np.asarray(x, np.float64)[np.asarray(cols)])
    # per-row sums row by row (differences of one running sum over the whole matrix lose the small rows as soon as
    # some row holds huge products: test_rows_whose_products_span_more_than_fp64_can_hold...)
    abs_sum = np.zeros(row_ptrs.size - 1, dtype=np.float64)
    nonempty = row_ptrs[1:] > row_ptrs[:-1]
    if prod.size and nonempty.any():
        abs_sum[nonempty] = np.add.reduceat(prod, row_ptrs[:-1][nonempty])
    expected = np.asarray(expected, np.float64)
    actual = np.asarray(actual, np.float64)
    ok_nonfinite = (~np.isfinite(expected)) & ((expected == actual) | (np.isnan(expected) & np.isnan(actual)))
    scale = np.maximum(np.maximum(np.abs(expected), abs_sum), 1e-30)
    err = np.where(ok_nonfinite, 0.0, np.abs(expected - actual) / scale)
    err = np.where(np.isfinite(expected) | ok_nonfinite, err, np.inf)
    return float(err.max()) if err.size else 0.0
