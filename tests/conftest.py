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
    """Fails loudly (no silent CPU route) when a gpu-marked test runs without a device."""
    spmv.require_gpu()
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
