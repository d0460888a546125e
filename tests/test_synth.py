"""Properties of the synthetic-input generators (numpy side; the device twins are
checked against these in tests/test_gpu_generators.py)."""
import numpy as np


def test_uniform_csr_structure(spmv):
    rp, ci, va = spmv.synth.uniform_csr(42, 0, 2000, 5000, 16)
    assert rp[0] == 0 and (np.diff(rp) == 16).all() and ci.size == 32000
    rows = ci.reshape(2000, 16)
    assert (np.diff(rows, axis=1) > 0).all()                 # unique, ascending inside a row
    assert rows.min() >= 0 and rows.max() < 5000
    assert va.dtype == np.float32 and va.min() >= -1 and va.max() < 1
    # column marginal is close to uniform
    hist = np.bincount(ci, minlength=5000)
    assert abs(hist.mean() - 6.4) < 1e-9 and hist.max() < 25


def test_uniform_csr_is_shardable(spmv):
    """rows depend on (seed, global row) only: any row range regenerates identically"""
    full = spmv.synth.uniform_csr(1, 0, 100, 1000, 8)
    part = spmv.synth.uniform_csr(1, 40, 30, 1000, 8)
    np.testing.assert_array_equal(full[1][40 * 8:70 * 8], part[1])
    np.testing.assert_array_equal(full[2][40 * 8:70 * 8], part[2])


def test_power_law_lengths_and_stratified_columns(spmv):
    lens = spmv.synth.power_law_lengths(42, 200000)
    assert lens.min() == 4 and lens.max() == 10000 and 9 < lens.mean() < 14
    assert lens.max() / (lens.min() + 1) >= 10               # selector sees skew >= 10
    rp, ci, va = spmv.synth.stratified_csr(42, 0, lens[:3000], 1_000_000)
    for r in (0, 17, 1500, 2999):
        seg = ci[rp[r]:rp[r + 1]]
        assert (np.diff(seg) > 0).all() and seg.min() >= 0 and seg.max() < 1_000_000


def test_column_stochastic_values(spmv):
    rp, ci, _ = spmv.synth.uniform_csr(3, 0, 500, 500, 8)
    va = spmv.synth.column_stochastic_values(ci, 500)
    sums = np.bincount(ci, weights=va.astype(np.float64), minlength=500)
    used = np.bincount(ci, minlength=500) > 0
    assert np.allclose(sums[used], 1.0, atol=1e-5)
