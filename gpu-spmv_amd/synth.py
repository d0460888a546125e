"""synth.py — numpy twin of csrc/generators.hip.

Every array is a pure function of (seed, row, slot), so the host (oracle side of
a parity test, CPU baseline) and the device (benchmark inputs built directly in
HBM) produce bit-identical synthetic matrices for BASELINE.md's configs:

* uniform_csr      — exactly k entries per row, columns = a uniform random
                     k-subset of [0, n_cols) in ascending order (configs 1, 2, 5)
* stratified_csr   — given row lengths, slot s of a row picks a column inside its
                     own stratum => unique, ascending (config 4, power-law rows)
* power_law_lengths— len = min(floor(4 * u^(-1/1.5)), 10000), one row forced to 10000
* vector           — uniform [-1, 1)
"""
from __future__ import annotations

import numpy as np

_U64 = np.uint64
STREAM_COLS, STREAM_VALS, STREAM_VEC, STREAM_LEN = 1, 2, 3, 4


def _mix64(z):
    z = (z + _U64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
    return z ^ (z >> _U64(31))


def draw(seed, stream, a, b):
    """One 64-bit draw per (seed, stream, a, b); a / b broadcast as numpy arrays."""
    with np.errstate(over="ignore"):
        a = np.asarray(a, dtype=np.uint64)
        b = np.asarray(b, dtype=np.uint64)
        h = _mix64(_U64(seed) ^ (_U64(stream) * _U64(0xD6E8FEB86659FD93)))
        h = _mix64(h ^ (a * _U64(0x9E3779B97F4A7C15)))
        h = _mix64(h ^ (b * _U64(0xC2B2AE3D27D4EB4F)))
    return h


def to_unit(h):
    """uniform in [-1, 1) from the top 24 bits"""
    return ((h >> _U64(40)).astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0)).astype(np.float32)


def to_range(h, m):
    """uniform integer in [0, m), m < 2**32 (m may be an array)"""
    with np.errstate(over="ignore"):
        return (((h >> _U64(32)) * np.asarray(m, dtype=np.uint64)) >> _U64(32)).astype(np.uint32)


def vector(seed, tag, n):
    return to_unit(draw(seed, STREAM_VEC, tag, np.arange(n, dtype=np.uint64)))


def uniform_csr(seed, row_begin, local_rows, n_cols, k):
    """(row_ptrs, col_indices, values) for rows [row_begin, row_begin + local_rows)."""
    assert 0 <= k <= n_cols and k <= 64
    rows = (np.arange(local_rows, dtype=np.uint64) + _U64(row_begin))[:, None]
    slots = np.arange(k, dtype=np.uint64)[None, :]
    picks = to_range(draw(seed, STREAM_COLS, rows, slots), n_cols - k + 1)
    picks = np.sort(picks, axis=1, kind="stable").astype(np.int64) + np.arange(k, dtype=np.int64)[None, :]
    vals = to_unit(draw(seed, STREAM_VALS, rows, slots))
    row_ptrs = (np.arange(local_rows + 1, dtype=np.int64) * k).astype(np.int32)
    return row_ptrs, picks.astype(np.int32).reshape(-1), vals.reshape(-1)


def stratified_csr(seed, row_begin, row_lengths, n_cols):
    row_lengths = np.asarray(row_lengths, dtype=np.int64)
    local_rows = row_lengths.size
    row_ptrs = np.zeros(local_rows + 1, dtype=np.int64)
    np.cumsum(row_lengths, out=row_ptrs[1:])
    nnz = int(row_ptrs[-1])
    assert nnz < 2**31
    row_of = np.repeat(np.arange(local_rows, dtype=np.int64), row_lengths)
    slot = np.arange(nnz, dtype=np.int64) - row_ptrs[row_of]
    length = row_lengths[row_of]
    lo = slot * n_cols // length
    hi = (slot + 1) * n_cols // length
    h = draw(seed, STREAM_COLS, (row_of + row_begin).astype(np.uint64), slot.astype(np.uint64))
    cols = lo + to_range(h, (hi - lo).astype(np.uint64)).astype(np.int64)
    vals = to_unit(draw(seed, STREAM_VALS, (row_of + row_begin).astype(np.uint64), slot.astype(np.uint64)))
    return row_ptrs.astype(np.int32), cols.astype(np.int32), vals


def power_law_lengths(seed, num_rows, x_min=4, alpha=1.5, max_len=10000, n_cols=None):
    """Pareto row lengths: min x_min, mean ~ 3 * x_min - 1, capped."""
    h = draw(seed, STREAM_LEN, np.arange(num_rows, dtype=np.uint64), 0)
    u = ((h >> _U64(11)).astype(np.float64) + 1.0) / float(1 << 53)       # (0, 1]
    lens = np.floor(x_min * u ** (-1.0 / alpha))
    cap = max_len if n_cols is None else min(max_len, n_cols)
    lens = np.minimum(lens, cap).astype(np.int64)
    lens[num_rows // 2] = cap          # at least one row at the cap => skewness >= 10
    return lens


def column_stochastic_values(col_indices, n_cols):
    """values = 1 / (number of stored entries in the column) — PageRank transition weights."""
    counts = np.bincount(col_indices, minlength=n_cols).astype(np.int32)
    return (np.float32(1.0) / counts[col_indices].astype(np.float32)).astype(np.float32)
