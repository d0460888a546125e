"""workloads.py — builds BASELINE.md's synthetic inputs directly in HBM through
the C ABI (device generators), for bench.py, the smoke test and the full-size
GPU tests.  Plain ctypes; torch is not needed here."""
from __future__ import annotations

import ctypes
from ctypes import byref, c_void_p

import numpy as np

from . import (CudaBuffer, SpMVConfig, csr_destroy, csr_wrap_device, ell_destroy, ell_wrap_device, lib,
               spmv_csr, spmv_ell, synth)


class DeviceCSR:
    """A CSR matrix whose arrays live only in HBM (owned CudaBuffers + a wrapped header)."""

    def __init__(self, rows, cols, nnz):
        self.rows, self.cols, self.nnz = rows, cols, nnz
        self.row_ptrs = CudaBuffer(rows + 1, "int32")
        self.col_indices = CudaBuffer(max(nnz, 1), "int32")
        self.values = CudaBuffer(max(nnz, 1), "float32")
        self.handle = csr_wrap_device(rows, cols, nnz, self.row_ptrs.get(), self.col_indices.get(),
                                      self.values.get())

    def to_host(self):
        return (self.row_ptrs.copyToHost(self.rows + 1), self.col_indices.copyToHost(self.nnz),
                self.values.copyToHost(self.nnz))

    def close(self):
        if self.handle is not None:
            csr_destroy(self.handle)
            self.handle = None
        for b in (self.row_ptrs, self.col_indices, self.values):
            b.release()


def _check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed: {lib().spmv_c_error_string(status).decode()}")


def uniform_csr_device(seed, n_rows, n_cols, k, row_begin=0, stream=None) -> DeviceCSR:
    """Rows [row_begin, row_begin + n_rows) of the uniform k-per-row matrix (synth.uniform_csr)."""
    A = DeviceCSR(n_rows, n_cols, n_rows * k)
    _check(lib().spmv_c_gen_uniform_rows(seed, row_begin, n_rows, n_cols, k, A.row_ptrs.get(),
                                         A.col_indices.get(), A.values.get(), c_void_p(stream)),
           "gen_uniform_rows")
    lib().spmv_c_device_synchronize()
    return A


def power_law_csr_device(seed, n_rows, n_cols, max_len=10000, stream=None) -> DeviceCSR:
    """BASELINE config 4: Pareto(1.5, 4) row lengths capped at max_len (lengths on the host,
    entries generated in HBM; twin of synth.stratified_csr)."""
    lens = synth.power_law_lengths(seed, n_rows, max_len=max_len, n_cols=n_cols)
    row_ptrs = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(lens, out=row_ptrs[1:])
    nnz = int(row_ptrs[-1])
    A = DeviceCSR(n_rows, n_cols, nnz)
    A.row_ptrs.copyFromHost(row_ptrs.astype(np.int32), n_rows + 1)
    _check(lib().spmv_c_gen_stratified_rows(seed, 0, n_rows, n_cols, A.row_ptrs.get(),
                                            A.col_indices.get(), A.values.get(), c_void_p(stream)),
           "gen_stratified_rows")
    lib().spmv_c_device_synchronize()
    return A


def vector_device(seed, tag, n, stream=None) -> CudaBuffer:
    x = CudaBuffer(n, "float32")
    _check(lib().spmv_c_gen_vector(seed, tag, n, x.get(), c_void_p(stream)), "gen_vector")
    lib().spmv_c_device_synchronize()
    return x


def make_column_stochastic(A: DeviceCSR, counts: CudaBuffer = None, stream=None) -> CudaBuffer:
    """values[j] = 1 / (#entries in column col[j]).  `counts` (int32[n_cols]) may be passed
    in already summed over all shards; otherwise it is counted from A alone."""
    if counts is None:
        counts = CudaBuffer(A.cols, "int32")
        counts.copyFromHost(np.zeros(A.cols, np.int32), A.cols)
        _check(lib().spmv_c_count_columns(A.nnz, A.col_indices.get(), A.cols, counts.get(), c_void_p(stream)),
               "count_columns")
    _check(lib().spmv_c_reciprocal_values(A.nnz, A.col_indices.get(), counts.get(), A.values.get(),
                                          c_void_p(stream)), "reciprocal_values")
    lib().spmv_c_device_synchronize()
    return counts


def time_spmv_csr(A, d_x, d_y, kernel_type, warmup=5, runs=20, use_texture=False):
    """Reference benchmark protocol (include/spmv/benchmark.h:39: 5 warm-up + 20 timed calls),
    per-call kernel-only event time as reported by spmv_csr.  Returns list of ms."""
    cfg = SpMVConfig(kernel_type=kernel_type, use_texture=use_texture)
    handle = A.handle if isinstance(A, DeviceCSR) else A
    cols = handle.contents.num_cols
    for _ in range(warmup):
        r = spmv_csr(handle, d_x, d_y, cfg, cols)
        _check(r.error_code, "spmv_csr")
    times = []
    for _ in range(runs):
        r = spmv_csr(handle, d_x, d_y, cfg, cols)
        _check(r.error_code, "spmv_csr")
        times.append(float(r.elapsed_ms))
    return times


class DeviceELL:
    """Column-major ELL slabs that live only in HBM."""

    def __init__(self, rows, cols, k):
        self.rows, self.cols, self.k = rows, cols, k
        self.col_indices = CudaBuffer(max(rows * k, 1), "int32")
        self.values = CudaBuffer(max(rows * k, 1), "float32")
        self.handle = ell_wrap_device(rows, cols, k, self.col_indices.get(), self.values.get())

    def close(self):
        if self.handle is not None:
            ell_destroy(self.handle)
            self.handle = None
        self.col_indices.release()
        self.values.release()


def uniform_ell_device(seed, n_rows, n_cols, k, stream=None) -> DeviceELL:
    """BASELINE config 3: fixed k entries per row, no padding, column-major (== ell_from_csr of uniform_csr)."""
    E = DeviceELL(n_rows, n_cols, k)
    _check(lib().spmv_c_gen_uniform_ell(seed, n_rows, n_cols, k, E.col_indices.get(), E.values.get(),
                                        c_void_p(stream)), "gen_uniform_ell")
    lib().spmv_c_device_synchronize()
    return E


def time_spmv_ell(E, d_x, d_y, warmup=5, runs=20, use_texture=False):
    handle = E.handle if isinstance(E, DeviceELL) else E
    cols = handle.contents.num_cols
    cfg = SpMVConfig(kernel_type=SpMVConfig.ELL_KERNEL, use_texture=use_texture)
    for _ in range(warmup):
        _check(spmv_ell(handle, d_x, d_y, cfg, cols).error_code, "spmv_ell")
    times = []
    for _ in range(runs):
        r = spmv_ell(handle, d_x, d_y, cfg, cols)
        _check(r.error_code, "spmv_ell")
        times.append(float(r.elapsed_ms))
    return times
