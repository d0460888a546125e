"""CPU oracle for the SpMV / PageRank hot path — TEST INFRASTRUCTURE ONLY.

ctypes + numpy front end of ``oracle/spmv_oracle.c`` (our plain-C restatement of
the reference's ``src/spmv_cpu.cpp`` / ``csr_matrix.cpp`` / ``ell_matrix.cpp`` /
``pagerank.cu``; every C function cites the lines it follows) and of
``oracle/_ref/ref_cpu`` (the reference's own CPU sources compiled by
``oracle/Makefile``; used to validate the restatement and to make
``tests/golden/ref_*.npz``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker.  Nothing under
``gpu-spmv_amd/`` imports it.
"""
from __future__ import annotations

import ctypes
import os
import struct
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_BINARY = os.path.join(_HERE, "_ref", "ref_cpu")

_lib = None


def build() -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True, stdout=subprocess.DEVNULL)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_bytes_csr.restype = ctypes.c_double
        _lib.oracle_bytes_ell.restype = ctypes.c_double
    return _lib


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ---- SpMV (src/spmv_cpu.cpp:6-32) -------------------------------------------------
def spmv_csr(row_ptrs, col_indices, values, x) -> np.ndarray:
    row_ptrs, col_indices, values, x = _i32(row_ptrs), _i32(col_indices), _f32(values), _f32(x)
    rows = row_ptrs.size - 1
    y = np.empty(rows, dtype=np.float32)
    lib().oracle_spmv_csr(ctypes.c_int(rows), _p(row_ptrs), _p(col_indices), _p(values), _p(x), _p(y))
    return y


def spmv_csr_parallel(row_ptrs, col_indices, values, x, threads) -> np.ndarray:
    """OpenMP static row blocks over `threads` host threads; bit-identical to spmv_csr."""
    row_ptrs, col_indices, values, x = _i32(row_ptrs), _i32(col_indices), _f32(values), _f32(x)
    rows = row_ptrs.size - 1
    y = np.empty(rows, dtype=np.float32)
    lib().oracle_spmv_csr_parallel(ctypes.c_int(rows), _p(row_ptrs), _p(col_indices), _p(values), _p(x), _p(y),
                                   ctypes.c_int(int(threads)))
    return y


def spmv_ell(num_rows, max_nnz_per_row, col_indices, values, x) -> np.ndarray:
    col_indices, values, x = _i32(col_indices), _f32(values), _f32(x)
    y = np.empty(num_rows, dtype=np.float32)
    lib().oracle_spmv_ell(ctypes.c_int(num_rows), ctypes.c_int(max_nnz_per_row), _p(col_indices),
                          _p(values), _p(x), _p(y))
    return y


# ---- containers (src/csr_matrix.cpp:50-95, src/ell_matrix.cpp:111-159) -------------
def csr_from_dense(dense):
    dense = _f32(dense)
    rows, cols = dense.shape
    nnz = lib().oracle_count_nonzeros(_p(dense), ctypes.c_int(rows), ctypes.c_int(cols))
    row_ptrs = np.empty(rows + 1, dtype=np.int32)
    col_indices = np.empty(nnz, dtype=np.int32)
    values = np.empty(nnz, dtype=np.float32)
    lib().oracle_csr_from_dense(_p(dense), ctypes.c_int(rows), ctypes.c_int(cols), _p(row_ptrs),
                                _p(col_indices), _p(values))
    return row_ptrs, col_indices, values


def ell_from_csr(row_ptrs, col_indices, values):
    row_ptrs, col_indices, values = _i32(row_ptrs), _i32(col_indices), _f32(values)
    rows = row_ptrs.size - 1
    k = lib().oracle_max_row_nnz(ctypes.c_int(rows), _p(row_ptrs))
    ell_cols = np.empty(rows * k, dtype=np.int32)
    ell_vals = np.empty(rows * k, dtype=np.float32)
    lib().oracle_ell_from_csr(ctypes.c_int(rows), ctypes.c_int(k), _p(row_ptrs), _p(col_indices),
                              _p(values), _p(ell_cols), _p(ell_vals))
    return k, ell_cols, ell_vals


def csr_stats(row_ptrs, nnz=None):
    """(avg, max, min, skewness) — src/csr_matrix.cpp:281-300."""
    row_ptrs = _i32(row_ptrs)
    rows = row_ptrs.size - 1
    if nnz is None:
        nnz = int(row_ptrs[-1]) if rows >= 0 and row_ptrs.size else 0
    out = np.zeros(4, dtype=np.float32)
    lib().oracle_csr_stats(ctypes.c_int(rows), ctypes.c_int(nnz), _p(row_ptrs), _p(out))
    return float(out[0]), int(out[1]), int(out[2]), float(out[3])


def auto_config(row_ptrs, num_cols):
    """(kernel_type, use_texture) with the reference's thresholds — src/spmv_cpu.cpp:34-50."""
    row_ptrs = _i32(row_ptrs)
    rows = row_ptrs.size - 1
    tex = ctypes.c_int(0)
    kt = lib().oracle_auto_config(ctypes.c_int(rows), ctypes.c_int(num_cols), ctypes.c_int(int(row_ptrs[-1])),
                                  _p(row_ptrs), ctypes.byref(tex))
    return int(kt), bool(tex.value)


def bytes_csr(rows, cols, nnz) -> float:
    return float(lib().oracle_bytes_csr(ctypes.c_int(rows), ctypes.c_int(cols), ctypes.c_int(nnz)))


def bytes_ell(rows, cols, k) -> float:
    return float(lib().oracle_bytes_ell(ctypes.c_int(rows), ctypes.c_int(cols), ctypes.c_int(k)))


# ---- PageRank (src/pagerank.cu:20-153) -------------------------------------------
def dangling_mask(row_ptrs, col_indices, values, num_cols) -> np.ndarray:
    row_ptrs, col_indices, values = _i32(row_ptrs), _i32(col_indices), _f32(values)
    mask = np.zeros(max(num_cols, 1), dtype=np.uint8)
    lib().oracle_dangling_mask(ctypes.c_int(row_ptrs.size - 1), ctypes.c_int(num_cols), _p(row_ptrs),
                               _p(col_indices), _p(values), _p(mask))
    return mask[:num_cols]


def pagerank(row_ptrs, col_indices, values, num_cols=None, damping=0.85, tolerance=1e-6,
             max_iterations=100, wide_sums=False):
    """Returns (ranks, iterations, final_residual, converged)."""
    row_ptrs, col_indices, values = _i32(row_ptrs), _i32(col_indices), _f32(values)
    n = row_ptrs.size - 1
    if num_cols is None:
        num_cols = n
    ranks = np.zeros(max(n, 1), dtype=np.float32)
    res = ctypes.c_float(0.0)
    conv = ctypes.c_int(0)
    iters = lib().oracle_pagerank(ctypes.c_int(n), ctypes.c_int(num_cols), _p(row_ptrs), _p(col_indices),
                                  _p(values), ctypes.c_float(damping), ctypes.c_float(tolerance),
                                  ctypes.c_int(max_iterations), ctypes.c_int(1 if wide_sums else 0),
                                  _p(ranks), ctypes.byref(res), ctypes.byref(conv))
    return ranks[:n], int(iters), float(res.value), bool(conv.value)


# ---- the compiled reference (oracle/_ref/ref_cpu) ---------------------------------
def have_reference_binary() -> bool:
    return os.path.exists(REF_BINARY) and os.access(REF_BINARY, os.X_OK)


def _parse_records(blob: bytes) -> dict:
    out, pos = {}, 0
    while pos < len(blob):
        (name_len,) = struct.unpack_from("<i", blob, pos)
        pos += 4
        name = blob[pos:pos + name_len].decode()
        pos += name_len
        kind = chr(blob[pos])
        pos += 1
        (count,) = struct.unpack_from("<q", blob, pos)
        pos += 8
        dtype = {"i": np.int32, "f": np.float32, "b": np.uint8}[kind]
        width = np.dtype(dtype).itemsize
        out[name] = np.frombuffer(blob, dtype=dtype, count=count, offset=pos).copy()
        pos += count * width
    return out


def reference_case(dense, x) -> dict:
    """Runs the reference's own csr_from_dense / ell_from_* / spmv_cpu_* / stats /
    selector / serialisers on (dense, x) and returns every output by name."""
    dense, x = _f32(dense), _f32(x)
    rows, cols = dense.shape
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(src, "wb") as f:
            f.write(struct.pack("<ii", rows, cols))
            f.write(dense.tobytes())
            f.write(x.tobytes())
        subprocess.run([REF_BINARY, "case", src, dst], check=True)
        with open(dst, "rb") as f:
            return _parse_records(f.read())


def reference_time_csr(row_ptrs, col_indices, values, x, reps=3, scratch_dir=None) -> float:
    """Best-of-`reps` seconds of the reference's spmv_cpu_csr on one host thread."""
    row_ptrs, col_indices, values, x = _i32(row_ptrs), _i32(col_indices), _f32(values), _f32(x)
    rows, nnz = row_ptrs.size - 1, col_indices.size
    with tempfile.TemporaryDirectory(dir=scratch_dir) as tmp:
        src = os.path.join(tmp, "csr.bin")
        with open(src, "wb") as f:
            f.write(struct.pack("<iii", rows, x.size, nnz))
            for arr in (row_ptrs, col_indices, values, x):
                f.write(memoryview(arr))
        out = subprocess.run([REF_BINARY, "time", src, str(reps)], check=True, capture_output=True, text=True)
    return float(out.stdout.split()[1])
