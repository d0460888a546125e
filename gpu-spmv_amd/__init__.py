"""gpu-spmv_amd — Python host mirror of the MI355X-native SpMV library.

The product is ``lib/libspmv_amd.so`` (hand-written HIP kernels for gfx950 + the
C++ API in ``namespace spmv`` + the C ABI of ``include/spmv_c.h``).  This module
binds that C ABI with ctypes and mirrors the reference's interface
(LessUp/gpu-spmv ``include/spmv/*.h``): same function names, argument meaning
and error behaviour, so tests written against it read like the reference's own
(``tests/test_spmv.cu`` etc.).

There is no CPU fallback: importing works without a GPU (the library loads, the
host-side containers work), but every device entry point needs a HIP device and
the library itself must have been built (``python __graft_entry__.py`` or
``make -C gpu-spmv_amd``) — otherwise ``LibraryNotBuilt`` is raised.

The directory name carries a hyphen, so import it with
``importlib.import_module("gpu-spmv_amd")``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64
from ctypes import c_size_t, c_uint8, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPMV_AMD_LIB") or os.path.join(_HERE, "lib", "libspmv_amd.so")   # SPMV_AMD_LIB: an A/B build of the same library (tools/)


class LibraryNotBuilt(ImportError):
    pass


class SpMVError:
    """reference include/spmv/common.h:13-23"""
    SUCCESS = 0
    INVALID_DIMENSION = -1
    CUDA_MALLOC = -2
    CUDA_MEMCPY = -3
    KERNEL_LAUNCH = -4
    INVALID_FORMAT = -5
    FILE_IO = -6
    OUT_OF_MEMORY = -7
    INVALID_ARGUMENT = -8


class CudaException(RuntimeError):
    """Raised by CudaBuffer on device allocation / copy failure (common.h:42-50)."""


# ---- struct mirrors (layouts asserted in csrc/capi.cpp) -----------------------------
class CSRMatrix(Structure):
    """reference include/spmv/csr_matrix.h:11-28"""
    _fields_ = [("num_rows", c_int32), ("num_cols", c_int32), ("nnz", c_int32),
                ("values", POINTER(c_float)), ("col_indices", POINTER(c_int32)),
                ("row_ptrs", POINTER(c_int32)),
                ("d_values", c_void_p), ("d_col_indices", c_void_p), ("d_row_ptrs", c_void_p),
                ("owns_host_memory", c_uint8), ("owns_device_memory", c_uint8)]


class ELLMatrix(Structure):
    """reference include/spmv/ell_matrix.h:12-28"""
    _fields_ = [("num_rows", c_int32), ("num_cols", c_int32), ("max_nnz_per_row", c_int32),
                ("values", POINTER(c_float)), ("col_indices", POINTER(c_int32)),
                ("d_values", c_void_p), ("d_col_indices", c_void_p),
                ("owns_host_memory", c_uint8), ("owns_device_memory", c_uint8)]


class CSRStats(Structure):
    """reference include/spmv/csr_matrix.h:64-69"""
    _fields_ = [("avg_nnz_per_row", c_float), ("max_nnz_per_row", c_int32),
                ("min_nnz_per_row", c_int32), ("skewness", c_float)]


class SpMVConfig(Structure):
    """reference include/spmv/spmv.h:11-24 (defaults SCALAR_CSR / 256 / False)"""
    SCALAR_CSR, VECTOR_CSR, MERGE_PATH, ELL_KERNEL = 0, 1, 2, 3
    _fields_ = [("kernel_type", c_int32), ("block_size", c_int32), ("use_texture", c_uint8)]

    def __init__(self, kernel_type=0, block_size=256, use_texture=False):
        super().__init__(kernel_type, block_size, 1 if use_texture else 0)


class SpMVResult(Structure):
    """reference include/spmv/spmv.h:27-36"""
    _fields_ = [("y", c_void_p), ("elapsed_ms", c_float), ("gflops", c_float),
                ("bandwidth_gb_s", c_float), ("error_code", c_int32)]


class BandwidthMetrics(Structure):
    """reference include/spmv/bandwidth.h:10-18"""
    _fields_ = [("theoretical_bandwidth_gb_s", c_float), ("achieved_bandwidth_gb_s", c_float),
                ("efficiency", c_float)]


class PageRankConfig(Structure):
    """reference include/spmv/pagerank.h:9-15 (defaults 0.85 / 1e-6 / 100)"""
    _fields_ = [("damping_factor", c_float), ("tolerance", c_float), ("max_iterations", c_int32)]

    def __init__(self, damping_factor=0.85, tolerance=1e-6, max_iterations=100):
        super().__init__(damping_factor, tolerance, max_iterations)


class _PageRankResultC(Structure):
    _fields_ = [("ranks", POINTER(c_float)), ("iterations", c_int32), ("final_residual", c_float),
                ("converged", c_uint8)]


class TopKNode(Structure):
    """reference include/spmv/pagerank.h:38-41"""
    _fields_ = [("node_id", c_int32), ("rank", c_float)]


class PrStatus(Structure):
    _fields_ = [("dangling_sum", c_float), ("final_residual", c_float), ("iterations", c_int32),
                ("converged", c_int32), ("done", c_int32), ("reserved", c_int32)]


class PageRankResult:
    """reference include/spmv/pagerank.h:18-25; `ranks` is a numpy copy (the C buffer is freed)."""

    def __init__(self, ranks, iterations, final_residual, converged):
        self.ranks = ranks
        self.iterations = iterations
        self.final_residual = final_residual
        self.converged = converged


# ---- library loading -----------------------------------------------------------------
_lib = None

_SIGNATURES = {
    # name: (restype, argtypes)
    "spmv_c_error_string": (c_char_p, [c_int]),
    "spmv_c_version": (c_char_p, []),
    "spmv_c_device_count": (c_int, []),
    "spmv_c_device_name": (c_int, [c_char_p, c_size_t]),
    "spmv_c_set_device": (c_int, [c_int]),
    "spmv_c_enable_peer_access": (c_int, [c_int]),
    "spmv_c_set_stream": (None, [c_void_p]),
    "spmv_c_device_malloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "spmv_c_device_free": (c_int, [c_void_p]),
    "spmv_c_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t]),
    "spmv_c_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t]),
    "spmv_c_device_synchronize": (c_int, []),
    "spmv_c_ipc_get_handle": (c_int, [c_void_p, c_char_p]),
    "spmv_c_ipc_open_handle": (c_int, [c_char_p, POINTER(c_void_p)]),
    "spmv_c_ipc_close": (c_int, [c_void_p]),
    "spmv_c_csr_create": (POINTER(CSRMatrix), [c_int, c_int, c_int]),
    "spmv_c_csr_destroy": (None, [POINTER(CSRMatrix)]),
    "spmv_c_csr_from_dense": (c_int, [POINTER(CSRMatrix), c_void_p, c_int, c_int]),
    "spmv_c_csr_to_dense": (c_int, [POINTER(CSRMatrix), c_void_p]),
    "spmv_c_csr_get_element": (c_float, [POINTER(CSRMatrix), c_int, c_int]),
    "spmv_c_csr_to_gpu": (c_int, [POINTER(CSRMatrix)]),
    "spmv_c_csr_from_gpu": (c_int, [POINTER(CSRMatrix)]),
    "spmv_c_csr_free_gpu": (None, [POINTER(CSRMatrix)]),
    "spmv_c_csr_invalidate_gpu_cache": (None, [POINTER(CSRMatrix)]),
    "spmv_c_csr_serialize": (c_int, [POINTER(CSRMatrix), c_char_p]),
    "spmv_c_csr_deserialize": (c_int, [POINTER(CSRMatrix), c_char_p]),
    "spmv_c_csr_compute_stats": (c_int, [POINTER(CSRMatrix), POINTER(CSRStats)]),
    "spmv_c_csr_wrap_device": (POINTER(CSRMatrix), [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "spmv_c_ell_create": (POINTER(ELLMatrix), [c_int, c_int, c_int]),
    "spmv_c_ell_destroy": (None, [POINTER(ELLMatrix)]),
    "spmv_c_ell_from_dense": (c_int, [POINTER(ELLMatrix), c_void_p, c_int, c_int]),
    "spmv_c_ell_from_csr": (c_int, [POINTER(ELLMatrix), POINTER(CSRMatrix)]),
    "spmv_c_ell_from_csr_gpu": (c_int, [POINTER(ELLMatrix), POINTER(CSRMatrix)]),
    "spmv_c_ell_to_dense": (c_int, [POINTER(ELLMatrix), c_void_p]),
    "spmv_c_ell_get_element": (c_float, [POINTER(ELLMatrix), c_int, c_int]),
    "spmv_c_ell_to_gpu": (c_int, [POINTER(ELLMatrix)]),
    "spmv_c_ell_from_gpu": (c_int, [POINTER(ELLMatrix)]),
    "spmv_c_ell_free_gpu": (None, [POINTER(ELLMatrix)]),
    "spmv_c_ell_invalidate_gpu_cache": (None, [POINTER(ELLMatrix)]),
    "spmv_c_ell_serialize": (c_int, [POINTER(ELLMatrix), c_char_p]),
    "spmv_c_ell_deserialize": (c_int, [POINTER(ELLMatrix), c_char_p]),
    "spmv_c_ell_index": (c_int, [c_int, c_int, c_int]),
    "spmv_c_ell_wrap_device": (POINTER(ELLMatrix), [c_int, c_int, c_int, c_void_p, c_void_p]),
    "spmv_c_cpu_csr": (None, [POINTER(CSRMatrix), c_void_p, c_void_p]),
    "spmv_c_cpu_ell": (None, [POINTER(ELLMatrix), c_void_p, c_void_p]),
    "spmv_c_spmv_csr": (c_int, [POINTER(CSRMatrix), c_void_p, c_void_p, POINTER(SpMVConfig), c_int,
                                POINTER(SpMVResult)]),
    "spmv_c_spmv_ell": (c_int, [POINTER(ELLMatrix), c_void_p, c_void_p, POINTER(SpMVConfig), c_int,
                                POINTER(SpMVResult)]),
    "spmv_c_auto_config": (c_int, [POINTER(CSRMatrix), POINTER(SpMVConfig)]),
    "spmv_c_validate_dimensions": (c_int, [c_int, c_int]),
    "spmv_c_csr_has_tiled_plan": (c_int, [POINTER(CSRMatrix)]),
    "spmv_c_set_tiled_promotion": (None, [c_int]),
    "spmv_c_get_tiled_promotion": (c_int, []),
    "spmv_c_tiled_shape": (c_int, [c_int64, c_int64, c_int64, POINTER(c_int32), POINTER(c_int32)]),
    "spmv_c_csr_tiled_info": (c_int, [POINTER(CSRMatrix), POINTER(c_int64)]),
    "spmv_c_csr_tiled_stats": (c_int, [POINTER(CSRMatrix), POINTER(c_double)]),
    "spmv_c_csr_tiled_checksum": (c_int, [POINTER(CSRMatrix), POINTER(c_uint64)]),
    "spmv_c_csr_tiled_folded": (c_int, [POINTER(CSRMatrix)]),
    "spmv_c_spmv_csr_async": (c_int, [POINTER(CSRMatrix), c_void_p, c_void_p, POINTER(SpMVConfig), c_int,
                                      c_void_p]),
    "spmv_c_spmv_ell_async": (c_int, [POINTER(ELLMatrix), c_void_p, c_void_p, POINTER(SpMVConfig), c_int,
                                      c_void_p]),
    "spmv_c_compute_bandwidth_csr": (c_int, [POINTER(CSRMatrix), c_float, POINTER(BandwidthMetrics)]),
    "spmv_c_compute_bandwidth_ell": (c_int, [POINTER(ELLMatrix), c_float, POINTER(BandwidthMetrics)]),
    "spmv_c_get_gpu_peak_bandwidth": (c_float, []),
    "spmv_c_pagerank": (c_int, [POINTER(CSRMatrix), POINTER(PageRankConfig), POINTER(_PageRankResultC)]),
    "spmv_c_pagerank_free": (None, [POINTER(_PageRankResultC)]),
    "spmv_c_pagerank_top_k": (None, [POINTER(_PageRankResultC), c_int, c_int, POINTER(TopKNode)]),
    "spmv_c_pagerank_multi_gpu": (c_int, [POINTER(CSRMatrix), POINTER(PageRankConfig), c_int, POINTER(_PageRankResultC)]),
    "spmv_c_pagerank_shard_bounds": (c_int, [POINTER(c_int32), c_int, c_int, POINTER(c_int32)]),
    "spmv_c_pr_shard_create": (c_void_p, [POINTER(CSRMatrix), c_int, c_int, c_void_p]),
    "spmv_c_pr_shard_create_chunked": (c_void_p, [POINTER(CSRMatrix), c_int, c_int, c_int, c_int, c_void_p]),
    "spmv_c_pr_shard_destroy": (None, [c_void_p]),
    "spmv_c_pr_expand": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "spmv_c_pr_reset": (c_int, [c_void_p, c_float, c_void_p]),
    "spmv_c_pr_step": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "spmv_c_pr_step_push": (c_int, [c_void_p, c_void_p, c_void_p, c_float, POINTER(c_void_p), c_int, c_void_p]),
    "spmv_c_pr_reduce": (c_int, [c_void_p, c_void_p, c_void_p]),
    "spmv_c_pr_commit": (c_int, [c_void_p, c_void_p, c_float, c_void_p]),
    "spmv_c_pr_reduce_commit": (c_int, [c_void_p, c_float, c_void_p]),
    "spmv_c_pr_commit_gathered": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_float, c_void_p]),
    "spmv_c_pr_status_get": (c_int, [c_void_p, POINTER(PrStatus), c_void_p]),
    "spmv_c_pr_column_sums": (c_int, [POINTER(CSRMatrix), c_void_p, c_void_p]),
    "spmv_c_pr_mask_from_column_sums": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "spmv_c_fill": (c_int, [c_void_p, c_size_t, c_float, c_void_p]),
    "spmv_c_gen_uniform_rows": (c_int, [c_uint64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "spmv_c_gen_uniform_ell": (c_int, [c_uint64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "spmv_c_gen_stratified_rows": (c_int, [c_uint64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    "spmv_c_gen_vector": (c_int, [c_uint64, c_uint64, c_size_t, c_void_p, c_void_p]),
    "spmv_c_count_columns": (c_int, [c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "spmv_c_reciprocal_values": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def _share_torchs_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (same soname,
    libamdhip64.so.7, loaded by path); if libspmv_amd.so pulled in /opt/rocm's copy first, a later
    `import torch` would bring up a second runtime that finds no GPU.  So when torch is installed,
    its copy is loaded first (without importing torch) and libspmv_amd.so binds to it by soname —
    the same pairing that results when torch happens to be imported first."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    candidate = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(candidate):
        try:
            ctypes.CDLL(candidate, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> ctypes.CDLL:
    """The loaded libspmv_amd.so; raises LibraryNotBuilt when it is missing."""
    global _lib
    if _lib is None:
        _share_torchs_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise LibraryNotBuilt(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                f"(or `make -C gpu-spmv_amd`).  There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError here = header / library mismatch
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def version() -> str:
    return lib().spmv_c_version().decode()


def device_count() -> int:
    return lib().spmv_c_device_count()


def device_name() -> str:
    buf = ctypes.create_string_buffer(256)
    lib().spmv_c_device_name(buf, 256)
    return buf.value.decode()


def require_gpu() -> None:
    if device_count() < 1:
        raise RuntimeError("gpu-spmv_amd: no HIP device visible; the SpMV path has no CPU fallback")


def spmv_error_string(code: int) -> str:
    return lib().spmv_c_error_string(int(code)).decode()


def set_stream(stream_handle) -> None:
    lib().spmv_c_set_stream(c_void_p(stream_handle))


def device_synchronize() -> None:
    lib().spmv_c_device_synchronize()


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(c_void_p)


_NP_BY_NAME = {"float": np.float32, "float32": np.float32, "int": np.int32, "int32": np.int32,
               "double": np.float64, "float64": np.float64, "uint8": np.uint8, "int64": np.int64,
               "uint64": np.uint64}


class CudaBuffer:
    """RAII device buffer — reference include/spmv/cuda_buffer.h:12-101.

    Same behaviour: sized construction allocates in HBM (CudaException on failure),
    copies raise RuntimeError("Copy size exceeds buffer size") when count > size,
    resize discards contents, zero size holds a null pointer, move-only.
    """

    def __init__(self, count: int = 0, dtype="float32"):
        self._dtype = np.dtype(_NP_BY_NAME.get(dtype, dtype))
        self._ptr = c_void_p(None)
        self._size = 0
        if count:
            self._allocate(int(count))

    def _allocate(self, count: int) -> None:
        self._size = count
        if count > 0:
            ptr = c_void_p(None)
            status = lib().spmv_c_device_malloc(byref(ptr), count * self._dtype.itemsize)
            if status != 0:
                self._size = 0
                raise CudaException("CUDA error: " + spmv_error_string(status))
            self._ptr = ptr

    def get(self):
        """device address (int) or None"""
        return self._ptr.value

    def size(self) -> int:
        return self._size

    def empty(self) -> bool:
        return self._ptr.value is None or self._size == 0

    def copyFromHost(self, host_data, count: int) -> None:
        if count > self._size:
            raise RuntimeError("Copy size exceeds buffer size")
        src = np.ascontiguousarray(host_data, dtype=self._dtype)
        status = lib().spmv_c_memcpy_h2d(self._ptr, _np_ptr(src), int(count) * self._dtype.itemsize)
        if status != 0:
            raise CudaException("CUDA error: " + spmv_error_string(status))

    def copyToHost(self, count: int = None) -> np.ndarray:
        if count is None:
            count = self._size
        if count > self._size:
            raise RuntimeError("Copy size exceeds buffer size")
        out = np.empty(count, dtype=self._dtype)
        status = lib().spmv_c_memcpy_d2h(_np_ptr(out), self._ptr, int(count) * self._dtype.itemsize)
        if status != 0:
            raise CudaException("CUDA error: " + spmv_error_string(status))
        return out

    def resize(self, new_count: int) -> None:
        if new_count == self._size:
            return
        self.release()
        self._allocate(int(new_count))

    def release(self) -> None:
        if self._ptr.value is not None:
            lib().spmv_c_device_free(self._ptr)
            self._ptr = c_void_p(None)
        self._size = 0

    def move(self) -> "CudaBuffer":
        """C++ move construction: the returned buffer owns the allocation, self is emptied."""
        other = CudaBuffer(0, self._dtype)
        other._ptr, other._size = self._ptr, self._size
        self._ptr, self._size = c_void_p(None), 0
        return other

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


# ---- CSR container (reference include/spmv/csr_matrix.h:31-71) ---------------------
def csr_create(rows, cols, nnz):
    p = lib().spmv_c_csr_create(rows, cols, nnz)
    return p if p else None


def csr_destroy(mat) -> None:
    if mat:
        lib().spmv_c_csr_destroy(mat)


def csr_from_dense(csr, dense, rows, cols) -> int:
    if dense is None:
        return lib().spmv_c_csr_from_dense(csr, None, rows, cols)
    d = np.ascontiguousarray(dense, dtype=np.float32)
    return lib().spmv_c_csr_from_dense(csr, _np_ptr(d), rows, cols)


def csr_to_dense(csr) -> np.ndarray:
    m = csr.contents
    out = np.empty((m.num_rows, m.num_cols), dtype=np.float32)
    status = lib().spmv_c_csr_to_dense(csr, _np_ptr(out))
    if status != 0:
        raise ValueError(spmv_error_string(status))
    return out


def csr_get_element(mat, row, col) -> float:
    return float(lib().spmv_c_csr_get_element(mat, row, col))


def csr_to_gpu(mat) -> int:
    return lib().spmv_c_csr_to_gpu(mat)


def csr_from_gpu(mat) -> int:
    return lib().spmv_c_csr_from_gpu(mat)


def csr_free_gpu(mat) -> None:
    lib().spmv_c_csr_free_gpu(mat)


def csr_invalidate_gpu_cache(mat) -> None:
    """After writing into the matrix's device arrays in place: drop the cached plan / tables."""
    lib().spmv_c_csr_invalidate_gpu_cache(mat)


def csr_serialize(mat, filename) -> int:
    return lib().spmv_c_csr_serialize(mat, os.fsencode(filename) if filename is not None else None)


def csr_deserialize(mat, filename) -> int:
    return lib().spmv_c_csr_deserialize(mat, os.fsencode(filename) if filename is not None else None)


def csr_compute_stats(mat) -> CSRStats:
    out = CSRStats()
    lib().spmv_c_csr_compute_stats(mat, byref(out))
    return out


def csr_wrap_device(rows, cols, nnz, d_row_ptrs, d_col_indices, d_values):
    """Header over caller-owned device arrays (e.g. torch tensors' data_ptr())."""
    p = lib().spmv_c_csr_wrap_device(rows, cols, nnz, c_void_p(d_row_ptrs), c_void_p(d_col_indices),
                                     c_void_p(d_values))
    return p if p else None


def csr_from_arrays(num_rows, num_cols, row_ptrs, col_indices, values):
    """Convenience: csr_create + fill of the host arrays (what the reference's callers do by hand)."""
    row_ptrs = np.ascontiguousarray(row_ptrs, dtype=np.int32)
    col_indices = np.ascontiguousarray(col_indices, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.float32)
    assert row_ptrs.size == num_rows + 1 and col_indices.size == values.size
    mat = csr_create(num_rows, num_cols, int(values.size))
    m = mat.contents
    ctypes.memmove(m.row_ptrs, _np_ptr(row_ptrs), row_ptrs.nbytes)
    if values.size:
        ctypes.memmove(m.col_indices, _np_ptr(col_indices), col_indices.nbytes)
        ctypes.memmove(m.values, _np_ptr(values), values.nbytes)
    return mat


def csr_host_arrays(mat):
    """(row_ptrs, col_indices, values) numpy copies of the host arrays."""
    m = mat.contents
    rp = np.ctypeslib.as_array(m.row_ptrs, shape=(m.num_rows + 1,)).copy()
    if m.nnz > 0:
        ci = np.ctypeslib.as_array(m.col_indices, shape=(m.nnz,)).copy()
        va = np.ctypeslib.as_array(m.values, shape=(m.nnz,)).copy()
    else:
        ci, va = np.empty(0, np.int32), np.empty(0, np.float32)
    return rp, ci, va


# ---- ELL container (reference include/spmv/ell_matrix.h:31-66) ---------------------
def ell_create(rows, cols, max_nnz_per_row):
    p = lib().spmv_c_ell_create(rows, cols, max_nnz_per_row)
    return p if p else None


def ell_destroy(mat) -> None:
    if mat:
        lib().spmv_c_ell_destroy(mat)


def ell_from_dense(ell, dense, rows, cols) -> int:
    if dense is None:
        return lib().spmv_c_ell_from_dense(ell, None, rows, cols)
    d = np.ascontiguousarray(dense, dtype=np.float32)
    return lib().spmv_c_ell_from_dense(ell, _np_ptr(d), rows, cols)


def ell_from_csr(ell, csr) -> int:
    return lib().spmv_c_ell_from_csr(ell, csr)


def ell_from_csr_gpu(ell, csr) -> int:
    """Extension: CSR -> ELL on the device (device slabs only; ell_from_gpu mirrors them to the host)."""
    return lib().spmv_c_ell_from_csr_gpu(ell, csr)


def ell_to_dense(ell) -> np.ndarray:
    m = ell.contents
    out = np.empty((m.num_rows, m.num_cols), dtype=np.float32)
    status = lib().spmv_c_ell_to_dense(ell, _np_ptr(out))
    if status != 0:
        raise ValueError(spmv_error_string(status))
    return out


def ell_get_element(mat, row, col) -> float:
    return float(lib().spmv_c_ell_get_element(mat, row, col))


def ell_to_gpu(mat) -> int:
    return lib().spmv_c_ell_to_gpu(mat)


def ell_from_gpu(mat) -> int:
    return lib().spmv_c_ell_from_gpu(mat)


def ell_free_gpu(mat) -> None:
    lib().spmv_c_ell_free_gpu(mat)


def ell_invalidate_gpu_cache(mat) -> None:
    lib().spmv_c_ell_invalidate_gpu_cache(mat)


def ell_serialize(mat, filename) -> int:
    return lib().spmv_c_ell_serialize(mat, os.fsencode(filename) if filename is not None else None)


def ell_deserialize(mat, filename) -> int:
    return lib().spmv_c_ell_deserialize(mat, os.fsencode(filename) if filename is not None else None)


def ell_index(row, k, num_rows) -> int:
    return lib().spmv_c_ell_index(row, k, num_rows)


def ell_wrap_device(rows, cols, max_nnz_per_row, d_col_indices, d_values):
    p = lib().spmv_c_ell_wrap_device(rows, cols, max_nnz_per_row, c_void_p(d_col_indices), c_void_p(d_values))
    return p if p else None


def ell_host_arrays(mat):
    m = mat.contents
    slots = m.num_rows * m.max_nnz_per_row
    if slots == 0:
        return np.empty(0, np.int32), np.empty(0, np.float32)
    ci = np.ctypeslib.as_array(m.col_indices, shape=(slots,)).copy()
    va = np.ctypeslib.as_array(m.values, shape=(slots,)).copy()
    return ci, va


# ---- SpMV (reference include/spmv/spmv.h:39-54) -------------------------------------
def spmv_cpu_csr(A, x) -> np.ndarray:
    """The library's host path (reference API parity); not used by any device entry point."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros(A.contents.num_rows, dtype=np.float32)
    lib().spmv_c_cpu_csr(A, _np_ptr(x), _np_ptr(y))
    return y


def spmv_cpu_ell(A, x) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros(A.contents.num_rows, dtype=np.float32)
    lib().spmv_c_cpu_ell(A, _np_ptr(x), _np_ptr(y))
    return y


def _dev(ptr):
    if isinstance(ptr, CudaBuffer):
        return c_void_p(ptr.get())
    return c_void_p(ptr)


def spmv_csr(A, d_x, d_y, config=None, vec_size=-1) -> SpMVResult:
    out = SpMVResult()
    lib().spmv_c_spmv_csr(A, _dev(d_x), _dev(d_y), byref(config) if config is not None else None,
                          vec_size, byref(out))
    return out


def spmv_ell(A, d_x, d_y, config=None, vec_size=-1) -> SpMVResult:
    out = SpMVResult()
    lib().spmv_c_spmv_ell(A, _dev(d_x), _dev(d_y), byref(config) if config is not None else None,
                          vec_size, byref(out))
    return out


def csr_tiled_checksum(A):
    """Four position-weighted checksums of the matrix's tiled plan (values, local columns, row deltas, cell
    table), or None without a plan."""
    out = (c_uint64 * 4)()
    if not lib().spmv_c_csr_tiled_checksum(A, out):
        return None
    return tuple(int(v) for v in out)


def csr_has_tiled_plan(A) -> bool:
    return bool(lib().spmv_c_csr_has_tiled_plan(A))


def set_tiled_promotion(calls: int) -> None:
    """spmv_set_tiled_promotion (include/spmv/spmv.h): VECTOR_CSR / MERGE_PATH callers without use_texture move to the
    LDS-tiled engine after `calls` calls on a large matrix; 0 = never."""
    lib().spmv_c_set_tiled_promotion(int(calls))


def get_tiled_promotion() -> int:
    return int(lib().spmv_c_get_tiled_promotion())


def tiled_shape(rows, cols, nnz):
    """(takes_it, strip_cols, tile_rows) the LDS-tiled engine would use for such a matrix."""
    w, r = c_int32(0), c_int32(0)
    takes = lib().spmv_c_tiled_shape(rows, cols, nnz, byref(w), byref(r))
    return bool(takes), w.value, r.value


def csr_tiled_info(A):
    """dict describing the matrix's cached tiled plan, or None."""
    out = (c_int64 * 8)()
    if not lib().spmv_c_csr_tiled_info(A, out):
        return None
    keys = ("strip_cols", "tile_rows", "num_strips", "num_tiles", "slots_in_cells", "long_rows",
            "slots_per_lane", "long_row_limit")
    info = dict(zip(keys, (int(v) for v in out)))
    info["values_folded"] = bool(lib().spmv_c_csr_tiled_folded(A))
    stats = (c_double * 4)()
    if lib().spmv_c_csr_tiled_stats(A, stats):
        info["build_ms"] = round(float(stats[0]), 3)
        info["plan_bytes"] = int(stats[1])
        info["entries_in_cells"] = int(stats[3])
    return info


def spmv_csr_async(A, d_x, d_y, config=None, vec_size=-1, stream=None) -> int:
    return lib().spmv_c_spmv_csr_async(A, _dev(d_x), _dev(d_y),
                                       byref(config) if config is not None else None, vec_size,
                                       c_void_p(stream))


def spmv_ell_async(A, d_x, d_y, config=None, vec_size=-1, stream=None) -> int:
    return lib().spmv_c_spmv_ell_async(A, _dev(d_x), _dev(d_y),
                                       byref(config) if config is not None else None, vec_size,
                                       c_void_p(stream))


def spmv_auto_config(A) -> SpMVConfig:
    out = SpMVConfig()
    status = lib().spmv_c_auto_config(A, byref(out))
    if status != 0:
        raise ValueError(spmv_error_string(status))
    return out


def spmv_validate_dimensions(num_cols, vec_size) -> bool:
    return bool(lib().spmv_c_validate_dimensions(num_cols, vec_size))


# ---- bandwidth (reference include/spmv/bandwidth.h:21-27) ---------------------------
def compute_bandwidth_csr(A, elapsed_ms) -> BandwidthMetrics:
    out = BandwidthMetrics()
    lib().spmv_c_compute_bandwidth_csr(A, elapsed_ms, byref(out))
    return out


def compute_bandwidth_ell(A, elapsed_ms) -> BandwidthMetrics:
    out = BandwidthMetrics()
    lib().spmv_c_compute_bandwidth_ell(A, elapsed_ms, byref(out))
    return out


def get_gpu_peak_bandwidth() -> float:
    return float(lib().spmv_c_get_gpu_peak_bandwidth())


# ---- PageRank (reference include/spmv/pagerank.h:29-43) -----------------------------
def pagerank(adj_matrix, config=None) -> PageRankResult:
    raw = _PageRankResultC()
    lib().spmv_c_pagerank(adj_matrix, byref(config) if config is not None else None, byref(raw))
    if not raw.ranks:
        return PageRankResult(None, 0, 0.0, False)
    n = adj_matrix.contents.num_rows
    if n < (1 << 18):
        ranks = np.ctypeslib.as_array(raw.ranks, shape=(n,)).copy() if n > 0 else np.empty(0, np.float32)
        lib().spmv_c_pagerank_free(byref(raw))
    else:
        # large result: a view of the library's (pinned, pooled) array, handed back by pagerank_free when the
        # numpy array is collected — no 4n-byte host copy
        ranks = np.ctypeslib.as_array(raw.ranks, shape=(n,))
        import weakref
        weakref.finalize(ranks, lambda held=raw: lib().spmv_c_pagerank_free(byref(held)))
    return PageRankResult(ranks, raw.iterations, float(raw.final_residual), bool(raw.converged))


def pagerank_multi_gpu(adj_matrix, config=None, num_gpus=1) -> PageRankResult:
    """include/spmv/pagerank.h extension: the row-sharded single-process RCCL loop over `num_gpus` devices."""
    raw = _PageRankResultC()
    lib().spmv_c_pagerank_multi_gpu(adj_matrix, byref(config) if config is not None else None, num_gpus, byref(raw))
    if not raw.ranks:
        return PageRankResult(None, 0, 0.0, False)
    n = adj_matrix.contents.num_rows
    ranks = np.ctypeslib.as_array(raw.ranks, shape=(n,)).copy()
    lib().spmv_c_pagerank_free(byref(raw))
    return PageRankResult(ranks, raw.iterations, float(raw.final_residual), bool(raw.converged))


def pagerank_top_k(result: PageRankResult, num_nodes: int, k: int):
    """[(node_id, rank)] of the k best-ranked nodes, descending."""
    if result is None or result.ranks is None or k <= 0 or num_nodes <= 0:
        return []
    ranks = np.ascontiguousarray(result.ranks, dtype=np.float32)
    raw = _PageRankResultC()
    raw.ranks = ranks.ctypes.data_as(POINTER(c_float))
    keep = min(k, num_nodes)
    nodes = (TopKNode * keep)()
    lib().spmv_c_pagerank_top_k(byref(raw), num_nodes, k, nodes)
    return [(n.node_id, float(n.rank)) for n in nodes]


from . import synth  # noqa: E402,F401  (numpy twin of the device generators)
