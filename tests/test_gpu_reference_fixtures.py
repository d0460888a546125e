"""The HIP kernels on the reference-held fixtures THEMSELVES: the 13 inputs of
tests/golden/ref_cases.npz with the y the reference's own CPU path produced for them
(/root/reference/src/spmv_cpu.cpp:6-32 compiled by oracle/Makefile, recorded by tests/golden/make_golden.py; inputs in the
spirit of /root/reference/tests/test_spmv.cu:40-118,161-218).  SCALAR_CSR and the ELL kernel keep the CPU's summation
order: bit equality with the reference's bits.  VECTOR_CSR / MERGE_PATH reorder a row's sum: 1e-5 (conftest.reorder_err).
The same fixtures through the LDS-tiled engine: tests/tiled_small_shapes_worker.py (thresholds lowered)."""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, reorder_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    data = np.load(os.path.join(GOLDEN_DIR, "ref_cases.npz"), allow_pickle=False)
    return data, [str(n) for n in data["case_names"]]


def test_csr_kernels_on_the_reference_fixtures(gpu, golden):
    data, names = golden
    assert len(names) == 13
    for name in names:
        rp, ci, va = data[f"{name}/csr_row_ptrs"], data[f"{name}/csr_col_indices"], data[f"{name}/csr_values"]
        x, want = data[f"{name}/x"], data[f"{name}/y_csr"]
        rows, cols, nnz = (int(v) for v in data[f"{name}/csr_shape"])
        A = gpu.csr_from_arrays(rows, cols, rp, ci, va)
        assert gpu.csr_to_gpu(A) == 0
        d_x, d_y = gpu.CudaBuffer(cols), gpu.CudaBuffer(rows)
        d_x.copyFromHost(x, cols)
        for kernel, exact in ((0, True), (1, False), (2, False)):
            for use_texture in (False, True):
                d_y.copyFromHost(np.full(rows, np.float32(np.nan)), rows)         # every row must be written
                res = gpu.spmv_csr(A, d_x, d_y, gpu.SpMVConfig(kernel_type=kernel, use_texture=use_texture), cols)
                assert res.error_code == 0, (name, kernel, res.error_code)
                got = d_y.copyToHost(rows)
                if exact:
                    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32), err_msg=f"{name} scalar")
                else:
                    assert reorder_err(rp, ci, va, x, want, got) <= 1e-5, (name, kernel, use_texture)
        # the selector's choice for this matrix, too (what a reference caller gets without thinking)
        cfg = gpu.spmv_auto_config(A)
        assert gpu.spmv_csr(A, d_x, d_y, cfg, cols).error_code == 0
        assert reorder_err(rp, ci, va, x, want, d_y.copyToHost(rows)) <= 1e-5, (name, "auto")
        gpu.csr_destroy(A)


def test_ell_kernel_on_the_reference_fixtures(gpu, golden):
    data, names = golden
    for name in names:
        rows, cols, k = (int(v) for v in data[f"{name}/ell_shape"])
        ecols, evals = data[f"{name}/ell_col_indices"], data[f"{name}/ell_values"]
        x, want = data[f"{name}/x"], data[f"{name}/y_ell"]
        E = gpu.ell_create(rows, cols, k)
        if ecols.size:
            ctypes.memmove(E.contents.col_indices, ecols.ctypes.data, ecols.nbytes)
            ctypes.memmove(E.contents.values, evals.ctypes.data, evals.nbytes)
        assert gpu.ell_to_gpu(E) == 0
        d_x, d_y = gpu.CudaBuffer(cols), gpu.CudaBuffer(rows)
        d_x.copyFromHost(x, cols)
        d_y.copyFromHost(np.full(rows, np.float32(np.nan)), rows)
        res = gpu.spmv_ell(E, d_x, d_y, gpu.SpMVConfig(kernel_type=gpu.SpMVConfig.ELL_KERNEL), cols)
        assert res.error_code == 0, (name, res.error_code)
        np.testing.assert_array_equal(d_y.copyToHost(rows).view(np.uint32), want.view(np.uint32), err_msg=name)
        gpu.ell_destroy(E)
