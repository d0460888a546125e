"""CPU-side tests of libspmv_amd.so: the C ABI loads without a GPU and exports every
symbol include/spmv_c.h declares; host-side containers, serialisers, statistics,
selector and the host SpMV path reproduce the reference's outputs bit for bit
(golden fixture made from the compiled reference).  No device compute here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT


@pytest.fixture(scope="module")
def golden():
    data = np.load(os.path.join(GOLDEN_DIR, "ref_cases.npz"), allow_pickle=False)
    return data, [str(n) for n in data["case_names"]]


def test_library_exports_every_declared_symbol(spmv):
    header = open(os.path.join(ROOT, "include", "spmv_c.h")).read()
    declared = set(re.findall(r"\b(spmv_c_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 60
    handle = ctypes.CDLL(spmv.LIB_PATH)
    missing = [name for name in sorted(declared) if not hasattr(handle, name)]
    assert not missing, missing
    assert declared == set(spmv.EXPORTED_SYMBOLS)      # the Python mirror binds all of them


def test_struct_sizes_match_reference_abi(spmv):
    """sizeof on x86-64 SysV"""
    assert ctypes.sizeof(spmv.CSRMatrix) == 72 and spmv.CSRMatrix.d_values.offset == 40
    assert spmv.CSRMatrix.owns_host_memory.offset == 64
    assert ctypes.sizeof(spmv.ELLMatrix) == 56
    assert ctypes.sizeof(spmv.SpMVConfig) == 12 and ctypes.sizeof(spmv.SpMVResult) == 24
    assert ctypes.sizeof(spmv.CSRStats) == 16 and ctypes.sizeof(spmv.PageRankConfig) == 12
    assert ctypes.sizeof(spmv._PageRankResultC) == 24 and ctypes.sizeof(spmv.TopKNode) == 8
    assert ctypes.sizeof(spmv.BandwidthMetrics) == 12


def test_error_strings(spmv):
    """reference tests/test_common.cpp:8-18"""
    E = spmv.SpMVError
    assert spmv.spmv_error_string(E.SUCCESS) == "Success"
    assert spmv.spmv_error_string(E.INVALID_DIMENSION) == "Invalid matrix/vector dimension"
    assert spmv.spmv_error_string(E.CUDA_MALLOC) == "CUDA memory allocation failed"
    assert spmv.spmv_error_string(E.CUDA_MEMCPY) == "CUDA memory copy failed"
    assert spmv.spmv_error_string(E.KERNEL_LAUNCH) == "CUDA kernel launch failed"
    assert spmv.spmv_error_string(E.INVALID_FORMAT) == "Invalid sparse matrix format"
    assert spmv.spmv_error_string(E.FILE_IO) == "File I/O error"
    assert spmv.spmv_error_string(E.OUT_OF_MEMORY) == "Out of memory"
    assert spmv.spmv_error_string(E.INVALID_ARGUMENT) == "Invalid argument"
    assert spmv.spmv_error_string(-99) == "Unknown error"


def test_defaults(spmv):
    c = spmv.SpMVConfig()
    assert (c.kernel_type, c.block_size, c.use_texture) == (spmv.SpMVConfig.SCALAR_CSR, 256, 0)
    p = spmv.PageRankConfig()
    assert p.damping_factor == pytest.approx(0.85) and p.tolerance == pytest.approx(1e-6) and p.max_iterations == 100


def test_cuda_buffer_host_side_semantics(spmv):
    """reference tests/test_common.cpp:21-98 — the parts that need no device"""
    b = spmv.CudaBuffer()
    assert b.get() is None and b.size() == 0 and b.empty()
    z = spmv.CudaBuffer(0)
    assert z.get() is None and z.empty()
    with pytest.raises(RuntimeError, match="Copy size exceeds buffer size"):
        z.copyFromHost(np.zeros(4, np.float32), 4)
    with pytest.raises(RuntimeError, match="Copy size exceeds buffer size"):
        z.copyToHost(1)
    z.release()
    assert z.size() == 0


def test_csr_container_against_reference_outputs(spmv, golden, tmp_path):
    data, names = golden
    for n in names:
        dense = data[f"{n}/dense"]
        rows, cols = dense.shape
        A = spmv.csr_create(0, 0, 0)
        assert spmv.csr_from_dense(A, dense, rows, cols) == int(data[f"{n}/csr_from_dense_status"][0])
        rp, ci, va = spmv.csr_host_arrays(A)
        np.testing.assert_array_equal(rp, data[f"{n}/csr_row_ptrs"], err_msg=n)       # bit-exact indexing
        np.testing.assert_array_equal(ci, data[f"{n}/csr_col_indices"], err_msg=n)
        np.testing.assert_array_equal(va.view(np.uint32), data[f"{n}/csr_values"].view(np.uint32), err_msg=n)
        m = A.contents
        assert [m.num_rows, m.num_cols, m.nnz] == list(data[f"{n}/csr_shape"])

        np.testing.assert_array_equal(spmv.csr_to_dense(A).reshape(-1), data[f"{n}/csr_to_dense"])
        diag = [spmv.csr_get_element(A, i, i) for i in range(min(rows, cols))]
        np.testing.assert_array_equal(np.array(diag, np.float32), data[f"{n}/csr_get_diag"])

        st = spmv.csr_compute_stats(A)
        ref = data[f"{n}/csr_stats"]
        assert (np.float32(st.avg_nnz_per_row), st.max_nnz_per_row, st.min_nnz_per_row, np.float32(st.skewness)) == \
               (ref[0], int(ref[1]), int(ref[2]), ref[3]), n
        cfg = spmv.spmv_auto_config(A)
        assert [cfg.kernel_type, cfg.block_size, cfg.use_texture] == list(data[f"{n}/auto_config"]), n

        # host SpMV path of the library (API parity with the reference's CPU path)
        y = spmv.spmv_cpu_csr(A, data[f"{n}/x"])
        np.testing.assert_array_equal(y.view(np.uint32), data[f"{n}/y_csr"].view(np.uint32), err_msg=n)

        # on-disk format is byte-identical to the reference's file
        path = tmp_path / f"{n}.csr"
        assert spmv.csr_serialize(A, str(path)) == 0
        assert path.read_bytes() == data[f"{n}/csr_file"].tobytes(), n
        B = spmv.csr_create(0, 0, 0)
        assert spmv.csr_deserialize(B, str(path)) == 0
        for a, b in zip(spmv.csr_host_arrays(A), spmv.csr_host_arrays(B)):
            np.testing.assert_array_equal(a, b)
        spmv.csr_destroy(B)
        spmv.csr_destroy(A)


def test_ell_container_against_reference_outputs(spmv, golden, tmp_path):
    data, names = golden
    for n in names:
        dense = data[f"{n}/dense"]
        rows, cols = dense.shape
        A = spmv.csr_create(0, 0, 0)
        spmv.csr_from_dense(A, dense, rows, cols)
        E = spmv.ell_create(0, 0, 0)
        assert spmv.ell_from_csr(E, A) == 0
        e = E.contents
        assert [e.num_rows, e.num_cols, e.max_nnz_per_row] == list(data[f"{n}/ell_shape"]), n
        ecols, evals = spmv.ell_host_arrays(E)
        np.testing.assert_array_equal(ecols, data[f"{n}/ell_col_indices"], err_msg=n)
        np.testing.assert_array_equal(evals.view(np.uint32), data[f"{n}/ell_values"].view(np.uint32), err_msg=n)

        E2 = spmv.ell_create(0, 0, 0)
        assert spmv.ell_from_dense(E2, dense, rows, cols) == 0
        c2, v2 = spmv.ell_host_arrays(E2)
        np.testing.assert_array_equal(c2, data[f"{n}/ell_dense_col_indices"], err_msg=n)
        np.testing.assert_array_equal(v2.view(np.uint32), data[f"{n}/ell_dense_values"].view(np.uint32))

        y = spmv.spmv_cpu_ell(E, data[f"{n}/x"])
        np.testing.assert_array_equal(y.view(np.uint32), data[f"{n}/y_ell"].view(np.uint32), err_msg=n)
        np.testing.assert_array_equal(spmv.ell_to_dense(E), dense)
        # padding slots are (-1, 0.0f) and slot (row, k) sits at k*rows + row (tests/test_ell.cpp:48-108)
        assert ((ecols >= 0) | (evals == 0)).all()
        if e.max_nnz_per_row:
            assert spmv.ell_index(rows - 1, e.max_nnz_per_row - 1, rows) == rows * e.max_nnz_per_row - 1

        path = tmp_path / f"{n}.ell"
        assert spmv.ell_serialize(E, str(path)) == 0
        assert path.read_bytes() == data[f"{n}/ell_file"].tobytes(), n
        E3 = spmv.ell_create(0, 0, 0)
        assert spmv.ell_deserialize(E3, str(path)) == 0
        for a, b in zip(spmv.ell_host_arrays(E), spmv.ell_host_arrays(E3)):
            np.testing.assert_array_equal(a, b)
        for h in (E, E2, E3):
            spmv.ell_destroy(h)
        spmv.csr_destroy(A)


def test_argument_validation_matches_reference(spmv, tmp_path):
    """create with negative sizes -> null; null/empty inputs -> INVALID_ARGUMENT; bad file -> FILE_IO;
    out-of-range element query -> 0 (reference src/csr_matrix.cpp:10-13,51-53,117-119,237-239)."""
    E = spmv.SpMVError
    assert spmv.csr_create(-1, 2, 3) is None and spmv.ell_create(1, -2, 3) is None
    A = spmv.csr_create(0, 0, 0)
    assert spmv.csr_from_dense(A, None, 3, 3) == E.INVALID_ARGUMENT
    assert spmv.csr_from_dense(A, np.ones((2, 2), np.float32), 0, 2) == E.INVALID_ARGUMENT
    assert spmv.csr_deserialize(A, str(tmp_path / "missing.bin")) == E.FILE_IO
    (tmp_path / "short.bin").write_bytes(b"\x01\x00\x00\x00")
    assert spmv.csr_deserialize(A, str(tmp_path / "short.bin")) == E.FILE_IO
    assert spmv.csr_serialize(A, str(tmp_path / "no_such_dir" / "x.bin")) == E.FILE_IO
    spmv.csr_from_dense(A, np.eye(3, dtype=np.float32), 3, 3)
    assert spmv.csr_get_element(A, 5, 0) == 0.0 and spmv.csr_get_element(A, 0, -1) == 0.0
    assert spmv.csr_get_element(A, 1, 1) == 1.0
    assert spmv.csr_from_gpu(A) == E.INVALID_ARGUMENT          # nothing on the device
    spmv.csr_destroy(A)
    assert spmv.spmv_validate_dimensions(5, 5) and not spmv.spmv_validate_dimensions(5, 6)


def test_selector_properties(spmv, oracle):
    """reference tests/test_kernel_selector.cpp:17-137 (P11 + threshold units)."""
    rng = np.random.default_rng(42)
    for _ in range(60):
        rows, cols = int(rng.integers(10, 200)), int(rng.integers(10, 200))
        dense = np.where(rng.random((rows, cols)) < rng.uniform(0.01, 0.5), 1.0, 0.0).astype(np.float32)
        A = spmv.csr_create(0, 0, 0)
        spmv.csr_from_dense(A, dense, rows, cols)
        cfg = spmv.spmv_auto_config(A)
        assert 32 <= cfg.block_size <= 1024 and cfg.block_size % 32 == 0
        assert cfg.kernel_type in (0, 1, 2)
        st = spmv.csr_compute_stats(A)
        if st.avg_nnz_per_row < 4.0:
            assert cfg.kernel_type == spmv.SpMVConfig.SCALAR_CSR
        elif st.skewness < 10.0:
            assert cfg.kernel_type == spmv.SpMVConfig.VECTOR_CSR
        spmv.csr_destroy(A)
    dense = np.zeros((100, 1000), np.float32)
    dense[:, ::100] = 1.0
    A = spmv.csr_create(0, 0, 0)
    spmv.csr_from_dense(A, np.zeros((10, 10001), np.float32) + np.eye(10, 10001, dtype=np.float32), 10, 10001)
    assert spmv.spmv_auto_config(A).use_texture == 1           # cols > 10000
    spmv.csr_destroy(A)


def test_bandwidth_model_host_side(spmv, oracle):
    """reference tests/test_bandwidth.cu:100-113: zero elapsed => zero metrics; byte model = oracle's."""
    A = spmv.csr_create(0, 0, 0)
    spmv.csr_from_dense(A, np.eye(8, dtype=np.float32), 8, 8)
    z = spmv.compute_bandwidth_csr(A, 0.0)
    assert (z.achieved_bandwidth_gb_s, z.theoretical_bandwidth_gb_s, z.efficiency) == (0.0, 0.0, 0.0)
    m = spmv.compute_bandwidth_csr(A, 1.0)
    assert m.achieved_bandwidth_gb_s == pytest.approx(oracle.bytes_csr(8, 8, 8) / 1e9 / 1e-3, rel=1e-6)
    assert 0 < m.theoretical_bandwidth_gb_s < 10000 and 0 <= m.efficiency <= 1
    spmv.csr_destroy(A)


def test_product_never_imports_the_oracle():
    """The library and its Python mirror must not route through oracle/ (tier rule 3)."""
    pkg = os.path.join(ROOT, "gpu-spmv_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "spmv_oracle" not in text and "liboracle" not in text, f


def test_container_properties_random(spmv, tmp_path):
    """Property loops of the reference's container tests on random dense matrices:
    P1 dense->CSR->dense round trip (tests/test_csr.cpp:18-43), P2 csr_get_element (:47-76),
    P3 CSR serialise round trip (:80-127), P4 ELL round trip, P5 padding = (-1, 0.0f)
    (tests/test_ell.cpp:48-80), P6 ell_index == k*rows+row (:84-108), P7 ELL serialise,
    ell_from_csr == ell_from_dense."""
    rng = np.random.default_rng(42)
    for it in range(40):
        rows, cols = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        dense = np.where(rng.random((rows, cols)) < rng.uniform(0.0, 0.5), rng.uniform(-10, 10, (rows, cols)), 0)
        dense = dense.astype(np.float32)
        A = spmv.csr_create(0, 0, 0)
        assert spmv.csr_from_dense(A, dense, rows, cols) == 0
        np.testing.assert_array_equal(spmv.csr_to_dense(A), dense)                                   # P1
        for _ in range(10):                                                                          # P2
            r, c = int(rng.integers(0, rows)), int(rng.integers(0, cols))
            assert spmv.csr_get_element(A, r, c) == dense[r, c]
        rp, ci, va = spmv.csr_host_arrays(A)
        assert rp[0] == 0 and rp[-1] == A.contents.nnz == np.count_nonzero(dense)
        assert all((np.diff(ci[rp[r]:rp[r + 1]]) > 0).all() for r in range(rows))                    # ascending columns
        path = str(tmp_path / "m.csr")
        assert spmv.csr_serialize(A, path) == 0                                                      # P3
        B = spmv.csr_create(0, 0, 0)
        assert spmv.csr_deserialize(B, path) == 0
        np.testing.assert_array_equal(spmv.csr_to_dense(B), dense)

        E = spmv.ell_create(0, 0, 0)
        assert spmv.ell_from_dense(E, dense, rows, cols) == 0
        np.testing.assert_array_equal(spmv.ell_to_dense(E), dense)                                   # P4
        k = E.contents.max_nnz_per_row
        assert k == int((dense != 0).sum(axis=1).max())
        ecols, evals = spmv.ell_host_arrays(E)
        if k:
            pad = ecols < 0
            assert (ecols[pad] == -1).all() and (evals[pad] == 0).all()                               # P5
            grid = ecols.reshape(k, rows)
            for r in range(rows):                                                                    # P6: column-major slots
                want = np.flatnonzero(dense[r])
                got = grid[:, r][grid[:, r] >= 0]
                np.testing.assert_array_equal(got, want)
                assert spmv.ell_index(r, k - 1, rows) == (k - 1) * rows + r
        for _ in range(10):
            r, c = int(rng.integers(0, rows)), int(rng.integers(0, cols))
            assert spmv.ell_get_element(E, r, c) == dense[r, c]
        E2 = spmv.ell_create(0, 0, 0)
        assert spmv.ell_from_csr(E2, A) == 0
        for a, b in zip(spmv.ell_host_arrays(E), spmv.ell_host_arrays(E2)):
            np.testing.assert_array_equal(a, b)
        epath = str(tmp_path / "m.ell")
        assert spmv.ell_serialize(E, epath) == 0                                                     # P7
        E3 = spmv.ell_create(0, 0, 0)
        assert spmv.ell_deserialize(E3, epath) == 0
        np.testing.assert_array_equal(spmv.ell_to_dense(E3), dense)
        for h in (E, E2, E3):
            spmv.ell_destroy(h)
        spmv.csr_destroy(A)
        spmv.csr_destroy(B)


def test_tiled_engine_shape_rules(spmv):
    """Host logic of the LDS-tiled engine (csrc/tiled.hip choose_shape / eligibility), no GPU needed:
    which matrices it takes, and that strips / tiles stay inside the LDS budgets and index widths."""
    takes, w, r = spmv.tiled_shape(10_000_000, 10_000_000, 160_000_000)          # BASELINE config 5
    assert takes and w in (4096, 8192, 16384, 32768) and r % 64 == 0 and 1024 <= r <= 9984
    tiles = (10_000_000 + r - 1) // r
    assert 1000 <= tiles <= 1024                                                   # two full rounds of the 512 resident tiles, no tail
    assert 160_000_000 / (((10_000_000 + w - 1) // w) * ((10_000_000 + r - 1) // r)) >= 100   # long enough runs
    takes, w, r = spmv.tiled_shape(1_250_000, 10_000_032, 20_000_000)              # a 1/8 row shard of it
    assert takes and w == 32768 and 450 <= (1_250_000 + r - 1) // r <= 512          # one full round
    assert spmv.tiled_shape(1_000_000, 1_000_000, 16_000_000)[0]                   # config 2
    assert not spmv.tiled_shape(1000, 1000, 8000)[0]                               # config 1: tiny
    assert not spmv.tiled_shape(2_000_000, 30_000, 32_000_000)[0]                  # x fits LDS: other kernel
    assert not spmv.tiled_shape(100_000, 1_000_000, 500_000)[0]                    # too few entries
    assert not spmv.tiled_shape(2_000_000_000, 2_000_000_000, 2_000_000_000)[0]    # cell table would be too large
    rng = np.random.default_rng(0)
    for _ in range(200):
        rows, cols = int(rng.integers(1, 50_000_000)), int(rng.integers(1, 50_000_000))
        nnz = int(rng.integers(1, 2_000_000_000))
        takes, w, r = spmv.tiled_shape(rows, cols, nnz)
        assert w in (4096, 8192, 16384, 32768) and r % 64 == 0 and 1024 <= r <= 9984   # u16 local indices, LDS fits
        tiles = (rows + r - 1) // r
        assert tiles < 512 or tiles % 512 == 0 or tiles % 512 > 450                  # whole rounds of resident tiles


def test_the_one_debug_variable_overrides_thresholds_and_shape(spmv, monkeypatch):
    """SPMV_DEBUG="key=value,key,..." (csrc/internal.h) is the library's only tuning / debugging variable and is read at
    every use: thresholds and shape of the tiled engine follow it from one call to the next; keys that are prefixes of
    other keys, unknown keys and malformed values are harmless."""
    small = (5_000, 5_000, 40_000)
    assert not spmv.tiled_shape(*small)[0]
    monkeypatch.setenv("SPMV_DEBUG", "min_cols=1,min_nnz=1")
    assert spmv.tiled_shape(*small)[0]
    monkeypatch.setenv("SPMV_DEBUG", "rank=plain,min_cols=1,min_nnz=1,strip=4096,tile=1024,nonsense,x=")
    assert spmv.tiled_shape(*small) == (True, 4096, 1024)
    monkeypatch.setenv("SPMV_DEBUG", "min_cols=1,min_nnz=1,strip=5000,tile=1000")          # not a power of two / not a multiple of 64: ignored
    takes, w, r = spmv.tiled_shape(*small)
    assert takes and w in (4096, 8192, 16384, 32768) and r % 64 == 0
    monkeypatch.setenv("SPMV_DEBUG", "tiles=7,stripe=3,min_colsx=1,min_nnz=1")             # longer keys do not match shorter ones
    assert not spmv.tiled_shape(*small)[0]
    monkeypatch.setenv("SPMV_DEBUG", "min_nnz=999999999,min_cols=1")
    assert not spmv.tiled_shape(10_000_000, 10_000_000, 160_000_000)[0]
    monkeypatch.delenv("SPMV_DEBUG")
    assert spmv.tiled_shape(10_000_000, 10_000_000, 160_000_000)[0] and not spmv.tiled_shape(*small)[0]


def test_the_library_reads_few_environment_variables(spmv):
    """at most ten SPMV_* variables in the product sources, every one of them listed in INTEGRATION.md."""
    import glob
    import re
    names = set()
    for path in glob.glob(os.path.join(ROOT, "gpu-spmv_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "**", "*.h"), recursive=True):
        names.update(re.findall(r'getenv\("(SPMV_[A-Z0-9_]+)"\)', open(path, errors="replace").read()))
    assert 0 < len(names) <= 10, sorted(names)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in names:
        assert name in doc, name


def test_shard_engine_rejects_bad_arguments_without_touching_a_gpu(spmv):
    """spmv_c_pr_shard_create validates before any device call: null matrix / mask, negative offsets and a
    node count wider than the matrix header all give NULL (include/spmv_c.h)."""
    import ctypes
    lib = spmv.lib()
    assert not lib.spmv_c_pr_shard_create(None, 0, 10, None)
    A = spmv.csr_create(4, 4, 0)
    mask = (ctypes.c_uint8 * 4)()
    assert not lib.spmv_c_pr_shard_create(A, 0, 4, None)                       # no mask
    assert not lib.spmv_c_pr_shard_create(A, -1, 4, mask)                      # negative offset
    assert not lib.spmv_c_pr_shard_create(A, 0, 5, mask)                       # more nodes than columns
    assert not lib.spmv_c_pr_shard_create(A, 1, 4, mask)                       # rows would run past the vector
    assert not lib.spmv_c_pr_shard_create(A, 0, 4, mask)                       # rows but no device arrays
    spmv.csr_destroy(A)


def test_native_multi_gpu_shard_bounds_on_the_host(spmv):
    """tests/cpp/multi_gpu_pagerank (bounds mode, no GPU): pagerank_shard_bounds — binary search on row_ptrs for
    equal nnz, SURVEY.md §8(e) — through the C++ API; and the same through the C ABI."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "multi_gpu_pagerank")
    if not os.path.exists(exe):
        pytest.skip("tests/cpp/bin not built (run __graft_entry__.build())")
    out = subprocess.run([exe, "bounds"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "all checks passed (bounds only)" in out.stdout, out.stdout + out.stderr
    import ctypes
    rp = np.array([0, 10, 10, 11, 12, 40, 41], dtype=np.int32)
    bounds = np.zeros(3, dtype=np.int32)
    lib = spmv.lib()
    assert lib.spmv_c_pagerank_shard_bounds(rp.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), 6, 2,
                                            bounds.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))) == 0
    assert bounds.tolist() == [0, 4, 6]


def test_shard_bounds_and_layouts_agree_on_random_inputs(spmv):
    """Property test (hypothesis): the two loops cut a graph the same way — the C++ `pagerank_shard_bounds`
    (native multi-GPU loop) and `Layout.equal_nnz_bounds` (one process per GPU) — and every layout the Python
    loop can build (any world, any block count, any alignment, with or without forced exchange) numbers the
    nodes injectively inside the vector, keeps the tails clear of them, and agrees with the engine's RowMap."""
    import ctypes
    import importlib
    hyp = pytest.importorskip("hypothesis")
    st = importlib.import_module("hypothesis.strategies")
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
    lib = spmv.lib()

    @hyp.settings(max_examples=150, deadline=None)
    @hyp.given(lens=st.lists(st.integers(0, 50), min_size=1, max_size=200), world=st.integers(1, 9),
               chunks=st.integers(1, 5), align=st.sampled_from([None, 2, 4, 64]), forced=st.booleans())
    def check(lens, world, chunks, align, forced):
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        n = len(lens)
        native = np.zeros(world + 1, dtype=np.int32)
        assert lib.spmv_c_pagerank_shard_bounds(rp.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n, world,
                                                native.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))) == 0
        bounds = prd.Layout.equal_nnz_bounds(rp, world)
        assert bounds.tolist() == native.tolist()
        exchange = True if (forced and world == 1) else None
        lays = [prd.Layout(n, world, r, bounds=bounds, chunks=chunks, align=align, exchange=exchange) for r in range(world)]
        lay = lays[0]
        pos = lay.positions()
        assert len(set(pos.tolist())) == n and (n == 0 or (0 <= pos.min() and pos.max() < lay.padded))
        np.testing.assert_array_equal(lay.remap_columns(np.arange(n, dtype=np.int32)), pos)
        taken = set(pos.tolist())
        for r, lr in enumerate(lays):
            mine = lr.local_positions()
            np.testing.assert_array_equal(mine, pos[lr.row_begin:lr.row_end])
            base, piece, block = lr.row_map()
            i = np.arange(lr.local_rows)
            np.testing.assert_array_equal(mine, base + i if piece == 0x7FFFFFFF else base + (i // piece) * block + i % piece)
            if lr.exchange:
                t = lr.tail_slice()
                assert t.start % 2 == 0 and t.stop <= lr.padded and not (set(range(t.start, t.stop)) & taken)
                assert lr.block_slice(lr.chunks - 1).start <= t.start and t.stop == lr.piece_slice(lr.chunks - 1).stop

    check()
