"""Worker of tests/test_gpu_tiled_small_shapes.py: run with SPMV_DEBUG=min_cols=1,min_nnz=1 set so
that SMALL matrices go through the LDS-tiled engine, and compare every case with the CPU oracle.
(The thresholds are read once per process, hence the separate process.)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
spmv = importlib.import_module("gpu-spmv_amd")
oracle = importlib.import_module("oracle")
from conftest import reorder_err  # noqa: E402


def run_case(rng, rows, cols, lens, fold, ell):
    lens = np.minimum(lens, cols).astype(np.int64)
    # distinct ascending columns per row (a draw with repeats, de-duplicated: rows may come out a little shorter)
    per_row = [np.unique(rng.integers(0, cols, size=n)) if n < cols else np.arange(cols) for n in lens]
    lens = np.array([r.size for r in per_row], dtype=np.int64)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = (np.concatenate(per_row) if per_row else np.empty(0)).astype(np.int32)
    if fold:      # every column one value: the plan folds the values away
        weight = rng.uniform(0.1, 2.0, size=cols).astype(np.float32)
        va = weight[ci]
    else:
        va = rng.uniform(-1, 1, size=ci.size).astype(np.float32)
    x = rng.uniform(-1, 1, size=cols).astype(np.float32)
    want = oracle.spmv_csr(rp, ci, va, x)
    d_x, d_y = spmv.CudaBuffer(cols), spmv.CudaBuffer(rows)
    d_x.copyFromHost(x, cols)
    if ell:
        import ctypes
        kk, ecols, evals = oracle.ell_from_csr(rp, ci, va)
        E = spmv.ell_create(rows, cols, kk)
        ctypes.memmove(E.contents.col_indices, ecols.ctypes.data, ecols.nbytes)
        ctypes.memmove(E.contents.values, evals.ctypes.data, evals.nbytes)
        assert spmv.ell_to_gpu(E) == 0
        cfg = spmv.SpMVConfig(kernel_type=spmv.SpMVConfig.ELL_KERNEL, use_texture=True)
        for _ in range(2):
            assert spmv.spmv_ell(E, d_x, d_y, cfg, cols).error_code == 0
        got = d_y.copyToHost(rows)
        spmv.ell_destroy(E)
        return reorder_err(rp, ci, va, x, want, got), True
    A = spmv.csr_from_arrays(rows, cols, rp, ci, va)
    assert spmv.csr_to_gpu(A) == 0
    worst, planned = 0.0, False
    for kernel in (1, 2):
        for _ in range(2):          # the second call reuses the plan (and meets whatever the first left behind)
            res = spmv.spmv_csr(A, d_x, d_y, spmv.SpMVConfig(kernel_type=kernel, use_texture=True), cols)
            assert res.error_code == 0
            worst = max(worst, reorder_err(rp, ci, va, x, want, d_y.copyToHost(rows)))
        planned = planned or bool(spmv.csr_has_tiled_plan(A))
    info = spmv.csr_tiled_info(A)
    if fold and info is not None and info["entries_in_cells"] > 0:
        assert info["values_folded"], (rows, cols, info)
    spmv.csr_destroy(A)
    return worst, planned


def main():
    spmv.require_gpu()
    rng = np.random.default_rng(2024)
    shapes = [(1, 1), (1, 5000), (3, 4097), (64, 4096), (65, 8193), (1000, 70_000), (4999, 33_000),
              (20_000, 100), (9793, 9793), (130, 200_000)]
    cases = planned_cases = 0
    worst = 0.0
    for rows, cols in shapes:
        for kind in ("uniform", "ragged", "long", "empty"):
            if kind == "uniform":
                lens = np.full(rows, min(cols, 7))
            elif kind == "ragged":
                lens = rng.integers(0, min(cols, 40) + 1, size=rows)
            elif kind == "long":       # a few rows far beyond the long-row limit, many empty ones
                lens = rng.integers(0, 3, size=rows)
                lens[rng.integers(0, rows, size=min(rows, 3))] = min(cols, 5000)
            else:
                lens = np.zeros(rows, dtype=np.int64)
                lens[rows // 2] = min(cols, 3)
            for fold in (False, True):
                for ell in (False, True):
                    if ell and (kind == "long" or rows * int(max(lens.max(), 1)) > 4_000_000):
                        continue
                    err, planned = run_case(rng, rows, cols, lens, fold, ell)
                    assert err <= 1e-5, (rows, cols, kind, fold, ell, err)
                    worst = max(worst, err)
                    cases += 1
                    planned_cases += int(planned)
    assert planned_cases >= cases // 2, (planned_cases, cases)     # the thresholds really were lowered

    # PageRank on small graphs through the same engine (fused update in phase 2, seeds of the hub rows,
    # dangling mask read off the folded column weights), twice per graph, against the oracle
    for n, k in ((50, 3), (1000, 8), (5000, 12), (9793, 5)):
        per_row = [np.unique(rng.integers(0, n, size=k)) for _ in range(n)]
        per_row[n // 3] = np.arange(0, n, 2)                       # a hub: far beyond the long-row limit
        dangling = np.array([1, n // 2, n - 1])
        per_row = [r[~np.isin(r, dangling)] for r in per_row]
        lens = np.array([r.size for r in per_row], dtype=np.int64)
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        ci = np.concatenate(per_row).astype(np.int32)
        va = spmv.synth.column_stochastic_values(ci, n)
        want, iters, _, conv = oracle.pagerank(rp, ci, va, num_cols=n, wide_sums=True)
        A = spmv.csr_from_arrays(n, n, rp, ci, va)
        assert spmv.csr_to_gpu(A) == 0
        for _ in range(2):
            r = spmv.pagerank(A, spmv.PageRankConfig(0.85, 1e-6, 100))
            assert spmv.csr_has_tiled_plan(A)
            assert r.converged == conv and abs(r.iterations - iters) <= 1, (n, r.iterations, iters)
            ref = want
            if r.iterations != iters:      # compare at equal iteration counts (tolerance 0: the last computed vector)
                k = min(r.iterations, iters)
                r = spmv.pagerank(A, spmv.PageRankConfig(0.85, 0.0, k))
                ref, *_ = oracle.pagerank(rp, ci, va, num_cols=n, tolerance=0.0, max_iterations=k, wide_sums=True)
            rel = float(np.max(np.abs(r.ranks.astype(np.float64) - ref) / ref))
            assert rel <= 1e-5, (n, rel)
        spmv.csr_destroy(A)
        cases += 1
    print("tiled small shapes: %d cases, %d through the tiled engine, worst error %.3g" % (cases, planned_cases, worst))
    golden_fixtures()


def run_length_patterns():
    """Cells (strip x tile) whose slot counts are chosen one by one, so that phase 2's slot stream meets every boundary
    case: runs of 0 / 1 / 3 / 4 / 5 slots, runs just under / at / over one pass (252 / 256 / 260) and two (511 / 512 / 513), long
    trains of empty runs, runs longer than a tile has rows, > 64 runs per wavefront (more than one window of run records:
    1100 strips = 69 runs per wavefront) — against the oracle, through VECTOR_CSR and MERGE_PATH with use_texture."""
    import scipy.sparse as sp
    strips, tiles = 1100, 2
    w, r = 4096, 1024                                  # forced by the test's environment (SPMV_DEBUG=strip=4096,tile=1024)
    assert "strip=4096" in os.environ.get("SPMV_DEBUG", "") and "tile=1024" in os.environ.get("SPMV_DEBUG", "")
    rows, cols = tiles * r, strips * w
    base = [0, 0, 3, 4, 5, 252, 256, 260, 0, 1, 511, 512, 513, 64, 64, 64, 64, 7] + [0] * 53 + [1000, 2, 0, 0, 1500, 8, 248, 12]
    rng = np.random.default_rng(99)
    rr, cc = [], []
    for t in range(tiles):
        pattern = np.array([base[(s + 11 * t) % len(base)] for s in range(strips)])
        pattern[rng.integers(0, strips, size=40)] = rng.integers(1, 700, size=40)     # and some arbitrary ones
        for s in np.nonzero(pattern)[0]:
            n = int(pattern[s])
            i = np.arange(n)
            rr.append(t * r + (i * 7 + s) % r)                     # n <= r: distinct rows; beyond: the column moves on
            cc.append(s * w + ((i // r) * 5 + (i * 13) % 5 + (s % 3)) % w)
    rr, cc = np.concatenate(rr), np.concatenate(cc)
    keys = np.unique(rr.astype(np.int64) * cols + cc)              # distinct (row, column) pairs, row-major
    rr, cc = (keys // cols).astype(np.int64), (keys % cols).astype(np.int32)
    va = rng.uniform(-1, 1, size=keys.size).astype(np.float32)
    rp = np.concatenate([[0], np.cumsum(np.bincount(rr, minlength=rows))]).astype(np.int32)
    x = rng.uniform(-1, 1, size=cols).astype(np.float32)
    want = oracle.spmv_csr(rp, cc, va, x)
    A = spmv.csr_from_arrays(rows, cols, rp, cc, va)
    assert spmv.csr_to_gpu(A) == 0
    d_x, d_y = spmv.CudaBuffer(cols), spmv.CudaBuffer(rows)
    d_x.copyFromHost(x, cols)
    for kernel in (1, 2):
        for _ in range(2):
            res = spmv.spmv_csr(A, d_x, d_y, spmv.SpMVConfig(kernel_type=kernel, use_texture=True), cols)
            assert res.error_code == 0
            err = reorder_err(rp, cc, va, x, want, d_y.copyToHost(rows))
            assert err <= 1e-5, (kernel, err)
    info = spmv.csr_tiled_info(A)
    assert info and info["num_strips"] == strips and info["tile_rows"] == r, info
    spmv.csr_destroy(A)
    print("run-length patterns: %d entries in %d strips x %d tiles" % (keys.size, strips, tiles))


def golden_fixtures():
    """The 13 inputs of tests/golden/ref_cases.npz through the tiled engine (CSR as VECTOR_CSR and MERGE_PATH with
    use_texture, the ELL slabs with use_texture), against the y the REFERENCE's own spmv_cpu_csr / spmv_cpu_ell
    produced for them (/root/reference/src/spmv_cpu.cpp:6-32, recorded by tests/golden/make_golden.py)."""
    import ctypes
    data = np.load(os.path.join(ROOT, "tests", "golden", "ref_cases.npz"), allow_pickle=False)
    through_plan = 0
    for name in (str(n) for n in data["case_names"]):
        rp, ci, va = data[f"{name}/csr_row_ptrs"], data[f"{name}/csr_col_indices"], data[f"{name}/csr_values"]
        x, want_csr, want_ell = data[f"{name}/x"], data[f"{name}/y_csr"], data[f"{name}/y_ell"]
        rows, cols, nnz = (int(v) for v in data[f"{name}/csr_shape"])
        d_x, d_y = spmv.CudaBuffer(max(cols, 1)), spmv.CudaBuffer(max(rows, 1))
        d_x.copyFromHost(x, cols)
        A = spmv.csr_from_arrays(rows, cols, rp, ci, va)
        assert spmv.csr_to_gpu(A) == 0
        for kernel in (1, 2):
            res = spmv.spmv_csr(A, d_x, d_y, spmv.SpMVConfig(kernel_type=kernel, use_texture=True), cols)
            assert res.error_code == 0, (name, kernel, res.error_code)
            err = reorder_err(rp, ci, va, x, want_csr, d_y.copyToHost(rows))
            assert err <= 1e-5, (name, kernel, err)
        through_plan += int(bool(spmv.csr_has_tiled_plan(A)))
        spmv.csr_destroy(A)
        erows, ecols_n, kk = (int(v) for v in data[f"{name}/ell_shape"])
        E = spmv.ell_create(erows, ecols_n, kk)
        ecols, evals = data[f"{name}/ell_col_indices"], data[f"{name}/ell_values"]
        if ecols.size:
            ctypes.memmove(E.contents.col_indices, ecols.ctypes.data, ecols.nbytes)
            ctypes.memmove(E.contents.values, evals.ctypes.data, evals.nbytes)
        assert spmv.ell_to_gpu(E) == 0
        res = spmv.spmv_ell(E, d_x, d_y, spmv.SpMVConfig(kernel_type=spmv.SpMVConfig.ELL_KERNEL, use_texture=True), cols)
        assert res.error_code == 0, (name, "ell", res.error_code)
        err = reorder_err(rp, ci, va, x, want_ell, d_y.copyToHost(rows))
        assert err <= 1e-5, (name, "ell", err)
        spmv.ell_destroy(E)
    assert through_plan >= 8, through_plan          # the non-trivial cases did go through a plan
    print("reference fixtures through the tiled engine: 13 cases, %d with a plan" % through_plan)


if __name__ == "__main__":
    if sys.argv[1:] == ["patterns"]:
        spmv.require_gpu()
        run_length_patterns()
    else:
        main()
