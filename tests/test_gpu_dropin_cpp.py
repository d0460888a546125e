"""Runs tests/cpp/dropin_smoke (a C++ caller written like the reference's gtest files,
compiled with plain g++ by __graft_entry__.build()) on the GPU: the C++ boundary
(namespace spmv, CudaBuffer, struct fields) is a source-level drop-in."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_benchmark_executable_runs(gpu):
    """benchmarks/spmv_benchmark.cpp: the working counterpart of the reference's benchmarks/main.cu."""
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "spmv_benchmark")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe, "200000", "300000", "8"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    for needle in ("scalar CSR", "vector CSR", "merge path", "ELL", "speed-up GPU/CPU", "\"gflops\"",
                   "iterations", "% of peak", "auto config"):
        assert needle in out.stdout, needle


def test_cpp_dropin_caller(gpu):
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "dropin_smoke")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "all checks passed" in out.stdout, out.stdout + out.stderr


def test_reference_gtest_cases_restated_in_cpp(gpu):
    """tests/cpp/reference_suite.cpp: the 48 cases of the reference's eight gtest files, same names, compiled
    with plain g++ against the drop-in headers (source-level compatibility of the C++ boundary)."""
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "reference_suite")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "all reference cases passed" in out.stdout, out.stdout[-4000:] + out.stderr[-2000:]
    assert out.stdout.count("[  OK  ]") == 48, out.stdout[-2000:]


def test_native_multi_gpu_pagerank_entry_point(gpu):
    """tests/cpp/multi_gpu_pagerank.cpp: pagerank_multi_gpu(adj, config, 1) — the single-process RCCL host loop
    behind the reference's API (include/spmv/pagerank.h extension) — equals pagerank() on a uniform and on a
    power-law graph, both without and with the RCCL all-gather in the loop (one device: SPMV_MULTI_GPU=force_rccl),
    and the equal-nnz shard boundaries are right."""
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "multi_gpu_pagerank")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe, "run", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "all checks passed (bounds + run)" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
