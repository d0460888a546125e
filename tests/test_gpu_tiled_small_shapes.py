"""The LDS-tiled engine normally takes only large matrices (> 32768 columns, >= 1 M entries).  This test
lowers the thresholds (environment, read once per process => a worker process) and pushes ~150 SMALL matrices
of awkward shapes through it — single row / column, sizes straddling the strip and tile sizes, ragged and
empty rows, rows far beyond the long-row limit, value-folded columns, ELL sources — against the CPU oracle."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_small_and_awkward_shapes_through_the_tiled_engine(gpu):
    # pr_plan_after=0: pagerank() builds the tiled plan before its first step (by default a matrix
    # without a plan starts on the direct kernel, see test_pagerank_switches_to_the_tiled_engine_mid_run)
    env = dict(os.environ, SPMV_DEBUG="min_cols=1,min_nnz=1,pr_plan_after=0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tiled_small_shapes_worker.py")],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "tiled small shapes:" in out.stdout
    assert "reference fixtures through the tiled engine: 13 cases" in out.stdout
    # cells with hand-picked slot counts (every boundary case of phase 2's passes), shape forced to 4096 x 1024
    env.update(SPMV_DEBUG="min_cols=1,min_nnz=1,pr_plan_after=0,strip=4096,tile=1024")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tiled_small_shapes_worker.py"), "patterns"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "run-length patterns:" in out.stdout
