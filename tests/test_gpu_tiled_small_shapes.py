"""The LDS-tiled engine normally takes only large matrices (>= 65536 columns, >= 1 M entries).  This test
lowers the thresholds (environment, read once per process => a worker process) and pushes ~150 SMALL matrices
of awkward shapes through it — single row / column, sizes straddling the strip and tile sizes, ragged and
empty rows, rows far beyond the long-row limit, value-folded columns, ELL sources — against the CPU oracle."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("stream", ["1", "0"])
def test_small_and_awkward_shapes_through_the_tiled_engine(gpu, stream):
    """stream = "1": phase 2 walks a wavefront's runs as one slot stream (the default); "0": run by run (the round-2
    form, kept behind SPMV_TILED_STREAM=0)."""
    # SPMV_PR_PLAN_AFTER=0: pagerank() builds the tiled plan before its first step (by default a matrix
    # without a plan starts on the direct kernel, see test_pagerank_switches_to_the_tiled_engine_mid_run)
    env = dict(os.environ, SPMV_TILED_MIN_COLS="1", SPMV_TILED_MIN_NNZ="1", SPMV_PR_PLAN_AFTER="0", SPMV_TILED_STREAM=stream)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tiled_small_shapes_worker.py")],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "tiled small shapes:" in out.stdout
    assert "reference fixtures through the tiled engine: 13 cases" in out.stdout
    # cells with hand-picked slot counts (every boundary case of phase 2's passes), shape forced to 4096 x 1024
    env.update(SPMV_TILED_STRIP="4096", SPMV_TILED_TILE="1024")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tiled_small_shapes_worker.py"), "patterns"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "run-length patterns:" in out.stdout
