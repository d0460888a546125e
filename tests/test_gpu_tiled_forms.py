"""The two forms of the tiled engine's phase 2 — the slot stream (default) and the run-by-run walk (SPMV_TILED_STREAM=0) — add
the same products to the same rows in fp64: their results must be the same BITS."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_stream_and_run_by_run_forms_give_the_same_bits(gpu, tmp_path):
    results = {}
    for form in ("1", "0"):
        out_path = str(tmp_path / f"form{form}.npz")
        env = dict(os.environ, SPMV_TILED_STREAM=form)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tiled_forms_worker.py"), out_path],
                             capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        results[form] = np.load(out_path)
    for name in ("uniform", "power_law"):
        assert np.array_equal(results["1"][name], results["0"][name]), name
