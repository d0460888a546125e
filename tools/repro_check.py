"""repro_check.py — developer probe: is the tiled engine's y bitwise reproducible run to run, and across a rebuild
of the plan?  (fp64 LDS accumulation makes a row's sum independent of arrival order whenever its products span
fewer than 29 binades.)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spmv = importlib.import_module("gpu-spmv_amd")
wl = importlib.import_module("gpu-spmv_amd.workloads")
spmv.require_gpu()
for name, A, kt in (("c2 1M x 16", wl.uniform_csr_device(42, 1_000_000, 1_000_000, 16), 1),
                    ("c4 1M power-law", wl.power_law_csr_device(42, 1_000_000, 1_000_000), 2),
                    ("c5 10M x 16", wl.uniform_csr_device(42, 10_000_000, 10_000_000, 16), 1)):
    x = wl.vector_device(42, 1, A.cols)
    y = spmv.CudaBuffer(A.rows)
    cfg = spmv.SpMVConfig(kt, 256, True)
    outs = []
    for rebuild in (False, False, False, True, False):
        if rebuild:
            spmv.csr_invalidate_gpu_cache(A.handle)
        assert spmv.spmv_csr(A.handle, x, y, cfg, A.cols).error_code == 0
        outs.append(y.copyToHost(A.rows).view(np.uint32).copy())
    diffs = [int(np.count_nonzero(o != outs[0])) for o in outs[1:]]
    print(f"{name:18s} rows {A.rows:9d}: rows differing from run 0 in runs 1-2 (same plan), 3 (rebuilt plan), 4: {diffs}", flush=True)
    x.release(); y.release(); A.close()
