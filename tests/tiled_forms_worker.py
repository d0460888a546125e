"""Worker of tests/test_gpu_tiled_forms.py: y of a uniform and a power-law matrix through the tiled engine, saved as raw bits
(the form of phase 2 is read from the environment once per process, hence one process per form)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spmv = importlib.import_module("gpu-spmv_amd")
wl = importlib.import_module("gpu-spmv_amd.workloads")


def main(out_path):
    spmv.require_gpu()
    outs = {}
    for name, A, kernel in (("uniform", wl.uniform_csr_device(11, 700_000, 900_000, 14), 1),
                            ("power_law", wl.power_law_csr_device(11, 600_000, 600_000), 2)):
        x = wl.vector_device(11, 1, A.cols)
        y = spmv.CudaBuffer(A.rows)
        assert spmv.spmv_csr(A.handle, x, y, spmv.SpMVConfig(kernel, 256, True), A.cols).error_code == 0
        assert spmv.csr_has_tiled_plan(A.handle)
        outs[name] = y.copyToHost(A.rows).view(np.uint32).copy()
        x.release()
        y.release()
        A.close()
    np.savez(out_path, **outs)


if __name__ == "__main__":
    main(sys.argv[1])
