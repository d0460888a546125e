"""plan_checksums.py — developer probe: the four plan checksums (spmv_c_csr_tiled_checksum) of the bench matrices.
The layout is a pure function of the matrix, so these numbers must not move when the BUILDER is reworked
(compare before / after).  usage: python tools/plan_checksums.py"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SPMV_TILED_FOLD", "0")
spmv = importlib.import_module("gpu-spmv_amd")
wl = importlib.import_module("gpu-spmv_amd.workloads")
spmv.require_gpu()
for name, make, kernel in (("c2", lambda: wl.uniform_csr_device(42, 1_000_000, 1_000_000, 16), 1),
                           ("c4", lambda: wl.power_law_csr_device(42, 1_000_000, 1_000_000), 2),
                           ("shard8", lambda: wl.uniform_csr_device(42, 1_250_000, 10_000_000, 16), 1),
                           ("c5", lambda: wl.uniform_csr_device(42, 10_000_000, 10_000_000, 16), 1)):
    A = make()
    x = wl.vector_device(42, 1, A.cols)
    y = spmv.CudaBuffer(A.rows)
    assert spmv.spmv_csr(A.handle, x, y, spmv.SpMVConfig(kernel, 256, True), A.cols).error_code == 0
    info = spmv.csr_tiled_info(A.handle)
    print(name, spmv.csr_tiled_checksum(A.handle), info["slots_in_cells"], "build %.2f ms" % info["build_ms"], flush=True)
    x.release(); y.release(); A.close()
