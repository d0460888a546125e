"""build_time.py — developer probe: how long does the first use_texture call (plan build) take?"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spmv = importlib.import_module("gpu-spmv_amd")
wl = importlib.import_module("gpu-spmv_amd.workloads")
spmv.require_gpu()
for name, (rows, cols, k) in {"c2": (1_000_000, 1_000_000, 16), "c5": (10_000_000, 10_000_000, 16)}.items():
    A = wl.uniform_csr_device(42, rows, cols, k)
    x = wl.vector_device(42, 1, cols); y = spmv.CudaBuffer(rows)
    cfg = spmv.SpMVConfig(1, 256, True)
    spmv.device_synchronize()
    t0 = time.perf_counter(); r = spmv.spmv_csr(A.handle, x, y, cfg, cols); t1 = time.perf_counter()
    r2 = spmv.spmv_csr(A.handle, x, y, cfg, cols); t2 = time.perf_counter()
    print(name, "first call %.1f ms (build + spmv), second %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    A.close()
