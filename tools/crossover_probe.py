"""crossover_probe.py — where does the LDS-tiled engine start to beat the direct vector-CSR kernel below its 65536-column threshold?
run as: SPMV_DEBUG=min_cols=32769 python tools/crossover_probe.py (round 4; profiles/r04_crossover.txt)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spmv = importlib.import_module("gpu-spmv_amd")
wl = importlib.import_module("gpu-spmv_amd.workloads")
spmv.require_gpu()
spmv.set_tiled_promotion(0)
for rows, k in ((1_000_000, 16), (4_000_000, 8)):
    for cols in (34_000, 40_960, 49_152, 65_536, 98_304):
        A = wl.uniform_csr_device(42, rows, cols, k)
        x = wl.vector_device(42, 1, cols); y = spmv.CudaBuffer(rows)
        out = []
        for kt, tex in ((1, False), (1, True)):
            t = wl.time_spmv_csr(A, x, y, kt, warmup=3, runs=10, use_texture=tex)
            out.append((float(np.mean(t)) * 1e3, bool(spmv.csr_has_tiled_plan(A.handle))))
        print(f"rows {rows} k {k} cols {cols}: direct {out[0][0]:.1f} us   use_texture {out[1][0]:.1f} us (tiled plan: {out[1][1]})", flush=True)
        x.release(); y.release(); A.close()
