"""min_nnz_probe.py — below how many entries does the direct vector-CSR kernel beat the LDS-tiled engine (two launches)?
run as: SPMV_DEBUG=min_nnz=1 python tools/min_nnz_probe.py (round 4; profiles/r04_crossover.txt)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spmv = importlib.import_module("gpu-spmv_amd")
wl = importlib.import_module("gpu-spmv_amd.workloads")
spmv.require_gpu()
spmv.set_tiled_promotion(0)
for rows, cols, k in ((32_768, 200_000, 8), (65_536, 200_000, 8), (131_072, 200_000, 8), (262_144, 200_000, 8), (65_536, 200_000, 16),
                      (131_072, 1_000_000, 8), (262_144, 1_000_000, 8)):
    A = wl.uniform_csr_device(42, rows, cols, k)
    x = wl.vector_device(42, 1, cols); y = spmv.CudaBuffer(rows)
    out = []
    for kt, tex in ((1, False), (1, True)):
        t = wl.time_spmv_csr(A, x, y, kt, warmup=3, runs=20, use_texture=tex)
        out.append((float(np.mean(t)) * 1e3, bool(spmv.csr_has_tiled_plan(A.handle))))
    print(f"rows {rows} cols {cols} k {k} ({rows * k} entries): direct {out[0][0]:.1f} us   use_texture {out[1][0]:.1f} us (tiled plan: {out[1][1]})", flush=True)
    x.release(); y.release(); A.close()
