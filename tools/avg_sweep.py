"""avg_sweep.py — developer probe behind the selector's `avg < 4 -> SCALAR_CSR` rule (reference
src/spmv_cpu.cpp:34-50): scalar (one thread per row, CPU order) vs vector-CSR vs merge-path on 4 M-row
matrices with EXACTLY k entries per row, k = 1 .. 16 (direct-gather kernels; x = 256 K columns so the gather is
L2-resident and the row-length effect dominates)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SPMV_TILED"] = "0"
spmv = importlib.import_module("gpu-spmv_amd")
wl = importlib.import_module("gpu-spmv_amd.workloads")

spmv.require_gpu()
rows, cols = 4_000_000, 262_144
print("k scalar_us vector_us merge_us  (algorithmic GB/s of the best)")
for k in (1, 2, 3, 4, 5, 6, 8, 12, 16):
    A = wl.uniform_csr_device(42, rows, cols, k)
    x = wl.vector_device(42, 1, cols)
    y = spmv.CudaBuffer(rows)
    t = {kt: float(np.mean(wl.time_spmv_csr(A, x, y, kt, warmup=2, runs=8))) * 1e3 for kt in (0, 1, 2)}
    nbytes = A.nnz * 8 + (rows + 1) * 4 + cols * 4 + rows * 4
    best = min(t.values())
    print(k, round(t[0], 1), round(t[1], 1), round(t[2], 1), " ", round(nbytes / best / 1e3, 1), flush=True)
    A.close(); x.release(); y.release()
