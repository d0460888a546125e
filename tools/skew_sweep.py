"""skew_sweep.py — developer probe behind the selector's skew threshold:
vector-CSR vs merge-path on 1 M-row matrices (direct-gather kernels, x = 256 K columns so the
gather is L2-resident and row-length effects dominate) whose row lengths are Pareto(1.5, 4)
capped at `cap`, i.e. skewness = cap / 5."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SPMV_TILED"] = "0"
spmv = importlib.import_module("gpu-spmv_amd")
wl = importlib.import_module("gpu-spmv_amd.workloads")

spmv.require_gpu()
rows, cols = 1_000_000, 262_144
print("cap skew avg_len vector_us merge_us scalar_us")
for cap in (8, 16, 24, 32, 48, 64, 96, 128, 256, 512, 2048, 10000):
    lens = spmv.synth.power_law_lengths(42, rows, max_len=cap, n_cols=cols)
    rp = np.zeros(rows + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    A = wl.DeviceCSR(rows, cols, int(rp[-1]))
    A.row_ptrs.copyFromHost(rp.astype(np.int32), rows + 1)
    spmv.lib().spmv_c_gen_stratified_rows(42, 0, rows, cols, A.row_ptrs.get(), A.col_indices.get(), A.values.get(), None)
    spmv.device_synchronize()
    x = wl.vector_device(42, 1, cols)
    y = spmv.CudaBuffer(rows)
    t = {k: float(np.mean(wl.time_spmv_csr(A, x, y, k, warmup=2, runs=8))) * 1e3 for k in (1, 2, 0)}
    print(cap, round(lens.max() / (lens.min() + 1), 1), round(float(lens.mean()), 2), round(t[1], 1), round(t[2], 1), round(t[0], 1), flush=True)
    A.close(); x.release(); y.release()
