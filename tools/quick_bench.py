"""quick_bench.py — developer probe: times every CSR kernel on the BASELINE configs
(device-generated inputs) and prints algorithmic GB/s.  Not the contract bench (bench.py)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spmv = importlib.import_module("gpu-spmv_amd")
wl = importlib.import_module("gpu-spmv_amd.workloads")


def report(name, A, kernels=(0, 1, 2)):
    x = wl.vector_device(42, 1, A.cols)
    y = spmv.CudaBuffer(A.rows)
    bytes_ = A.nnz * 8 + (A.rows + 1) * 4 + A.cols * 4 + A.rows * 4
    for kt in kernels:
        tex = kt >= 10
        t = wl.time_spmv_csr(A, x, y, kt % 10, warmup=3, runs=10, use_texture=tex)
        avg, best = float(np.mean(t)), float(np.min(t))
        print(f"{name:28s} kernel={kt} tiled={int(spmv.csr_has_tiled_plan(A.handle))} avg={avg*1e3:9.1f}us min={best*1e3:9.1f}us "
              f"GB/s={bytes_/avg/1e6:8.1f} frac={bytes_/avg/1e6/8000:.3f} GFLOPS={2*A.nnz/avg/1e6:8.1f}",
              flush=True)
    x.release(); y.release()


def main():
    spmv.require_gpu()
    print(spmv.version(), spmv.device_name(), flush=True)
    which = sys.argv[1:] or ["c1", "c2", "c4", "c5"]
    if "c1" in which:
        A = wl.uniform_csr_device(42, 1000, 1000, 8); report("c1 1k x 8", A); A.close()
    if "c2" in which:
        A = wl.uniform_csr_device(42, 1_000_000, 1_000_000, 16); report("c2 1M x 16", A, kernels=(0, 1, 2, 11)); A.close()
    if "c4" in which:
        A = wl.power_law_csr_device(42, 1_000_000, 1_000_000)
        print("c4 nnz", A.nnz, flush=True); report("c4 1M power-law", A, kernels=(1, 2, 12)); A.close()
    if "crossover" in which:  # where does the LDS-tiled engine start to beat the direct gather?
        for cols in (65_536, 131_072, 262_144, 524_288):
            A = wl.uniform_csr_device(42, 1_000_000, cols, 16); report("1M rows x %d cols" % cols, A, kernels=(1, 11)); A.close()
    if "smallx" in which:     # x (30 K columns) fits one CU's LDS: direct gather vs x resident in LDS
        A = wl.uniform_csr_device(42, 2_000_000, 30_000, 16); report("2M x 30K cols, 16/row", A, kernels=(1, 11)); A.close()
    if "shard8" in which:     # what one rank of 8 sees of C5: 1.25 M rows x 10 M columns
        A = wl.uniform_csr_device(42, 1_250_000, 10_000_000, 16); report("c5 shard 1/8", A, kernels=(11, 1)); A.close()
    if "shard2" in which:
        A = wl.uniform_csr_device(42, 5_000_000, 10_000_000, 16); report("c5 shard 1/2", A, kernels=(11,)); A.close()
    if "c2only" in which:
        A = wl.uniform_csr_device(42, 1_000_000, 1_000_000, 16); report("c2 1M x 16", A, kernels=(11,)); A.close()
    if "c4only" in which:
        A = wl.power_law_csr_device(42, 1_000_000, 1_000_000); report("c4 1M power-law", A, kernels=(12,)); A.close()
    if "pagerank" in which:   # the drop-in pagerank() call end to end on the column-stochastic C5 (SPMV_TRACE=1 for phases)
        import time
        A = wl.uniform_csr_device(42, 10_000_000, 10_000_000, 16)
        counts = wl.make_column_stochastic(A)
        for attempt in range(3):
            t0 = time.perf_counter()
            res = spmv.pagerank(A.handle, spmv.PageRankConfig(0.85, 1e-6, 100))
            dt = time.perf_counter() - t0
            print(f"pagerank() call {attempt}: {dt*1e3:.2f} ms, {res.iterations} iterations, converged={res.converged}, "
                  f"sum={float(res.ranks.sum(dtype=np.float64)):.9f}", flush=True)
        counts.release(); A.close()
    if "pr_small" in which:   # launch-bound PageRank: small graphs, 100 iterations forced (tolerance 0)
        import time
        for n, k in ((10_000, 8), (100_000, 8), (1_000_000, 8)):
            A = wl.uniform_csr_device(7, n, n, k)
            counts = wl.make_column_stochastic(A)
            for attempt in range(3):
                t0 = time.perf_counter()
                res = spmv.pagerank(A.handle, spmv.PageRankConfig(0.85, 0.0, 100))
                dt = time.perf_counter() - t0
            print(f"pagerank() n={n} k={k}: {dt*1e3:.2f} ms for {res.iterations} iterations = {dt*1e6/max(res.iterations,1):.1f} us/iteration", flush=True)
            counts.release(); A.close()
    if "c5pl" in which:       # power-law rows at C5 scale (what a real PageRank graph looks like)
        A = wl.power_law_csr_device(42, 10_000_000, 10_000_000)
        print("c5pl nnz", A.nnz, flush=True); report("10M power-law", A, kernels=(12, 2))
        print("plan", spmv.csr_tiled_info(A.handle), flush=True); A.close()
    if "onestrip" in which:   # run with SPMV_DEBUG=min_cols=1: ONE strip, so every tile's entries are one contiguous run
        A = wl.uniform_csr_device(42, 10_000_000, 32_768, 16); report("10M x 32K cols (1 strip)", A, kernels=(11,))
        print("plan", spmv.csr_tiled_info(A.handle), flush=True); A.close()
    if "c5only" in which:
        A = wl.uniform_csr_device(42, 10_000_000, 10_000_000, 16); report("c5 10M x 16", A, kernels=(11,)); A.close()
    if "c5" in which:
        A = wl.uniform_csr_device(42, 10_000_000, 10_000_000, 16); report("c5 10M x 16", A, kernels=(1, 11)); A.close()


if __name__ == "__main__":
    main()
