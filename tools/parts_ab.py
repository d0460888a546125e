"""parts_ab.py — same-box, same-process A/B of the row-parts overlap experiment (SPMV_TILED_PARTS): for every
configuration the plan is rebuilt (the knob is read at plan build), y is compared bit for bit with the
one-part result, and the reference protocol's kernel-only time is taken; three interleaved rounds, medians.
usage: python tools/parts_ab.py [c5|c2|c4] [parts ...]"""
import importlib
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spmv = importlib.import_module("gpu-spmv_amd")
wl = importlib.import_module("gpu-spmv_amd.workloads")


def main():
    spmv.require_gpu()
    which = sys.argv[1] if len(sys.argv) > 1 else "c5"
    parts = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]
    if which == "c5":
        A = wl.uniform_csr_device(42, 10_000_000, 10_000_000, 16)
    elif which == "c2":
        A = wl.uniform_csr_device(42, 1_000_000, 1_000_000, 16)
    else:
        A = wl.power_law_csr_device(42, 1_000_000, 1_000_000)
    x = wl.vector_device(42, 1, A.cols)
    y = spmv.CudaBuffer(A.rows)
    kt = 2 if which == "c4" else 1
    times = {p: [] for p in parts}
    base = None
    for rnd in range(3):
        for p in parts:
            os.environ["SPMV_TILED_PARTS"] = str(p)
            spmv.csr_invalidate_gpu_cache(A.handle)
            t = wl.time_spmv_csr(A, x, y, kt, warmup=3, runs=10, use_texture=True)
            got = y.copyToHost(A.rows)
            if base is None:
                base = got.copy()
            same = bool(np.array_equal(got.view(np.uint32), base.view(np.uint32)))
            times[p].append(float(np.mean(t)) * 1e3)
            print(f"{which} round {rnd} parts={p}: avg {np.mean(t)*1e3:8.1f} us  min {np.min(t)*1e3:8.1f} us  "
                  f"bit-equal to parts={parts[0]}: {same}", flush=True)
    for p in parts:
        print(f"{which} parts={p}: median {statistics.median(times[p]):8.1f} us   all {[round(v, 1) for v in times[p]]}")


if __name__ == "__main__":
    main()
