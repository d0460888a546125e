import importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
spmv = importlib.import_module("gpu-spmv_amd")
oracle = importlib.import_module("oracle")
spmv.require_gpu()
rng = np.random.default_rng(1)
for rows, cols, k in ((1, 5000, 7), (3, 4097, 5), (2, 9000, 6)):
    per_row = [np.unique(rng.integers(0, cols, size=k)) for _ in range(rows)]
    lens = np.array([r.size for r in per_row])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate(per_row).astype(np.int32)
    va = np.arange(1, ci.size + 1, dtype=np.float32)
    x = np.ones(cols, np.float32)
    want = oracle.spmv_csr(rp, ci, va, x)
    A = spmv.csr_from_arrays(rows, cols, rp, ci, va)
    spmv.csr_to_gpu(A)
    d_x, d_y = spmv.CudaBuffer(cols), spmv.CudaBuffer(rows)
    d_x.copyFromHost(x, cols)
    r = spmv.spmv_csr(A, d_x, d_y, spmv.SpMVConfig(kernel_type=1, use_texture=True), cols)
    got = d_y.copyToHost(rows)
    print(rows, cols, "cols", ci.tolist(), "strips", (ci // 4096).tolist(), "want", want.tolist(), "got", got.tolist(), spmv.csr_tiled_info(A))
    spmv.csr_destroy(A)
