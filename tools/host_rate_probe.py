"""How fast can the host ENQUEUE iterations of the one-process-per-GPU loop?  One rank through RCCL
(Layout(exchange=True)), a shard small enough that the GPU is never the bottleneck (1/8 of C5 scaled down):
host microseconds per iterate() for the one-collective form and for the overlapped form.  At eight ranks a
step takes ~200 us on the GPU: the loop must stay well under that.  usage: python tools/host_rate_probe.py"""
import importlib
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spmv = importlib.import_module("gpu-spmv_amd")
prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")
spmv.require_gpu()
dev = torch.device("cuda:0")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29544", rank=0, world_size=1, device_id=dev)
n, k = 400_000, 8
stream = torch.cuda.current_stream().cuda_stream
for chunks in (1, 2, 4):
    lay = prd.Layout(n, 1, 0, chunks=chunks, exchange=True)
    rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ci = torch.empty(n * k, dtype=torch.int32, device=dev)
    va = torch.empty(n * k, dtype=torch.float32, device=dev)
    assert spmv.lib().spmv_c_gen_uniform_rows(7, 0, n, n, k, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), stream) == 0
    va.fill_(1.0 / k)
    ci.copy_(lay.remap_columns(ci))
    eng = prd.HipEngine(rp, ci, va, lay)
    loop = prd.ShardedPageRank(eng, lay).prepare()
    loop.reset()
    for i in range(20):
        loop.iterate(i, 0.85, 0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20, 420):
        loop.iterate(i, 0.85, 0.0)
    host = (time.perf_counter() - t0) / 400
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / 400
    print("blocks=%d: host enqueue %.1f us per iteration, with the GPU %.1f us" % (chunks, host * 1e6, total * 1e6), flush=True)
    eng.close()
    loop.close()
dist.destroy_process_group()
