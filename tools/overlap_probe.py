"""What the overlapped exchange costs on the compute side: rank 0's shard of the C5 matrix in an 8-rank layout
(1.25 M rows x 10 M columns), one step as a whole against the same step with phase 1 issued block by block
(engine.expand per block, as ShardedPageRank.iterate does when the blocks arrive).  One GPU, no collectives:
the other ranks' pieces are simply whatever the start vector holds.  Usage: python tools/overlap_probe.py [blocks]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SPMV_TILED_FOLD", "0")
spmv = importlib.import_module("gpu-spmv_amd")
prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")

n, k, world, seed = 10_000_000, 16, 8, 42
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
spmv.require_gpu()
stream = torch.cuda.current_stream().cuda_stream


def shard(chunks):
    lay = prd.Layout(n, world, 0, chunks=chunks)
    rows = lay.local_rows
    rp = torch.empty(rows + 1, dtype=torch.int32, device=dev)
    ci = torch.empty(rows * k, dtype=torch.int32, device=dev)
    va = torch.empty(rows * k, dtype=torch.float32, device=dev)
    assert spmv.lib().spmv_c_gen_uniform_rows(seed, lay.row_begin, rows, n, k, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), stream) == 0
    va.fill_(1.0 / k)
    ci.copy_(lay.remap_columns(ci))
    eng = prd.HipEngine(rp, ci, va, lay)
    loop = prd.ShardedPageRank(eng, lay)
    mask = torch.zeros(lay.padded, dtype=torch.uint8, device=dev)
    loop.num_dangling = 0
    eng.set_dangling_mask(mask)
    loop.reset()
    return lay, eng, loop


def time_it(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ev]) * 1e3
    return float(np.median(t)), float(t.min())


for chunks in (1, blocks):
    lay, eng, loop = shard(chunks)
    r0, r1 = loop.r
    tail = loop._my_tail(r1)
    whole = time_it(lambda: eng.step(r0, r1, 0.85, tail))
    info = spmv.csr_tiled_info(eng._A)
    print("layout chunks=%d piece=%d padded=%d  plan %dx%d strips x tiles, W=%d  step as a whole: median %.1f us (min %.1f)"
          % (lay.chunks, lay.piece, lay.padded, info["num_strips"], info["num_tiles"], info["strip_cols"], *whole))
    if chunks > 1:
        def pieces():
            for c in range(lay.chunks - 1):
                eng.expand(r0, (c + 1) * lay.block)
            eng.step(r0, r1, 0.85, tail)
        split = time_it(pieces)
        print("            phase 1 in %d launches (expand per block) + the rest: median %.1f us (min %.1f)" % (lay.chunks, *split))
    eng.close()
    loop.close()
