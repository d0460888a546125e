"""push_rehearsal.py — developer probe: 2+ ranks sharing ONE GPU (gloo bootstrap) run the sharded
PageRank in gather mode and in push mode (IPC-mapped peer vectors) and report how the vectors differ.
launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/push_rehearsal.py"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spmv = importlib.import_module("gpu-spmv_amd")
prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
n, k = 400_000, 8
lay = prd.Layout(n, world, rank)
rp, ci, _ = spmv.synth.uniform_csr(3, 0, n, n, k)
va = spmv.synth.column_stochastic_values(ci, n)
b, e = lay.row_begin, lay.row_end
eng = prd.HipEngine(torch.from_numpy((rp[b:e + 1] - rp[b]).astype(np.int32)).to(dev),
                    torch.from_numpy(lay.remap_columns(ci[rp[b]:rp[e]]).astype(np.int32)).to(dev),
                    torch.from_numpy(va[rp[b]:rp[e]]).to(dev), lay)
pr = prd.ShardedPageRank(eng, lay).prepare()
print(rank, "push enabled:", pr.enable_push(), getattr(pr, "_push_error", ""), flush=True)


def trial(mode, steps=4):
    pr.mode = mode
    pr.reset()
    for i in range(steps):
        pr.iterate(i, 0.85, 0.0)
    torch.cuda.synchronize()
    dist.barrier()
    return pr.r[steps & 1][pr._pos].clone(), eng.status()


ref, st_ref = trial("gather")
got, st_got = trial("push")
diff = (got - ref).abs()
print(rank, "status", st_ref, st_got, "max abs diff", float(diff.max()), "at", int(diff.argmax()),
      "rel", float((diff / ref.abs().clamp_min(1e-30)).max()),
      "mine-slice diff", float(diff[b:e].max()), "other-slice diff", float(torch.cat([diff[:b], diff[e:]]).max()), flush=True)
dist.barrier()
eng.close()
pr.close()
dist.destroy_process_group()
