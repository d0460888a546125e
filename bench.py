#!/usr/bin/env python3
"""bench.py — the contract benchmark.

Workload (BASELINE.json: the matrix the headline metric is quoted on): uniform random
CSR, 10,000,000 x 10,000,000, exactly 16 entries per row (160 M entries, fp32 values,
int32 indices, column-stochastic), generated directly in HBM.

A "step" is one pass of the hot path over that matrix: ONE vector-CSR SpMV, run as a
PageRank power iteration (fused SpMV + damping/teleport update + residual partials,
then — for N > 1 — the RCCL all-reduce of two scalars and the all-gather of the rank
slices).  Rows are sharded over the N ranks (strong scaling: total work fixed).

    value = (algorithmic CSR bytes of the WHOLE matrix, reference byte model
             src/bandwidth.cpp:34-42) x steps / wall time          [GB/s, whole job]

also reported: GFLOPS, PageRank iterations/s, the roofline of the dominant kernel
(HIP events around the step kernel alone), the drop-in spmv_csr() numbers per kernel
type for BASELINE configs 2-5 (reference protocol: 5 warm-up + 20 timed calls,
kernel-only event time), and the CPU baseline (the reference's spmv_cpu_csr on the same
matrix, one host thread).

usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--nnz-per-row k] [--no-extras]
       (N > 1: launched by torch.distributed.run, one rank per GPU)
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--rows", type=int, default=10_000_000)
    p.add_argument("--nnz-per-row", type=int, default=16)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--no-extras", action="store_true", help="skip the per-config spmv_csr table and the CPU baseline")
    return p.parse_args()


def csr_bytes(rows, cols, nnz):
    return nnz * 8 + (rows + 1) * 4 + cols * 4 + rows * 4



# ------------------------------------------------------------------------------------------------
# One JSON line, whatever happens after the measurement.
#
# The measurement itself (W + K steps of the one-collective loop, then the step kernel under HIP events)
# comes first.  Everything after it — CPU baseline, spmv_csr table, parity report, folded plan, opt-in exchange
# trials — can only ADD keys.  By default bench.py runs as a SUPERVISOR that never touches the GPU: it starts
# the real run as a child (same interpreter, same arguments, SPMV_BENCH_CHILD=1) and reads the child's
# stdout.  The child writes a complete line as soon as the measurement stands and again after every extra;
# the supervisor prints the LAST line it received when the child ends — normally, by a Python exception, by
# a GPU fault that aborts the process, by a signal, or by the deadline for the extras (a hung collective) —
# and exits 0 once a measured line exists.  Under a profiler (rocprofv3 preloads a library that has
# already initialised the GPU, so this process must not start another program) or with
# SPMV_BENCH_INPROCESS=1 the run stays in this process and prints its line from a `finally`.
def _under_profiler():
    """True when a tool library has (or may have) initialised the GPU in THIS process before main() runs: then this
    process must not start another program (the pool forbids that hop) and the run stays in-process.  Recognised: a
    preloaded profiler / HIP / HSA library, the rocprofiler tool-library variables, any ROCPROF* / ROCP_ variable.
    A wrapper this does not recognise sets SPMV_BENCH_INPROCESS=1 (tools/README.md)."""
    preload = os.environ.get("LD_PRELOAD", "").lower()
    if any(tag in preload for tag in ("rocprof", "libhsa", "libamdhip", "roctracer", "roctx", "omnitrace", "rocsys", "rocprofiler")):
        return True
    if os.environ.get("HSA_TOOLS_LIB") or os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("ROCP_TOOL_LIB"):
        return True
    return any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launcher_command(gpus, argv):
    """`python bench.py --gpus N` started plainly (no rank variables): the supervisor — a process that never touches
    the GPU — starts torch.distributed.run itself, one measuring child per GPU (the driver's own launch line)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)


def _ending(returncode):
    import signal
    if returncode is None:
        return "still running (killed by the supervisor)"
    if returncode < 0:
        try:
            return "signal " + signal.Signals(-returncode).name
        except ValueError:
            return "signal %d" % -returncode
    return "exit code %d" % returncode


def supervise(child_cmd=None, self_launched=False):
    """Deadlines (all wall clock of THIS process, so a SIGKILL at the caller's limit is never what ends the run):
    SPMV_BENCH_BUDGET (default 540 s: the driver runs bench.py under 600 s) bounds the whole run; a child that has not
    measured by SPMV_BENCH_MEASURE_DEADLINE (default budget - 120 s) is killed and the exit code is non-zero; the extras
    end SPMV_BENCH_EXTRAS_DEADLINE seconds after the measurement (default 300) or 20 s before the budget, whichever
    comes first."""
    import signal
    import subprocess
    import threading

    started = time.time()
    rank = int(os.environ.get("RANK", "0"))
    env = dict(os.environ, SPMV_BENCH_CHILD="1")
    if self_launched:
        env["SPMV_BENCH_SELF_LAUNCHED"] = "1"          # ranks other than 0 then write nothing to the shared stdout
    child = subprocess.Popen(child_cmd or [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                             stdout=subprocess.PIPE, env=env)       # same process group: whoever ends the job ends the child too
    state = {"line": None, "measured_at": None, "final": False}

    def reader():
        for raw in child.stdout:
            text = raw.decode(errors="replace").strip()
            if text == "MEASURED":                       # ranks other than 0: the timed region is over
                state["measured_at"] = time.time()
                continue
            try:
                obj = json.loads(text)
            except ValueError:
                print(text, file=sys.stderr, flush=True)  # a library wrote to the child's stdout: not ours
                continue
            if not isinstance(obj, dict) or "metric" not in obj:
                print(text, file=sys.stderr, flush=True)
                continue
            state["line"] = obj
            state["final"] = not obj.get("provisional", False)
            if state["measured_at"] is None:
                state["measured_at"] = time.time()

    thread = threading.Thread(target=reader, daemon=True)
    thread.start()

    def finish(reason):
        if child.poll() is None:
            child.kill()
            try:
                child.wait(timeout=10)
            except Exception:                           # noqa: BLE001
                pass
            killed = True
        else:
            killed = False
        thread.join(timeout=5)
        obj = state["line"]
        ending = _ending(child.returncode) + (" after the supervisor killed it" if killed else "")
        if obj is not None and rank == 0:
            if obj.pop("provisional", False):          # (a final line is complete whatever ends the child afterwards)
                obj["incomplete"] = reason or "the run ended before its last extra (%s)" % ending
            # how the measuring child ended, always: a fault behind the final line (teardown, an opt-in trial) stays findable
            obj["child_exit"] = ending
            sys.stdout.write(json.dumps(obj) + "\n")
            sys.stdout.flush()
        if child.returncode not in (0, None) or killed:
            print("[bench] measuring child: %s%s" % (ending, "; " + reason if reason else ""), file=sys.stderr, flush=True)
        measured = state["measured_at"] is not None
        rc = child.returncode if child.returncode is not None else -9
        os._exit(0 if measured else (rc if rc > 0 else 1))

    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, lambda number, frame: finish("signal %d while the extras were running" % number))
    budget = float(os.environ.get("SPMV_BENCH_BUDGET", "540"))
    measure_deadline = float(os.environ.get("SPMV_BENCH_MEASURE_DEADLINE", str(max(budget - 120.0, 30.0))))
    extras_deadline = float(os.environ.get("SPMV_BENCH_EXTRAS_DEADLINE", "300"))
    final_at = None
    while child.poll() is None:
        time.sleep(0.2)
        now = time.time()
        if state["final"] and final_at is None:
            final_at = now
        if state["measured_at"] is None:
            if now - started > measure_deadline:
                finish("no measurement within %.0f s: child killed" % measure_deadline)
            continue
        out_of_budget = now - started > budget - 20.0
        if not state["final"] and (now - state["measured_at"] > extras_deadline or out_of_budget):
            finish("the extras did not finish within %s: child killed"
                   % ("the run's budget of %.0f s" % budget if out_of_budget else "%.0f s of the measurement" % extras_deadline))
        if final_at is not None and (now - final_at > extras_deadline or out_of_budget):   # trials or the teardown hang behind a complete line
            finish(None)
    finish(None)


def _fail_here(stage):
    """Test hook: SPMV_BENCH_FAIL_IN=<stage>:<raise|abort|hang> breaks the named stage on purpose."""
    spec = os.environ.get("SPMV_BENCH_FAIL_IN", "")
    if not spec:
        return
    name, _, how = spec.partition(":")
    if name != stage:
        return
    if how == "abort":
        os.abort()
    if how == "hang":
        while True:
            time.sleep(1)
    raise RuntimeError("SPMV_BENCH_FAIL_IN asked stage %r to fail" % stage)


def main():
    args = parse()
    if os.environ.get("SPMV_BENCH_CHILD") != "1":
        if os.environ.get("SPMV_BENCH_INPROCESS", "0") == "1" or _under_profiler():
            print("[bench] running in this process (profiler or SPMV_BENCH_INPROCESS=1): no supervisor", file=sys.stderr, flush=True)
        elif args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            print("[bench] --gpus %d without a launcher: starting torch.distributed.run" % args.gpus, file=sys.stderr, flush=True)
            supervise(launcher_command(args.gpus, sys.argv[1:]), self_launched=True)      # never returns
        else:
            supervise()                               # never returns
    # The contract: stdout carries exactly ONE line, the JSON.  Libraries write there too (RCCL prints a version
    # banner on stdout when a communicator is created), so file descriptor 1 is pointed at stderr for the run and
    # the line goes out through a private duplicate of the real stdout.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world

    spmv = importlib.import_module("gpu-spmv_amd")
    wl = importlib.import_module("gpu-spmv_amd.workloads")
    prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the SpMV path has no CPU fallback)")
    # SPMV_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 code path
    # on a single-GPU box (numbers from such a run mean nothing); the real thing is RCCL.
    backend = os.environ.get("SPMV_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    spmv.lib().spmv_c_set_device(dev_index)
    spmv.require_gpu()
    device = torch.device("cuda", dev_index)
    # SPMV_BENCH_FORCE_EXCHANGE=1 with one rank: tails, collectives and the exchange trials stay in the loop
    # (a rehearsal of the N > 1 code path on the real backend; the number it prints is not the N = 1 number)
    force_exchange = world == 1 and os.environ.get("SPMV_BENCH_FORCE_EXCHANGE", "0") == "1"
    if world > 1 or force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # a peer that never arrives, or a collective a peer never joins, ends as an error after this long, not as a hang
        import datetime
        limit = datetime.timedelta(seconds=int(os.environ.get("SPMV_DIST_TIMEOUT", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)

    # The bench matrix is column-stochastic (a_ij = 1 / outdeg(j)), a structure the tiled engine can
    # exploit by folding the values into one weight per column.  The headline is
    # measured on the GENERAL path (value stream read per entry), so that it holds for any values;
    # the folded step is reported next to it as `pagerank_step_values_folded`.  SPMV_TILED_FOLD=1 in
    # the environment moves the headline onto the folded path (config.values_folded says which ran).
    os.environ.setdefault("SPMV_TILED_FOLD", "0")

    n, k = args.rows, args.nnz_per_row
    nnz_total = n * k
    layout = prd.Layout(n, world, rank, exchange=True if force_exchange else None)
    shard_len, row_begin, local_rows = layout.shard_len, layout.row_begin, layout.local_rows

    # ---- build this rank's rows in HBM (torch owns the memory; the C ABI fills it) ----
    stream = torch.cuda.current_stream().cuda_stream
    row_ptrs = torch.empty(local_rows + 1, dtype=torch.int32, device=device)
    cols = torch.empty(max(local_rows * k, 1), dtype=torch.int32, device=device)
    vals = torch.empty(max(local_rows * k, 1), dtype=torch.float32, device=device)
    status = spmv.lib().spmv_c_gen_uniform_rows(args.seed, row_begin, local_rows, n, k, row_ptrs.data_ptr(),
                                                cols.data_ptr(), vals.data_ptr(), stream)
    assert status == 0, spmv.spmv_error_string(status)
    counts = torch.zeros(n, dtype=torch.int32, device=device)
    spmv.lib().spmv_c_count_columns(local_rows * k, cols.data_ptr(), n, counts.data_ptr(), stream)
    if world > 1 or force_exchange:
        dist.all_reduce(counts)
    spmv.lib().spmv_c_reciprocal_values(local_rows * k, cols.data_ptr(), counts.data_ptr(), vals.data_ptr(), stream)
    del counts
    cols_v, vals_v = cols[: local_rows * k], vals[: local_rows * k]
    if layout.exchange:
        cols_v.copy_(layout.remap_columns(cols_v))     # node ids -> positions in the padded rank vector

    engine = prd.HipEngine(row_ptrs, cols_v, vals_v, layout)
    pr = prd.ShardedPageRank(engine, layout).prepare()
    pr.reset()

    damping, never = 0.85, 0.0          # tolerance 0: the loop never converges, every step does full work

    # ---- N > 1: how the new slices travel.  The recorded number ALWAYS runs on the RCCL all-gather (one
    # collective per step: the exchange BASELINE.json names).  The push exchange (step kernels store the new
    # slices straight into the peers' IPC-mapped vectors + a 16-byte all-reduce) has never run across real
    # devices; it is tried only when SPMV_PR_EXCHANGE=push|auto is set explicitly, and only AFTER the result
    # line of the gather run is already printed (see the end of main), so nothing it does can lose that line.
    def barrier():
        torch.cuda.synchronize()
        if world > 1 or force_exchange:
            dist.barrier()
        torch.cuda.synchronize()

    exchange = "gather" if layout.exchange else "none"
    pr.mode = "gather"

    def timed(loop, its_engine):
        """W untimed + K timed iterations of `loop`; seconds for the K, max over the ranks."""
        loop.reset()
        step = 0
        for _ in range(args.warmup):
            loop.iterate(step, damping, never)
            step += 1
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loop.iterate(step, damping, never)
            step += 1
        barrier()
        seconds = time.perf_counter() - t0
        if world > 1 or force_exchange:
            t = torch.tensor([seconds], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            seconds = float(t.item())
        iters_done = its_engine.status()[0]
        assert iters_done == args.warmup + args.steps, (iters_done, args.warmup, args.steps)
        return seconds

    elapsed = timed(pr, engine)
    if rank != 0 and os.environ.get("SPMV_BENCH_CHILD") == "1" and os.environ.get("SPMV_BENCH_SELF_LAUNCHED") != "1":
        os.write(result_fd, b"MEASURED\n")          # tells this rank's supervisor that the timed region is over

    bytes_per_step = csr_bytes(n, n, nnz_total)
    ms_per_step = elapsed / args.steps * 1e3
    value = bytes_per_step * args.steps / elapsed / 1e9

    # ---- roofline of the dominant kernel: the fused step kernel alone, HIP events on its stream ----
    local_bytes = csr_bytes(local_rows, n, local_rows * k)     # this rank's share of the algorithmic bytes
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record()
        engine._check(spmv.lib().spmv_c_pr_step(engine._shard, pr.r[0].data_ptr(), pr.r[1].data_ptr(), damping,
                                                engine._stream()), "pr_step")
        b.record()
    torch.cuda.synchronize()
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    achieved = local_bytes / (kernel_ms * 1e-3) / 1e9
    tiled = spmv.csr_has_tiled_plan(engine._A)
    plan_info = spmv.csr_tiled_info(engine._A)
    # roofline.traffic is NOT measured by this run: it is the PMC figure of a separate rocprofv3 pass over this
    # same command (tools/pmc_traffic.sh -> profiles/pmc_traffic.json; FETCH_SIZE x 2 + WRITE_SIZE, see
    # MI355X_MICROARCH.md).  It is reported only while that file describes the plan shape that ran here.
    traffic, traffic_source = None, None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_file) and world == 1 and tiled:
        try:
            pmc = json.load(open(pmc_file))
            same_shape = all(pmc.get("plan", {}).get(k) == plan_info.get(k)
                             for k in ("strip_cols", "tile_rows", "num_strips", "num_tiles", "slots_in_cells"))
            import hashlib
            same_sources = all(
                hashlib.sha256(open(os.path.join(ROOT, "gpu-spmv_amd", "csrc", name), "rb").read()).hexdigest() == digest
                for name, digest in pmc.get("sources_sha256", {"missing": ""}).items()) if pmc.get("sources_sha256") else False
            if same_shape and same_sources:
                traffic = pmc.get("pr_step_kernel_bytes_per_launch")
                traffic_source = "profiles/pmc_traffic.json (commit %s; %s)" % (pmc.get("commit", "?"), pmc.get("collected", "separate rocprofv3 --pmc passes"))
            elif same_shape:
                traffic_source = ("profiles/pmc_traffic.json was collected on other kernel sources (commit %s): not reported"
                                  % pmc.get("commit", "?"))
            else:
                traffic_source = "profiles/pmc_traffic.json describes another plan shape: not reported"
        except Exception:
            traffic = None
    step_kernels = ("tiled_expand_kernel + tiled_pagerank_reduce_kernel (two launches per step: bucketed-slot "
                    "propagation-blocking engine, x strips and y tiles in LDS)"
                    if tiled else "pr_step_kernel (fused vector-CSR direct-gather SpMV + PageRank update)")
    # what the bound is: `frac` prices the ALGORITHMIC bytes; the engine moves `traffic` bytes
    # (1.84x of them: the product round trip), and that traffic over the same time is what the memory system
    # actually carried — a kernel saturated on wasted bytes, not an idle one
    traffic_frac = round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None
    roofline = {"bound": "hbm", "kernel": step_kernels,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "traffic_frac": traffic_frac,
                "floor_note": ("frac = algorithmic bytes / kernel time / peak; traffic_frac = the bytes the kernels really moved "
                               "(PMC) over the same time: the two-phase engine streams 15 B per entry (7 B of bucketed matrix + "
                               "a 4 B product written and read back) against the byte model's 8 B; both kernels run within ~10 % of "
                               "what plain streams of their bytes take on the same box (profiles/r04_phase2_bound.txt: phase 1 "
                               "307-312 us against 297, phase 2 157 against ~140-156), and running the phases side by side or part "
                               "by part through the Infinity Cache does not pay (profiles/r04_parts_overlap_rejected.txt, "
                               "r04_infinity_cache_parts_rejected.txt)")
                              if tiled else None,
                "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": local_bytes,
                "tiled_plan": plan_info}

    result = {
        "metric": "spmv_effective_bandwidth", "value": round(value, 1), "unit": "GB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (counter-based uniform random CSR generated in HBM, seed %d)" % args.seed,
        "config": {"workload": "csr_uniform_%dx%d_%d_per_row_pagerank_step" % (n, n, k), "rows": n, "cols": n,
                   "nnz": nnz_total, "avg_nnz_per_row": k,
                   "kernel": ("LDS-tiled engine (what SpMVConfig::use_texture selects for VECTOR_CSR on this matrix): a second, "
                              "bucketed copy of the entries (7 B per slot + a 4 B product slot) streamed in two phases per step; "
                              "the direct vector-CSR kernel on the CSR arrays is the row c5_10Mx16/vector of spmv_csr_api")
                             if tiled else "VECTOR_CSR direct gather (fused PageRank step)",
                   "plan_build_ms": plan_info.get("build_ms") if plan_info else None,
                   "plan_bytes": plan_info.get("plan_bytes") if plan_info else None,
                   "plan_note": "one-time per matrix, outside the timed region; break-even against the direct kernel "
                                "after ~3 SpMVs (pagerank() starts on the direct kernel and builds it after 4 steps)",
                   "parallelism": "row-shard x%d%s" % (world, "" if world == 1 else
                       " + %d RCCL all-gather(s) per step (%d f32/rank each, partial sums in the slice tails)"
                       % (layout.chunks, layout.piece)),
                   "exchange": exchange,
                   "exchange_note": None if world == 1 else
                       "one RCCL all-gather per step (the recorded default); the overlapped and push exchanges are opt-in "
                       "(SPMV_PR_OVERLAP / SPMV_PR_EXCHANGE), run after this line and report on stderr",
                   "values_folded": bool(plan_info and plan_info.get("values_folded"))},
        "gflops": round(2.0 * nnz_total * args.steps / elapsed / 1e9, 1),
        "pagerank_iters_per_sec": round(args.steps / elapsed, 2),
        "frac_of_hbm_peak_whole_job": round(value / (HBM_PEAK_GBS * world), 4),
        "roofline": roofline,
    }

    child = os.environ.get("SPMV_BENCH_CHILD") == "1"
    printed = {"final": False}

    def emit(final):
        """The line as it stands.  Child: after the measurement and after every extra (the supervisor keeps the last one).
        In-process: once, at the end or from the `finally` below."""
        if rank != 0 or printed["final"]:
            return
        if not final and not child:
            return
        line = dict(result)
        if not final:
            line["provisional"] = True
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())
        printed["final"] = final

    def extra(name, fn):
        """An extra can add keys; a Python exception in it costs that extra only."""
        try:
            _fail_here(name)
            fn()
            result.setdefault("extras_done", []).append(name)
        except Exception as exc:                                    # noqa: BLE001
            result.setdefault("extras_failed", {})[name] = repr(exc)
        emit(False)

    try:
        emit(False)                                   # the measurement stands from here on
        if rank == 0 and world == 1 and not args.no_extras:
            def do_cpu_baseline():
                result["cpu_baseline"], result["cpu_baseline_all_cores"] = cpu_baseline(spmv, row_ptrs, cols_v, vals_v, n, nnz_total)

            def do_pagerank_api():
                # the drop-in pagerank() call end to end on the same matrix (d = 0.85, tol = 1e-6, <= 100 iterations), with
                # the tiled plan the step engine built above already cached: first call of the process (allocates the
                # workspace kept with the matrix and the pinned result array) and a warm one
                calls = []
                for _ in range(2):
                    t0 = time.perf_counter()
                    full = spmv.pagerank(engine._A, spmv.PageRankConfig(0.85, 1e-6, 100))
                    calls.append(time.perf_counter() - t0)
                    rank_sum = float(full.ranks.sum(dtype=np.float64))
                    iterations, converged, residual = full.iterations, bool(full.converged), full.final_residual
                    del full                                   # hands the pinned result array back to the pool
                result["pagerank_api"] = {"iterations": iterations, "converged": converged, "final_residual": residual,
                                          "seconds_total": round(calls[1], 5), "seconds_first_call": round(calls[0], 5),
                                          "ms_per_iteration_incl_setup": round(calls[1] / max(iterations, 1) * 1e3, 3),
                                          "rank_sum": rank_sum,
                                          "note": "seconds_total = a call with plan and workspace warm, result delivered in the "
                                                  "library's pinned array; seconds_first_call also allocates them"}

            def do_cpu_pagerank():
                # the reference's HOST loop (src/pagerank.cu:50-153, restated in oracle_pagerank) on the same
                # matrix, stopped after as many iterations as the GPU call above took
                gpu_iterations = (result.get("pagerank_api") or {}).get("iterations") or 3
                if result.get("cpu_baseline"):
                    result["cpu_baseline"]["pagerank"] = cpu_pagerank_baseline(row_ptrs, cols_v, vals_v, n, gpu_iterations,
                                                                               result.get("pagerank_api"))

            def do_api_table():
                result["spmv_csr_api"] = api_table(spmv, wl, engine, n, k, args.seed)

            def do_parity_report():
                result["parity_report"] = parity_report(spmv, wl, args.seed)

            def do_folded():
                # the same step with the values folded into column weights (what this column-stochastic matrix
                # allows): a second plan over the same device arrays, built with folding on
                if result["config"]["values_folded"]:
                    return
                os.environ["SPMV_TILED_FOLD"] = "1"
                folded = prd.HipEngine(row_ptrs.clone(), cols_v, vals_v, layout)
                pr_f = prd.ShardedPageRank(folded, layout).prepare()
                pr_f.reset()
                for i in range(3):
                    pr_f.iterate(i, damping, never)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(3, 3 + args.steps):
                    pr_f.iterate(i, damping, never)
                torch.cuda.synchronize()
                t_f = (time.perf_counter() - t0) / args.steps
                info_f = spmv.csr_tiled_info(folded._A)
                folded_traffic = None
                try:
                    folded_traffic = json.load(open(pmc_file)).get("folded", {}).get("bytes_per_step")
                except Exception:
                    pass
                result["pagerank_step_values_folded"] = {
                    "ms_per_step": round(t_f * 1e3, 4), "effective_gb_s": round(bytes_per_step / t_f / 1e9, 1),
                    "frac_of_hbm_peak": round(bytes_per_step / t_f / 1e9 / HBM_PEAK_GBS, 4),
                    "values_folded": bool(info_f and info_f.get("values_folded")),
                    "traffic": folded_traffic,
                    "note": "same matrix, same arithmetic (w_j * x_j rounded once per column); applies only when every "
                            "stored entry of a column is bit-identical"}
                folded.close()
                pr_f.close()

            result["cpu_baseline"] = None
            extra("cpu_baseline", do_cpu_baseline)     # the contract's other object: first, and it touches no kernel
            extra("pagerank_api", do_pagerank_api)
            extra("cpu_baseline_pagerank", do_cpu_pagerank)
            extra("spmv_csr_api", do_api_table)
            extra("parity_report", do_parity_report)
            extra("values_folded", do_folded)
        elif rank == 0:
            result["cpu_baseline"] = None
        emit(True)

        # ---- N > 1, opt-in, AFTER the line: other exchanges against the recorded all-gather.  Whatever happens in
        # here — an exception, a fault, a hung collective on hardware these paths have never seen — the line is out.
        overlap = [int(c) for c in os.environ.get("SPMV_PR_OVERLAP", "").split(",") if c.strip()]
        overlap = [c for c in overlap if c > 1]
        if layout.exchange and overlap:
            _fail_here("trial")
            overlap_trial(spmv, prd, dist, torch, device, args, layout, pr, engine, row_ptrs, cols, vals, vals_v, stream,
                          damping, never, timed, elapsed, overlap)
        if world > 1 and os.environ.get("SPMV_PR_EXCHANGE", "gather") in ("push", "auto"):
            push_trial(pr, dist, torch, device, world, rank, backend, damping, never, barrier, ms_per_step)
    finally:
        emit(True)
    engine.close()
    pr.close()          # unmaps peers, barriers, then frees the rank vectors
    if world > 1 or force_exchange:
        dist.destroy_process_group()


def overlap_trial(spmv, prd, dist, torch, device, args, layout, pr, engine, row_ptrs, cols, vals, vals_v, stream,
                  damping, never, timed, gather_seconds, candidates):
    """Opt-in (SPMV_PR_OVERLAP=<C>[,<C>...]), after the result line: the same K steps with the OVERLAPPED exchange
    (Layout(chunks=C): C all-gathers per step, the products of block c computed while block c + 1 is on the links;
    pagerank_dist.py).  Same collectives, same kernels, another numbering of the vector — so each form is first
    checked against the one-collective run (4 steps from the start vector, every node within 1e-5 relative on every
    rank) and then timed over the same W + K steps.  Reports on stderr."""
    n, k, world, rank = layout.n, args.nnz_per_row, layout.world, layout.rank
    row_begin, local_rows = layout.row_begin, layout.local_rows

    def after(loop, steps):
        loop.reset()
        for i in range(steps):
            loop.iterate(i, damping, never)
        torch.cuda.synchronize()
        return loop.r[steps & 1][loop._pos].clone()

    ref = after(pr, 4)
    report = {"gather_ms_per_step": round(gather_seconds / args.steps * 1e3, 4), "overlapped": []}
    try:
        for blocks in candidates:
            lay2 = prd.Layout(n, world, rank, chunks=blocks, exchange=True)
            cols2 = torch.empty_like(cols)
            scratch_ptrs, scratch_vals = torch.empty_like(row_ptrs), torch.empty_like(vals)
            status = spmv.lib().spmv_c_gen_uniform_rows(args.seed, row_begin, local_rows, n, k, scratch_ptrs.data_ptr(),
                                                        cols2.data_ptr(), scratch_vals.data_ptr(), stream)   # the node ids again
            assert status == 0, spmv.spmv_error_string(status)
            torch.cuda.synchronize()
            del scratch_ptrs, scratch_vals
            cols2_v = cols2[: local_rows * k]
            cols2_v.copy_(lay2.remap_columns(cols2_v))
            # (its own copy of the row pointers: the library keys a matrix's cached plan by that array)
            engine2 = prd.HipEngine(row_ptrs.clone(), cols2_v, vals_v, lay2)
            pr2 = prd.ShardedPageRank(engine2, lay2).prepare()
            got = after(pr2, 4)
            worst = float(((got - ref).abs() / ref.abs().clamp_min(1e-30)).max())
            agree = torch.tensor([1 if worst <= 1e-5 else 0], dtype=torch.int32, device=device)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            trial = {"blocks": blocks, "max_rel_diff_after_4_steps": worst, "agrees": bool(int(agree.item()))}
            if trial["agrees"]:
                trial["ms_per_step"] = round(timed(pr2, engine2) / args.steps * 1e3, 4)
            report["overlapped"].append(trial)
            engine2.close()
            pr2.close()
    except Exception as exc:                                    # noqa: BLE001
        report["aborted"] = repr(exc)
    if rank == 0:
        print(json.dumps({"exchange_trials": report}), file=sys.stderr, flush=True)


def push_trial(pr, dist, torch, device, world, rank, backend, damping, never, barrier, gather_ms):
    """Opt-in experiment, run after the result line is out: the push exchange against the all-gather.
    Reports on stderr (the contract's stdout carries exactly one JSON line)."""
    report = {"push_exchange": "unavailable"}
    peer_devices = list(range(world)) if backend == "nccl" else None
    if pr.enable_push(peer_devices):
        def trial(mode):
            pr.mode = mode
            pr.reset()
            for i in range(4):
                pr.iterate(i, damping, never)
            torch.cuda.synchronize()
            return pr.r[0][pr._pos].clone()
        ref, got = trial("gather"), trial("push")
        worst = float(((got - ref).abs() / ref.abs().clamp_min(1e-30)).max())
        flag = torch.tensor([1 if worst <= 1e-5 else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        report = {"push_exchange": "agrees with the all-gather path" if int(flag.item()) else "DISAGREES",
                  "worst_rel_difference_after_4_steps": worst}
        if int(flag.item()) == 1:
            pr.mode = "push"
            pr.reset()
            for i in range(3):
                pr.iterate(i, damping, never)
            barrier()
            t0 = time.perf_counter()
            for i in range(3, 23):
                pr.iterate(i, damping, never)
            barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            # every later step against the gather path too: a stale slice would show here
            after = pr.r[23 & 1][pr._pos].clone()
            pr.mode = "gather"
            pr.reset()
            for i in range(23):
                pr.iterate(i, damping, never)
            torch.cuda.synchronize()
            drift = float(((after - pr.r[23 & 1][pr._pos]).abs() / pr.r[23 & 1][pr._pos].abs().clamp_min(1e-30)).max())
            report.update({"push_ms_per_step": round(float(t.item()) / 20 * 1e3, 4), "gather_ms_per_step": round(gather_ms, 4),
                           "worst_rel_difference_after_23_steps": drift})
    else:
        report["reason"] = getattr(pr, "_push_error", "a peer failed")
    pr.mode = "gather"
    if rank == 0:
        print(json.dumps(report), file=sys.stderr, flush=True)


def api_table(spmv, wl, engine, n, k, seed):
    """Drop-in spmv_csr()/spmv_ell() numbers, reference protocol (5 warm-up + 20 timed, event time)."""
    table = {}

    def run(name, handle, rows, cols, nnz, kernels):
        x = wl.vector_device(seed, 1, cols)
        y = spmv.CudaBuffer(rows)
        b = csr_bytes(rows, cols, nnz)
        for kt, label in kernels:
            # "..._direct": the kernel the caller spelled, promotion to the tiled engine switched off for the row
            promotion = spmv.get_tiled_promotion()
            if label.endswith("_direct"):
                spmv.set_tiled_promotion(0)
            try:
                t = wl.time_spmv_csr(handle, x, y, kt % 10, use_texture=kt >= 10)
            finally:
                spmv.set_tiled_promotion(promotion)
            avg = float(np.mean(t))
            table[f"{name}/{label}"] = {"avg_us": round(avg * 1e3, 1), "min_us": round(float(np.min(t)) * 1e3, 1),
                                        "GBps": round(b / avg / 1e6, 1), "frac": round(b / avg / 1e6 / HBM_PEAK_GBS, 4),
                                        "gflops": round(2.0 * nnz / avg / 1e6, 1)}
        x.release()
        y.release()

    # "+lds_tiles" = SpMVConfig::use_texture (what spmv_auto_config sets for cols > 10000); "vector" / "merge" = what the
    # reference's own callers pass ({VECTOR_CSR, 256, false}, benchmarks/main.cu:52-56): the direct kernel for the first
    # calls, the tiled engine once the matrix has been promoted (the protocol's 5 warm-up calls are past that point);
    # "..._direct" = the same call with promotion off
    table["note"] = ("vector / merge rows: SpMVConfig without use_texture, measured after the library's promotion to the LDS-tiled "
                     "engine (spmv_set_tiled_promotion, default after 4 calls); *_direct rows: promotion off")
    run("c5_10Mx16", engine._A, n, n, n * k, [(11, "vector+lds_tiles"), (1, "vector"), (2, "merge"), (1, "vector_direct"), (2, "merge_direct")])
    A = wl.uniform_csr_device(seed, 1_000_000, 1_000_000, 16)
    run("c2_1Mx16", A.handle, A.rows, A.cols, A.nnz, [(1, "vector_direct"), (2, "merge_direct"), (0, "scalar"), (1, "vector"), (2, "merge"), (11, "vector+lds_tiles")])
    A.close()
    P = wl.power_law_csr_device(seed, 1_000_000, 1_000_000)
    run("c4_1M_powerlaw_nnz%d" % P.nnz, P.handle, P.rows, P.cols, P.nnz,
        [(2, "merge_direct"), (1, "vector_direct"), (2, "merge"), (12, "merge+lds_tiles")])
    P.close()
    # config 3: ELL 1M x 32 (column-major), built from a uniform CSR on the host side of the C ABI
    E = wl.uniform_ell_device(seed, 1_000_000, 1_000_000, 32)
    x = wl.vector_device(seed, 1, 1_000_000)
    y = spmv.CudaBuffer(1_000_000)
    b = 1_000_000 * 32 * 8 + 1_000_000 * 4 * 2
    for label, tex in (("c3_ell_1Mx32/ell+lds_tiles", True), ("c3_ell_1Mx32/ell", False)):
        t = wl.time_spmv_ell(E, x, y, use_texture=tex)
        avg = float(np.mean(t))
        table[label] = {"avg_us": round(avg * 1e3, 1), "min_us": round(float(np.min(t)) * 1e3, 1),
                        "GBps": round(b / avg / 1e6, 1), "frac": round(b / avg / 1e6 / HBM_PEAK_GBS, 4),
                        "gflops": round(2.0 * 32e6 / avg / 1e6, 1)}
    E.close()
    x.release()
    y.release()
    return table


def parity_report(spmv, wl, seed):
    """Per config and kernel: the backward-error figure the tests gate on (|got - want| <= 1e-5 * max(|want|,
    sum_j |a_ij x_j|)) next to the number of rows that would fail the PLAIN relative form 1e-5 * |want| (rows
    whose terms cancel; any kernel that reorders a row's sum has them on signed data).  Checker: oracle/."""
    oracle = importlib.import_module("oracle")
    out = {}
    promotion = spmv.get_tiled_promotion()
    spmv.set_tiled_promotion(0)            # every row is the kernel it names (a cached plan would otherwise take the vector / merge rows)

    def check(name, A, kernels):
        rp, ci, va = A.to_host()
        x = spmv.synth.vector(seed, 1, A.cols)
        want = oracle.spmv_csr(rp, ci, va, x).astype(np.float64)
        prod = np.abs(va.astype(np.float64) * x.astype(np.float64)[ci])
        rp64 = rp.astype(np.int64)
        abs_sum = np.zeros(rp64.size - 1, dtype=np.float64)
        nonempty = rp64[1:] > rp64[:-1]
        if prod.size and nonempty.any():
            abs_sum[nonempty] = np.add.reduceat(prod, rp64[:-1][nonempty])
        d_x, d_y = spmv.CudaBuffer(A.cols), spmv.CudaBuffer(A.rows)
        d_x.copyFromHost(x, A.cols)
        for kt, label in kernels:
            r = spmv.spmv_csr(A.handle, d_x, d_y, spmv.SpMVConfig(kt % 10, 256, kt >= 10), A.cols)
            assert r.error_code == 0
            got = d_y.copyToHost(A.rows).astype(np.float64)
            diff = np.abs(got - want)
            out[f"{name}/{label}"] = {
                "rows": int(A.rows),
                "worst_backward_error": float(np.max(diff / np.maximum(np.maximum(np.abs(want), abs_sum), 1e-30))),
                "rows_failing_plain_relative_1e-5": int(np.count_nonzero(diff > 1e-5 * np.abs(want))),
                "worst_plain_relative_error": float(np.max(diff / np.maximum(np.abs(want), 1e-30)))}
        d_x.release()
        d_y.release()

    A = wl.uniform_csr_device(seed, 1_000_000, 1_000_000, 16)
    check("c2_1Mx16", A, [(11, "vector+lds_tiles"), (1, "vector"), (2, "merge"), (0, "scalar")])
    A.close()
    P = wl.power_law_csr_device(seed, 1_000_000, 1_000_000)
    check("c4_1M_powerlaw", P, [(12, "merge+lds_tiles"), (2, "merge"), (1, "vector")])
    P.close()
    spmv.set_tiled_promotion(promotion)
    return out


def host_core_share():
    """How many host cores this job may really use: the cgroup's CPU quota when there is one (a GPU box of the pool shows 256
    CPUs and grants 16), else the affinity mask; at most 64 threads (the matrix is 1.3 GB: past that the memory system, not
    the cores, bounds a CSR SpMV)."""
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            share = max(1, int(int(quota) / int(period)))
            return min(share, visible, 64), "the cgroup CPU quota (%d of %d visible CPUs)" % (share, visible)
    except Exception:                                   # noqa: BLE001 - cgroup v1 or no cgroup: fall through
        pass
    return min(visible, 64), "the affinity mask (%d CPUs), capped at 64" % visible


def cpu_baseline(spmv, row_ptrs, cols, vals, n, nnz):
    """The reference's CPU path (src/spmv_cpu.cpp:6-16) on the SAME matrix, one host thread:
    oracle/_ref/ref_cpu (the reference's own sources, kind "reference") when it was shipped,
    else the plain-C restatement in oracle/ (kind "port")."""
    oracle = importlib.import_module("oracle")
    wl = importlib.import_module("gpu-spmv_amd.workloads")
    rp, ci, va = row_ptrs.cpu().numpy(), cols.cpu().numpy(), vals.cpu().numpy()
    x = spmv.synth.vector(42, 1, n)
    reps = 3
    if oracle.have_reference_binary():
        scratch = "/dev/shm" if os.path.isdir("/dev/shm") else None
        best = oracle.reference_time_csr(rp, ci, va, x, reps=reps, scratch_dir=scratch)
        kind = "reference"
    else:
        oracle.spmv_csr(rp, ci, va, x)
        best = 1e30
        for _ in range(reps):
            t0 = time.perf_counter()
            oracle.spmv_csr(rp, ci, va, x)
            best = min(best, time.perf_counter() - t0)
        kind = "port"
    b = csr_bytes(n, n, nnz)
    single = {"value": round(b / best / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": kind,
              "gflops": round(2.0 * nnz / best / 1e9, 3), "seconds_per_spmv": round(best, 4),
              "sample": "the full workload matrix (%d rows, %d entries), best of %d single-thread SpMV passes after 1 warm-up"
                        % (n, nnz, reps),
              "host_cpus_visible": os.cpu_count()}
    # (ii) the same loop over OpenMP static row blocks — the reference has no OpenMP,
    # so this row is a port; threads = the box's CPU share for one GPU (16) or fewer
    threads, share_source = host_core_share()
    y1 = oracle.spmv_csr_parallel(rp, ci, va, x, threads)
    best_par = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        oracle.spmv_csr_parallel(rp, ci, va, x, threads)
        best_par = min(best_par, time.perf_counter() - t0)
    parallel = {"value": round(b / best_par / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
                "gflops": round(2.0 * nnz / best_par / 1e9, 3), "seconds_per_spmv": round(best_par, 4),
                "sample": "same matrix, OpenMP static row blocks, best of %d after 1 warm-up; threads = %s" % (reps, share_source),
                "checksum": float(np.float64(y1.sum(dtype=np.float64)))}
    return single, parallel


def cpu_pagerank_baseline(row_ptrs, cols, vals, n, iterations, gpu_call):
    """The reference's PageRank HOST loop (src/pagerank.cu:50-153: dangling scan, then per iteration the dangling sum, one
    spmv_cpu_csr, the update and the fp32 residual) as restated in oracle/spmv_oracle.c, one host thread, on the same
    column-stochastic matrix, tolerance 0 and max_iterations = the GPU call's iteration count."""
    oracle = importlib.import_module("oracle")
    rp, ci, va = row_ptrs.cpu().numpy(), cols.cpu().numpy(), vals.cpu().numpy()
    t0 = time.perf_counter()
    ranks, done, residual, _ = oracle.pagerank(rp, ci, va, num_cols=n, damping=0.85, tolerance=0.0, max_iterations=int(iterations))
    seconds = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.dangling_mask(rp, ci, va, n)                   # the loop's one-time part (src/pagerank.cu:20-48), timed alone
    setup = time.perf_counter() - t0
    per_iter = max(seconds - setup, 1e-9) / max(done, 1)
    out = {"value": round(1.0 / per_iter, 3), "unit": "iterations/s", "cores": 1, "kind": "port",
           "iterations": int(done), "seconds_total": round(seconds, 3), "seconds_dangling_scan": round(setup, 3),
           "final_residual": residual, "rank_sum": float(ranks.sum(dtype=np.float64)),
           "sample": "oracle_pagerank (the host loop of src/pagerank.cu:50-153 restated in C; src/pagerank.cu itself needs the CUDA "
                     "runtime to link) on the full workload matrix, %d iterations (what the GPU pagerank() call took), tolerance 0; "
                     "value = iterations / (total - dangling scan); rank_sum is what the reference's arithmetic gives at this size "
                     "(src/pagerank.cu:140-150 normalises by an fp32 running sum over n values: at n = 1e7 it stops growing near "
                     "0.93 — SURVEY.md H5; the GPU path sums in double)" % done}
    if gpu_call and gpu_call.get("iterations"):
        out["gpu_iterations_per_s_incl_setup"] = round(gpu_call["iterations"] / max(gpu_call["seconds_total"], 1e-9), 1)
    return out


if __name__ == "__main__":
    main()
