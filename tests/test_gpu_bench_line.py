"""bench.py end to end on the GPU box (small matrix): stdout is exactly one JSON line with the contract's keys, and the
line survives a failing extra and a failing exchange trial (protocol of the reference's
include/spmv/benchmark.h:34-40: warm-up, timed iterations, one record out)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--rows", "400000", "--steps", "3", "--warmup", "1"]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def one_line(out):
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_one_line_with_the_contract_keys(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras"] + SMALL,
                         capture_output=True, text=True, timeout=900)
    line = one_line(out)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["steps"] == 3 and line["n_gpus"] == 1 and line["value"] > 0
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] < 1 and roof["peak"] == 8000.0
    assert "incomplete" not in line and "provisional" not in line


@pytest.mark.gpu
def test_an_extra_that_kills_the_process_costs_the_extra_only(gpu):
    env = dict(os.environ, SPMV_BENCH_FAIL_IN="cpu_baseline:abort")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True,
                         timeout=900, env=env)
    line = one_line(out)
    assert line["value"] > 0 and "incomplete" in line and line["roofline"]["frac"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["raise", "abort"])
def test_two_ranks_a_failing_exchange_trial_still_leaves_the_line(gpu, how):
    """Two ranks sharing the one GPU over gloo (a rehearsal of the N > 1 code path; its number means nothing)."""
    env = dict(os.environ, SPMV_BENCH_BACKEND="gloo", SPMV_PR_OVERLAP="2", SPMV_BENCH_FAIL_IN="trial:" + how)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL,
                         capture_output=True, text=True, timeout=1200, env=env)
    line = one_line(out)
    assert line["n_gpus"] == 2 and line["config"]["exchange"] == "gather" and line["value"] > 0


@pytest.mark.gpu
def test_plain_launch_with_two_gpus_starts_its_own_launcher(gpu):
    """`python bench.py --gpus 2` the way the driver starts `--gpus 1` (no torch.distributed.run in front, no rank variables):
    the supervisor starts the launcher itself and exactly one line with n_gpus = 2 comes out."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SPMV_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-extras"] + SMALL,
                         capture_output=True, text=True, timeout=1200, env=env)
    line = one_line(out)
    assert line["n_gpus"] == 2 and line["config"]["exchange"] == "gather" and line["value"] > 0
    assert line["child_exit"] == "exit code 0" and "incomplete" not in line
    assert "starting torch.distributed.run" in out.stderr
