"""bench.py's supervisor (no GPU needed): whatever ends the measuring child after its measurement — an exception,
an abort, a hang — exactly one JSON line comes out and the exit code is 0 ("make bench.py unable to
lose the line").  The child here is a stand-in that speaks the child's protocol on stdout."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_supervisor(tmp_path, child_body, env=None, timeout=60):
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent(child_body))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        bench.supervise([sys.executable, {str(child)!r}])
    """))
    full_env = dict(os.environ, **(env or {}))
    return subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=timeout, env=full_env)


LINE = {"metric": "spmv_effective_bandwidth", "value": 1.0, "roofline": {"frac": 0.3}}


def test_the_last_line_wins_and_the_exit_code_is_zero(tmp_path):
    out = run_supervisor(tmp_path, f"""
        import json, sys
        line = {LINE!r}
        print(json.dumps(dict(line, provisional=True)), flush=True)
        print("a library banner on stdout", flush=True)
        print(json.dumps(dict(line, cpu_baseline={{"value": 2.0}})), flush=True)
    """)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1
    got = json.loads(lines[0])
    assert got["cpu_baseline"] == {"value": 2.0} and "provisional" not in got and "incomplete" not in got
    assert "a library banner" in out.stderr


@pytest.mark.parametrize("ending", ["raise RuntimeError('extra failed')", "import os; os.abort()"])
def test_a_child_that_dies_in_an_extra_keeps_its_measurement(tmp_path, ending):
    out = run_supervisor(tmp_path, f"""
        import json, sys
        print(json.dumps(dict({LINE!r}, provisional=True)), flush=True)
        {ending}
    """)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1
    got = json.loads(lines[0])
    assert got["value"] == 1.0 and "provisional" not in got and "incomplete" in got


def test_a_hung_extra_is_cut_off_at_the_deadline(tmp_path):
    out = run_supervisor(tmp_path, f"""
        import json, sys, time
        print(json.dumps(dict({LINE!r}, provisional=True)), flush=True)
        time.sleep(600)
    """, env={"SPMV_BENCH_EXTRAS_DEADLINE": "1"})
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip())
    assert "did not finish" in got["incomplete"]


def test_no_measurement_no_line_and_a_failing_exit_code(tmp_path):
    out = run_supervisor(tmp_path, "raise SystemExit(3)")
    assert out.returncode == 3 and out.stdout.strip() == ""


def test_other_ranks_exit_zero_once_they_have_measured(tmp_path):
    out = run_supervisor(tmp_path, """
        import os
        print("MEASURED", flush=True)
        os.abort()
    """, env={"RANK": "1"})
    assert out.returncode == 0 and out.stdout.strip() == ""


def test_a_hang_behind_the_final_line_is_cut_off_and_the_line_is_complete(tmp_path):
    out = run_supervisor(tmp_path, f"""
        import json, sys, time
        print(json.dumps({LINE!r}), flush=True)
        time.sleep(600)
    """, env={"SPMV_BENCH_EXTRAS_DEADLINE": "1"})
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip())
    assert got["value"] == 1.0 and "incomplete" not in got


def test_a_child_that_never_measures_is_cut_off_by_the_supervisors_own_clock(tmp_path):
    """No signal from outside: the supervisor's deadline fires by itself (a SIGKILL at the caller's limit must
    never be what ends the run), the child is killed, no line, non-zero exit."""
    out = run_supervisor(tmp_path, """
        import time
        time.sleep(600)
    """, env={"SPMV_BENCH_MEASURE_DEADLINE": "1"}, timeout=60)
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert "no measurement within" in out.stderr


def test_the_runs_budget_bounds_the_extras_without_any_signal(tmp_path):
    out = run_supervisor(tmp_path, f"""
        import json, sys, time
        print(json.dumps(dict({LINE!r}, provisional=True)), flush=True)
        time.sleep(600)
    """, env={"SPMV_BENCH_BUDGET": "22"}, timeout=60)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip())
    assert "budget" in got["incomplete"] and "killed" in got["child_exit"]


def test_how_the_child_ended_is_in_the_line_even_behind_a_final_one(tmp_path):
    out = run_supervisor(tmp_path, f"""
        import json, os
        print(json.dumps({LINE!r}), flush=True)
        os.abort()
    """)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip())
    assert got["child_exit"] == "signal SIGABRT" and "incomplete" not in got
    assert "SIGABRT" in out.stderr


def test_gpus_n_without_rank_variables_starts_the_launcher(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_supervise(cmd=None, self_launched=False):
        seen["cmd"], seen["self_launched"] = cmd, self_launched
        raise SystemExit(0)

    monkeypatch.setattr(bench, "supervise", fake_supervise)
    for name in ("SPMV_BENCH_CHILD", "SPMV_BENCH_INPROCESS", "WORLD_SIZE", "RANK", "LD_PRELOAD", "HSA_TOOLS_LIB"):
        monkeypatch.delenv(name, raising=False)
    monkeypatch.setattr(bench, "_under_profiler", lambda: False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    with pytest.raises(SystemExit):
        bench.main()
    cmd = seen["cmd"]
    assert seen["self_launched"] and cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    # with a launcher's rank variables in place nothing changes: the plain supervisor
    monkeypatch.setenv("WORLD_SIZE", "8")
    with pytest.raises(SystemExit):
        bench.main()
    assert seen["cmd"] is None and not seen["self_launched"]


@pytest.mark.parametrize("name,value", [("HSA_TOOLS_LIB", "/opt/rocm/lib/librocprofiler-sdk-tool.so"),
                                        ("LD_PRELOAD", "/usr/lib/libomnitrace-dl.so"),
                                        ("ROCP_TOOL_LIBRARIES", "x.so"), ("ROCPROF_COUNTERS", "1")])
def test_tool_libraries_that_initialise_the_gpu_keep_the_run_in_process(monkeypatch, name, value):
    sys.path.insert(0, ROOT)
    import bench
    for var in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCP_TOOL_LIB"):
        monkeypatch.delenv(var, raising=False)
    for var in [k for k in os.environ if k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_"))]:
        monkeypatch.delenv(var, raising=False)
    assert not bench._under_profiler()
    monkeypatch.setenv(name, value)
    assert bench._under_profiler()


def test_cpu_pagerank_baseline_runs_the_reference_host_loop_for_the_asked_iterations():
    """bench.py's `cpu_baseline.pagerank`: the oracle's restatement of the reference's host loop,
    tolerance 0, exactly as many iterations as the GPU call took; iterations/s excludes the one-time dangling scan."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    n, k = 2000, 6
    rng = np.random.default_rng(3)
    cols = np.sort(rng.integers(0, n, size=(n, k)), axis=1).astype(np.int32).ravel()
    counts = np.bincount(cols, minlength=n).astype(np.float32)
    vals = (1.0 / counts[cols]).astype(np.float32)
    row_ptrs = (np.arange(n + 1) * k).astype(np.int32)
    out = bench.cpu_pagerank_baseline(torch.from_numpy(row_ptrs), torch.from_numpy(cols), torch.from_numpy(vals), n, 3,
                                      {"iterations": 3, "seconds_total": 0.0025})
    assert out["iterations"] == 3 and out["cores"] == 1 and out["unit"] == "iterations/s" and out["value"] > 0
    assert out["kind"] == "port" and abs(out["rank_sum"] - 1.0) < 1e-3 and out["gpu_iterations_per_s_incl_setup"] == 1200.0
