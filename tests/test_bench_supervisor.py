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
