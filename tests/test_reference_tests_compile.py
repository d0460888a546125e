"""The drop-in boundary, checked by the reference's own callers: every file of /root/reference/tests compiles
(-fsyntax-only) against include/spmv/*.h (tools/check_reference_tests_compile.sh; nothing of the reference is copied).
Runs in the build container only — the reference does not travel to the GPU box."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests"), reason="the reference tree is not on this machine")
def test_the_references_own_tests_compile_against_our_headers():
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "check_reference_tests_compile.sh")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok ") == 8 and "FAILED" not in out.stdout, out.stdout
