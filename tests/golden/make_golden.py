"""make_golden.py — regenerates tests/golden/ref_cases.npz.

Runs the REFERENCE's own CPU path (csr_from_dense, ell_from_csr / ell_from_dense,
spmv_cpu_csr, spmv_cpu_ell, csr_compute_stats, spmv_auto_config, csr/ell_serialize —
compiled from /root/reference by oracle/Makefile into oracle/_ref/ref_cpu) on a fixed
set of inputs and stores inputs + every output.  The fixture is data only; it lets the
GPU box (which has no /root/reference) check the oracle restatement and the library
against real reference outputs.

usage (in the build container):  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

CASES = [
    # name, rows, cols, density, value range
    ("tiny_3x3_readme", None, None, None, None),
    ("design_doc_3x4", None, None, None, None),
    ("single_5", None, None, None, None),
    ("zero_row", None, None, None, None),
    ("all_zero_3x3", None, None, None, None),
    ("r37_c41_d30", 37, 41, 0.30, 10.0),
    ("r200_c200_d05", 200, 200, 0.05, 10.0),
    ("r128_c64_d15", 128, 64, 0.15, 10.0),
    ("r1_c150_d20", 1, 150, 0.20, 10.0),
    ("r150_c1_d50", 150, 1, 0.50, 10.0),
    ("r97_c193_d01", 97, 193, 0.01, 10.0),
    ("r64_c300_skewed", 64, 300, None, 10.0),
    ("r300_c280_short", 300, 280, 0.008, 1.0),
]


def build_input(name, rows, cols, density, scale, rng):
    if name == "tiny_3x3_readme":            # reference README.md:75-99
        return np.array([[1, 0, 2], [0, 3, 4], [5, 0, 0]], np.float32), np.ones(3, np.float32)
    if name == "design_doc_3x4":             # .kiro/specs/spmv-gpu/design.md:372-385
        return np.array([[1, 0, 2, 0], [0, 3, 4, 0], [0, 0, 0, 5]], np.float32), np.array([1, 2, 3, 4], np.float32)
    if name == "single_5":                   # tests/test_spmv.cu:161-186
        return np.array([[5.0]], np.float32), np.array([2.0], np.float32)
    if name == "zero_row":                   # tests/test_spmv.cu:188-218
        return np.array([[1, 2, 0], [0, 0, 0], [3, 0, 4]], np.float32), np.ones(3, np.float32)
    if name == "all_zero_3x3":               # tests/test_csr.cpp:139-151
        return np.zeros((3, 3), np.float32), np.ones(3, np.float32)
    if name == "r64_c300_skewed":
        dense = np.zeros((rows, cols), np.float32)
        dense[rng.random((rows, cols)) < 0.02] = 1.0
        dense[5, :] = 1.0                      # one full row => skewness >= 10
        dense *= rng.uniform(-scale, scale, size=(rows, cols)).astype(np.float32)
        return dense, rng.uniform(-scale, scale, cols).astype(np.float32)
    mask = rng.random((rows, cols)) < density
    vals = rng.uniform(-scale, scale, size=(rows, cols)).astype(np.float32)
    vals[vals == 0] = 1.0
    return np.where(mask, vals, np.float32(0)).astype(np.float32), rng.uniform(-scale, scale, cols).astype(np.float32)


def main():
    oracle.build()
    assert oracle.have_reference_binary(), "oracle/_ref/ref_cpu missing (needs /root/reference)"
    rng = np.random.default_rng(42)
    bundle = {"case_names": np.array([c[0] for c in CASES])}
    for name, rows, cols, density, scale in CASES:
        dense, x = build_input(name, rows, cols, density, scale, rng)
        out = oracle.reference_case(dense, x)
        bundle[f"{name}/dense"] = dense
        bundle[f"{name}/x"] = x
        for key, value in out.items():
            bundle[f"{name}/{key}"] = value
    path = os.path.join(ROOT, "tests", "golden", "ref_cases.npz")
    np.savez_compressed(path, **bundle)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
