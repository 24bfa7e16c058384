"""
K_lik's table-driven exp(-y) and log (fcdiff_amd/csrc/fcd_fastmath.h, shared by host and device) against long-double
libm on the host: the header is plain C++ under g++, so its accuracy claim (<= 1.5 ulp on the ranges the kernel feeds
it) is checked here without a GPU.  The kernel's own outputs are checked against the reference fixtures and the oracle
in test_gpu_parity.py.
"""
import os
import subprocess

from conftest import ROOT


def test_fastmath_header_against_libm(tmp_path):
    exe = str(tmp_path / "fastmath_check")
    subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(ROOT, "tests", "fastmath_check.cpp"), "-lm"])
    out = subprocess.check_output([exe, "3000000"]).decode().split()
    (worst_exp, worst_log, edge_ok) = (float(out[0]), float(out[1]), int(out[2]))
    assert worst_exp <= 1.5, worst_exp          # fcd_exp_neg: measured 1.005 ulp
    assert worst_log <= 1.5, worst_log          # fcd_log_normal: measured 1.27 ulp
    assert edge_ok == 1
