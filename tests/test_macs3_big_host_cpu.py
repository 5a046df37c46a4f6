"""The serial MACS / MUL 3D core for containers above 64 cells (tap-net_amd/csrc/tap_macs3_big.h -- what the
one-thread-per-container fallback kernel runs, and the control skeleton the wave-per-container kernel follows
statement for statement) compiled for the HOST with g++ and stepped against the CPU oracle on random blocks:
tests/host/m3b_host.cpp compares positions, stable flags, the height-map and the valid / empty counters after every
step, and that both sides raise on the same step.  Runs without a GPU; the GPU kernels built on this header are compared
with the same oracle by tests/test_gpu_parity.py::test_macs3d_big_containers and the fallback test next to it."""
import os
import shutil
import subprocess

import pytest

from oracle_lib import ORACLE_DIR, build as build_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# W, L, H, n, episodes, flags (1 hard, 2 P, 4 S, 8 zero ratio = 'mul', 16 tie-break = 'mcs'), block side bound, height bound, seed
CASES = [(10, 10, 50, 12, 150, 22, 6, 6, 1), (12, 9, 40, 16, 120, 23, 7, 5, 2), (9, 16, 60, 20, 80, 22, 5, 7, 3),
         (3, 40, 30, 14, 80, 30, 9, 4, 4), (20, 20, 30, 30, 25, 22, 9, 5, 5), (10, 10, 14, 30, 80, 22, 6, 6, 6),   # the last one overflows
         (16, 5, 200, 40, 40, 19, 5, 9, 7), (11, 11, 50, 12, 100, 14, 6, 6, 8),
         # block sides of 9 .. 16 (round 5: tap_stable_wide.h beyond the 8 x 8 support masks)
         (20, 20, 40, 14, 60, 22, 13, 5, 9), (24, 18, 40, 12, 40, 23, 17, 5, 10), (30, 12, 50, 16, 40, 30, 17, 6, 11)]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    build_oracle()
    exe = str(tmp_path_factory.mktemp("m3b") / "m3b_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fno-fast-math", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "tap-net_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-I" + ORACLE_DIR,
                           os.path.join(ROOT, "tests", "host", "m3b_host.cpp"), "-L" + ORACLE_DIR, "-ltap_oracle",
                           "-Wl,-rpath," + ORACLE_DIR, "-o", exe])
    return exe


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%dx%d-n%d-f%d" % (c[0], c[1], c[2], c[3], c[5]))
def test_serial_macs3d_core_on_the_host_against_the_oracle(harness, case):
    p = subprocess.run([harness] + [str(v) for v in case], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:]
    assert " 0 mismatching episodes" in p.stdout, p.stdout[-2000:]
