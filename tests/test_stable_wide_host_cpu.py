"""tools.is_stable beyond the 8 x 8 support masks: tap-net_amd/csrc/tap_stable_wide.h (what big.hip, tap_macs3_wave.h
and tap_macs3_big.h call for block sides of 9 .. 16) compiled for the HOST and checked (a) against the reference's own
answers on the sampled patterns of tests/golden/stable3d_wide.npz, (b) against the oracle's hull-based restatement on
random patterns of every footprint up to 16 x 16, and (c) the oracle itself against the fixture.  No GPU needed; the
kernels that call the header are compared with the oracle in tests/test_gpu_parity.py."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def wide(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    so = str(tmp_path_factory.mktemp("sw") / "libsw.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "tap-net_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "stable_wide_host.cpp"), "-o", so])
    lib = C.CDLL(so)
    lib.sw_is_stable.restype = C.c_int
    lib.sw_is_stable.argtypes = [C.c_int, C.c_int, C.c_void_p]
    return lambda bx, by, m: lib.sw_is_stable(bx, by, np.ascontiguousarray(m, dtype=np.uint8).ctypes.data_as(C.c_void_p))


def _oracle(bx, by, m):
    m = np.ascontiguousarray(m, dtype=np.uint8)
    return O.lib().orc_is_stable_3d_mask(bx, by, m.ctypes.data_as(C.c_void_p))


def test_reference_patterns(wide):
    z = G.load("stable3d_wide.npz")
    n = 0
    for si, (bx, by) in enumerate(z["shapes"]):
        bx, by = int(bx), int(by)
        masks = np.unpackbits(z["m%d" % si], axis=1)[:, :bx * by]
        for m, want in zip(masks, z["r%d" % si]):
            assert _oracle(bx, by, m) == int(want), (bx, by, m.reshape(bx, by))
            assert wide(bx, by, m) == int(want), (bx, by, m.reshape(bx, by))
            n += 1
    assert n == 4000


def test_random_patterns_against_the_oracle(wide):
    rng = np.random.RandomState(3)
    for bx in range(1, 17):
        for by in range(1, 17):
            cells = bx * by
            for rep in range(60):
                k = rng.randint(0, cells // 2 + 2)
                m = np.zeros(cells, np.uint8)
                if rep % 5 == 0 and by > 1:                   # a single column / row / diagonal: the collinear branch
                    m.reshape(bx, by)[rng.choice(bx, size=min(max(k, 2), bx), replace=False), rng.randint(by)] = 1
                elif rep % 5 == 1 and bx > 1:
                    m.reshape(bx, by)[rng.randint(bx), rng.choice(by, size=min(max(k, 2), by), replace=False)] = 1
                else:
                    m[rng.choice(cells, size=min(k, cells), replace=False)] = 1
                assert wide(bx, by, m) == _oracle(bx, by, m), (bx, by, m.reshape(bx, by))
