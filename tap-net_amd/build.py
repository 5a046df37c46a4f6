"""Compile libtapenv.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
TARGET = os.path.join(_HERE, "libtapenv.so")


def needs_build():
    if not os.path.exists(TARGET):
        return True
    t = os.path.getmtime(TARGET)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "tapenv.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if force or needs_build():
        cmd = ["make", "-C", CSRC] + (["-B"] if force else [])
        subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return TARGET
