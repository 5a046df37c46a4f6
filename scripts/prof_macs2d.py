#!/usr/bin/env python3
"""Phase clocks of the MACS 2D placement (tap_macs.h, -DTAP_PROF marks M2_PROF) on the stand-alone step at c4's
shape: cycles per phase of the first lane group of every workgroup, at a few steps of a 20-block episode.
DESIGN.md section 4 (tap_macs_place row) quotes these.

Build recipe (the product build has no clocks in it):
    cd tap-net_amd/csrc && mkdir -p ../../build_prof && \
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -I../../include -I. -DTAP_PROF \
          -c macs.hip -o ../../build_prof/macs.o && \
    hipcc --offload-arch=gfx950 -shared $(ls build/*.o | grep -v /macs.o) ../../build_prof/macs.o -o ../../build_prof/libtapenv.so
Run (on the GPU box):  TAP_LIB_PATH=build_prof/libtapenv.so python scripts/prof_macs2d.py
"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_net_amd as T
from tap_net_amd import _lib
DEV = "cuda:0"
B, n = 8192, 20
rng = np.random.RandomState(3)
blocks = torch.as_tensor(rng.randint(1, 5, size=(B, n, 2)).astype(np.int32), device=DEV)
env = T.BatchedContainer(B, [7, 100], n, "C+P+S-mcs-soft", "diff", packing_strategy="MACS", device=DEV)
names = ["1a levels", "1b tops", "2 walks", "3 score", "3 tie-break", "commit", "n_ems", "n_slots"]
L = _lib.lib()
buf = (C.c_uint * (8192 * 8))()
for rep in range(2):
    env.reset()
    for t in range(n):
        env.add_new_blocks(blocks[:, t].contiguous())
        torch.cuda.synchronize()
        if rep == 1 and t in (1, 5, 10, 15, 19):
            L.tap_prof_read_macs2(buf)
            a = np.frombuffer(buf, dtype=np.uint32).reshape(8192, 8)[: B // 32].astype(np.int64)
            print("step %2d: total %6.0f  " % (t, a[:, :6].sum(1).mean()) + "  ".join("%s %5.0f" % (names[i], a[:, i].mean()) for i in range(8)))
