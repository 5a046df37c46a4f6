#!/usr/bin/env python3
"""Phase clocks of the MACS 3D placement (tap_macs3.h, -DTAP_PROF marks M3_PROF) on the stand-alone step at c6's
shape (5x5x50, B = 4096).  DESIGN.md section 9 (c6) quotes these.

Build recipe (the product build has no clocks in it):
    cd tap-net_amd/csrc && mkdir -p ../../build_prof && \
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -I../../include -I. -DTAP_PROF \
          -c macs.hip -o ../../build_prof/macs.o && \
    hipcc --offload-arch=gfx950 -shared $(ls build/*.o | grep -v /macs.o) ../../build_prof/macs.o -o ../../build_prof/libtapenv.so
Run (on the GPU box):  TAP_LIB_PATH=build_prof/libtapenv.so python scripts/prof_macs3d.py
"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_net_amd as T
from tap_net_amd import _lib
DEV = "cuda:0"
B, n = 4096, 10
rng = np.random.RandomState(3)
blocks = torch.as_tensor(rng.randint(1, 5, size=(B, n, 3)).astype(np.int32), device=DEV)
env = T.BatchedContainer(B, [5, 5, 50], n, "C+P+S-mcs-soft", "diff", packing_strategy="MACS", device=DEV)
names = {11: "scan+stab", 12: "maskbuild(last round)", 0: "1a lists", 8: "1b sides", 9: "1b offsets", 10: "1b tops", 1: "1b append", 2: "walks", 5: "score", 6: "tie table", 7: "tie sum", 3: "select", 4: "commit"}
L = _lib.lib()
buf = (C.c_uint * (8192 * 16))()
for rep in range(2):
    env.reset()
    for t in range(n):
        env.add_new_blocks(blocks[:, t].contiguous())
        torch.cuda.synchronize()
        if rep == 1:
            L.tap_prof_read_macs3(buf)
            a = np.frombuffer(buf, dtype=np.uint32).reshape(8192, 16)[: B // 8].astype(np.int64)   # 8 envs per WG (G=32, 256 threads)
            order = [0, 8, 9, 10, 1, 11, 12, 2, 5, 6, 7, 3, 4]
            tot = a[:, order].sum(1)
            print("step %d: total %6.0f cycles (median %6.0f)  " % (t, tot.mean(), np.median(tot)) +
                  "  ".join("%s %5.0f" % (names[i], a[:, i].mean()) for i in order))
