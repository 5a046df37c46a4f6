#!/usr/bin/env python3
"""On-box bandwidth calibration for the roofline figures: libtapenv's probe kernels (the hot kernels'
own access shape: 16 B per lane, plain / nontemporal stores) at 20 MB / 160 MB / 1.2 GB.

    python scripts/calibrate_bw.py [--out profiles/r03_bw_calibration.json]

Every figure is the median over replays of a hipGraph holding `launches` back-to-back probe launches (so the number
includes the gap between graph nodes, like a pass of bench.py).  `hot` = the same buffers every launch (a 20 MB
buffer then lives in the 256 MB Infinity Cache, which is the regime of the BASELINE batch); `cold` = each launch
of the graph works on its own buffers, >= 1.2 GB in total, so nothing is cache-resident.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tap_net_amd as T          # noqa: E402
from tap_net_amd import _lib     # noqa: E402

KINDS = {0: "copy", 1: "copy_nt", 2: "fill", 3: "fill_nt", 4: "read", 5: "fill_wt", 6: "fill_wt_slab_shape", 7: "fill_wt_wave_linear", 8: "fill_wt_run_of_rows"}


def run(kind, nbytes, cold, dev, reps=30, slots=None):
    n = nbytes // 4
    if slots is None:
        slots = max(2, int(1.3e9 // nbytes)) if cold else 1
    slots = min(slots, 64)
    src = [torch.rand(n, device=dev) for _ in range(slots)] if kind in (0, 1, 4) else [None] * slots
    dst = [torch.empty(n, device=dev) for _ in range(slots)] if kind != 4 else [None] * slots
    launches = slots if cold else 16
    c = _lib.ctx(dev)
    L = _lib.lib()

    def body():
        st = _lib.stream_of(dev)
        for i in range(launches):
            k = i % slots
            _lib.check(L.tap_bw_probe(c, kind, _lib.ptr(dst[k]), _lib.ptr(src[k]), C.c_size_t(n * 4), st), c)

    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3 / launches)
    us = statistics.median(times)
    moved = n * 4 * (2 if kind in (0, 1) else 1)
    if kind in (0, 1):      # correctness of the probe itself
        assert torch.equal(dst[0], src[0])
    return dict(kind=KINDS[kind], bytes=n * 4, cold=bool(cold), slots=slots, launches_per_graph=launches,
                us_per_launch=round(us, 3), us_min=round(min(times), 3), GBps=round(moved / us / 1e3, 1),
                bytes_moved_per_launch=moved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--pingpong-sizes", default="",
                    help="comma-separated bytes: fill / nontemporal fill alternating between TWO buffers of this size "
                         "(the bench sweep's `dynamic` ping-pong at large batch: 2.5 GB per buffer at B = 1 M, c2 shape)")
    ap.add_argument("--store-shapes", action="store_true",
                    help="19.66 MB (the c2 step's tensor): linear fill with plain / nontemporal / write-through stores against "
                         "the bit-shadow expansion's own store shape (kinds 6, 7 of tap_bw_probe)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rows = []
    if args.store_shapes:
        for kind in (2, 3, 5, 6, 7, 8):
            for cold in (False, True):
                r = run(kind, 19_660_800, cold, dev)
                rows.append(r)
                print(json.dumps(r), flush=True)
        if args.out:
            with open(args.out, "w") as f:
                json.dump(dict(device=torch.cuda.get_device_name(0), when=time.strftime("%Y-%m-%d %H:%M:%S"), rows=rows), f, indent=1)
        return
    if args.pingpong_sizes:
        for nbytes in (int(x) for x in args.pingpong_sizes.split(",")):
            for kind in (2, 3):
                r = run(kind, nbytes, True, dev, reps=10, slots=2)
                r["pingpong"] = True
                rows.append(r)
                print(json.dumps(r), flush=True)
        if args.out:
            with open(args.out, "w") as f:
                json.dump(dict(device=torch.cuda.get_device_name(0), when=time.strftime("%Y-%m-%d %H:%M:%S"), rows=rows), f, indent=1)
        return
    for nbytes in (19_660_800, 160_000_000, 1_200_000_000):
        for kind in (0, 1, 2, 3, 4):
            for cold in ((False, True) if nbytes < 1e9 else (False,)):
                r = run(kind, nbytes, cold, dev)
                rows.append(r)
                print(json.dumps(r), flush=True)
    best = lambda k, pred: max((r["GBps"] for r in rows if r["kind"] in k and pred(r)), default=None)  # noqa: E731
    big = lambda r: r["bytes"] >= 1e9 or r["cold"]                                                     # noqa: E731
    summary = dict(
        device=torch.cuda.get_device_name(0), when=time.strftime("%Y-%m-%d %H:%M:%S"),
        hbm_peak_spec_GBps=8000.0,
        copy_GBps_beyond_cache=best(("copy", "copy_nt"), big),
        fill_GBps_beyond_cache=best(("fill", "fill_nt"), big),
        read_GBps_beyond_cache=best(("read",), big),
        copy_GBps_20MB_hot=best(("copy", "copy_nt"), lambda r: r["bytes"] < 3e7 and not r["cold"]),
        fill_GBps_20MB_hot=best(("fill", "fill_nt"), lambda r: r["bytes"] < 3e7 and not r["cold"]),
        fill_GBps_20MB_cold=best(("fill", "fill_nt"), lambda r: r["bytes"] < 3e7 and r["cold"]),
        rows=rows)
    print(json.dumps({k: v for k, v in summary.items() if k != "rows"}))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
