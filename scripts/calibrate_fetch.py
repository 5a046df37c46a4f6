#!/usr/bin/env python3
"""What rocprofv3's FETCH_SIZE tallies for SPARSE reads on this part (the guide's "x 2" is calibrated on wide coalesced
reads only; the rolling step's excess traffic is scattered 8 / 32 / 12-byte records).  Launches libtapenv's sparse-read
probes (tap_bw_probe kinds 9 .. 13: one record per stride, every record once) and the dense read (kind 4) over a region
several times the Infinity Cache; run it under a FETCH_SIZE pass and condense:

    cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o fetch -- python scripts/calibrate_fetch.py run
    python scripts/calibrate_fetch.py report out/fetch_counter_collection.csv [profiles/rNN_fetch_calibration.json]
"""
import csv, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REGION = 1 << 30                                        # 1 GiB, four times the Infinity Cache
KINDS = {4: ("dense 16 B per lane", None, None), 9: ("8 of every 128 B", 8, 128), 10: ("32 of every 128 B", 32, 128),
         11: ("32 of every 64 B", 32, 64), 12: ("64 of every 128 B", 64, 128), 13: ("32 of every 256 B", 32, 256)}
NAMES = {"k_bw_probe<4>": 4, "k_fetch_probe<8, 128>": 9, "k_fetch_probe<32, 128>": 10, "k_fetch_probe<32, 64>": 11,
         "k_fetch_probe<64, 128>": 12, "k_fetch_probe<32, 256>": 13}


def run():
    import torch
    sys.path.insert(0, ROOT)
    import tap_net_amd as T   # noqa: F401
    from tap_net_amd import _lib
    dev = torch.device("cuda:0")
    src = torch.rand(REGION // 4 + 64, device=dev)
    off = (-src.data_ptr()) % 256 // 4
    src = src[off: off + REGION // 4]
    c, L, st = _lib.ctx(dev), _lib.lib(), _lib.stream_of(dev)
    for rep in range(3):
        for kind in KINDS:
            _lib.check(L.tap_bw_probe(c, kind, None, _lib.ptr(src), C.c_size_t(REGION), st), c)
    torch.cuda.synchronize()
    if os.environ.get("FETCHCAL_TIME"):                   # (not under the profiler) event times per launch
        for kind, (what, touch, stride) in KINDS.items():
            ts = []
            for rep in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(L.tap_bw_probe(c, kind, None, _lib.ptr(src), C.c_size_t(REGION), st), c)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            us = sorted(ts)[len(ts) // 2]
            print(json.dumps(dict(kind=kind, reads=what, us=round(us, 1), region_GBps=round(REGION / us / 1e3, 1),
                                  lines128_GBps=round((REGION if not stride or stride <= 128 else REGION // (stride // 128)) / us / 1e3, 1))))


def report(path, out=None):
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != "FETCH_SIZE":
                continue
            for name, kind in NAMES.items():
                if name in r["Kernel_Name"]:
                    rows.setdefault(kind, []).append(float(r["Counter_Value"]))
    res = []
    for kind, (what, touch, stride) in KINDS.items():
        v = rows.get(kind)
        if not v:
            continue
        raw = sum(v) / len(v) * 1024.0                                   # FETCH_SIZE is in KiB
        rec = REGION // stride if stride else None
        res.append(dict(kind=kind, reads=what, launches=len(v), fetch_size_raw_bytes=raw,
                        bytes_touched=(rec * touch if rec else REGION),
                        raw_per_record=(raw / rec if rec else None),
                        raw_over_touched=raw / (rec * touch if rec else REGION),
                        raw_over_region=raw / REGION))
    doc = dict(script="scripts/calibrate_fetch.py", region_bytes=REGION,
               how="rocprofv3 --pmc FETCH_SIZE (KiB, raw: no x 2), mean over the launches of each probe; one record per stride, "
                   "every record read once per launch, the region four times the Infinity Cache", rows=res)
    print(json.dumps(doc, indent=1))
    if out:
        json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "report":
        report(*sys.argv[2:4])
    else:
        run()
