"""Is c5's 495 / 545 M bimodality (constant inside a process, different between processes) a matter of where the hot
buffers landed?  Build the pass several times in ONE process, shifting the allocator between builds, and time each."""
import os, sys, json, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # scripts/ -> repo root
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = bench.CONFIGS["c5"]
name, D, cs, n, B, reward, strategy = cfg
keep = []
first = None
for k in range(7):
    if k:
        keep.append(torch.empty((3 + 5 * k) * 1024 * 1024 + 4096 * k, dtype=torch.uint8, device=dev))   # shift what follows
    if k < 4 or first is None:
        hp = bench.RollingHotPath(cfg, B, 0, dev, window=bench.WINDOW["c5"], fused_rolling=True, mix=True)
        hp.config = "c5"
    dt, graphs = bench.time_passes(hp, 20, 3, True, 1, repeats=5)
    addrs = {nm: getattr(hp, nm) for nm in ("static", "static2", "cur", "feat", "state")}
    a = {nm: hex(t.data_ptr()) for nm, t in addrs.items()}
    a["dyn2"] = hex(hp.dyn[2].data_ptr())
    print(json.dumps(dict(build=k, rebuilt=(k < 4), value_M=round(B * n * 20 / dt / 1e6, 1), addrs=a)), flush=True)
    if k >= 3:
        first = hp            # builds 4..6: the SAME buffers timed again (graphs re-captured) -> is it the buffers or the moment?
