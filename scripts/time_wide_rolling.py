#!/usr/bin/env python3
"""Two fused rolling episodes over 100-block instances (two-word graphs, one wavefront per instance) for a kernel trace:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wide -o wide -- python scripts/time_wide_rolling.py
(k_rolling_step<3,32,-2> is the row to read; DESIGN 4, rolling_window_wave2)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tap_net_amd as T                      # noqa: E402
from tap_net_amd import generate             # noqa: E402
DEV = "cuda:0"
B, N, D, init = 8192, 100, 3, [7, 7, 500]
g = torch.Generator(device=DEV); g.manual_seed(1)
blocks = torch.randint(1, 5, (B, N, D), device=DEV, generator=g, dtype=torch.int32)
positions, _, _ = generate.pack_blocks(blocks, init, 'C+P+S-lb-soft')
gen = torch.Generator(device=DEV)
def policy(step, static, dynamic, current_mask, **_):
    return torch.multinomial(current_mask, 1, generator=gen).squeeze(1)
for rep in range(2):
    gen.manual_seed(7)
    out = T.run_rolling_episode(blocks, positions, init, policy, 5, 4 * N + 10, child_graph_size=10, fused=True)
torch.cuda.synchronize()
print("reward mean", float(out["reward"].mean()))
