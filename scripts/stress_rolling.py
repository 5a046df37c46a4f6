"""Randomised sweep of the instance generator and the rolling-window path on the GPU, re-using the
oracle comparisons of tests/test_gpu_parity.py with random shapes.

    python scripts/stress_rolling.py 60
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tap_net_amd as T                                                   # noqa: E402
import test_gpu_parity as P                                               # noqa: E402

t0 = time.time(); done = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    rs = np.random.RandomState(9000 + seed)
    D = 2 + seed % 2
    N = int(rs.randint(12, 65))
    child = int(rs.randint(3, min(N, 22)))
    W = int(rs.randint(5, 9))
    init = [W, 250] if D == 2 else [W, int(rs.randint(5, 9)), 250]
    if D == 3:
        init[1] = init[0]                                                  # generate_instances builds square 3D containers
    B = int(rs.choice([32, 64, 100]))
    cfg = (D, N, child, init, B)
    try:
        P.test_rolling_episode_vs_oracle(T, cfg)
        P.test_generate_instances_vs_oracle(T, (D, int(rs.randint(5, 30)), init, int(rs.randint(1, 3))))
        done += 1
    except AssertionError as e:
        print("FAIL", cfg, repr(e)[:300])
    except T.TapError as e:
        print("TapError", cfg, e)
print("rolling/generator configurations ok: %d, %.0f s" % (done, time.time() - t0))
