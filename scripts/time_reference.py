"""Time the REFERENCE's own numpy/Python path (only where /root/reference exists, i.e. the build
container) next to the C oracle on the same work, so the oracle's timing on the GPU box can be
translated: ref_on_box ~= oracle_on_box * (ref_here / oracle_here)   (CPU baseline plan).

Work = what bench.py's cpu_baseline times: per env and step update_dynamic + update_mask + add_new_block,
plus calc_ratio at the end; B = 128 (BASELINE configs[0]), median of 3 repeats.  Two figures per config, as
SURVEY 8(d) asks: one process on one core, and -- the reference being single-threaded (trainer.py:155-156,
num_workers = 0) -- one forked process per CPU of this container on disjoint batch slices, summed.

    python scripts/time_reference.py            -> one JSON line per config
"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                                                         # noqa: E402
import oracle_lib as O                                               # noqa: E402
import ref_loader                                                    # noqa: E402
from tap_net_amd import synth                                        # noqa: E402  (host-side generators only)

tools, pack = ref_loader.load()[:2]
O.set_threads(1)                                                     # the oracle's "1 core" figure really is one thread
torch.set_num_threads(1)
CASES = [("c1/c2", 2, [5, 50], 10, "C+P+S-lb-soft", "LB_GREEDY"), ("c3", 3, [5, 5, 50], 10, "C+P+S-lb-soft", "LB_GREEDY"),
         ("c4", 2, [7, 100], 20, "C+P+S-mcs-soft", "MACS")]
B = 128
for name, D, cs, n, reward, strategy in CASES:
    static, dynamic = synth.rand_instances(B, n, D, seed=12345)
    tape = synth.random_feasible_tape(static, dynamic, n, seed=12346)
    R = static.shape[2] // n
    st_np, dyn_np, tape_np = static.numpy(), dynamic.numpy(), tape.numpy()

    def ref_pass():
        envs = [tools.Container(list(cs), n, reward, "diff", packing_strategy=strategy) for _ in range(B)]
        dyn, mask = dynamic, torch.ones(B, n * R)
        for t in range(n):
            ptr = tape[:, t]
            dyn = pack.update_dynamic(dyn, static, ptr, "bot", True)
            _, mask = pack.update_mask(mask, dyn, static, ptr, "bot", True)
            blk = st_np[np.arange(B), 1:, tape_np[:, t]]
            for b in range(B):
                envs[b].add_new_block(blk[b])
        return [e.calc_ratio() for e in envs]

    def orc_pass():
        dyn, mask = dyn_np, np.ones((B, n * R), np.float32)
        blocks = np.stack([st_np[np.arange(B), 1:, tape_np[:, t]] for t in range(n)], axis=1).astype(np.int32)
        for t in range(n):
            dyn = O.update_dynamic(dyn, st_np, tape_np[:, t], n, 3)
            _, mask = O.update_mask(mask, dyn, tape_np[:, t], n, R)
        return O.run_episodes(O.make_desc(cs, n, reward, "diff", strategy), blocks, nthreads=1, want_heightmaps=False)

    def med(fn, reps):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    t_ref = med(ref_pass, 3)
    t_orc = med(lambda: [orc_pass() for _ in range(20)], 3) / 20
    # all CPUs: P forked processes, each running the whole B = 128 pass on its own copy (disjoint slices of a
    # P*B batch), wall time of the slowest; torch pinned to one thread per process
    import multiprocessing as mp
    P = len(os.sched_getaffinity(0))
    def worker(q):
        torch.set_num_threads(1)
        t0 = time.perf_counter(); ref_pass(); q.put(time.perf_counter() - t0)
    ctxm = mp.get_context("fork")
    runs = []
    for _ in range(3):
        q = ctxm.Queue()
        ps = [ctxm.Process(target=worker, args=(q,)) for _ in range(P)]
        t0 = time.perf_counter()
        for p_ in ps: p_.start()
        for p_ in ps: p_.join()
        runs.append(time.perf_counter() - t0)
    t_par = float(np.median(runs))
    model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
    print(json.dumps(dict(config=name, B=B, n=n, reference_env_steps_per_s=B * n / t_ref, oracle_env_steps_per_s=B * n / t_orc,
                          ratio_oracle_over_reference=t_ref / t_orc, cores=1,
                          procs=P, reference_env_steps_per_s_procs=P * B * n / t_par,
                          cpu_model=model[0] if model else None,
                          note="reference = tools.Container + pack.update_dynamic/update_mask (torch CPU), this container; "
                               "procs = one forked process per CPU, each a whole B = 128 pass")), flush=True)
