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

tools, pack, generate = ref_loader.load()
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


# ---- c5: rolling.validate's loop (rolling.py:589-637) without the networks: per instance InitialContainer, then per
# decoding step convert_to_input + the initial-mask rule + Container.add_new_block + remove_block, and a whole episode
# (update_dynamic / update_mask / add_new_block) on the last window -- the work bench.py's cpu_baseline_rolling gives
# the oracle.  Instances: the four 50-block 3D instances the reference's generate_blocks wrote into
# tests/golden/rolling.npz, eight times over (B = 32).
import ast                                                           # noqa: E402
z = np.load(os.path.join(ROOT, "tests", "golden", "rolling.npz"))
cases = [ast.literal_eval(str(c)) for c in z["cases"]]
picks = [i for i, c in enumerate(cases) if c["D"] == 3 and c["N"] == 50 and c["child"] == 10]
N, child, D, R = 50, 10, 3, 6
cs = [5, 5, 250]
inst = []
for i in picks:
    blocks, pos, ptr = z["r%d_blocks" % i].astype(np.int64), z["r%d_pos" % i].astype(np.int64), z["r%d_ptr" % i].astype(np.int64)
    perms = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]
    blocks_all = np.concatenate([blocks[:, list(p_)] for p_ in perms], axis=0)            # rolling.py:484-486 layout
    # a feasible tape for the last window, from the fixture's last static / dynamic
    st_last, dy_last = z["r%d_static" % i][-1].astype(np.float32), z["r%d_dynamic" % i][-1].astype(np.float32)
    tail = synth.random_feasible_tape(torch.from_numpy(st_last[None]), torch.from_numpy(dy_last[None]), child, seed=5)[0].numpy()
    inst.append((blocks_all, pos, ptr, tail, cases[i]["init"]))
inst = inst * 8
B5 = len(inst)


def ref_rolling():
    for blocks_all, pos, ptr, tail, init in inst:
        ic = generate.InitialContainer(blocks_all, pos, N, list(init), True, child, 'bot')
        env = tools.Container(list(cs), N, "C+P+S-lb-soft", "diff", packing_strategy="LB_GREEDY")
        for t in range(N - child):
            static, dynamic = ic.convert_to_input()
            move = dynamic[:child].sum(0); small = dynamic[child:2 * child].sum(0); large = dynamic[2 * child:].sum(0)
            _ = (small * large + move) == 0                                                # model.py:297-307
            p = int(ptr[t])
            env.add_new_block(static[1:, p].astype(np.float32))
            ic.remove_block(ic.sub_graph_nodes[p % child])
        static, dynamic = ic.convert_to_input()
        st_t = torch.from_numpy(static[None].astype(np.float32)); dyn = torch.from_numpy(dynamic[None].astype(np.float32))
        mask = torch.ones(1, child * R)
        for t in range(child):
            pt = torch.tensor([int(tail[t])])
            dyn = pack.update_dynamic(dyn, st_t, pt, "bot", True)
            _, mask = pack.update_mask(mask, dyn, st_t, pt, "bot", True)
            env.add_new_block(static[1:, int(tail[t])].astype(np.float32))
        env.calc_ratio()


def orc_rolling():
    for blocks_all, pos, ptr, tail, init in inst:
        ro = O.Rolling(blocks_all[:N].astype(np.int32), pos.astype(np.int32), list(init), child)
        e = O.Env(cs, N, "C+P+S-lb-soft", "diff", "LB_GREEDY")
        for t in range(N - child):
            rc, st, dy, _ = ro.convert_to_input()
            O.initial_mask(dy[None], child)
            p = int(ptr[t])
            e.add_new_block(st[1:, p])
            ro.remove(p % child)
        rc, st, dy, _ = ro.convert_to_input()
        mask, dyn = np.ones((1, child * R), np.float32), dy[None]
        for t in range(child):
            p = np.array([tail[t]], dtype=np.int64)
            e.add_new_block(st[1:, int(p[0])])
            dyn = O.update_dynamic(dyn, st[None], p, child, 3)
            _, mask = O.update_mask(mask, dyn, p, child, R)
        e.calc_ratio()


t_ref = med(ref_rolling, 3)
t_orc = med(lambda: [orc_rolling() for _ in range(3)], 3) / 3
P = len(os.sched_getaffinity(0))


def worker5(q):
    torch.set_num_threads(1)
    t0 = time.perf_counter(); ref_rolling(); q.put(time.perf_counter() - t0)


runs = []
for _ in range(2):
    q = ctxm.Queue()
    ps = [ctxm.Process(target=worker5, args=(q,)) for _ in range(P)]
    t0 = time.perf_counter()
    for p_ in ps: p_.start()
    for p_ in ps: p_.join()
    runs.append(time.perf_counter() - t0)
t_par = float(np.median(runs))
print(json.dumps(dict(config="c5", B=B5, n=N, reference_env_steps_per_s=B5 * N / t_ref, oracle_env_steps_per_s=B5 * N / t_orc,
                      ratio_oracle_over_reference=t_ref / t_orc, cores=1, procs=P,
                      reference_env_steps_per_s_procs=P * B5 * N / t_par, cpu_model=model[0] if model else None,
                      note="rolling.validate's loop without the networks: generate.InitialContainer + convert_to_input + "
                           "remove_block per decoding step, tools.Container.add_new_block, a whole episode on the last window; "
                           "the reference generator's own four 50-block 3D instances (tests/golden/rolling.npz) x 8")), flush=True)
