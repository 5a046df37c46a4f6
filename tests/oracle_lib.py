"""ctypes front-end for the CPU oracle (oracle/libtap_oracle.so).

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.environ.get("TAP_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libtap_oracle.so")   # TAP_ORACLE_LIB: the sanitizer build (oracle/Makefile: check-asan)

LB_GREEDY, MACS, LB = 0, 1, 2
F_HARD, F_USE_P, F_USE_S, F_MCS_ZERO, F_MCS_TIE = 1, 2, 4, 8, 16
R_C, R_CxS, R_CP, R_CPxS, R_CPS, R_2CPS, R_CxPxS, R_CP_HALF = range(8)
FEAT = {"full": 0, "zero": 1, "diff": 2}

_RATIO_TABLE = {  # tools.py:3919-3957
    "comp": R_C, "soft": R_CxS, "hard": R_CxS, "pyrm": R_CP,
    "pyrm-soft": R_CPxS, "pyrm-hard": R_CPxS, "mcs-soft": R_CPxS, "mcs-hard": R_CPxS,
    "pyrm-soft-sum": R_CPS, "pyrm-soft-SUM": R_2CPS, "pyrm-hard-sum": R_CPS,
    "pyrm-hard-SUM": R_2CPS, "CPS": R_CxPxS,
}


class Desc(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("D", "W", "L", "H", "n_max", "strategy", "flags", "ratio_mode", "feature")]


def build(force=False):
    src = [os.path.join(ORACLE_DIR, f) for f in ("tap_oracle.c", "tap_oracle.h", "Makefile")]
    if (force or not os.path.exists(_LIB_PATH)
            or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_env_new.restype = C.c_void_p
        L.orc_env_new.argtypes = [C.POINTER(Desc)]
        L.orc_env_free.argtypes = [C.c_void_p]
        L.orc_env_clear.argtypes = [C.c_void_p]
        L.orc_env_add_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_env_feature.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_env_cps.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_env_ratio.restype = C.c_double
        L.orc_env_ratio.argtypes = [C.c_void_p]
        for name, rt in (("heightmap", C.POINTER(C.c_int32)), ("positions", C.POINTER(C.c_int32)),
                         ("stable", C.POINTER(C.c_uint8)), ("container", C.POINTER(C.c_int32)),
                         ("valid", C.c_int64), ("empty", C.c_int64), ("count", C.c_int32),
                         ("error", C.c_int32)):
            f = getattr(L, "orc_env_" + name)
            f.restype = rt
            f.argtypes = [C.c_void_p]
        L.orc_feature_len.argtypes = [C.POINTER(Desc)]
        L.orc_is_stable_2d.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_is_stable_3d_mask.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.orc_run_episodes.argtypes = [C.POINTER(Desc), C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.c_int]
        L.orc_calc_positions_lb_greedy.argtypes = [C.POINTER(Desc), C.c_int] + [C.c_void_p] * 5
        L.orc_calc_positions_mcs.argtypes = [C.POINTER(Desc), C.c_int] + [C.c_void_p] * 5
        L.orc_render_scores.argtypes = [C.POINTER(Desc), C.POINTER(Desc), C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_reward.argtypes = [C.POINTER(Desc), C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_instance_from_blocks.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4
        L.orc_rolling_new.restype = C.c_void_p
        L.orc_rolling_new.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_rolling_free.argtypes = [C.c_void_p]
        L.orc_rolling_window.argtypes = [C.c_void_p] * 4
        L.orc_rolling_remove.argtypes = [C.c_void_p, C.c_int]
        L.orc_rng_words.restype = C.c_void_p
        L.orc_rng_words.argtypes = [C.c_void_p, C.c_int64]
        L.orc_rng_counter.restype = C.c_void_p
        L.orc_rng_counter.argtypes = [C.c_uint64]
        L.orc_rng_key.restype = C.c_uint64
        L.orc_rng_key.argtypes = [C.c_uint64] * 4
        L.orc_rng_free.argtypes = [C.c_void_p]
        L.orc_rng_consumed.restype = C.c_int64
        L.orc_rng_consumed.argtypes = [C.c_void_p]
        L.orc_rng_exhausted.argtypes = [C.c_void_p]
        L.orc_bpp3d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_ppsg_try_layout.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_generate_blocks_with_gt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                  C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ppsg_try_layout_d.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_bpp2d_easy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p]
        L.orc_generate_blocks_with_gt_2d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                     C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                                     C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ppsg_gt2d.restype = C.c_int64
        L.orc_ppsg_gt2d.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                    C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_ppsg_order_2d.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
        L.orc_ppsg_gt.restype = C.c_int64
        L.orc_ppsg_gt.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                  C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_ppsg_order.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_initial_mask.argtypes = [C.c_int] * 4 + [C.c_void_p] * 2
        L.orc_update_dynamic.argtypes = [C.c_int] * 6 + [C.c_void_p] * 4
        L.orc_update_mask.argtypes = [C.c_int] * 4 + [C.c_void_p] * 5
        _lib = L
    return _lib


def make_desc(container_size, blocks_num, reward_type, heightmap_type="diff",
              packing_strategy="LB_GREEDY"):
    """Same arguments as tools.Container.__init__ (tools.py:3611)."""
    D = len(container_size)
    W = int(container_size[0])
    L = int(container_size[1]) if D == 3 else 1
    H = int(container_size[-1])
    # tools.py:3617-3620: the reward string overrides the strategy
    if reward_type in ("C+P+S-mul-soft", "C+P+S-mul-hard"):
        packing_strategy = "MUL"
    elif reward_type in ("C+P+S-mcs-soft", "C+P+S-mcs-hard"):
        packing_strategy = "MACS"
    strategy = MACS if packing_strategy in ("MACS", "MUL") else LB if packing_strategy == "LB" else LB_GREEDY
    flags = 0
    if reward_type.endswith("hard"):
        flags |= F_HARD
    if "P" in reward_type:
        flags |= F_USE_P
    if "S" in reward_type:
        flags |= F_USE_S
    if reward_type.startswith("mcs"):
        flags |= F_MCS_ZERO
    if "mcs" in reward_type:
        flags |= F_MCS_TIE
    if reward_type == "C+P-lb-soft":
        rm = R_CP_HALF
    else:
        rm = _RATIO_TABLE.get(reward_type, R_CPS)
    return Desc(D, W, L, H, int(blocks_num), strategy, flags, rm, FEAT[heightmap_type])


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Env:
    """One reference-shaped Container backed by the oracle."""

    def __init__(self, container_size, blocks_num, reward_type, heightmap_type="diff",
                 packing_strategy="LB_GREEDY"):
        self.desc = make_desc(container_size, blocks_num, reward_type, heightmap_type, packing_strategy)
        self._h = lib().orc_env_new(C.byref(self.desc))
        if not self._h:
            raise ValueError("oracle rejected the container description")
        self.flen = lib().orc_feature_len(C.byref(self.desc))
        self.D = self.desc.D

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_env_free(self._h)
            self._h = None

    def add_new_block(self, block):
        blk = np.ascontiguousarray(np.asarray(block).astype(np.int32))
        feat = np.zeros(max(self.flen, 1), np.int32)
        rc = lib().orc_env_add_block(self._h, _p(blk), _p(feat))
        return rc, self._shape_feature(feat[:self.flen])

    def _shape_feature(self, f):
        d = self.desc
        if d.D == 3:
            return f.reshape((2, d.W, d.L)) if d.feature == 2 else f.reshape((d.W, d.L))
        return f

    def get_heightmap(self):
        """Container.get_heightmap (tools.py:3824-3856): the feature of the current state"""
        feat = np.zeros(max(self.flen, 1), np.int32)
        lib().orc_env_feature(self._h, _p(feat))
        return self._shape_feature(feat[:self.flen])

    def clear(self):
        lib().orc_env_clear(self._h)

    def calc_ratio(self):
        return lib().orc_env_ratio(self._h)

    @property
    def heightmap(self):
        d = self.desc
        a = np.ctypeslib.as_array(lib().orc_env_heightmap(self._h), (d.W * d.L,)).copy()
        return a.reshape((d.W, d.L)) if d.D == 3 else a

    @property
    def positions(self):
        d = self.desc
        return np.ctypeslib.as_array(lib().orc_env_positions(self._h), (d.n_max, d.D)).copy()

    @property
    def stable(self):
        return np.ctypeslib.as_array(lib().orc_env_stable(self._h), (self.desc.n_max,)).astype(bool)

    @property
    def container(self):
        d = self.desc
        shape = (d.W, d.L, d.H) if d.D == 3 else (d.W, d.H)
        return np.ctypeslib.as_array(lib().orc_env_container(self._h), shape).copy()

    @property
    def valid_size(self):
        return lib().orc_env_valid(self._h)

    @property
    def empty_size(self):
        return lib().orc_env_empty(self._h)

    @property
    def error(self):
        return lib().orc_env_error(self._h)


def run_episodes(desc, blocks, nthreads=1, want_features=True, want_heightmaps=True):
    """blocks: (B, n, D) ints -> dict of per-step / final outputs for B fresh episodes."""
    blocks = np.ascontiguousarray(blocks, dtype=np.int32)
    B, n, D = blocks.shape
    assert D == desc.D
    fl = lib().orc_feature_len(C.byref(desc))
    cells = desc.W * desc.L
    out = dict(
        positions=np.zeros((B, n, D), np.int32), stable=np.zeros((B, n), np.uint8),
        features=np.zeros((B, n, fl), np.int32) if want_features else None,
        heightmaps=np.zeros((B, n, cells), np.int32) if want_heightmaps else None,
        ratio=np.zeros(B, np.float64), cps=np.zeros((B, 3), np.float64),
        counters=np.zeros((B, 3), np.int64), errs=np.zeros(B, np.int32))
    nerr = lib().orc_run_episodes(C.byref(desc), B, n, _p(blocks), _p(out["positions"]),
                                  _p(out["stable"]), _p(out["features"]), _p(out["heightmaps"]),
                                  _p(out["ratio"]), _p(out["cps"]), _p(out["counters"]),
                                  _p(out["errs"]), nthreads)
    out["nerr"] = nerr
    return out


def calc_positions_lb_greedy(blocks, container_size, reward_type):
    blocks = np.ascontiguousarray(blocks, dtype=np.int32)
    n, D = blocks.shape
    desc = make_desc(container_size, n, reward_type, "full", "LB_GREEDY")
    pos = np.zeros((n, D), np.int32)
    st = np.zeros(n, np.uint8)
    ratio = C.c_double()
    scores = np.zeros(5, np.int64)
    rc = lib().orc_calc_positions_lb_greedy(C.byref(desc), n, _p(blocks), _p(pos), _p(st),
                                            C.byref(ratio), _p(scores))
    return rc, pos, st.astype(bool), ratio.value, scores


def calc_positions_mcs(blocks, container_size, reward_type):
    """tools.calc_positions_mcs (tools.py:3213-3315)."""
    blocks = np.ascontiguousarray(blocks, dtype=np.int32)
    n, D = blocks.shape
    desc = make_desc(container_size, n, reward_type, "full", "MACS")
    pos = np.zeros((n, D), np.int32)
    st = np.zeros(n, np.uint8)
    ratio = C.c_double()
    scores = np.zeros(5, np.int64)
    rc = lib().orc_calc_positions_mcs(C.byref(desc), n, _p(blocks), _p(pos), _p(st), C.byref(ratio), _p(scores))
    return rc, pos, st.astype(bool), ratio.value, scores


def render_scores(static, tour, reward_type, input_type, allow_rot, container_width, container_height,
                  packing_strategy, initial_container_height=None):
    """The per-sample figures pack.render writes (pack.py:670-807): -> (ratio (B,), scores (B,5) fp64, errs)."""
    static = np.ascontiguousarray(static, dtype=np.float32)
    tour = np.ascontiguousarray(tour, dtype=np.int64)
    B, rows, nR = static.shape
    mul = input_type in ("mul", "mul-with")
    D = rows - (2 if mul else 1)
    R = [1, 1, 2, 6][D] if allow_rot else 1
    n = nR // R
    tour = np.ascontiguousarray(tour[:, :n])
    cs = [container_width, container_height] if D == 2 else [container_width, container_width, container_height]
    cs_mul = cs if D == 2 else [container_width, container_width, initial_container_height]   # pack.py:718-724
    strat = "MACS" if packing_strategy in ("MACS", "MUL") else "LB_GREEDY"
    desc = make_desc(cs, n, reward_type, "full", strat)
    desc_mul = make_desc(cs_mul, n, reward_type, "full", strat) if mul else desc
    ratio = np.zeros(B, np.float64)
    scores = np.zeros((B, 5), np.float64)
    errs = np.zeros(B, np.int32)
    nerr = lib().orc_render_scores(C.byref(desc), C.byref(desc_mul), B, n, nR, rows, _p(static), _p(tour), int(mul),
                                   _p(ratio), _p(scores), _p(errs))
    assert nerr >= 0
    return ratio, scores, errs


def reward_mul(static, tour, reward_type, container_width, container_height, R):
    """pack.reward for input types 'mul' / 'mul-with' (pack.py:451-466) on top of the oracle's
    calc_positions_lb_greedy: static (B, 1+D+1, n*R) with the target id in the last row."""
    static = np.asarray(static, dtype=np.float32)
    tour = np.asarray(tour, dtype=np.int64)
    B, rows, nR = static.shape
    n, D = nR // R, rows - 2
    cs = [container_width, container_height] if D == 2 else [container_width, container_width, container_height]
    out = np.zeros(B, np.float32)
    for b in range(B):
        sample = static[b][:, tour[b][:n]]                       # gather by tour, first n steps
        blocks = sample[1:1 + D].T.astype(np.int32)
        ids = sample[-1]
        sc = []
        for target in (0, 1):
            mine = blocks[ids == target]
            if len(mine) == 0:
                sc.append(0.0)
                continue
            rc, _, _, ratio, _ = calc_positions_lb_greedy(mine, cs, reward_type)
            assert rc == 0
            sc.append(ratio)
        out[b] = np.float32((sc[0] + sc[1]) / 2)
    return -out


def reward(static, tour, reward_type, container_width, container_height, nthreads=1):
    static = np.ascontiguousarray(static, dtype=np.float32)
    tour = np.ascontiguousarray(tour, dtype=np.int64)
    B, rows, nR = static.shape
    n = tour.shape[1]
    D = rows - 1
    cs = [container_width, container_height] if D == 2 else [container_width, container_width, container_height]
    desc = make_desc(cs, n, reward_type, "full", "LB_GREEDY")
    out = np.zeros(B, np.float32)
    nerr = lib().orc_reward(C.byref(desc), B, n, nR, rows, _p(static), _p(tour), _p(out), nthreads)
    return nerr, out


def set_threads(n):
    """OpenMP threads of initial_mask / update_dynamic / update_mask (timing only)."""
    lib().orc_set_threads(int(n))


def initial_mask(dynamic, n):
    dynamic = np.ascontiguousarray(dynamic, dtype=np.float32)
    B, rows, nR = dynamic.shape
    out = np.zeros((B, nR), np.float32)
    lib().orc_initial_mask(B, n, nR, rows, _p(dynamic), _p(out))
    return out


def update_dynamic(dynamic, static, ptr, n, update_time=3):
    dynamic = np.ascontiguousarray(dynamic, dtype=np.float32)
    static = np.ascontiguousarray(static, dtype=np.float32)
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    B, rows, nR = dynamic.shape
    out = np.empty_like(dynamic)
    lib().orc_update_dynamic(B, n, nR, rows, update_time, static.shape[1], _p(dynamic), _p(static),
                             _p(ptr), _p(out))
    return out


def update_mask(mask, dynamic, ptr, n, R):
    mask = np.ascontiguousarray(mask, dtype=np.float32)
    dynamic = np.ascontiguousarray(dynamic, dtype=np.float32)
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    B, rows, nR = dynamic.shape
    cur = np.empty_like(mask)
    new = np.empty_like(mask)
    lib().orc_update_mask(B, n, R, rows, _p(mask), _p(dynamic), _p(ptr), _p(cur), _p(new))
    return cur, new


def is_stable_3d_mask(bx, by, mask):
    m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(-1)
    return bool(lib().orc_is_stable_3d_mask(bx, by, _p(m)))


def instance_from_blocks(blocks, init_size, arm_size=1):
    """blocks (n, D) -> (accepted, positions (n,D), static (1+D, nR), dynamic (3n, nR))."""
    blocks = np.ascontiguousarray(blocks, dtype=np.int32)
    n, D = blocks.shape
    R = 2 if D == 2 else 6
    cs = np.ascontiguousarray(init_size, dtype=np.int32)
    pos = np.zeros((n, D), np.int32)
    st = np.zeros((1 + D, n * R), np.float32)
    dyn = np.zeros((3 * n, n * R), np.float32)
    rc = lib().orc_instance_from_blocks(D, _p(cs), n, arm_size, _p(blocks), _p(pos), _p(st), _p(dyn))
    return rc, pos, st, dyn


class Rolling:
    """generate.InitialContainer backed by the oracle."""

    def __init__(self, blocks, positions, init_size, child):
        blocks = np.ascontiguousarray(blocks, dtype=np.int32)
        positions = np.ascontiguousarray(positions, dtype=np.int32)
        self.N, self.D = blocks.shape
        self.child, self.R = child, (2 if self.D == 2 else 6)
        cs = np.ascontiguousarray(init_size, dtype=np.int32)
        self._h = lib().orc_rolling_new(self.D, _p(cs), self.N, child, _p(blocks), _p(positions))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_rolling_free(self._h)
            self._h = None

    def convert_to_input(self):
        st = np.zeros((1 + self.D, self.child * self.R), np.float32)
        dy = np.zeros((3 * self.child, self.child * self.R), np.float32)
        nodes = np.zeros(self.child, np.int32)
        rc = lib().orc_rolling_window(self._h, _p(st), _p(dy), _p(nodes))
        return rc, st, dy, nodes

    def remove(self, local_index):
        lib().orc_rolling_remove(self._h, int(local_index))


# ---- PPSG (generate.py:17-301) -------------------------------------------------------------------------

def numpy_mt_words(seed, count):
    """The 32-bit words numpy's legacy RandomState(seed) hands out, in order (what np.random.* consumes after
    np.random.seed(seed)): numpy's MT19937, not reference code."""
    return np.random.RandomState(int(seed))._bit_generator.random_raw(int(count)).astype(np.uint32)


class Rng:
    """Word source for the PPSG restatement: an explicit stream (``words``) or the counter generator (``key``)."""

    def __init__(self, words=None, key=None):
        self._words = None if words is None else np.ascontiguousarray(words, dtype=np.uint32)
        self._h = (lib().orc_rng_words(_p(self._words), len(self._words)) if words is not None
                   else lib().orc_rng_counter(C.c_uint64(int(key))))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_rng_free(self._h)
            self._h = None

    @property
    def consumed(self):
        return lib().orc_rng_consumed(self._h)

    @property
    def exhausted(self):
        return bool(lib().orc_rng_exhausted(self._h))


def rng_key(seed, a, b, c):
    return int(lib().orc_rng_key(int(seed), int(a), int(b), int(c)))


def bpp3d(rng, n, gt_size, size_range=(1, 5)):
    gt = np.ascontiguousarray(gt_size, dtype=np.int32)
    blocks, pos = np.zeros((n, 3), np.int32), np.zeros((n, 3), np.int32)
    rc = lib().orc_bpp3d(rng._h, n, _p(gt), int(size_range[0]), int(size_range[1]), _p(blocks), _p(pos))
    return rc, blocks, pos


def ppsg_try_layout(blocks, init_size, arm_size=1, input_simple=False):
    blocks = np.ascontiguousarray(blocks, dtype=np.int32)
    n, D = blocks.shape
    cs = np.ascontiguousarray(init_size, dtype=np.int32)
    pos = np.zeros((n, D), np.int32)
    rc = lib().orc_ppsg_try_layout_d(D, n, _p(cs), arm_size, _p(blocks), int(input_simple), _p(pos))
    return rc, pos


def gauss_split_cdf(max_len, size_range=(1, 5)):
    """The table BPP_Generator_2D_easy's Gaussian split draws from (generate.py:463-469), made with numpy itself:
    row L (2*max_size - 1 <= L <= max_len) = the normalised cumulative sum np.random.choice builds of
    prob / np.sum(prob).  -> float64 (max_len + 1, max_len)."""
    mn, mx = int(size_range[0]), int(size_range[1])
    tab = np.zeros((max_len + 1, max(1, max_len)), np.float64)
    mu, sigma = 0.5, 0.16
    for L in range(2 * mx - 1, max_len + 1):
        m = L - 2 * mn
        if m < 1:
            continue
        prob_x = np.linspace(mu - 3 * sigma, mu + 3 * sigma, m)
        prob = np.exp(-(prob_x - mu) ** 2 / (2 * sigma ** 2)) / (np.sqrt(2 * np.pi) * sigma)
        prob = prob / np.sum(prob)
        cdf = prob.cumsum()
        cdf /= cdf[-1]
        tab[L, :m] = cdf
    return tab


def bpp2d_easy(rng, n, gt_size, size_range=(1, 5), gauss=None):
    gt = np.ascontiguousarray(gt_size, dtype=np.int32)
    g = gauss_split_cdf(int(max(gt)), size_range) if gauss is None else gauss
    blocks, pos = np.zeros((n, 2), np.int32), np.zeros((n, 2), np.int32)
    rc = lib().orc_bpp2d_easy(rng._h, n, _p(gt), int(size_range[0]), int(size_range[1]), _p(g), g.shape[1], g.shape[0],
                              _p(blocks), _p(pos))
    return rc, blocks, pos


def generate_blocks_with_gt_2d(rng, n, gt_size, init_size, arm_size=1, size_range=(1, 5), input_simple=False,
                               allow_rot=True, max_bpp=10 ** 7):
    gt = np.ascontiguousarray(gt_size, dtype=np.int32)
    cs = np.ascontiguousarray(init_size, dtype=np.int32)
    g = gauss_split_cdf(int(max(gt)), size_range)
    blocks, pos = np.zeros((n, 2), np.int32), np.zeros((n, 2), np.int32)
    stats = np.zeros(2, np.int64)
    rc = lib().orc_generate_blocks_with_gt_2d(rng._h, n, _p(gt), _p(cs), arm_size, int(size_range[0]), int(size_range[1]),
                                              int(input_simple), int(allow_rot), int(max_bpp), _p(g), g.shape[1], g.shape[0],
                                              _p(blocks), _p(pos), _p(stats))
    return rc, blocks, pos, stats


def ppsg_gt2d(seed, instance, gen, n, W, H, size_range=(1, 5), max_attempts=10 ** 7, gauss=None):
    g = gauss_split_cdf(int(max(W, H)), size_range) if gauss is None else gauss
    blocks, pos = np.zeros((n, 2), np.int32), np.zeros((n, 2), np.int32)
    used = lib().orc_ppsg_gt2d(int(seed), int(instance), int(gen), n, W, H, int(size_range[0]), int(size_range[1]),
                               int(max_attempts), _p(g), g.shape[1], g.shape[0], _p(blocks), _p(pos))
    return used, blocks, pos


def generate_blocks_with_gt(rng, n, gt_size, init_size, arm_size=1, size_range=(1, 5), input_simple=False,
                            allow_rot=True, max_bpp=10 ** 7):
    gt = np.ascontiguousarray(gt_size, dtype=np.int32)
    cs = np.ascontiguousarray(init_size, dtype=np.int32)
    blocks, pos = np.zeros((n, 3), np.int32), np.zeros((n, 3), np.int32)
    stats = np.zeros(2, np.int64)
    rc = lib().orc_generate_blocks_with_gt(rng._h, n, _p(gt), _p(cs), arm_size, int(size_range[0]), int(size_range[1]),
                                           int(input_simple), int(allow_rot), int(max_bpp), _p(blocks), _p(pos), _p(stats))
    return rc, blocks, pos, stats


def ppsg_gt(seed, instance, gen, S, ns, W, heights, size_range=(1, 5), max_attempts=10 ** 7):
    h = np.ascontiguousarray(heights, dtype=np.int32)
    blocks, pos = np.zeros((S * ns, 3), np.int32), np.zeros((S * ns, 3), np.int32)
    used = lib().orc_ppsg_gt(int(seed), int(instance), int(gen), S, ns, W, _p(h), int(size_range[0]), int(size_range[1]),
                             int(max_attempts), _p(blocks), _p(pos))
    return used, blocks, pos


def ppsg_order(seed, instance, gen, trial, gt_size, gt_blocks, gt_positions):
    gb = np.ascontiguousarray(gt_blocks, dtype=np.int32)
    gp = np.ascontiguousarray(gt_positions, dtype=np.int32)
    gs = np.ascontiguousarray(gt_size, dtype=np.int32)
    out = np.zeros_like(gb)
    fn = lib().orc_ppsg_order if gb.shape[1] == 3 else lib().orc_ppsg_order_2d
    rc = fn(int(seed), int(instance), int(gen), int(trial), gb.shape[0], _p(gs), _p(gb), _p(gp), _p(out))
    return rc, out
