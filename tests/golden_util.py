"""Helpers to read the golden fixtures written by tests/golden/make_golden.py."""
import ast
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def cases(name):
    """Yield (meta dict, {blocks, features, heightmaps, positions, stable, valid, empty, ratio, cps})."""
    z = load(name)
    for i, m in enumerate(z["cases"]):
        meta = ast.literal_eval(str(m))
        pre = "c%d_" % i
        yield meta, {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


def unpack_dynamic(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(np.float32)


RENDER_FILES = ("ratio", "valid_size", "box_size", "empty_size", "stable_num", "packing_height", "time", "ids")


def render_cases():
    """Yield (meta dict, static, tour, {suffix: bytes of the file the reference wrote}) from render.npz."""
    z = load("render.npz")
    data = {}
    for D in (2, 3):
        st = load("dataset_%dd.npz" % D)["static"].astype(np.float32)
        tour = load("episode_%dd.npz" % D)["tour_idx"].astype(np.int64)
        n = tour.shape[1]
        ids = np.repeat(z["target_ids_%dd" % D].astype(np.float32)[:, None, :], 1, axis=1)
        ids = np.tile(ids, (1, 1, st.shape[2] // n))
        data[D] = (st, np.concatenate((st, ids), 1), tour)
    for m in z["meta"]:
        k, D, input_type, strategy, reward, W, H, H0, cnt, vt = str(m).split("|")
        k, D, W, H, H0, cnt = int(k), int(D), int(W), int(H), int(H0), int(cnt)
        st, st_mul, tour = data[D]
        meta = dict(k=k, D=D, input_type=input_type, strategy=strategy, reward=reward, W=W, H=H, H0=H0, count=cnt,
                    valid_time=float(vt))
        files = {suf: z["c%d_%dd_%s" % (k, D, suf)].tobytes() for suf in RENDER_FILES}
        yield meta, (st_mul if input_type.startswith("mul") else st)[:cnt], tour[:cnt], files
