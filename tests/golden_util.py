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
