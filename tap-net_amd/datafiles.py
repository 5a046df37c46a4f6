"""The reference's on-disk dataset format -- the six text files
``pack.create_dataset`` writes (pack.py:580-667) and ``PACKDataset`` / ``RollingDataset`` read back
(pack.py:58-136, rolling.py:472-492).  Host-side plumbing for interoperability: instances generated
on the device (``generate.generate_instances``) can be saved for the reference, and data sets written
by the reference load through ``pack.PACKDataset``.

Per sample (n blocks, D dims, R = D! rotations; generate.py:929-971):
  blocks.txt     R lines of D*n ints, dimension-major: rotation r lists dim p_r[0] of every block, then p_r[1], ...
  pos.txt        1 line of D*n ints, dimension-major
  dep_move.txt   1 line of n*n ints = deps_move.T.flatten()      (entry a*n+b = "b rests above a")
  dep_small.txt  R lines of n*n ints = side matrix of rotation r, flatten()   (entry a*n+b = "a blocks b")
  dep_large.txt  R lines, same for the other side
  container.txt  1 line of n ints in {0,1}: target container of each block ('mul' input types)
"""
import os

import numpy as np


def _rows(a):
    return "".join(" ".join(str(int(v)) for v in row) + "\n" for row in a)


def write_dataset(data_dir, static, dynamic, positions, container_ids=None, seed=None):
    """Write N instances given in PACKDataset layout ('bot', allow_rot=True): static (N,1+D,n*R),
    dynamic (N,3n,n*R), positions (N,n,D).  ``container_ids`` (N,n) defaults to random 0/1 like
    pack.py:652."""
    st = np.asarray(static.cpu() if hasattr(static, "cpu") else static)
    dy = np.asarray(dynamic.cpu() if hasattr(dynamic, "cpu") else dynamic)
    pos = np.asarray(positions.cpu() if hasattr(positions, "cpu") else positions)
    N, rows, nR = st.shape
    D = rows - 1
    n = dy.shape[1] // 3
    R = nR // n
    if container_ids is None:
        container_ids = np.random.RandomState(seed).randint(0, 2, size=(N, n))
    os.makedirs(data_dir, exist_ok=True)
    blocks = st[:, 1:, :].reshape(N, D, R, n).transpose(0, 2, 1, 3).reshape(N * R, D * n)
    small = dy[:, n:2 * n, :].reshape(N, n, R, n).transpose(0, 2, 1, 3).reshape(N * R, n * n)
    large = dy[:, 2 * n:, :].reshape(N, n, R, n).transpose(0, 2, 1, 3).reshape(N * R, n * n)
    move = dy[:, :n, :n].transpose(0, 2, 1).reshape(N, n * n)
    files = {"blocks": blocks, "pos": pos.transpose(0, 2, 1).reshape(N, D * n), "dep_move": move,
             "dep_small": small, "dep_large": large, "container": np.asarray(container_ids).reshape(N, n)}
    for name, arr in files.items():
        with open(os.path.join(data_dir, name + ".txt"), "w") as f:
            f.write(_rows(arr))
    return data_dir


def read_raw(data_dir, blocks_num, block_dim):
    """-> dict of the six arrays, one leading axis per sample (and per rotation where applicable)."""
    out = {}
    R = int(np.prod(np.arange(1, block_dim + 1)))
    for name in ("blocks", "pos", "dep_move", "dep_small", "dep_large", "container"):
        a = np.loadtxt(os.path.join(data_dir, name + ".txt"), ndmin=2).astype(np.int64)
        out[name] = a
    N = out["pos"].shape[0]
    out["blocks"] = out["blocks"].reshape(N, R, block_dim, blocks_num)
    out["pos"] = out["pos"].reshape(N, block_dim, blocks_num)
    return out
