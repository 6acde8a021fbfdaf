"""helpers to read tests/golden/*.npz"""
import os

import numpy as np

import golden_cases as gc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def dense(g, nm, s, Nc, comps, default):
    shape = (Nc, comps) if comps > 1 else (Nc,)
    a = np.full(shape, default, dtype=np.float64)
    idx = g[f"{nm}_idx_s{s}"]
    if idx.size:
        a[idx] = g[f"{nm}_val_s{s}"]
    return a


def batch_offsets(c: gc.Case, n):
    if c.n_yade == 1:
        return np.array([0, n], dtype=np.int32)
    W = c.n_yade - 1
    return np.array([gc.split_range(n, W, w)[0] for w in range(W)] + [n], dtype=np.int32)


def check_inputs_reproducible(c: gc.Case, g):
    f = gc.fluid_fields(c)
    got = [gc.sha(f[n]) for n in ("U", "gradP", "divT", "ddtU", "vGrad")]
    assert list(g["field_sha"]) == got, "polynomial fluid fields are not bit-reproducible on this host"
    assert str(g["centres_sha"][0]) == gc.sha(gc.cell_centres(c))
    return f


# Tolerances (floating point; index work is compared bit-exactly).
# Oracle vs reference on the same host/libm: identical operation order => a few ulp at most.
RTOL_ORACLE = 1e-13
# GPU vs oracle: device exp/pow differ from glibc in the last ulp(s) and atomics reorder the per-cell sums.
RTOL_GPU = 1e-10
