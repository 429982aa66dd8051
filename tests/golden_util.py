"""Shared helpers for tests that consume tests/golden/*.npz (made by oracle/make_golden.py)."""
import os

import numpy as np

from oracle import vlp_oracle as O
from oracle.make_golden import CASES, fingerprint, sample  # noqa: F401  (pure helpers; no reference access)
from vlp_amd import synthetic as S

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    """Returns (golden npz dict, oracle params (fp32), batch, model kwargs) or raises Skip-able
    RuntimeError when the RNG stream of this torch build does not reproduce the fixture inputs."""
    mk, bk = CASES[name]
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    p = O.init_params(vocab_size=mk["vocab_size"], layers=mk["layers"], tasks=mk["tasks"], seed=mk["seed"])
    batch = S.make_batch(**bk)
    fp = fingerprint(p, batch)
    if not np.allclose(fp, g["fingerprint"], rtol=1e-9, atol=0):
        raise RuntimeError("RNG stream differs from the one that generated the fixture")
    return g, p, batch, mk


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
