"""Shared helpers for the -m gpu parity tests (they drive the product only through the C ABI)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def pkg():
    return entry.load_package()


def to_pkg_params(p):
    """oracle.Params -> package TfheParams"""
    m = pkg()
    return m.TfheParams(p.k, p.glwe_poly_degree, p.n, m.DecomposerParams(p.pbs.log_base, p.pbs.levels),
                        m.DecomposerParams(p.ks.log_base, p.ks.levels), log_p=p.log_p,
                        padding_bits=p.padding_bits)


def rand_u32(rng, shape):
    return rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32)
