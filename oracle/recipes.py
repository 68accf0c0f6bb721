"""Bit-reproducible synthetic input recipes shared by oracle/gen_golden.py, tests/ and bench.py.

Built only from numpy ``default_rng(seed).integers`` (PCG64, platform independent),
so a fixture needs to store just the seed and the reference's outputs.

  lattice(seed, shape)  integers in [-3, 3] as fp32: every squared L2 distance is an
                        exact fp32 integer -> any correct fp32 distance kernel yields
                        the same D bit for bit (parity level P1).
  dyadic(seed, shape)   sum of four integers in [-32, 32] divided by 64: bell-shaped,
                        every L1 distance over <= 768 dims is exact in fp32 (level P2).
"""
import numpy as np


def lattice(seed, shape):
    return np.random.default_rng(seed).integers(-3, 4, size=shape).astype(np.float32)


def dyadic(seed, shape):
    r = np.random.default_rng(seed).integers(-32, 33, size=(4,) + tuple(shape))
    return (r.sum(0).astype(np.float32) / np.float32(64.0)).astype(np.float32)
