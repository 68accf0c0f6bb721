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


# name: (seed, P, n_distinct, N, K, layout)   -- duplicate-token problems, selection from a stored D
DUPLICATE_CASES = {
    "dup_tile2": (71, 3, 24, 48, 6, "tile"),          # every token twice, copies 24 apart
    "dup_frames": (72, 2, 49, 196, 20, "tile"),       # 4 identical frames of 49 tokens (padded-clip shape)
    "dup_repeat3": (73, 2, 20, 60, 7, "repeat"),      # copies adjacent
    "dup_random": (74, 3, 40, 100, 12, "random"),     # some tokens unique, some 2-5 copies
    "dup_n588": (75, 1, 147, 588, 30, "tile"),        # big clusters; ATen's second accumulator level is live
    "dup_n7": (76, 4, 4, 7, 2, "tile"),               # N < 8: ATen's scalar summation path
    "dup_n5": (77, 4, 3, 5, 2, "random"),
    # every row of D is a permutation of one row: with K = 1 all N row sums are equal in real arithmetic and the
    # medoid is decided purely by how each fp32 sum rounds -- the sharpest probe of the summation order
    "perm_n50": (81, 4, 0, 50, 1, "permrows"),
    "perm_n196": (82, 4, 0, 196, 1, "permrows"),
    "perm_n197": (83, 2, 0, 197, 1, "permrows"),
    "perm_n588": (84, 2, 0, 588, 1, "permrows"),
    "perm_n640": (85, 2, 0, 640, 1, "permrows"),
    "perm_n6": (86, 6, 0, 6, 1, "permrows"),
    "perm_n196_k2": (87, 3, 0, 196, 2, "permrows"),
}


def duplicate_token_problem(seed, P, n_distinct, N, layout):
    """Stored-distance problems whose tokens collide (level P0 with exact ties).

    Returns (D [P,N,N] fp32, X [P,N,4] fp32).  Distances between the n_distinct base tokens are
    integers in [1, 2^24) scaled by 2^-20 (full 24-bit mantissas, so fp32 row sums round and their
    value depends on the order of summation); token n is a copy of base token idx[n].  D is then what
    cluster_utils.py:35-41 makes of it in fp32: (d - max) - 1, diagonal - 1 more.  X (copies share a
    row) only feeds the first-medoid choice and the reference's centre-shift stop test.
    """
    rng = np.random.default_rng(seed)
    f = np.float32
    Ds, Xs = [], []
    for _ in range(P):
        if layout == "permrows":
            row = rng.integers(1, 1 << 24, size=N).astype(f) * f(2.0 ** -20)
            D = np.stack([-(row[rng.permutation(N)]) - f(1.0) for _ in range(N)]).astype(f)
            x = rng.integers(-3, 4, size=(N, 4)).astype(f)
            x[:, 0] += np.arange(N, dtype=f) * f(8.0)
            Ds.append(D)
            Xs.append(x)
            continue
        base = rng.integers(1, 1 << 24, size=(n_distinct, n_distinct)).astype(f) * f(2.0 ** -20)
        base = np.minimum(base, base.T)
        base[np.arange(n_distinct), np.arange(n_distinct)] = 0.0
        if layout == "tile":
            idx = np.arange(N) % n_distinct
        elif layout == "repeat":
            idx = np.arange(N) // (N // n_distinct)
        else:
            idx = rng.integers(0, n_distinct, size=N)
        d = base[idx][:, idx].astype(f)
        D = ((d - d.max()).astype(f) - f(1.0)).astype(f)
        D[np.arange(N), np.arange(N)] -= f(1.0)
        xb = rng.integers(-3, 4, size=(n_distinct, 4)).astype(f)
        xb[:, 0] += np.arange(n_distinct, dtype=f) * f(8.0)          # distinct base rows
        Ds.append(D)
        Xs.append(xb[idx])
    return np.stack(Ds), np.stack(Xs)
