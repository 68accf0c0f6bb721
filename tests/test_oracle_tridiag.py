"""CPU: the numerical specification of the direct eigensolver (oracle/probe_tridiag.py = centerclip_amd/csrc/eig.hip phase by
phase, same precisions) against the reference's fixtures - the stored L_sym / singular values of torch.linalg.svd
(modules/cluster/spectral.py:54-61) - and on a tight cluster, where the two variants the kernel does NOT use are shown to fail."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import probe_tridiag as pt                                  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_spec_reproduces_the_reference_spectrum():
    r2 = np.load(os.path.join(GOLD, "r2_golden.npz"))
    Ls, S, U = r2["sp_lsym"], r2["sp_s"], r2["sp_u"].astype(np.float64)
    for L, s, u in zip(Ls, S, U):
        for K in (4, 16):
            lam, Z = pt.smallest_eigenpairs(L, K)
            res, orth, _ = pt.quality(L, lam, Z)
            assert res < 2e-6 and orth < 3e-6
            assert np.abs(lam[::-1] - s[-K:]).max() < 1e-5               # the reference's singular values (descending)
            if s[-K - 1] - s[-K] > 1e-3:                                   # open gap: the same invariant subspace
                Zd = Z.astype(np.float64)
                assert np.abs(Zd @ Zd.T - u[:, -K:] @ u[:, -K:].T).max() < 1e-3


def test_spec_on_fixture_laplacians():
    sg = np.load(os.path.join(GOLD, "spectral_golden.npz"))
    for key in ("knn_lsym", "knn_graph_lsym"):
        for L in sg[key]:
            res, orth, ev = pt.quality(L, *pt.smallest_eigenpairs(L, 6))
            assert res < 2e-6 and orth < 3e-6 and ev < 2e-6


def test_tight_cluster_needs_fp64_shifts_and_the_analytic_reflector():
    rng = np.random.default_rng(3)
    L = pt.planted(rng, 48, 12, 1e-6, False)                               # eigenvalue 0 of multiplicity 12 up to the coupling
    res, orth, ev = pt.quality(L, *pt.smallest_eigenpairs(L, 12))
    assert res < 2e-6 and orth < 3e-6 and ev < 2e-6
    bad = pt.quality(L, *pt.smallest_eigenpairs(L, 12, store_analytic=False))
    assert bad[0] > 20 * res                                               # reflector taken from the updated row: tau does not fit
    low = pt.quality(L, *pt.smallest_eigenpairs(L, 12, high=False, iters=1))
    assert low[0] > 20 * res                                               # fp32 shifts: Gram-Schmidt amplifies the rest of the spectrum
    T = [np.diag(d.astype(np.float64)) + np.diag(e.astype(np.float64), 1) + np.diag(e.astype(np.float64), -1)
         for d, e, _, _ in (pt.tridiagonalise_fused(L), pt.tridiagonalise(L))]
    assert np.abs(np.linalg.eigvalsh(T[0]) - np.linalg.eigvalsh(T[1])).max() < 2e-6       # both forms: the same spectrum
