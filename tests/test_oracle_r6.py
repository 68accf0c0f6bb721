"""CPU (no GPU): the oracle's literal k-medoids (oracle/cluster_oracle.py literal_batch_kmedoids_with_split, the restatement of
modules/cluster/fast_kmeans.py:14-97 INCLUDING the chunk-mean stop test :85-88) against the round-6 fixtures the imported
reference produced for LOOSE thresholds (oracle/gen_golden_r6.py -> tests/golden/r6_golden.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import cluster_oracle as co
from oracle.recipes import LOOSE_THRESHOLD_CASES, P1_WIDE_CASES, lattice, loose_threshold_inputs

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g6():
    return np.load(os.path.join(HERE, "golden", "r6_golden.npz"))


@pytest.mark.parametrize("tag", sorted(LOOSE_THRESHOLD_CASES))
def test_literal_oracle_matches_reference_at_loose_thresholds(g6, tag):
    seed, P, N, W, K, split, iters, distance, pre_norm, id_sort, _, _ = LOOSE_THRESHOLD_CASES[tag]
    X = torch.from_numpy(loose_threshold_inputs(tag))
    thr = float(g6[f"{tag}_threshold"][0])
    a, m = co.literal_batch_kmedoids_with_split(X, K, distance, thr, iters, id_sort, 2.0, split, pre_norm)
    assert np.array_equal(m.numpy(), g6[f"{tag}_medoids"].astype(np.int64))
    assert np.array_equal(a.numpy(), g6[f"{tag}_assign"].astype(np.int64))
    assert int(g6[f"{tag}_differs_from_fixed_point"][0]) == 1          # the fixtures are cases the fixed-point test gets wrong


def test_aten_row_sums_restates_torch_for_the_shift_shapes():
    """The three sums of the stop test (over W, over K, over the chunk) go through ATen's contiguous row sum: the oracle's
    restatement (aten_row_sums) == torch.sum bit for bit at those lengths."""
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 4, 5, 7, 8, 10, 16, 25, 49, 64, 100, 512, 768, 1024):
        M = (rng.standard_normal((6, n)) * 3).astype(np.float32) ** 2
        assert np.array_equal(co.aten_row_sums(M), torch.from_numpy(M).sum(dim=-1).numpy()), n


def test_literal_oracle_matches_reference_above_4095_tokens(g6):
    """The oracle's literal k-medoids at N = 4,500 (the smaller of the two wide fixtures: the [B, K, N, N] temporaries of the
    literal form are 0.5 GB there) == the reference's indices."""
    seed, P, N, W, K, split, iters = P1_WIDE_CASES["p1w_4500"]
    X = torch.from_numpy(lattice(seed, (P, N, W)))
    a, m = co.literal_batch_kmedoids_with_split(X, K, "euclidean", 1e-6, iters, True, 2.0, split, False)
    assert np.array_equal(m.numpy(), g6["p1w_4500_medoids"].astype(np.int64))
    assert np.array_equal(a.numpy(), g6["p1w_4500_assign"].astype(np.int64))
