"""Pin the CPU oracle (oracle/cluster_oracle.py) against fixtures captured from the
imported reference (tests/golden/cluster_golden.npz, written by oracle/gen_golden.py).

CPU only.  Parity levels follow SURVEY.md §8(c):
  C3/C4        literal op restatement == reference output (same ATen ops; tolerance 0 on
               this torch build, 1e-5 relative guard for a different BLAS on another host)
  P0           selection from the STORED fp32 distance tensor: indices bit-exact
  P1 / P2      from-X on lattice / dyadic inputs: indices bit-exact
  C1           module output (gather + CLS mean + restack): exact
"""
import os
import numpy as np
import pytest
import torch

from oracle import cluster_oracle as co
from oracle.recipes import (DUPLICATE_CASES, GRAD_CASES, VARIANT_CASES, grad_output, duplicate_token_problem, dyadic, fullmant, lattice,
                            variant_input)

t = torch.from_numpy


@pytest.mark.parametrize("tag", ["n12", "n32"])
def test_c3_pairwise_distance(cluster_golden, tag):
    g = cluster_golden
    X = t(g[f"c3_{tag}_x"])
    for metric, p, mtag in (("euclidean", 2.0, "l2"), ("euclidean", 1.0, "l1"), ("cosine", 2.0, "cos")):
        for an in (False, True):
            for sn in (False, True):
                d = co.literal_pairwise_distance(X, X, metric=metric, self_nearest=sn, all_negative=an, p=p)
                np.testing.assert_allclose(d.numpy(), g[f"c3_{tag}_{mtag}_an{int(an)}_sn{int(sn)}"], rtol=1e-5, atol=1e-5)
    d2 = co.literal_pairwise_distance(X[0], X[0], metric="euclidean", self_nearest=True, all_negative=True)
    np.testing.assert_allclose(d2.numpy(), g[f"c3_{tag}_l2_2d"], rtol=1e-5, atol=1e-5)


def test_c3_unknown_metric_raises():
    with pytest.raises(NotImplementedError):
        co.literal_pairwise_distance(torch.zeros(2, 3), torch.zeros(2, 3), metric="manhattan")


def test_c4_kkz_with_ties(cluster_golden):
    g = cluster_golden
    D, X = t(g["c4_d"]), t(g["c4_x"])
    med = co.literal_kkz(torch.norm(X, dim=-1), D, 7)
    assert np.array_equal(med.numpy(), g["c4_batch"])
    for b in range(D.shape[0]):
        first = int(torch.argmax(torch.norm(X[b], dim=-1)))
        _, m, _ = co.select_streamlined(D[b].numpy(), first, 7, iter_limit=0, id_sort=False)
        assert np.array_equal(m, g["c4_batch"][b])
        # the reference's non-batched KKZ (cluster_utils.py:95-101, not on the hot path) walks
        # COLUMNS of D where the batched one walks rows: on a non-symmetric D they differ.
        _, mt, _ = co.select_streamlined(D[b].numpy().T.copy(), first, 7, iter_limit=0, id_sort=False)
        assert np.array_equal(mt, g["c4_single"][b])


@pytest.mark.parametrize("tag", ["p0_small_l2", "p0_small_cos", "p0_real_l2", "p0_n392_l2"])
def test_p0_selection_from_stored_distance(cluster_golden, tag):
    g = cluster_golden
    D, nrm, K = t(g[f"{tag}_d"]), t(g[f"{tag}_norm"]), int(g[f"{tag}_k"])
    a, m, _ = co.literal_select(D, nrm, K, X=None, iter_limit=100, id_sort=True)
    assert np.array_equal(m.numpy(), g[f"{tag}_medoids"].astype(np.int64))
    assert np.array_equal(a.numpy(), g[f"{tag}_assign"].astype(np.int64))
    a, m, _ = co.literal_select(D, nrm, K, X=None, iter_limit=100, id_sort=False)
    assert np.array_equal(m.numpy(), g[f"{tag}_medoids_nosort"].astype(np.int64))
    assert np.array_equal(a.numpy(), g[f"{tag}_assign_nosort"].astype(np.int64))
    # the streamlined formulation (the HIP kernel's algorithm) reproduces the same indices
    for b in range(D.shape[0]):
        first = int(torch.argmax(nrm[b]))
        a_s, m_s, _ = co.select_streamlined(D[b].numpy(), first, K, iter_limit=100, id_sort=True)
        assert np.array_equal(m_s, g[f"{tag}_medoids"][b].astype(np.int64))
        assert np.array_equal(a_s, g[f"{tag}_assign"][b].astype(np.int64))


def test_p0_update_step_ties(cluster_golden):
    g = cluster_golden
    D, X = g["p0_tie_d"], g["p0_tie_x"]
    a, m, _ = co.literal_select(t(D), torch.norm(t(X), dim=-1), 5, X=t(X), threshold=1e-6, iter_limit=50)
    assert np.array_equal(m.numpy(), g["p0_tie_medoids"].astype(np.int64))
    assert np.array_equal(a.numpy(), g["p0_tie_assign"].astype(np.int64))
    for b in range(D.shape[0]):
        first = int(np.argmax(np.linalg.norm(X[b], axis=-1)))
        a_s, m_s, _ = co.select_streamlined(D[b], first, 5, iter_limit=50)
        assert np.array_equal(m_s, g["p0_tie_medoids"][b].astype(np.int64))
        assert np.array_equal(a_s, g["p0_tie_assign"][b].astype(np.int64))


@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 8, 9, 31, 32, 33, 49, 196, 197, 511, 512, 543, 544, 588, 640, 1024, 8225])
def test_aten_sum_association_restatement(n):
    """oracle.aten_row_sums restates the association of ATen's CPU sum (the arithmetic of fast_kmeans.py:82);
    it must equal torch.sum bit for bit, with and without the masked-out zeros of the reference's sub_matrix."""
    rng = np.random.default_rng(1000 + n)
    x = (rng.standard_normal((48, n)) * 10 ** rng.uniform(-2, 3, (48, 1))).astype(np.float32)
    masked = -np.abs(x) * (rng.random((48, n)) < 0.1)
    for m in (x, masked.astype(np.float32)):
        ref = t(m).view(2, 3, 8, n).sum(-1).reshape(48).numpy()
        assert np.array_equal(co.aten_row_sums(m), ref)


@pytest.mark.parametrize("tag", list(DUPLICATE_CASES))
def test_p0_duplicate_tokens(cluster_dup_golden, tag):
    """Colliding tokens: candidates of one cluster tie exactly in real arithmetic and are separated only by
    how their fp32 row sums round, so the selection depends on the summation order of fast_kmeans.py:82."""
    g = cluster_dup_golden
    seed, P, nd, N, K, layout = DUPLICATE_CASES[tag]
    D, X = duplicate_token_problem(seed, P, nd, N, layout)
    for sort, sfx in ((True, ""), (False, "_nosort")):
        a, m, _ = co.literal_select(t(D), torch.norm(t(X), dim=-1), K, X=t(X), threshold=1e-6, iter_limit=100,
                                    id_sort=sort)
        assert np.array_equal(m.numpy(), g[f"{tag}_medoids{sfx}"].astype(np.int64))
        assert np.array_equal(a.numpy(), g[f"{tag}_assign{sfx}"].astype(np.int64))
        for b in range(P):
            first = int(np.argmax(np.linalg.norm(X[b], axis=-1)))
            a_s, m_s, _ = co.select_streamlined(D[b], first, K, iter_limit=100, id_sort=sort)
            assert np.array_equal(m_s, g[f"{tag}_medoids{sfx}"][b].astype(np.int64))
            assert np.array_equal(a_s, g[f"{tag}_assign{sfx}"][b].astype(np.int64))


P1 = ["p1_cfg2", "p1_cfg3", "p1_cfg4", "p1_cfg5", "p1_ragged", "p1_k_eq_n", "p1_k1"]


@pytest.mark.parametrize("tag", P1)
def test_p1_lattice_from_x(cluster_golden, tag):
    g = cluster_golden
    seed, P, N, W, K, split, iters = [int(v) for v in g[f"{tag}_cfg"]]
    if P * N * N * K > 4e8:       # keep the CPU suite short: check a prefix of whole chunks
        P = split * max(1, min(P // split, 1))
    X = t(lattice(seed, (int(g[f"{tag}_cfg"][1]), N, W)))[:P]
    a, m = co.literal_batch_kmedoids_with_split(X, K, "euclidean", 1e-6, iters, True, 2.0, split, False)
    assert np.array_equal(m.numpy(), g[f"{tag}_medoids"][:P].astype(np.int64))
    assert np.array_equal(a.numpy(), g[f"{tag}_assign"][:P].astype(np.int64))


@pytest.mark.parametrize("tag", ["p1_cfg3", "p1_ragged", "p1_k_eq_n", "p1_k1"])
def test_p1_streamlined_with_exact_zero_diagonal(cluster_golden, tag):
    """The kernel's own arithmetic (Gram-diagonal norms, exact zero diagonal, member-list sums)
    gives the reference's indices on lattice inputs - chunk by chunk, as the reference splits."""
    g = cluster_golden
    seed, P, N, W, K, split, iters = [int(v) for v in g[f"{tag}_cfg"]]
    X = lattice(seed, (P, N, W))
    P = min(P, 2 * split)
    for c0 in range(0, P, split):
        Xc = X[c0:min(c0 + split, P)]
        D = co.exact_zero_diag_distance(Xc)
        for b in range(Xc.shape[0]):
            first = int(np.argmax(np.sqrt((Xc[b].astype(np.float64) ** 2).sum(-1)).astype(np.float32)))
            a_s, m_s, _ = co.select_streamlined(D[b], first, K, iter_limit=iters)
            assert np.array_equal(m_s, g[f"{tag}_medoids"][c0 + b].astype(np.int64))
            assert np.array_equal(a_s, g[f"{tag}_assign"][c0 + b].astype(np.int64))


@pytest.mark.parametrize("tag", ["p2_cfg2", "p2_cfg3", "p2_small"])
def test_p2_l1_from_x(cluster_golden, tag):
    g = cluster_golden
    seed, P, N, W, K, split, iters = [int(v) for v in g[f"{tag}_cfg"]]
    X = t(dyadic(seed, (P, N, W)))
    a, m = co.literal_batch_kmedoids_with_split(X, K, "euclidean", 1e-6, iters, True, 1.0, split, False)
    assert np.array_equal(m.numpy(), g[f"{tag}_medoids"].astype(np.int64))
    assert np.array_equal(a.numpy(), g[f"{tag}_assign"].astype(np.int64))


C1 = ["c1_12_3", "c1_12_4", "c1_12_6", "c1_64_8", "c1_12_12", "c1_b16"]


@pytest.mark.parametrize("tag", C1)
def test_c1_token_cluster_module(cluster_golden, tag):
    g = cluster_golden
    seed, B, T, T_new, n, W, K, split = [int(v) for v in g[f"{tag}_cfg"]]
    x = t(lattice(seed, (1 + n, B * T, W)))
    y = co.literal_token_cluster(x, T, T_new, K, "euclidean", 1e-6, 100, 2.0, split, False)
    assert tuple(y.shape) == (1 + K, B * T_new, W)
    assert np.array_equal(y.numpy(), g[f"{tag}_out"])


# ------------------------------------------------------------------------------- N2 variants
@pytest.mark.parametrize("tag", list(VARIANT_CASES))
def test_n2_variant_oracle_reproduces_reference(cluster_variants_golden, tag):
    """literal_token_cluster_variant == the reference module's output (fixture) for aggregation / cluster_embed /
    adaptive_cls / pooling / sparse_sampling; the k-medoids cases are evaluated on the reference's own assignment,
    so generic-float inputs do not depend on this host's cdist rounding."""
    g, cfg = cluster_variants_golden, VARIANT_CASES[tag]
    x, embed, mult = variant_input(cfg)
    kw = {}
    if f"{tag}_assign" in g.files:
        kw = dict(assign=t(g[f"{tag}_assign"].astype(np.int64)), medoids=t(g[f"{tag}_medoids"].astype(np.int64)))
    y = co.literal_token_cluster_variant(t(x), cfg["T"], cfg["T_new"], cfg["K"], cfg["algorithm"], cfg["aggregation"],
                                         None if embed is None else t(embed), None if mult is None else t(mult), **kw)
    assert np.array_equal(y.numpy(), g[f"{tag}_out"], equal_nan=True)
    if cfg["inp"] == "lattice" and kw:            # exact inputs: the oracle's own k-medoids finds the same assignment
        y2 = co.literal_token_cluster_variant(t(x), cfg["T"], cfg["T_new"], cfg["K"], cfg["algorithm"], cfg["aggregation"],
                                              None if embed is None else t(embed), None if mult is None else t(mult))
        assert np.array_equal(y2.numpy(), g[f"{tag}_out"], equal_nan=True)


@pytest.mark.parametrize("tag", list(GRAD_CASES))
def test_n4_variant_oracle_gradients_reproduce_reference(tag):
    """torch.autograd through the oracle's restatement of TokenClusterInter.forward == the gradients the reference module
    produced (tests/golden/cluster_grad_golden.npz, oracle/gen_golden_grad.py): d/dx exactly (gathers, divisions by
    counts), the parameter gradients to rounding (sums over segments / batch in an unspecified order)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cluster_grad_golden.npz"))
    cfg = GRAD_CASES[tag]
    x, embed, mult = variant_input(cfg)
    xt = t(x).clone().requires_grad_(True)
    et = None if embed is None else t(embed).clone().requires_grad_(True)
    mt = None if mult is None else t(mult).clone().requires_grad_(True)
    kw = {}
    if f"{tag}_assign" in g.files:
        kw = dict(assign=t(g[f"{tag}_assign"].astype(np.int64)), medoids=t(g[f"{tag}_medoids"].astype(np.int64)))
    y = co.literal_token_cluster_variant(xt, cfg["T"], cfg["T_new"], cfg["K"], cfg["algorithm"], cfg["aggregation"], et, mt, **kw)
    y.backward(t(grad_output(cfg)))
    assert np.array_equal(xt.grad.numpy(), g[f"{tag}_gx"])
    if et is not None:
        np.testing.assert_allclose(et.grad.numpy(), g[f"{tag}_gembed"], rtol=1e-6, atol=1e-6)
    if mt is not None:
        np.testing.assert_allclose(mt.grad.numpy(), g[f"{tag}_gmult"], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("n,C", [(1, 64), (3, 32), (5, 40), (16, 64), (17, 96), (33, 64), (196, 768), (257, 64), (588, 32),
                                 (640, 100)])
def test_aten_outer_sum_association_restatement(n, C):
    """oracle.aten_outer_sums restates ATen's sum over a non-innermost dimension (segment / cluster means)."""
    M = fullmant(500 + n, (n, C))
    masked = M * (np.random.default_rng(n).random((n, 1)) < 0.2)
    for m in (M, masked.astype(np.float32)):
        tt = t(m).reshape(1, n, C).repeat(3, 1, 1)
        assert np.array_equal(co.aten_outer_sums(m), tt.sum(dim=1)[1].numpy())
        assert np.array_equal(co.aten_outer_sums(m) / np.float32(n), tt.mean(dim=1)[2].numpy())


@pytest.mark.parametrize("tag", ["p1n_ragged", "p1n_b16_64f"])
def test_p1_lattice_above_1023_tokens(tag):
    """Round 5: the oracle against the reference's indices above 1,023 tokens per problem (oracle/gen_golden_r5.py): the
    cascade of ATen's row sum has folded more than once there."""
    g5 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r5_golden.npz"))
    seed, P, N, W, K, split, iters = [int(v) for v in g5[f"{tag}_cfg"]]
    a, m = co.literal_batch_kmedoids_with_split(t(lattice(seed, (P, N, W))), K, "euclidean", 1e-6, iters, True, 2.0, split, False)
    assert np.array_equal(m.numpy(), g5[f"{tag}_medoids"].astype(np.int64))
    assert np.array_equal(a.numpy(), g5[f"{tag}_assign"].astype(np.int64))
