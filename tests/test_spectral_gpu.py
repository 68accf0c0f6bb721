"""cluster_algo 'spectral' on the device (modules/cluster/spectral.py, cluster.py:262-272).

What is asserted, and why not medoid indices on generic inputs: the reference takes the K trailing singular vectors of a fp32
LAPACK SVD; where singular values coincide to rounding the basis of that eigenspace is solver-specific, and on token-like
inputs the reference does not reproduce its own medoids under a float64 solve.  So:
  * graph construction (heat kernel / KNN / spatial-temporal mask) and L_sym: vs the reference fixtures, 1e-6;
  * the eigensolver: residual, orthonormality, the reference's singular values to 1e-5, the projector onto the K trailing
    vectors where the K-th gap is open;
  * planted partitions (gap ~ 2,000x): the module's assignment equals the reference's, every medoid sits in its own group, and
    with aggregation='mean' - where the output depends on the partition only - the module output equals the reference
    module's bit for bit; also inside the fused encoder;
  * generic Gaussian input: the normalised cut of the partition within 10 % of the reference partition's.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cluster_oracle as co                                 # noqa: E402
from oracle.recipes import SPECTRAL_CASES, planted_tokens, planted_group               # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def sg():
    return np.load(os.path.join(GOLD, "spectral_golden.npz"))


@pytest.fixture(scope="module")
def r2():
    return np.load(os.path.join(GOLD, "r2_golden.npz"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_knn_graph_laplacian_matches_reference(sg):
    from centerclip_amd.cluster.spectral import spectral_laplacian
    X = dev(sg["knn_x"])
    sigma, knn_k = float(sg["knn_cfg"][0]), int(sg["knn_cfg"][1])
    for tag, g in (("knn", None), ("knn_graph", torch.from_numpy(sg["knn_graph_mask"]))):
        L, W = spectral_laplacian(X, sigma=sigma, mode="KNN", knn_k=knn_k, spatial_temporal_graph=g, return_affinity=True)
        # the same edges survive (expf vs torch.exp differ by an ulp: the k-th value test is consistent inside either)
        assert np.array_equal(W.cpu().numpy() != 0, sg[f"{tag}_w"] != 0)
        np.testing.assert_allclose(W.cpu().numpy(), sg[f"{tag}_w"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(L.cpu().numpy(), sg[f"{tag}_lsym"], rtol=0, atol=2e-6)


def test_eigensolver_against_the_reference_svd(r2):
    """The stored L_sym / U / S of the reference (torch.linalg.svd): eigenpairs, singular values, projector."""
    from centerclip_amd.cluster.spectral import spectral_embedding
    L = dev(r2["sp_lsym"])
    S, U = torch.from_numpy(r2["sp_s"]), torch.from_numpy(r2["sp_u"]).double()
    for K in (1, 4, 8, 16, 48):
        Q, ev = spectral_embedding(L, K, correct_sign=True)
        assert Q.shape == (3, 48, K) and ev.shape == (3, K)
        Qd, Ld, evd = Q.cpu().double(), L.cpu().double(), ev.cpu().double()
        assert float((Ld @ Qd - Qd * evd[:, None, :]).abs().max()) < 2e-5                  # L q = lambda q
        assert float((Qd.transpose(1, 2) @ Qd - torch.eye(K, dtype=torch.float64)).abs().max()) < 1e-5
        assert float((ev.cpu() - S[:, -K:]).abs().max()) < 1e-5                            # the reference's order
        if K < 48:
            gap = (S[:, -K - 1] - S[:, -K]).min()
            if gap > 1e-3:
                Ur = U[:, :, -K:]
                assert float((Qd @ Qd.transpose(1, 2) - Ur @ Ur.transpose(1, 2)).abs().max()) < 1e-3
        # the sign rule of batch_sign_flip_rasmus_bro: sum_i sign(q_i) q_i^2 > 0 for every column
        assert bool(((torch.sign(Qd) * Qd * Qd).sum(dim=1) > 0).all())
    # without the sign correction the vectors are the same up to sign
    Q0, _ = spectral_embedding(L, 8, correct_sign=False)
    Q1, _ = spectral_embedding(L, 8, correct_sign=True)
    assert bool(((Q0 - Q1).abs().amax(dim=1) < 1e-6).logical_or((Q0 + Q1).abs().amax(dim=1) < 1e-6).all())


@pytest.mark.parametrize("N,K", [(2, 1), (5, 2), (65, 7), (98, 49), (130, 10), (147, 49), (196, 49), (197, 5), (230, 12), (392, 49),
                                 (588, 100), (640, 128), (300, 150), (640, 160), (784, 160), (832, 192)])
def test_eigensolver_shapes(N, K):
    """Odd / tiny / multi-register / LDS-resident / global-memory problem sizes - among them the reference's own spectral
    settings, 12 -> 6 / 4 / 3 frames of 49 tokens with K = 49 (scripts/lsmdc.sh:128-152) and ViT-B/16's four frames of 196 tokens
    with K = 160 (scripts/activitynet.sh:104-122; more vectors than one back-transformation pass holds): eigenpairs of a random
    graph Laplacian against float64 eigh (values; vectors through the residual)."""
    from centerclip_amd.cluster.spectral import spectral_laplacian, spectral_embedding
    gen = torch.Generator().manual_seed(N)
    X = (torch.randn(2, N, 16, generator=gen) * 0.7).to(DEV)
    L = spectral_laplacian(X, sigma=2.0)
    Q, ev = spectral_embedding(L, K, correct_sign=True)
    w = torch.linalg.eigvalsh(L.cpu().double())                          # ascending
    want = torch.flip(w[:, :K], dims=[1])                                 # the reference's order: descending among the K smallest
    assert float((ev.cpu().double() - want).abs().max()) < 2e-5
    Qd = Q.cpu().double()
    assert float((L.cpu().double() @ Qd - Qd * ev.cpu().double()[:, None, :]).abs().max()) < 3e-5
    assert float((Qd.transpose(1, 2) @ Qd - torch.eye(K, dtype=torch.float64)).abs().max()) < 2e-5


@pytest.mark.parametrize("solver", ["direct", "jacobi"])
def test_eigensolver_tight_clusters_both_solvers(solver):
    """Planted partitions, eigenvalue 0 of multiplicity K up to the coupling: the direct solver (eig.hip: fp64 shifts, the
    analytic reflector) reaches the fp32 floor, the Jacobi kernel - selected through the solver argument, it is what shapes outside
    the direct solver's scope run - its documented bounds.  Shapes cover the three register layouts of the LDS kernel, the
    global-memory kernel (N > 196) and the fallback (K too large for the LDS layout at N = 196)."""
    from centerclip_amd import _lib as L
    from centerclip_amd.cluster.spectral import spectral_embedding
    from oracle import probe_tridiag as pt
    rng = np.random.default_rng(7)
    cases = [(196, 49, 1e-6), (196, 49, 0.0), (96, 24, 1e-8), (48, 12, 1e-4), (196, 64, 1e-6), (200, 50, 1e-6), (392, 98, 1e-6)]
    for N, K, leak in cases:
        parts = 49 if (N, K) == (196, 64) else K
        Lm = np.stack([pt.planted(rng, N, parts, leak, perm) for perm in (False, True)])
        Q, ev = spectral_embedding(dev(Lm), K, correct_sign=True, solver="jacobi" if solver == "jacobi" else "auto")
        Qd, Ld, evd = Q.cpu().double(), torch.from_numpy(Lm).double(), ev.cpu().double()
        res = float((Ld @ Qd - Qd * evd[:, None, :]).abs().max())
        orth = float((Qd.transpose(1, 2) @ Qd - torch.eye(K, dtype=torch.float64)).abs().max())
        everr = float((evd - torch.linalg.eigvalsh(Ld)[:, :K].flip(-1)).abs().max())
        direct = solver == "direct" and not (N == 196 and K > 49)                  # (that shape: the Jacobi kernel)
        assert res < (3e-6 if direct else 2e-5), (N, K, leak, res)
        assert orth < (3e-6 if direct else 1e-5), (N, K, leak, orth)
        assert everr < (2e-6 if direct else 3e-4), (N, K, leak, everr)       # (one-sided Jacobi: |mu| of nearly equal rows)


@pytest.mark.parametrize("P,N,K", [(3, 3, 1), (3, 3, 3), (2, 4, 2), (5, 33, 33), (2, 64, 64), (300, 40, 7), (2, 196, 1), (2, 197, 1),
                                   (2, 200, 128), (3, 257, 64), (2, 321, 65), (1, 449, 3), (70, 210, 9)])
def test_eigensolver_edge_shapes(P, N, K):
    """Corners of the direct solver's dispatch (eig.hip): smallest sizes, K = N, K = 1, more problems than CUs, the first
    sizes of the global-memory kernel and of each of its register layouts, 8 lanes per vector (K > 64); zero-padded Q columns."""
    from centerclip_amd.cluster.spectral import spectral_laplacian
    gen = torch.Generator().manual_seed(1000 * N + K)
    X = (torch.randn(P, N, 12, generator=gen) * 0.6).to(DEV)
    L = spectral_laplacian(X, sigma=2.0)
    Q4, ev, _ = torch.ops.centerclip.spectral_embedding(L.contiguous(), K, True)
    K4 = (K + 3) // 4 * 4
    assert Q4.shape == (P, N, K4) and ev.shape == (P, K)
    assert float(Q4[:, :, K:].abs().max()) == 0.0 if K4 > K else True
    Qd, Ld, evd = Q4[:, :, :K].cpu().double(), L.cpu().double(), ev.cpu().double()
    want = torch.flip(torch.linalg.eigvalsh(Ld)[:, :K], dims=[1])
    assert float((evd - want).abs().max()) < 3e-6
    assert float((Ld @ Qd - Qd * evd[:, None, :]).abs().max()) < 3e-6
    assert float((Qd.transpose(1, 2) @ Qd - torch.eye(K, dtype=torch.float64)).abs().max()) < 4e-6
    assert bool(((torch.sign(Qd) * Qd * Qd).sum(dim=1) > 0).all())


def _module(cfg, agg):
    from centerclip_amd.cluster import TokenClusterInter
    return TokenClusterInter(algorithm="spectral", block_id=7, before_cluster_num=cfg["n"], cluster_num=cfg["K"],
                             before_block_frames=cfg["T"], after_block_frames=cfg["T_new"], original_frame=cfg["T"],
                             distance="euclidean", threshold=1e-6, iter_limit=100, id_sort=True, aggregation=agg, split_size=16,
                             norm_p=2.0, spectral_graph=cfg["graph"], spectral_sigma=cfg["sigma"], spectral_knn_k=cfg["knn_k"],
                             spectral_spatial_temporal_graph=bool(cfg.get("spg")), transformer_width=cfg["W"],
                             svd_correct_sign=1).to(DEV).eval()


@pytest.mark.parametrize("tag", list(SPECTRAL_CASES))
def test_spectral_module_on_planted_partitions(sg, tag):
    cfg = SPECTRAL_CASES[tag]
    x = dev(planted_tokens(cfg))
    T, Tn, n, K = cfg["T"], cfg["T_new"], cfg["n"], cfg["K"]
    N = (T // Tn) * n
    # cluster means: a function of the partition only -> the reference module's output, bit for bit (both layouts)
    mod = _module(cfg, "mean")
    y, res = mod(x)
    assert res is None and np.array_equal(y.cpu().numpy(), sg[f"{tag}_mean_out"])
    yf = mod.cluster_frame_major(x.permute(1, 0, 2).contiguous())
    assert np.array_equal(yf.permute(1, 0, 2).cpu().numpy(), sg[f"{tag}_mean_out"])
    # medoid tokens: one medoid per planted group (which member is picked inside a group of near-identical embedding rows is
    # decided by rounding noise of the decomposition, for the reference too)
    mod = _module(cfg, None)
    y, _ = mod(x)
    med = mod.last_medoids.cpu()
    assert med.shape == sg[f"{tag}_none_medoids"].shape
    assert torch.equal(planted_group(med, N, K), torch.arange(K).expand_as(med))
    ref_med = torch.from_numpy(sg[f"{tag}_none_medoids"].astype(np.int64))
    assert torch.equal(planted_group(ref_med, N, K), torch.arange(K).expand_as(ref_med))
    # the output is the gather of exactly those tokens + the CLS means
    want = co.literal_token_cluster_variant(x.cpu(), T, Tn, K, "kmediods++", None, medoids=med,
                                            assign=torch.zeros(med.shape[0], N, dtype=torch.long))
    assert torch.equal(y.cpu(), want)
    # autograd: the selection is a constant of the backward pass, the gradient is that of the gather + CLS mean
    xg = x.clone().requires_grad_(True)
    yg, _ = mod(xg)
    G = torch.randn_like(yg)
    yg.backward(G)
    xo = x.cpu().clone().requires_grad_(True)
    co.literal_token_cluster_variant(xo, T, Tn, K, "kmediods++", None, medoids=mod.last_medoids.cpu(),
                                     assign=torch.zeros(med.shape[0], N, dtype=torch.long)).backward(G.cpu())
    assert torch.equal(xg.grad.cpu(), xo.grad)


def test_spectral_generic_input_normalised_cut(sg):
    """No index target on generic inputs: the partition's normalised cut (what spectral clustering minimises) is compared with
    the reference partition's on the same affinity."""
    from centerclip_amd.cluster.spectral import batch_spectral_clustering, spectral_laplacian
    X = dev(sg["generic_x"])
    asg, med = batch_spectral_clustering(X, 8, mode="HeatKernel", metric="euclidean", threshold=1e-6, iter_limit=100, norm_p=2.0,
                                         correct_sign=True, split_size=16, sigma=2.0)
    assert asg.shape == (4, 64) and med.shape == (4, 8) and bool((med[:, 1:] > med[:, :-1]).all())
    assert bool((torch.gather(asg, 1, med) == torch.arange(8, device=DEV)).all())           # medoid k carries label k
    _, W = spectral_laplacian(X, sigma=2.0, return_affinity=True)
    mine = co.normalized_cut(W.cpu(), asg.cpu(), 8)
    ref = co.normalized_cut(W.cpu(), torch.from_numpy(sg["generic_assign"].astype(np.int64)), 8)
    assert bool((mine <= ref * 1.10 + 1e-9).all()), (mine.tolist(), ref.tolist())
    # an external decomposition plugs in (the reference's own choice on this device)
    asg2, med2 = batch_spectral_clustering(X, 8, sigma=2.0, norm_p=2.0, threshold=1e-6, iter_limit=100, correct_sign=True,
                                           split_size=16, eigensolver=lambda Ls: torch.linalg.svd(Ls, full_matrices=False))
    assert asg2.shape == asg.shape and med2.shape == med.shape


def test_spectral_block_inside_the_fused_encoder():
    """A ViT whose cluster block uses cluster_algo='spectral': the fused encoder (cc_vit_encode) runs the whole selection on the
    device; its medoids are those of the module applied to the hidden state, and the features follow the oracle forward
    given those medoids."""
    from argparse import Namespace
    from centerclip_amd.clip import build_clip_model
    import test_r2_gpu as t2
    gc = np.load(t2.CLIPG)
    sd = {k[3:]: torch.from_numpy(gc[k].astype(np.float32) if gc[k].dtype == np.float16 else gc[k]) for k in gc.files if k.startswith("sd/")}
    T = int(gc["cfg"][11])
    args = Namespace(cluster_inter=1, cluster_algo='spectral', max_frames=T, target_frames_blocks=[4, 2, 2],
                     cluster_num_blocks=[16, 6, 6], cluster_distance='euclidean', cluster_threshold=1e-6, cluster_iter_limit=100,
                     minkowski_norm_p=2.0, pretrained_clip_name='ViT-B/32', aggregation=None, pre_norm=False, spectral_sigma=2.0,
                     spectral_graph='HeatKernel', spectral_knn_k=1, spectral_spg=0, svd_correct_sign=1, cluster_embedding=0,
                     cluster_frame_embedding=0, save_feature_path=None)
    model, _ = build_clip_model(dict(sd), args=args)
    model = model.to(DEV).eval()
    video = torch.from_numpy(gc["video"]).to(DEV)
    feat, _ = model.visual.encode(video, T, want_medoids=True)
    med = model.visual.last_medoids
    assert med is not None and med.shape[1] == 6 and bool((med[:, 1:] > med[:, :-1]).all())
    feat2, _ = model.visual.encode(video, T)
    assert torch.equal(feat, feat2) and bool(torch.isfinite(feat).all())
    from oracle import clip_oracle as clo
    ref = clo.visual_forward(sd, video.cpu(), T, cluster_plan={1: (2, 6)}, forced_medoids={1: med.cpu()})
    nrm = lambda v: v / v.norm(dim=-1, keepdim=True)
    assert float((nrm(feat.cpu()) - nrm(ref)).abs().max()) <= 1e-3


def test_cross_set_pairwise_distance_matches_reference(sg):
    """pairwise_distance(data1, data2) with two different token sets (cluster_utils.py:8-43; not on the retrieval path):
    batched and 2-d inputs, N1 != N2, Minkowski p in {1, 2, 3}, cosine, both flags - against the reference's outputs."""
    from centerclip_amd.cluster import pairwise_distance
    from oracle.recipes import CROSS_DIST_CASES, cross_dist_inputs
    for tag, cfg in CROSS_DIST_CASES.items():
        a, b = cross_dist_inputs(cfg)
        d = pairwise_distance(dev(a), dev(b), metric=cfg["metric"], self_nearest=cfg["self_nearest"],
                              all_negative=cfg["all_negative"], p=cfg["p"])
        want = sg[f"xd_{tag}"]
        assert tuple(d.shape) == want.shape, tag
        np.testing.assert_allclose(d.cpu().numpy(), want, rtol=0, atol=2e-4 if cfg["p"] == 2.0 and cfg["metric"] == "euclidean" else 2e-5)
    with pytest.raises(Exception):                                     # the reference's diagonal indexing needs N2 <= N1
        pairwise_distance(dev(np.zeros((4, 8), np.float32)), dev(np.zeros((6, 8), np.float32)), self_nearest=True)
