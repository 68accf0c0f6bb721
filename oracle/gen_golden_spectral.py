"""Spectral-clustering fixtures captured from the IMPORTED reference (dev container only; see gen_golden.py for the rules):

  knn_*      constructW(mode='KNN') + the normalised Laplacian, with and without the spatial-temporal mask (spectral.py:42-52,
             89-105)
  planted_*  the reference's TokenClusterInter(algorithm='spectral') on tokens drawn around K well-separated centres
             (recipes.planted_tokens): with aggregation='mean' the module output depends on the partition only, which every
             correct eigensolver recovers; stored with the medoids / assignment batch_spectral_clustering returned and the
             singular values around the K-th (the gap that makes the case solver-independent)
  xd_*       pairwise_distance(data1, data2) of two different token sets (recipes.CROSS_DIST_CASES)
  generic_*  a Gaussian input with the reference's assignment, for the normalised-cut comparison (no index target: the
             reference's own float64 run disagrees with it)

    python oracle/gen_golden_spectral.py   ->  tests/golden/spectral_golden.npz
"""
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = "/root/reference"
sys.path.insert(0, HERE)
from recipes import SPECTRAL_CASES, planted_tokens  # noqa: E402


def main():
    sys.path.insert(0, os.path.join(REF, "modules"))
    import cluster.cluster as cc
    import cluster.spectral as sp
    t = torch.from_numpy
    out = {}
    # ---- KNN graph
    g = torch.Generator().manual_seed(77)
    B, N, W, sigma, knn_k = 3, 48, 32, 2.5, 6
    X = torch.randn(B, N, W, generator=g) * 0.6
    graph = sp.spatial_temporal_graph(N, 16, s_kernel=3, t_kernel=3)
    for tag, gr in (("knn", None), ("knn_graph", graph)):
        Wm = sp.constructW(X, X, sigma=sigma, mode='KNN', knn_k=knn_k, spatial_temporal_graph=gr)
        d = Wm.sum(dim=-1)
        inv = torch.diag_embed(torch.pow(d, -0.5))
        out[f"{tag}_w"] = Wm.numpy()
        out[f"{tag}_lsym"] = torch.bmm(torch.bmm(inv, torch.diag_embed(d) - Wm), inv).numpy()
    out["knn_x"], out["knn_cfg"] = X.numpy(), np.array([sigma, knn_k], dtype=np.float32)
    out["knn_graph_mask"] = graph.numpy().astype(np.uint8)
    # ---- the module on planted partitions
    for tag, cfg in SPECTRAL_CASES.items():
        x = planted_tokens(cfg)
        for agg in (None, "mean"):
            mod = cc.TokenClusterInter(algorithm="spectral", block_id=7, before_cluster_num=cfg["n"], cluster_num=cfg["K"],
                                       before_block_frames=cfg["T"], after_block_frames=cfg["T_new"], original_frame=cfg["T"],
                                       distance="euclidean", threshold=1e-6, iter_limit=100, id_sort=True, aggregation=agg,
                                       split_size=16, norm_p=2.0, spectral_graph=cfg["graph"], spectral_sigma=cfg["sigma"],
                                       spectral_knn_k=cfg["knn_k"], spectral_spatial_temporal_graph=bool(cfg.get("spg")),
                                       transformer_width=cfg["W"], svd_correct_sign=1)
            mod.eval()
            captured = {}
            orig = cc.batch_spectral_clustering

            def spy(*a, **k):
                r = orig(*a, **k)
                captured["assign"], captured["medoids"] = r[0].clone(), r[1].clone()
                captured["x"] = a[0].clone()
                return r
            cc.batch_spectral_clustering = spy
            try:
                with torch.no_grad():
                    y, _ = mod(t(x))
            finally:
                cc.batch_spectral_clustering = orig
            name = "none" if agg is None else "mean"
            out[f"{tag}_{name}_out"] = y.contiguous().numpy()
            out[f"{tag}_{name}_assign"] = captured["assign"].numpy().astype(np.int16)
            out[f"{tag}_{name}_medoids"] = captured["medoids"].numpy().astype(np.int16)
        # the spectrum around K of the Laplacian the module built (same call as the module makes)
        Wm = sp.constructW(captured["x"], captured["x"], sigma=cfg["sigma"], mode=cfg["graph"], knn_k=mod.spectral_knn_k,
                           spatial_temporal_graph=mod.spg)
        d = Wm.sum(dim=-1)
        inv = torch.diag_embed(torch.pow(d, -0.5))
        S = torch.linalg.svd(torch.bmm(torch.bmm(inv, torch.diag_embed(d) - Wm), inv), full_matrices=False)[1]
        K = cfg["K"]
        out[f"{tag}_spectrum"] = S[:, -(K + 2):].numpy()
        print(tag, "done; singular values around K:", S[0, -(K + 2):].tolist(), flush=True)
    # ---- generic input: assignment for the normalised-cut comparison
    g = torch.Generator().manual_seed(78)
    Xg = torch.randn(4, 64, 32, generator=g) * 0.5
    asg, med = sp.batch_spectral_clustering(Xg, 8, mode='HeatKernel', metric='euclidean', threshold=1e-6, iter_limit=100,
                                            norm_p=2.0, correct_sign=True, split_size=16, sigma=2.0)
    out["generic_x"], out["generic_assign"], out["generic_medoids"] = Xg.numpy(), asg.numpy().astype(np.int16), med.numpy().astype(np.int16)
    # ---- pairwise_distance(data1, data2) for two different token sets (cluster_utils.py:8-43)
    import cluster.cluster_utils as cu
    from recipes import CROSS_DIST_CASES, cross_dist_inputs
    for tag, cfg in CROSS_DIST_CASES.items():
        a, b = cross_dist_inputs(cfg)
        d = cu.pairwise_distance(t(a), t(b), metric=cfg["metric"], self_nearest=cfg["self_nearest"],
                                 all_negative=cfg["all_negative"], p=cfg["p"])
        out[f"xd_{tag}"] = d.numpy()
    path = os.path.join(GOLD, "spectral_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
