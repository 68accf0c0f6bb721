"""Generate tests/golden/*.npz by IMPORTING the read-only reference (dev container only).

    python oracle/gen_golden.py [cluster|dup|variants|metrics|clip|all]

The reference (/root/reference, Python/PyTorch) never travels to the GPU box:
what travels is the data this script writes - inputs (or the integer seeds that
regenerate them exactly) and the outputs the reference produced for them.
No reference source text is copied anywhere.  Inputs are built only from
numpy ``default_rng(seed).integers`` so they are bit-reproducible on any host:

  lattice(seed, shape)      integers in [-3, 3]            -> every squared L2
                            distance is an exact fp32 integer (parity level P1)
  dyadic(seed, shape)       sum of four integers in [-32,32] / 64 -> bell-shaped,
                            every L1 distance is exact in fp32 (parity level P2)

See SURVEY.md §8(c) for the parity levels P0-P3.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = "/root/reference"


sys.path.insert(0, HERE)
from recipes import lattice, dyadic  # noqa: E402


# ----------------------------------------------------------------- cluster rows
def gen_cluster():
    sys.path.insert(0, os.path.join(REF, "modules"))
    import cluster.cluster_utils as cu
    import cluster.fast_kmeans as fk
    import cluster.cluster as cc

    out = {}
    t = torch.from_numpy

    # ---- C3 pairwise_distance: N<=25 takes ATen's direct path, N>=26 the Gram path
    for tag, (B, N, W) in {"n12": (2, 12, 8), "n32": (2, 32, 16)}.items():
        X = dyadic(100 + N, (B, N, W))
        out[f"c3_{tag}_x"] = X
        for metric, p, mtag in (("euclidean", 2.0, "l2"), ("euclidean", 1.0, "l1"), ("cosine", 2.0, "cos")):
            for an in (False, True):
                for sn in (False, True):
                    d = cu.pairwise_distance(t(X), t(X), metric=metric, self_nearest=sn, all_negative=an, p=p)
                    out[f"c3_{tag}_{mtag}_an{int(an)}_sn{int(sn)}"] = d.numpy()
        d2 = cu.pairwise_distance(t(X[0]), t(X[0]), metric="euclidean", self_nearest=True, all_negative=True, p=2.0)
        out[f"c3_{tag}_l2_2d"] = d2.numpy()

    # ---- C4 KKZ from a stored D, including exact ties (integer-valued D)
    rng = np.random.default_rng(7)
    Dt = -(rng.integers(1, 6, size=(3, 20, 20)).astype(np.float32)) - 1.0      # many ties
    idx = np.arange(20)
    Dt[:, idx, idx] = -9.0
    Xt = lattice(8, (3, 20, 4))
    out["c4_d"], out["c4_x"] = Dt, Xt
    out["c4_batch"] = cu.KKZ_init(t(Xt), t(Dt), 7, batch=True).numpy()
    out["c4_single"] = np.stack([cu.KKZ_init(t(Xt[b]), t(Dt[b]), 7, batch=False).numpy() for b in range(3)])

    # ---- C5 / P0: selection from a STORED fp32 D (reference's own cdist output on this host)
    def run_from_d(X, K, metric, p, thr=1e-6, iters=100):
        Xt_ = t(X)
        D = cu.pairwise_distance(Xt_, Xt_, metric=metric, all_negative=True, self_nearest=True, p=p)
        nrm = torch.norm(Xt_, dim=-1)
        a, m = fk.batch_fast_kmedoids(Xt_, K, distance=metric, threshold=thr, iter_limit=iters, id_sort=True, norm_p=p)
        a_ns, m_ns = fk.batch_fast_kmedoids(Xt_, K, distance=metric, threshold=thr, iter_limit=iters, id_sort=False, norm_p=p)
        return D.numpy(), nrm.numpy(), a.numpy(), m.numpy(), a_ns.numpy(), m_ns.numpy()

    p0_cases = {
        "p0_small_l2": (dyadic(21, (4, 40, 24)) * 3, 10, "euclidean", 2.0),
        "p0_small_cos": (dyadic(22, (4, 40, 24)) * 3, 10, "cosine", 2.0),
        "p0_real_l2": (np.random.default_rng(23).standard_normal((2, 196, 768)).astype(np.float32), 49, "euclidean", 2.0),
        "p0_n392_l2": (np.random.default_rng(24).standard_normal((1, 392, 768)).astype(np.float32), 49, "euclidean", 2.0),
    }
    for tag, (X, K, metric, p) in p0_cases.items():
        D, nrm, a, m, a_ns, m_ns = run_from_d(X, K, metric, p)
        out[f"{tag}_d"], out[f"{tag}_norm"] = D, nrm
        out[f"{tag}_assign"], out[f"{tag}_medoids"] = a.astype(np.int16), m.astype(np.int16)
        out[f"{tag}_assign_nosort"], out[f"{tag}_medoids_nosort"] = a_ns.astype(np.int16), m_ns.astype(np.int16)
        out[f"{tag}_k"] = np.int64(K)

    # crafted tie case for the update step: integer D, symmetric, 2- and 3-member ties
    Dtie = -(np.random.default_rng(31).integers(1, 4, size=(2, 16, 16)).astype(np.float32)) - 1.0
    Dtie = np.minimum(Dtie, Dtie.transpose(0, 2, 1))
    i16 = np.arange(16)
    Dtie[:, i16, i16] = -6.0
    Xtie = lattice(32, (2, 16, 4)) + np.arange(16, dtype=np.float32)[None, :, None] * 0.0
    Xtie[:, :, 0] += np.arange(16, dtype=np.float32)[None, :] * 8.0     # distinct rows -> stop test == fixed point
    orig = fk.pairwise_distance
    fk.pairwise_distance = lambda *a_, **k_: t(Dtie).clone()
    try:
        a, m = fk.batch_fast_kmedoids(t(Xtie), 5, threshold=1e-6, iter_limit=50, id_sort=True)
    finally:
        fk.pairwise_distance = orig
    out["p0_tie_d"], out["p0_tie_x"] = Dtie, Xtie
    out["p0_tie_assign"], out["p0_tie_medoids"] = a.numpy().astype(np.int16), m.numpy().astype(np.int16)

    # ---- P1: from-X on integer lattices at the real shapes (seeds + indices only)
    #      name: (seed, P, N, W, K, split, iter_limit)
    p1 = {
        "p1_cfg2": (41, 48, 196, 768, 49, 16, 100),
        "p1_cfg3": (42, 32, 147, 768, 49, 16, 100),
        "p1_cfg4": (43, 8, 392, 768, 49, 16, 100),
        "p1_cfg5": (44, 4, 588, 768, 100, 4, 100),
        "p1_ragged": (45, 7, 60, 96, 9, 3, 100),       # P not a multiple of split_size
        "p1_k_eq_n": (46, 2, 12, 32, 12, 16, 100),     # K == N: every token is a medoid
        "p1_k1": (47, 3, 30, 32, 2, 16, 100),
    }
    for tag, (seed, P, N, W, K, split, iters) in p1.items():
        X = lattice(seed, (P, N, W))
        a, m = fk.batch_fast_kmedoids_with_split(t(X), K, distance="euclidean", threshold=1e-6, iter_limit=iters,
                                                 id_sort=True, norm_p=2.0, split_size=split, pre_norm=False)
        out[f"{tag}_cfg"] = np.array([seed, P, N, W, K, split, iters], dtype=np.int64)
        out[f"{tag}_assign"], out[f"{tag}_medoids"] = a.numpy().astype(np.int16), m.numpy().astype(np.int16)
        print(tag, "done", flush=True)

    # ---- P2: from-X with norm_p = 1 on dyadic inputs (every L1 distance exact in fp32)
    p2 = {
        "p2_cfg2": (51, 16, 196, 768, 49, 16, 100),
        "p2_cfg3": (52, 16, 147, 768, 49, 16, 100),
        "p2_small": (53, 5, 50, 64, 7, 2, 100),
    }
    for tag, (seed, P, N, W, K, split, iters) in p2.items():
        X = dyadic(seed, (P, N, W))
        a, m = fk.batch_fast_kmedoids_with_split(t(X), K, distance="euclidean", threshold=1e-6, iter_limit=iters,
                                                 id_sort=True, norm_p=1.0, split_size=split, pre_norm=False)
        out[f"{tag}_cfg"] = np.array([seed, P, N, W, K, split, iters], dtype=np.int64)
        out[f"{tag}_assign"], out[f"{tag}_medoids"] = a.numpy().astype(np.int16), m.numpy().astype(np.int16)
        print(tag, "done", flush=True)

    # ---- C1: TokenClusterInter.forward, integer payloads so the permutation is visible
    #      name: (seed, B, T, T_new, n, W, K, split)
    c1 = {
        "c1_12_3": (61, 3, 12, 3, 49, 64, 49, 16),
        "c1_12_4": (62, 2, 12, 4, 49, 64, 49, 16),
        "c1_12_6": (63, 2, 12, 6, 49, 64, 20, 16),
        "c1_64_8": (64, 1, 64, 8, 49, 32, 49, 16),
        "c1_12_12": (65, 2, 12, 12, 49, 64, 25, 16),   # frames kept, tokens 49 -> 25 (cfg-1 style)
        "c1_b16": (66, 2, 12, 4, 196, 32, 100, 4),     # ViT-B/16-shaped, split 4
    }
    for tag, (seed, B, T, T_new, n, W, K, split) in c1.items():
        x = lattice(seed, (1 + n, B * T, W))
        mod = cc.TokenClusterInter(algorithm="kmediods++", block_id=7, before_cluster_num=n, cluster_num=K,
                                   before_block_frames=T, after_block_frames=T_new, original_frame=T,
                                   distance="euclidean", threshold=1e-6, iter_limit=100, id_sort=True,
                                   aggregation=None, split_size=split, norm_p=2.0, transformer_width=W)
        y, res = mod(t(x))
        assert res is None
        out[f"{tag}_cfg"] = np.array([seed, B, T, T_new, n, W, K, split], dtype=np.int64)
        out[f"{tag}_out"] = y.contiguous().numpy()
        print(tag, "done", tuple(y.shape), flush=True)

    np.savez_compressed(os.path.join(GOLD, "cluster_golden.npz"), **out)
    sz = os.path.getsize(os.path.join(GOLD, "cluster_golden.npz"))
    print("wrote cluster_golden.npz", sz, "bytes,", len(out), "arrays")


def gen_cluster_dup():
    """Duplicate-token problems (exact ties between non-identical candidates whose row sums round): the
    reference's selection from a stored D.  Seeds + outputs only -> tests/golden/cluster_dup_golden.npz."""
    sys.path.insert(0, os.path.join(REF, "modules"))
    import cluster.fast_kmeans as fk
    from recipes import DUPLICATE_CASES, duplicate_token_problem
    t = torch.from_numpy
    out = {}
    for tag, (seed, P, nd, N, K, layout) in DUPLICATE_CASES.items():
        D, X = duplicate_token_problem(seed, P, nd, N, layout)
        orig = fk.pairwise_distance
        fk.pairwise_distance = lambda *a_, **k_: t(D).clone()
        try:
            a, m = fk.batch_fast_kmedoids(t(X), K, threshold=1e-6, iter_limit=100, id_sort=True)
            a_ns, m_ns = fk.batch_fast_kmedoids(t(X), K, threshold=1e-6, iter_limit=100, id_sort=False)
        finally:
            fk.pairwise_distance = orig
        out[f"{tag}_assign"], out[f"{tag}_medoids"] = a.numpy().astype(np.int16), m.numpy().astype(np.int16)
        out[f"{tag}_assign_nosort"], out[f"{tag}_medoids_nosort"] = a_ns.numpy().astype(np.int16), m_ns.numpy().astype(np.int16)
        print(tag, "done", flush=True)
    np.savez_compressed(os.path.join(GOLD, "cluster_dup_golden.npz"), **out)
    print("wrote cluster_dup_golden.npz", os.path.getsize(os.path.join(GOLD, "cluster_dup_golden.npz")), "bytes")


def gen_cluster_variants():
    """N2: the other TokenClusterInter branches (aggregation, cluster_embedding, adaptive_cls, pooling,
    sparse_sampling in eval mode) run by the reference module itself -> tests/golden/cluster_variants_golden.npz.
    Inputs are regenerated from seeds (recipes.variant_input); stored: outputs, and for the k-medoids cases
    the assignment / medoids the reference used."""
    sys.path.insert(0, os.path.join(REF, "modules"))
    import cluster.cluster as cc
    import cluster.fast_kmeans as fk
    from recipes import VARIANT_CASES, variant_input
    t = torch.from_numpy
    out = {}
    for tag, cfg in VARIANT_CASES.items():
        x, embed, mult = variant_input(cfg)
        mod = cc.TokenClusterInter(algorithm=cfg["algorithm"], block_id=7, before_cluster_num=cfg["n"],
                                   cluster_num=cfg["K"], before_block_frames=cfg["T"], after_block_frames=cfg["T_new"],
                                   original_frame=cfg["T"], distance="euclidean", threshold=1e-6, iter_limit=100,
                                   id_sort=True, aggregation=cfg["aggregation"], split_size=16, norm_p=2.0,
                                   cluster_embedding=bool(cfg.get("embed")), adaptive_cls=bool(cfg.get("adaptive")),
                                   transformer_width=cfg["W"])
        mod.eval()
        with torch.no_grad():
            if embed is not None:
                mod.cluster_embed.copy_(t(embed))
            if mult is not None:
                mod.cls_multiplier.copy_(t(mult).reshape(1, -1, 1, 1))
            captured = {}
            orig = cc.batch_fast_kmedoids_with_split

            def spy(*a, **k):
                r = orig(*a, **k)
                captured["assign"], captured["medoids"] = r[0].clone(), r[1].clone()
                return r
            cc.batch_fast_kmedoids_with_split = spy
            try:
                y, res = mod(t(x))
            finally:
                cc.batch_fast_kmedoids_with_split = orig
        assert res is None
        out[f"{tag}_out"] = y.contiguous().numpy()
        if captured:
            out[f"{tag}_assign"] = captured["assign"].numpy().astype(np.int16)
            out[f"{tag}_medoids"] = captured["medoids"].numpy().astype(np.int16)
        print(tag, "done", tuple(y.shape), flush=True)
    np.savez_compressed(os.path.join(GOLD, "cluster_variants_golden.npz"), **out)
    print("wrote cluster_variants_golden.npz", os.path.getsize(os.path.join(GOLD, "cluster_variants_golden.npz")), "bytes")


def gen_metrics_multi():
    """N1, multi-sentence protocol (main.py:466-480): the reference's tensor_text_to_video_metrics /
    tensor_video_to_text_sim + compute_metrics on a small ragged problem -> tests/golden/metrics_multi_golden.npz."""
    from gen_golden_clip import _import_reference
    rm = _import_reference()[3]
    rng = np.random.default_rng(12)
    lens = [3, 1, 5, 2, 4, 1, 2, 5, 3, 1, 2, 4]                      # sentences per video
    G = len(lens)
    sim = (rng.integers(-2000, 2000, size=(sum(lens), G)).astype(np.float32) / np.float32(64.0))   # distinct enough, a few ties
    cut = np.cumsum(lens)
    max_len = max(lens)
    blocks = []
    for s_, e_ in zip([0] + list(cut[:-1]), cut):
        blocks.append(np.concatenate((sim[s_:e_], np.full((max_len - e_ + s_, G), -np.inf, dtype=np.float32)), axis=0))
    sim3 = np.stack(blocks, axis=0)
    tv = rm.tensor_text_to_video_metrics(sim3.copy())
    v2t_sim = rm.tensor_video_to_text_sim(sim3.copy())
    vt = rm.compute_metrics(v2t_sim.numpy() if torch.is_tensor(v2t_sim) else v2t_sim)
    out = {"sim3": sim3, "v2t_sim": np.asarray(v2t_sim),
           "tv": np.array([tv["R1"], tv["R5"], tv["R10"], tv["MedianR"], tv["MeanR"], tv["Std_Rank"]], dtype=np.float64),
           "vt": np.array([vt["R1"], vt["R5"], vt["R10"], vt["MR"], vt["MeanR"]], dtype=np.float64)}
    np.savez_compressed(os.path.join(GOLD, "metrics_multi_golden.npz"), **out)
    print("wrote metrics_multi_golden.npz", tv, vt)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    os.makedirs(GOLD, exist_ok=True)
    if what in ("cluster", "all"):
        gen_cluster()
    if what in ("dup", "all"):
        gen_cluster_dup()
    if what in ("variants", "all"):
        gen_cluster_variants()
    if what in ("metrics", "all"):
        gen_metrics_multi()
    if what in ("clip", "all"):
        from gen_golden_clip import gen_clip
        gen_clip()
