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
    # beyond 640 tokens (ViT-B/16 with 4 frames per segment = 784; the supported maximum 1023)
    "perm_n784": (88, 1, 0, 784, 1, "permrows"),
    "perm_n1023": (89, 1, 0, 1023, 1, "permrows"),
    "dup_n1000": (90, 1, 250, 1000, 20, "tile"),
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


def fullmant(seed, shape, scale_bits=20):
    """Generic-looking fp32 values that are bit-reproducible: integers in (-2^23, 2^23) times 2^-scale_bits.
    Full 24-bit significands, so sums of them round and expose the order of summation."""
    r = np.random.default_rng(seed).integers(-(1 << 23) + 1, 1 << 23, size=shape)
    return (r.astype(np.float32) * np.float32(2.0 ** -scale_bits)).astype(np.float32)


# N2 variant fixtures: name -> dict(seed, kind of input, B, T, T_new, n, W, K, algorithm, aggregation, embed, adaptive)
VARIANT_CASES = {
    "mean_lattice": dict(seed=91, inp="lattice", B=2, T=12, T_new=4, n=49, W=64, K=20, algorithm="kmediods++", aggregation="mean"),
    "mean_generic": dict(seed=92, inp="fullmant", B=2, T=12, T_new=3, n=49, W=64, K=12, algorithm="kmediods++", aggregation="mean"),
    "mean_big_clusters": dict(seed=93, inp="fullmant", B=1, T=8, T_new=1, n=49, W=32, K=3, algorithm="kmediods++", aggregation="mean"),
    "embed_adaptive": dict(seed=94, inp="lattice", B=2, T=12, T_new=4, n=49, W=64, K=20, algorithm="kmediods++", aggregation=None,
                           embed=True, adaptive=True),
    "mean_embed_adaptive": dict(seed=95, inp="lattice", B=2, T=12, T_new=6, n=49, W=32, K=10, algorithm="kmediods++",
                                aggregation="mean", embed=True, adaptive=True),
    "pooling_12_3": dict(seed=96, inp="fullmant", B=2, T=12, T_new=3, n=49, W=64, K=49, algorithm="pooling", aggregation=None),
    "pooling_64_2": dict(seed=97, inp="fullmant", B=1, T=64, T_new=2, n=16, W=32, K=16, algorithm="pooling", aggregation=None),
    "sparse_12_3": dict(seed=98, inp="fullmant", B=2, T=12, T_new=3, n=49, W=64, K=20, algorithm="sparse_sampling", aggregation=None),
    "cls_mean_fd32": dict(seed=99, inp="lattice_cls_generic", B=1, T=64, T_new=2, n=16, W=32, K=9, algorithm="kmediods++",
                          aggregation=None),
}


# N4 gradient fixtures (TokenClusterInter under autograd): the variant cases whose selection is an exact target + the
# shipped default branch.  The upstream gradient G [1+K', B*T_new, W] is fullmant(seed + 5000).
GRAD_CASES = dict(VARIANT_CASES)
for _k in ("mean_generic", "pooling_12_3"):
    GRAD_CASES.pop(_k)
GRAD_CASES["default_12_3"] = dict(seed=89, inp="lattice", B=2, T=12, T_new=3, n=49, W=64, K=20, algorithm="kmediods++",
                                  aggregation=None)


def grad_output(cfg):
    Kp = cfg["n"] if cfg["algorithm"] == "pooling" else cfg["K"]
    return fullmant(cfg["seed"] + 5000, (1 + Kp, cfg["B"] * cfg["T_new"], cfg["W"]))


# N4 spectral fixtures: tokens around K well-separated centres, the groups being contiguous runs of the token index inside
# every segment (so that the rank of any medoid of a group - the cluster label - does not depend on which member is picked).
SPECTRAL_CASES = {
    "planted_heat": dict(seed=61, B=2, T=4, T_new=2, n=16, W=32, K=4, sigma=2.0, graph="HeatKernel", knn_k=0, sep=6.0),
    "planted_knn_spg": dict(seed=62, B=2, T=6, T_new=2, n=16, W=32, K=6, sigma=2.0, graph="KNN", knn_k=0, spg=1, sep=6.0),
    "planted_k5_odd": dict(seed=63, B=3, T=3, T_new=1, n=25, W=32, K=5, sigma=2.0, graph="HeatKernel", knn_k=0, sep=5.0),
    # the real layouts (round 4): ViT-B/32 12 -> 3 frames (N = 196, K = 49: the LDS eigensolver at its largest shape),
    # 64 -> 8 frames (N = 392: the global-memory eigensolver; heat kernel, and KNN graph + spatial-temporal mask), ViT-B/16 12 -> 4 frames with
    # K = 100 (N = 588, groups of 5 and 6 tokens), full width
    "planted_196_k49": dict(seed=64, B=2, T=4, T_new=1, n=49, W=768, K=49, sigma=2.0, graph="HeatKernel", knn_k=0, sep=6.0),
    "planted_392_k49": dict(seed=65, B=1, T=16, T_new=2, n=49, W=768, K=49, sigma=2.0, graph="HeatKernel", knn_k=0, sep=6.0),
    # (with the spatial-temporal mask a group has to be connected under it: K = 56 makes the groups the rows of the 7 x 7 grid)
    "planted_392_k56_knn": dict(seed=67, B=1, T=8, T_new=1, n=49, W=768, K=56, sigma=2.0, graph="KNN", knn_k=0, spg=1, sep=6.0),
    "planted_588_k100": dict(seed=66, B=1, T=3, T_new=1, n=196, W=768, K=100, sigma=2.0, graph="HeatKernel", knn_k=0, sep=6.0),
    # ViT-B/16 60 -> 15 frames with K = 160 (scripts/activitynet.sh:104-122): N = 784, groups of 4 and 5 tokens - more vectors
    # than one back-transformation pass of the direct eigensolver holds
    "planted_784_k160": dict(seed=68, B=1, T=4, T_new=1, n=196, W=768, K=160, sigma=2.0, graph="HeatKernel", knn_k=0, sep=6.0),
}


def planted_group(j, N, K):
    """Planted group of token j of a segment: contiguous runs of floor / ceil(N / K) tokens (N / K when it divides)."""
    return (j * K) // N


def planted_tokens(cfg):
    """x [1+n, B*T, W] fp32: patch token j = f*n + i of a segment lies at centre[j // (N/K)] (|centre_a - centre_b| >= sep)
    plus noise of magnitude 0.05; CLS tokens are generic."""
    rng = np.random.default_rng(cfg["seed"])
    B, T, Tn, n, W, K = (cfg[k] for k in ("B", "T", "T_new", "n", "W", "K"))
    fd = T // Tn
    N = fd * n
    x = np.zeros((1 + n, B * T, W), dtype=np.float32)
    x[0] = fullmant(cfg["seed"] + 1, (B * T, W))
    for b in range(B):
        for s_ in range(Tn):
            centres = np.zeros((K, W), dtype=np.float32)
            for k in range(K):
                centres[k, (3 * k) % W] = cfg["sep"]                  # orthogonal axes: pairwise distance sep * sqrt(2)
                centres[k, (3 * k + 1) % W] = 0.5 * k
            for f in range(fd):
                for i in range(n):
                    j = f * n + i
                    noise = (rng.integers(-64, 65, size=W).astype(np.float32) * np.float32(0.05 / 64.0))
                    x[1 + i, b * T + s_ * fd + f] = centres[planted_group(j, N, K)] + noise
    return x


# pairwise_distance(data1, data2) with two different sets: batched / 2-d, N1 != N2, every metric and flag
CROSS_DIST_CASES = {
    "l2_batched": dict(seed=71, shape1=(2, 40, 32), shape2=(2, 30, 32), metric="euclidean", p=2.0, self_nearest=True, all_negative=True),
    "l2_plain": dict(seed=72, shape1=(3, 17, 48), shape2=(3, 17, 48), metric="euclidean", p=2.0, self_nearest=False, all_negative=False),
    "l1_2d": dict(seed=73, shape1=(33, 20), shape2=(21, 20), metric="euclidean", p=1.0, self_nearest=True, all_negative=False),
    "l3": dict(seed=74, shape1=(2, 18, 16), shape2=(2, 35, 16), metric="euclidean", p=3.0, self_nearest=False, all_negative=True),
    "cos_batched": dict(seed=75, shape1=(2, 50, 64), shape2=(2, 28, 64), metric="cosine", p=2.0, self_nearest=True, all_negative=True),
    "cos_2d": dict(seed=76, shape1=(9, 12), shape2=(31, 12), metric="cosine", p=2.0, self_nearest=False, all_negative=False),
}


def cross_dist_inputs(cfg):
    return fullmant(cfg["seed"], cfg["shape1"], 21), fullmant(cfg["seed"] + 500, cfg["shape2"], 21)


def variant_input(cfg):
    """x [1+n, B*T, W] fp32 for a VARIANT_CASES entry (+ cluster_embed [K,W], cls_multiplier [T] when asked)."""
    L, BT, W = 1 + cfg["n"], cfg["B"] * cfg["T"], cfg["W"]
    if cfg["inp"] == "lattice":
        x = lattice(cfg["seed"], (L, BT, W))
    elif cfg["inp"] == "fullmant":
        x = fullmant(cfg["seed"], (L, BT, W))
    else:                                   # lattice patches (robust clustering), generic CLS row
        x = lattice(cfg["seed"], (L, BT, W))
        x[0] = fullmant(cfg["seed"] + 1000, (BT, W))
    embed = fullmant(cfg["seed"] + 2000, (cfg["K"], W), 24) if cfg.get("embed") else None
    mult = (fullmant(cfg["seed"] + 3000, (cfg["T"],), 24) + np.float32(1.0)) if cfg.get("adaptive") else None
    return x, embed, mult


def _four_squares(r):
    """Non-negative integers (a, b, c, d) with a^2 + b^2 + c^2 + d^2 = r (Lagrange), first in lexicographic order."""
    a = 0
    while a * a <= r:
        b = 0
        while a * a + b * b <= r:
            c = 0
            while a * a + b * b + c * c <= r:
                d2 = r - a * a - b * b - c * c
                d = int(round(d2 ** 0.5))
                if d * d == d2:
                    return a, b, c, d
                c += 1
            b += 1
        a += 1
    raise ValueError(r)


def norm32_tokens(seed, shape):
    """Integer-valued fp32 tokens whose L2 norm is EXACTLY 32 (sum of squares 1024): W - 4 entries in [-3, 3], the last
    four complete the sum of squares.  32 + 1e-6 rounds to 32 in fp32, so the reference's pre-normalisation
    X / (|X| + 1e-6) (fast_kmeans.py:21-22) is an exact division by 2^5: a bit-exact target for pre_norm=True."""
    rng = np.random.default_rng(seed)
    *lead, W = shape
    x = rng.integers(-3, 4, size=tuple(lead) + (W,)).astype(np.int64)
    flat = x.reshape(-1, W)
    for row in flat:
        row[W - 4:] = 0
        rem = 1024 - int((row * row).sum())
        assert rem >= 0, "W too large for a norm-32 token"
        sq = _four_squares(rem)
        signs = rng.integers(0, 2, size=4) * 2 - 1
        row[W - 4:] = np.array(sq) * signs
    return flat.reshape(shape).astype(np.float32)


# name: (seed, P, N, W, K, split)   -- pre_norm=True fixtures (r2_golden.npz)
PRENORM_CASES = {
    "pn_small": (111, 5, 60, 96, 9, 2),
    "pn_cfg2": (112, 16, 196, 128, 49, 16),
}


# ---------------------------------------------------------------------------------------------- round 3: S3 fixtures
def s3_case(E, T, T_new, seed=301, Nt=37, Nv=21, bt=8, bv=8):
    """Cached-feature lists as main.eval_epoch hands them to main._run_on_single_gpu (main.py:502-534): ragged text /
    video batches (the last ones short), video masks with the ORIGINAL frame count T and zeros in them (one clip fully
    masked) so that get_similarity_logits has to derive the per-segment masks itself.
    -> (sequence_output list [b,1,E], visual_output list [b,T_new,E], batch_list_t [(input_mask, segment_ids)],
        batch_list_v [(video_mask,)]) as torch tensors."""
    import torch
    text = fullmant(seed, (Nt, 1, E), 22)
    vis = fullmant(seed + 1, (Nv, T_new, E), 22)
    rng = np.random.default_rng(seed + 2)
    vmask = np.ones((Nv, 1, T), dtype=np.int64)
    for v in range(Nv):
        if v % 4 == 1:
            vmask[v, 0, rng.integers(1, T):] = 0          # trailing padding frames
    vmask[5, 0, :] = 0                                    # fully masked clip: NaN column in the reference
    amask = np.ones((Nt, 1, 8), dtype=np.int64)
    seq_list, vis_list, list_t, list_v = [], [], [], []
    for s in range(0, Nt, bt):
        seq_list.append(torch.from_numpy(text[s:s + bt]))
        m = torch.from_numpy(amask[s:s + bt])
        list_t.append((m, torch.zeros_like(m)))
    for s in range(0, Nv, bv):
        vis_list.append(torch.from_numpy(vis[s:s + bv]))
        list_v.append((torch.from_numpy(vmask[s:s + bv]),))
    return seq_list, vis_list, list_t, list_v


# main.eval_epoch over a list-backed loader (ev_* of r3_golden.npz): per-video sentence counts, batch size
EVAL_CASES = {
    # (seeds picked by oracle/gen_golden_r3.py's robustness check: rank metrics stable under 5e-4 * exp(logit_scale))
    "single": dict(seed=319, sentences=[1] * 6, batch=4),                  # one caption per clip, last batch short
    "multi": dict(seed=326, sentences=[3, 1, 4, 2, 3], batch=4),           # multi_sentence_per_video protocol
}


def eval_case_batches(case, cfg):
    """Batches (input_ids, input_mask, segment_ids, video, video_mask) of an EVAL_CASES entry for the small model whose
    geometry is cfg = r2_golden's s1_cfg, + the dataset attributes main.eval_epoch reads (main.py:391-399).  One item
    per sentence; under the multi-sentence protocol an item carries the video of its clip, and the clip's video is taken
    from the item of its last sentence (cut_off_points).  Videos are dyadic (exactly representable, seed-reproducible)."""
    import torch
    RES, CTX, VOCAB, T = int(cfg[1]), int(cfg[5]), int(cfg[6]), int(cfg[11])
    rng = np.random.default_rng(case["seed"])
    sentences = case["sentences"]
    nvid = len(sentences)
    videos = dyadic(case["seed"] + 1, (nvid, 1, T, 3, RES, RES)) * np.float32(1.5)
    vmasks = np.ones((nvid, 1, T), dtype=np.int64)
    vmasks[1, 0, T - 1:] = 0
    items = []
    for v, ns in enumerate(sentences):
        for _ in range(ns):
            ln = int(rng.integers(4, CTX + 1))
            ids = np.zeros((1, CTX), dtype=np.int64)
            ids[0, 0], ids[0, ln - 1] = VOCAB - 2, VOCAB - 1
            ids[0, 1:ln - 1] = rng.integers(1, VOCAB - 2, size=ln - 2)
            items.append((ids, (ids > 0).astype(np.int64), np.zeros_like(ids), videos[v], vmasks[v]))
    batches = []
    for s in range(0, len(items), case["batch"]):
        chunk = items[s:s + case["batch"]]
        batches.append(tuple(torch.from_numpy(np.stack([it[k] for it in chunk])) for k in range(5)))
    attrs = {}
    if any(ns != 1 for ns in sentences):
        attrs = dict(multi_sentence_per_video=True, cut_off_points=list(np.cumsum(sentences)),
                     sentence_num=len(items), video_num=nvid)
    return batches, attrs


def loss_grad_case(tag, n, T_new, E):
    """Features for the lg_* fixtures (training loss + gradients): sequence_output [n,1,E], visual_output [n,T_new,E] with
    generic full-mantissa values, video_mask [n,T_new] with zeros (never a fully masked clip: the loss would be NaN)."""
    seed = 700 + sum(ord(ch) for ch in tag)
    seq = fullmant(seed, (n, 1, E), 21)
    vis = fullmant(seed + 1, (n, T_new, E), 21)
    rng = np.random.default_rng(seed + 2)
    vmask = np.ones((n, T_new), dtype=np.int64)
    for v in range(n):
        if v % 3 == 1 and T_new > 1:
            vmask[v, rng.integers(1, T_new)] = 0
    return seq, vis, vmask


def conv3d_patch_weight(seed, shape):
    """fp16-representable Conv3d weights for the linear_patch='3d' fixture: integers in [-2048, 2048] / 2^15."""
    r = np.random.default_rng(seed).integers(-2048, 2049, size=shape)
    return (r.astype(np.float32) * np.float32(2.0 ** -15)).astype(np.float32)


PATCH3D_SEED = 901        # linear_patch='3d' fixture (p3d_* of r3_golden.npz)
MINOR_SEED = 911          # mean_residual / training-mode sparse_sampling fixtures (mr_*, ss_train_*)


# N4: backward of one ResidualAttentionBlock (modules/clip.py:196-253).  x [L, N, W] LND, dz the gradient fed into the output
BLOCK_GRAD_CASES = {
    "bg_visual": dict(seed=91, L=50, N=6, W=128, heads=2, causal=False),
    "bg_text": dict(seed=92, L=12, N=5, W=128, heads=2, causal=True),
    # sequences longer than 64 tokens (the attention backward's two-launch form): ViT-B/16's 197 tokens per frame, CLIP's
    # native 77-token context with the causal mask
    "bg_visual_b16": dict(seed=93, L=197, N=3, W=128, heads=2, causal=False),
    "bg_text77": dict(seed=94, L=77, N=4, W=128, heads=2, causal=True),
}


def block_grad_inputs(cfg):
    """-> (x [L,N,W], dz [L,N,W], state dict of the block under the reference's parameter names), fp32; weights rounded
    through fp16 (as convert_weights yields) so that the fp16 operands of the HIP path carry exactly these values."""
    rng = np.random.default_rng(cfg["seed"])
    L, N, W = cfg["L"], cfg["N"], cfg["W"]
    f16 = lambda a: a.astype(np.float16).astype(np.float32)
    x = rng.standard_normal((L, N, W)).astype(np.float32) * 1.5 + 0.2
    dz = rng.standard_normal((L, N, W)).astype(np.float32) * 0.01
    sd = {
        "attn.in_proj_weight": f16(rng.standard_normal((3 * W, W)) * W ** -0.5),
        "attn.in_proj_bias": f16(rng.standard_normal(3 * W) * 0.1),
        "attn.out_proj.weight": f16(rng.standard_normal((W, W)) * W ** -0.5),
        "attn.out_proj.bias": f16(rng.standard_normal(W) * 0.1),
        "ln_1.weight": f16(1.0 + 0.2 * rng.standard_normal(W)), "ln_1.bias": f16(0.1 * rng.standard_normal(W)),
        "mlp.c_fc.weight": f16(rng.standard_normal((4 * W, W)) * W ** -0.5),
        "mlp.c_fc.bias": f16(rng.standard_normal(4 * W) * 0.1),
        "mlp.c_proj.weight": f16(rng.standard_normal((W, 4 * W)) * (4 * W) ** -0.5),
        "mlp.c_proj.bias": f16(rng.standard_normal(W) * 0.1),
        "ln_2.weight": f16(1.0 + 0.2 * rng.standard_normal(W)), "ln_2.bias": f16(0.1 * rng.standard_normal(W)),
    }
    sd = {k: v.astype(np.float32) for k, v in sd.items()}
    return x, dz, sd


# Round 6: k-medoids with a loose `threshold` (the literal chunk-mean stop test, fast_kmeans.py:85-88).
# name: (seed, P, N, W, K, split, iter_limit, distance, pre_norm, id_sort, (ca, sa, wa, cb, sb, wb), recipe) - inputs
# recipe(seed, (P, N, W)) with recipe "dyadic" or "norm32" (pre_norm: X / (|X| + 1e-6) is then an exact division by 32);
# threshold = wa * shift[chunk ca][step sa] + wb * shift[chunk cb][step sb] of the reference's own center_shift sequences
# (oracle/gen_golden_r6.py records them), i.e. a value that falls between two steps the reference really takes
LOOSE_THRESHOLD_CASES = {
    "lt_two_chunks": (161, 6, 196, 64, 49, 4, 60, "euclidean", False, True, (0, 1, 0.5, 1, 1, 0.5), "dyadic"),
    "lt_unsorted": (161, 6, 196, 64, 49, 4, 60, "euclidean", False, False, (0, 1, 0.5, 1, 1, 0.5), "dyadic"),
    "lt_wide": (162, 3, 100, 768, 10, 2, 60, "euclidean", False, True, (0, 0, 0.5, 1, 0, 0.5), "dyadic"),
    "lt_prenorm": (170, 4, 98, 96, 25, 2, 60, "euclidean", True, True, (0, 1, 0.5, 1, 1, 0.5), "norm32"),
    "lt_tiny": (164, 4, 7, 8, 3, 4, 60, "euclidean", False, True, (0, 0, 1.5, 0, 0, 0.0), "dyadic"),
    "lt_iter_limit": (165, 4, 196, 64, 49, 4, 2, "euclidean", False, True, (0, 2, 0.1, 0, 2, 0.0), "dyadic"),
}


def loose_threshold_inputs(tag):
    seed, P, N, W = LOOSE_THRESHOLD_CASES[tag][:4]
    return (norm32_tokens if LOOSE_THRESHOLD_CASES[tag][11] == "norm32" else dyadic)(seed, (P, N, W))


# Round 6: k-medoids above 4,095 tokens per problem.  name: (seed, P, N, W, K, split, iter_limit) - lattice(seed, (P, N, W))
P1_WIDE_CASES = {"p1w_4500": (181, 2, 4500, 16, 6, 1, 100), "p1w_8191": (182, 1, 8191, 8, 4, 16, 100)}
