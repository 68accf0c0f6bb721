"""CPU oracle for the token-cluster hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``centerclip_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker / the timed CPU baseline.

It is a plain-PyTorch (CPU, fp32) restatement of the reference algorithm, with
two independent formulations:

* ``literal_*``  - the same ATen op sequence the reference issues (cdist,
  chunk-max shift, KKZ re-gather, [B,K,N,N] masked sum, chunk-mean stop test),
  so on one host it reproduces the reference bit for bit and costs what the
  reference costs.  This is what ``bench.py`` times as ``cpu_baseline``
  (kind "port").
* ``select_streamlined`` - a per-problem numpy restatement built only from the
  exact equivalences listed in SURVEY.md §8(a) (running-min KKZ, cluster row
  sums in the association of ATen's CPU sum -- ``aten_row_sums`` --, fixed-point
  stop).  It is the executable spec of what the HIP selection kernel does.

Pinning: the reference ships no golden vectors for this path, so
the oracle is pinned by fixtures generated from the imported reference in the
dev container (``oracle/gen_golden.py`` -> ``tests/golden/cluster_*.npz``) and
checked in ``tests/test_oracle_cluster.py``.

Reference citations are relative to /root/reference.
"""
import numpy as np
import torch

METRICS = ("euclidean", "cosine")


# --------------------------------------------------------------------------- C3
def literal_pairwise_distance(a, b, metric="euclidean", self_nearest=True,
                              all_negative=False, p=2.0):
    """modules/cluster/cluster_utils.py:8-43.

    euclidean -> torch.cdist(p) (:22); cosine -> 1 - a_hat @ b_hat^T with
    x_hat = x / (|x| + 1e-6) (:24-30); all_negative subtracts the max of the
    WHOLE tensor handed in (one split chunk) and 1 (:35-36); self_nearest
    subtracts another 1 on the diagonal (:38-41).
    """
    if metric == "euclidean":
        d = torch.cdist(a, b, p=p)
    elif metric == "cosine":
        ah = a / (a.norm(dim=-1, keepdim=True) + 1e-6)
        bh = b / (b.norm(dim=-1, keepdim=True) + 1e-6)
        if a.ndim == 3:
            d = 1.0 - torch.bmm(ah, bh.transpose(-2, -1))
        else:
            d = 1.0 - torch.matmul(ah, bh.transpose(-2, -1))
    else:
        raise NotImplementedError("{} metric is not implemented".format(metric))
    if all_negative:
        d = d - torch.max(d) - 1.0
    if self_nearest:
        idx = torch.arange(d.shape[-1], dtype=torch.long)
        d[..., idx, idx] -= 1.0
    return d


# --------------------------------------------------------------------------- C4
def literal_kkz(l2_norm, D, K):
    """Batched KKZ init, modules/cluster/cluster_utils.py:93,106-118.

    l2_norm [B,N] = torch.norm(X, dim=-1) (:93).  First medoid = first argmax
    of the norm; medoid i = argmax_n min_{j<i} D[b, m_j, n] (rows of D).
    """
    B, N = l2_norm.shape
    rows = torch.arange(B, dtype=torch.long).unsqueeze(1)
    med = torch.arange(K, dtype=torch.long).unsqueeze(0).repeat(B, 1)
    med[:, 0] = torch.max(l2_norm, dim=1)[1]
    for i in range(1, K):
        nearest = torch.min(D[rows, med[:, :i], :], dim=1)[0]        # [B,N]
        med[:, i] = torch.max(nearest, dim=1)[1]
    return med


# --------------------------------------------------------------------------- C5
def literal_select(D, l2_norm, K, X=None, threshold=1e-5, iter_limit=60, id_sort=True):
    """k-medoids selection from a finished distance tensor D [B,N,N].

    modules/cluster/fast_kmeans.py:65-97.  ``X`` is only needed for the
    reference's stop test (chunk mean of sum_k |X[m_k]-X[m_k_prev]|_2 < thr,
    :85-88); with X=None the equivalent fixed-point test "no medoid of the
    chunk changed" is used (SURVEY §8a equivalence 4).
    Returns (assign [B,N] i64, medoids [B,K] i64, iterations executed).
    """
    B, N, _ = D.shape
    big = D.unsqueeze(1).repeat(1, K, 1, 1)                           # :65
    med = literal_kkz(l2_norm, D, K)                                  # :67
    rows = torch.arange(B, dtype=torch.long).unsqueeze(1)
    kid = torch.arange(K, dtype=torch.long).reshape(1, K, 1)
    steps = 0
    assign = None
    for _ in range(iter_limit):
        steps += 1
        prev = med
        assign = torch.min(D[rows, med, :], dim=1)[1]                 # :75-76
        member = assign.unsqueeze(1) == kid                           # [B,K,N]
        masked = big * member.unsqueeze(-1) * member.unsqueeze(-2)    # :81
        med = torch.argmin(torch.sum(masked, dim=-1), dim=-1)         # :82
        if X is not None:
            shift = torch.sum((X[rows, med, :] - X[rows, prev, :]) ** 2, dim=-1).sqrt().sum(dim=-1).mean()
            if shift < threshold:                                     # :85-88
                break
        elif torch.equal(med, prev):
            break
    if id_sort:
        med = torch.sort(med, dim=1)[0]                               # :90-94
        assign = torch.min(D[rows, med, :], dim=1)[1]
    return assign, med, steps


def literal_batch_kmedoids(X, K, distance="euclidean", threshold=1e-5, iter_limit=60,
                           id_sort=True, norm_p=2.0, return_steps=False):
    """modules/cluster/fast_kmeans.py:45-97 (one split chunk)."""
    assert distance in METRICS and X.ndim == 3
    D = literal_pairwise_distance(X, X, metric=distance, all_negative=True,
                                  self_nearest=True, p=norm_p)
    assign, med, steps = literal_select(D, torch.norm(X, dim=-1), K, X=X, threshold=threshold,
                                        iter_limit=iter_limit, id_sort=id_sort)
    return (assign, med, steps) if return_steps else (assign, med)


# --------------------------------------------------------------------------- C2
def literal_batch_kmedoids_with_split(X, K, distance="euclidean", threshold=1e-5, iter_limit=60,
                                      id_sort=True, norm_p=2.0, split_size=4, pre_norm=False):
    """modules/cluster/fast_kmeans.py:14-40: optional L2 pre-norm, then the
    batch is cut into chunks of ``split_size`` problems that are solved
    independently (the chunk is the scope of the max in C3 and of the stop test)."""
    X = X.float()
    if pre_norm:
        X = X / (X.norm(dim=-1, keepdim=True) + 1e-6)
    outs = [literal_batch_kmedoids(c, K, distance, threshold, iter_limit, id_sort, norm_p)
            for c in torch.split(X, split_size, dim=0)]
    return torch.cat([o[0] for o in outs], 0), torch.cat([o[1] for o in outs], 0)


# --------------------------------------------------------------------------- C1
def regroup_segments(x_lnd, T, T_new):
    """Layout contract of TokenClusterInter.forward (modules/cluster/cluster.py:239-250).

    x_lnd [1+n, B*T, W] (column = b*T + t).  Returns
      tokens [T_new*B, fd*n, W]  problem p = s*B + b, token j = f*n + i
      cls    [B, T, W]
    """
    L, BT, W = x_lnd.shape
    n, B, fd = L - 1, BT // T, T // T_new
    x = x_lnd.permute(1, 0, 2)                                        # [B*T, L, W]
    cls = x[:, 0, :].reshape(B, T, W)
    patches = x[:, 1:, :].reshape(B, T_new, fd, n, W)                 # t = s*fd + f
    tokens = patches.permute(1, 0, 2, 3, 4).reshape(T_new * B, fd * n, W)
    return tokens.contiguous(), cls


def literal_token_cluster(x_lnd, T, T_new, K, distance="euclidean", threshold=1e-6, iter_limit=100,
                          norm_p=2.0, split_size=16, pre_norm=False, return_ids=False):
    """kmediods++ / aggregation=None branch of TokenClusterInter.forward
    (modules/cluster/cluster.py:206-216,239-260,287-289,303-310,350-352).
    Output [1+K, B*T_new, W] with column b*T_new + s; CLS of a segment is the
    mean of its fd frame CLS tokens (:307-308)."""
    L, BT, W = x_lnd.shape
    B, fd = BT // T, T // T_new
    tokens, cls = regroup_segments(x_lnd, T, T_new)
    assign, med = literal_batch_kmedoids_with_split(tokens, K, distance, threshold, iter_limit,
                                                    True, norm_p, split_size, pre_norm)
    P = tokens.shape[0]
    picked = tokens[torch.arange(P).unsqueeze(-1), med]               # [T_new*B, K, W]  (:289)
    picked = picked.reshape(T_new, B, K, W).permute(1, 0, 2, 3).reshape(B * T_new, K, W)   # (:303)
    seg_cls = torch.stack([c.mean(dim=1) for c in torch.split(cls, fd, dim=1)], dim=1)      # (:307)
    out = torch.cat([seg_cls.reshape(B * T_new, 1, W), picked], dim=1).permute(1, 0, 2).contiguous()
    return (out, med, assign) if return_ids else out


# --------------------------------------------------------------------------- N2 variants
def literal_token_cluster_variant(x_lnd, T, T_new, K, algorithm="kmediods++", aggregation=None, cluster_embed=None,
                                  cls_multiplier=None, distance="euclidean", threshold=1e-6, iter_limit=100,
                                  norm_p=2.0, split_size=16, pre_norm=False, assign=None, medoids=None):
    """The other branches of TokenClusterInter.forward (modules/cluster/cluster.py), same ATen op
    sequence as the reference:
      kmediods++ + aggregation (:291-301)  token k = sum(res_tmp * mask_k, dim=1) / sum(mask_k)
      cluster_embed (:304-305), cls_multiplier / adaptive_cls (:244-245)
      pooling (:319-324), sparse_sampling in eval mode (:326-343 with cluster_utils.py:136-170)
    ``assign`` / ``medoids`` given: skip the k-medoids and use them (to compare the aggregation
    "given identical assignment").  Returns [1+K', B*T_new, W]."""
    L, BT, W = x_lnd.shape
    n, B, fd = L - 1, BT // T, T // T_new
    x = x_lnd.permute(1, 0, 2)
    if algorithm == "pooling":
        res_x = x.reshape(B, T, L, W)
        frame_split = [it.mean(dim=1) for it in torch.split(res_x, fd, dim=1)]
        return torch.stack(frame_split, dim=1).contiguous().reshape(B * T_new, L, W).permute(1, 0, 2).contiguous()
    all_cls = x[:, 0, :].reshape(B, T, 1, W)
    if cls_multiplier is not None:
        all_cls = all_cls * cls_multiplier.reshape(1, T, 1, 1)
    seg_cls = torch.stack([it.mean(dim=1) for it in torch.split(all_cls, fd, dim=1)], dim=1).reshape(B * T_new, 1, W)
    res_x = x[:, 1:, :].reshape(B, T, n, W)
    if algorithm == "sparse_sampling":
        res_all = []
        for it in torch.split(res_x, fd, dim=1):
            it_tmp = it.reshape(B, -1, W)
            total = it_tmp.shape[1]
            if total > K:
                tick = total / float(K)
                ind = np.array([int(tick / 2.0 + tick * i) for i in range(K)])
            else:
                ind = np.clip(np.arange(0, K), 0, total)
            res_all.append(it_tmp[:, torch.from_numpy(ind).long(), :])
        x_tmp = torch.stack(res_all, dim=1).contiguous().reshape(B * T_new, K, W)
        return torch.cat([seg_cls, x_tmp], dim=1).permute(1, 0, 2).contiguous()
    res_tmp = torch.cat(torch.split(res_x, fd, dim=1), dim=0).contiguous().reshape(B * T_new, -1, W)
    if assign is None or medoids is None:
        assign, medoids = literal_batch_kmedoids_with_split(res_tmp, K, distance, threshold, iter_limit, True,
                                                            norm_p, split_size, pre_norm)
    if aggregation in [None, "None"]:
        x_tmp = res_tmp[torch.arange(res_tmp.shape[0]).unsqueeze(-1), medoids, ...]
    else:
        parts = []
        for i in range(K):
            mask = (assign == i).unsqueeze(-1)
            parts.append(torch.sum(res_tmp * mask, dim=1, keepdim=True) / torch.sum(mask.float(), dim=1, keepdim=True))
        x_tmp = torch.cat(parts, dim=1)
    x_tmp = torch.stack(torch.split(x_tmp, B, dim=0), dim=1).reshape(B * T_new, K, W)
    if cluster_embed is not None:
        x_tmp = x_tmp + cluster_embed
    return torch.cat([seg_cls, x_tmp], dim=1).permute(1, 0, 2).contiguous()


def _cascade_rows(X, ilp):
    """multi_row_sum / row_sum of SumKernel.cpp along axis 0 of X [n, ...]: ``ilp`` interleaved
    accumulators over the rows (1 = plain sequential), each folded into the next level after every
    16 of its additions (256, 4096 for the higher levels); levels, then left-over rows, then the
    interleaved accumulators are added in order."""
    f = np.float32
    n = X.shape[0]
    size = n // ilp
    levels = 4
    lp = max(4, (int(np.ceil(np.log2(size))) if size > 1 else 0) // levels)
    step = 1 << lp
    mask = step - 1
    shp = (ilp,) + X.shape[1:]
    acc = [np.zeros(shp, f) for _ in range(levels)]
    i = 0
    while i + step <= size:
        for _ in range(step):
            acc[0] = acc[0] + X[i * ilp:(i + 1) * ilp]
            i += 1
        for j in range(1, levels):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = np.zeros(shp, f)
            if (i & (mask << (j * lp))) != 0:
                break
    while i < size:
        acc[0] = acc[0] + X[i * ilp:(i + 1) * ilp]
        i += 1
    for j in range(1, levels):
        acc[0] = acc[0] + acc[j]
    out = acc[0][0]
    for t in range(size * ilp, n):
        out = out + X[t]
    for k in range(1, ilp):
        out = out + acc[0][k]
    return out


def aten_outer_sums(M):
    """Column sums of M [n, C] (fp32) in the association ATen's CPU ``sum`` / ``mean`` uses when the
    reduced dimension is not the innermost one (SumKernel.cpp ``vectorized_outer_sum``; third-party
    dependency of the reference, restated from its published source): the columns are taken in
    groups of 32 (4 vectors of 8; threads split the columns on 32-column boundaries, so this does not
    depend on the thread count) and summed row after row with the cascade of ``_cascade_rows``
    (ilp = 1); the columns left after the last full group of 32 go through ``row_sum``, which
    additionally interleaves 4 accumulators over the rows (ilp = 4).  Every transformer width of
    the model family is a multiple of 32, so the hot path only ever sees the first form.
    This is the arithmetic of the segment means (cluster.py:307-308,319-324) and of the cluster
    means (:296-298); checked bit-for-bit against torch in tests/test_oracle_cluster.py."""
    M = np.ascontiguousarray(M, dtype=np.float32)
    C = M.shape[1]
    full = (C // 32) * 32
    out = np.empty(C, np.float32)
    if full:
        out[:full] = _cascade_rows(M[:, :full], 1)
    if full < C:
        out[full:] = _cascade_rows(M[:, full:], 4)
    return out


# ------------------------------------------------------- streamlined (kernel spec)
def aten_row_sums(M):
    """Row sums of M [R, n] (fp32) in the association ATen's CPU ``sum`` uses for a
    contiguous inner reduction -- the arithmetic behind ``torch.sum(sub_matrix,
    dim=-1)`` at modules/cluster/fast_kmeans.py:82.

    The algorithm is a third-party dependency of the reference (PyTorch,
    aten/src/ATen/native/cpu/SumKernel.cpp ``cascade_sum``; unchanged from 1.7 --
    the reference's requirement -- to the 2.10 installed here; ``sum_stub`` is
    registered up to AVX2, so the vector width is 8 floats on every x86 build).
    Restated from its published source:

      * n >= 8: view the first (n//8)*8 elements as n//8 vectors of 8 lanes.  Four
        interleaved vector accumulators (vector v -> accumulator v % 4) run over
        passes of four vectors; every ``level_step`` = 16 passes accumulator level 0
        is folded into level 1 (and so on, 4 levels), left-over passes stay in level
        0, then levels 1..3 are added to level 0.  Vectors beyond the last full pass
        are added to accumulator 0, then accumulators 1..3 are added to 0 in order.
        The scalar tail is summed ascending from 0, then lanes 0..7 are added.
      * n < 8: the same scheme on scalars (4 interleaved partial sums).

    Adding the zeros of masked-out entries is exact, so this also gives the value
    of the masked sum.  Checked bit-for-bit against torch.sum for n = 1..8225 in
    tests/test_oracle_cluster.py.
    """
    f = np.float32
    M = np.ascontiguousarray(M, dtype=f)
    R, n = M.shape

    def multi_row(X):                     # X [R, size, ...] -> sum over axis 1
        size = X.shape[1]
        ilp = 4
        size_ilp = size // ilp
        shp = (R, ilp) + X.shape[2:]
        levels = 4
        lp = max(4, (int(np.ceil(np.log2(size_ilp))) if size_ilp > 1 else 0) // levels)
        step = 1 << lp
        mask = step - 1
        acc = [np.zeros(shp, f) for _ in range(levels)]
        i = 0
        while i + step <= size_ilp:
            for _ in range(step):
                acc[0] = acc[0] + X[:, i * ilp:(i + 1) * ilp]
                i += 1
            for j in range(1, levels):
                acc[j] = acc[j] + acc[j - 1]
                acc[j - 1] = np.zeros(shp, f)
                if (i & (mask << (j * lp))) != 0:
                    break
        while i < size_ilp:
            acc[0] = acc[0] + X[:, i * ilp:(i + 1) * ilp]
            i += 1
        for j in range(1, levels):
            acc[0] = acc[0] + acc[j]
        part = acc[0]
        p0 = part[:, 0]
        for t in range(size_ilp * ilp, size):
            p0 = p0 + X[:, t]
        for k in range(1, ilp):
            p0 = p0 + part[:, k]
        return p0

    V = 8
    if n < V:
        return multi_row(M.reshape(R, n))
    vs = n // V
    lanes = multi_row(M[:, :vs * V].reshape(R, vs, V))
    out = np.zeros(R, f)
    for t in range(vs * V, n):
        out = out + M[:, t]
    for l in range(V):
        out = out + lanes[:, l]
    return out


def select_streamlined(D, first, K, iter_limit=60, id_sort=True):
    """Per-problem selection from one finished D [N,N] (numpy fp32), using only
    the exact equivalences of SURVEY §8(a):

      1. KKZ with a running minimum over rows of D;
      2. update via cluster membership: s_i = sum_j D[i,j]*[a_j == a_i] in ATen's
         association (``aten_row_sums``), medoid = member with the smallest s_i
         (lowest index on ties; an empty cluster yields index 0, as argmin over an
         all-zero row does in the reference);
      4. stop when the medoid vector is unchanged (fixed point) or at iter_limit;
      6. final sort + re-assignment.
    ``first`` is the first KKZ medoid (argmax of the token L2 norms).
    Returns (assign [N] i64, medoids [K] i64, iterations).
    """
    D = np.asarray(D, dtype=np.float32)
    N = D.shape[0]
    med = np.empty(K, dtype=np.int64)
    med[0] = first
    nearest = D[first].copy()
    for i in range(1, K):
        m = int(np.argmax(nearest))                 # first max
        med[i] = m
        nearest = np.minimum(nearest, D[m])
    steps = 0
    assign = np.zeros(N, dtype=np.int64)
    for _ in range(iter_limit):
        steps += 1
        assign = np.argmin(D[med], axis=0)          # first k on ties
        same_cluster = assign[:, None] == assign[None, :]
        s = aten_row_sums(np.where(same_cluster, D, np.float32(0.0)))
        new = np.zeros(K, dtype=np.int64)
        for k in range(K):
            mem = np.nonzero(assign == k)[0]
            if mem.size:
                new[k] = mem[int(np.argmin(s[mem]))]  # first (lowest index) minimum
        same = np.array_equal(new, med)
        med = new
        if same:
            break
    if id_sort:
        med = np.sort(med)
        assign = np.argmin(D[med], axis=0)
    return assign.astype(np.int64), med, steps


def exact_zero_diag_distance(X, metric="euclidean", p=2.0, chunk_max=None):
    """The HIP kernel's own distance arithmetic, restated on the CPU (numpy):
    Gram via an fp32 dot per pair, squared norms taken from the Gram diagonal
    (=> d(i,i) == 0 and D symmetric), sqrt(max(n_i+n_j-2g,0)); shift by the
    chunk max and 1, diagonal minus another 1.  On exactly representable
    (integer-lattice) inputs this equals literal_pairwise_distance bit for bit
    (SURVEY §8c P1); on generic floats it differs from ATen's cdist in the last
    bits (P3).  X [B,N,W] -> D [B,N,N] fp32."""
    X = np.asarray(X, dtype=np.float32)
    if metric == "euclidean" and p == 2.0:
        g = np.einsum("bnw,bmw->bnm", X.astype(np.float64), X.astype(np.float64)).astype(np.float32)
        sq = np.einsum("bnn->bn", g)
        d2 = (sq[:, :, None] + sq[:, None, :]) - np.float32(2.0) * g
        d = np.sqrt(np.maximum(d2, np.float32(0.0)), dtype=np.float32)
    elif metric == "euclidean":
        diff = np.abs(X[:, :, None, :].astype(np.float64) - X[:, None, :, :].astype(np.float64))
        d = (diff ** p).sum(-1) ** (1.0 / p) if p != 1.0 else diff.sum(-1)
        d = d.astype(np.float32)
    else:
        nrm = np.sqrt((X.astype(np.float64) ** 2).sum(-1)).astype(np.float32) + np.float32(1e-6)
        g = np.einsum("bnw,bmw->bnm", X.astype(np.float64), X.astype(np.float64)).astype(np.float32)
        d = np.float32(1.0) - g / nrm[:, :, None] / nrm[:, None, :]
    mx = np.float32(d.max() if chunk_max is None else chunk_max)
    d = (d - mx) - np.float32(1.0)
    idx = np.arange(d.shape[-1])
    d[:, idx, idx] -= np.float32(1.0)
    return d.astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------
# N4: forward pieces of spectral clustering (modules/cluster/spectral.py) - pinned by tests/golden/r2_golden.npz (sp_*)
def spectral_laplacian(X, sigma=2.5, graph=None, mode="HeatKernel", knn_k=10, mutual=False):
    """constructW('HeatKernel' | 'KNN') + normalised Laplacian (spectral.py:42-52,79-107; squared distances as
    batched_cdist_l2, cluster_utils.py:121-133).  X [B,N,L] -> (L_sym [B,N,N], W [B,N,N])."""
    x = X.float()
    n1 = x.pow(2).sum(dim=-1, keepdim=True)
    d2 = torch.baddbmm(n1.transpose(-2, -1), x, x.transpose(-2, -1), alpha=-2).add(n1)
    W = torch.exp(-1.0 * d2 / (2 * sigma ** 2))
    if mode == "KNN":                                      # spectral.py:89-100
        k_value = torch.topk(W, knn_k, dim=-1, largest=True)[0][:, :, -1:]
        keep = W >= k_value
        keep = torch.logical_and(keep, keep.transpose(-2, -1)) if mutual else torch.logical_or(keep, keep.transpose(-2, -1))
        W = W * keep
    elif mode != "HeatKernel":
        raise NotImplementedError(mode)
    if graph is not None:
        W = W * graph
    deg = W.sum(dim=-1)
    inv = torch.diag_embed(torch.pow(deg, -0.5))
    return torch.bmm(torch.bmm(inv, torch.diag_embed(deg) - W), inv), W


def svd_sign_flip(U, S, VT):
    """batch_sign_flip_rasmus_bro (spectral.py:110-137)."""
    SVT = S.unsqueeze(-1) * VT
    sign_left = torch.sum(torch.sign(SVT) * torch.square(SVT), dim=2)
    return torch.sign(sign_left).unsqueeze(1) * U


def spatial_temporal_graph(N, tokens_per_frame, s_kernel=5, t_kernel=5):
    """spectral.py:139-165: token i (frame t, grid row h, column w) is connected to the tokens of the frames t-ht..t+ht at
    rows h-hs..h+hs and columns w-hs..w+hs that exist."""
    side = int(tokens_per_frame ** 0.5)
    frames = N // tokens_per_frame
    g = torch.zeros(N, N, dtype=torch.bool)
    ht, hs = t_kernel // 2, s_kernel // 2
    for i in range(N):
        t_, h_, w_ = i // tokens_per_frame, i % tokens_per_frame // side, i % tokens_per_frame % side
        for t in range(max(t_ - ht, 0), min(t_ + ht, frames - 1) + 1):
            for y in range(max(h_ - hs, 0), min(h_ + hs, side - 1) + 1):
                for x in range(max(w_ - hs, 0), min(w_ + hs, side - 1) + 1):
                    g[i, t * tokens_per_frame + y * side + x] = True
    return g


def literal_spectral_clustering(X, K, mode="HeatKernel", knn_k=10, metric="euclidean", threshold=1e-5, iter_limit=60,
                                norm_p=1.0, correct_sign=False, split_size=8, sigma=2.5, graph=None):
    """batch_spectral_clustering (spectral.py:17-75) with this host's LAPACK SVD as the decomposition - the reference's own
    choice; its medoids are an exact target only where the K-th and (K+1)-th singular value are separated."""
    B = X.shape[0]
    L_sym, _ = spectral_laplacian(X, sigma, graph, mode, knn_k)
    U, S, Vh = torch.linalg.svd(L_sym, full_matrices=False)
    if correct_sign:
        U = svd_sign_flip(U, S, Vh)
    Q = U[:, :, -K:]
    Q = Q / (Q.norm(p=2, dim=-1, keepdim=True) + 1e-6)
    if split_size > 1 and B > split_size:
        return literal_batch_kmedoids_with_split(Q, K, metric, threshold, iter_limit, True, norm_p, split_size)
    return literal_batch_kmedoids(Q, K, metric, threshold, iter_limit, True, norm_p)


def normalized_cut(W, assign, K):
    """sum_k cut(C_k, rest) / vol(C_k) of a partition (the quantity spectral clustering relaxes), per problem; float64."""
    W = W.double()
    deg = W.sum(dim=-1)
    out = torch.zeros(W.shape[0], dtype=torch.float64)
    for k in range(K):
        m = (assign == k).double()
        vol = (deg * m).sum(dim=-1)
        inside = torch.einsum("bi,bij,bj->b", m, W, m)
        out += torch.where(vol > 0, (vol - inside) / vol.clamp_min(1e-300), torch.zeros_like(vol))
    return out
