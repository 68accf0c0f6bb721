"""Mirror of modules/cluster/fast_kmeans.py:14-97 (batched k-medoids with KKZ init)."""
import ctypes

import torch

from .. import _lib as L


def _run(X, K, distance, threshold, iter_limit, id_sort, norm_p, split_size, pre_norm, return_iters=False):
    assert distance in ['euclidean', 'cosine'] and X.ndim == 3        # fast_kmeans.py:60
    L.require_device(X)
    x = X.float().contiguous()                                         # custom_fwd(cast_inputs=float32)
    P, N, W = x.shape
    lay = L.TokenLayout(P, 1, 1, N, N * W, 0, 0, W)
    lib = L.lib()
    medoids = torch.empty(P, K, dtype=torch.long, device=x.device)
    assign = torch.empty(P, N, dtype=torch.long, device=x.device)
    iters = torch.empty(P, dtype=torch.int32, device=x.device) if return_iters else None
    ws = L.workspace(lib.cc_cluster_workspace_bytes(P, N, W, int(bool(pre_norm))), x.device)
    L.check(lib.cc_batch_kmedoids_f32(L.ptr(x), ctypes.byref(lay), W, int(K), L.METRIC_IDS[distance], float(norm_p),
                                      float(threshold), int(iter_limit), int(bool(id_sort)), int(split_size),
                                      int(bool(pre_norm)), L.ptr(medoids), L.ptr(assign), L.ptr(iters),
                                      L.ptr(ws), ws.numel(), L.stream_ptr(x.device)), "cc_batch_kmedoids_f32")
    return (assign, medoids, iters) if return_iters else (assign, medoids)


@torch.no_grad()
def batch_fast_kmedoids_with_split(X, K, distance='euclidean', threshold=1e-5, iter_limit=60,
                                   id_sort=True, norm_p=2.0, split_size=4, pre_norm=False):
    """X [P,N,L] -> (assign [P,N] int64, medoids [P,K] int64).

    ``split_size`` no longer bounds memory (nothing of size [B,K,N,N] exists here); it is kept
    because it defines which problems share the max of the all-negative shift
    (cluster_utils.py:36 takes the max over one chunk).  ``threshold`` is accepted for
    signature parity: each problem iterates to its fixed point (equivalence 4).
    """
    return _run(X, K, distance, threshold, iter_limit, id_sort, norm_p,
                split_size if X.shape[0] > split_size else X.shape[0], pre_norm)


@torch.no_grad()
def batch_fast_kmedoids(X, K, distance='euclidean', threshold=1e-5, iter_limit=60, id_sort=True, norm_p=2.0):
    """One chunk: all problems share the shift constant (fast_kmeans.py:45-97)."""
    return _run(X, K, distance, threshold, iter_limit, id_sort, norm_p, X.shape[0], False)


@torch.no_grad()
def kmedoids_from_distance(distance_matrix, l2_norm, K, iter_limit=60, id_sort=True, return_iters=False):
    """Selection only, from a finished (already shifted) distance tensor [P,N,N] and the token
    norms [P,N] (parity level P0): KKZ + iterations + sort, fast_kmeans.py:65-97."""
    L.require_device(distance_matrix, l2_norm)
    d = distance_matrix.float().contiguous()
    nrm = l2_norm.float().contiguous()
    P, N, _ = d.shape
    lib = L.lib()
    medoids = torch.empty(P, K, dtype=torch.long, device=d.device)
    assign = torch.empty(P, N, dtype=torch.long, device=d.device)
    iters = torch.empty(P, dtype=torch.int32, device=d.device)
    L.check(lib.cc_kmedoids_from_dist_f32(L.ptr(d), L.ptr(nrm), P, N, int(K), int(iter_limit), int(bool(id_sort)),
                                          L.ptr(medoids), L.ptr(assign), L.ptr(iters), None, 0,
                                          L.stream_ptr(d.device)), "cc_kmedoids_from_dist_f32")
    return (assign, medoids, iters) if return_iters else (assign, medoids)
