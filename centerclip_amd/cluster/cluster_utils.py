"""Mirror of modules/cluster/cluster_utils.py:8-43 (pairwise_distance) and :78-118 (KKZ_init)."""
import torch

from .. import _lib as L
from .. import torch_ops  # noqa: F401  (registers torch.ops.centerclip)


def _as_batch(x):
    if x.ndim == 2:
        return x.unsqueeze(0), True
    if x.ndim == 3:
        return x, False
    raise AssertionError("expected a [N, L] or [B, N, L] tensor")


@torch.no_grad()
def pairwise_distance(data1, data2, metric='euclidean', self_nearest=True, all_negative=False, p=2.0):
    """Pairwise distance -> [N1, N2] or [B, N1, N2] fp32, same arguments as the reference.

    The hot path only ever passes ``data2 is data1`` (fast_kmeans.py:61-62): that case runs on the Gram kernel of the
    cluster op; two different sets go through a plain tiled kernel (cc_pairwise_distance_cross_f32).
    Unknown metric -> NotImplementedError, as in the reference (cluster_utils.py:33).
    """
    if metric not in L.METRIC_IDS:
        raise NotImplementedError("{} metric is not implemented".format(metric))
    if data2 is not data1 and not (data1.shape == data2.shape and data1.data_ptr() == data2.data_ptr()):
        L.require_device(data1, data2)
        a, squeeze = _as_batch(data1.float().contiguous())
        b, _ = _as_batch(data2.float().contiguous())
        dist = torch.ops.centerclip.pairwise_distance_cross(a, b, L.METRIC_IDS[metric], float(p), bool(all_negative),
                                                            bool(self_nearest))
        return dist[0] if squeeze else dist
    L.require_device(data1)
    x, squeeze = _as_batch(data1.float().contiguous())
    dist = torch.ops.centerclip.pairwise_distance(x, L.METRIC_IDS[metric], float(p), bool(all_negative), bool(self_nearest))
    return dist[0] if squeeze else dist


@torch.no_grad()
def KKZ_init(X, distance_matrix, K, batch=False):
    """KKZ initialisation (first medoid = largest L2 norm, then farthest-point traversal over
    rows of ``distance_matrix``).  batch=True: X [B,N,L], D [B,N,N] -> [B,K];
    batch=False: X [N,L], D [N,N] -> [K].

    Note: the reference's non-batched branch (cluster_utils.py:95-101) indexes COLUMNS of D
    where the batched branch indexes rows; for the symmetric matrices the hot path produces
    the two agree.  This mirror follows the batched (hot-path) semantics: for batch=False the
    matrix is transposed first so the reference's column walk is reproduced exactly.
    """
    L.require_device(X, distance_matrix)
    x = X.float().contiguous()
    d = distance_matrix.float()
    if not batch:
        x, d = x.unsqueeze(0), d.transpose(-2, -1).unsqueeze(0)
    d = d.contiguous()
    norms = torch.ops.centerclip.token_norms(x)
    _assign, medoids, _iters = torch.ops.centerclip.kmedoids_from_dist(d, norms, int(K), 0, False)
    return medoids if batch else medoids[0]
