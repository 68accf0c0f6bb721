"""Mirror of modules/cluster/fast_kmeans.py:14-97 (batched k-medoids with KKZ init)."""
import torch

from .. import _lib as L
from .. import torch_ops  # noqa: F401  (registers torch.ops.centerclip)


def _run(X, K, distance, threshold, iter_limit, id_sort, norm_p, split_size, pre_norm, return_iters=False):
    assert distance in ['euclidean', 'cosine'] and X.ndim == 3        # fast_kmeans.py:60
    L.require_device(X)
    x = X.float().contiguous()                                         # custom_fwd(cast_inputs=float32)
    assign, medoids, iters = torch.ops.centerclip.batch_kmedoids(x, int(K), L.METRIC_IDS[distance], float(norm_p),
                                                                  float(threshold), int(iter_limit), bool(id_sort),
                                                                  int(split_size), bool(pre_norm))
    return (assign, medoids, iters) if return_iters else (assign, medoids)


@torch.no_grad()
def batch_fast_kmedoids_with_split(X, K, distance='euclidean', threshold=1e-5, iter_limit=60,
                                   id_sort=True, norm_p=2.0, split_size=4, pre_norm=False):
    """X [P,N,L] -> (assign [P,N] int64, medoids [P,K] int64).

    ``split_size`` no longer bounds memory (nothing of size [B,K,N,N] exists here); it is kept
    because it defines which problems share the max of the all-negative shift
    (cluster_utils.py:36 takes the max over one chunk) and the stop test.  ``threshold`` <= 1e-5 (the reference's default; the
    scripts pass 1e-6): each problem iterates to its fixed point in ONE launch, which is the final state of the reference's
    chunk-mean stop test (fast_kmeans.py:85-88; SURVEY.md §8a, equivalence 4).  A looser ``threshold`` (round 6) runs that test
    literally: one iteration of every problem per launch, the chunk's center_shift in ATen's summation order after each,
    chunks that passed it left alone - 2 * iter_limit + 2 launches, the reference's indices (tests/golden/r6_golden.npz).
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
    assign, medoids, iters = torch.ops.centerclip.kmedoids_from_dist(distance_matrix.float().contiguous(),
                                                                      l2_norm.float().contiguous(), int(K),
                                                                      int(iter_limit), bool(id_sort))
    return (assign, medoids, iters) if return_iters else (assign, medoids)
