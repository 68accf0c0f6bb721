"""Mirror of modules/cluster/spectral.py (cluster_algo 'spectral'), every step a HIP kernel:

    constructW ('HeatKernel' or 'KNN' [+ spatial_temporal_graph])  ->  normalised Laplacian  ->  the K eigenvectors with the
    smallest eigenvalues  ->  batch_sign_flip_rasmus_bro  ->  row-normalise  ->  k-medoids on them

The decomposition (the reference: the trailing K left singular vectors of a fp32 LAPACK SVD of L_sym) is
cc_spectral_embedding_f32: a direct solver (Householder tridiagonalisation, fp64 Sturm multi-section and inverse iteration
for the K wanted pairs, back-transformation; the matrix in LDS for N <= 196, in a global scratch up to N = 832, K = 192), with a
batched one-sided Jacobi solver for the shapes outside its scope.  What "the same result" can mean for it: eigenpairs to working
precision and the reference's singular values to 1e-5 - yes; the same *vectors* only up to sign and, where eigenvalues
coincide to rounding, up to a rotation of that eigenspace, which no two solvers share.  The k-medoids tail only sees row
distances of the K selected vectors, which are invariant to both when the K-th and (K+1)-th eigenvalue are separated
(planted-partition inputs: tested against the reference module); on token-like inputs that gap is ~1e-5 relative and the
reference does not reproduce its own medoids under a float64 solve (0 of 20 problems at N = 196, K = 49; the probe script
named in DESIGN.md §6) - there the tests report the objective instead of asserting indices.
`batch_spectral_clustering(..., eigensolver=callable)` still accepts an external decomposition (L_sym -> (U, S, Vh)).
"""
import torch

from .. import _lib as L
from .. import torch_ops  # noqa: F401  (registers torch.ops.centerclip)
from .fast_kmeans import batch_fast_kmedoids, batch_fast_kmedoids_with_split


GRAPH_MODES = {'HeatKernel': 0, 'KNN': 1}


def spatial_temporal_graph(N, tokens_per_frame, s_kernel=5, t_kernel=5):
    """[N, N] bool: token i is connected to token j when their frames are at most t_kernel // 2 apart and their grid
    positions at most s_kernel // 2 apart in both directions (spectral.py:139-165).  A constant of the module, built once."""
    side = int(tokens_per_frame ** 0.5)
    idx = torch.arange(N)
    t, hh, ww = idx // tokens_per_frame, idx % tokens_per_frame // side, idx % tokens_per_frame % side
    frames = N // tokens_per_frame
    near = lambda v, half: (v[:, None] - v[None, :]).abs() <= half
    ok = near(t, t_kernel // 2) & near(hh, s_kernel // 2) & near(ww, s_kernel // 2)
    # neighbours are enumerated from valid frames / rows / columns of the grid only; tokens of a partial last frame
    # (t == frames) or beyond the square grid are never the TARGET of an edge
    target_ok = (t < frames) & (hh < side)
    return ok & target_ok[None, :]


@torch.no_grad()
def spectral_laplacian(X, sigma=2.5, mode='HeatKernel', spatial_temporal_graph=None, return_affinity=False, knn_k=10,
                       mutual=False):
    """X [B,N,L] -> L_sym [B,N,N] = D^-1/2 (D - W) D^-1/2 with W = constructW(X, X, sigma, mode, knn_k, mutual)
    (spectral.py:42-52,79-107)."""
    if mode not in GRAPH_MODES:
        raise NotImplementedError(mode)
    L.require_device(X)                                   # (the constant graph mask may live on the host, as in the reference)
    g = None
    if spatial_temporal_graph is not None:
        g = spatial_temporal_graph.to(device=X.device).ne(0).to(torch.uint8).reshape(X.shape[1], X.shape[1]).contiguous()
    lap, aff = torch.ops.centerclip.spectral_graph_laplacian(X.float().contiguous(), False, 0, 0, float(sigma),
                                                             GRAPH_MODES[mode], int(knn_k), bool(mutual), g)
    return (lap, aff) if return_affinity else lap


@torch.no_grad()
def spectral_embedding(L_sym, K, correct_sign=False, solver="auto"):
    """-> (Q [B,N,K] = the reference's U[:, :, -K:] up to sign / rotations inside degenerate eigenspaces, eigenvalues [B,K]).
    solver: 'auto' (the direct solver wherever it applies) or 'jacobi' (the one-sided Jacobi kernel for every shape)."""
    L.require_device(L_sym)
    Q, ev, _ = torch.ops.centerclip.spectral_embedding(L_sym.float().contiguous(), int(K), bool(correct_sign),
                                                       {"auto": 0, "jacobi": 1}[solver])
    return Q[:, :, :K], ev


@torch.no_grad()
def batch_sign_flip_rasmus_bro(U, S, VT, backend="pytorch"):
    """Sign correction of the left singular vectors (spectral.py:110-137): U [B,M,K], S [B,K], VT [B,K,N]."""
    L.require_device(U, S, VT)
    return torch.ops.centerclip.svd_sign_flip(U.float().contiguous(), S.float().contiguous(), VT.float().contiguous())


@torch.no_grad()
def spectral_embedding_kmedoids(Q, K, metric='euclidean', threshold=1e-5, iter_limit=60, id_sort=True, norm_p=1.0,
                                split_size=8):
    """The tail of batch_spectral_clustering (spectral.py:58-73): Q [B,N,K] (the K trailing singular vectors) ->
    Q / (|Q| + 1e-6) row-wise -> k-medoids.  The row normalisation is the k-medoids op's own pre_norm pass."""
    B = Q.shape[0]
    if Q.shape[-1] % 4:          # the token kernels read 16-byte pieces: zero columns change no norm and no distance
        Q = torch.nn.functional.pad(Q, (0, 4 - Q.shape[-1] % 4))
    if split_size > 1 and B > split_size:
        return batch_fast_kmedoids_with_split(Q, K, distance=metric, threshold=threshold, iter_limit=iter_limit,
                                              id_sort=id_sort, norm_p=norm_p, split_size=split_size, pre_norm=True)
    return batch_fast_kmedoids_with_split(Q, K, distance=metric, threshold=threshold, iter_limit=iter_limit,
                                          id_sort=id_sort, norm_p=norm_p, split_size=B, pre_norm=True)


@torch.no_grad()
def batch_spectral_clustering(X, K, mode='HeatKernel', knn_k=10, metric='euclidean', threshold=1e-5, iter_limit=60,
                              id_sort=True, norm_p=1.0, correct_sign=False, split_size=8, sigma=2.5,
                              spatial_temporal_graph=None, eigensolver=None):
    """modules/cluster/spectral.py:17-75 -> (cluster_assignment [B,N], medoids [B,K]).  eigensolver: optional callable
    L_sym -> (U, S, Vh) replacing the built-in decomposition."""
    assert metric in ['euclidean', 'cosine'] and X.ndim == 3
    L_sym = spectral_laplacian(X, sigma=sigma, mode=mode, spatial_temporal_graph=spatial_temporal_graph, knn_k=knn_k)
    return _cluster_from_laplacian(L_sym, K, metric, threshold, iter_limit, id_sort, norm_p, correct_sign, split_size,
                                   eigensolver)


def _cluster_from_laplacian(L_sym, K, metric, threshold, iter_limit, id_sort, norm_p, correct_sign, split_size, eigensolver):
    if eigensolver is None:
        Q4, _, _ = torch.ops.centerclip.spectral_embedding(L_sym, int(K), bool(correct_sign))   # zero-padded to K % 4 == 0
        return spectral_embedding_kmedoids(Q4, K, metric, threshold, iter_limit, id_sort, norm_p, split_size)
    U, S, Vh = eigensolver(L_sym)
    if correct_sign:
        U = batch_sign_flip_rasmus_bro(U, S, Vh)
    Q = U[:, :, -K:].contiguous()
    return spectral_embedding_kmedoids(Q, K, metric, threshold, iter_limit, id_sort, norm_p, split_size)


@torch.no_grad()
def spectral_clustering_of_tokens(x, frame_major, T, T_new, K, mode='HeatKernel', knn_k=10, metric='euclidean',
                                  threshold=1e-5, iter_limit=60, id_sort=True, norm_p=1.0, correct_sign=False, split_size=8,
                                  sigma=2.5, graph=None, eigensolver=None):
    """batch_spectral_clustering on the patch tokens of TokenClusterInter's activations ([1+n, B*T, W] or frame-major),
    regrouped into T_new segments per clip through strides (no copy): -> (assign [T_new*B, fd*n], medoids [T_new*B, K])."""
    L.require_device(x)
    L_sym, _ = torch.ops.centerclip.spectral_graph_laplacian(x, bool(frame_major), int(T), int(T_new), float(sigma),
                                                             GRAPH_MODES[mode], int(knn_k), False, graph)
    return _cluster_from_laplacian(L_sym, K, metric, threshold, iter_limit, id_sort, norm_p, correct_sign, split_size,
                                   eigensolver)
