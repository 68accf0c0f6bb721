"""Mirror of the well-posed pieces of modules/cluster/spectral.py (cluster_algo 'spectral'):

    constructW ('HeatKernel' [+ spatial_temporal_graph])  ->  normalised Laplacian  ->  [ eigen-decomposition ]  ->
    batch_sign_flip_rasmus_bro  ->  row-normalise the K trailing vectors  ->  k-medoids on them

Everything except the bracketed step runs as HIP kernels.  The eigen-decomposition is deliberately NOT built: the reference
takes the K singular vectors of L_sym with the smallest singular values from a fp32 LAPACK SVD, and on token-like inputs the
gap between the K-th and (K+1)-th value is ~1e-5 relative, so the subspace depends on the solver's rounding (its own fp64
run reproduces 0 of 20 medoid sets at N = 196; measured by the probe script named in DESIGN.md §6) - there is no parity
target to build to.
`batch_spectral_clustering` therefore takes the decomposition as a callable (``eigensolver(L_sym) -> (U, S, Vh)``, e.g.
``torch.linalg.svd`` on the device) and raises without one; parity is asserted on L_sym, on the sign flip and on the
tail given the reference's own embedding.
"""
import torch

from .. import _lib as L
from .. import torch_ops  # noqa: F401  (registers torch.ops.centerclip)
from .fast_kmeans import batch_fast_kmedoids, batch_fast_kmedoids_with_split


@torch.no_grad()
def spectral_laplacian(X, sigma=2.5, mode='HeatKernel', spatial_temporal_graph=None, return_affinity=False):
    """X [B,N,L] -> L_sym [B,N,N] = D^-1/2 (D - W) D^-1/2 with W = constructW(X, X, sigma, mode) (spectral.py:42-52)."""
    if mode != 'HeatKernel':
        raise NotImplementedError("only the 'HeatKernel' graph is built (spectral.py:86-88); got %r" % (mode,))
    L.require_device(X)                                   # (the constant graph mask may live on the host, as in the reference)
    g = None
    if spatial_temporal_graph is not None:
        g = spatial_temporal_graph.to(device=X.device).ne(0).to(torch.uint8).contiguous()
    lap, aff = torch.ops.centerclip.spectral_laplacian(X.float().contiguous(), float(sigma), g)
    return (lap, aff) if return_affinity else lap


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
    """modules/cluster/spectral.py:17-75 with the decomposition supplied by the caller (see the module docstring).
    -> (cluster_assignment [B,N], medoids [B,K])."""
    assert metric in ['euclidean', 'cosine'] and X.ndim == 3
    if eigensolver is None:
        raise NotImplementedError("the eigen-decomposition of spectral clustering is not built (no parity definition): "
                                  "pass eigensolver=callable(L_sym) -> (U, S, Vh)")
    L_sym = spectral_laplacian(X, sigma=sigma, mode=mode, spatial_temporal_graph=spatial_temporal_graph)
    U, S, Vh = eigensolver(L_sym)
    if correct_sign:
        U = batch_sign_flip_rasmus_bro(U, S, Vh)
    Q = U[:, :, -K:].contiguous()
    return spectral_embedding_kmedoids(Q, K, metric, threshold, iter_limit, id_sort, norm_p, split_size)
