"""ctypes binding of libcenterclip_hip.so (C ABI: include/centerclip_hip.h).

The HIP library IS the product path: there is no CPU or eager-PyTorch fallback.  If the
shared library is missing or a tensor is not on a ROCm device, calls fail loudly.
"""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libcenterclip_hip.so")

CC_OK = 0
METRIC_IDS = {"euclidean": 0, "cosine": 1}


class CenterClipHipError(RuntimeError):
    pass


class TokenLayout(ctypes.Structure):
    """struct cc_token_layout"""
    _fields_ = [("B", ctypes.c_int32), ("S", ctypes.c_int32), ("fd", ctypes.c_int32), ("n", ctypes.c_int32),
                ("stride_b", ctypes.c_int64), ("stride_s", ctypes.c_int64),
                ("stride_f", ctypes.c_int64), ("stride_i", ctypes.c_int64)]


_lib = None


class ClusterVariant(ctypes.Structure):
    """cc_cluster_variant (include/centerclip_hip.h): algorithm 0 kmedoids / 1 pooling / 2 sparse_sampling / 3 spectral,
    aggregation 0 medoid / 1 mean."""
    _fields_ = [("algorithm", ctypes.c_int32), ("aggregation", ctypes.c_int32), ("cluster_embed", ctypes.c_void_p),
                ("cls_multiplier", ctypes.c_void_p), ("fixed_ids", ctypes.c_void_p),
                ("spectral_sigma", ctypes.c_float), ("spectral_graph_mode", ctypes.c_int32),
                ("spectral_knn_k", ctypes.c_int32), ("spectral_correct_sign", ctypes.c_int32),
                ("spectral_graph", ctypes.c_void_p), ("mean_residual", ctypes.c_int32)]


def _declare(lib):
    c = ctypes
    vp, i32, i64, f32, sz = c.c_void_p, c.c_int32, c.c_int64, c.c_float, c.c_size_t
    lay = c.POINTER(TokenLayout)
    lib.cc_version.restype = c.c_char_p
    lib.cc_status_string.restype = c.c_char_p
    lib.cc_status_string.argtypes = [c.c_int]
    lib.cc_cluster_workspace_bytes.restype = sz
    lib.cc_cluster_workspace_bytes.argtypes = [i32, i32, i32, i32]
    lib.cc_token_norms_f32.argtypes = [vp, lay, i32, vp, vp, sz, vp]
    lib.cc_pairwise_distance_f32.argtypes = [vp, lay, i32, i32, f32, i32, i32, i32, vp, vp, vp, sz, vp]
    lib.cc_pairwise_distance_cross_f32.argtypes = [vp, vp, i32, i32, i32, i32, i32, f32, i32, i32, vp, vp, sz, vp]
    lib.cc_pairwise_distance_cross_f32.restype = c.c_int
    lib.cc_kmedoids_from_dist_f32.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.cc_batch_kmedoids_f32.argtypes = [vp, lay, i32, i32, i32, f32, f32, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.cc_token_cluster_f32.argtypes = [vp, i64, i64, i32, i32, i32, i32, i32, i32, i32, f32, f32, i32, i32, i32,
                                         vp, i64, i64, vp, vp, vp, vp, sz, vp]
    lib.cc_token_cluster_variant_f32.argtypes = [vp, i64, i64, i32, i32, i32, i32, i32, i32, i32, f32, f32, i32, i32, i32,
                                                 c.POINTER(ClusterVariant), vp, i64, i64, vp, vp, vp, vp, sz, vp]
    lib.cc_token_aggregate_f32.argtypes = [vp, i64, i64, i32, i32, i32, i32, i32, i32, vp, c.POINTER(ClusterVariant), vp,
                                           i64, i64, vp]
    lib.cc_token_apply_selection_f32.argtypes = [vp, i64, i64, i32, i32, i32, i32, i32, i32, c.POINTER(ClusterVariant), vp, vp,
                                                 vp, i64, i64, vp]
    lib.cc_token_apply_selection_f32.restype = c.c_int
    lib.cc_token_cluster_backward_f32.argtypes = [vp, i64, i64, i32, i32, i32, i32, i32, i32, c.POINTER(ClusterVariant), vp, vp,
                                                  vp, i64, i64, vp, i64, i64, vp, vp, vp]
    lib.cc_token_cluster_backward_f32.restype = c.c_int
    lib.cc_spectral_laplacian_f32.argtypes = [vp, lay, i32, f32, vp, vp, vp, vp, vp, sz, vp]
    lib.cc_svd_sign_flip_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    lib.cc_spectral_graph_laplacian_f32.argtypes = [vp, lay, i32, f32, i32, i32, i32, vp, vp, vp, vp, vp, sz, vp]
    lib.cc_spectral_graph_laplacian_f32.restype = c.c_int
    lib.cc_spectral_embedding_workspace_bytes.argtypes = [i32, i32]
    lib.cc_spectral_embedding_workspace_bytes.restype = sz
    lib.cc_spectral_workspace_bytes.argtypes = [i32, i32, i32]
    lib.cc_spectral_workspace_bytes.restype = sz
    lib.cc_spectral_embedding_f32.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, sz, vp]
    lib.cc_spectral_embedding_solver_f32.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp, vp, i32, vp, sz, vp]
    lib.cc_spectral_embedding_solver_f32.restype = c.c_int
    lib.cc_spectral_embedding_f32.restype = c.c_int
    lib.cc_spectral_laplacian_f32.restype = c.c_int
    lib.cc_svd_sign_flip_f32.restype = c.c_int
    for name in ("cc_token_norms_f32", "cc_pairwise_distance_f32", "cc_kmedoids_from_dist_f32",
                 "cc_batch_kmedoids_f32", "cc_token_cluster_f32", "cc_token_cluster_variant_f32",
                 "cc_token_aggregate_f32"):
        getattr(lib, name).restype = c.c_int
    return lib


def _declare_optional(lib):
    """Entry points added by later translation units (transformer / similarity kernels)."""
    try:
        from . import _lib_clip
        _lib_clip.declare(lib)
    except ImportError:
        pass


def lib():
    """Load the shared library once; raise if it has not been built (python -m centerclip_amd.build)."""
    global _lib
    if _lib is None:
        path = os.environ.get("CENTERCLIP_HIP_LIB", LIB_PATH)    # dev aid: A/B two builds in one GPU session
        if not os.path.exists(path):
            raise CenterClipHipError(
                "libcenterclip_hip.so not found at %s - build it with `python -m centerclip_amd.build` "
                "(the HIP library is the only execution path; there is no CPU fallback)" % path)
        handle = ctypes.CDLL(path)
        _declare(handle)
        _declare_optional(handle)
        _lib = handle
    return _lib


def check(status, what):
    if status != CC_OK:
        raise CenterClipHipError("%s failed: %s (%d)" % (what, lib().cc_status_string(status).decode(), status))


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise CenterClipHipError("centerclip_amd runs on MI355X only: got a %s tensor (no CPU fallback)" % t.device)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_workspaces = {}
_retired = []      # outgrown workspaces stay allocated: a hipGraph captured earlier may still hold their addresses


def workspace(nbytes, device):
    """Caller-owned scratch for the C ABI: one growing uint8 buffer per (device, stream).  A buffer that is outgrown is
    retired, never freed - a captured hipGraph replays with the addresses it was captured with, and the caching
    allocator must not hand that memory to another tensor (growth is geometric, so the retired total stays below the
    live buffer's size)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _retired.append(buf)
        grow = 0 if buf is None else buf.numel() * 3 // 2
        buf = torch.empty(max(int(nbytes), grow, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf
