"""``torch.library`` registration of the C-ABI entry points (namespace ``centerclip``): the boundary the north star
names - "called from Python via PyTorch-ROCm custom ops that keep the modules/clip.py and modules/clip4clip.py forward
signatures" (SURVEY.md §8b).  Every op

  * has a schema in the dispatcher (``torch.ops.centerclip.<name>``), so it is visible to torch.compile / export /
    profilers like any ATen op;
  * has a CUDA(=ROCm) implementation that only enqueues kernels of libcenterclip_hip.so on the current stream through
    the ctypes binding (no CPU implementation is registered: a CPU tensor fails in the dispatcher, loudly);
  * has a fake (meta) kernel giving output shapes / dtypes without touching the device.

The module mirrors (clip.py, clip4clip.py, cluster/*, metrics.py) and ops.py call these ops; nothing else in the package
calls the ctypes layer for compute.  The encoders take their weights through an integer *model handle*
(``register_model``): the C structs hold raw device pointers, which a dispatcher schema cannot carry.
"""
import ctypes
import math
from typing import Optional, Tuple

import torch
from torch.library import custom_op

from . import _lib as L
from ._lib_clip import EPI

NS = "centerclip"
LN_MAX_SLOTS = 32

# ------------------------------------------------------------------------------------------ model handles
_MODELS = {}
_next_handle = [1]


def register_model(struct, meta, keep):
    """Park a packed cc_vit_model / cc_text_model (+ the tensors its pointers refer to) -> int handle."""
    h = _next_handle[0]
    _next_handle[0] += 1
    _MODELS[h] = (struct, meta, keep)
    return h


def release_model(handle):
    _MODELS.pop(handle, None)


def _model(handle):
    try:
        return _MODELS[handle]
    except KeyError:
        raise L.CenterClipHipError("unknown centerclip model handle %r" % (handle,))


def _st(t):
    return L.stream_ptr(t.device)


def _e(*shape, like, dtype):
    return torch.empty(shape, device=like.device, dtype=dtype)


# ------------------------------------------------------------------------------------------ Linear / LN / attention
@custom_op(NS + "::linear_f16", mutates_args=(), device_types="cuda")
def linear_f16(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], epilogue: str, tile: int) -> torch.Tensor:
    M, K = a.shape
    N = w.shape[0]
    out = _e(M, N, like=a, dtype=torch.float16 if epilogue.startswith("f16") else torch.float32)
    L.check(L.lib().cc_linear_f16(L.ptr(a), L.ptr(w), L.ptr(bias), L.ptr(out), M, N, K, N, EPI[epilogue], tile, _st(a)),
            "cc_linear_f16")
    return out


@linear_f16.register_fake
def _(a, w, bias, epilogue, tile):
    return a.new_empty((a.shape[0], w.shape[0]), dtype=torch.float16 if epilogue.startswith("f16") else torch.float32)


@custom_op(NS + "::linear_f16_out", mutates_args=("out",), device_types="cuda")
def linear_f16_out(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, epilogue: str,
                   tile: int) -> None:
    """epilogue 'f32_resid': out += a w^T + b in place; the other epilogues overwrite ``out`` (row stride honoured)."""
    M, K = a.shape
    L.check(L.lib().cc_linear_f16(L.ptr(a), L.ptr(w), L.ptr(bias), L.ptr(out), M, w.shape[0], K, out.stride(0),
                                  EPI[epilogue], tile, _st(a)), "cc_linear_f16")


@linear_f16_out.register_fake
def _(a, w, bias, out, epilogue, tile):
    return None


@custom_op(NS + "::layernorm", mutates_args=(), device_types="cuda")
def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float, out_f16: bool) -> torch.Tensor:
    W = x.shape[-1]
    rows = x.numel() // W
    out = torch.empty(x.shape, device=x.device, dtype=torch.float16 if out_f16 else torch.float32)
    L.check(L.lib().cc_layernorm_f32(L.ptr(x), W, L.ptr(weight), L.ptr(bias), L.ptr(out), W, rows, W, float(eps),
                                     int(out_f16), _st(x)), "cc_layernorm_f32")
    return out


@layernorm.register_fake
def _(x, weight, bias, eps, out_f16):
    return x.new_empty(x.shape, dtype=torch.float16 if out_f16 else torch.float32)


@custom_op(NS + "::attention_f16", mutates_args=(), device_types="cuda")
def attention_f16(qkv: torch.Tensor, nseq: int, L_tok: int, heads: int, causal: bool, seq_rows: int,
                  tok_rows: int) -> torch.Tensor:
    W = qkv.shape[1] // 3
    out = _e(nseq * L_tok, W, like=qkv, dtype=torch.float16)
    L.check(L.lib().cc_attention_strided_f16(L.ptr(qkv), L.ptr(out), nseq, L_tok, heads, W, int(causal), seq_rows, tok_rows,
                                             _st(qkv)), "cc_attention_strided_f16")
    return out


@attention_f16.register_fake
def _(qkv, nseq, L_tok, heads, causal, seq_rows, tok_rows):
    return qkv.new_empty((nseq * L_tok, qkv.shape[1] // 3))


@custom_op(NS + "::fold_layernorm_linear", mutates_args=(), device_types="cuda")
def fold_layernorm_linear(weight: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor,
                          beta: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    N, K = weight.shape
    wf = _e(N, K, like=weight, dtype=torch.float16)
    c1 = _e(N, like=weight, dtype=torch.float32)
    c2 = _e(N, like=weight, dtype=torch.float32)
    L.check(L.lib().cc_fold_layernorm_linear_f32(L.ptr(weight), L.ptr(bias), L.ptr(gamma), L.ptr(beta), N, K, L.ptr(wf),
                                                 L.ptr(c1), L.ptr(c2), _st(weight)), "cc_fold_layernorm_linear_f32")
    return wf, c1, c2


@fold_layernorm_linear.register_fake
def _(weight, bias, gamma, beta):
    N, K = weight.shape
    return weight.new_empty((N, K), dtype=torch.float16), weight.new_empty((N,)), weight.new_empty((N,))


@custom_op(NS + "::row_stats", mutates_args=(), device_types="cuda")
def row_stats(h: torch.Tensor, centre: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    M, W = h.shape
    h16 = _e(M, W, like=h, dtype=torch.float16)
    stats = _e(M, 2, like=h, dtype=torch.float32)
    shift = _e(M if centre else 0, like=h, dtype=torch.float32)
    L.check(L.lib().cc_row_stats_f16(L.ptr(h), L.ptr(h16), L.ptr(stats), L.ptr(shift) if centre else None, M, W, _st(h)),
            "cc_row_stats_f16")
    return h16, stats, shift


@row_stats.register_fake
def _(h, centre):
    M, W = h.shape
    return h.new_empty((M, W), dtype=torch.float16), h.new_empty((M, 2)), h.new_empty((M if centre else 0,))


@custom_op(NS + "::linear_ln_f16", mutates_args=(), device_types="cuda")
def linear_ln_f16(h16: torch.Tensor, w_ln: torch.Tensor, c1: torch.Tensor, c2: torch.Tensor, stats: torch.Tensor,
                  slots: int, gelu: bool, eps: float, tile: int) -> torch.Tensor:
    M, K = h16.shape
    N = w_ln.shape[0]
    out = _e(M, N, like=h16, dtype=torch.float16)
    L.check(L.lib().cc_linear_ln_f16(L.ptr(h16), L.ptr(w_ln), L.ptr(c1), L.ptr(c2), L.ptr(stats), int(slots), float(eps),
                                     L.ptr(out), M, N, K, int(gelu), tile, _st(h16)), "cc_linear_ln_f16")
    return out


@linear_ln_f16.register_fake
def _(h16, w_ln, c1, c2, stats, slots, gelu, eps, tile):
    return h16.new_empty((h16.shape[0], w_ln.shape[0]))


@custom_op(NS + "::inproj_attention_f16", mutates_args=(), device_types="cuda")
def inproj_attention_f16(h16: torch.Tensor, w_ln: torch.Tensor, c1: torch.Tensor, c2: torch.Tensor, stats: torch.Tensor,
                         slots: int, eps: float, nseq: int, L_tok: int, heads: int, causal: bool,
                         seq_off: Optional[torch.Tensor], seq_len: Optional[torch.Tensor]) -> torch.Tensor:
    """LN-folded in_proj + attention in one launch (frame-major rows); -> att [nseq * L_tok, W] fp16."""
    M, W = h16.shape
    out = torch.zeros(M, W, device=h16.device, dtype=torch.float16) if seq_off is not None else _e(M, W, like=h16, dtype=torch.float16)
    L.check(L.lib().cc_inproj_attention_f16(L.ptr(h16), L.ptr(w_ln), L.ptr(c1), L.ptr(c2), L.ptr(stats), int(slots), float(eps),
                                            L.ptr(out), nseq, L_tok, heads, int(causal), L.ptr(seq_off), L.ptr(seq_len), None,
                                            _st(h16)), "cc_inproj_attention_f16")
    return out


@inproj_attention_f16.register_fake
def _(h16, w_ln, c1, c2, stats, slots, eps, nseq, L_tok, heads, causal, seq_off, seq_len):
    return h16.new_empty(h16.shape)


@custom_op(NS + "::linear_resid_stats_f16", mutates_args=("h", "h16", "stats", "shift_out"), device_types="cuda")
def linear_resid_stats_f16(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], h: torch.Tensor,
                           h16: torch.Tensor, stats: torch.Tensor, shift_in: Optional[torch.Tensor],
                           stats_in: Optional[torch.Tensor], slots_in: int, shift_out: Optional[torch.Tensor],
                           tile: int) -> None:
    """h += a w^T + b; h16 = fp16(h - c_row); stats (flat, >= M * resid_stats_slots * 2 floats) <- partial sums."""
    M, K = a.shape
    N = w.shape[0]
    slots = ctypes.c_int32(0)
    L.check(L.lib().cc_linear_resid_stats_f16(L.ptr(a), L.ptr(w), L.ptr(bias), L.ptr(h), L.ptr(h16), L.ptr(stats),
                                              ctypes.byref(slots), L.ptr(shift_in), L.ptr(stats_in), int(slots_in),
                                              L.ptr(shift_out), M, N, K, tile, _st(a)), "cc_linear_resid_stats_f16")


@linear_resid_stats_f16.register_fake
def _(a, w, bias, h, h16, stats, shift_in, stats_in, slots_in, shift_out, tile):
    return None


def resid_stats_slots(M, N, K, tile=0):
    """Partial-sum slots per row the residual Linear writes for this shape (a host-side query of the tile choice)."""
    n = L.lib().cc_linear_resid_stats_slots(int(M), int(N), int(K), int(tile))
    if n <= 0:
        raise L.CenterClipHipError("cc_linear_resid_stats_slots(%d, %d, %d, %d): unsupported shape / tile" % (M, N, K, tile))
    return n


@custom_op(NS + "::head_project", mutates_args=(), device_types="cuda")
def head_project(h: torch.Tensor, row_mul: int, row_idx: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                 proj: torch.Tensor, R: int) -> torch.Tensor:
    W, E = proj.shape
    out = _e(R, E, like=h, dtype=torch.float32)
    L.check(L.lib().cc_head_project_f32(L.ptr(h), row_mul, L.ptr(row_idx), L.ptr(gamma), L.ptr(beta), L.ptr(proj),
                                        L.ptr(out), R, W, E, _st(h)), "cc_head_project_f32")
    return out


@head_project.register_fake
def _(h, row_mul, row_idx, gamma, beta, proj, R):
    return h.new_empty((R, proj.shape[1]))


# ------------------------------------------------------------------------------------------ token cluster
def _variant(algorithm, aggregation, cluster_embed, cls_mult, fixed_ids, spectral=None):
    var = L.ClusterVariant()
    var.algorithm, var.aggregation = int(algorithm), int(aggregation)
    var.cluster_embed = cluster_embed.data_ptr() if cluster_embed is not None else None
    var.cls_multiplier = cls_mult.data_ptr() if cls_mult is not None else None
    var.fixed_ids = fixed_ids.data_ptr() if fixed_ids is not None else None
    if spectral is not None:                       # (sigma, graph mode, knn_k, correct_sign, graph [N,N] uint8 | None)
        sigma, mode, knn_k, sign, graph = spectral
        var.spectral_sigma, var.spectral_graph_mode, var.spectral_knn_k = float(sigma), int(mode), int(knn_k)
        var.spectral_correct_sign = int(bool(sign))
        var.spectral_graph = graph.data_ptr() if graph is not None else None
    return var


def _cluster_ws(lib, P, N, W, pre_norm, K, algorithm, device):
    need = lib.cc_cluster_workspace_bytes(P, N, W, int(pre_norm))
    if algorithm == 3:
        need += lib.cc_spectral_workspace_bytes(P, N, K)
    return L.workspace(need, device)


@custom_op(NS + "::token_cluster", mutates_args=(), device_types="cuda")
def token_cluster(x: torch.Tensor, frame_major: bool, T: int, T_new: int, K: int, metric: int, norm_p: float,
                  threshold: float, iter_limit: int, split_size: int, pre_norm: bool, algorithm: int, aggregation: int,
                  cluster_embed: Optional[torch.Tensor], cls_mult: Optional[torch.Tensor], fixed_ids: Optional[torch.Tensor],
                  want_medoids: bool, spectral_sigma: float = 0.0, spectral_mode: int = 0, spectral_knn_k: int = 0,
                  spectral_sign: bool = False, spectral_graph: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """TokenClusterInter.forward (modules/cluster/cluster.py:206-352).  x contiguous fp32: [1+n, B*T, W] (LND, the
    reference's layout) or [B*T, 1+n, W] (frame_major) -> same layout with T_new segments of 1+K tokens; medoids
    [T_new*B, K] int64 (empty unless want_medoids and algorithm 0 / 3).  algorithm 3 = 'spectral' (the spectral_* arguments:
    sigma, graph mode 0 HeatKernel / 1 KNN, knn_k, sign correction, optional [N,N] uint8 spatial-temporal mask)."""
    if frame_major:
        BT, Lt, W = x.shape
        tok, frame = W, Lt * W
    else:
        Lt, BT, W = x.shape
        tok, frame = BT * W, W
    B, n = BT // T, Lt - 1
    if frame_major:
        out = _e(B * T_new, 1 + K, W, like=x, dtype=torch.float32)
        o_tok, o_frame = W, (1 + K) * W
    else:
        out = _e(1 + K, B * T_new, W, like=x, dtype=torch.float32)
        o_tok, o_frame = B * T_new * W, W
    kmed = algorithm in (0, 3)
    med = _e(B * T_new if (want_medoids and kmed) else 0, K, like=x, dtype=torch.long)
    var = _variant(algorithm, aggregation, cluster_embed, cls_mult, fixed_ids,
                   (spectral_sigma, spectral_mode, spectral_knn_k, spectral_sign, spectral_graph) if algorithm == 3 else None)
    lib = L.lib()
    N = (T // T_new) * n
    ws = _cluster_ws(lib, B * T_new, N, W, pre_norm, K, algorithm, x.device)
    L.check(lib.cc_token_cluster_variant_f32(L.ptr(x), tok, frame, B, T, T_new, n, W, K, metric, float(norm_p),
                                             float(threshold), int(iter_limit), int(split_size), int(pre_norm),
                                             ctypes.byref(var), L.ptr(out), o_tok, o_frame,
                                             L.ptr(med) if med.numel() else None, None, None, L.ptr(ws), ws.numel(),
                                             _st(x)), "cc_token_cluster_variant_f32")
    return out, med


@token_cluster.register_fake
def _(x, frame_major, T, T_new, K, metric, norm_p, threshold, iter_limit, split_size, pre_norm, algorithm, aggregation,
      cluster_embed, cls_mult, fixed_ids, want_medoids, spectral_sigma=0.0, spectral_mode=0, spectral_knn_k=0,
      spectral_sign=False, spectral_graph=None):
    if frame_major:
        BT, Lt, W = x.shape
        out = x.new_empty((BT // T * T_new, 1 + K, W))
    else:
        Lt, BT, W = x.shape
        out = x.new_empty((1 + K, BT // T * T_new, W))
    rows = BT // T * T_new if (want_medoids and algorithm in (0, 3)) else 0
    return out, x.new_empty((rows, K), dtype=torch.long)


def _cluster_strides(shape, frame_major):
    if frame_major:
        BT, Lt, W = shape
        return BT, Lt, W, W, Lt * W
    Lt, BT, W = shape
    return BT, Lt, W, BT * W, W


@custom_op(NS + "::token_cluster_train", mutates_args=(), device_types="cuda")
def token_cluster_train(x: torch.Tensor, frame_major: bool, T: int, T_new: int, K: int, metric: int, norm_p: float,
                        threshold: float, iter_limit: int, split_size: int, pre_norm: bool, algorithm: int, aggregation: int,
                        cluster_embed: Optional[torch.Tensor], cls_mult: Optional[torch.Tensor],
                        fixed_ids: Optional[torch.Tensor], spectral_sigma: float = 0.0, spectral_mode: int = 0,
                        spectral_knn_k: int = 0, spectral_sign: bool = False,
                        spectral_graph: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """token_cluster that also returns what its backward needs: medoids [T_new*B, K] and assign [T_new*B, fd*n] (both
    empty for 'pooling' / 'sparse_sampling').  Differentiable with respect to x, cluster_embed and cls_mult
    (torch.ops.centerclip.token_cluster_backward); the selection is a constant of the backward pass, as in the reference,
    whose k-medoids runs under no_grad (fast_kmeans.py:13,44)."""
    BT, Lt, W, tok, frame = _cluster_strides(x.shape, frame_major)
    B, n = BT // T, Lt - 1
    oshape = (B * T_new, 1 + K, W) if frame_major else (1 + K, B * T_new, W)
    out = _e(*oshape, like=x, dtype=torch.float32)
    _, _, _, o_tok, o_frame = _cluster_strides(oshape, frame_major)
    kmed = algorithm in (0, 3)
    N = (T // T_new) * n
    med = _e(B * T_new if kmed else 0, K, like=x, dtype=torch.long)
    assign = _e(B * T_new if kmed else 0, N, like=x, dtype=torch.long)
    var = _variant(algorithm, aggregation, cluster_embed, cls_mult, fixed_ids,
                   (spectral_sigma, spectral_mode, spectral_knn_k, spectral_sign, spectral_graph) if algorithm == 3 else None)
    lib = L.lib()
    ws = _cluster_ws(lib, B * T_new, N, W, pre_norm, K, algorithm, x.device)
    L.check(lib.cc_token_cluster_variant_f32(L.ptr(x), tok, frame, B, T, T_new, n, W, K, metric, float(norm_p),
                                             float(threshold), int(iter_limit), int(split_size), int(pre_norm),
                                             ctypes.byref(var), L.ptr(out), o_tok, o_frame,
                                             L.ptr(med) if kmed else None, L.ptr(assign) if kmed else None, None,
                                             L.ptr(ws), ws.numel(), _st(x)), "cc_token_cluster_variant_f32")
    return out, med, assign


@token_cluster_train.register_fake
def _(x, frame_major, T, T_new, K, metric, norm_p, threshold, iter_limit, split_size, pre_norm, algorithm, aggregation,
      cluster_embed, cls_mult, fixed_ids, spectral_sigma=0.0, spectral_mode=0, spectral_knn_k=0, spectral_sign=False,
      spectral_graph=None):
    BT, Lt, W, _, _ = _cluster_strides(x.shape, frame_major)
    B, n = BT // T, Lt - 1
    out = x.new_empty((B * T_new, 1 + K, W) if frame_major else (1 + K, B * T_new, W))
    rows = B * T_new if algorithm in (0, 3) else 0
    return out, x.new_empty((rows, K), dtype=torch.long), x.new_empty((rows, (T // T_new) * n), dtype=torch.long)


@custom_op(NS + "::token_apply_selection", mutates_args=(), device_types="cuda")
def token_apply_selection(x: torch.Tensor, frame_major: bool, T: int, T_new: int, K: int, aggregation: int,
                          medoids: torch.Tensor, assign: torch.Tensor, cluster_embed: Optional[torch.Tensor],
                          cls_mult: Optional[torch.Tensor]) -> torch.Tensor:
    """The gather / aggregation half of TokenClusterInter.forward (cluster.py:287-310) for a selection made elsewhere
    (cluster_algo 'spectral'): medoids [T_new*B, K] or, for cluster means, assign [T_new*B, fd*n]."""
    BT, Lt, W, tok, frame = _cluster_strides(x.shape, frame_major)
    B, n = BT // T, Lt - 1
    oshape = (B * T_new, 1 + K, W) if frame_major else (1 + K, B * T_new, W)
    out = _e(*oshape, like=x, dtype=torch.float32)
    _, _, _, o_tok, o_frame = _cluster_strides(oshape, frame_major)
    var = _variant(0, aggregation, cluster_embed, cls_mult, None)
    L.check(L.lib().cc_token_apply_selection_f32(L.ptr(x), tok, frame, B, T, T_new, n, W, K, ctypes.byref(var),
                                                 L.ptr(medoids) if medoids.numel() else None,
                                                 L.ptr(assign) if assign.numel() else None, L.ptr(out), o_tok, o_frame,
                                                 _st(x)), "cc_token_apply_selection_f32")
    return out


@token_apply_selection.register_fake
def _(x, frame_major, T, T_new, K, aggregation, medoids, assign, cluster_embed, cls_mult):
    BT, Lt, W, _, _ = _cluster_strides(x.shape, frame_major)
    B = BT // T
    return x.new_empty((B * T_new, 1 + K, W) if frame_major else (1 + K, B * T_new, W))


@custom_op(NS + "::token_cluster_backward", mutates_args=(), device_types="cuda")
def token_cluster_backward(grad_out: torch.Tensor, x: torch.Tensor, frame_major: bool, T: int, T_new: int, K: int,
                           algorithm: int, aggregation: int, medoids: torch.Tensor, assign: torch.Tensor,
                           cls_mult: Optional[torch.Tensor], fixed_ids: Optional[torch.Tensor], want_embed: bool,
                           want_mult: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Gradient of token_cluster_train for a fixed selection (cc_token_cluster_backward_f32): -> (grad_x like x,
    grad_cluster_embed [K, W] or empty, grad_cls_mult [T] or empty)."""
    BT, Lt, W, tok, frame = _cluster_strides(x.shape, frame_major)
    B, n = BT // T, Lt - 1
    _, _, _, g_tok, g_frame = _cluster_strides(grad_out.shape, frame_major)
    gx = torch.empty_like(x)
    g_embed = _e(K if want_embed else 0, W, like=x, dtype=torch.float32)
    g_mult = _e(T if want_mult else 0, like=x, dtype=torch.float32)
    var = _variant(algorithm, aggregation, None, cls_mult, fixed_ids)
    L.check(L.lib().cc_token_cluster_backward_f32(L.ptr(grad_out), g_tok, g_frame, B, T, T_new, n, W, K, ctypes.byref(var),
                                                  L.ptr(medoids) if medoids.numel() else None,
                                                  L.ptr(assign) if assign.numel() else None, L.ptr(x), tok, frame,
                                                  L.ptr(gx), tok, frame, L.ptr(g_embed) if want_embed else None,
                                                  L.ptr(g_mult) if want_mult else None, _st(x)),
            "cc_token_cluster_backward_f32")
    return gx, g_embed, g_mult


@token_cluster_backward.register_fake
def _(grad_out, x, frame_major, T, T_new, K, algorithm, aggregation, medoids, assign, cls_mult, fixed_ids, want_embed,
      want_mult):
    W = x.shape[-1]
    return torch.empty_like(x), x.new_empty((K if want_embed else 0, W)), x.new_empty((T if want_mult else 0,))


def _token_cluster_setup(ctx, inputs, output):
    (x, frame_major, T, T_new, K, _m, _p, _t, _i, _s, _pn, algorithm, aggregation, cluster_embed, cls_mult, fixed_ids) = inputs[:16]
    algorithm = 0 if algorithm == 3 else algorithm          # spectral: the backward is that of the gather / cluster means
    _, med, assign = output
    ctx.cfg = (frame_major, T, T_new, K, algorithm, aggregation)
    ctx.has = (cluster_embed is not None, cls_mult is not None, fixed_ids is not None)
    ctx.save_for_backward(x, med, assign, *(t for t in (cls_mult, fixed_ids) if t is not None))


def _token_cluster_bwd(ctx, grad_out, _gmed, _gassign):
    frame_major, T, T_new, K, algorithm, aggregation = ctx.cfg
    has_embed, has_mult, has_ids = ctx.has
    saved = list(ctx.saved_tensors)
    x, med, assign = saved[:3]
    rest = saved[3:]
    cls_mult = rest.pop(0) if has_mult else None
    fixed_ids = rest.pop(0) if has_ids else None
    want_embed = has_embed and ctx.needs_input_grad[13]
    want_mult = has_mult and ctx.needs_input_grad[14]
    gx, ge, gm = torch.ops.centerclip.token_cluster_backward(grad_out.contiguous().float(), x, frame_major, T, T_new, K,
                                                             algorithm, aggregation, med, assign, cls_mult, fixed_ids,
                                                             want_embed, want_mult)
    return (gx,) + (None,) * 12 + (ge if want_embed else None, gm if want_mult else None) + (None,) * 6


token_cluster_train.register_autograd(_token_cluster_bwd, setup_context=_token_cluster_setup)


@custom_op(NS + "::batch_kmedoids", mutates_args=(), device_types="cuda")
def batch_kmedoids(x: torch.Tensor, K: int, metric: int, norm_p: float, threshold: float, iter_limit: int, id_sort: bool,
                   split_size: int, pre_norm: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """batch_fast_kmedoids_with_split (modules/cluster/fast_kmeans.py:14-97): x [P,N,W] -> (assign [P,N] int64,
    medoids [P,K] int64, iterations [P] int32)."""
    P, N, W = x.shape
    lay = L.TokenLayout(P, 1, 1, N, N * W, 0, 0, W)
    lib = L.lib()
    medoids = _e(P, K, like=x, dtype=torch.long)
    assign = _e(P, N, like=x, dtype=torch.long)
    iters = _e(P, like=x, dtype=torch.int32)
    ws = L.workspace(lib.cc_cluster_workspace_bytes(P, N, W, int(pre_norm)), x.device)
    L.check(lib.cc_batch_kmedoids_f32(L.ptr(x), ctypes.byref(lay), W, int(K), metric, float(norm_p), float(threshold),
                                      int(iter_limit), int(id_sort), int(split_size), int(pre_norm), L.ptr(medoids),
                                      L.ptr(assign), L.ptr(iters), L.ptr(ws), ws.numel(), _st(x)), "cc_batch_kmedoids_f32")
    return assign, medoids, iters


@batch_kmedoids.register_fake
def _(x, K, metric, norm_p, threshold, iter_limit, id_sort, split_size, pre_norm):
    P, N, W = x.shape
    return (x.new_empty((P, N), dtype=torch.long), x.new_empty((P, K), dtype=torch.long),
            x.new_empty((P,), dtype=torch.int32))


@custom_op(NS + "::kmedoids_from_dist", mutates_args=(), device_types="cuda")
def kmedoids_from_dist(dist: torch.Tensor, norms: torch.Tensor, K: int, iter_limit: int,
                       id_sort: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    P, N, _ = dist.shape
    medoids = _e(P, K, like=dist, dtype=torch.long)
    assign = _e(P, N, like=dist, dtype=torch.long)
    iters = _e(P, like=dist, dtype=torch.int32)
    L.check(L.lib().cc_kmedoids_from_dist_f32(L.ptr(dist), L.ptr(norms), P, N, int(K), int(iter_limit), int(id_sort),
                                              L.ptr(medoids), L.ptr(assign), L.ptr(iters), None, 0, _st(dist)),
            "cc_kmedoids_from_dist_f32")
    return assign, medoids, iters


@kmedoids_from_dist.register_fake
def _(dist, norms, K, iter_limit, id_sort):
    P, N, _ = dist.shape
    return (dist.new_empty((P, N), dtype=torch.long), dist.new_empty((P, K), dtype=torch.long),
            dist.new_empty((P,), dtype=torch.int32))


@custom_op(NS + "::pairwise_distance", mutates_args=(), device_types="cuda")
def pairwise_distance(x: torch.Tensor, metric: int, p: float, all_negative: bool, self_nearest: bool) -> torch.Tensor:
    """pairwise_distance(x, x, ...) (modules/cluster/cluster_utils.py:8-43): x [P,N,W] -> [P,N,N]."""
    P, N, W = x.shape
    lay = L.TokenLayout(P, 1, 1, N, N * W, 0, 0, W)
    dist = _e(P, N, N, like=x, dtype=torch.float32)
    lib = L.lib()
    ws = L.workspace(lib.cc_cluster_workspace_bytes(P, N, W, 0), x.device)
    L.check(lib.cc_pairwise_distance_f32(L.ptr(x), ctypes.byref(lay), W, metric, float(p), int(all_negative),
                                         int(self_nearest), P, L.ptr(dist), None, L.ptr(ws), ws.numel(), _st(x)),
            "cc_pairwise_distance_f32")
    return dist


@pairwise_distance.register_fake
def _(x, metric, p, all_negative, self_nearest):
    P, N, W = x.shape
    return x.new_empty((P, N, N))


@custom_op(NS + "::pairwise_distance_cross", mutates_args=(), device_types="cuda")
def pairwise_distance_cross(x1: torch.Tensor, x2: torch.Tensor, metric: int, p: float, all_negative: bool,
                            self_nearest: bool) -> torch.Tensor:
    """pairwise_distance(data1, data2, ...) for two different sets (cluster_utils.py:8-43): [P,N1,W], [P,N2,W] -> [P,N1,N2]."""
    P, N1, W = x1.shape
    N2 = x2.shape[1]
    dist = _e(P, N1, N2, like=x1, dtype=torch.float32)
    ws = L.workspace(256, x1.device)
    L.check(L.lib().cc_pairwise_distance_cross_f32(L.ptr(x1), L.ptr(x2), P, N1, N2, W, metric, float(p), int(all_negative),
                                                   int(self_nearest), L.ptr(dist), L.ptr(ws), ws.numel(), _st(x1)),
            "cc_pairwise_distance_cross_f32")
    return dist


@pairwise_distance_cross.register_fake
def _(x1, x2, metric, p, all_negative, self_nearest):
    return x1.new_empty((x1.shape[0], x1.shape[1], x2.shape[1]))


@custom_op(NS + "::token_norms", mutates_args=(), device_types="cuda")
def token_norms(x: torch.Tensor) -> torch.Tensor:
    P, N, W = x.shape
    lay = L.TokenLayout(P, 1, 1, N, N * W, 0, 0, W)
    norms = _e(P, N, like=x, dtype=torch.float32)
    lib = L.lib()
    ws = L.workspace(lib.cc_cluster_workspace_bytes(P, N, W, 0), x.device)
    L.check(lib.cc_token_norms_f32(L.ptr(x), ctypes.byref(lay), W, L.ptr(norms), L.ptr(ws), ws.numel(), _st(x)),
            "cc_token_norms_f32")
    return norms


@token_norms.register_fake
def _(x):
    return x.new_empty(x.shape[:2])


@custom_op(NS + "::spectral_laplacian", mutates_args=(), device_types="cuda")
def spectral_laplacian(x: torch.Tensor, sigma: float, graph: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Normalised graph Laplacian of the heat-kernel affinity (modules/cluster/spectral.py:42-52,79-107): x [P,N,W] fp32,
    graph [N,N] uint8 or None -> (L_sym [P,N,N], W [P,N,N])."""
    P, N, Wd = x.shape
    lay = L.TokenLayout(P, 1, 1, N, N * Wd, 0, 0, Wd)
    lap = _e(P, N, N, like=x, dtype=torch.float32)
    aff = _e(P, N, N, like=x, dtype=torch.float32)
    lib = L.lib()
    ws = L.workspace(lib.cc_cluster_workspace_bytes(P, N, Wd, 0), x.device)
    L.check(lib.cc_spectral_laplacian_f32(L.ptr(x), ctypes.byref(lay), Wd, float(sigma), L.ptr(graph), L.ptr(lap), L.ptr(aff),
                                          None, L.ptr(ws), ws.numel(), _st(x)), "cc_spectral_laplacian_f32")
    return lap, aff


@spectral_laplacian.register_fake
def _(x, sigma, graph):
    P, N, _W = x.shape
    return x.new_empty((P, N, N)), x.new_empty((P, N, N))


@custom_op(NS + "::spectral_graph_laplacian", mutates_args=(), device_types="cuda")
def spectral_graph_laplacian(x: torch.Tensor, frame_major: bool, T: int, T_new: int, sigma: float, mode: int, knn_k: int,
                             mutual: bool, graph: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """constructW + normalised Laplacian (spectral.py:42-52,79-107) for the heat-kernel (mode 0) and the KNN graph (mode 1).
    x: [P,N,W] (T = T_new = 0), or the activations of TokenClusterInter ([1+n, B*T, W] LND / [B*T, 1+n, W] frame-major) whose
    patch tokens are regrouped into T_new segments per clip through strides (problem p = s*B + b, cluster.py:247-250).
    -> (L_sym [P,N,N], W [P,N,N])."""
    if T == 0:
        P, N, Wd = x.shape
        lay = L.TokenLayout(P, 1, 1, N, N * Wd, 0, 0, Wd)
        base = x
    else:
        BT, Lt, Wd, tok, frame = _cluster_strides(x.shape, frame_major)
        B, n, fd = BT // T, Lt - 1, T // T_new
        P, N = B * T_new, fd * n
        lay = L.TokenLayout(B, T_new, fd, n, T * frame, fd * frame, frame, tok)
        base = x.reshape(-1)[tok:]                       # first patch token of frame 0 (the CLS token is skipped)
    lap = _e(P, N, N, like=x, dtype=torch.float32)
    aff = _e(P, N, N, like=x, dtype=torch.float32)
    lib = L.lib()
    ws = L.workspace(lib.cc_cluster_workspace_bytes(P, N, Wd, 0), x.device)
    L.check(lib.cc_spectral_graph_laplacian_f32(L.ptr(base), ctypes.byref(lay), Wd, float(sigma), int(mode), int(knn_k),
                                                int(mutual), L.ptr(graph), L.ptr(lap), L.ptr(aff), None, L.ptr(ws), ws.numel(),
                                                _st(x)), "cc_spectral_graph_laplacian_f32")
    return lap, aff


@spectral_graph_laplacian.register_fake
def _(x, frame_major, T, T_new, sigma, mode, knn_k, mutual, graph):
    if T == 0:
        P, N, _W = x.shape
    else:
        BT, Lt, _W, _, _ = _cluster_strides(x.shape, frame_major)
        P, N = BT // T * T_new, (T // T_new) * (Lt - 1)
    return x.new_empty((P, N, N)), x.new_empty((P, N, N))


@custom_op(NS + "::spectral_embedding", mutates_args=(), device_types="cuda")
def spectral_embedding(laplacian: torch.Tensor, K: int, correct_sign: bool, solver: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The decomposition of batch_spectral_clustering (spectral.py:54-61, cc_spectral_embedding_solver_f32): laplacian [P,N,N]
    -> (Q [P,N,K4] with K4 = K rounded up to 4 and zero padding columns - the k-medoids kernels read 16-byte pieces -,
    eigenvalues [P,K] in the reference's order, sweeps [P]).  solver: 0 = CC_EIG_AUTO, 1 = CC_EIG_JACOBI."""
    P, N, _ = laplacian.shape
    K4 = (K + 3) // 4 * 4
    Q = _e(P, N, K4, like=laplacian, dtype=torch.float32)
    ev = _e(P, K, like=laplacian, dtype=torch.float32)
    sw = _e(P, like=laplacian, dtype=torch.int32)
    lib = L.lib()
    ws = L.workspace(lib.cc_spectral_embedding_workspace_bytes(P, N), laplacian.device)
    L.check(lib.cc_spectral_embedding_solver_f32(L.ptr(laplacian), P, N, int(K), int(correct_sign), L.ptr(Q), K4, L.ptr(ev),
                                                 L.ptr(sw), int(solver), L.ptr(ws), ws.numel(), _st(laplacian)),
            "cc_spectral_embedding_solver_f32")
    return Q, ev, sw


@spectral_embedding.register_fake
def _(laplacian, K, correct_sign, solver=0):
    P, N, _ = laplacian.shape
    return (laplacian.new_empty((P, N, (K + 3) // 4 * 4)), laplacian.new_empty((P, K)),
            laplacian.new_empty((P,), dtype=torch.int32))


@custom_op(NS + "::svd_sign_flip", mutates_args=(), device_types="cuda")
def svd_sign_flip(U: torch.Tensor, S: torch.Tensor, VT: torch.Tensor) -> torch.Tensor:
    """batch_sign_flip_rasmus_bro (spectral.py:110-137): U [P,M,K], S [P,K], VT [P,K,N] -> sign-corrected copy of U."""
    P, M, K = U.shape
    out = U.clone()
    L.check(L.lib().cc_svd_sign_flip_f32(L.ptr(out), L.ptr(S), L.ptr(VT), P, M, K, VT.shape[2], _st(U)),
            "cc_svd_sign_flip_f32")
    return out


@svd_sign_flip.register_fake
def _(U, S, VT):
    return torch.empty_like(U)


# ------------------------------------------------------------------------------------------ encoders
def _check_forced_medoids(lib, m, B: int, forced_medoids: Optional[torch.Tensor]) -> None:
    """The C entry points take forced_medoids as a bare pointer: the id tensors of ALL cluster blocks back to back.  A
    buffer of any other length (e.g. the ids of the last block only, as rounds 1-4 took them) would be read past its end."""
    if forced_medoids is None:
        return
    want = int(lib.cc_vit_forced_medoids_count(ctypes.byref(m), B))
    if (forced_medoids.dtype != torch.long or not forced_medoids.is_contiguous() or forced_medoids.numel() != want):
        raise ValueError(f"forced_medoids: {want} contiguous int64 ids expected (every cluster block's [B*T_new, K] ids "
                         f"back to back), got {forced_medoids.numel()} of {forced_medoids.dtype}")


@custom_op(NS + "::vit_encode", mutates_args=(), device_types="cuda")
def vit_encode(frames: torch.Tensor, handle: int, B: int, T: int, want_hidden: bool, want_medoids: bool,
               forced_medoids: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """CLIP.encode_image / VisualTransformer.forward (modules/clip.py:460-469,304-349): frames [B*T,3,H,W] fp32 or
    uint8 ([B*T,3,H,W] / [B*T,H,W,3]) -> (features [B*T_final, E], hidden [B*T_final, L_final, W] before ln_post or
    empty, medoids of the last k-medoids block or empty)."""
    from .clip import frames_descriptor
    m, meta, _keep = _model(handle)
    fr, frames = frames_descriptor(frames)
    frames_out, ltok, med_shape = meta["final"](T)
    lib = L.lib()
    feats = _e(B * frames_out, meta["embed_dim"], like=frames, dtype=torch.float32)
    hidden = _e(B * frames_out if want_hidden else 0, ltok, meta["width"], like=frames, dtype=torch.float32)
    med = torch.empty((B * med_shape[0], med_shape[1]) if (want_medoids and med_shape) else (0, 0), device=frames.device,
                      dtype=torch.long)
    _check_forced_medoids(lib, m, B, forced_medoids)
    ws = L.workspace(lib.cc_vit_workspace_bytes(ctypes.byref(m), B, T), frames.device)
    L.check(lib.cc_vit_encode_frames(ctypes.byref(m), ctypes.byref(fr), B, T, L.ptr(feats),
                                     L.ptr(hidden) if want_hidden else None, L.ptr(med) if med.numel() else None,
                                     L.ptr(forced_medoids), L.ptr(ws), ws.numel(), _st(frames)), "cc_vit_encode_frames")
    return feats, hidden, med


@vit_encode.register_fake
def _(frames, handle, B, T, want_hidden, want_medoids, forced_medoids):
    _m, meta, _keep = _model(handle)
    frames_out, ltok, med_shape = meta["final"](T)
    feats = frames.new_empty((B * frames_out, meta["embed_dim"]), dtype=torch.float32)
    hidden = frames.new_empty((B * frames_out if want_hidden else 0, ltok, meta["width"]), dtype=torch.float32)
    med = frames.new_empty((B * med_shape[0], med_shape[1]) if (want_medoids and med_shape) else (0, 0), dtype=torch.long)
    return feats, hidden, med


@custom_op(NS + "::text_encode", mutates_args=(), device_types="cuda")
def text_encode(ids: torch.Tensor, handle: int, want_hidden: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """CLIP.encode_text (modules/clip.py:471-496): ids [B, n_ctx] int64 -> (features [B, E], hidden [B, n_ctx, W]
    before ln_final, or empty)."""
    m, meta, _keep = _model(handle)
    Bt, Lt = ids.shape
    lib = L.lib()
    out = _e(Bt, meta["embed_dim"], like=ids, dtype=torch.float32)
    hidden = _e(Bt if want_hidden else 0, Lt, meta["width"], like=ids, dtype=torch.float32)
    ws = L.workspace(lib.cc_text_workspace_bytes(ctypes.byref(m), Bt, Lt), ids.device)
    L.check(lib.cc_text_encode_hidden(ctypes.byref(m), L.ptr(ids), Bt, Lt, L.ptr(out), L.ptr(hidden) if want_hidden else None,
                                      L.ptr(ws), ws.numel(), _st(ids)), "cc_text_encode_hidden")
    return out, hidden


@text_encode.register_fake
def _(ids, handle, want_hidden):
    _m, meta, _keep = _model(handle)
    Bt, Lt = ids.shape
    return (ids.new_empty((Bt, meta["embed_dim"]), dtype=torch.float32),
            ids.new_empty((Bt if want_hidden else 0, Lt, meta["width"]), dtype=torch.float32))


@custom_op(NS + "::clip_encode_out", mutates_args=("vfeat", "tfeat", "medoids_out"), device_types="cuda")
def clip_encode_out(frames: torch.Tensor, ids: torch.Tensor, vhandle: int, thandle: int, B: int, T: int,
                    vfeat: torch.Tensor, tfeat: torch.Tensor, medoids_out: Optional[torch.Tensor],
                    forced_medoids: Optional[torch.Tensor]) -> None:
    """Both towers of one CLIP4Clip.forward call in a single enqueue (modules/clip4clip.py:199-243); the features are
    written into caller-owned contiguous buffers (e.g. slices of the packed all-gather record).  medoids_out / forced_medoids:
    the ids of the last k-medoids block, reported / imposed (test hook, see cc_vit_encode)."""
    from .clip import frames_descriptor
    vm, vmeta, _k0 = _model(vhandle)
    tm, tmeta, _k1 = _model(thandle)
    fr, frames = frames_descriptor(frames)
    Bt, Lt = ids.shape
    lib = L.lib()
    _check_forced_medoids(lib, vm, B, forced_medoids)
    # (caller-owned outputs reach the library as bare pointers)
    frames_out, _ltok, med_shape = vmeta["final"](T)
    if (vfeat.dtype != torch.float32 or not vfeat.is_contiguous() or vfeat.numel() != B * frames_out * vmeta["embed_dim"]
            or tfeat.dtype != torch.float32 or not tfeat.is_contiguous() or tfeat.numel() != Bt * tmeta["embed_dim"]):
        raise ValueError("clip_encode_out: vfeat [B * %d, %d] and tfeat [%d, %d] contiguous fp32 expected"
                         % (frames_out, vmeta["embed_dim"], Bt, tmeta["embed_dim"]))
    if medoids_out is not None and (med_shape is None or medoids_out.dtype != torch.long or not medoids_out.is_contiguous()
                                    or medoids_out.numel() != B * med_shape[0] * med_shape[1]):
        raise ValueError("clip_encode_out: medoids_out = the last k-medoids block's [B * T_new, K] int64 ids")
    ws = L.workspace(lib.cc_clip_workspace_bytes(ctypes.byref(vm), B, T, ctypes.byref(tm), Bt, Lt), frames.device)
    L.check(lib.cc_clip_encode_frames(ctypes.byref(vm), ctypes.byref(fr), B, T, L.ptr(vfeat), L.ptr(medoids_out),
                                      L.ptr(forced_medoids), ctypes.byref(tm), L.ptr(ids), Bt, Lt, L.ptr(tfeat), L.ptr(ws),
                                      ws.numel(), _st(frames)), "cc_clip_encode_frames")


@clip_encode_out.register_fake
def _(frames, ids, vhandle, thandle, B, T, vfeat, tfeat, medoids_out, forced_medoids):
    return None


@custom_op(NS + "::clip_encode", mutates_args=(), device_types="cuda")
def clip_encode(frames: torch.Tensor, ids: torch.Tensor, vhandle: int, thandle: int, B: int,
                T: int) -> Tuple[torch.Tensor, torch.Tensor]:
    _vm, vmeta, _k0 = _model(vhandle)
    frames_out, _ltok, _med = vmeta["final"](T)
    vfeat = torch.empty(B * frames_out, vmeta["embed_dim"], device=frames.device, dtype=torch.float32)
    tfeat = torch.empty(ids.shape[0], vmeta["embed_dim"], device=frames.device, dtype=torch.float32)
    torch.ops.centerclip.clip_encode_out(frames, ids, vhandle, thandle, B, T, vfeat, tfeat, None, None)
    return vfeat, tfeat


@clip_encode.register_fake
def _(frames, ids, vhandle, thandle, B, T):
    _vm, vmeta, _k0 = _model(vhandle)
    frames_out, _ltok, _med = vmeta["final"](T)
    return (frames.new_empty((B * frames_out, vmeta["embed_dim"]), dtype=torch.float32),
            frames.new_empty((ids.shape[0], vmeta["embed_dim"]), dtype=torch.float32))


# ------------------------------------------------------------------------------------------ similarity tail / metrics
@custom_op(NS + "::loose_similarity", mutates_args=(), device_types="cuda")
def loose_similarity(text: torch.Tensor, visual: torch.Tensor, video_mask: torch.Tensor, logit_scale: float, group: int,
                     vis_group_stride: int, mask_group_stride: int, Bv: int, Tn: int,
                     want_pooled: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """CLIP4Clip._loose_similarity, meanP (modules/clip4clip.py:305-316,357-366): text [Bt,E] fp32, visual fp32 holding
    Bv videos of Tn frames (plain [Bv,Tn,E], or records of `group` videos every vis_group_stride floats), video_mask
    int64 (a strided [Bv,Tn] view, or records every mask_group_stride elements) -> (logits [Bt,Bv], pooled [Bv,E] or
    empty)."""
    Bt, E = text.shape
    lib = L.lib()
    logits = _e(Bt, Bv, like=text, dtype=torch.float32)
    pooled = _e(Bv if want_pooled else 0, E, like=text, dtype=torch.float32)
    ws = L.workspace(lib.cc_similarity_workspace_bytes(Bt, Bv, E), text.device)
    if video_mask.dim() == 2:
        mrs, mcs = video_mask.stride(0), video_mask.stride(1)
    else:                                    # flat record buffer: rows of Tn mask entries inside each record
        mrs, mcs = Tn, 1
    L.check(lib.cc_loose_similarity_grouped_f32(L.ptr(text), L.ptr(visual), L.ptr(video_mask), int(group),
                                                int(vis_group_stride), int(mask_group_stride), mrs, mcs, Bt, Bv, Tn, E,
                                                float(logit_scale), L.ptr(logits), Bv,
                                                L.ptr(pooled) if want_pooled else None, L.ptr(ws), ws.numel(), _st(text)),
            "cc_loose_similarity_grouped_f32")
    return logits, pooled


@loose_similarity.register_fake
def _(text, visual, video_mask, logit_scale, group, vis_group_stride, mask_group_stride, Bv, Tn, want_pooled):
    Bt, E = text.shape
    return text.new_empty((Bt, Bv)), text.new_empty((Bv if want_pooled else 0, E))


@custom_op(NS + "::video_pool_normalize", mutates_args=(), device_types="cuda")
def video_pool_normalize(visual: torch.Tensor, video_mask: torch.Tensor) -> torch.Tensor:
    Bv, Tn, E = visual.shape
    pooled = _e(Bv, E, like=visual, dtype=torch.float32)
    L.check(L.lib().cc_video_pool_normalize_f32(L.ptr(visual), L.ptr(video_mask), Bv, Tn, E, L.ptr(pooled), _st(visual)),
            "cc_video_pool_normalize_f32")
    return pooled


@video_pool_normalize.register_fake
def _(visual, video_mask):
    return visual.new_empty((visual.shape[0], visual.shape[2]))


@custom_op(NS + "::normalize_rows", mutates_args=(), device_types="cuda")
def normalize_rows(x: torch.Tensor) -> torch.Tensor:
    R, E = x.shape
    out = torch.empty_like(x)
    L.check(L.lib().cc_normalize_rows_f32(L.ptr(x), L.ptr(out), R, E, _st(x)), "cc_normalize_rows_f32")
    return out


@normalize_rows.register_fake
def _(x):
    return torch.empty_like(x)


@custom_op(NS + "::normalize_rows_planes", mutates_args=(), device_types="cuda")
def normalize_rows_planes(x: torch.Tensor, video_side: bool) -> torch.Tensor:
    """rows / |row| written as the split-fp16 planes the similarity GEMM multiplies: [R, 3E] fp16 (text side [hi|hi|lo],
    video side [hi|lo|hi]) - the operand of scaled_dot_planes, produced when a batch is encoded."""
    R, E = x.shape
    planes = _e(R, 3 * E, like=x, dtype=torch.float16)
    L.check(L.lib().cc_normalize_rows_planes_f32(L.ptr(x), None, L.ptr(planes), int(video_side), R, E, _st(x)),
            "cc_normalize_rows_planes_f32")
    return planes


@normalize_rows_planes.register_fake
def _(x, video_side):
    return x.new_empty((x.shape[0], 3 * x.shape[1]), dtype=torch.float16)


@custom_op(NS + "::video_pool_normalize_planes", mutates_args=(), device_types="cuda")
def video_pool_normalize_planes(visual: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """_mean_pooling_for_similarity_visual + the final normalisation (clip4clip.py:305-316,357-360) -> video-side planes
    [Bv, 3E] fp16."""
    Bv, Tn, E = visual.shape
    planes = _e(Bv, 3 * E, like=visual, dtype=torch.float16)
    L.check(L.lib().cc_video_pool_normalize_planes_f32(L.ptr(visual), L.ptr(mask), Bv, Tn, E, None, L.ptr(planes), _st(visual)),
            "cc_video_pool_normalize_planes_f32")
    return planes


@video_pool_normalize_planes.register_fake
def _(visual, mask):
    return visual.new_empty((visual.shape[0], 3 * visual.shape[2]), dtype=torch.float16)


@custom_op(NS + "::scaled_dot_planes", mutates_args=(), device_types="cuda")
def scaled_dot_planes(text_planes: torch.Tensor, video_planes: torch.Tensor, n_video: int, mult: float,
                      products: int = 3) -> torch.Tensor:
    """[Bt, n_video] = mult * text . video^T from the planes: the GEMM launch alone.  video_planes holds at least
    padded_video_rows(n_video) rows, zeros behind n_video.  products: 3 (default: both operands to 22 bits), 2 (the text
    side rounded to fp16) or 1 (both sides fp16) of the three fp16 products per multiply-add - see cc_scaled_dot_planes_products_f32."""
    Bt, E3 = text_planes.shape
    out = _e(Bt, int(n_video), like=text_planes, dtype=torch.float32)
    L.check(L.lib().cc_scaled_dot_planes_products_f32(L.ptr(text_planes), L.ptr(video_planes), Bt, int(n_video),
                                                      video_planes.shape[0], E3 // 3, float(mult), int(products), L.ptr(out),
                                                      int(n_video), _st(text_planes)), "cc_scaled_dot_planes_products_f32")
    return out


@scaled_dot_planes.register_fake
def _(text_planes, video_planes, n_video, mult, products=3):
    return text_planes.new_empty((text_planes.shape[0], n_video), dtype=torch.float32)


def padded_video_rows(n_video):
    """Rows a video-side plane buffer needs for n_video videos (whole GEMM tiles; the rows behind n_video must be zero)."""
    return int(L.lib().cc_similarity_padded_rows(int(n_video)))


@custom_op(NS + "::scaled_dot_nt", mutates_args=(), device_types="cuda")
def scaled_dot_nt(a: torch.Tensor, b: torch.Tensor, mult: float) -> torch.Tensor:
    Bt, E = a.shape
    Bv = b.shape[0]
    out = _e(Bt, Bv, like=a, dtype=torch.float32)
    lib = L.lib()
    ws = L.workspace(lib.cc_similarity_workspace_bytes(Bt, Bv, E), a.device)
    L.check(lib.cc_scaled_dot_nt_f32(L.ptr(a), L.ptr(b), Bt, Bv, E, float(mult), L.ptr(out), Bv, L.ptr(ws), ws.numel(),
                                     _st(a)), "cc_scaled_dot_nt_f32")
    return out


@scaled_dot_nt.register_fake
def _(a, b, mult):
    return a.new_empty((a.shape[0], b.shape[0]))


@custom_op(NS + "::scaled_dot_nt_out", mutates_args=("out",), device_types="cuda")
def scaled_dot_nt_out(a: torch.Tensor, b: torch.Tensor, mult: float, out: torch.Tensor) -> None:
    Bt, E = a.shape
    lib = L.lib()
    ws = L.workspace(lib.cc_similarity_workspace_bytes(Bt, b.shape[0], E), a.device)
    L.check(lib.cc_scaled_dot_nt_f32(L.ptr(a), L.ptr(b), Bt, b.shape[0], E, float(mult), L.ptr(out), out.stride(0),
                                     L.ptr(ws), ws.numel(), _st(a)), "cc_scaled_dot_nt_f32")


@scaled_dot_nt_out.register_fake
def _(a, b, mult, out):
    return None


@custom_op(NS + "::rank_counts", mutates_args=(), device_types="cuda")
def rank_counts(sim: torch.Tensor, transpose: bool, diag_offset: int) -> torch.Tensor:
    """utils/metrics.py:11-26 without the sort: per row (#greater, #equal) than the ground-truth entry."""
    rows, cols = (sim.shape[1], sim.shape[0]) if transpose else sim.shape
    rs, cs = (sim.stride(1), sim.stride(0)) if transpose else (sim.stride(0), sim.stride(1))
    counts = _e(rows, 2, like=sim, dtype=torch.int32)
    L.check(L.lib().cc_rank_counts_f32(L.ptr(sim), rows, cols, rs, cs, int(diag_offset), L.ptr(counts), _st(sim)),
            "cc_rank_counts_f32")
    return counts


@rank_counts.register_fake
def _(sim, transpose, diag_offset):
    return sim.new_empty((sim.shape[1] if transpose else sim.shape[0], 2), dtype=torch.int32)


@custom_op(NS + "::rank_counts_cols", mutates_args=(), device_types="cuda")
def rank_counts_cols(sim: torch.Tensor, gt_cols: torch.Tensor) -> torch.Tensor:
    """utils/metrics.py:38-65: sim [R, C] contiguous, gt_cols [R] int32 -> (#greater, #equal, #equal before) per row."""
    R, C = sim.shape
    counts = _e(R, 3, like=sim, dtype=torch.int32)
    L.check(L.lib().cc_rank_counts_cols_f32(L.ptr(sim), R, C, C, 1, L.ptr(gt_cols), L.ptr(counts), _st(sim)),
            "cc_rank_counts_cols_f32")
    return counts


@rank_counts_cols.register_fake
def _(sim, gt_cols):
    return sim.new_empty((sim.shape[0], 3), dtype=torch.int32)


@custom_op(NS + "::contrastive_loss_grad", mutates_args=(), device_types="cuda")
def contrastive_loss_grad(text: torch.Tensor, visual: torch.Tensor, video_mask: torch.Tensor,
                          logit_scale: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """The training branch's loss with its gradient (clip4clip.py:245-262,357-366; losses.py:8-18): text [n,E], visual
    [n,Tn,E], video_mask [n,Tn] int64 -> (loss3 [3], d_text, d_visual, d_logit_scale [1]) for d loss3[2] = 1."""
    n, Tn, E = visual.shape
    loss3 = _e(3, like=text, dtype=torch.float32)
    d_text, d_visual = torch.empty_like(text), torch.empty_like(visual)
    d_ls = _e(1, like=text, dtype=torch.float32)
    ws = L.workspace(L.lib().cc_contrastive_grad_workspace_bytes(n, Tn, E), text.device)
    L.check(L.lib().cc_contrastive_loss_grad_f32(L.ptr(text), L.ptr(visual), L.ptr(video_mask), video_mask.stride(0),
                                                 video_mask.stride(1), n, Tn, E, float(logit_scale), 1.0, L.ptr(loss3),
                                                 L.ptr(d_text), L.ptr(d_visual), L.ptr(d_ls), L.ptr(ws), ws.numel(), _st(text)),
            "cc_contrastive_loss_grad_f32")
    return loss3, d_text, d_visual, d_ls


@contrastive_loss_grad.register_fake
def _(text, visual, video_mask, logit_scale):
    return text.new_empty((3,)), torch.empty_like(text), torch.empty_like(visual), text.new_empty((1,))


@custom_op(NS + "::contrastive_loss_grad_dev", mutates_args=(), device_types="cuda")
def contrastive_loss_grad_dev(text: torch.Tensor, visual: torch.Tensor, video_mask: torch.Tensor,
                              logit_scale: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """contrastive_loss_grad with logit_scale read from the device (a 0-d / 1-element fp32 tensor: the parameter itself)."""
    n, Tn, E = visual.shape
    loss3 = _e(3, like=text, dtype=torch.float32)
    d_text, d_visual = torch.empty_like(text), torch.empty_like(visual)
    d_ls = _e(1, like=text, dtype=torch.float32)
    ws = L.workspace(L.lib().cc_contrastive_grad_workspace_bytes(n, Tn, E), text.device)
    L.check(L.lib().cc_contrastive_loss_grad_dev_f32(L.ptr(text), L.ptr(visual), L.ptr(video_mask), video_mask.stride(0),
                                                     video_mask.stride(1), n, Tn, E, 0.0, L.ptr(logit_scale), 1.0, L.ptr(loss3),
                                                     L.ptr(d_text), L.ptr(d_visual), L.ptr(d_ls), L.ptr(ws), ws.numel(), _st(text)),
            "cc_contrastive_loss_grad_dev_f32")
    return loss3, d_text, d_visual, d_ls


@contrastive_loss_grad_dev.register_fake
def _(text, visual, video_mask, logit_scale):
    return text.new_empty((3,)), torch.empty_like(text), torch.empty_like(visual), text.new_empty((1,))


@custom_op(NS + "::rank_counts_ref", mutates_args=(), device_types="cuda")
def rank_counts_ref(sim: torch.Tensor, ref_vals: torch.Tensor, transpose: bool) -> torch.Tensor:
    """(#greater, #equal) than ref_vals[i] per row of sim (per COLUMN with transpose=True, through the strides): the
    partial video->text counts of a row block of the similarity matrix (clip-sharded eval)."""
    rows, cols = (sim.shape[1], sim.shape[0]) if transpose else sim.shape
    rs, cs = (sim.stride(1), sim.stride(0)) if transpose else (sim.stride(0), sim.stride(1))
    counts = _e(rows, 2, like=sim, dtype=torch.int32)
    L.check(L.lib().cc_rank_counts_ref_f32(L.ptr(sim), rows, cols, rs, cs, L.ptr(ref_vals), L.ptr(counts), _st(sim)),
            "cc_rank_counts_ref_f32")
    return counts


@rank_counts_ref.register_fake
def _(sim, ref_vals, transpose):
    return sim.new_empty((sim.shape[1] if transpose else sim.shape[0], 2), dtype=torch.int32)


@custom_op(NS + "::group_max_rows", mutates_args=(), device_types="cuda")
def group_max_rows(sim: torch.Tensor, group: torch.Tensor, n_groups: int) -> torch.Tensor:
    """[n_groups, cols] maxima of sim's rows per group id (int32 [rows]), NaN ignored, -inf for groups without a row:
    tensor_video_to_text_sim (utils/metrics.py:68-76)."""
    rows, cols = sim.shape
    out = _e(int(n_groups), cols, like=sim, dtype=torch.float32)
    L.check(L.lib().cc_group_max_rows_f32(L.ptr(sim) if rows else None, rows, cols, sim.stride(0) if rows else cols,
                                          L.ptr(group) if rows else None, int(n_groups), L.ptr(out), _st(sim)),
            "cc_group_max_rows_f32")
    return out


@group_max_rows.register_fake
def _(sim, group, n_groups):
    return sim.new_empty((n_groups, sim.shape[1]))


@custom_op(NS + "::contrastive_loss", mutates_args=(), device_types="cuda")
def contrastive_loss(sim: torch.Tensor) -> torch.Tensor:
    """CrossEn(sim), CrossEn(sim.T) and their mean (modules/losses.py:8-18, clip4clip.py:250-253) -> [3] fp32."""
    n = sim.shape[0]
    out = _e(3, like=sim, dtype=torch.float32)
    ws = L.workspace(2 * n * 4, sim.device)
    L.check(L.lib().cc_contrastive_loss_f32(L.ptr(sim), n, sim.stride(0), sim.stride(1), L.ptr(out), L.ptr(ws), ws.numel(),
                                            _st(sim)), "cc_contrastive_loss_f32")
    return out


@contrastive_loss.register_fake
def _(sim):
    return sim.new_empty((3,))


OPS = ("contrastive_loss", "contrastive_loss_grad", "contrastive_loss_grad_dev", "spectral_laplacian", "spectral_graph_laplacian", "spectral_embedding", "svd_sign_flip", "linear_f16", "linear_f16_out", "layernorm", "attention_f16", "fold_layernorm_linear", "row_stats",
       "linear_ln_f16", "inproj_attention_f16", "linear_resid_stats_f16", "head_project", "token_cluster", "token_cluster_train", "token_cluster_backward", "token_apply_selection",
       "batch_kmedoids", "kmedoids_from_dist",
       "pairwise_distance", "pairwise_distance_cross", "token_norms", "vit_encode", "text_encode", "clip_encode_out", "clip_encode",
       "loose_similarity", "video_pool_normalize", "normalize_rows", "scaled_dot_nt", "scaled_dot_nt_out", "rank_counts",
       "rank_counts_cols", "rank_counts_ref", "group_max_rows", "normalize_rows_planes", "video_pool_normalize_planes",
       "scaled_dot_planes")


def logit_multiplier(logit_scale):
    """exp(logit_scale) on the host (clip4clip.py:357-358 computes it with a device op; the value is a parameter)."""
    return math.exp(float(logit_scale))
