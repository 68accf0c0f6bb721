"""Thin tensor-level wrappers over the single-op C entry points (used by the module mirrors and
by the op-level parity tests).  Every function enqueues on the current torch stream."""
import torch

from . import _lib as L
from ._lib_clip import EPI


def linear_f16(a, w, bias, epilogue="f16", out=None, tile=0):
    """y = a @ w.T + bias with a fused epilogue; a [M,K] fp16, w [N,K] fp16, bias [N] fp32|None.
    epilogue: 'f16' | 'f16_gelu' (QuickGELU) | 'f32' | 'f32_resid' (out += ..., out required)."""
    L.require_device(a, w, bias, out)
    assert a.dtype == torch.float16 and w.dtype == torch.float16 and a.is_contiguous() and w.is_contiguous()
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        assert epilogue != "f32_resid"
        out = torch.empty(M, N, device=a.device, dtype=torch.float16 if epilogue.startswith("f16") else torch.float32)
    L.check(L.lib().cc_linear_f16(L.ptr(a), L.ptr(w), L.ptr(bias), L.ptr(out), M, N, K, out.stride(0), EPI[epilogue],
                                  tile, L.stream_ptr(a.device)), "cc_linear_f16")
    return out


def layernorm(x, weight, bias, eps=1e-5, out_f16=False):
    """LayerNorm over the last dim of a contiguous fp32 [..., W] tensor (statistics in fp32)."""
    L.require_device(x, weight, bias)
    assert x.dtype == torch.float32 and x.is_contiguous()
    W = x.shape[-1]
    rows = x.numel() // W
    out = torch.empty(x.shape, device=x.device, dtype=torch.float16 if out_f16 else torch.float32)
    L.check(L.lib().cc_layernorm_f32(L.ptr(x), W, L.ptr(weight), L.ptr(bias), L.ptr(out), W, rows, W, float(eps),
                                     int(out_f16), L.stream_ptr(x.device)), "cc_layernorm_f32")
    return out


def attention_f16(qkv, nseq, L_tok, heads, causal=False):
    """qkv [nseq*L, 3W] fp16 -> [nseq*L, W] fp16 (softmax(q k^T / 8 [+ causal mask]) v per head)."""
    L.require_device(qkv)
    assert qkv.dtype == torch.float16 and qkv.is_contiguous()
    W = qkv.shape[1] // 3
    out = torch.empty(nseq * L_tok, W, device=qkv.device, dtype=torch.float16)
    L.check(L.lib().cc_attention_f16(L.ptr(qkv), L.ptr(out), nseq, L_tok, heads, W, int(causal),
                                     L.stream_ptr(qkv.device)), "cc_attention_f16")
    return out


def loose_similarity(text, visual, video_mask, logit_scale, return_pooled=False):
    """meanP retrieval logits [Bt,Bv] (CLIP4Clip._loose_similarity, eval branch)."""
    L.require_device(text, visual, video_mask)
    text = text.float().contiguous()
    visual = visual.float().contiguous()
    mask = video_mask.to(torch.long)          # a strided int64 view (every fd-th frame of the frame mask) is used as is
    Bt, E = text.shape
    Bv, Tn, _ = visual.shape
    assert mask.shape == (Bv, Tn)
    lib = L.lib()
    logits = torch.empty(Bt, Bv, device=text.device, dtype=torch.float32)
    pooled = torch.empty(Bv, E, device=text.device, dtype=torch.float32) if return_pooled else None
    ws = L.workspace(lib.cc_similarity_workspace_bytes(Bt, Bv, E), text.device)
    L.check(lib.cc_loose_similarity_strided_f32(L.ptr(text), L.ptr(visual), L.ptr(mask), mask.stride(0), mask.stride(1),
                                                Bt, Bv, Tn, E, float(logit_scale), L.ptr(logits), Bv, L.ptr(pooled),
                                                L.ptr(ws), ws.numel(), L.stream_ptr(text.device)),
            "cc_loose_similarity_strided_f32")
    return (logits, pooled) if return_pooled else logits


def video_pool_normalize(visual, video_mask):
    L.require_device(visual, video_mask)
    visual = visual.float().contiguous()
    mask = video_mask.to(torch.long).contiguous()
    Bv, Tn, E = visual.shape
    pooled = torch.empty(Bv, E, device=visual.device, dtype=torch.float32)
    L.check(L.lib().cc_video_pool_normalize_f32(L.ptr(visual), L.ptr(mask), Bv, Tn, E, L.ptr(pooled),
                                                L.stream_ptr(visual.device)), "cc_video_pool_normalize_f32")
    return pooled


def scaled_dot_nt(a, b, mult=1.0, out=None):
    """out[Bt,Bv] = mult * a @ b.T in exact fp32 (rows already normalised)."""
    L.require_device(a, b, out)
    a, b = a.float().contiguous(), b.float().contiguous()
    Bt, E = a.shape
    Bv = b.shape[0]
    if out is None:
        out = torch.empty(Bt, Bv, device=a.device, dtype=torch.float32)
    L.check(L.lib().cc_scaled_dot_nt_f32(L.ptr(a), L.ptr(b), Bt, Bv, E, float(mult), L.ptr(out), out.stride(0),
                                         L.stream_ptr(a.device)), "cc_scaled_dot_nt_f32")
    return out


LN_MAX_SLOTS = 32


def fold_layernorm_linear(weight, bias, gamma, beta):
    """-> (w' fp16 [N,K], c1 [N], c2 [N]): LN(x) W^T + b == rstd (x w'^T - mu c1) + c2."""
    L.require_device(weight, gamma, beta)
    w = weight.detach().float().contiguous()
    N, K = w.shape
    b = bias.detach().float().contiguous() if bias is not None else None
    wf = torch.empty(N, K, device=w.device, dtype=torch.float16)
    c1 = torch.empty(N, device=w.device, dtype=torch.float32)
    c2 = torch.empty(N, device=w.device, dtype=torch.float32)
    L.check(L.lib().cc_fold_layernorm_linear_f32(L.ptr(w), L.ptr(b), L.ptr(gamma.detach().float().contiguous()),
                                                 L.ptr(beta.detach().float().contiguous()), N, K, L.ptr(wf), L.ptr(c1),
                                                 L.ptr(c2), L.stream_ptr(w.device)), "cc_fold_layernorm_linear_f32")
    return wf, c1, c2


def row_stats(h):
    """h [M,W] fp32 -> (h16, stats [M, 1, 2]: one slot per row)."""
    L.require_device(h)
    M, W = h.shape
    h16 = torch.empty(M, W, device=h.device, dtype=torch.float16)
    compact = torch.empty(M, 2, device=h.device, dtype=torch.float32)
    L.check(L.lib().cc_row_stats_f16(L.ptr(h.contiguous()), L.ptr(h16), L.ptr(compact), M, W, L.stream_ptr(h.device)),
            "cc_row_stats_f16")
    return h16, compact


def linear_ln_f16(h16, w_ln, c1, c2, stats, slots, gelu=False, eps=1e-5, out=None, tile=0):
    """LayerNorm-folded Linear (stats laid out [M, slots, 2])."""
    M, K = h16.shape
    N = w_ln.shape[0]
    if out is None:
        out = torch.empty(M, N, device=h16.device, dtype=torch.float16)
    L.check(L.lib().cc_linear_ln_f16(L.ptr(h16), L.ptr(w_ln), L.ptr(c1), L.ptr(c2), L.ptr(stats), int(slots), float(eps),
                                     L.ptr(out), M, N, K, int(gelu), tile, L.stream_ptr(h16.device)), "cc_linear_ln_f16")
    return out


def linear_resid_stats_f16(a, w, bias, h, tile=0, h16=None, stats=None):
    """h += a @ w.T + bias in place; returns (h16, stats [M, slots, 2], slots).  ``h16`` [M,N] fp16 and ``stats``
    (flat fp32, >= M*32*2) may be preallocated."""
    import ctypes
    M, K = a.shape
    N = w.shape[0]
    if h16 is None:
        h16 = torch.empty(M, N, device=a.device, dtype=torch.float16)
    if stats is None:
        stats = torch.empty(M * LN_MAX_SLOTS * 2, device=a.device, dtype=torch.float32)
    slots = ctypes.c_int32(0)
    L.check(L.lib().cc_linear_resid_stats_f16(L.ptr(a), L.ptr(w), L.ptr(bias), L.ptr(h), L.ptr(h16), L.ptr(stats),
                                              ctypes.byref(slots), M, N, K, tile, L.stream_ptr(a.device)),
            "cc_linear_resid_stats_f16")
    return h16, stats[:M * slots.value * 2].view(M, slots.value, 2), slots.value
