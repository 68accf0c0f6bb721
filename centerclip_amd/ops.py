"""Tensor-level wrappers over the ``torch.ops.centerclip`` custom ops (torch_ops.py) - input checks and dtype /
contiguity normalisation only.  Used by the module mirrors and by the op-level parity tests.  Every function enqueues on
the current torch stream; there is no CPU path (a CPU tensor raises CenterClipHipError)."""
import torch

from . import _lib as L
from . import torch_ops as T          # noqa: F401  (registers the ops)

_ops = torch.ops.centerclip
LN_MAX_SLOTS = T.LN_MAX_SLOTS


def _need(cond, what):
    """The kernels behind these wrappers take bare pointers: a tensor of the wrong dtype / shape / layout would be read or
    written past its end on the device, so it is refused here."""
    if not cond:
        raise ValueError("centerclip_amd.ops: " + what)


def _is(t, dtype, shape=None):
    return (t.dtype == dtype and t.is_contiguous() and (shape is None or tuple(t.shape) == tuple(shape)))


def linear_f16(a, w, bias, epilogue="f16", out=None, tile=0):
    """y = a @ w.T + bias with a fused epilogue; a [M,K] fp16, w [N,K] fp16, bias [N] fp32|None.
    epilogue: 'f16' | 'f16_gelu' (QuickGELU) | 'f32' | 'f32_resid' (out += ..., out required)."""
    L.require_device(a, w, bias, out)
    _need(a.dim() == 2 and w.dim() == 2 and _is(a, torch.float16) and _is(w, torch.float16, (w.shape[0], a.shape[1])),
          "linear_f16: a [M, K] and w [N, K] contiguous fp16")
    _need(bias is None or _is(bias, torch.float32, (w.shape[0],)), "linear_f16: bias [N] contiguous fp32")
    _need(epilogue in T.EPI, "linear_f16: epilogue %r" % (epilogue,))
    if out is None:
        _need(epilogue != "f32_resid", "linear_f16: epilogue 'f32_resid' accumulates into out")
        return _ops.linear_f16(a, w, bias, epilogue, tile)
    _need(out.dim() == 2 and tuple(out.shape) == (a.shape[0], w.shape[0]) and out.stride(1) == 1 and out.stride(0) >= w.shape[0]
          and out.dtype == (torch.float16 if epilogue.startswith("f16") else torch.float32),
          "linear_f16: out [M, N] (unit column stride) of the epilogue's dtype")
    _ops.linear_f16_out(a, w, bias, out, epilogue, tile)
    return out


def layernorm(x, weight, bias, eps=1e-5, out_f16=False):
    """LayerNorm over the last dim of a contiguous fp32 [..., W] tensor (statistics in fp32)."""
    L.require_device(x, weight, bias)
    assert x.dtype == torch.float32 and x.is_contiguous()
    return _ops.layernorm(x, weight, bias, float(eps), bool(out_f16))


def attention_f16(qkv, nseq, L_tok, heads, causal=False, seq_rows=None, tok_rows=1):
    """qkv [nseq*L, 3W] fp16 -> [nseq*L, W] fp16 (softmax(q k^T / 8 [+ causal mask]) v per head).  Token t of sequence
    s is row s*seq_rows + t*tok_rows (default: frame-major, seq_rows = L; LND activations: seq_rows = 1, tok_rows = nseq)."""
    L.require_device(qkv)
    assert qkv.dtype == torch.float16 and qkv.is_contiguous()
    return _ops.attention_f16(qkv, nseq, L_tok, heads, bool(causal), L_tok if seq_rows is None else int(seq_rows),
                              int(tok_rows))


def loose_similarity(text, visual, video_mask, logit_scale, return_pooled=False):
    """meanP retrieval logits [Bt,Bv] (CLIP4Clip._loose_similarity, eval branch)."""
    L.require_device(text, visual, video_mask)
    text = text.float().contiguous()
    visual = visual.float().contiguous()
    mask = video_mask.to(torch.long)          # a strided int64 view (every fd-th frame of the frame mask) is used as is
    Bv, Tn, _ = visual.shape
    assert mask.shape == (Bv, Tn)
    logits, pooled = _ops.loose_similarity(text, visual, mask, float(logit_scale), Bv, 0, 0, Bv, Tn, bool(return_pooled))
    return (logits, pooled) if return_pooled else logits


def loose_similarity_packed(text, records, B, Tn, E, vis_off, mask_off, logit_scale):
    """The same on a packed all-gather buffer: ``records`` [G, rec] uint8 holds, per rank, B videos of Tn frames (fp32,
    at byte offset vis_off) and their [B, Tn] int64 mask (at byte offset mask_off) -> logits [Bt, G*B]."""
    L.require_device(text, records)
    G, rec = records.shape
    assert rec % 8 == 0 and vis_off % 4 == 0 and mask_off % 8 == 0 and records.is_contiguous()
    vis = records.view(-1)[vis_off:].view(torch.float32)
    msk = records.view(-1)[mask_off:].view(torch.long)
    logits, _ = _ops.loose_similarity(text.float().contiguous(), vis, msk, float(logit_scale), B, rec // 4, rec // 8,
                                      G * B, Tn, False)
    return logits


def video_pool_normalize(visual, video_mask):
    L.require_device(visual, video_mask)
    return _ops.video_pool_normalize(visual.float().contiguous(), video_mask.to(torch.long).contiguous())


def normalize_rows(x):
    """rows / |row| in fp32 (the text half of _loose_similarity)."""
    L.require_device(x)
    return _ops.normalize_rows(x.float().contiguous())


def scaled_dot_nt(a, b, mult=1.0, out=None):
    """out[Bt,Bv] = mult * a @ b.T in exact fp32 (rows already normalised)."""
    L.require_device(a, b, out)
    a, b = a.float().contiguous(), b.float().contiguous()
    if out is None:
        return _ops.scaled_dot_nt(a, b, float(mult))
    _ops.scaled_dot_nt_out(a, b, float(mult), out)
    return out


def fold_layernorm_linear(weight, bias, gamma, beta):
    """-> (w' fp16 [N,K], c1 [N], c2 [N]): LN(x) W^T + b == rstd (x w'^T - mu c1) + c2."""
    L.require_device(weight, gamma, beta)
    b = bias.detach().float().contiguous() if bias is not None else None
    return _ops.fold_layernorm_linear(weight.detach().float().contiguous(), b, gamma.detach().float().contiguous(),
                                      beta.detach().float().contiguous())


def row_stats(h, centre=True):
    """h [M,W] fp32 -> (h16 = fp16(h - row mean), stats [M, 2] (one slot per row), shift [M] = the row means);
    centre=False: h16 = fp16(h), shift is empty."""
    L.require_device(h)
    return _ops.row_stats(h.contiguous(), bool(centre))


def linear_ln_f16(h16, w_ln, c1, c2, stats, slots, gelu=False, eps=1e-5, out=None, tile=0):
    """LayerNorm-folded Linear (stats laid out [M, slots, 2])."""
    L.require_device(h16, w_ln, c1, c2, stats)
    _need(h16.dim() == 2 and w_ln.dim() == 2 and _is(h16, torch.float16) and _is(w_ln, torch.float16, (w_ln.shape[0], h16.shape[1])),
          "linear_ln_f16: h16 [M, K] and w_ln [N, K] contiguous fp16")
    _need(_is(c1, torch.float32, (w_ln.shape[0],)) and _is(c2, torch.float32, (w_ln.shape[0],)), "linear_ln_f16: c1, c2 [N] fp32")
    _need(1 <= int(slots) <= LN_MAX_SLOTS and stats.dtype == torch.float32 and stats.is_contiguous()
          and stats.numel() >= h16.shape[0] * int(slots) * 2, "linear_ln_f16: stats [M, slots, 2] fp32, 1 <= slots <= %d" % LN_MAX_SLOTS)
    y = _ops.linear_ln_f16(h16, w_ln, c1, c2, stats, int(slots), bool(gelu), float(eps), tile)
    if out is not None:
        out.copy_(y)
        return out
    return y


def inproj_attention_f16(h16, w_ln, c1, c2, stats, slots, nseq, L_tok, heads, causal=False, eps=1e-5, seq_off=None, seq_len=None):
    """LayerNorm-folded in_proj + multi-head attention in ONE launch (clip.py:210-214) on frame-major rows (row = s * L_tok + t,
    or seq_len[s] tokens from row seq_off[s]); q, k, v stay in LDS.  -> att [nseq * L_tok, W] fp16, bit-identical to
    linear_ln_f16 followed by attention_f16 for L_tok <= 56, equal to the fp16 rounding for 56 < L_tok <= 256.  W = heads * 64."""
    L.require_device(h16, w_ln, c1, c2, stats, seq_off, seq_len)
    W = h16.shape[1] if h16.dim() == 2 else -1
    _need(h16.dim() == 2 and _is(h16, torch.float16) and _is(w_ln, torch.float16, (3 * W, W)) and W == int(heads) * 64,
          "inproj_attention_f16: h16 [M, W] and w_ln [3 W, W] contiguous fp16, W = heads * 64")
    _need(_is(c1, torch.float32, (3 * W,)) and _is(c2, torch.float32, (3 * W,)), "inproj_attention_f16: c1, c2 [3 W] fp32")
    _need(1 <= int(slots) <= LN_MAX_SLOTS and stats.dtype == torch.float32 and stats.is_contiguous()
          and stats.numel() >= h16.shape[0] * int(slots) * 2, "inproj_attention_f16: stats [M, slots, 2] fp32")
    _need((seq_off is None) == (seq_len is None), "inproj_attention_f16: seq_off and seq_len come together")
    if seq_off is None:
        _need(int(nseq) * int(L_tok) == h16.shape[0], "inproj_attention_f16: M = nseq * L_tok rows")
    else:
        _need(_is(seq_off, torch.int32, (int(nseq),)) and _is(seq_len, torch.int32, (int(nseq),)),
              "inproj_attention_f16: seq_off, seq_len [nseq] int32")
    return _ops.inproj_attention_f16(h16, w_ln, c1, c2, stats, int(slots), float(eps), int(nseq), int(L_tok), int(heads),
                                     bool(causal), seq_off, seq_len)


def linear_resid_stats_f16(a, w, bias, h, tile=0, h16=None, stats=None, shift_in=None, stats_in=None, shift_out=None):
    """h += a @ w.T + bias in place; returns (h16, stats [M, slots, 2], slots, shift_out).  h16 = fp16(h - c_row) with
    c_row = shift_in + mean of the previous centred copy (stats_in [M, slots_in, 2]); without stats_in c_row = 0.
    ``h16`` [M,N] fp16 and ``stats`` (flat fp32, >= M*32*2) may be preallocated."""
    L.require_device(a, w, bias, h, h16, stats, shift_in, stats_in, shift_out)
    _need(a.dim() == 2 and w.dim() == 2 and _is(a, torch.float16) and _is(w, torch.float16, (w.shape[0], a.shape[1])),
          "linear_resid_stats_f16: a [M, K] and w [N, K] contiguous fp16")
    M, K = a.shape
    N = w.shape[0]
    _need(_is(h, torch.float32, (M, N)), "linear_resid_stats_f16: h [M, N] contiguous fp32")
    _need(bias is None or _is(bias, torch.float32, (N,)), "linear_resid_stats_f16: bias [N] fp32")
    _need(h16 is None or _is(h16, torch.float16, (M, N)), "linear_resid_stats_f16: h16 [M, N] contiguous fp16")
    _need(shift_in is None or _is(shift_in, torch.float32, (M,)), "linear_resid_stats_f16: shift_in [M] fp32")
    _need(shift_out is None or _is(shift_out, torch.float32, (M,)), "linear_resid_stats_f16: shift_out [M] fp32")
    if h16 is None:
        h16 = torch.empty(M, N, device=a.device, dtype=torch.float16)
    if stats is None:
        stats = torch.empty(M * LN_MAX_SLOTS * 2, device=a.device, dtype=torch.float32)
    slots_in = 0
    if stats_in is not None:
        slots_in = stats_in.shape[1] if stats_in.dim() == 3 else 1
        stats_in = stats_in.contiguous()
        if shift_out is None:
            shift_out = torch.empty(M, device=a.device, dtype=torch.float32)
    slots = T.resid_stats_slots(M, N, K, tile)
    _need(stats.dtype == torch.float32 and stats.is_contiguous() and stats.numel() >= M * slots * 2,
          "linear_resid_stats_f16: stats fp32 with >= M * %d * 2 entries" % slots)
    _need(stats_in is None or (stats_in.dtype == torch.float32 and stats_in.numel() == M * slots_in * 2),
          "linear_resid_stats_f16: stats_in [M, slots_in, 2] fp32")
    _ops.linear_resid_stats_f16(a, w, bias, h, h16, stats, shift_in, stats_in, slots_in, shift_out, tile)
    return h16, stats[:M * slots * 2].view(M, slots, 2), slots, shift_out


def head_project(h, gamma, beta, proj, row_mul=1, row_idx=None, rows=None):
    """LayerNorm + projection of selected rows of h [*, W] (fp32): row r -> h[r*row_mul + row_idx[r]]."""
    L.require_device(h, gamma, beta, proj, row_idx)
    h2 = h.float().contiguous().view(-1, h.shape[-1])
    R = int(rows) if rows is not None else h2.shape[0] // row_mul
    _need(proj.dim() == 2 and proj.shape[0] == h2.shape[1] and gamma.numel() == h2.shape[1] and beta.numel() == h2.shape[1],
          "head_project: gamma, beta [W], proj [W, E]")
    _need(R >= 0 and (row_idx is not None or R == 0 or (R - 1) * int(row_mul) < h2.shape[0]), "head_project: rows beyond h")
    _need(row_idx is None or (row_idx.dtype == torch.int32 and row_idx.is_contiguous() and row_idx.numel() >= R),
          "head_project: row_idx [rows] int32")
    return _ops.head_project(h2, int(row_mul), row_idx, gamma.detach().float().contiguous(),
                             beta.detach().float().contiguous(), proj.detach().float().contiguous(), R)
