"""N4, first slice of training: forward WITH saved activations and backward of one ResidualAttentionBlock
(modules/clip.py:196-253; the reference gets the backward from torch.autograd inside main.py:321
``scaler.scale(loss).backward()``).

    x [L, N, W] (LND, as the reference's blocks see it)
    y = x + out_proj(MHA(in_proj(ln_1(x))))          z = y + c_proj(QuickGELU(c_fc(ln_2(y))))

* forward: the op-level HIP entry points of the inference path (LayerNorm -> fp16, cc_linear_f16, cc_attention_f16,
  residual epilogue), keeping what the backward needs: x, ln_1(x), qkv, the attention output, y, ln_2(y), the c_fc output
  before and after QuickGELU.
* backward: the four Linear layers' dgrad (dX = dY W) and wgrad (dW = dY^T X) run on the SAME fp16 MFMA GEMM kernel with
  swapped operand roles - ``cc_linear_f16(a, w)`` computes a w^T, so dX = linear(dY, W^T) and dW = linear(dY^T, X^T) with the
  row count (padded to 64) as the contraction; gradients enter the matrix cores as fp16 with a per-tensor power-of-two scale
  chosen on the device (cc_cast_transpose_f16 / cc_linear_unscaled_f16: no host synchronisation); LayerNorm, QuickGELU, attention
  and bias gradients are the fp32 kernels of csrc/backward.hip.
* ``ResidualAttentionBlockFunction`` wires both into torch.autograd (d/dx and the 12 parameter gradients), so a block can sit
  in a graph that ends in losses.contrastive_loss; dist.GradientBuckets then averages the gradients over the ranks.

Round 4, later: the towers themselves (encode_image_train / encode_text_train below: patch embedding, ln_pre, the blocks with
a token-cluster module in front, the heads), BertAdam (utils/optimization.py) on cc_bertadam_step_f32 and train_epoch
(main.py:291-378) - CLIP4Clip.forward in training mode runs on them, so a training step reaches every parameter.  What is NOT
here: linear_patch='3d' and mean_residual in training, and fusion (train_epoch takes the reference's GradScaler; the master
weights are fp32 and the HIP backward scales per tensor on the device) - per-op launches from Python, checked against torch.autograd on the
reference model (tests/test_r4_gpu.py, fixture tests/golden/r4_golden.npz) to 1e-2 of each tensor's largest entry.
Transposed fp16 copies (W^T, dY^T, X^T) come from cc_cast_transpose_f16 (one read per matrix); the scale of a gradient operand
is divided out in the consuming GEMM's epilogue (cc_linear_unscaled_f16).
"""
import torch

from . import _lib as L
from . import ops
from .torch_ops import _st


def _check(rc, what):
    L.check(rc, what)


def _pad64(n):
    return -(-n // 64) * 64


def _cast_transpose(x, scaled, want_out=True, col_sums=False, amax=None, want_t=True, col_partials=False):
    """One read of a matrix -> its fp16 operand copies for a Linear's backward (cc_cast_transpose_f16):
    x fp32 [M, C] -> (x16 [M, C], x16^T [C, Mp] zero padded to a multiple of 64, scale or None); x fp16 -> (x, x^T, None).
    scaled: the device-chosen power-of-two scale of the gradients (returned as a 1-element device tensor); amax: a 2-float
    device tensor whose first entry already holds the largest |x| (written by the kernel that produced x) - the pass over x that
    finds it is skipped, the scale lands in the second entry."""
    x = x.contiguous()
    M, C = x.shape
    Mp = _pad64(M)
    out_t = torch.empty(C, Mp, device=x.device, dtype=torch.float16) if want_t else None    # (want_t False: the fp16 copy only)
    lib = L.lib()
    if x.dtype == torch.float16:
        _check(lib.cc_cast_transpose_f16(None, L.ptr(x), None, L.ptr(out_t), M, C, Mp, 0, None, None, None, None, 0, _st(x)),
               "cc_cast_transpose_f16")
        return x, out_t, None
    out = torch.empty(M, C, device=x.device, dtype=torch.float16) if want_out else None
    scratch = (amax if amax is not None else torch.empty(2, device=x.device, dtype=torch.float32)) if scaled else None
    cs = torch.empty(C, device=x.device, dtype=torch.float32) if (col_sums and not col_partials) else None
    # col_partials: the per-tile partial column sums [Mp / 64, C] stay in a tensor of their own and are returned instead of the
    # sums - cc_wgrad_tn_f16 adds them in the launch that adds its slices (the shared workspace is that call's scratch)
    ws = None
    if col_partials:
        ws = torch.empty(Mp // 64, C, device=x.device, dtype=torch.float32)
    elif col_sums:
        ws = L.workspace(lib.cc_cast_transpose_colsum_workspace_bytes(Mp, C), x.device)
    _check(lib.cc_cast_transpose_f16(L.ptr(x), None, L.ptr(out), L.ptr(out_t), M, C, Mp, (2 if amax is not None else 1) if scaled else 0,
                                     L.ptr(scratch[0:1]) if scaled else None, L.ptr(scratch[1:2]) if scaled else None, L.ptr(cs),
                                     L.ptr(ws), ws.numel() * ws.element_size() if ws is not None else 0, _st(x)), "cc_cast_transpose_f16")
    if col_partials:
        return out, out_t, (scratch[1:2] if scaled else None), ws
    if col_sums:
        return out, out_t, (scratch[1:2] if scaled else None), cs
    return out, out_t, (scratch[1:2] if scaled else None)


def _cast_scaled(x32):
    """fp32 tensor -> (fp16 copy scaled by a device-chosen power of two, the scale as a 1-element device tensor)."""
    x32 = x32.contiguous()
    out = torch.empty(x32.shape, device=x32.device, dtype=torch.float16)
    scratch = torch.empty(2, device=x32.device, dtype=torch.float32)
    _check(L.lib().cc_cast_scaled_f16(L.ptr(x32), L.ptr(out), x32.numel(), L.ptr(scratch[0:1]), L.ptr(scratch[1:2]), _st(x32)),
           "cc_cast_scaled_f16")
    return out, scratch[1:2]


def _unscale(x32, scale):
    _check(L.lib().cc_unscale_f32(L.ptr(x32), x32.numel(), L.ptr(scale), None, _st(x32)), "cc_unscale_f32")
    return x32


def _linear_unscaled(a16, w16, scale):
    """(a w^T) / scale in fp32: the GEMM with the operand's device-chosen scale undone in its epilogue."""
    M, K = a16.shape
    N = w16.shape[0]
    assert a16.dtype == torch.float16 and w16.dtype == torch.float16 and a16.is_contiguous() and w16.is_contiguous()
    assert w16.shape[1] == K and scale.dtype == torch.float32
    out = torch.empty(M, N, device=a16.device, dtype=torch.float32)
    _check(L.lib().cc_linear_unscaled_f16(L.ptr(a16), L.ptr(w16), L.ptr(out), M, N, K, L.ptr(scale), _st(a16)),
           "cc_linear_unscaled_f16")
    return out


def _wgrad_tn(dy16, x16, scale, col_partial=None):
    """dW [N1, N2] fp32 = (dy16^T x16) / scale from the row-major fp16 matrices dy16 [M, N1], x16 [M, N2] (cc_wgrad_tn_f16).
    col_partial [chunks, N1] (cc_cast_transpose_f16's partial column sums of dY): also returns the bias gradient [N1]."""
    M, N1 = dy16.shape
    N2 = x16.shape[1]
    assert dy16.dtype == torch.float16 and x16.dtype == torch.float16 and dy16.is_contiguous() and x16.is_contiguous()
    assert x16.shape[0] == M and scale.dtype == torch.float32
    lib = L.lib()
    dw = torch.empty(N1, N2, device=dy16.device, dtype=torch.float32)
    db = None
    if col_partial is not None:
        assert col_partial.dtype == torch.float32 and col_partial.is_contiguous() and col_partial.shape[1] == N1
        db = torch.empty(N1, device=dy16.device, dtype=torch.float32)
    ws = L.workspace(lib.cc_wgrad_tn_workspace_bytes(M, N1, N2), dy16.device)
    _check(lib.cc_wgrad_tn_f16(L.ptr(dy16), L.ptr(x16), L.ptr(dw), M, N1, N2, L.ptr(scale), L.ptr(col_partial),
                               col_partial.shape[0] if col_partial is not None else 0, L.ptr(db), L.ptr(ws), ws.numel(), _st(dy16)),
           "cc_wgrad_tn_f16")
    return dw if col_partial is None else (dw, db)


def _linear_resid(a16, w16, bias, resid):
    """resid + a w^T + bias in fp32 (cc_linear_resid_f16): the residual epilogue reading the rows it adds from ``resid`` - the
    forward keeps its input for the backward, so it cannot accumulate in place and used to copy it first."""
    M, K = a16.shape
    N = w16.shape[0]
    assert a16.dtype == torch.float16 and w16.dtype == torch.float16 and a16.is_contiguous() and w16.is_contiguous()
    assert w16.shape[1] == K and resid.dtype == torch.float32 and resid.is_contiguous() and tuple(resid.shape) == (M, N)
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() == N)
    out = torch.empty(M, N, device=a16.device, dtype=torch.float32)
    _check(L.lib().cc_linear_resid_f16(L.ptr(a16), L.ptr(w16), L.ptr(bias), L.ptr(resid), L.ptr(out), M, N, K, 0, _st(a16)),
           "cc_linear_resid_f16")
    return out


def _column_sums(x32):
    rows, cols = x32.shape
    out = torch.empty(cols, device=x32.device, dtype=torch.float32)
    lib = L.lib()
    ws = L.workspace(lib.cc_column_sums_workspace_bytes(rows, cols), x32.device)
    _check(lib.cc_column_sums_f32(L.ptr(x32), rows, cols, L.ptr(out), L.ptr(ws), ws.numel(), _st(x32)), "cc_column_sums_f32")
    return out


def _ln_backward(x, gamma, dy, dres, eps=1e-5, amax=None):
    rows, W = x.shape
    dx = torch.empty_like(x)
    dg, db = torch.empty(W, device=x.device), torch.empty(W, device=x.device)
    lib = L.lib()
    ws = L.workspace(lib.cc_layernorm_backward_workspace_bytes(rows, W), x.device)
    _check(lib.cc_layernorm_backward_f32(L.ptr(x), W, L.ptr(gamma), L.ptr(dy), L.ptr(dres), L.ptr(dx), L.ptr(dg), L.ptr(db),
                                         rows, W, float(eps), L.ptr(amax), L.ptr(ws), ws.numel(), _st(x)), "cc_layernorm_backward_f32")
    return dx, dg, db


def _wt16(w):
    """W [N, K] (fp32 master weight or fp16) -> W^T [K, Np] fp16, the dgrad's operand (columns behind N are zeros and are
    sliced away: cc_linear_f16 takes the row stride from the shape, so the view must be made contiguous only when N % 64)."""
    w = w.detach()
    N, K = w.shape
    _, wt, _ = _cast_transpose(w.float() if w.dtype not in (torch.float16, torch.float32) else w, scaled=False, want_out=False)
    return wt if wt.shape[1] == N else wt[:, :N].contiguous()


def _w16_pair(w):
    """fp32 master weight [N, K] -> (W fp16 for the forward GEMM, W^T [K, N] fp16 for the backward's dgrad) from ONE read."""
    w = w.detach()
    if w.dtype != torch.float32:
        return w.to(torch.float16).contiguous(), None
    w16, wt, _ = _cast_transpose(w, scaled=False)
    N = w.shape[0]
    return w16, (wt if wt.shape[1] == N else wt[:, :N].contiguous())


def _grad_linear(dy32, x16, w16_t, need_dx=True, amax=None, need_dw=True):
    """Gradients of y = x W^T + b for dy [M, N] fp32, x [M, K] fp16, W^T [K, N] fp16 -> (dx [M, K], dW [N, K], db [N]) fp32.
    The gradient is read ONCE for its two fp16 layouts (row-major for dX = dY W, transposed + padded for dW = dY^T X).
    need_dw False (a frozen layer, main.py's freeze_layer_num): no transposed copies, no wgrad GEMM, dW = None."""
    # round 5: the weight gradient multiplies dY and X as they lie in memory (cc_wgrad_tn_f16: LDS transposing reads) wherever both
    # widths are multiples of its 128-wide tile - every layer of the CLIP towers; other widths keep the transposed copies
    M, N1 = dy32.shape
    tn = need_dw and N1 % 128 == 0 and x16.shape[1] % 128 == 0
    dy16, dy16_t, scale, db = _cast_transpose(dy32, scaled=True, col_sums=True, amax=amax,   # (+ the bias gradient, same read)
                                              want_t=need_dw and not tn, col_partials=tn)
    dw = None
    if tn:
        dw, db = _wgrad_tn(dy16, x16, scale, col_partial=db)                                  # dY^T X (+ the bias sums' last step)
    elif need_dw:
        _, x16_t, _ = _cast_transpose(x16, scaled=False)
        dw = _linear_unscaled(dy16_t, x16_t, scale)                                           # dY^T X
    # (dX last: the kernel that consumes it runs next and finds it in the memory-side cache)
    dx = _linear_unscaled(dy16, w16_t, scale) if need_dx else None                            # dY W
    return dx, dw, db


def block_forward_train(block, x_lnd):
    """-> (z [L, N, W] fp32, saved dict).  ``block``: a centerclip_amd.clip.ResidualAttentionBlock without a cluster module."""
    if block.tokencluster_inter is not None:
        raise NotImplementedError("block backward: blocks with a token-cluster module are not covered by this slice")
    L.require_device(x_lnd)
    Lt, N, W = x_lnd.shape
    M = N * Lt
    causal = block.attn_mask is not None
    f32 = lambda t: t.detach().float().contiguous()
    x = x_lnd.detach().float().permute(1, 0, 2).contiguous().view(M, W)              # frame-major rows (row = seq*L + token)
    wq, wo, wf, wp = (_w16_pair(w) for w in (block.attn.in_proj_weight, block.attn.out_proj.weight, block.mlp["c_fc"].weight,
                                             block.mlp["c_proj"].weight))
    n1 = ops.layernorm(x, f32(block.ln_1.weight), f32(block.ln_1.bias), eps=block.ln_1.eps, out_f16=True)
    qkv = ops.linear_f16(n1, wq[0], f32(block.attn.in_proj_bias), "f16")
    att = ops.attention_f16(qkv, N, Lt, block.n_head, causal=causal)
    y = _linear_resid(att, wo[0], f32(block.attn.out_proj.bias), x)           # x + out_proj(att): x itself is kept for the backward
    n2 = ops.layernorm(y, f32(block.ln_2.weight), f32(block.ln_2.bias), eps=block.ln_2.eps, out_f16=True)
    u_pre = ops.linear_f16(n2, wf[0], f32(block.mlp["c_fc"].bias), "f16")
    u = torch.empty_like(u_pre)
    _check(L.lib().cc_quick_gelu_f16(L.ptr(u_pre), L.ptr(u), u.numel(), _st(u)), "cc_quick_gelu_f16")
    z = _linear_resid(u, wp[0], f32(block.mlp["c_proj"].bias), y)
    wt = dict(in_proj=wq[1], out_proj=wo[1], c_fc=wf[1], c_proj=wp[1])                  # W^T of the same read, for the dgrads
    saved = dict(x=x, n1=n1, qkv=qkv, att=att, y=y, n2=n2, u_pre=u_pre, u=u, shape=(Lt, N, W), causal=causal, wt=wt)
    # (a VIEW of the frame-major rows: the next block's permute + contiguous then costs nothing - a chain of plain blocks never
    # copies its activations between the two layouts)
    return z.view(N, Lt, W).permute(1, 0, 2), saved


def block_backward(block, saved, dz_lnd, need=None):
    """dz [L, N, W] -> (dx [L, N, W], {parameter name: gradient}) for the forward that produced ``saved``.  ``need``
    (optional): {parameter name: bool} - weight gradients that nobody asked for (frozen layers) are not computed (None)."""
    need = need or {}
    nw = lambda key: bool(need.get(key, True))
    Lt, N, W = saved["shape"]
    M = N * Lt
    wt = saved.get("wt", {})
    f16t = lambda w, key: wt[key] if wt.get(key) is not None else _wt16(w)              # W^T as the dgrad's operand
    f32 = lambda t: t.detach().float().contiguous()
    dz = dz_lnd.detach().float().permute(1, 0, 2).contiguous().view(M, W)
    g = {}
    # z = y + c_proj(u)
    du, g["mlp.c_proj.weight"], g["mlp.c_proj.bias"] = _grad_linear(dz, saved["u"], f16t(block.mlp["c_proj"].weight, "c_proj"), need_dw=nw("mlp.c_proj.weight"))
    # u = QuickGELU(u_pre)
    # (the three gradients this function produces AND multiplies publish their largest magnitude from the producing kernel:
    #  the fp16 cast of each then needs no pass of its own to choose the scale)
    am = torch.zeros(3, 2, device=dz.device, dtype=torch.float32)
    du_pre = torch.empty_like(du)
    _check(L.lib().cc_quick_gelu_backward_f16(L.ptr(saved["u_pre"]), L.ptr(du), L.ptr(du_pre), du.numel(), L.ptr(am[0]), _st(du)),
           "cc_quick_gelu_backward_f16")
    # u_pre = c_fc(ln_2(y))
    dn2, g["mlp.c_fc.weight"], g["mlp.c_fc.bias"] = _grad_linear(du_pre, saved["n2"], f16t(block.mlp["c_fc"].weight, "c_fc"), amax=am[0], need_dw=nw("mlp.c_fc.weight"))
    dy, g["ln_2.weight"], g["ln_2.bias"] = _ln_backward(saved["y"], f32(block.ln_2.weight), dn2, dz, eps=block.ln_2.eps, amax=am[1])   # + the residual branch
    # y = x + out_proj(att)
    datt, g["attn.out_proj.weight"], g["attn.out_proj.bias"] = _grad_linear(dy, saved["att"], f16t(block.attn.out_proj.weight, "out_proj"), amax=am[1], need_dw=nw("attn.out_proj.weight"))
    dqkv = torch.empty(M, 3 * W, device=dz.device, dtype=torch.float32)
    ab_bytes = L.lib().cc_attention_backward_workspace_bytes(N, Lt, block.n_head)       # (0 for Lt <= 64)
    ab_ws = L.workspace(ab_bytes, dz.device) if ab_bytes else None
    _check(L.lib().cc_attention_backward_f16(L.ptr(saved["qkv"]), L.ptr(datt), L.ptr(dqkv), N, Lt, block.n_head, W,
                                             int(saved["causal"]), L.ptr(am[2]), L.ptr(ab_ws), ab_bytes, _st(dz)),
           "cc_attention_backward_f16")
    dn1, g["attn.in_proj_weight"], g["attn.in_proj_bias"] = _grad_linear(dqkv, saved["n1"], f16t(block.attn.in_proj_weight, "in_proj"), amax=am[2], need_dw=nw("attn.in_proj_weight"))
    dx, g["ln_1.weight"], g["ln_1.bias"] = _ln_backward(saved["x"], f32(block.ln_1.weight), dn1, dy, eps=block.ln_1.eps)
    return dx.view(N, Lt, W).permute(1, 0, 2), g


_PARAM_ORDER = ("attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias", "ln_1.weight",
                "ln_1.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias", "ln_2.weight",
                "ln_2.bias")


class ResidualAttentionBlockFunction(torch.autograd.Function):
    """z = block(x) with the HIP forward / backward above inside torch.autograd: gradients reach x and the block's 12
    parameter tensors (passed as arguments so that autograd sees them)."""

    @staticmethod
    def forward(ctx, block, x, *params):
        z, saved = block_forward_train(block, x)
        ctx.block, ctx.saved = block, saved
        # (the activations and the forward-time W^T copies live in ctx.saved, outside autograd's version tracking: remember the
        #  parameters' versions, so that a weight changed in place between forward and backward is an error, as it is for
        #  tensors kept with save_for_backward, and not a silently stale W^T)
        ctx.versions = tuple(p._version for p in params)
        return z

    @staticmethod
    def backward(ctx, dz):
        named = dict(ctx.block.named_parameters())
        if tuple(named[k]._version for k in _PARAM_ORDER) != ctx.versions:
            raise RuntimeError("ResidualAttentionBlockFunction: a parameter of the block was modified in place between forward "
                               "and backward (the saved W^T copies are those of the forward)")
        need = {k: bool(ctx.needs_input_grad[2 + i]) for i, k in enumerate(_PARAM_ORDER)}
        dx, g = block_backward(ctx.block, ctx.saved, dz, need=need)
        return (None, dx) + tuple((g[k].view_as(named[k]).to(named[k].dtype) if (need[k] and g[k] is not None) else None)
                                  for k in _PARAM_ORDER)


def block_apply(block, x_lnd):
    """Differentiable block forward: ``z = block_apply(block, x); loss(z).backward()`` fills x.grad and block.*.grad."""
    named = dict(block.named_parameters())
    return ResidualAttentionBlockFunction.apply(block, x_lnd, *[named[k] for k in _PARAM_ORDER])


# ================================================================================================ the towers, differentiable
# What main.py:291-378 (train_epoch) needs from the model: CLIP4Clip.forward in training mode with gradients reaching every
# parameter.  The towers below are the reference's forward (modules/clip.py:320-345 visual, :471-496 text) composed of the HIP
# forward / backward pieces: patch embedding and the projection heads as GEMMs (LinearFunction: dgrad / wgrad on the forward
# kernel), LayerNorms (LayerNormFunction), the blocks (ResidualAttentionBlockFunction), the token-cluster module (its own
# autograd, cluster/cluster.py).  What stays torch glue: reshapes / permutes / concatenation, the broadcast adds of the class
# and positional embeddings, the embedding-table gather with its scatter-add gradient, the EOT row gather.

class LinearFunction(torch.autograd.Function):
    """y [M, N] fp32 = x [M, K] @ w[N, K]^T (+ b): fp16 MFMA operands, fp32 accumulate; gradients as _grad_linear."""

    @staticmethod
    def forward(ctx, x, w, b):
        x16 = x.detach().to(torch.float16).contiguous()
        ctx.save_for_backward(x16, w)
        ctx.has_bias = b is not None
        ctx.need_dx = x.requires_grad
        return ops.linear_f16(x16, w.detach().to(torch.float16).contiguous(), None if b is None else b.detach().float().contiguous(),
                              "f32")

    @staticmethod
    def backward(ctx, dy):
        x16, w = ctx.saved_tensors
        dx, dw, db = _grad_linear(dy.contiguous().float(), x16, _wt16(w) if ctx.need_dx else None, need_dx=ctx.need_dx)
        return dx, dw.to(w.dtype), (db if ctx.has_bias else None)


class LayerNormFunction(torch.autograd.Function):
    """LayerNorm over the last dim of x [rows, W] fp32 (modules/clip.py:183-189), backward = cc_layernorm_backward_f32."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.detach().float().contiguous()
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return ops.layernorm(x, gamma.detach().float().contiguous(), beta.detach().float().contiguous(), eps)

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dx, dg, db = _ln_backward(x, gamma.detach().float().contiguous(), dy.contiguous().float(), None, ctx.eps)
        return dx, dg.to(gamma.dtype), db.to(gamma.dtype), None


def _layernorm(ln, x2d):
    return LayerNormFunction.apply(x2d, ln.weight, ln.bias, ln.eps)


def _blocks(transformer, x_lnd):
    """The resblocks on LND activations; a block's token-cluster module runs in front of it (clip.py:236-242)."""
    for blk in transformer.resblocks:
        if blk.tokencluster_inter is not None:
            if getattr(blk.tokencluster_inter, "mean_residual", False):
                raise NotImplementedError("training towers: mean_residual is not built")
            x_lnd, _ = blk.tokencluster_inter(x_lnd)
            x_lnd = _plain(blk, x_lnd)
        else:
            x_lnd = block_apply(blk, x_lnd)
    return x_lnd


class _NoCluster:
    """A view of a block without its cluster module (block_forward_train refuses blocks that carry one: here the module has
    already run)."""

    def __init__(self, blk):
        self._blk = blk
        self.tokencluster_inter = None

    def __getattr__(self, name):
        return getattr(self._blk, name)


def _plain(blk, x_lnd):
    view = _NoCluster(blk)
    named = dict(blk.named_parameters())
    return ResidualAttentionBlockFunction.apply(view, x_lnd, *[named[k] for k in _PARAM_ORDER])


def encode_image_train(clip, video, video_frame):
    """CLIP.encode_image (modules/clip.py:460-469 with VisualTransformer.forward :320-345, linear_patch '2d') with gradients:
    video [F, 3, H, W] fp32 -> (features [F', embed_dim], cluster_loss)."""
    vis = clip.visual
    if vis.linear_patch != '2d':
        raise NotImplementedError("training towers: linear_patch='3d' is not built")
    L.require_device(video)
    F, p, W = video.shape[0], vis.patch_size, vis.width
    g = vis.input_resolution // p
    # conv1 (kernel = stride = p, no bias) as a GEMM over the patch rows (c, kh, kw) - a reshape of the frames, no gather
    a = video.float().view(F, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(F * g * g, 3 * p * p)
    x = LinearFunction.apply(a, vis.conv1.weight.view(W, -1), None).view(F, g * g, W)
    cls = vis.class_embedding.to(x.dtype) + torch.zeros(F, 1, W, dtype=x.dtype, device=x.device)
    x = torch.cat([cls, x], dim=1) + vis.positional_embedding.to(x.dtype)
    x = _layernorm(vis.ln_pre, x.reshape(F * (g * g + 1), W)).view(F, g * g + 1, W)
    x = _blocks(vis.transformer, x.permute(1, 0, 2).contiguous()).permute(1, 0, 2)          # NLD -> LND -> NLD
    cls_rows = x[:, 0, :].contiguous()                        # ln_post(x) @ proj, of which encode_image keeps the CLS row
    feats = LinearFunction.apply(_layernorm(vis.ln_post, cls_rows), vis.proj.t(), None)
    return feats, torch.zeros((), device=video.device)


def encode_text_train(clip, ids):
    """CLIP.encode_text (modules/clip.py:471-496) with gradients: ids [B, n_ctx] -> [B, embed_dim]."""
    L.require_device(ids)
    B, n_ctx = ids.shape
    W = clip.transformer.width
    x = clip.token_embedding(ids).float() + clip.positional_embedding[:n_ctx].float()
    x = _blocks(clip.transformer, x.permute(1, 0, 2).contiguous()).permute(1, 0, 2).contiguous()
    eot = x[torch.arange(B, device=x.device), ids.argmax(dim=-1)]                             # the EOT token has the largest id
    return LinearFunction.apply(_layernorm(clip.ln_final, eot.contiguous()), clip.text_projection.t(), None)


# ================================================================================================ BertAdam
def warmup_cosine(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    import math
    return 0.5 * (1.0 + math.cos(math.pi * x))


def warmup_constant(x, warmup=0.002):
    return x / warmup if x < warmup else 1.0


def warmup_linear(x, warmup=0.002):
    return x / warmup if x < warmup else max((x - 1.) / (warmup - 1.), 0)


SCHEDULES = {'warmup_cosine': warmup_cosine, 'warmup_constant': warmup_constant, 'warmup_linear': warmup_linear}


class BertAdam(torch.optim.Optimizer):
    """utils/optimization.py:55-170 (the optimizer main.py:161-167 builds): same constructor, same state names
    ('step', 'next_m', 'next_v'), same per-tensor clipping / decoupled weight decay / schedule; the tensor arithmetic of a
    step is one cc_bertadam_step_f32 call per parameter (no host synchronisation)."""

    def __init__(self, params, lr, warmup=-1, t_total=-1, schedule='warmup_linear', b1=0.9, b2=0.999, e=1e-6,
                 weight_decay=0.01, max_grad_norm=1.0, capturable=False):
        # capturable (not in the reference): the scheduled learning rate of each group reaches the kernels through a device
        # float, so a step captured into a hipGraph can be replayed with the schedule's next value (GraphedTrainStep)
        self.capturable = bool(capturable)
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if schedule not in SCHEDULES:
            raise ValueError("Invalid schedule parameter: {}".format(schedule))
        if not 0.0 <= warmup < 1.0 and not warmup == -1:
            raise ValueError("Invalid warmup: {} - should be in [0.0, 1.0[ or -1".format(warmup))
        if not 0.0 <= b1 < 1.0:
            raise ValueError("Invalid b1 parameter: {} - should be in [0.0, 1.0[".format(b1))
        if not 0.0 <= b2 < 1.0:
            raise ValueError("Invalid b2 parameter: {} - should be in [0.0, 1.0[".format(b2))
        if not e >= 0.0:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(e))
        super().__init__(params, dict(lr=lr, schedule=schedule, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e,
                                      weight_decay=weight_decay, max_grad_norm=max_grad_norm))

    @staticmethod
    def _lr(group, step):
        if group['t_total'] != -1:
            return group['lr'] * SCHEDULES[group['schedule']](step / group['t_total'], group['warmup'])
        return group['lr']

    def get_lr(self):
        lr = []
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    return [0]
                lr.append(self._lr(group, state['step']))
        return lr

    _MULTI_MAX_N = 8192                     # CC_BERTADAM_MULTI_MAX_N (include/centerclip_hip.h)

    def _multi_small(self, items, hyper, capturing, device):
        """All small tensors of groups with the same (b1, b2, e, max_grad_norm) in ONE launch (cc_bertadam_multi_f32): the
        records (cc_bertadam_item: four tensor pointers, the group's device learning rate, n, weight decay) are staged through
        pinned memory and re-sent only when a pointer changed.  A captured step owns its own staging buffers (the graph replays
        the host-to-device copy), allocated during the eager warm-up that precedes the capture."""
        import numpy as np
        rec = np.zeros(len(items), dtype=np.dtype([('p', '<u8'), ('g', '<u8'), ('m', '<u8'), ('v', '<u8'), ('lr_dev', '<u8'),
                                                   ('n', '<i4'), ('lr', '<f4'), ('wd', '<f4'), ('pad', '<i4')]))
        for i, (p, grad, m, v, lr_dev, wd) in enumerate(items):
            rec[i] = (p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), lr_dev.data_ptr(), p.numel(), 0.0, wd, 0)
        raw = rec.tobytes()
        slot = self._multi.setdefault(hyper, {})
        if capturing:
            host, dev = slot.pop("spare", (None, None))
            if host is None or host.numel() != len(raw):
                raise RuntimeError("BertAdam: run one eager step with the same parameters before capturing (staging buffers)")
            host.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            dev.copy_(host, non_blocking=True)
            self._multi_keep.append((host, dev))                  # the graph reads both on every replay
        else:
            if slot.get("raw") != raw:
                host = torch.frombuffer(bytearray(raw), dtype=torch.uint8).pin_memory()
                if slot.get("dev") is None or slot["dev"].numel() != len(raw):
                    slot["dev"] = torch.empty(len(raw), dtype=torch.uint8, device=device)
                slot["dev"].copy_(host, non_blocking=True)
                slot["raw"] = raw
            if "spare" not in slot or slot["spare"][0].numel() != len(raw):
                slot["spare"] = (torch.empty(len(raw), dtype=torch.uint8).pin_memory(),
                                 torch.empty(len(raw), dtype=torch.uint8, device=device))
            dev = slot["dev"]
        b1, b2, e, max_norm = hyper
        _check(L.lib().cc_bertadam_multi_f32(L.ptr(dev), len(items), b1, b2, e, max_norm, _st(dev)), "cc_bertadam_multi_f32")

    def _multi_large(self, items, hyper, capturing, device):
        """All large tensors of groups with the same (b1, b2, e, max_grad_norm) in TWO launches (cc_bertadam_multi_large_f32:
        every tensor's norm workgroups, then every tensor's step workgroups) instead of two per tensor - ~100 tensors of a
        ViT-B/32 CLIP: 204 launches -> 2.  Records (cc_bertadam_big_item) staged like the small tensors' (see _multi_small)."""
        import numpy as np
        lib = L.lib()
        rec = np.zeros(len(items), dtype=np.dtype([('p', '<u8'), ('g', '<u8'), ('m', '<u8'), ('v', '<u8'), ('lr_dev', '<u8'),
                                                   ('n', '<i8'), ('lr', '<f4'), ('wd', '<f4'), ('nb0', '<i4'), ('nb', '<i4'),
                                                   ('sb0', '<i4'), ('sb', '<i4')]))
        nb0 = sb0 = 0
        for i, (p, grad, m, v, lr_dev, wd) in enumerate(items):
            nb, sb = int(lib.cc_bertadam_norm_blocks(p.numel())), int(lib.cc_bertadam_step_blocks(p.numel()))
            rec[i] = (p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), lr_dev.data_ptr(), p.numel(), 0.0, wd, nb0, nb, sb0, sb)
            nb0 += nb
            sb0 += sb
        raw = rec.tobytes()
        slot = self._multi.setdefault((hyper, "large"), {})
        if slot.get("partial") is None or slot["partial"].numel() < nb0:
            if capturing:
                raise RuntimeError("BertAdam: run one eager step with the same parameters before capturing (partial sums)")
            slot["partial"] = torch.empty(nb0, dtype=torch.float64, device=device)
        if capturing:
            host, dev = slot.pop("spare", (None, None))
            if host is None or host.numel() != len(raw):
                raise RuntimeError("BertAdam: run one eager step with the same parameters before capturing (staging buffers)")
            host.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            dev.copy_(host, non_blocking=True)
            self._multi_keep.append((host, dev))
        else:
            if slot.get("raw") != raw:
                host = torch.frombuffer(bytearray(raw), dtype=torch.uint8).pin_memory()
                if slot.get("dev") is None or slot["dev"].numel() != len(raw):
                    slot["dev"] = torch.empty(len(raw), dtype=torch.uint8, device=device)
                slot["dev"].copy_(host, non_blocking=True)
                slot["raw"] = raw
            if "spare" not in slot or slot["spare"][0].numel() != len(raw):
                slot["spare"] = (torch.empty(len(raw), dtype=torch.uint8).pin_memory(),
                                 torch.empty(len(raw), dtype=torch.uint8, device=device))
            dev = slot["dev"]
        b1, b2, e, max_norm = hyper
        part = slot["partial"]
        _check(lib.cc_bertadam_multi_large_f32(L.ptr(dev), len(items), nb0, sb0, b1, b2, e, max_norm, L.ptr(part),
                                               part.numel() * 8, _st(dev)), "cc_bertadam_multi_large_f32")

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = L.lib()
        capturing = self.capturable and torch.cuda.is_current_stream_capturing()
        if not hasattr(self, "_lr_dev"):
            self._lr_dev = {}                                     # group index -> 1-element device tensor (not optimizer state)
            self._multi, self._multi_keep = {}, []
        small, large = {}, {}                                     # (b1, b2, e, max_grad_norm) -> records of the small / large tensors
        for gi, group in enumerate(self.param_groups):
            lr_set = False
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("BertAdam (HIP): fp32 contiguous parameters (the master weights)")
                L.require_device(p)
                grad = p.grad if (p.grad.dtype == torch.float32 and p.grad.is_contiguous()) else None
                if grad is None:
                    p.grad = p.grad.float().contiguous()
                    grad = p.grad
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['next_m'] = torch.zeros_like(p)
                    state['next_v'] = torch.zeros_like(p)
                ws = L.workspace(lib.cc_bertadam_workspace_bytes(), p.device)
                lr_dev = None
                if self.capturable:
                    lr_dev = self._lr_dev.get(gi)
                    if lr_dev is None or lr_dev.device != p.device:
                        lr_dev = self._lr_dev[gi] = torch.zeros(1, device=p.device, dtype=torch.float32)
                    if not capturing and not lr_set:
                        lr_dev.fill_(float(self._lr(group, state['step'])))
                        lr_set = True
                if self.capturable and p.numel() <= self._MULTI_MAX_N:
                    hyper = (float(group['b1']), float(group['b2']), float(group['e']), float(group['max_grad_norm']))
                    small.setdefault(hyper, []).append((p, grad, state['next_m'], state['next_v'], lr_dev, float(group['weight_decay'])))
                elif self.capturable:
                    hyper = (float(group['b1']), float(group['b2']), float(group['e']), float(group['max_grad_norm']))
                    large.setdefault(hyper, []).append((p, grad, state['next_m'], state['next_v'], lr_dev, float(group['weight_decay'])))
                else:
                    _check(lib.cc_bertadam_step_f32(L.ptr(p), L.ptr(grad), L.ptr(state['next_m']), L.ptr(state['next_v']), p.numel(),
                                                    float(self._lr(group, state['step'])), float(group['b1']), float(group['b2']),
                                                    float(group['e']), float(group['weight_decay']), float(group['max_grad_norm']),
                                                    L.ptr(lr_dev), L.ptr(ws), ws.numel(), _st(p)), "cc_bertadam_step_f32")
                if not capturing:
                    state['step'] += 1
        for hyper, items in small.items():
            self._multi_small(items, hyper, capturing, items[0][0].device)
        for hyper, items in large.items():
            self._multi_large(items, hyper, capturing, items[0][0].device)
        return loss

    def refresh_lr(self):
        """capturable: write every group's scheduled learning rate (from the host-side step counts) into its device float -
        call before replaying a captured step."""
        for gi, group in enumerate(self.param_groups):
            steps = [self.state[p]['step'] for p in group['params'] if p in self.state and len(self.state[p])]
            if steps and getattr(self, "_lr_dev", {}).get(gi) is not None:
                self._lr_dev[gi].fill_(float(self._lr(group, steps[0])))

    def advance(self):
        """capturable: count one replayed step for every parameter that has state."""
        for group in self.param_groups:
            for p in group['params']:
                if p in self.state and len(self.state[p]):
                    self.state[p]['step'] += 1


def prep_optim_params_groups(args, model, coef_lr=1.):
    """utils/optimization.py:173-208 (BertAdam branch): CLIP parameters at lr * coef_lr, newly added modules at lr, no weight
    decay for biases / LayerNorm."""
    model = getattr(model, 'module', model)
    named = list(model.named_parameters())
    no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']
    no_clip = args.new_added_modules
    dec = [(n, p) for n, p in named if not any(nd in n for nd in no_decay)]
    nodec = [(n, p) for n, p in named if any(nd in n for nd in no_decay)]
    is_clip = lambda n: "clip." in n and not any(nd in n for nd in no_clip)
    return [{'params': [p for n, p in dec if is_clip(n)], 'weight_decay': args.wd, 'lr': args.lr * coef_lr},
            {'params': [p for n, p in nodec if is_clip(n)], 'weight_decay': 0.0, 'lr': args.lr * coef_lr},
            {'params': [p for n, p in dec if not is_clip(n)], 'weight_decay': args.wd},
            {'params': [p for n, p in nodec if not is_clip(n)], 'weight_decay': 0.0}]


# ================================================================================================ train_epoch
def train_epoch(epoch, args, model, train_dataloader, device, optimizer, global_step, scheduler=None, buckets=None,
                log=None, scaler=None):
    """main.py:291-378 for this path: zero_grad -> forward (training branch of CLIP4Clip.forward) -> backward ->
    [gradient average over the ranks, dist.GradientBuckets] -> [clip_grad_norm_] -> optimizer.step -> clamp logit_scale.
    ``model``: a centerclip_amd.clip4clip.CLIP4Clip in training mode.  -> (mean loss, global_step).

    ``scaler`` (main.py:309-330, the reference's ``--fp16`` branch): a ``torch.cuda.amp.GradScaler`` (or anything with its
    scale / unscale_ / step / update).  The forward here always feeds the matrix cores fp16 operands with fp32 accumulation
    and keeps fp32 master weights - what ``autocast`` gives the reference - so the branch adds what the scaler itself does:
    the loss is multiplied by the scale before backward (the HIP backward picks a power-of-two scale per gradient tensor on
    the device, so the factor passes through exactly), gradients are unscaled (and averaged over the ranks) before clipping,
    a step whose gradients hold an inf / NaN is skipped and the scale backed off, as GradScaler.step / update do."""
    model.train()
    total_loss, nb = 0.0, 0
    for step, batch in enumerate(train_dataloader):
        optimizer.zero_grad()
        if scheduler is not None:
            scheduler(optimizer, global_step=global_step)
        input_ids, input_mask, segment_ids, video, video_mask = tuple(t.to(device=device, non_blocking=True) for t in batch)
        output = model(input_ids, segment_ids, input_mask, video, video_mask)
        loss = output['loss'].mean()
        if args.gradient_accumulation_steps > 1:
            loss = loss / args.gradient_accumulation_steps
        if scaler is not None:
            scaler.scale(loss).backward()
        else:
            loss.backward()
        if (step + 1) % args.gradient_accumulation_steps == 0:
            if buckets is not None:
                buckets.reduce()
            if scaler is not None:
                if getattr(args, "clip_grad_norm", None) is not None:
                    scaler.unscale_(optimizer)           # (clipping sees the true gradients, main.py:324-326)
                    torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip_grad_norm)
                scaler.step(optimizer)                   # skipped when a gradient holds an inf / NaN
                scaler.update()
            else:
                if getattr(args, "clip_grad_norm", None) is not None:
                    torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip_grad_norm)
                optimizer.step()
            global_step += 1
        with torch.no_grad():                                    # (main.py:336-340; tracked, so the cached copies refresh)
            model.clip.logit_scale.clamp_(0.1, 4.6052)
        if log is not None:
            log(epoch, step, float(loss.detach()), float(output['sim_loss'].detach()), global_step)
        total_loss += float(loss.detach())
        nb += 1
    return total_loss / max(nb, 1), global_step


class GraphedTrainStep:
    """One training step (forward, backward, optimizer, logit_scale clamp - main.py:300-340 for one batch) captured into a
    hipGraph and replayed on static input buffers: no op of the step synchronises with the host, so the replay runs at the GPU
    time of its kernels instead of the host's launch rate (cfg-2 shape: 19 ms against 40-100 ms launched op by op).
    Single process (a captured step cannot contain the RCCL exchange of GradientBuckets); fixed batch shape; an optimizer
    built with capturable=True.  The first call warms up eagerly (2 steps on the given batch) and captures - on a snapshot:
    parameters, moments and step counts are put back before the one replay that counts, so that EVERY call, the first
    included, is exactly one optimizer step (main.py:300-340) and the schedule position equals the caller's step count."""

    def __init__(self, model, optimizer, gradient_accumulation_steps=1):
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs BertAdam(..., capturable=True)")
        if gradient_accumulation_steps != 1:
            raise NotImplementedError("GraphedTrainStep: gradient accumulation is not built")
        self.model, self.optimizer = model, optimizer
        self.graph = self.static = self.loss = None

    def _step(self):
        self.optimizer.zero_grad(set_to_none=True)       # (captured: the gradients live in the graph's pool, no fill + accumulate)
        out = self.model(self.static[0], self.static[2], self.static[1], self.static[3], self.static[4])
        loss = out['loss'].mean()
        loss.backward()
        self.optimizer.step()
        with torch.no_grad():
            self.model.clip.logit_scale.clamp_(0.1, 4.6052)
        return loss.detach()

    def _tensors(self):
        seen, out = set(), []
        for p in list(self.model.parameters()) + [p for g in self.optimizer.param_groups for p in g['params']]:
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
        return out

    def _snapshot(self):
        """Copies of everything a step changes: every parameter, and per parameter the optimizer's (step, next_m, next_v)."""
        snap = []
        for p in self._tensors():
            st = self.optimizer.state.get(p, {})
            snap.append((p, p.detach().clone(), st.get('step'), st['next_m'].clone() if 'next_m' in st else None,
                         st['next_v'].clone() if 'next_v' in st else None))
        return snap

    @torch.no_grad()
    def _restore(self, snap):
        """In place (the captured graph holds the addresses of the parameters and of the moments the warm-up created);
        Tensor.copy_ bumps the version counter, so cached fp16 / folded copies of the weights refresh."""
        for p, value, step, m, v in snap:
            p.copy_(value)
            st = self.optimizer.state.get(p)
            if not st:
                continue
            st['step'] = 0 if step is None else step
            st['next_m'].zero_() if m is None else st['next_m'].copy_(m)
            st['next_v'].zero_() if v is None else st['next_v'].copy_(v)

    def __call__(self, batch):
        """batch = (input_ids, input_mask, segment_ids, video, video_mask) as the dataloaders yield it -> the step's loss (a
        device tensor that the next call overwrites)."""
        dev = next(self.model.parameters()).device
        if self.graph is None:
            self.model.train()
            self.static = [t.to(dev).clone() for t in batch]
            snap = self._snapshot()
            for _ in range(2):                                    # allocator / staging-buffer warm-up (the optimizer's records)
                self._step()
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):                    # (the capture pass does not execute)
                self.loss = self._step()
            self._restore(snap)                                   # the two warm-up steps never happened
            self.optimizer.refresh_lr()
            self.graph.replay()
            self.optimizer.advance()
            return self.loss
        for dst, src in zip(self.static, batch):
            dst.copy_(src, non_blocking=True)
        self.optimizer.refresh_lr()
        self.graph.replay()
        self.optimizer.advance()
        return self.loss
