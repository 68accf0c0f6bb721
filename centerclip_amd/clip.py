"""MI355X-native mirror of the reference's ``modules/clip.py`` (ViT + text transformer subset):
same class / method names, forward signatures and state-dict keys, executed by
the HIP encoders of libcenterclip_hip.so through the C ABI.  No PyTorch compute fallback.

Built: VisualTransformer (ViT-B/32, ViT-B/16, linear_patch '2d' and '3d'), the text Transformer,
ResidualAttentionBlock.forward / Transformer.forward on LND activations (composed from the op-level
entry points), CLIP.encode_image / encode_text (incl. return_hidden=True), CLIP.forward,
build_clip_model, load_clip_state_dict (local files).  Not built (out of the hot path, SURVEY §2.1
#2): ModifiedResNet, weight download.

All compute goes through ``torch.ops.centerclip.*`` (torch_ops.py).
"""
import contextlib
import ctypes
import torch
from torch import nn

import os

from . import _lib as L
from . import ops
from . import torch_ops as T
from ._lib_clip import BlockWeights, TextModel, VitModel, CC_MAX_LAYERS, ROWS_ALL_TEXT, ROWS_ALL_LAST_BLOCK
from .cluster import get_cluster_inter


# the constants of the reference's loader (dataloaders/decode.py:43-48), applied by the uint8 input path
PIXEL_MEAN = (0.48145466, 0.4578275, 0.40821073)
PIXEL_STD = (0.26862954, 0.26130258, 0.27577711)


def frames_descriptor(x, mean=PIXEL_MEAN, std=PIXEL_STD):
    """Tensor -> (cc_frames struct, tensor kept alive).  Float input is what the reference's encode_image takes
    ([N,3,H,W], already normalised); uint8 input is the decoder's output, [N,3,H,W] or [N,H,W,3], and gets the
    loader's u8/255 -> (x - mean)/std (dataloaders/transforms.py:19-34,166) inside the patch gather (N3)."""
    from ._lib_clip import Frames
    fr = Frames()
    if x.dtype == torch.uint8:
        if x.ndim != 4 or (x.shape[1] != 3 and x.shape[-1] != 3):
            raise ValueError("uint8 frames must be [N,3,H,W] or [N,H,W,3], got %s" % (tuple(x.shape),))
        x = x.contiguous()
        fr.format = 1 if x.shape[1] == 3 else 2
        fr.mean = (ctypes.c_float * 3)(*[float(v) for v in mean])
        fr.std = (ctypes.c_float * 3)(*[float(v) for v in std])
    else:
        x = x.float().contiguous()
        fr.format = 0
    fr.data = x.data_ptr()
    return fr, x


_ZERO = {}


def zero_scalar(device):
    """The `cluster_loss` the retrieval path returns (torch.zeros([]) in the reference, clip.py:265): one cached tensor
    per device instead of a fill kernel per forward."""
    key = str(device)
    if key not in _ZERO:
        _ZERO[key] = torch.zeros([], device=device)
    return _ZERO[key]


class LayerNorm(nn.Module):
    """LayerNorm computed in fp32 whatever the input dtype (modules/clip.py:183-189)."""

    def __init__(self, width, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(width))
        self.bias = nn.Parameter(torch.zeros(width))
        self.eps = eps

    def forward(self, x):
        y = ops.layernorm(x.float().contiguous(), self.weight.float(), self.bias.float(), self.eps)
        return y.type(x.dtype)


class _Attn(nn.Module):
    """Parameter container with nn.MultiheadAttention's state-dict names (clip.py:205)."""

    def __init__(self, width):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)


class ResidualAttentionBlock(nn.Module):
    """One block under the reference's parameter names (clip.py:197-217).  Inside the fused encoders
    (VisualTransformer.encode, CLIP.encode_text / encode_pair) the compute of all blocks runs from one enqueue;
    ``forward`` is the block-level drop-in of clip.py:228-253 on the reference's LND activations, composed from the same
    op-level entry points (folded LayerNorm -> in_proj, attention, out_proj + residual, folded LayerNorm -> c_fc +
    QuickGELU, c_proj + residual).  ``attn_mask`` follows the reference: None for the visual tower, the (callable)
    causal mask builder for the text tower - the kernels take it as a causal flag.
    ``tokencluster_inter`` is decided per block by get_cluster_inter."""

    def __init__(self, d_model, n_head, attn_mask=None, block_id=1, args=None):
        super().__init__()
        self.attn = _Attn(d_model)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.ModuleDict({"c_fc": nn.Linear(d_model, d_model * 4), "c_proj": nn.Linear(d_model * 4, d_model)})
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = attn_mask
        self.n_head = n_head
        self.block_id = block_id
        self.tokencluster_inter = get_cluster_inter(d_model, block_id, args)
        self._fold = None

    def _folded(self):
        """fp16 operands + LayerNorm-folded weights of this block, rebuilt when a parameter changes."""
        key = _Pack.signature(self)
        if self._fold is None or self._fold[0] != key:
            f16 = lambda t: t.detach().to(torch.float16).contiguous()
            f32 = lambda t: t.detach().float().contiguous()
            in_w, in_c1, in_c2 = ops.fold_layernorm_linear(self.attn.in_proj_weight, self.attn.in_proj_bias,
                                                           self.ln_1.weight, self.ln_1.bias)
            fc_w, fc_c1, fc_c2 = ops.fold_layernorm_linear(self.mlp["c_fc"].weight, self.mlp["c_fc"].bias,
                                                           self.ln_2.weight, self.ln_2.bias)
            self._fold = (key, dict(in_w=in_w, in_c1=in_c1, in_c2=in_c2, fc_w=fc_w, fc_c1=fc_c1, fc_c2=fc_c2,
                                    out_w=f16(self.attn.out_proj.weight), out_b=f32(self.attn.out_proj.bias),
                                    proj_w=f16(self.mlp["c_proj"].weight), proj_b=f32(self.mlp["c_proj"].bias)))
        return self._fold[1]

    def forward(self, x_tuple):
        """(x [L, N, W] LND, video_frame, cluster_loss) -> same tuple   (clip.py:228-253)"""
        x, video_frame, cluster_loss = x_tuple
        L.require_device(x)
        res_x = None
        if self.tokencluster_inter is not None:                 # place 1, before the self-attention (clip.py:236-242)
            x, res_x = self.tokencluster_inter(x)
        Lq, N, W = x.shape
        M = Lq * N
        w = self._folded()
        xin = x.float().contiguous().view(M, W)
        h16, st, sh = ops.row_stats(xin)                        # ln_1 reads the clustered x ...
        # ... and the residual stream (updated in place below) starts from x, or from the frame means (mean_residual, :242)
        h = (xin if res_x is None else res_x.float().contiguous().view(M, W)).clone()
        qkv = ops.linear_ln_f16(h16, w["in_w"], w["in_c1"], w["in_c2"], st, 1, eps=self.ln_1.eps)
        att = ops.attention_f16(qkv, N, Lq, self.n_head, causal=self.attn_mask is not None, seq_rows=1, tok_rows=N)
        h16, st1, slots1, _ = ops.linear_resid_stats_f16(att, w["out_w"], w["out_b"], h, shift_in=sh,
                                                         stats_in=st.view(M, 1, 2))
        u = ops.linear_ln_f16(h16, w["fc_w"], w["fc_c1"], w["fc_c2"], st1, slots1, gelu=True, eps=self.ln_2.eps)
        ops.linear_f16(u, w["proj_w"], w["proj_b"], "f32_resid", out=h)
        return (h.view(Lq, N, W).type(x.dtype), video_frame, cluster_loss)


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, attn_mask=None, args=None):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask, i + 1, args)
                                         for i in range(layers)])

    def forward(self, x, video_frame=-1, visual=False):
        """x [L, N, W] (LND) through the blocks one by one (clip.py:264-269); visual=True returns the whole tuple."""
        cluster_loss = zero_scalar(x.device)
        out = self.resblocks((x, video_frame, cluster_loss))
        return out if visual else out[0]


class _Pack:
    """fp16/fp32 device copies of the weights + the ctypes structs handed to the C ABI.
    Rebuilt when any parameter's version or the device changes."""

    def __init__(self):
        self.key = None
        self.keep = []
        self.struct = None
        self.handle = None

    @staticmethod
    def signature(module):
        return tuple((p.data_ptr(), p._version, str(p.device), p.dtype) for p in module.parameters())

    def f32(self, t):
        t = t.detach().float().contiguous()
        self.keep.append(t)
        return t.data_ptr()

    def f16(self, t):
        t = t.detach().to(torch.float16).contiguous()
        self.keep.append(t)
        return t.data_ptr()

    def fold(self, linear_w, linear_b, ln):
        """LayerNorm folded into the consuming Linear (cc_fold_layernorm_linear_f32) -> (w' f16, c1, c2) pointers."""
        w = linear_w.detach().float().contiguous()
        b = linear_b.detach().float().contiguous()
        gamma, beta = ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous()
        N, K = w.shape
        wf = torch.empty(N, K, device=w.device, dtype=torch.float16)
        c1 = torch.empty(N, device=w.device, dtype=torch.float32)
        c2 = torch.empty(N, device=w.device, dtype=torch.float32)
        L.check(L.lib().cc_fold_layernorm_linear_f32(L.ptr(w), L.ptr(b), L.ptr(gamma), L.ptr(beta), N, K, L.ptr(wf),
                                                     L.ptr(c1), L.ptr(c2), L.stream_ptr(w.device)),
                "cc_fold_layernorm_linear_f32")
        self.keep += [wf, c1, c2]
        return wf.data_ptr(), c1.data_ptr(), c2.data_ptr()

    def blocks(self, transformer):
        arr = (BlockWeights * len(transformer.resblocks))()
        for i, blk in enumerate(transformer.resblocks):
            b = arr[i]
            b.in_proj_ln_weight_f16, b.in_proj_ln_c1, b.in_proj_ln_c2 = self.fold(
                blk.attn.in_proj_weight, blk.attn.in_proj_bias, blk.ln_1)
            b.c_fc_ln_weight_f16, b.c_fc_ln_c1, b.c_fc_ln_c2 = self.fold(
                blk.mlp["c_fc"].weight, blk.mlp["c_fc"].bias, blk.ln_2)
            b.ln_1_weight, b.ln_1_bias = self.f32(blk.ln_1.weight), self.f32(blk.ln_1.bias)
            b.in_proj_weight_f16, b.in_proj_bias = self.f16(blk.attn.in_proj_weight), self.f32(blk.attn.in_proj_bias)
            b.out_proj_weight_f16 = self.f16(blk.attn.out_proj.weight)
            b.out_proj_bias = self.f32(blk.attn.out_proj.bias)
            b.ln_2_weight, b.ln_2_bias = self.f32(blk.ln_2.weight), self.f32(blk.ln_2.bias)
            b.c_fc_weight_f16, b.c_fc_bias = self.f16(blk.mlp["c_fc"].weight), self.f32(blk.mlp["c_fc"].bias)
            b.c_proj_weight_f16, b.c_proj_bias = self.f16(blk.mlp["c_proj"].weight), self.f32(blk.mlp["c_proj"].bias)
        self.keep.append(arr)
        return arr


class VisualTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim, linear_patch='2d',
                 video_frames=None, args=None):
        super().__init__()
        assert linear_patch in ['2d', '3d']
        self.input_resolution, self.patch_size, self.output_dim, self.width = input_resolution, patch_size, output_dim, width
        self.heads = heads
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, args=args)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self.linear_patch = linear_patch
        self.video_frames = video_frames
        if linear_patch == '3d':                     # clip.py:296-299: Conv3d over (t, h, w), kernel (3, p, p), zero padding 1 along t
            self.conv2 = nn.Conv3d(3, width, kernel_size=(3, patch_size, patch_size), stride=(1, patch_size, patch_size),
                                   padding=(1, 0, 0), bias=False)
        self.register_buffer("position_ids", torch.arange(self.positional_embedding.shape[0]).expand(1, -1))
        self._pack = _Pack()
        self.last_medoids = None
        # False (shipped): block 12 computes out_proj / c_fc / c_proj for the CLS rows only (the rows ln_post + proj read);
        # True: every row, as the reference does (cc_vit_model.row_policy, bench.py / the tests compare both)
        self.all_last_block_rows = False

    # -- C-ABI model struct -----------------------------------------------------------------
    def _model(self):
        """-> handle of the packed cc_vit_model in the torch_ops registry (rebuilt when a parameter changes)."""
        sig = _Pack.signature(self) + (("rows", bool(self.all_last_block_rows)),)
        pk = self._pack
        if pk.key == sig:
            return pk.handle
        if pk.handle is not None:
            T.release_model(pk.handle)
        pk.keep, pk.key = [], sig
        m = VitModel()
        m.row_policy = ROWS_ALL_LAST_BLOCK if self.all_last_block_rows else 0
        m.layers, m.width, m.heads = self.transformer.layers, self.width, self.heads
        m.patch, m.resolution, m.embed_dim = self.patch_size, self.input_resolution, self.output_dim
        m.conv1_weight_f16 = pk.f16(self.conv1.weight.reshape(self.width, -1))
        if self.linear_patch == '3d':                # [W, 3(c), 3(t), p, p] flattened = the im2col row order of the 3-d gather
            m.conv2_weight_f16 = pk.f16(self.conv2.weight.reshape(self.width, -1))
        m.class_embedding, m.positional_embedding = pk.f32(self.class_embedding), pk.f32(self.positional_embedding)
        m.ln_pre_weight, m.ln_pre_bias = pk.f32(self.ln_pre.weight), pk.f32(self.ln_pre.bias)
        m.ln_post_weight, m.ln_post_bias = pk.f32(self.ln_post.weight), pk.f32(self.ln_post.bias)
        m.proj = pk.f32(self.proj)
        m.blocks = ctypes.cast(pk.blocks(self.transformer), ctypes.POINTER(BlockWeights))
        first = None
        tokens = (self.input_resolution // self.patch_size) ** 2
        variants = (L.ClusterVariant * CC_MAX_LAYERS)()
        any_variant = False
        dev = self.proj.device
        for i, blk in enumerate(self.transformer.resblocks):
            tc = blk.tokencluster_inter
            if tc is not None:
                m.cluster_frames[i], m.cluster_tokens[i] = tc.after_block_frames, tc.cluster_num
                if tc.algorithm == 'pooling' and tc.cluster_num != tokens:
                    raise ValueError("'pooling' keeps the token count: cluster_num_blocks[%d] must be %d" % (i, tokens))
                if getattr(tc, "mean_residual", False) and tc.cluster_num != tokens:
                    raise ValueError("mean_residual keeps the token count (cluster.py:229): cluster_num_blocks[%d] must be %d"
                                     % (i, tokens))
                variants[i], keep = tc.variant(tc.frame_duration * tokens, dev)     # N2: per-block variant
                pk.keep.extend(keep)
                any_variant = any_variant or not tc.is_default_variant
                tokens = tc.cluster_num
                first = first or tc
        if any_variant:
            pk.keep.append(variants)
            m.cluster_variants = ctypes.cast(variants, ctypes.c_void_p)
        if first is not None:
            m.cluster_metric = L.METRIC_IDS[first.distance]
            m.cluster_norm_p, m.cluster_threshold = float(first.norm_p), float(first.threshold)
            m.cluster_iter_limit, m.cluster_split_size = int(first.iter_limit), int(first.split_size)
            m.cluster_pre_norm = int(bool(first.pre_norm))
        pk.struct = m
        meta = dict(embed_dim=self.output_dim, width=self.width, final=self._final_meta)
        pk.handle = T.register_model(m, meta, pk)
        return pk.handle

    def _final_meta(self, video_frame):
        """(frames per clip, tokens incl. CLS) after all cluster blocks and (segments, K) of the LAST k-medoids block (the
        one whose ids cc_vit_encode reports) or None."""
        frames, ltok = self.final_shape(video_frame)
        med = None
        for blk in self.transformer.resblocks:
            tc = blk.tokencluster_inter
            if tc is not None and tc.algorithm in ('kmediods++', 'spectral'):
                med = (tc.after_block_frames, tc.cluster_num)
        return frames, ltok, med

    def final_shape(self, video_frame):
        """(frames, tokens incl. CLS) per clip after all cluster blocks."""
        frames, tokens = video_frame, (self.input_resolution // self.patch_size) ** 2
        for blk in self.transformer.resblocks:
            tc = blk.tokencluster_inter
            if tc is not None:
                frames, tokens = tc.after_block_frames, tc.cluster_num
        return frames, tokens + 1

    def encode(self, x, video_frame=-1, want_hidden=False, want_medoids=False, forced_medoids=None):
        """[B*T, 3, H, W] -> (features [B*T_final, output_dim], hidden [B*T_final, L, W] | None).  With want_medoids the
        ids of the LAST k-medoids block ([T_new*B, K] of that block) are left in ``last_medoids``."""
        L.require_device(x)
        if x.dtype != torch.uint8:
            x = x.float()
        x = x.contiguous()
        BT = x.shape[0]
        T_ = video_frame if video_frame and video_frame > 0 else 1
        has_cluster = any(b.tokencluster_inter is not None for b in self.transformer.resblocks)
        if self.linear_patch == '3d':
            assert video_frame and video_frame > 0, "linear_patch='3d' needs video_frame (clip.py:307)"
        elif not has_cluster:
            T_ = 1
        assert BT % T_ == 0
        if forced_medoids is not None:
            # one id tensor [B * T_new, K] per cluster block (a list / tuple for plans with several blocks): back to back
            if isinstance(forced_medoids, (list, tuple)):
                forced_medoids = torch.cat([m.to(device=x.device, dtype=torch.long).reshape(-1) for m in forced_medoids])
            forced_medoids = forced_medoids.to(device=x.device, dtype=torch.long).contiguous()
        feats, hidden, med = torch.ops.centerclip.vit_encode(x, self._model(), BT // T_, T_, bool(want_hidden),
                                                             bool(want_medoids and has_cluster), forced_medoids)
        self.last_medoids = med if med.numel() else None
        return feats, (hidden if want_hidden else None)

    def forward(self, x, video_frame=-1):
        """-> (hidden [N', L', W] before ln_post, cluster_loss)   (clip.py:304-349).  Runs the fused encoder (the
        reference's self.transformer(x, video_frame, visual=True) call is inside it); the block-by-block form is
        Transformer.forward."""
        _, hidden = self.encode(x, video_frame, want_hidden=True)
        return hidden, zero_scalar(x.device)


class CLIP(nn.Module):
    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size,
                 context_length, vocab_size, transformer_width, transformer_heads, transformer_layers,
                 linear_patch='2d', video_frames=None, args=None):
        super().__init__()
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError("ModifiedResNet visual towers are not built (ViT only)")
        self.context_length = context_length
        self.visual = VisualTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim, linear_patch, video_frames, args)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads,
                                       attn_mask=self.build_attention_mask)
        self.vocab_size = vocab_size
        self.embed_dim = embed_dim
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]))
        self._text_pack = _Pack()
        # row policies of the text tower (cc_text_model.row_policy): False (shipped) = rows up to each caption's EOT only /
        # the last block on the EOT rows only; True = every row, as the reference computes them
        self.all_text_rows = False
        self.all_last_block_rows = False
        self.initialize_parameters()

    @contextlib.contextmanager
    def row_policy(self, all_text_rows=False, all_last_block_rows=False):
        """Temporarily compute the rows nothing downstream reads, exactly as the reference does (both towers).  The policy is
        part of the packed model (per model, no process state); a hipGraph captured inside the block keeps its policy."""
        saved = (self.all_text_rows, self.all_last_block_rows, self.visual.all_last_block_rows)
        self.all_text_rows, self.all_last_block_rows = bool(all_text_rows), bool(all_last_block_rows)
        self.visual.all_last_block_rows = bool(all_last_block_rows)
        try:
            yield self
        finally:
            self.all_text_rows, self.all_last_block_rows, self.visual.all_last_block_rows = saved

    def invalidate(self):
        """Drop the packed device copies (fp16 operands, folded LayerNorms).  The packs are keyed on (data_ptr, _version)
        of every parameter; writes through ``p.data`` do not bump ``_version`` - call this after such a write (or write
        under ``torch.no_grad()`` instead, which is tracked)."""
        for pk in (self._text_pack, self.visual._pack):
            if pk.handle is not None:
                T.release_model(pk.handle)
            pk.key, pk.handle, pk.keep = None, None, []
        for tr in (self.transformer, self.visual.transformer):
            for blk in tr.resblocks:
                blk._fold = None

    def initialize_parameters(self):
        """Same statistics as modules/clip.py:419-446."""
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        for tr in (self.transformer, self.visual.transformer):
            proj_std = (tr.width ** -0.5) * ((2 * tr.layers) ** -0.5)
            attn_std, fc_std = tr.width ** -0.5, (2 * tr.width) ** -0.5
            for block in tr.resblocks:
                nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
                nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
                nn.init.normal_(block.mlp["c_fc"].weight, std=fc_std)
                nn.init.normal_(block.mlp["c_proj"].weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=self.transformer.width ** -0.5)

    def build_attention_mask(self, context_length):
        mask = torch.zeros(context_length, context_length)
        mask.fill_(float("-inf"))
        mask.triu_(1)
        return mask

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image, return_hidden=False, video_frame=-1):
        """-> (x [N', embed_dim], cluster_loss), or (x, hidden [N', L', embed_dim]) with return_hidden (clip.py:460-469:
        hidden = ln_post(h) @ proj on every token, x = its CLS row)."""
        if return_hidden:
            _, h = self.visual.encode(image, video_frame, want_hidden=True)
            hidden = ops.head_project(h, self.visual.ln_post.weight, self.visual.ln_post.bias, self.visual.proj)
            hidden = hidden.view(h.shape[0], h.shape[1], -1)
            return hidden[:, 0, :], hidden
        feats, _ = self.visual.encode(image, video_frame)
        return feats, zero_scalar(image.device)

    def _text_model(self):
        tower = nn.ModuleList([self.transformer, self.token_embedding, self.ln_final])
        sig = _Pack.signature(tower) + ((self.positional_embedding._version, self.text_projection._version,
                                         self.positional_embedding.data_ptr(), self.text_projection.data_ptr(),
                                         bool(self.all_text_rows), bool(self.all_last_block_rows)),)
        pk = self._text_pack
        if pk.key == sig:
            return pk.handle
        if pk.handle is not None:
            T.release_model(pk.handle)
        pk.keep, pk.key = [], sig
        m = TextModel()
        m.row_policy = (ROWS_ALL_TEXT if self.all_text_rows else 0) | (ROWS_ALL_LAST_BLOCK if self.all_last_block_rows else 0)
        m.layers, m.width, m.heads = self.transformer.layers, self.transformer.width, self.transformer.heads
        m.context_length, m.vocab_size, m.embed_dim = self.context_length, self.vocab_size, self.embed_dim
        m.token_embedding, m.positional_embedding = pk.f32(self.token_embedding.weight), pk.f32(self.positional_embedding)
        m.ln_final_weight, m.ln_final_bias = pk.f32(self.ln_final.weight), pk.f32(self.ln_final.bias)
        m.text_projection = pk.f32(self.text_projection)
        m.blocks = ctypes.cast(pk.blocks(self.transformer), ctypes.POINTER(BlockWeights))
        pk.struct = m
        pk.handle = T.register_model(m, dict(embed_dim=self.embed_dim, width=self.transformer.width), pk)
        return pk.handle

    def encode_pair(self, image, text, video_frame=-1, out=None):
        """encode_image + encode_text of one CLIP4Clip.forward call in a single enqueue (cc_clip_encode):
        block i of the text tower shares its launches with block i of the ViT.
        -> (image features [N', embed_dim], text features [B, embed_dim]); ``out`` = preallocated (vfeat, tfeat)."""
        L.require_device(image, text)
        vis = self.visual
        if image.dtype != torch.uint8:
            image = image.float()
        image = image.contiguous()
        ids = text.to(torch.long).contiguous()
        T_ = video_frame if video_frame and video_frame > 0 else 1
        if vis.linear_patch == '3d':
            assert video_frame and video_frame > 0, "linear_patch='3d' needs video_frame (clip.py:307)"
        elif not any(b.tokencluster_inter is not None for b in vis.transformer.resblocks):
            T_ = 1
        B = image.shape[0] // T_
        forced = getattr(vis, "forced_medoids", None)       # test hook ("given identical medoid sets")
        keep = getattr(vis, "keep_medoids", False)
        if out is None and forced is None and not keep:
            return torch.ops.centerclip.clip_encode(image, ids, vis._model(), self._text_model(), B, T_)
        h = vis._model()
        frames, _, med_shape = vis._final_meta(T_)
        if out is None:
            out = (torch.empty(B * frames, self.embed_dim, device=image.device, dtype=torch.float32),
                   torch.empty(ids.shape[0], self.embed_dim, device=image.device, dtype=torch.float32))
        med = torch.empty(B * med_shape[0], med_shape[1], device=image.device, dtype=torch.long) if (keep and med_shape) else None
        if forced is not None:
            forced = forced.to(device=image.device, dtype=torch.long).contiguous()
        torch.ops.centerclip.clip_encode_out(image, ids, h, self._text_model(), B, T_, out[0], out[1], med, forced)
        vis.last_medoids = med
        return out

    def encode_text(self, text, return_hidden=False):
        """ids [B, n_ctx] -> [B, embed_dim]: EOT row of ln_final(x) @ text_projection (clip.py:471-496); with
        return_hidden also hidden [B, n_ctx, embed_dim] = ln_final(x) @ text_projection on every token."""
        L.require_device(text)
        ids = text.to(torch.long).contiguous()
        feats, h = torch.ops.centerclip.text_encode(ids, self._text_model(), bool(return_hidden))
        if return_hidden:
            hidden = ops.head_project(h, self.ln_final.weight, self.ln_final.bias, self.text_projection)
            return feats, hidden.view(h.shape[0], h.shape[1], -1)
        return feats

    def forward(self, image, text):
        """-> (logits_per_image, logits_per_text)   (clip.py:498-512)"""
        image_features, _ = self.encode_image(image)
        text_features = self.encode_text(text)
        img, txt = ops.normalize_rows(image_features), ops.normalize_rows(text_features)
        mult = T.logit_multiplier(self.logit_scale.detach())
        return ops.scaled_dot_nt(img, txt, mult), ops.scaled_dot_nt(txt, img, mult)


_PT_NAME = {"ViT-B/32": "ViT-B-32.pt", "ViT-B/16": "ViT-B-16.pt"}      # clip.py:29-36 (ViT entries)


def load_clip_state_dict(pretrained_clip_name="ViT-B/32", pretrained_dir=os.path.expanduser("~/models/pretrained")):
    """The local-file half of modules/clip.py:load_clip_state_dict: <pretrained_dir>/ViT-B-32.pt as a TorchScript
    archive (the OpenAI release format) or a plain state dict.  Nothing is downloaded (no network on the target boxes):
    a missing file raises."""
    if pretrained_clip_name not in _PT_NAME:
        raise NotImplementedError("only the ViT checkpoints are supported, got %r" % (pretrained_clip_name,))
    model_path = os.path.join(pretrained_dir, _PT_NAME[pretrained_clip_name])
    if not os.path.exists(model_path):
        raise FileNotFoundError("%s not found (weight download is not built; place the checkpoint there)" % model_path)
    try:
        return torch.jit.load(model_path, map_location="cpu").eval().state_dict()
    except RuntimeError:
        return torch.load(model_path, map_location="cpu")


def build_clip_model(state_dict, convert_fp16=True, linear_patch='2d', cut_top_layer=0, load_state_dict=True,
                     is_eval=True, video_frames=None, args=None):
    """Infer the architecture from an OpenAI-CLIP style state dict and build the model
    (modules/clip.py:539-635).  Returns (model, config dict).  ``convert_fp16`` is accepted for
    signature parity: GEMM operands are always fp16 copies, master parameters stay fp32."""
    if "visual.proj" not in state_dict:
        raise NotImplementedError("ModifiedResNet checkpoints are not supported (ViT only)")
    vision_width = state_dict["visual.conv1.weight"].shape[0]
    vision_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
    grid_size = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    image_resolution = vision_patch_size * grid_size
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    transformer_width = state_dict["ln_final.weight"].shape[0]
    transformer_heads = transformer_width // 64
    transformer_layers = len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")))
    config = dict(embed_dim=embed_dim, image_resolution=image_resolution, vision_layers=vision_layers,
                  vision_width=vision_width, vision_patch_size=vision_patch_size, context_length=context_length,
                  vocab_size=vocab_size, transformer_width=transformer_width, transformer_heads=transformer_heads,
                  transformer_layers=transformer_layers)
    model = CLIP(embed_dim, image_resolution, vision_layers - cut_top_layer, vision_width, vision_patch_size,
                 context_length, vocab_size, transformer_width, transformer_heads, transformer_layers - cut_top_layer,
                 linear_patch=linear_patch, video_frames=video_frames, args=args)
    if load_state_dict:
        sd = {k: v for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}
        model.load_state_dict(sd, strict=False)
    return (model.eval() if is_eval else model), config
