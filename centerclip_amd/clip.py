"""MI355X-native mirror of the reference's ``modules/clip.py`` (ViT + text transformer subset):
same class / method names, forward signatures and state-dict keys, executed by
the HIP encoders of libcenterclip_hip.so through the C ABI.  No PyTorch compute fallback.

Built: VisualTransformer (ViT-B/32, ViT-B/16, linear_patch '2d'), the text Transformer,
CLIP.encode_image / encode_text, build_clip_model.  Not built (out of the hot path, SURVEY §2.1
#2): ModifiedResNet, linear_patch='3d', weight download, return_hidden=True.
"""
import ctypes
import torch
from torch import nn

from . import _lib as L
from ._lib_clip import BlockWeights, TextModel, VitModel, CC_MAX_LAYERS
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


class LayerNorm(nn.Module):
    """LayerNorm computed in fp32 whatever the input dtype (modules/clip.py:183-189)."""

    def __init__(self, width, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(width))
        self.bias = nn.Parameter(torch.zeros(width))
        self.eps = eps

    def forward(self, x):
        from . import ops
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
    """Parameters of one block under the reference's names (clip.py:197-217); the compute runs
    inside the fused encoders.  ``tokencluster_inter`` is decided per block by get_cluster_inter."""

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


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, attn_mask=None, args=None):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask, i + 1, args)
                                         for i in range(layers)])


class _Pack:
    """fp16/fp32 device copies of the weights + the ctypes structs handed to the C ABI.
    Rebuilt when any parameter's version or the device changes."""

    def __init__(self):
        self.key = None
        self.keep = []
        self.struct = None

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
        if linear_patch != '2d':
            raise NotImplementedError("linear_patch='3d' is not built")
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
        self.register_buffer("position_ids", torch.arange(self.positional_embedding.shape[0]).expand(1, -1))
        self._pack = _Pack()
        self.last_medoids = None

    # -- C-ABI model struct -----------------------------------------------------------------
    def _model(self):
        sig = _Pack.signature(self)
        pk = self._pack
        if pk.key == sig:
            return pk.struct
        pk.keep, pk.key = [], sig
        m = VitModel()
        m.layers, m.width, m.heads = self.transformer.layers, self.width, self.heads
        m.patch, m.resolution, m.embed_dim = self.patch_size, self.input_resolution, self.output_dim
        m.conv1_weight_f16 = pk.f16(self.conv1.weight.reshape(self.width, -1))
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
        return m

    def final_shape(self, video_frame):
        """(frames, tokens incl. CLS) per clip after all cluster blocks."""
        frames, tokens = video_frame, (self.input_resolution // self.patch_size) ** 2
        for blk in self.transformer.resblocks:
            tc = blk.tokencluster_inter
            if tc is not None:
                frames, tokens = tc.after_block_frames, tc.cluster_num
        return frames, tokens + 1

    def encode(self, x, video_frame=-1, want_hidden=False, want_medoids=False, forced_medoids=None):
        """[B*T, 3, H, W] -> (features [B*T_final, output_dim], hidden [B*T_final, L, W] | None)."""
        L.require_device(x)
        fr, x = frames_descriptor(x)
        BT = x.shape[0]
        T = video_frame if video_frame and video_frame > 0 else 1
        has_cluster = any(b.tokencluster_inter is not None for b in self.transformer.resblocks)
        if not has_cluster:
            T = 1
        assert BT % T == 0
        B = BT // T
        m = self._model()
        lib = L.lib()
        frames, ltok = self.final_shape(T)
        feats = torch.empty(B * frames, self.output_dim, device=x.device, dtype=torch.float32)
        hidden = torch.empty(B * frames, ltok, self.width, device=x.device, dtype=torch.float32) if want_hidden else None
        med = None
        if want_medoids and has_cluster:
            med = torch.empty(B * frames, ltok - 1, device=x.device, dtype=torch.long)
        ws = L.workspace(lib.cc_vit_workspace_bytes(ctypes.byref(m), B, T), x.device)
        if forced_medoids is not None:
            forced_medoids = forced_medoids.to(device=x.device, dtype=torch.long).contiguous()
        L.check(lib.cc_vit_encode_frames(ctypes.byref(m), ctypes.byref(fr), B, T, L.ptr(feats), L.ptr(hidden),
                                         L.ptr(med), L.ptr(forced_medoids), L.ptr(ws), ws.numel(),
                                         L.stream_ptr(x.device)), "cc_vit_encode_frames")
        self.last_medoids = med
        return feats, hidden

    def forward(self, x, video_frame=-1):
        """-> (hidden [N', L', W] before ln_post, cluster_loss)   (clip.py:304-349)"""
        _, hidden = self.encode(x, video_frame, want_hidden=True)
        return hidden, torch.zeros([], device=x.device)


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
        self.initialize_parameters()

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
        """-> (x [N', embed_dim], cluster_loss)   (clip.py:460-469)"""
        if return_hidden:
            raise NotImplementedError("return_hidden=True is not built (only the CLS row is projected)")
        feats, _ = self.visual.encode(image, video_frame)
        return feats, torch.zeros([], device=image.device)

    def _text_model(self):
        tower = nn.ModuleList([self.transformer, self.token_embedding, self.ln_final])
        sig = _Pack.signature(tower) + ((self.positional_embedding._version, self.text_projection._version,
                                         self.positional_embedding.data_ptr(), self.text_projection.data_ptr()),)
        pk = self._text_pack
        if pk.key == sig:
            return pk.struct
        pk.keep, pk.key = [], sig
        m = TextModel()
        m.layers, m.width, m.heads = self.transformer.layers, self.transformer.width, self.transformer.heads
        m.context_length, m.vocab_size, m.embed_dim = self.context_length, self.vocab_size, self.embed_dim
        m.token_embedding, m.positional_embedding = pk.f32(self.token_embedding.weight), pk.f32(self.positional_embedding)
        m.ln_final_weight, m.ln_final_bias = pk.f32(self.ln_final.weight), pk.f32(self.ln_final.bias)
        m.text_projection = pk.f32(self.text_projection)
        m.blocks = ctypes.cast(pk.blocks(self.transformer), ctypes.POINTER(BlockWeights))
        pk.struct = m
        return m

    def encode_pair(self, image, text, video_frame=-1):
        """encode_image + encode_text of one CLIP4Clip.forward call in a single enqueue (cc_clip_encode):
        block i of the text tower shares its launches with block i of the ViT.
        -> (image features [N', embed_dim], text features [B, embed_dim])"""
        L.require_device(image, text)
        vis = self.visual
        fr, x = frames_descriptor(image)
        ids = text.to(torch.long).contiguous()
        T = video_frame if video_frame and video_frame > 0 else 1
        if not any(b.tokencluster_inter is not None for b in vis.transformer.resblocks):
            T = 1
        B = x.shape[0] // T
        Bt, Lt = ids.shape
        vm, tm = vis._model(), self._text_model()
        frames, _ = vis.final_shape(T)
        lib = L.lib()
        vfeat = torch.empty(B * frames, self.embed_dim, device=x.device, dtype=torch.float32)
        tfeat = torch.empty(Bt, self.embed_dim, device=x.device, dtype=torch.float32)
        ws = L.workspace(lib.cc_clip_workspace_bytes(ctypes.byref(vm), B, T, ctypes.byref(tm), Bt, Lt), x.device)
        L.check(lib.cc_clip_encode_frames(ctypes.byref(vm), ctypes.byref(fr), B, T, L.ptr(vfeat), None,
                                          ctypes.byref(tm), L.ptr(ids), Bt, Lt, L.ptr(tfeat), L.ptr(ws), ws.numel(),
                                          L.stream_ptr(x.device)), "cc_clip_encode_frames")
        return vfeat, tfeat

    def encode_text(self, text, return_hidden=False):
        """ids [B, n_ctx] -> [B, embed_dim]: EOT row of ln_final(x) @ text_projection (clip.py:471-496)."""
        if return_hidden:
            raise NotImplementedError("return_hidden=True is not built")
        L.require_device(text)
        ids = text.to(torch.long).contiguous()
        Bt, Lt = ids.shape
        m = self._text_model()
        lib = L.lib()
        out = torch.empty(Bt, self.embed_dim, device=ids.device, dtype=torch.float32)
        ws = L.workspace(lib.cc_text_workspace_bytes(ctypes.byref(m), Bt, Lt), ids.device)
        L.check(lib.cc_text_encode(ctypes.byref(m), L.ptr(ids), Bt, Lt, L.ptr(out), L.ptr(ws), ws.numel(),
                                   L.stream_ptr(ids.device)), "cc_text_encode")
        return out


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
