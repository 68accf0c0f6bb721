"""MI355X-native mirror of the retrieval path of ``modules/clip4clip.py`` (meanP / loose_type):
CLIP4Clip.from_pretrained (local checkpoint), forward, get_sequence_output, get_visual_output,
get_similarity_logits, _loose_similarity, get_video_mask_after_cluster - same signatures and return conventions
(SURVEY.md §8b, rows S1/S2).  All compute goes through ``torch.ops.centerclip.*``.

Training mode returns the reference's loss (feature all-gather -> logits -> symmetric CrossEn, clip4clip.py:245-262),
differentiable with respect to the features and logit_scale; the towers have no backward (encoder backward,
DDP gradient all-reduce and the optimiser are out of scope).
Not built: seqTransf / tightTransf heads, weight download.
"""
import torch
from torch import nn

from . import _lib as L
from . import ops
from . import torch_ops as T
from .clip import build_clip_model, load_clip_state_dict, zero_scalar
from . import dist as ccdist
from .dist import AllGather, PackedAllGather, all_gather
from .losses import contrastive_loss


class CLIP4Clip(nn.Module):
    def __init__(self, clip_state_dict, task_config):
        super().__init__()
        self.task_config = task_config
        self.loose_type = bool(getattr(task_config, "loose_type", True))
        self.linear_patch = getattr(task_config, "linear_patch", '2d')
        self.sim_header = getattr(task_config, "sim_header", 'meanP')
        if self.linear_patch not in ('2d', '3d'):
            raise ValueError("linear_patch must be '2d' or '3d'")
        if self.sim_header != "meanP" or not self.loose_type:
            raise NotImplementedError("only sim_header='meanP' with loose_type is built (all shipped scripts use it)")
        self.cluster_inter = getattr(task_config, "cluster_inter", 0)
        self.cluster_algo = getattr(task_config, "cluster_algo", None)
        self.deep_cluster = getattr(task_config, "deep_cluster", 0)
        self.video_frames = getattr(task_config, "max_frames", None)
        self.final_frames = task_config.target_frames_blocks[-1]
        self.f_frame_duration = self.video_frames // self.final_frames
        self.pre_visual_pooling = getattr(task_config, "pre_visual_pooling", 0)
        self.clip, self.clip_config = build_clip_model(clip_state_dict, convert_fp16=True,
                                                       linear_patch=self.linear_patch, cut_top_layer=0,
                                                       load_state_dict=True, is_eval=False,
                                                       video_frames=self.video_frames, args=task_config)

    @classmethod
    def from_state_dict(cls, clip_state_dict, task_config):
        """Build from an OpenAI-CLIP style state dict (keys without the 'clip.' prefix)."""
        return cls(clip_state_dict, task_config)

    def replica(self):
        """A second instance with the same configuration and a copy of the weights, on the same device and in the same mode.
        The fused encoders keep their folded weights and scratch per instance, so two batches can only be in flight at once
        on two instances (eval_epoch(..., in_flight=2): +15 % clips/s - the k-medoids selection and the launch tails of one
        batch run under the other's GEMMs).  Not in the reference (it evaluates one batch at a time)."""
        m = type(self)({k: v.detach() for k, v in self.clip.state_dict().items()}, self.task_config)
        m = m.to(next(self.parameters()).device)
        m.load_state_dict(self.state_dict(), strict=False)
        return m.train(self.training)

    @classmethod
    def from_pretrained(cls, cross_model_name=None, state_dict=None, cache_dir=None, type_vocab_size=2, *inputs,
                        **kwargs):
        """modules/clip4clip.py:27-123 for the built configuration (meanP, linear_patch '2d'): the CLIP weights come
        from <task_config.pretrained_dir>/ViT-B-32.pt (or ViT-B-16.pt), a fine-tuned ``state_dict`` (keys prefixed
        'clip.') overrides them, ``temperature_new`` > 1 replaces logit_scale.  ``cross_model_name`` / ``cache_dir`` /
        ``type_vocab_size`` configure the cross encoder, which only the tightTransf head uses: accepted, unused."""
        task_config = kwargs['task_config']
        if state_dict is None:
            state_dict = {}
        name = getattr(task_config, 'pretrained_clip_name', "ViT-B/32")
        clip_state_dict = load_clip_state_dict(name, pretrained_dir=task_config.pretrained_dir)
        model = cls(clip_state_dict, task_config)
        override = {k[len("clip."):]: v for k, v in state_dict.items() if k.startswith("clip.")}
        if model.linear_patch == '3d' and "visual.conv2.weight" not in override and "visual.conv2.weight" not in clip_state_dict:
            # the reference's initialisation trick (clip4clip.py:46-76): conv2 = conv1 in the centre time slice, zeros around
            w1 = override.get("visual.conv1.weight", clip_state_dict["visual.conv1.weight"]).float()
            w2 = torch.zeros_like(model.clip.visual.conv2.weight)
            w2[:, :, (w2.shape[2] - 1) // 2] = w1
            override["visual.conv2.weight"] = w2
        if override:
            model.clip.load_state_dict(override, strict=False)
        if getattr(task_config, "temperature_new", 0.0) > 1.0:
            with torch.no_grad():                    # (tracked by the packs' version keys, unlike a .data write)
                model.clip.logit_scale.fill_(task_config.temperature_new)
        return model

    # ------------------------------------------------------------------ forward (clip4clip.py:199-263)
    def forward(self, input_ids=None, token_type_ids=None, attention_mask=None, video=None, video_mask=None,
                pre_visual_pooling=False):
        output_dict = {'sequence_output': None, 'visual_output': None, 'loss': None}
        sequence_output = visual_output = None
        cluster_loss = None
        if input_ids is not None:
            input_ids = input_ids.view(-1, input_ids.shape[-1])
            if attention_mask is not None:
                attention_mask = attention_mask.view(-1, attention_mask.shape[-1])
        if video is not None:
            video = torch.as_tensor(video)
            if video.dtype != torch.uint8:           # uint8 frames go to the patch gather as they are (N3)
                video = video.float()
            b, pair, video_frame = video.shape[:3]
            video = video.reshape((-1,) + tuple(video.shape[3:]))   # [B*T, C, H, W], or [B*T, H, W, C] for uint8 HWC
            video_mask = video_mask.view(-1, video_mask.shape[-1])
            if self.cluster_inter or self.deep_cluster:
                video_mask = self.get_video_mask_after_cluster(video_mask)
        if self.training and torch.is_grad_enabled() and input_ids is not None and video is not None:
            # training (main.py:311): the towers with their backward (centerclip_amd.train), gradients reach every parameter
            from . import train as cctrain
            vfeat, cluster_loss = cctrain.encode_image_train(self.clip, video, video_frame)
            tfeat = cctrain.encode_text_train(self.clip, input_ids)
            sequence_output = tfeat.view(input_ids.size(0), -1, tfeat.size(-1))
            visual_output = vfeat.view(video_mask.size(0), -1, vfeat.size(-1))
        elif input_ids is not None and video is not None:
            # both towers in one enqueue: the text tower's blocks share their launches with the ViT's
            vfeat, tfeat = self.clip.encode_pair(video, input_ids, video_frame=video_frame)
            sequence_output = tfeat.view(input_ids.size(0), -1, tfeat.size(-1))
            visual_output = vfeat.view(video_mask.size(0), -1, vfeat.size(-1))
            cluster_loss = zero_scalar(vfeat.device)
        elif input_ids is not None:
            sequence_output = self.get_sequence_output(input_ids, token_type_ids, attention_mask)
        elif video is not None:
            visual_output, cluster_loss = self.get_visual_output(video, video_mask, video_frame=video_frame)
        output_dict['sequence_output'] = sequence_output
        if video is not None:
            if self.training or not self.pre_visual_pooling:
                output_dict['visual_output'] = visual_output
            else:                                    # clip4clip.py:238-243: normalise, masked mean, normalise
                output_dict['visual_output'] = ops.video_pool_normalize(visual_output, video_mask)
        if self.training:
            # the reference's training branch (clip4clip.py:245-262): features of all ranks -> logits -> symmetric CrossEn.
            # The loss is differentiable with respect to the features and logit_scale (losses.contrastive_loss: forward and
            # gradient in one kernel chain); the features come from the differentiable towers above.
            seq, vis, vmask = sequence_output.contiguous(), visual_output.contiguous(), video_mask.contiguous()
            if ccdist.world_size() > 1:              # ONE collective for the three tensors, gradient slices of the own shard back
                vis, vmask, seq = PackedAllGather.apply(vis, vmask, seq)
            # (logit_scale is read on the device: the optimizer changes it every step, and a host copy would be a synchronisation)
            sim_loss, _, _ = contrastive_loss(seq, vis, vmask, self.clip.logit_scale)
            output_dict['loss'] = sim_loss + cluster_loss
            output_dict['cluster_loss'] = cluster_loss
            output_dict['sim_loss'] = sim_loss
        return output_dict

    def encode_into(self, sink, input_ids, video, video_mask):
        """The multi-GPU step's producer: forward()'s two towers with the features written straight into a
        dist.PackedFeatures record (sink.vis / sink.seq; the segment mask is copied into sink.mask), so that the exchange
        of clip4clip.py:351-355 is ONE all_gather_into_tensor of preallocated buffers with no packing step."""
        input_ids = input_ids.view(-1, input_ids.shape[-1])
        video = torch.as_tensor(video)
        if video.dtype != torch.uint8:
            video = video.float()
        video_frame = video.shape[2]
        video = video.reshape((-1,) + tuple(video.shape[3:]))
        video_mask = video_mask.view(-1, video_mask.shape[-1])
        if self.cluster_inter or self.deep_cluster:
            video_mask = self.get_video_mask_after_cluster(video_mask)
        self.clip.encode_pair(video, input_ids, video_frame=video_frame, out=(sink.vis, sink.seq))
        sink.mask.copy_(video_mask)
        return sink

    def invalidate(self):
        """After a write through ``.data`` (not version-tracked): drop every cached copy of the parameters."""
        self._ls_key = None
        self.clip.invalidate()

    def _logit_scale_value(self):
        """Python float of clip.logit_scale, cached per parameter version (no device->host sync per call).  Writes through
        ``.data`` are not seen - use ``with torch.no_grad(): p.fill_()`` or call invalidate()."""
        p = self.clip.logit_scale
        key = (p.data_ptr(), p._version)
        if getattr(self, "_ls_key", None) != key:
            self._ls_key, self._ls_val = key, float(p.detach())
        return self._ls_val

    def get_sequence_output(self, input_ids, token_type_ids=None, attention_mask=None):
        """-> [bs_pair, 1, D] fp32   (clip4clip.py:265-272)"""
        bs_pair = input_ids.size(0)
        hidden = self.clip.encode_text(input_ids).float()
        return hidden.view(bs_pair, -1, hidden.size(-1))

    def get_visual_output(self, video, video_mask=None, video_frame=-1):
        """-> ([bs_pair, T_final, D] fp32, cluster_loss)   (clip4clip.py:274-281)"""
        bs_pair = video_mask.size(0)
        hidden, cluster_loss = self.clip.encode_image(video, video_frame=video_frame)
        return hidden.view(bs_pair, -1, hidden.size(-1)).float(), cluster_loss

    def get_video_mask_after_cluster(self, video_mask):
        """Mask of a segment = mask of its last frame (clip4clip.py:436-447)."""
        if self.cluster_algo in ['kmediods++', 'pooling', 'sparse_sampling', 'spectral']:
            # same columns as the reference's arange(f_frame_duration - 1, T, T // final_frames) index, as a strided
            # view: no arange / gather kernels in the step
            return video_mask[:, self.f_frame_duration - 1::video_mask.shape[-1] // self.final_frames]
        return video_mask

    def _loose_similarity(self, sequence_output, visual_output, attention_mask, video_mask):
        """exp(logit_scale) * t_hat @ v_bar^T   (clip4clip.py:324-367).  In training mode the features of all ranks are
        gathered first (:351-355) - one packed RCCL all-gather instead of three + a barrier."""
        sequence_output, visual_output = sequence_output.contiguous(), visual_output.contiguous()
        if self.training:
            visual_output, video_mask, sequence_output = all_gather(visual_output, video_mask.contiguous(),
                                                                    sequence_output)
        text = sequence_output.squeeze(1)
        scale = self._logit_scale_value()
        if visual_output.ndim == 2:      # already pooled + normalised (eval with pre_visual_pooling, :357)
            L.require_device(text, visual_output)
            return ops.scaled_dot_nt(ops.normalize_rows(text), visual_output, mult=T.logit_multiplier(scale))
        return ops.loose_similarity(text, visual_output, video_mask, scale)

    def get_similarity_logits(self, sequence_output, visual_output, attention_mask, video_mask, shaped=False):
        """-> (logits [Bt, Bv], ())   (clip4clip.py:412-434)"""
        if shaped is False:
            if attention_mask is not None:
                attention_mask = attention_mask.view(-1, attention_mask.shape[-1])
            video_mask = video_mask.view(-1, video_mask.shape[-1])
        if visual_output.ndim == 3 and video_mask.shape[1] != visual_output.shape[1]:
            video_mask = self.get_video_mask_after_cluster(video_mask)
        assert self.sim_header in ["meanP"]
        return self._loose_similarity(sequence_output, visual_output, attention_mask, video_mask), ()
