"""MI355X-native mirror of the retrieval path of ``modules/clip4clip.py`` (meanP / loose_type):
CLIP4Clip.forward (eval branch), get_sequence_output, get_visual_output, get_similarity_logits,
get_video_mask_after_cluster - same signatures and return conventions.

Not built: the training branch (loss, DDP), seqTransf / tightTransf heads,
from_pretrained's weight-download plumbing.  ``CLIP4Clip.from_state_dict`` replaces it.
"""
import torch
from torch import nn

from . import _lib as L
from . import ops
from .clip import build_clip_model
from .dist import all_gather


class CLIP4Clip(nn.Module):
    def __init__(self, clip_state_dict, task_config):
        super().__init__()
        self.task_config = task_config
        self.loose_type = bool(getattr(task_config, "loose_type", True))
        self.linear_patch = getattr(task_config, "linear_patch", '2d')
        self.sim_header = getattr(task_config, "sim_header", 'meanP')
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

    # ------------------------------------------------------------------ forward (clip4clip.py:199-263)
    def forward(self, input_ids=None, token_type_ids=None, attention_mask=None, video=None, video_mask=None,
                pre_visual_pooling=False):
        if self.training:
            raise NotImplementedError("the training branch (loss + DDP) is out of scope; call .eval()")
        output_dict = {'sequence_output': None, 'visual_output': None, 'loss': None}
        if input_ids is not None:
            input_ids = input_ids.view(-1, input_ids.shape[-1])
        if video is not None:
            video = torch.as_tensor(video)
            if video.dtype != torch.uint8:           # uint8 frames go to the patch gather as they are (N3)
                video = video.float()
            b, pair, video_frame = video.shape[:3]
            video = video.reshape((-1,) + tuple(video.shape[3:]))   # [B*T, C, H, W], or [B*T, H, W, C] for uint8 HWC
            video_mask = video_mask.view(-1, video_mask.shape[-1])
            if self.cluster_inter or self.deep_cluster:
                video_mask = self.get_video_mask_after_cluster(video_mask)
        if input_ids is not None and video is not None:
            # both towers in one enqueue: the text tower's blocks share their launches with the ViT's
            vfeat, tfeat = self.clip.encode_pair(video, input_ids, video_frame=video_frame)
            output_dict['sequence_output'] = tfeat.view(input_ids.size(0), -1, tfeat.size(-1))
            visual_output = vfeat.view(video_mask.size(0), -1, vfeat.size(-1))
        elif input_ids is not None:
            output_dict['sequence_output'] = self.get_sequence_output(input_ids, token_type_ids, attention_mask)
        elif video is not None:
            visual_output, _ = self.get_visual_output(video, video_mask, video_frame=video_frame)
        if video is not None:
            if self.pre_visual_pooling:
                visual_output = ops.video_pool_normalize(visual_output, video_mask)
            output_dict['visual_output'] = visual_output
        return output_dict

    def _logit_scale_value(self):
        """Python float of clip.logit_scale, cached per parameter version (no device->host sync per call)."""
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

    def _loose_similarity(self, sequence_output, visual_output, attention_mask, video_mask, gather=False):
        """exp(logit_scale) * t_hat @ v_bar^T   (clip4clip.py:324-367).  gather=True reproduces the
        training-time feature all-gather (one packed RCCL all-gather instead of three + barrier)."""
        sequence_output, visual_output = sequence_output.contiguous(), visual_output.contiguous()
        if gather:
            visual_output, video_mask, sequence_output = all_gather(visual_output, video_mask, sequence_output)
        text = sequence_output.squeeze(1)
        scale = self._logit_scale_value()
        if visual_output.ndim == 2:      # already pooled + normalised (pre_visual_pooling)
            L.require_device(text)
            tn = ops.scaled_dot_nt(text / text.norm(dim=-1, keepdim=True), visual_output, mult=float(torch.tensor(scale).exp()))
            return tn
        return ops.loose_similarity(text, visual_output, video_mask, scale)

    def get_similarity_logits(self, sequence_output, visual_output, attention_mask, video_mask, shaped=False):
        """-> (logits [Bt, Bv], ())   (clip4clip.py:412-434)"""
        if shaped is False:
            attention_mask = attention_mask.view(-1, attention_mask.shape[-1])
            video_mask = video_mask.view(-1, video_mask.shape[-1])
        if visual_output.ndim == 3 and video_mask.shape[1] != visual_output.shape[1]:
            video_mask = self.get_video_mask_after_cluster(video_mask)
        assert self.sim_header in ["meanP"]
        return self._loose_similarity(sequence_output, visual_output, attention_mask, video_mask), ()
