"""Mirror of modules/losses.py:8-18 (CrossEn) and of the loss of CLIP4Clip.forward's training branch
(modules/clip4clip.py:245-262), on the device.

* ``CrossEn`` / ``symmetric_contrastive_loss``: forward values from a similarity matrix (what a validation pass reports).
* ``contrastive_loss``: the training branch's loss FROM THE FEATURES, differentiable - a torch.autograd.Function whose
  backward hands out what torch.autograd derives on the reference module for sequence_output, visual_output and
  logit_scale (cc_contrastive_loss_grad_f32: one fixed-order fp32 kernel chain, forward and gradient together).
  The towers themselves have no backward in this library (encoder backward and the DDP gradient all-reduce are
  out of scope): the gradients stop at the features, where a caller's own encoder backward would pick them up.
"""
import torch
from torch import nn

from . import _lib as L
from . import torch_ops  # noqa: F401  (registers torch.ops.centerclip)


class CrossEn(nn.Module):
    """cross entropy loss: mean over rows of -log_softmax(sim_matrix, -1)[i, i]"""

    def forward(self, sim_matrix):
        L.require_device(sim_matrix)
        assert sim_matrix.dim() == 2 and sim_matrix.shape[0] == sim_matrix.shape[1]
        return torch.ops.centerclip.contrastive_loss(sim_matrix.float())[0]


def symmetric_contrastive_loss(sim_matrix):
    """(CrossEn(sim) + CrossEn(sim.T)) / 2 in one enqueue (modules/clip4clip.py:250-253) -> (sim_loss, loss1, loss2)."""
    L.require_device(sim_matrix)
    out = torch.ops.centerclip.contrastive_loss(sim_matrix.float())
    return out[2], out[0], out[1]


class _ContrastiveLoss(torch.autograd.Function):
    """(sequence_output [n,1,E] or [n,E], visual_output [n,Tn,E], video_mask [n,Tn], logit_scale 0-d tensor) ->
    (sim_loss, loss_t2v, loss_v2t).  The gradient of sim_loss is computed together with the value (it is cheap) and scaled
    by the incoming gradient in backward; loss_t2v / loss_v2t are reported values (no gradient flows through them)."""

    @staticmethod
    def forward(ctx, sequence_output, visual_output, video_mask, logit_scale, scale_value=None):
        L.require_device(sequence_output, visual_output)
        text = sequence_output.reshape(sequence_output.shape[0], -1).float().contiguous()
        vis = visual_output.float().contiguous()
        mask = video_mask.reshape(vis.shape[0], -1).to(torch.long)
        if text.shape[0] != vis.shape[0]:
            raise ValueError("the contrastive loss pairs text i with video i: %d texts, %d videos" % (text.shape[0], vis.shape[0]))
        if scale_value is None and logit_scale.is_cuda and logit_scale.dtype == torch.float32:
            # the parameter's own memory: no device -> host read (a training step stays asynchronous and can be captured)
            loss3, d_text, d_vis, d_ls = torch.ops.centerclip.contrastive_loss_grad_dev(text, vis, mask, logit_scale.detach())
        else:
            if scale_value is None:                  # (a device -> host read; callers with a cached value pass it)
                scale_value = float(logit_scale)
            loss3, d_text, d_vis, d_ls = torch.ops.centerclip.contrastive_loss_grad(text, vis, mask, float(scale_value))
        ctx.save_for_backward(d_text, d_vis, d_ls)
        ctx.shapes = (sequence_output.shape, visual_output.shape, sequence_output.dtype, visual_output.dtype)
        l_tv, l_vt, loss = loss3[0], loss3[1], loss3[2]          # bind the views once: the marks below apply to THESE objects
        ctx.mark_non_differentiable(l_tv, l_vt)
        return loss, l_tv, l_vt

    @staticmethod
    def backward(ctx, g, _g1, _g2):
        d_text, d_vis, d_ls = ctx.saved_tensors
        s_shape, v_shape, s_dtype, v_dtype = ctx.shapes
        return ((g * d_text).reshape(s_shape).to(s_dtype), (g * d_vis).reshape(v_shape).to(v_dtype), None,
                (g * d_ls).reshape(()), None)


def contrastive_loss(sequence_output, visual_output, video_mask, logit_scale, scale_value=None):
    """The loss of the reference's training branch at world size 1 - gather the features first (dist.AllGather.apply keeps
    the gradient edge) for more ranks: -> (sim_loss, CrossEn(sim), CrossEn(sim.T)), sim_loss differentiable with respect to
    sequence_output, visual_output and logit_scale."""
    if not torch.is_tensor(logit_scale):
        logit_scale = torch.tensor(float(logit_scale), device=sequence_output.device)
    return _ContrastiveLoss.apply(sequence_output, visual_output, video_mask, logit_scale, scale_value)
