"""Mirror of modules/losses.py:8-18 (CrossEn), forward values on the device (no backward: training is out of scope,
SURVEY §8f N4 - this is the loss a validation pass reports)."""
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
