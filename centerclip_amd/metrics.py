"""Retrieval metrics from a device-resident similarity matrix - mirror of ``utils/metrics.py`` (compute_metrics
:11-26, tensor_text_to_video_metrics :38-65, tensor_video_to_text_sim :68-76) with the rank extraction done on
the GPU (cc_rank_counts_f32 / cc_rank_counts_cols_f32): a few int32 per row travel to the host instead of the whole
[Nt, Nv] matrix, and no sort is needed."""
import numpy as np
import torch

from . import _lib as L
from . import torch_ops  # noqa: F401  (registers torch.ops.centerclip)


def rank_counts(sim, transpose=False, diag_offset=0):
    """sim [R, C] fp32 on the device -> int32 [R', 2] (#greater, #equal incl. self) per row of sim
    (or per row of sim.T when transpose=True, without materialising the transpose)."""
    L.require_device(sim)
    assert sim.dtype == torch.float32 and sim.dim() == 2
    return torch.ops.centerclip.rank_counts(sim, bool(transpose), int(diag_offset))


def metrics_from_counts(counts):
    """The dict compute_metrics returns, from per-row (#greater, #equal) counts."""
    c = np.asarray(counts.cpu() if torch.is_tensor(counts) else counts, dtype=np.int64)
    ind = np.concatenate([np.arange(g, g + e) for g, e in c]) if len(c) else np.zeros(0, dtype=np.int64)
    m = {}
    m['R1'] = float(np.sum(ind == 0)) * 100 / len(ind)
    m['R5'] = float(np.sum(ind < 5)) * 100 / len(ind)
    m['R10'] = float(np.sum(ind < 10)) * 100 / len(ind)
    m['MR'] = np.median(ind) + 1
    m["MedianR"] = m['MR']
    m["MeanR"] = np.mean(ind) + 1
    m["cols"] = [int(i) for i in list(ind)]
    return m


def compute_metrics(x):
    """Drop-in for utils.metrics.compute_metrics on a CUDA/ROCm tensor: text->video when called on sim,
    video->text when called on sim.T (a transposed VIEW is fine - strides are honoured)."""
    if not torch.is_tensor(x):
        raise TypeError("centerclip_amd.metrics.compute_metrics ranks a device tensor; the reference's NumPy version "
                        "handles host arrays")
    return metrics_from_counts(rank_counts(x))


def tensor_text_to_video_metrics(sim_tensor, top_k=(1, 5, 10)):
    """Drop-in for utils.metrics.tensor_text_to_video_metrics (:38-65) on a device tensor.
    sim_tensor [G, Lmax, C]: group g holds the sentences of video g, padded with -inf rows (main.py:466-476).
    The reference double-argsorts every row and reads the rank of the ground-truth column off the diagonal; here the
    rank of sentence (g, l) is #{j: sim[g,l,j] > sim[g,l,g]} plus its position among ties under a stable descending
    sort (the reference's argsort leaves the order of exact ties open).  Padding rows (non-finite ground truth) are
    dropped, as in the reference."""
    if not torch.is_tensor(sim_tensor):
        raise TypeError("centerclip_amd.metrics ranks device tensors; the reference's version handles host arrays")
    L.require_device(sim_tensor)
    x = sim_tensor.float().contiguous()
    G, Lmax, C = x.shape
    gt = torch.arange(G, dtype=torch.int32, device=x.device).repeat_interleave(Lmax).contiguous()
    counts = torch.ops.centerclip.rank_counts_cols(x.view(G * Lmax, C), gt)
    truth = x.reshape(G * Lmax, C).gather(1, gt.long().unsqueeze(1)).squeeze(1)
    valid = torch.isfinite(truth).cpu().numpy()
    return multi_sentence_metrics_from_counts(counts.cpu().numpy(), valid, top_k)


def multi_sentence_metrics_from_counts(counts3, valid, top_k=(1, 5, 10)):
    """The dict tensor_text_to_video_metrics returns, from per-sentence (#greater, #equal, #equal before) counts and the
    mask of real (non-padding, finite ground truth) sentences; the order of the sentences does not matter."""
    c = np.asarray(counts3).astype(np.int64)
    ranks = (c[:, 0] + c[:, 2])[np.asarray(valid, dtype=bool)]
    # the reference divides torch tensors here, i.e. in fp32 (metrics.py:58)
    res = {"R%d" % k: float(np.float32(np.sum(ranks < k) * 100) / np.float32(len(ranks))) for k in top_k}
    res["MedianR"] = float(np.sort(ranks + 1)[(len(ranks) - 1) // 2])          # torch.median: the lower middle value
    res["MeanR"] = float(np.mean(ranks + 1))
    res["Std_Rank"] = float(np.std(ranks + 1))
    res["MR"] = res["MedianR"]
    return res


def tensor_video_to_text_sim(sim_tensor):
    """Drop-in for utils.metrics.tensor_video_to_text_sim (:68-76): NaN -> -inf, max over the sentences of a group,
    transposed -> [C, G], ready for compute_metrics.  Stays on the device."""
    L.require_device(sim_tensor)
    x = sim_tensor.float()
    x = torch.where(x != x, torch.full_like(x, float("-inf")), x)
    values, _ = torch.max(x, dim=1, keepdim=True)
    return torch.squeeze(values).T.contiguous()
