"""Retrieval metrics from a device-resident similarity matrix - mirror of ``utils/metrics.py:11-26``
(compute_metrics) with the rank extraction done on the GPU (cc_rank_counts_f32): two int32 per row
travel to the host instead of the whole [Nt, Nv] matrix, and no sort is needed."""
import numpy as np
import torch

from . import _lib as L


def rank_counts(sim, transpose=False, diag_offset=0):
    """sim [R, C] fp32 on the device -> int32 [R', 2] (#greater, #equal incl. self) per row of sim
    (or per row of sim.T when transpose=True, without materialising the transpose)."""
    L.require_device(sim)
    assert sim.dtype == torch.float32 and sim.dim() == 2
    rows, cols = (sim.shape[1], sim.shape[0]) if transpose else sim.shape
    rs, cs = (sim.stride(1), sim.stride(0)) if transpose else (sim.stride(0), sim.stride(1))
    counts = torch.empty(rows, 2, dtype=torch.int32, device=sim.device)
    L.check(L.lib().cc_rank_counts_f32(L.ptr(sim), rows, cols, rs, cs, int(diag_offset), L.ptr(counts),
                                       L.stream_ptr(sim.device)), "cc_rank_counts_f32")
    return counts


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
