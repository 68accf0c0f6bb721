"""Collectives of the retrieval path over torch.distributed (backend "nccl" is RCCL on ROCm; the same
code runs on "gloo" for the CPU tests): one process per GPU, clips sharded across ranks.

* all_gather(*tensors): the three per-step feature all-gathers + barrier of the reference
  (modules/utils.py:47-64 called at modules/clip4clip.py:351-355) packed into ONE all-gather of a
  byte buffer - the messages are <= 2 MB (latency-bound over xGMI), so fewer, larger collectives.
* shard_rows / sharded_similarity: the eval similarity matrix (main.py:502-534, rank-0 only in the
  reference) row-sharded over ranks: every rank keeps its text rows, receives all pooled video
  embeddings with one all-gather and computes its [Nt/G, Nv] row block.
"""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def all_gather(*tensors):
    """Concatenate each tensor over ranks along dim 0 (rank order).  Tensors may differ in dtype and
    trailing shape but must have identical shapes on every rank.  Without an initialised process
    group this is the identity (world size 1)."""
    if not is_dist() or dist.get_world_size() == 1:
        return tensors if len(tensors) > 1 else tensors[0]
    world = dist.get_world_size()
    flat = [t.contiguous().view(torch.uint8).reshape(-1) for t in tensors]
    sizes = [f.numel() for f in flat]
    packed = torch.cat(flat)
    gathered = torch.empty(world * packed.numel(), dtype=torch.uint8, device=packed.device)
    dist.all_gather_into_tensor(gathered, packed)
    gathered = gathered.view(world, -1)
    outs, off = [], 0
    for t, sz in zip(tensors, sizes):
        part = gathered[:, off:off + sz].contiguous().view(t.dtype).reshape((world * t.shape[0],) + tuple(t.shape[1:]))
        outs.append(part)
        off += sz
    return tuple(outs) if len(outs) > 1 else outs[0]


def shard_rows(n, rank=None, world=None):
    """Contiguous [start, stop) of n rows owned by this rank (DistributedSampler-style, no padding)."""
    if world is None:
        world = dist.get_world_size() if is_dist() else 1
    if rank is None:
        rank = dist.get_rank() if is_dist() else 0
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_rows(local_rows, n_total):
    """All-gather row blocks of possibly unequal height (shard_rows layout) -> [n_total, ...]."""
    if not is_dist() or dist.get_world_size() == 1:
        return local_rows
    world = dist.get_world_size()
    maxh = -(-n_total // world)
    pad = torch.zeros((maxh,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
    pad[:local_rows.shape[0]] = local_rows
    out = torch.empty((world * maxh,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(out, pad)
    out = out.view((world, maxh) + tuple(local_rows.shape[1:]))
    parts = []
    for r in range(world):
        s, e = shard_rows(n_total, r, world)
        parts.append(out[r, :e - s])
    return torch.cat(parts, 0)


def sharded_similarity(text_local, pooled_video_local, n_video_total, logit_mult, dot_fn):
    """Row block [Nt_local, Nv] of the similarity matrix: all-gather the (pooled, normalised) video
    embeddings, then one local NT GEMM.  ``dot_fn(a, b, mult)`` is the device kernel."""
    video_all = gather_rows(pooled_video_local, n_video_total)
    return dot_fn(text_local, video_all, logit_mult)
