"""Collectives of the retrieval path over torch.distributed (backend "nccl" is RCCL on ROCm; the same code runs on
"gloo" for the CPU tests): one process per GPU, clips sharded across ranks.

* all_gather(*tensors) / AllGather: the per-step feature all-gathers + barrier of the reference
  (modules/utils.py:25-64 called at modules/clip4clip.py:351-355) packed into ONE all-gather of a byte buffer - the
  messages are <= 2 MB (latency-bound over xGMI), so fewer, larger collectives.  Values as in the reference; the
  reference also splices the INPUT tensor back into its rank's slot (utils.py:56) so that autograd reaches the local
  shard - all_gather() returns plain gathered bytes (no autograd edge); use AllGather.apply for the differentiable form.
* PackedAllGather: the differentiable form for SEVERAL tensors in one collective (the training branch's exchange).
* GradientBuckets: data-parallel gradient averaging as bucketed reduce-scatter + all-gather (main.py:124,321).
* PackedFeatures: the same exchange with NO packing step at all - the encoders write their features straight into a
  preallocated record (visual | text | mask), one all_gather_into_tensor moves the records, and the similarity kernel
  reads the gathered records in place (cc_loose_similarity_grouped_f32).
* shard_rows / gather_rows / sharded_similarity: the eval similarity matrix (main.py:502-534, rank-0 only in the
  reference) row-sharded over ranks: every rank keeps its text rows, receives all pooled video embeddings with one
  all-gather and computes its [Nt/G, Nv] row block with the HIP NT GEMM.
* gather_varlen / all_gather_ints / all_reduce_: the exchange steps of the clip-sharded evaluation loop
  (centerclip_amd/eval.py, eval_epoch(shard=True)).
"""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


def all_gather(*tensors):
    """Concatenate each tensor over ranks along dim 0 (rank order).  Tensors may differ in dtype and
    trailing shape but must have identical shapes on every rank.  Without an initialised process
    group this is the identity (world size 1).  No autograd edge reaches the inputs (AllGather.apply is the
    differentiable form, utils.py:25-44)."""
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


class AllGather(torch.autograd.Function):
    """modules/utils.py:25-44: all-gather along dim 0 whose backward hands every rank the gradient slice of its own
    shard.  Forward = all_gather above (one collective)."""

    @staticmethod
    def forward(ctx, tensor, args=None):
        ctx.rank = rank()
        ctx.batch_size = tensor.shape[0]
        return all_gather(tensor)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output[ctx.batch_size * ctx.rank: ctx.batch_size * (ctx.rank + 1)], None


class PackedAllGather(torch.autograd.Function):
    """The training branch's exchange (modules/utils.py:25-64 applied to visual_output, video_mask and sequence_output at
    modules/clip4clip.py:351-355: three all-gathers + a barrier) as ONE collective: the tensors - any dtypes, any
    trailing shapes - travel in one packed byte buffer (all_gather above); backward hands every floating-point input the
    gradient slice of its own shard (utils.py:38-44), integer inputs (the mask) get none.

        vis_all, mask_all, seq_all = PackedAllGather.apply(visual_output, video_mask, sequence_output)
    """

    @staticmethod
    def forward(ctx, *tensors):
        ctx.rank = rank()
        ctx.rows = [t.shape[0] for t in tensors]
        ctx.diff = [t.is_floating_point() for t in tensors]
        outs = all_gather(*tensors)
        outs = outs if isinstance(outs, tuple) else (outs,)
        ctx.mark_non_differentiable(*[o for o, d in zip(outs, ctx.diff) if not d])
        return outs

    @staticmethod
    def backward(ctx, *grads):
        out = []
        for g, b, d in zip(grads, ctx.rows, ctx.diff):
            out.append(g[b * ctx.rank: b * (ctx.rank + 1)] if (d and g is not None) else None)
        return tuple(out)


class GradientBuckets:
    """The gradient exchange of data-parallel training (main.py:124 wraps the model in DistributedDataParallel; its
    all-reduce averages the gradients over the ranks after scaler.scale(loss).backward(), main.py:321) as bucketed
    reduce-scatter + all-gather over RCCL:

    * the parameters' gradients are packed into flat fp32 buckets of ~`bucket_bytes` (in reverse registration order: the
      order in which backward produces them), each padded to a multiple of the world size;
    * per bucket ONE reduce_scatter_tensor (every rank receives the sum of its 1/G slice) and ONE all_gather_into_tensor of
      the averaged slices.  On a node whose GPUs are joined by point-to-point xGMI links both halves move (G-1)/G of the
      bucket per rank spread over all 7 links at once - the direct pattern - instead of a ring's G-1 dependent hops; bucket
      size is the knob that keeps each transfer above the link's latency-bound regime (25 MB default: 3.1 MB per peer at
      G = 8).  The two halves of consecutive buckets can overlap (async_op) - reduce() issues every bucket's reduce-scatter
      before waiting for the first.
    * the averaged gradients are scattered back into .grad (views of the flat buckets after the first call: no copies).
    * a parameter that NO rank produced a gradient for keeps ``.grad = None`` (as a single process and DistributedDataParallel
      leave it: the optimizer then skips it - no weight decay, no moment decay, no step count): every bucket carries one
      has-gradient flag per parameter behind its gradients, summed by the same collective; the flags are read on the host
      only on a rank that itself holds a ``None`` gradient.

    Values: the mean over ranks of every gradient, as DistributedDataParallel produces (its bucket order differs, so sums
    may differ in the last bit; asserted against an all-reduce mean in tests/test_dist_cpu.py)."""

    def __init__(self, params, bucket_bytes=25 * 1024 * 1024):
        self.params = [p for p in params if p.requires_grad]
        self.world = world_size()
        order = list(reversed(self.params))
        self.buckets = []                       # (flat buffer, [(param, offset, numel)])
        self._flags = {}                        # id(flat) -> [flag pattern, its device copy]
        cur, cur_n = [], 0
        cap = max(1, bucket_bytes // 4)
        for p in order:
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(self._make(cur, cur_n))
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(self._make(cur, cur_n))

    def _make(self, plist, n):
        pad = -(-(n + len(plist)) // self.world) * self.world      # gradients | one has-gradient flag per parameter | padding
        flat = torch.zeros(pad, dtype=torch.float32, device=plist[0].device)
        slots, off = [], 0
        for p in plist:
            slots.append((p, off, p.numel()))
            off += p.numel()
        return flat, slots

    def reduce(self):
        """Average .grad of every parameter over the ranks (in place).  A parameter without a gradient on this rank counts as
        zeros in the mean; one without a gradient on every rank keeps ``.grad = None``."""
        missing = False
        for flat, slots in self.buckets:
            flags = []
            for p, off, n in slots:
                dst = flat[off:off + n]
                flags.append(0.0 if p.grad is None else 1.0)
                if p.grad is None:
                    dst.zero_()
                    missing = True
                elif p.grad.data_ptr() != dst.data_ptr():
                    dst.copy_(p.grad.reshape(-1))
            n_grad = slots[-1][1] + slots[-1][2]
            # (the flags of a bucket normally never change between steps: one upload when the pattern changes, a device-to-
            #  device copy per step - a pageable host tensor copied every step would stall the host once per bucket)
            key = tuple(flags)
            fc = self._flags.setdefault(id(flat), [None, None])
            if fc[0] != key:
                fc[0], fc[1] = key, torch.tensor(flags, dtype=torch.float32).to(flat.device)
            flat[n_grad:n_grad + len(slots)].copy_(fc[1])
        if self.world > 1:
            works, shards = [], []
            for flat, _ in self.buckets:
                shard = torch.empty(flat.numel() // self.world, dtype=flat.dtype, device=flat.device)
                works.append(dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, async_op=True))
                shards.append(shard)
            for (flat, _), w, shard in zip(self.buckets, works, shards):
                w.wait()
                shard.div_(self.world)
                dist.all_gather_into_tensor(flat, shard)
        for flat, slots in self.buckets:
            n_grad = slots[-1][1] + slots[-1][2]
            # (the mean of the flags: > 0 where at least one rank had a gradient; a host read only where it can matter)
            have = flat[n_grad:n_grad + len(slots)].cpu().tolist() if missing else None
            for i, (p, off, n) in enumerate(slots):
                if have is not None and p.grad is None and have[i] == 0.0:
                    continue                                       # globally unused: stays None
                p.grad = flat[off:off + n].view_as(p)
        return self


class PackedFeatures:
    """One rank's record of a step's features, laid out for a single all_gather_into_tensor:

        [ visual  B*Tn*E fp32 | text  B*E fp32 | mask  B*Tn int64 | pad to 16 B ]

    ``vis`` / ``seq`` / ``mask`` are views into the send record - hand ``(vis, seq)`` to CLIP.encode_pair(out=...) so the
    projection heads write there directly, copy the segment mask into ``mask`` - then ``gather()`` returns the
    [G, rec_bytes] uint8 buffer of all ranks' records (preallocated; world size 1: a view of the send record) and
    ``logits(text)`` runs the similarity tail on it in place."""

    def __init__(self, B, Tn, E, device, world=None):
        self.B, self.Tn, self.E = B, Tn, E
        self.world = world_size() if world is None else world
        self.vis_off = 0
        self.seq_off = B * Tn * E * 4
        self.mask_off = self.seq_off + B * E * 4
        assert self.mask_off % 8 == 0
        self.rec = -(-(self.mask_off + B * Tn * 8) // 16) * 16
        self.send = torch.zeros(self.rec, dtype=torch.uint8, device=device)
        self.vis = self.send[self.vis_off:self.seq_off].view(torch.float32).view(B * Tn, E)
        self.seq = self.send[self.seq_off:self.mask_off].view(torch.float32).view(B, E)
        self.mask = self.send[self.mask_off:self.mask_off + B * Tn * 8].view(torch.long).view(B, Tn)
        self.recv = (torch.zeros(self.world * self.rec, dtype=torch.uint8, device=device) if self.world > 1
                     else self.send)
        self.bytes_per_gather = self.world * self.rec

    def gather(self):
        if self.world > 1:
            dist.all_gather_into_tensor(self.recv, self.send)
        return self.recv.view(self.world, self.rec)

    def logits(self, text, logit_scale):
        """text [Bt, E] x all gathered videos -> [Bt, G*B] (this rank's row block when text is the local text)."""
        from . import ops
        return ops.loose_similarity_packed(text, self.recv.view(self.world, self.rec), self.B, self.Tn, self.E,
                                           self.vis_off, self.mask_off, logit_scale)

    def gathered_text(self):
        """[G*B, E] text features of all ranks (a strided view of the gathered records)."""
        r = self.recv.view(self.world, self.rec)[:, self.seq_off:self.mask_off]
        return r.contiguous().view(torch.float32).view(self.world * self.B, self.E)


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


def all_gather_ints(values, device):
    """A few Python ints per rank -> list (rank order) of lists; identity without a process group."""
    if not is_dist() or dist.get_world_size() == 1:
        return [list(values)]
    world = dist.get_world_size()
    mine = torch.as_tensor(list(values), dtype=torch.long, device=device)
    out = torch.empty(world * mine.numel(), dtype=torch.long, device=device)
    dist.all_gather_into_tensor(out, mine)
    return out.view(world, -1).cpu().tolist()


def gather_varlen(rows, tags):
    """All-gather row blocks whose heights differ per rank (a rank may hold none): rows [n_r, ...], tags [n_r] (any
    integer tensor travelling with the rows, e.g. dataset positions) -> (rows of all ranks concatenated in rank order,
    their tags).  Two collectives: the heights, then one padded all_gather_into_tensor per tensor."""
    if not is_dist() or dist.get_world_size() == 1:
        return rows, tags
    world = dist.get_world_size()
    heights = [h[0] for h in all_gather_ints([rows.shape[0]], rows.device)]
    maxh = max(max(heights), 1)

    def padded_gather(t):
        pad = torch.zeros((maxh,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        out = torch.empty((world * maxh,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, pad)
        out = out.view((world, maxh) + tuple(t.shape[1:]))
        return torch.cat([out[r, :heights[r]] for r in range(world)], 0)
    return padded_gather(rows.contiguous()), padded_gather(tags.to(rows.device).contiguous())


def all_reduce_(t, op):
    """In-place all-reduce ("sum" / "max") over the default group; identity without one."""
    if is_dist() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
    return t


def sharded_similarity(text_local, pooled_video_local, n_video_total, logit_mult, dot_fn=None):
    """Row block [Nt_local, Nv] of the similarity matrix: all-gather the (pooled, normalised) video
    embeddings, then one local NT GEMM.  ``dot_fn(a, b, mult)`` defaults to the HIP exact-fp32 MFMA kernel
    (ops.scaled_dot_nt); the CPU (gloo) tests pass their own."""
    video_all = gather_rows(pooled_video_local, n_video_total)
    if dot_fn is None:
        from . import ops
        dot_fn = ops.scaled_dot_nt
    return dot_fn(text_local, video_all, logit_mult)
