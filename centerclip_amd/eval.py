"""Mirror of the evaluation loop of ``main.py``: eval_epoch (:381-499 - feature caching, similarity matrix, retrieval
metrics of both directions, single- and multi-sentence protocols) and _run_on_single_gpu (:502-534).

Single process (default, ``shard=False`` - what the reference does: it evaluates on rank 0 only, main.py:232,251).  The
reference forms the [Nt, Nv] matrix as (#text batches x #video batches) small get_similarity_logits calls with a
device->host copy each (3,969 of them for MSR-VTT at batch_size_val=16).  Here the videos are pooled / normalised once
per batch as they are cached, the cache is concatenated on the device and the matrix is ONE NT GEMM; the metrics are
extracted on the device (centerclip_amd.metrics).  No collective is ever issued in this mode, so a ported main.py that
keeps ``if is_master(): eval_epoch(...)`` does not deadlock under DDP.

Clip-sharded (``shard=True``, EVERY rank calls it).  The encoders are > 99.9 % of the evaluation, so the
dataset - not the GEMM - is what is sharded:
  * batches are dealt round robin (batch b -> rank b % G) when every rank iterates the same loader, or taken as they come
    when the loader carries a ``DistributedSampler(shuffle=False)`` (its padding duplicates are dropped);
  * every rank keeps its text rows; the pooled + normalised video rows ([Nv, E], 2 MB for 1k videos) are all-gathered
    once and put in dataset order;
  * each rank forms its [Nt/G, Nv] row block with the HIP NT GEMM and ranks its rows on the device;
  * text->video: the per-row counts (3 ints per sentence) are all-gathered;  video->text: the ground-truth entry of a
    column lives in one rank's block - its values are summed into place with one all-reduce, every rank counts its rows
    against them (cc_rank_counts_ref_f32) and the counts are all-reduced (multi-sentence: the per-(group, video) maxima
    are all-reduced with MAX instead, utils/metrics.py:68-76).
The [Nt, Nv] matrix never exists on one device and never travels.  A sharded loader in the single-process mode, or
ranks disagreeing on the dataset size, raise instead of returning a silently wrong matrix.
"""
import contextlib
import time

import numpy as np
import torch

from . import dist as ccdist
from . import ops
from . import torch_ops as T
from .metrics import metrics_from_counts, multi_sentence_metrics_from_counts


class HipBackend:
    """The device operations of the loop (the product path).  tests/test_dist_cpu.py swaps in a torch-CPU stand-in to run
    the sharding / collective logic over gloo without a GPU."""
    normalize_rows = staticmethod(ops.normalize_rows)
    pool_normalize = staticmethod(ops.video_pool_normalize)
    dot_nt = staticmethod(ops.scaled_dot_nt)

    # The operands of the final matrix as by-products of encoding a batch (split-fp16 planes, [rows, 3E] fp16): what the
    # epoch's last step multiplies is then ONE GEMM launch.  A backend's "operand rows" are opaque to the loop: it only
    # concatenates, scatters and (sharded) all-gathers them.
    @staticmethod
    def text_operand(feats):
        return torch.ops.centerclip.normalize_rows_planes(feats.float().contiguous(), False)

    @staticmethod
    def video_operand(visual_output, mask):
        if visual_output.dim() == 2:                               # pooled + normalised already (pre_visual_pooling)
            return torch.ops.centerclip.normalize_rows_planes(visual_output.float().contiguous(), True)
        return torch.ops.centerclip.video_pool_normalize_planes(visual_output.float().contiguous(), mask.to(torch.long).contiguous())

    video_operand_rows = staticmethod(T.padded_video_rows)          # rows to allocate (zeroed) for n videos

    # fp16 products per multiply-add of the final matrix (HipBackend.with_products).  2 = fp16(text) x the video operand to
    # 22 bits: 5e-5 at the worst entry of a 10k x 1k cosine matrix against float64 (1e-5 rms) - 20x inside the 1e-3 the
    # contract asks of similarities, and the towers' own fp16-operand error is larger - for 2/3 of the matrix-core work
    # (29.8 vs 38.4 us at 10k x 1k).  3 (both operands to 22 bits, 2e-7) on request: eval_epoch(similarity_products=3).
    similarity_products = 2

    @classmethod
    def dot_operands(cls, text_op, video_op, n_video, mult):
        return torch.ops.centerclip.scaled_dot_planes(text_op.contiguous(), video_op, int(n_video), float(mult),
                                                      int(cls.similarity_products))

    @classmethod
    def with_products(cls, products):
        """A backend whose final matrix issues `products` (3, 2 or 1) of the three fp16 products per multiply-add: 3 = both
        operands to 22 bits (the reference's fp32 product to its own rounding), 2 = the text side rounded to fp16 (the
        default), 1 = both sides fp16 - 2/3 and 1/3 of the GEMM's work, ~1e-5 rms on a cosine, inside the contract's 1e-3
        (cc_scaled_dot_planes_products_f32; ``eval_epoch(..., similarity_products=3)``)."""
        if products not in (1, 2, 3):
            raise ValueError("similarity products: 1, 2 or 3")
        return type("HipBackendP%d" % products, (cls,), {"similarity_products": int(products)})

    @staticmethod
    def counts_cols(sim, gt_cols):
        return torch.ops.centerclip.rank_counts_cols(sim.contiguous(), gt_cols)

    @staticmethod
    def counts_ref_columns(sim, ref_vals):
        return torch.ops.centerclip.rank_counts_ref(sim, ref_vals.contiguous(), True)

    @staticmethod
    def group_max(sim, groups, n_groups):
        return torch.ops.centerclip.group_max_rows(sim.contiguous(), groups.to(torch.int32).contiguous(), int(n_groups))


class _Cache:
    """Features of the items this process encoded, with their positions in the dataset."""

    def __init__(self):
        self.text, self.text_pos, self.video, self.video_pos = [], [], [], []

    def add_text(self, feats, pos):
        self.text.append(feats.reshape(feats.shape[0], -1))
        self.text_pos.append(pos)

    def add_video(self, rows, pos):
        self.video.append(rows)
        self.video_pos.append(pos)

    @staticmethod
    def _cat(parts, width, device, dtype):
        return torch.cat(parts, 0) if parts else torch.zeros((0, width), device=device, dtype=dtype)


def _video_operand(core, visual_output, video_mask, be):
    """[b, T', E] per-segment features + the loader's mask (or [b, E] pooled rows) -> the video-side operand rows of the
    final matrix (clip4clip.py:305-316,357-360 folded into their production)."""
    if visual_output.dim() == 2:
        return be.video_operand(visual_output, None)
    vm = video_mask.view(-1, video_mask.shape[-1])
    if vm.shape[1] != visual_output.shape[1]:
        vm = core.get_video_mask_after_cluster(vm)
    return be.video_operand(visual_output.contiguous(), vm.contiguous())


def _similarity_matrix(model, batch_list_t, batch_list_v, batch_sequence_output_list, batch_visual_output_list,
                       backend=HipBackend):
    """-> device tensor [Nt, Nv] from the cached per-batch lists (this process' cache only; no collective)."""
    core = model.module if hasattr(model, 'module') else model
    text = torch.cat([backend.text_operand(s.reshape(s.shape[0], -1)) for s in batch_sequence_output_list], 0)
    vids = torch.cat([_video_operand(core, v, masks[0], backend) for v, masks in zip(batch_visual_output_list, batch_list_v)], 0)
    n_video = vids.shape[0]
    video_all = torch.zeros(max(n_video, backend.video_operand_rows(n_video)), vids.shape[1], device=vids.device, dtype=vids.dtype)
    video_all[:n_video] = vids
    mult = T.logit_multiplier(core._logit_scale_value())
    return backend.dot_operands(text, video_all, n_video, mult)


def _run_on_single_gpu(model, batch_list_t, batch_list_v, batch_sequence_output_list, batch_visual_output_list,
                       args=None):
    """calculate the similarity between visual output and text output -> NumPy [Nt, Nv]   (main.py:502-534)"""
    return _similarity_matrix(model, batch_list_t, batch_list_v, batch_sequence_output_list,
                              batch_visual_output_list).cpu().detach().numpy()


def _is_distributed_sampler(loader):
    from torch.utils.data.distributed import DistributedSampler
    return isinstance(getattr(loader, "sampler", None), DistributedSampler)


def _item_positions(loader, world, rank, shard):
    """-> (fn(batch index, batch size) -> LongTensor of dataset positions or None when the batch is another rank's,
           dataset length or None)."""
    if _is_distributed_sampler(loader):
        if not shard:
            raise RuntimeError("eval_epoch: the loader carries a DistributedSampler (every rank sees a different shard) but "
                               "shard=False forms the matrix from this process' cache alone - pass shard=True on every "
                               "rank, or evaluate on one rank with an unsharded loader as the reference does")
        sampler = loader.sampler
        if getattr(sampler, "shuffle", False):
            raise RuntimeError("eval_epoch(shard=True) needs DistributedSampler(shuffle=False): retrieval metrics pair row i "
                               "with video i")
        order = torch.as_tensor(list(iter(sampler)), dtype=torch.long)
        n = len(sampler.dataset)
        seen = [0]

        def positions(bid, b):
            k = torch.arange(seen[0], seen[0] + b)
            seen[0] += b
            pos = order[k]
            pos[k * sampler.num_replicas + sampler.rank >= n] = -1        # the sampler's padding repeats the head
            return pos
        return positions, n
    seen = [0]

    def positions(bid, b):
        start = seen[0]
        seen[0] += b
        if shard and bid % world != rank:
            return None
        return torch.arange(start, start + b)
    return positions, None


class _GraphedLane:
    """One lane of eval_epoch(graphed=True): per batch signature (shapes + dtypes of the loader's five tensors) a set of
    device-resident input tensors, and a hipGraph of `forward -> operand rows` captured on them.  A batch is then ONE
    host -> device copy per tensor into the resident inputs + one graph launch (~70 kernel launches of the eager step leave
    the host's critical path); the operand rows are cloned out of the graph's buffers before the next replay reuses them."""

    def __init__(self, net, core, be, stream):
        self.net, self.core, self.be, self.stream, self.graphs = net, core, be, stream, {}

    def _step(self, bufs):
        input_ids, input_mask, segment_ids, video, video_mask = bufs
        out = self.net(input_ids, segment_ids, input_mask, video, video_mask)
        seq = out['sequence_output']
        return (self.be.text_operand(seq.reshape(seq.shape[0], -1)), _video_operand(self.core, out['visual_output'], video_mask, self.be))

    def __call__(self, batch, device):
        key = tuple((tuple(t.shape), t.dtype) for t in batch)
        entry = self.graphs.get(key)
        if entry is None:                                         # first batch of this signature: eager (warm-up), then capture
            bufs = tuple(t.to(device) for t in batch)
            result = tuple(o.clone() for o in self._step(bufs))
            torch.cuda.current_stream(device).synchronize()
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, stream=self.stream):
                outs = self._step(bufs)
            self.graphs[key] = (gph, bufs, outs)
            return result
        gph, bufs, outs = entry
        for d, h in zip(bufs, batch):
            d.copy_(h, non_blocking=True)                          # (asynchronous from pinned memory; ordered before the replay)
        gph.replay()
        return tuple(o.clone() for o in outs)


def eval_epoch(model, test_dataloader, device, args=None, log=None, shard=False, backend=HipBackend, in_flight=None,
               similarity_products=None, graphed=False):
    """main.py:381-499 -> (R1, all_infer_time, info_str).  ``shard=True``: clip-sharded over the ranks of the default
    process group (module docstring) - every rank must call it and every rank returns the same numbers.
    ``in_flight`` (not in the reference; > 1 GPU only; default: 2 on a GPU, 1 on the CPU stand-in): n keeps n batches in flight -
    batch b runs on instance b % n of the model (``CLIP4Clip.replica()``) on a stream of its own, so the host -> device copy and
    the small-grid kernels of one batch (k-medoids selection, launch tails) run under another's GEMMs: from pinned uint8 host
    batches of the cfg-2 shape 2.82 -> 1.82 ms per 16-clip batch (5,680 -> 8,780 clips/s; 4 lanes: 9,530), identical features.
    ``similarity_products`` (not in the reference): 3, 2 or 1 fp16 products per multiply-add of the final matrix
    (``HipBackend.with_products``); None = the backend's own (2: the text side rounded to fp16, the video side to 22 bits).
    ``graphed`` (not in the reference; GPU only, single-sentence protocol): every full batch is a copy into device-resident
    inputs + ONE hipGraph launch per lane (captured on the first batch of each shape; a ragged last batch gets a graph of its
    own) instead of ~70 eager kernel launches - identical rows.  Measured (same loader): one lane 2.82 -> 2.41 ms per batch; with
    two or more lanes the eager launches already overlap and the graphs add nothing (2.15 vs 1.82 ms at two lanes), so it is
    for the case where a second model instance is not wanted (``in_flight=1``)."""
    log = log or (lambda s: None)
    if in_flight is None:
        in_flight = 2 if (torch.cuda.is_available() and torch.device(device).type == "cuda") else 1
    be = backend.with_products(similarity_products) if similarity_products is not None else backend
    world, rank = (ccdist.world_size(), ccdist.rank()) if shard else (1, 0)
    ds = test_dataloader.dataset
    multi = bool(getattr(ds, 'multi_sentence_per_video', False))
    last_sentence = None                   # multi-sentence protocol: dataset position of the last sentence of each video
    if multi:
        last_sentence = [c - 1 for c in ds.cut_off_points]
        log("Eval under the multi-sentence per video clip setting.")
        log("sentence num: {}, video num: {}".format(ds.sentence_num, ds.video_num))
        video_of_last = {p: v for v, p in enumerate(last_sentence)}
    core = model.module if hasattr(model, 'module') else model
    positions, n_items = _item_positions(test_dataloader, world, rank, shard)
    cache = _Cache()
    model.eval()
    lanes = None                               # in_flight > 1: [(model instance, stream)], batch b on lane b % in_flight
    if in_flight > 1:
        if not (torch.cuda.is_available() and torch.device(device).type == "cuda"):
            raise ValueError("eval_epoch(in_flight > 1) needs a GPU")
        here = torch.cuda.current_stream(device)
        # the extra instances (weights, folded weights, encoder scratch) and the lane streams are kept on the model and reused by
        # later calls as long as no parameter changed (a version key); a changed model builds fresh ones
        key = (in_flight, str(torch.device(device)), tuple((p.data_ptr(), p._version) for p in core.parameters()))
        kept = getattr(core, "_eval_lanes", None)
        if kept is None or kept[0] != key:
            kept = (key, [torch.cuda.Stream(device) for _ in range(in_flight)], [core.replica().eval() for _ in range(in_flight - 1)])
            core._eval_lanes = kept
        lanes = list(zip([model] + kept[2], kept[1]))
        for _, s_ in lanes:
            s_.wait_stream(here)
    graphs, graph_stream = None, None
    if graphed:
        if not (torch.cuda.is_available() and torch.device(device).type == "cuda"):
            raise ValueError("eval_epoch(graphed=True) needs a GPU")
        # lane index -> _GraphedLane, kept on the model next to the lanes (same validity key: the weights' versions)
        gkey = (in_flight, str(torch.device(device)), be.text_operand, be.video_operand,
                tuple((p.data_ptr(), p._version) for p in core.parameters()))
        kept_g = getattr(core, "_eval_graphs", None)
        if kept_g is None or kept_g[0] != gkey:
            kept_g = (gkey, {}, torch.cuda.Stream(device))
            core._eval_graphs = kept_g
        graphs, graph_stream = kept_g[1], kept_g[2]
    t_start = time.time()
    total = 0
    with torch.no_grad():
        for bid, batch in enumerate(test_dataloader):
            b = batch[0].shape[0]
            pos = positions(bid, b)
            total += b
            if pos is None:
                continue
            keep = (pos >= 0).nonzero().flatten()
            if keep.numel() == 0:
                continue
            if keep.numel() < b:                       # (only a DistributedSampler's padded tail)
                batch, pos = tuple(t[keep] for t in batch), pos[keep]
            net, lane_stream = lanes[bid % in_flight] if lanes else (model, None)
            with (torch.cuda.stream(lane_stream) if lane_stream is not None else contextlib.nullcontext()):
                def keep_rows(t):
                    # rows produced on a lane's stream are read on the caller's stream at the end: tell the caching allocator,
                    # so their blocks are not handed out again on the lane while that read is still queued
                    if lane_stream is not None and t.is_cuda:
                        t.record_stream(here)
                    return t
                if graphs is not None and not multi:
                    lane = graphs.setdefault(bid % in_flight, _GraphedLane(net, core, be, lane_stream if lane_stream is not None
                                                                           else graph_stream))
                    t_op, v_op = lane(batch, device)
                    cache.add_text(keep_rows(t_op), pos)
                    cache.add_video(keep_rows(v_op), pos)
                    continue
                input_ids, input_mask, segment_ids, video, video_mask = (t.to(device) for t in batch)
                if not multi:
                    out = net(input_ids, segment_ids, input_mask, video, video_mask)
                    cache.add_text(keep_rows(be.text_operand(out['sequence_output'].reshape(b if keep.numel() == b else keep.numel(), -1))), pos)
                    cache.add_video(keep_rows(_video_operand(core, out['visual_output'], video_mask, be)), pos)
                    continue
                seq = net(input_ids, segment_ids, input_mask)['sequence_output']
                cache.add_text(keep_rows(be.text_operand(seq.reshape(seq.shape[0], -1))), pos)
                rows = [i for i, p in enumerate(pos.tolist()) if p in video_of_last]      # items that carry their clip's video
                if rows:
                    vout = net(video=video[rows, ...], video_mask=video_mask[rows, ...])['visual_output']
                    cache.add_video(keep_rows(_video_operand(core, vout, video_mask[rows, ...], be)),
                                    torch.as_tensor([video_of_last[int(pos[i])] for i in rows], dtype=torch.long))
        if torch.cuda.is_available():
            torch.cuda.synchronize()               # (also joins the lanes' streams: the cached rows are read on this one below)
        all_infer_time = time.time() - t_start
        log('The total model inference time of the program is {:.2f} Seconds\n'.format(all_infer_time))
        if args is not None and getattr(args, "inference_speed_test", False):
            return 0
        n_text = n_items if n_items is not None else total
        n_video = len(last_sentence) if multi else n_text
        tv_metrics, vt_metrics, shape = _sharded_metrics(core, cache, n_text, n_video, last_sentence, device, world, be)
    if multi:
        log("sim matrix size: {} sentences x {} videos ({} groups)".format(shape[0], shape[1], n_video))
    else:
        log("sim matrix size: {}, {}".format(*shape))
        log('\t Length-T: {}, Length-V:{}'.format(*shape))
    fmt = ' (metric) >>>  {p}R@1: {:.1f} - {p}R@5: {:.1f} - {p}R@10: {:.1f} - {p}Median R: {:.1f} - {p}Mean R: {:.1f}'
    keys = ('R1', 'R5', 'R10', 'MR', 'MeanR')
    info_str = ["Text-to-Video:", fmt.format(*[tv_metrics[k] for k in keys], p=""),
                "Video-to-Text:", fmt.format(*[vt_metrics[k] for k in keys], p="V2T$")]
    for line in info_str:
        log(line)
    return tv_metrics['R1'], all_infer_time, info_str


def _sharded_metrics(core, cache, n_text, n_video, last_sentence, device, world, be):
    """Row block of the similarity matrix for this process' text rows + the rank metrics of both directions.  world == 1:
    the block is the whole matrix and no collective runs.  -> (tv_metrics, vt_metrics, (Nt, Nv))."""
    text_pos = torch.cat(cache.text_pos) if cache.text_pos else torch.zeros(0, dtype=torch.long)
    video_pos = torch.cat(cache.video_pos) if cache.video_pos else torch.zeros(0, dtype=torch.long)
    # the cached rows are the backend's operand rows of the final product (HIP: split-fp16 planes, [rows, 3E] fp16)
    parts = cache.text + cache.video
    E = parts[0].shape[-1] if parts else 0
    dtype = parts[0].dtype if parts else torch.float32
    if world > 1:                                       # ranks must agree on the geometry before any payload moves
        sizes = ccdist.all_gather_ints([n_text, n_video, E, len(text_pos), len(video_pos)], device)
        if len({tuple(s[:2]) for s in sizes}) != 1:
            raise RuntimeError("eval_epoch(shard=True): ranks disagree on the dataset size %s" % (sizes,))
        E = max(s[2] for s in sizes)
        if sum(s[3] for s in sizes) != n_text or sum(s[4] for s in sizes) != n_video:
            raise RuntimeError("eval_epoch(shard=True): the ranks' shards do not add up to the dataset (%s) - every rank "
                               "must iterate the same unsharded loader or a DistributedSampler(shuffle=False)" % (sizes,))
    text = _Cache._cat(cache.text, E, device, dtype)
    local_video = _Cache._cat(cache.video, E, device, dtype)
    # ---- exchange step: the video operand rows of every rank, in dataset order (zero rows pad up to whole GEMM tiles)
    video_all = torch.zeros(max(n_video, be.video_operand_rows(n_video)), E, device=device, dtype=dtype)
    if world > 1:
        rows, pos = ccdist.gather_varlen(local_video, video_pos.to(device))
        video_all[pos] = rows
    else:
        video_all[video_pos.to(device)] = local_video
    text_pos = text_pos.to(device)
    if last_sentence is None:
        gt_cols = text_pos
    else:                                               # sentence at position p describes video #{cut points < p}
        bounds = torch.as_tensor(last_sentence, device=device)
        gt_cols = torch.searchsorted(bounds, text_pos)
    mult = T.logit_multiplier(core._logit_scale_value())
    nloc = text.shape[0]
    block = be.dot_operands(text, video_all, n_video, mult) if nloc else torch.zeros(0, n_video, device=device)
    # ---- text -> video: rank of the ground-truth column in every local row
    c3 = be.counts_cols(block, gt_cols.to(torch.int32)) if nloc else torch.zeros(0, 3, dtype=torch.int32, device=device)
    truth = block.gather(1, gt_cols.view(-1, 1)).squeeze(1) if nloc else torch.zeros(0, device=device)
    rec = torch.cat([text_pos.view(-1, 1).to(torch.float64), c3.to(torch.float64), truth.view(-1, 1).to(torch.float64)], 1)
    if world > 1:
        rec, _ = ccdist.gather_varlen(rec, text_pos)
    order = torch.argsort(rec[:, 0])
    rec = rec[order].cpu().numpy()
    counts_tv, truth_all = rec[:, 1:4].astype(np.int64), rec[:, 4]
    if last_sentence is None:
        tv = metrics_from_counts(counts_tv[:, :2])
        # ---- video -> text: column v against its ground-truth entry sim[v, v], owned by the rank that holds text row v
        ref = torch.zeros(n_video, device=device)
        ref[text_pos] = truth
        if world > 1:
            ccdist.all_reduce_(ref, "sum")
        c2 = (be.counts_ref_columns(block, ref) if nloc else torch.zeros(n_video, 2, dtype=torch.int32, device=device)).long()
        if world > 1:
            ccdist.all_reduce_(c2, "sum")
        vt = metrics_from_counts(c2)
    else:
        tv = multi_sentence_metrics_from_counts(counts_tv, np.isfinite(truth_all))
        # ---- video -> text: best sentence of every group per video (utils/metrics.py:68-76), then ranks of the diagonal
        best = be.group_max(block if nloc else block.new_zeros((0, n_video)), gt_cols, n_video)      # [group, video]
        if world > 1:
            ccdist.all_reduce_(best, "max")
        vt = metrics_from_counts(be.counts_cols(best.t().contiguous(), torch.arange(n_video, dtype=torch.int32, device=device))[:, :2])
    return tv, vt, (n_text, n_video)


__all__ = ["eval_epoch", "_run_on_single_gpu", "HipBackend", "np"]
