"""Mirror of the evaluation loop of ``main.py``: eval_epoch (:381-499, the feature-caching half and the metrics) and
_run_on_single_gpu (:502-534, the similarity matrix).

The reference forms the [Nt, Nv] matrix as len(batch_list_t) x len(batch_list_v) small get_similarity_logits calls with a
device->host copy each (3,969 of them for MSR-VTT at batch_size_val=16).  Here the cached features are concatenated on
the device, the videos are pooled / normalised once (they were re-normalised for every text block), and the matrix is
ONE exact-fp32 MFMA NT GEMM - or, with an initialised process group, this rank's row block of it (dist.sharded_similarity,
row-sharded over the 8 GPUs of a node; the reference leaves ranks 1..7 idle during eval, main.py:232).  Every entry is
what the block-by-block loop yields: pooling and the dot product are independent per (text, video) pair.
"""
import time

import numpy as np
import torch

from . import dist as ccdist
from . import ops
from . import torch_ops as T
from .metrics import compute_metrics, tensor_text_to_video_metrics, tensor_video_to_text_sim


def _similarity_matrix(model, batch_list_t, batch_list_v, batch_sequence_output_list, batch_visual_output_list):
    """-> device tensor [Nt, Nv] (the full matrix; world > 1: computed row-sharded and gathered)."""
    if hasattr(model, 'module'):
        model = model.module
    text = torch.cat([s.reshape(s.shape[0], -1) for s in batch_sequence_output_list], 0)     # [b, 1, D] -> [b, D]
    first_v = batch_visual_output_list[0]
    if first_v.dim() == 2:                       # eval with pre_visual_pooling: already pooled + normalised
        pooled = torch.cat(batch_visual_output_list, 0)
    else:
        visual = torch.cat(batch_visual_output_list, 0)
        masks = []
        for (video_mask, *_tmp), v in zip(batch_list_v, batch_visual_output_list):
            vm = video_mask.view(-1, video_mask.shape[-1])
            if vm.shape[1] != v.shape[1]:
                vm = model.get_video_mask_after_cluster(vm)
            masks.append(vm)
        pooled = ops.video_pool_normalize(visual, torch.cat(masks, 0))
    tn = ops.normalize_rows(text)
    mult = T.logit_multiplier(model._logit_scale_value())
    world = ccdist.world_size()
    if world == 1:
        return ops.scaled_dot_nt(tn, pooled, mult)
    # every rank holds the full cached features here (as rank 0 does in the reference); shard the rows of the matrix
    s, e = ccdist.shard_rows(tn.shape[0])
    block = ops.scaled_dot_nt(tn[s:e], pooled, mult)
    return ccdist.gather_rows(block, tn.shape[0])


def _run_on_single_gpu(model, batch_list_t, batch_list_v, batch_sequence_output_list, batch_visual_output_list,
                       args=None):
    """calculate the similarity between visual output and text output -> NumPy [Nt, Nv]   (main.py:502-534)"""
    return _similarity_matrix(model, batch_list_t, batch_list_v, batch_sequence_output_list,
                              batch_visual_output_list).cpu().detach().numpy()


def eval_epoch(model, test_dataloader, device, args=None, log=None):
    """main.py:381-499: cache the features of every batch, form the similarity matrix, report R@1/5/10, MdR, MnR in both
    directions (single- and multi-sentence protocols).  -> (R1, all_infer_time, info_str).
    The metrics are extracted on the device (centerclip_amd.metrics); the matrix never travels to the host."""
    log = log or (lambda s: None)
    multi_sentence_ = False
    cut_off_points_, sentence_num_, video_num_ = [], -1, -1
    ds = test_dataloader.dataset
    if hasattr(ds, 'multi_sentence_per_video') and ds.multi_sentence_per_video:
        multi_sentence_ = True
        cut_off_points_ = [itm - 1 for itm in ds.cut_off_points]
        sentence_num_, video_num_ = ds.sentence_num, ds.video_num
        log("Eval under the multi-sentence per video clip setting.")
        log("sentence num: {}, video num: {}".format(sentence_num_, video_num_))
    core = model.module if hasattr(model, 'module') else model
    model.eval()
    with torch.no_grad():
        batch_list_t, batch_list_v = [], []
        batch_sequence_output_list, batch_visual_output_list = [], []
        total_video_num = 0
        infer_start_t = time.time()
        for bid, batch in enumerate(test_dataloader):
            batch = tuple(t.to(device) for t in batch)
            input_ids, input_mask, segment_ids, video, video_mask = batch
            if multi_sentence_:
                b, *_t = video.shape
                sequence_output = model(input_ids, segment_ids, input_mask)['sequence_output']
                batch_sequence_output_list.append(sequence_output)
                batch_list_t.append((input_mask, segment_ids,))
                s_, e_ = total_video_num, total_video_num + b
                filter_inds = [itm - s_ for itm in cut_off_points_ if itm >= s_ and itm < e_]
                if len(filter_inds) > 0:
                    video, video_mask = video[filter_inds, ...], video_mask[filter_inds, ...]
                    visual_output = model(video=video, video_mask=video_mask)['visual_output']
                    batch_visual_output_list.append(visual_output)
                    batch_list_v.append((video_mask,))
                total_video_num += b
            else:
                output = model(input_ids, segment_ids, input_mask, video, video_mask)
                batch_sequence_output_list.append(output['sequence_output'])
                batch_list_t.append((input_mask, segment_ids,))
                batch_visual_output_list.append(output['visual_output'])
                batch_list_v.append((video_mask,))
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        all_infer_time = time.time() - infer_start_t
        log('The total model inference time of the program is {:.2f} Seconds\n'.format(all_infer_time))
        if args is not None and getattr(args, "inference_speed_test", False):
            return 0
        sim = _similarity_matrix(core, batch_list_t, batch_list_v, batch_sequence_output_list, batch_visual_output_list)
    if multi_sentence_:
        log("before reshape, sim matrix size: {} x {}".format(sim.shape[0], sim.shape[1]))
        cut_off_points2len_ = [itm + 1 for itm in cut_off_points_]
        bounds = list(zip([0] + cut_off_points2len_[:-1], cut_off_points2len_))
        max_length = max(e_ - s_ for s_, e_ in bounds)
        sim3 = torch.full((len(bounds), max_length, sim.shape[1]), float("-inf"), device=sim.device)
        for g, (s_, e_) in enumerate(bounds):
            sim3[g, :e_ - s_] = sim[s_:e_]
        log("after reshape, sim matrix size: {} x {} x {}".format(*sim3.shape))
        tv_metrics = tensor_text_to_video_metrics(sim3)
        vt_metrics = compute_metrics(tensor_video_to_text_sim(sim3))
    else:
        log("sim matrix size: {}, {}".format(sim.shape[0], sim.shape[1]))
        tv_metrics = compute_metrics(sim)
        vt_metrics = compute_metrics(sim.T)
        log('\t Length-T: {}, Length-V:{}'.format(sim.shape[0], sim.shape[1]))
    info_str = ["Text-to-Video:",
                ' (metric) >>>  R@1: {:.1f} - R@5: {:.1f} - R@10: {:.1f} - Median R: {:.1f} - Mean R: {:.1f}'.format(
                    tv_metrics['R1'], tv_metrics['R5'], tv_metrics['R10'], tv_metrics['MR'], tv_metrics['MeanR']),
                "Video-to-Text:",
                ' (metric) >>>  V2T$R@1: {:.1f} - V2T$R@5: {:.1f} - V2T$R@10: {:.1f} - V2T$Median R: {:.1f} - '
                'V2T$Mean R: {:.1f}'.format(vt_metrics['R1'], vt_metrics['R5'], vt_metrics['R10'], vt_metrics['MR'],
                                            vt_metrics['MeanR'])]
    for info in info_str:
        log(info)
    return tv_metrics['R1'], all_infer_time, info_str


__all__ = ["eval_epoch", "_run_on_single_gpu", "np"]
