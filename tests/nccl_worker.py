"""Worker of tests/test_r2_gpu.py::test_nccl_packed_all_gather_and_sharded_similarity - run by torch.distributed.run,
one process per GPU, backend nccl (= RCCL).  Checks, against data every rank can regenerate from seeds:
  * dist.PackedFeatures: ONE all_gather_into_tensor of the preallocated records, logits read in place
  * dist.all_gather / AllGather (reference semantics, modules/utils.py:25-64) incl. the own-shard backward slice
  * dist.sharded_similarity with the HIP NT GEMM == the single-rank matrix
  * eval.eval_epoch(shard=True) (clip-sharded evaluation loop, HIP kernels + the collectives) == the single-process
    eval_epoch on rank 0, single- and multi-sentence protocols
  * dist.PackedAllGather (one collective, own-shard gradient slices) and dist.GradientBuckets (reduce-scatter + all-gather
    gradient averaging) against per-tensor references
``--share-gpu``: every rank uses cuda:0 and the collectives run over gloo (a 1-GPU box can still drive world 2 through the
HIP-backed loop).  Prints NCCL_WORKER_OK world=<n> on rank 0."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def feats(rank, B, Tn, E):
    g = torch.Generator().manual_seed(100 + rank)
    vis = torch.randn(B * Tn, E, generator=g)
    seq = torch.randn(B, E, generator=g)
    mask = (torch.rand(B, Tn, generator=g) > 0.3).long()
    mask[:, 0] = 1
    return vis, seq, mask


class _Loader(list):
    pass


def sharded_eval_leg(rank, world, dev):
    """Every rank: eval_epoch(shard=True) over the same list-backed loader (batches dealt round robin); rank 0 alone: the
    single-process form (no collective).  Same R@1 and metric strings."""
    from argparse import Namespace
    import numpy as np
    from centerclip_amd.clip4clip import CLIP4Clip
    from centerclip_amd.eval import eval_epoch
    from oracle.recipes import EVAL_CASES, eval_case_batches
    g2 = np.load(os.path.join(ROOT, "tests", "golden", "r2_golden.npz"))
    sd = {k[6:]: torch.from_numpy(g2[k].astype(np.float32) if g2[k].dtype == np.float16 else g2[k])
          for k in g2.files if k.startswith("s1_sd/")}
    cfg = g2["s1_cfg"]
    T = int(cfg[11])
    a = Namespace(cluster_inter=0, deep_cluster=0, cluster_algo='kmediods++', max_frames=T, target_frames_blocks=[T, T, T],
                  cluster_num_blocks=[16, 6, 6], cluster_distance='euclidean', cluster_threshold=1e-6, cluster_iter_limit=100,
                  minkowski_norm_p=2.0, aggregation=None, pretrained_clip_name='ViT-B/32', pre_norm=False, loose_type=True,
                  sim_header='meanP', linear_patch='2d', pre_visual_pooling=0)
    model = CLIP4Clip.from_state_dict(sd, a).to(dev).eval()
    for name in sorted(EVAL_CASES):
        batches, attrs = eval_case_batches(EVAL_CASES[name], cfg)
        loader = _Loader(batches)
        loader.dataset = Namespace(**attrs)
        got = eval_epoch(model, loader, dev, shard=True)
        box = [eval_epoch(model, loader, dev) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        want = box[0]
        assert abs(got[0] - want[0]) < 1e-9 and list(got[2]) == list(want[2]), (name, got, want)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    share = "--share-gpu" in sys.argv
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if share:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    from centerclip_amd import ops
    from centerclip_amd.dist import AllGather, PackedFeatures, all_gather, gather_rows, shard_rows, sharded_similarity
    B, Tn, E = 16, 3, 512
    vis, seq, mask = feats(rank, B, Tn, E)
    pf = PackedFeatures(B, Tn, E, dev)
    assert pf.world == world
    pf.vis.copy_(vis)
    pf.seq.copy_(seq)
    pf.mask.copy_(mask)
    pf.gather()
    got = pf.logits(pf.seq, 1.5)
    allv = torch.cat([feats(r, B, Tn, E)[0].view(B, Tn, E) for r in range(world)]).to(dev)
    allm = torch.cat([feats(r, B, Tn, E)[2] for r in range(world)]).to(dev)
    alls = torch.cat([feats(r, B, Tn, E)[1] for r in range(world)]).to(dev)
    want = ops.loose_similarity(seq.to(dev), allv, allm, 1.5)
    assert got.shape == (B, world * B) and torch.equal(got, want), "packed records"
    assert torch.equal(pf.gathered_text(), alls)
    # generic packed all_gather, reference semantics
    gv, gm, gs = all_gather(vis.view(B, Tn, E).to(dev), mask.to(dev), seq.view(B, 1, E).to(dev))
    assert torch.equal(gv, allv) and torch.equal(gm, allm) and torch.equal(gs.squeeze(1), alls)
    x = seq.to(dev).clone().requires_grad_(True)
    y = AllGather.apply(x)
    y.backward(torch.arange(world * B, device=dev, dtype=torch.float32)[:, None].expand(-1, E).contiguous())
    assert torch.equal(x.grad[:, 0], torch.arange(rank * B, (rank + 1) * B, device=dev, dtype=torch.float32))
    # row-sharded eval similarity with the HIP kernel
    Nt, Nv = 1003, 257
    g = torch.Generator().manual_seed(7)
    t = torch.nn.functional.normalize(torch.randn(Nt, E, generator=g), dim=-1).to(dev)
    v = torch.nn.functional.normalize(torch.randn(Nv, E, generator=g), dim=-1).to(dev)
    t0, t1 = shard_rows(Nt)
    v0, v1 = shard_rows(Nv)
    block = sharded_similarity(t[t0:t1], v[v0:v1], Nv, 2.0)
    full = gather_rows(block, Nt)
    assert torch.equal(full, ops.scaled_dot_nt(t, v, 2.0)), "sharded similarity"
    sharded_eval_leg(rank, world, dev)
    # the training branch's exchange as one collective with the gradient slices back (N4), and the bucketed gradient
    # averaging (reduce-scatter + all-gather) against an all-reduce mean
    from centerclip_amd.dist import PackedAllGather, GradientBuckets
    xv = vis.view(B, Tn, E).to(dev).clone().requires_grad_(True)
    xs = seq.view(B, 1, E).to(dev).clone().requires_grad_(True)
    pv, pm, ps = PackedAllGather.apply(xv, mask.to(dev), xs)
    assert torch.equal(pv.detach(), allv) and torch.equal(pm, allm) and torch.equal(ps.detach().squeeze(1), alls)
    wv = torch.arange(world * B, device=dev, dtype=torch.float32)[:, None, None]
    ((pv * wv).sum() + (ps * wv).sum() * 2).backward()
    own = torch.arange(rank * B, (rank + 1) * B, device=dev, dtype=torch.float32)
    assert torch.equal(xv.grad[:, 0, 0], own) and torch.equal(xs.grad[:, 0, 0], 2 * own)
    params = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in ((768, 768), (3072,), (5, 7), (1,))]
    gb = GradientBuckets(params, bucket_bytes=1 << 20)
    gg = torch.Generator().manual_seed(500 + rank)
    want_g = []
    for p_ in params:
        p_.grad = torch.randn(p_.shape, generator=gg).to(dev)
        t_ = p_.grad.clone()
        dist.all_reduce(t_)
        want_g.append(t_ / world)
    gb.reduce()
    assert all(torch.allclose(p_.grad, w_, rtol=0, atol=1e-5) for p_, w_ in zip(params, want_g)), "gradient buckets"
    # one step of train_epoch (main.py:291-378) on the small model with the gradients averaged by GradientBuckets over RCCL:
    # every rank ends with the same parameters
    import numpy as np
    from argparse import Namespace
    from centerclip_amd.clip4clip import CLIP4Clip
    from centerclip_amd.train import BertAdam, prep_optim_params_groups, train_epoch
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_golden.npz"))
    sd = {k[3:]: torch.from_numpy(gold[k].astype(np.float32) if gold[k].dtype == np.float16 else gold[k]) for k in gold.files
          if k.startswith("sd/")}
    Tt = int(gold["cfg"][11])
    tcfg = Namespace(cluster_inter=1, cluster_algo='kmediods++', max_frames=Tt, target_frames_blocks=[4, 2, 2],
                     cluster_num_blocks=[16, 6, 6], cluster_distance='euclidean', cluster_threshold=1e-6, cluster_iter_limit=100,
                     minkowski_norm_p=2.0, pretrained_clip_name='ViT-B/32', aggregation=None, pre_norm=False, loose_type=True,
                     sim_header='meanP', linear_patch='2d')
    tm = CLIP4Clip.from_state_dict(sd, tcfg).float().to(dev)
    gv = torch.Generator().manual_seed(900 + rank)                      # a different batch on every rank
    tvideo = torch.randn(2, 1, Tt, 3, 64, 64, generator=gv)
    tids = torch.from_numpy(gold["t_ids"])[:2]
    tbatch = (tids, (tids > 0).long(), torch.zeros_like(tids), tvideo, torch.ones(2, 1, Tt, dtype=torch.long))
    targs = Namespace(lr=1e-3, wd=0.2, new_added_modules=["Cross"], gradient_accumulation_steps=1, clip_grad_norm=None)
    topt = BertAdam(prep_optim_params_groups(targs, tm), lr=targs.lr, warmup=0.1, t_total=10, schedule='warmup_cosine', b1=0.9,
                    b2=0.98, e=1e-6, max_grad_norm=1.0)
    tloss, tgs = train_epoch(0, targs, tm, [tbatch], dev, topt, 0, buckets=GradientBuckets(tm.parameters()))
    assert tgs == 1 and np.isfinite(tloss)
    flat = torch.cat([p_.detach().reshape(-1) for p_ in tm.parameters()])
    ref_flat = flat.clone()
    dist.broadcast(ref_flat, src=0)
    assert torch.equal(flat, ref_flat), "parameters differ between ranks after a data-parallel step"
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        print("NCCL_WORKER_OK world=%d backend=%s" % (world, dist.get_backend()), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
