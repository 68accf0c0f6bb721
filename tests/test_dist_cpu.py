"""N>1 path on CPU: world_size-2 gloo process groups exercise the packed feature all-gather and the
row-sharded similarity matrix (centerclip_amd/dist.py).  The device kernels are not involved here:
the collective plumbing is torch.distributed (RCCL on the GPU box, gloo here)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn_name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, globals()[fn_name](rank, world)))
    finally:
        dist.destroy_process_group()


def _run(fn_name, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _features(rank, B=3, Tn=2, E=8):
    g = torch.Generator().manual_seed(10 + rank)
    vis = torch.randn(B, Tn, E, generator=g)
    mask = (torch.rand(B, Tn, generator=g) > 0.3).long()
    seq = torch.randn(B, 1, E, generator=g)
    return vis, mask, seq


def case_packed_all_gather(rank, world):
    from centerclip_amd.dist import all_gather
    vis, mask, seq = _features(rank)
    gv, gm, gs = all_gather(vis, mask, seq)
    # reference semantics (modules/utils.py:47-64): cat of the per-rank tensors in rank order
    exp = [torch.cat([_features(r)[i] for r in range(world)], 0) for i in range(3)]
    ok = torch.equal(gv, exp[0]) and torch.equal(gm, exp[1]) and torch.equal(gs, exp[2])
    single = all_gather(vis)
    return bool(ok and gm.dtype == torch.long and torch.equal(single, exp[0]))


def case_sharded_similarity(rank, world):
    from centerclip_amd.dist import shard_rows, sharded_similarity, gather_rows
    Nt, Nv, E = 7, 5, 8                                   # odd sizes: unequal shards
    g = torch.Generator().manual_seed(3)
    text = torch.nn.functional.normalize(torch.randn(Nt, E, generator=g), dim=-1)
    video = torch.nn.functional.normalize(torch.randn(Nv, E, generator=g), dim=-1)
    t0, t1 = shard_rows(Nt, rank, world)
    v0, v1 = shard_rows(Nv, rank, world)
    block = sharded_similarity(text[t0:t1], video[v0:v1], Nv, 2.0, lambda a, b, m: m * a @ b.t())
    full = gather_rows(block, Nt)
    return bool(torch.allclose(full, 2.0 * text @ video.t(), atol=1e-6) and block.shape == (t1 - t0, Nv))


def test_packed_all_gather_world2():
    assert all(_run("case_packed_all_gather").values())


def test_row_sharded_similarity_world2():
    assert all(_run("case_sharded_similarity").values())


def test_shard_rows_partition():
    from centerclip_amd.dist import shard_rows
    for n in (0, 1, 7, 16, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_all_gather_is_identity_without_process_group():
    from centerclip_amd.dist import all_gather
    a, b = torch.arange(6).view(2, 3), torch.ones(2, 1)
    x, y = all_gather(a, b)
    assert x is a and y is b and all_gather(a) is a


def case_packed_features(rank, world):
    """dist.PackedFeatures over gloo: ONE all_gather_into_tensor of the preallocated records; the gathered buffer holds
    every rank's (visual | text | mask) at the documented offsets."""
    from centerclip_amd.dist import PackedFeatures
    B, Tn, E = 3, 2, 8
    pf = PackedFeatures(B, Tn, E, torch.device("cpu"))
    vis, mask, seq = _features(rank, B, Tn, E)
    pf.vis.copy_(vis.view(B * Tn, E))
    pf.seq.copy_(seq.view(B, E))
    pf.mask.copy_(mask)
    rec = pf.gather()
    ok = rec.shape == (world, pf.rec) and pf.rec % 16 == 0 and pf.world == world
    for r in range(world):
        v, m, s = _features(r, B, Tn, E)
        ok = ok and torch.equal(rec[r, pf.vis_off:pf.seq_off].view(torch.float32).view(B, Tn, E), v)
        ok = ok and torch.equal(rec[r, pf.seq_off:pf.mask_off].view(torch.float32).view(B, 1, E), s)
        ok = ok and torch.equal(rec[r, pf.mask_off:pf.mask_off + B * Tn * 8].view(torch.long).view(B, Tn), m)
    ok = ok and torch.equal(pf.gathered_text(), torch.cat([_features(r, B, Tn, E)[2].view(B, E) for r in range(world)]))
    return bool(ok)


def case_allgather_autograd(rank, world):
    """AllGather (modules/utils.py:25-44): forward = concatenation in rank order, backward = this rank's slice."""
    from centerclip_amd.dist import AllGather
    x = torch.full((2, 3), float(rank + 1), requires_grad=True)
    y = AllGather.apply(x)
    y.backward(torch.arange(2 * world, dtype=torch.float32)[:, None].expand(-1, 3).contiguous())
    want = torch.cat([torch.full((2, 3), float(r + 1)) for r in range(world)])
    return bool(torch.equal(y.detach(), want) and torch.equal(x.grad[:, 0], torch.arange(2 * rank, 2 * rank + 2, dtype=torch.float32)))


def case_packed_allgather_autograd(rank, world):
    """PackedAllGather (the training branch's exchange, modules/utils.py:25-64 at clip4clip.py:351-355 as ONE collective):
    forward == three AllGather.apply / all_gather calls, backward == their gradients (own-shard slices), exactly."""
    from centerclip_amd.dist import AllGather, PackedAllGather, all_gather
    vis0, mask, seq0 = _features(rank)
    w = [torch.randn(world * 3, 2, 8, generator=torch.Generator().manual_seed(5)),
         torch.randn(world * 3, 1, 8, generator=torch.Generator().manual_seed(6))]
    out = []
    for packed in (True, False):
        vis, seq = vis0.clone().requires_grad_(True), seq0.clone().requires_grad_(True)
        if packed:
            gv, gm, gs = PackedAllGather.apply(vis, mask, seq)
        else:
            gv, gs, gm = AllGather.apply(vis), AllGather.apply(seq), all_gather(mask)
        loss = (gv * w[0] * gm[:, :, None].float()).sum() + (gs * w[1]).pow(2).sum()
        loss.backward()
        out.append((gv.detach(), gm, gs.detach(), vis.grad, seq.grad, gm.requires_grad))
    a, b = out
    return bool(all(torch.equal(x, y) for x, y in zip(a[:5], b[:5])) and not a[5])


def case_gradient_buckets(rank, world):
    """GradientBuckets (main.py:124,321: DDP's gradient averaging as bucketed reduce-scatter + all-gather): every .grad ==
    the mean over the ranks (an all_reduce per tensor), several buckets, a parameter without gradient, sizes that do not
    divide by the world size; a second call reuses the flat buckets."""
    from centerclip_amd.dist import GradientBuckets
    torch.manual_seed(3)
    shapes = [(7, 5), (13,), (64, 33), (1,), (3, 3, 3), (129,)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    unused = torch.nn.Parameter(torch.zeros(6))                          # no rank ever produces a gradient for it
    frozen = torch.nn.Parameter(torch.zeros(4), requires_grad=False)
    gb = GradientBuckets(params + [unused, frozen], bucket_bytes=4096)  # -> several buckets
    ok = len(gb.buckets) > 2
    for step in range(2):
        g = torch.Generator().manual_seed(100 * step + rank)
        local = [torch.randn(s, generator=g) for s in shapes]
        for i, (p, gr) in enumerate(zip(params, local)):
            p.grad = None if (i == 3 and rank == 0) else gr.clone()       # one rank has no gradient for one parameter
        want = []
        for i, gr in enumerate(local):
            t = torch.zeros_like(gr) if (i == 3 and rank == 0) else gr.clone()
            dist.all_reduce(t)
            want.append(t / world)
        gb.reduce()
        ok = ok and all(torch.allclose(p.grad, w_, rtol=0, atol=1e-6) for p, w_ in zip(params, want))
        ok = ok and frozen.grad is None
        ok = ok and unused.grad is None                                  # globally unused: None, as DDP / one process leave it
    return bool(ok)


def test_packed_allgather_autograd_world2():
    assert all(_run("case_packed_allgather_autograd").values())


def test_gradient_buckets_world2():
    assert all(_run("case_gradient_buckets").values())


def test_gradient_buckets_single_process_keeps_gradients():
    from centerclip_amd.dist import GradientBuckets
    p = torch.nn.Parameter(torch.zeros(5))
    p.grad = torch.arange(5.0)
    q = torch.nn.Parameter(torch.zeros(3))                               # never gets a gradient
    GradientBuckets([p, q]).reduce()
    assert torch.equal(p.grad, torch.arange(5.0)) and q.grad is None


def test_packed_features_world2():
    assert all(_run("case_packed_features").values())


def test_allgather_autograd_world2():
    assert all(_run("case_allgather_autograd").values())


def test_packed_features_single_process_is_a_view():
    from centerclip_amd.dist import PackedFeatures
    pf = PackedFeatures(4, 3, 8, torch.device("cpu"), world=1)
    pf.vis.fill_(2.0)
    assert pf.gather().data_ptr() == pf.send.data_ptr() and pf.recv is pf.send
    assert pf.vis.shape == (12, 8) and pf.seq.shape == (4, 8) and pf.mask.shape == (4, 3) and pf.mask.dtype == torch.long


# ------------------------------------------------------------------------------------------------ clip-sharded eval loop
class _TorchBackend:
    """Torch-CPU stand-in for eval.HipBackend (same semantics as the HIP ops) so that the sharding / collective logic of
    eval_epoch(shard=True) runs over gloo without a GPU.  Test infrastructure only."""

    @staticmethod
    def normalize_rows(x):
        return x / x.norm(dim=-1, keepdim=True)

    @staticmethod
    def pool_normalize(v, m):
        v = v / v.norm(dim=-1, keepdim=True)
        m = m.to(torch.float).unsqueeze(-1)
        s = m.sum(dim=1)
        s[s == 0.] = 1.
        v = (v * m).sum(dim=1) / s
        return v / v.norm(dim=-1, keepdim=True)

    @staticmethod
    def dot_nt(a, b, mult):
        return mult * a @ b.t()

    # operand rows of the final product: the stand-in keeps plain normalised fp32 rows
    @classmethod
    def text_operand(cls, feats):
        return cls.normalize_rows(feats.float())

    @classmethod
    def video_operand(cls, visual_output, mask):
        return cls.normalize_rows(visual_output.float()) if visual_output.dim() == 2 else cls.pool_normalize(visual_output.float(), mask)

    @staticmethod
    def video_operand_rows(n):
        return n + 3                                              # (exercises the zero padding rows)

    @classmethod
    def dot_operands(cls, text_op, video_op, n_video, mult):
        assert float(video_op[n_video:].abs().sum()) == 0.0
        return cls.dot_nt(text_op, video_op[:n_video], mult)

    @staticmethod
    def counts_cols(sim, gt):
        d = sim.gather(1, gt.long().view(-1, 1))
        before = (sim == d) & (torch.arange(sim.shape[1])[None, :] < gt.long().view(-1, 1))
        return torch.stack([(sim > d).sum(1), (sim == d).sum(1), before.sum(1)], 1).to(torch.int32)

    @staticmethod
    def counts_ref_columns(sim, ref):
        return torch.stack([(sim > ref[None, :]).sum(0), (sim == ref[None, :]).sum(0)], 1).to(torch.int32)

    @staticmethod
    def group_max(sim, groups, n_groups):
        best = torch.full((n_groups, sim.shape[1]), float("-inf"))
        if sim.shape[0]:
            clean = torch.where(sim != sim, torch.full_like(sim, float("-inf")), sim)
            best.scatter_reduce_(0, groups.long().view(-1, 1).expand(-1, sim.shape[1]), clean, "amax", include_self=True)
        return best


class _TableModel(torch.nn.Module):
    """Stands in for CLIP4Clip: features are looked up from the inputs (ids carry a row number, 'frames' are features)."""

    def __init__(self, E=8):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.table = torch.randn(64, E, generator=g)

    def forward(self, input_ids=None, token_type_ids=None, attention_mask=None, video=None, video_mask=None):
        out = {'sequence_output': None, 'visual_output': None}
        if input_ids is not None:
            out['sequence_output'] = self.table[input_ids.view(-1, input_ids.shape[-1])[:, 0]].unsqueeze(1)
        if video is not None:
            out['visual_output'] = video[:, 0].float()                  # [b, T, E]
        return out

    def get_video_mask_after_cluster(self, m):
        return m

    def _logit_scale_value(self):
        return 0.5


def _eval_dataset(multi, E=8, T=3):
    """11 single-caption clips, or 5 clips with 3/1/4/2/3 sentences (13 items).  item = (ids, mask, seg, video, vmask)."""
    g = torch.Generator().manual_seed(17)
    sentences = [3, 1, 4, 2, 3] if multi else [1] * 11
    videos = torch.randn(len(sentences), 1, T, E, generator=g)
    vmask = torch.ones(len(sentences), 1, T, dtype=torch.long)
    vmask[1, 0, T - 1] = 0
    items = []
    for v, ns in enumerate(sentences):
        for _ in range(ns):
            ids = torch.zeros(1, 4, dtype=torch.long)
            ids[0, 0] = int(torch.randint(0, 64, (1,), generator=g))
            items.append((ids, (ids >= 0).long(), torch.zeros_like(ids), videos[v], vmask[v]))
    attrs = {}
    if multi:
        attrs = dict(multi_sentence_per_video=True, cut_off_points=list(torch.tensor(sentences).cumsum(0).tolist()),
                     sentence_num=len(items), video_num=len(sentences))
    return items, attrs


class _Items(torch.utils.data.Dataset):
    def __init__(self, items, attrs):
        self.items = items
        for k, v in attrs.items():
            setattr(self, k, v)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def case_sharded_eval(rank, world):
    """eval_epoch(shard=True) on 2 ranks == eval_epoch(shard=False) on one: batches dealt round robin over an unsharded
    loader, and a DistributedSampler loader (11 / 13 items: the sampler pads, the padding must be dropped).  shard=False
    issues no collective although a process group is up (rank 0 calls it alone), and refuses a DistributedSampler loader."""
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from centerclip_amd.eval import eval_epoch
    model = _TableModel()
    dev = torch.device("cpu")
    ok = True
    for multi in (False, True):
        items, attrs = _eval_dataset(multi)
        ds = _Items(items, attrs)
        plain = DataLoader(ds, batch_size=3, shuffle=False)
        want = None
        if rank == 0:                                                # single-process form, called by ONE rank only
            want = eval_epoch(model, plain, dev, backend=_TorchBackend)
        box = [want]
        dist.broadcast_object_list(box, src=0)
        want = box[0]
        got_rr = eval_epoch(model, plain, dev, shard=True, backend=_TorchBackend)
        sharded = DataLoader(ds, batch_size=3, sampler=DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=False))
        got_ds = eval_epoch(model, sharded, dev, shard=True, backend=_TorchBackend)
        for got in (got_rr, got_ds):
            ok = ok and abs(got[0] - want[0]) < 1e-9 and list(got[2]) == list(want[2])
        try:
            eval_epoch(model, sharded, dev, backend=_TorchBackend)
            ok = False
        except RuntimeError as exc:
            ok = ok and "DistributedSampler" in str(exc)
    return bool(ok)


def test_clip_sharded_eval_epoch_world2():
    assert all(_run("case_sharded_eval").values())


def test_eval_epoch_torch_backend_matches_numpy_reference_definition():
    """The stand-in backend + the loop (world 1) against the metric definitions of utils/metrics.py restated in NumPy - so
    that the world-2 comparison above is anchored to something independent of the loop itself."""
    import numpy as np
    from torch.utils.data import DataLoader
    from centerclip_amd.eval import eval_epoch
    model = _TableModel()
    items, attrs = _eval_dataset(False)
    r1, _, info = eval_epoch(model, DataLoader(_Items(items, attrs), batch_size=4), torch.device("cpu"), backend=_TorchBackend)
    text = torch.stack([model.table[it[0][0, 0]] for it in items])
    vis = _TorchBackend.pool_normalize(torch.stack([it[3][0] for it in items]), torch.stack([it[4][0] for it in items]))
    sim = (torch.exp(torch.tensor(0.5)) * _TorchBackend.normalize_rows(text) @ vis.t()).numpy()

    def ranks(x):
        sx = np.sort(-x, axis=1)
        return np.where(sx - np.diag(-x)[:, None] == 0)[1]
    ind_tv, ind_vt = ranks(sim), ranks(sim.T)
    assert abs(r1 - 100.0 * np.mean(ind_tv == 0)) < 1e-9
    assert info[1].endswith("Mean R: {:.1f}".format(np.mean(ind_tv) + 1)) and info[3].endswith("Mean R: {:.1f}".format(np.mean(ind_vt) + 1))


@pytest.mark.parametrize("name", ["multi", "single"])
def test_eval_epoch_metric_strings_against_reference_fixture(name):
    """The loop's metric logic and string format against the reference's own eval_epoch output (tests/golden/r3_golden.npz):
    the reference's similarity matrix is handed to the loop in place of its GEMM (torch stand-in backend; the HIP kernels are
    pinned the same way in tests/test_r3_gpu.py)."""
    import numpy as np
    from argparse import Namespace
    from centerclip_amd.eval import eval_epoch
    from oracle.recipes import EVAL_CASES, eval_case_batches
    here = os.path.dirname(os.path.abspath(__file__))
    g3 = np.load(os.path.join(here, "golden", "r3_golden.npz"))
    cfg = np.load(os.path.join(here, "golden", "r2_golden.npz"))["s1_cfg"]
    batches, attrs = eval_case_batches(EVAL_CASES[name], cfg)
    ref = torch.from_numpy(g3[f"ev_{name}_sim"])

    class Model(torch.nn.Module):                       # any features: the matrix is supplied
        def forward(self, input_ids=None, token_type_ids=None, attention_mask=None, video=None, video_mask=None):
            out = {'sequence_output': None, 'visual_output': None}
            if input_ids is not None:
                out['sequence_output'] = torch.ones(input_ids.shape[0], 1, 4)
            if video is not None:
                out['visual_output'] = torch.ones(video.shape[0], video.shape[2], 4)
            return out

        def get_video_mask_after_cluster(self, m):
            return m

        def _logit_scale_value(self):
            return 0.0

    class Given(_TorchBackend):
        dot_nt = staticmethod(lambda a, b, mult: ref.clone())

    class Loader(list):
        pass
    loader = Loader(batches)
    loader.dataset = Namespace(**attrs)
    r1, _, info = eval_epoch(Model(), loader, torch.device("cpu"), backend=Given)
    assert list(info) == [str(s) for s in g3[f"ev_{name}_info"]] and abs(r1 - float(g3[f"ev_{name}_r1"])) < 1e-4
