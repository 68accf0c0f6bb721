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
