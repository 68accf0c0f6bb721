"""Round-3 parity tests on the MI355X (``-m gpu``), all through torch.ops.centerclip / the C ABI:

  * THE STEP bench.py TIMES: bench.task_config / random_state_dict / synthetic_batch at cfg 2 (B = 16), CLIP4Clip.forward ->
    get_similarity_logits captured into a hipGraph and replayed, caption compaction + the few-rows last block ON (as
    shipped) and OFF, against oracle/clip_oracle.py at full width given the HIP path's own medoids            [S1, V1, T1]
  * the same towers (encode_pair: text rider inside the ViT's launches) at the per-GPU shapes of cfg 3 / 4 / 5        [V1, T1]
  * k-medoids on integer lattices at the per-GPU problem counts of cfg 3 / 4 / 5: several split chunks at the real N,
    indices bit-exact against the reference's own output                                                     [C2, C5]
  * eval._run_on_single_gpu against the matrix the reference's main._run_on_single_gpu produced from the same stored
    features; eval_epoch against the reference's main.eval_epoch (matrix, R@1, metric strings)                 [S3, N1]

Tolerances: north star - 1e-3 on L2-normalised embeddings / cosine similarities (x exp(logit_scale) on logits),
bit-exact indices.  Fixtures: tests/golden/r3_golden.npz (oracle/gen_golden_r3.py, generated from the imported reference).
"""
import itertools
import math
import os
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import clip_oracle as clo
from oracle.recipes import EVAL_CASES, eval_case_batches, lattice, s3_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
R3 = os.path.join(HERE, "golden", "r3_golden.npz")
R2 = os.path.join(HERE, "golden", "r2_golden.npz")


@pytest.fixture(scope="module")
def g3():
    return np.load(R3)


@pytest.fixture(scope="module")
def g2():
    return np.load(R2)


def nrm(x):
    return x / x.norm(dim=-1, keepdim=True)


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


def _captured(fn):
    """Run fn eagerly (warm-up: allocations, model packing), capture it into a hipGraph, replay twice -> its outputs."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fn()
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    return out, graph


# ------------------------------------------------------------------------------------------------ the timed step
@pytest.fixture(scope="module")
def timed_step():
    """bench.py's cfg-2 model, batch and step, + the oracle's answer for that batch given the HIP path's medoids."""
    bench = _bench()
    from centerclip_amd.clip4clip import CLIP4Clip
    c = bench.CFG2
    sd = bench.random_state_dict(c, seed=0)
    model = CLIP4Clip.from_state_dict(dict(sd), bench.task_config(c)).to(DEV).eval()
    ids, amask, video, vmask = bench.synthetic_batch(c, DEV, seed=100)
    seg = torch.zeros_like(ids)

    def step():
        out = model(ids, seg, amask, video, vmask)
        logits, *_ = model.get_similarity_logits(out["sequence_output"], out["visual_output"], amask, vmask)
        return out["sequence_output"], out["visual_output"], logits

    # the HIP path's own medoids (same kernels, ids kept): the oracle is evaluated "given identical medoid sets" 
    model.clip.visual.keep_medoids = True
    with torch.no_grad():
        kept = [t.clone() for t in step()]
    med = model.clip.visual.last_medoids.cpu()
    model.clip.visual.keep_medoids = False
    P = c["B"] * c["T_new"]
    assert med.shape == (P, c["K"]) and bool((med[:, 1:] > med[:, :-1]).all()) and int(med.max()) < (c["T"] // c["T_new"]) * 49
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with torch.no_grad():
        ref = clo.clip4clip_forward(sd, ids.cpu(), video.cpu(), vmask.cpu(), c["T"], c["T_new"],
                                    {c["cluster_block"] - 1: (c["T_new"], c["K"])}, float(sd["logit_scale"]),
                                    forced_medoids={c["cluster_block"] - 1: med})
    return dict(model=model, step=step, kept=kept, ref=ref, mult=math.exp(float(sd["logit_scale"])), c=c,
                inputs=(ids, amask, video, vmask), bench=bench)


def _check_step(out, ref, mult, what):
    seq, vis, logits = (t.float().cpu() for t in out)
    rseq, rvis, rlogits = ref
    assert seq.shape == rseq.shape and vis.shape == rvis.shape and logits.shape == rlogits.shape
    dt, dv = float((nrm(seq) - nrm(rseq)).abs().max()), float((nrm(vis) - nrm(rvis)).abs().max())
    dl = float((logits - rlogits).abs().max())
    print(f"[timed step, {what}] normalised embeddings: text {dt:.2e} visual {dv:.2e}; logits {dl:.2e} (x{mult:.2f})")
    assert dt <= 1e-3 and dv <= 1e-3 and dl <= 1e-3 * mult


def test_timed_step_hipgraph_replay_matches_oracle(timed_step):
    """The exact object bench.py times - the cfg-2 step as a replayed hipGraph, paired towers, compacted captions, last
    block on the rows the heads read - against the fp32 oracle (clip4clip.py:199-243,357-366)."""
    ts = timed_step
    with torch.no_grad():
        out, graph = _captured(ts["step"])
    _check_step(out, ts["ref"], ts["mult"], "hipGraph replay, row policies on")
    # the replayed graph, the eager launches and the medoid-keeping form run the same kernels on the same data
    for a, b in zip(out, ts["kept"]):
        assert torch.equal(a, b)
    # a second batch through the SAME captured graph (inputs refreshed in place, other caption lengths: the compaction
    # offsets are computed on the device): identical to eager launches on that batch, no state leaks between replays
    ids, amask, video, vmask = ts["inputs"]
    saved = [t.clone() for t in ts["inputs"]]
    fresh = ts["bench"].synthetic_batch(ts["c"], DEV, seed=321)
    try:
        for dst, src in zip(ts["inputs"], fresh):
            dst.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        replayed = [t.clone() for t in out]
        with torch.no_grad():
            eager = ts["step"]()
        for a, b in zip(replayed, eager):
            assert torch.equal(a, b)
        assert not torch.equal(replayed[2], ts["kept"][2])
    finally:
        for dst, src in zip(ts["inputs"], saved):
            dst.copy_(src)
    del graph


def test_timed_step_with_row_policies_off_matches_oracle_and_the_shipped_step(timed_step):
    """Caption compaction and the few-rows last block switched off (every text row, every row of block 12): still within
    the oracle's tolerance, text features bit-identical (compaction is exact), visual features within the rounding of the
    fp16 intermediates of the shipped step."""
    ts = timed_step
    with torch.no_grad(), ts["model"].clip.row_policy(all_text_rows=True, all_last_block_rows=True):
        out, graph = _captured(ts["step"])
    _check_step(out, ts["ref"], ts["mult"], "hipGraph replay, row policies off")
    seq_on, vis_on, log_on = ts["kept"]
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    assert rel(out[0], seq_on) <= 2e-4 and rel(out[1], vis_on) <= 2e-4
    assert float((out[2] - log_on).abs().max()) <= 2e-4 * ts["mult"]
    del graph


# ------------------------------------------------------------------------------------------------ cfg 3 / 4 / 5 towers
PAIRED = {
    # name: (patch, T, T_new, K, cluster block (1-based), B, words, model name)
    # (round 5: cfg 3 and cfg 5 at their full per-GPU batch - 64 and 16 clips - as bench.py's forward_other_configs times them)
    "cfg3_msvd_12to4_b64": (32, 12, 4, 49, 7, 64, 32, 'ViT-B/32'),
    "cfg4_activitynet_64to8_b8": (32, 64, 8, 49, 7, 8, 77, 'ViT-B/32'),
    "cfg5_vitb16_12to4_k100_b16": (16, 12, 4, 100, 7, 16, 32, 'ViT-B/16'),
}


@pytest.mark.parametrize("name", sorted(PAIRED))
def test_paired_towers_at_per_gpu_shapes(name):
    """encode_pair (both towers in one enqueue: the text problem rides in the ViT's launches) at the per-GPU batch of
    BASELINE.json configs 3-5, full width, against the fp32 oracle given the HIP path's medoids.  Multi-chunk k-medoids
    (P = B * T_new problems in split-size chunks) is part of the forward."""
    from centerclip_amd.clip import CLIP
    patch, T, T_new, K, cb, B, words, pname = PAIRED[name]
    torch.manual_seed(31)
    args = Namespace(cluster_inter=1, cluster_algo='kmediods++', max_frames=T,
                     target_frames_blocks=[T] * (cb - 1) + [T_new] * (13 - cb), cluster_num_blocks=[K] * 12,
                     cluster_distance='euclidean', cluster_threshold=1e-6, cluster_iter_limit=100, minkowski_norm_p=2.0,
                     pretrained_clip_name=pname, aggregation=None, pre_norm=False)
    model = CLIP(512, 224, 12, 768, patch, 77, 49408, 512, 8, 12, video_frames=T, args=args)
    with torch.no_grad():
        for p_ in model.parameters():
            p_.copy_(p_.half().float())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).eval()
    gen = torch.Generator().manual_seed(7)
    video = torch.randn(B * T, 3, 224, 224, generator=gen)
    ids = torch.zeros(B, words, dtype=torch.long)
    for b in range(B):
        ln = int(torch.randint(4, words + 1, (1,), generator=gen))
        ids[b, 0], ids[b, ln - 1] = 49406, 49407
        ids[b, 1:ln - 1] = torch.randint(1, 49405, (ln - 2,), generator=gen)
    model.visual.keep_medoids = True
    with torch.no_grad():
        vfeat, tfeat = model.encode_pair(video.to(DEV), ids.to(DEV), video_frame=T)
    med = model.visual.last_medoids.cpu()
    model.visual.keep_medoids = False
    n = (224 // patch) ** 2
    assert med.shape == (B * T_new, K) and bool((med[:, 1:] > med[:, :-1]).all()) and int(med.max()) < (T // T_new) * n
    with torch.no_grad():                                   # the shipped call (no medoid buffer) gives the same features
        v2, t2 = model.encode_pair(video.to(DEV), ids.to(DEV), video_frame=T)
    assert torch.equal(v2, vfeat) and torch.equal(t2, tfeat)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with torch.no_grad():
        vref = clo.visual_forward(sd, video, T, cluster_plan={cb - 1: (T_new, K)}, forced_medoids={cb - 1: med})
        tref = clo.text_forward(sd, ids)
    dv, dt = float((nrm(vfeat.cpu()) - nrm(vref)).abs().max()), float((nrm(tfeat.cpu()) - nrm(tref)).abs().max())
    print(f"[{name}] normalised embeddings vs oracle: visual {dv:.2e} text {dt:.2e}")
    assert vfeat.shape == vref.shape == (B * T_new, 512) and dv <= 1e-3 and dt <= 1e-3


# ------------------------------------------------------------------------------------------------ multi-chunk k-medoids
P1M = ["p1m_cfg3", "p1m_cfg4", "p1m_cfg5", "p1m_ragged"]


@pytest.mark.parametrize("tag", P1M)
def test_p1_lattice_multi_chunk(g3, tag):
    """Parity level P1 at the per-GPU problem counts: 16 / 4 / 4 split chunks (+ a ragged last chunk) at N = 147 / 392 /
    588 / 196 - the chunk-wide distance maximum (cluster_utils.py:36) is taken per chunk, the reference's indices must be
    reproduced bit for bit."""
    from centerclip_amd import cluster as cl
    seed, P, N, W, K, split, iters = [int(v) for v in g3[f"{tag}_cfg"]]
    X = torch.from_numpy(lattice(seed, (P, N, W))).to(DEV)
    a, m = cl.batch_fast_kmedoids_with_split(X, K, distance="euclidean", threshold=1e-6, iter_limit=iters,
                                             id_sort=True, norm_p=2.0, split_size=split, pre_norm=False)
    assert np.array_equal(m.cpu().numpy(), g3[f"{tag}_medoids"].astype(np.int64))
    assert np.array_equal(a.cpu().numpy(), g3[f"{tag}_assign"].astype(np.int64))


@pytest.mark.parametrize("name", ["cfg4 ActivityNet-shaped (per GPU)", "cfg5 ViT-B/16"])
def test_cluster_shapes_bench_times_have_a_structural_check(name):
    """The cfg-4 / cfg-5 token-cluster calls bench.py times (P = 64 problems, Gaussian tokens: parity level P3, no index
    target): the contract every valid result obeys - sorted distinct medoids in range, CLS rows = segment means, every
    output row a copy of the input row its medoid names."""
    bench = _bench()
    from centerclip_amd.cluster import TokenClusterInter
    c = bench.CLUSTER_SHAPES[name]
    B, T, Tn, K, n = c["B"], c["T"], c["T_new"], c["K"], c["n"]
    W, fd = 768, c["T"] // c["T_new"]
    x = torch.randn(B * T, 1 + n, W, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    mod = TokenClusterInter(before_cluster_num=n, cluster_num=K, before_block_frames=T, after_block_frames=Tn,
                            original_frame=T, threshold=1e-6, iter_limit=100, split_size=c["split"], norm_p=2.0)
    y = mod.cluster_frame_major(x, keep_ids=True).cpu()
    med = mod.last_medoids.cpu()
    assert y.shape == (B * Tn, 1 + K, W) and med.shape == (Tn * B, K)
    assert bool((med[:, 1:] > med[:, :-1]).all()) and int(med.min()) >= 0 and int(med.max()) < fd * n
    xs = x.cpu().view(B, Tn, fd, 1 + n, W)
    tokens = xs[:, :, :, 1:, :].reshape(B, Tn, fd * n, W)
    for b, s in itertools.product(range(B), range(Tn)):
        ids_ = med[s * B + b]                                   # problem p = s*B + b (cluster.py:247-250)
        assert torch.equal(y[b * Tn + s, 1:], tokens[b, s, ids_])
    cls = xs[:, :, :, 0, :].mean(dim=2).reshape(B * Tn, W)
    assert float((y[:, 0] - cls).abs().max()) <= 1e-5


# ------------------------------------------------------------------------------------------------ S3: the eval loop
def _small_model(g2, cluster_inter):
    from centerclip_amd.clip4clip import CLIP4Clip
    sd = {k[6:]: torch.from_numpy(g2[k].astype(np.float32) if g2[k].dtype == np.float16 else g2[k])
          for k in g2.files if k.startswith("s1_sd/")}
    cfg = g2["s1_cfg"]
    T, T_new = int(cfg[11]), int(cfg[12])
    a = Namespace(cluster_inter=cluster_inter, deep_cluster=0, cluster_algo='kmediods++', max_frames=T,
                  target_frames_blocks=[4, T_new, T_new] if cluster_inter else [T, T, T], cluster_num_blocks=[16, 6, 6],
                  cluster_distance='euclidean', cluster_threshold=1e-6, cluster_iter_limit=100, minkowski_norm_p=2.0,
                  aggregation=None, pretrained_clip_name='ViT-B/32', pre_norm=False, loose_type=True, sim_header='meanP',
                  linear_patch='2d', pre_visual_pooling=0)
    return CLIP4Clip.from_state_dict(sd, a).to(DEV).eval(), sd, cfg


def test_run_on_single_gpu_against_the_reference_matrix(g2, g3):
    """eval._run_on_single_gpu (ONE NT GEMM over the concatenated cache) against what the reference's own
    main._run_on_single_gpu (main.py:502-534: a get_similarity_logits call + D2H copy per batch pair) made of the same
    stored features - ragged batches, masks with the original frame count and zeros, a fully masked clip (NaN column)."""
    from centerclip_amd.eval import _run_on_single_gpu
    model, sd, cfg = _small_model(g2, cluster_inter=1)
    seq_list, vis_list, list_t, list_v = s3_case(int(cfg[0]), int(cfg[11]), int(cfg[12]))
    to = lambda ts: [tuple(t.to(DEV) for t in item) if isinstance(item, tuple) else item.to(DEV) for item in ts]
    with torch.no_grad():
        sim = _run_on_single_gpu(model, to(list_t), to(list_v), to(seq_list), to(vis_list))
    ref = g3["s3_sim"]
    assert isinstance(sim, np.ndarray) and sim.shape == ref.shape == (37, 21)
    assert np.array_equal(np.isnan(sim), np.isnan(ref)) and np.isnan(ref[:, 5]).all()
    mult = math.exp(float(sd["logit_scale"]))
    ok = ~np.isnan(ref)
    err = float(np.abs(sim[ok] - ref[ok]).max())
    print(f"[S3] max |sim - reference| = {err:.2e} (logit multiplier {mult:.2f})")
    # default: 2 fp16 products per multiply-add (the text side rounded to fp16) - 1e-4 of a cosine, the contract asks 1e-3;
    # the 3-product form (both operands to 22 bits), on request, to the fp32 reference's own rounding
    assert err <= 1e-4 * mult
    from centerclip_amd.eval import _similarity_matrix, HipBackend
    with torch.no_grad():
        sim3 = _similarity_matrix(model, to(list_t), to(list_v), to(seq_list), to(vis_list),
                                  backend=HipBackend.with_products(3)).cpu().numpy()
    assert float(np.abs(sim3[ok] - ref[ok]).max()) <= 5e-5 * mult
    # the pairwise form of the same API agrees (one get_similarity_logits call per text batch x video batch)
    with torch.no_grad():
        blocks = [[model.get_similarity_logits(s.to(DEV), v.to(DEV), lt[0].to(DEV), lv[0].to(DEV))[0].cpu()
                   for v, lv in zip(vis_list, list_v)] for s, lt in zip(seq_list, list_t)]
    pairwise = torch.cat([torch.cat(row, dim=1) for row in blocks], dim=0).numpy()
    assert float(np.abs(pairwise[ok] - sim3[ok]).max()) <= 2e-5 * mult and float(np.abs(pairwise[ok] - sim[ok]).max()) <= 1e-4 * mult


class _Loader(list):
    pass


@pytest.mark.parametrize("name", sorted(EVAL_CASES))
def test_eval_epoch_against_the_reference(g2, g3, name):
    """eval_epoch against the reference's own main.eval_epoch (main.py:381-499) on the same list-backed loader: the
    similarity matrix within 1e-3 * exp(logit_scale), and - the fixture's rank metrics are stable under delta - R@1 and
    the four metric strings character for character.  The metric kernels alone are pinned exactly by feeding them the
    reference's matrix."""
    from centerclip_amd import eval as ev
    model, sd, cfg = _small_model(g2, cluster_inter=0)
    batches, attrs = eval_case_batches(EVAL_CASES[name], cfg)
    loader = _Loader(batches)
    loader.dataset = Namespace(**attrs)
    seen = {}

    class Spy(ev.HipBackend):                                # the loop's one GEMM over the cached operand planes, recorded
        @staticmethod
        def dot_operands(t_op, v_op, n_video, mult):
            seen["sim"] = ev.HipBackend.dot_operands(t_op, v_op, n_video, mult)
            return seen["sim"]
    r1, t_inf, info = ev.eval_epoch(model, loader, torch.device(DEV), args=Namespace(inference_speed_test=False), backend=Spy)
    ref = g3[f"ev_{name}_sim"]
    sim = seen["sim"].cpu().numpy()
    mult = math.exp(float(sd["logit_scale"]))
    err = float(np.abs(sim - ref).max())
    print(f"[eval_epoch {name}] max |sim - reference| = {err:.2e}, delta {float(g3[f'ev_{name}_delta']):.2e}")
    assert sim.shape == ref.shape and err <= 1e-3 * mult
    assert err <= float(g3[f"ev_{name}_delta"]), "similarities outside the band in which the fixture's ranks are pinned"
    assert abs(r1 - float(g3[f"ev_{name}_r1"])) < 1e-4 and t_inf > 0
    assert list(info) == [str(s) for s in g3[f"ev_{name}_info"]]
    # N1 alone, exactly: the reference's matrix through the device metric kernels -> the reference's strings
    class Given(ev.HipBackend):
        dot_operands = staticmethod(lambda t_op, v_op, n_video, mult: torch.from_numpy(ref).to(DEV))
    r1b, _, info_b = ev.eval_epoch(model, loader, torch.device(DEV), args=Namespace(inference_speed_test=False), backend=Given)
    assert list(info_b) == [str(s) for s in g3[f"ev_{name}_info"]] and abs(r1b - float(g3[f"ev_{name}_r1"])) < 1e-4


@pytest.mark.parametrize("name", sorted(EVAL_CASES))
@pytest.mark.parametrize("cluster", [0, 1])
def test_eval_epoch_two_batches_in_flight(g2, name, cluster):
    """eval_epoch(in_flight=2): batch b on instance b % 2 of the model (CLIP4Clip.replica) and a stream of its own - the one
    GEMM over the cached operand planes sees bit-identical operands, so matrix, R@1 and metric strings equal in_flight=1."""
    from centerclip_amd import eval as ev
    model, sd, cfg = _small_model(g2, cluster_inter=cluster)
    batches, attrs = eval_case_batches(EVAL_CASES[name], cfg)
    loader = _Loader(batches)
    loader.dataset = Namespace(**attrs)
    sims = []

    class Spy(ev.HipBackend):
        @staticmethod
        def dot_operands(t_op, v_op, n_video, mult):
            sims.append(ev.HipBackend.dot_operands(t_op, v_op, n_video, mult).clone())
            return sims[-1]
    one = ev.eval_epoch(model, loader, torch.device(DEV), args=Namespace(inference_speed_test=False), backend=Spy, in_flight=1)
    two = ev.eval_epoch(model, loader, torch.device(DEV), args=Namespace(inference_speed_test=False), backend=Spy, in_flight=2)
    three = ev.eval_epoch(model, loader, torch.device(DEV), args=Namespace(inference_speed_test=False), backend=Spy, in_flight=3)
    assert torch.equal(sims[0], sims[1]) and torch.equal(sims[0], sims[2])
    assert one[0] == two[0] == three[0] and list(one[2]) == list(two[2]) == list(three[2])
    rep = model.replica()
    assert rep is not model and rep.training == model.training
    for (k, a), (_, b) in zip(model.state_dict().items(), rep.state_dict().items()):
        assert torch.equal(a, b) and a.data_ptr() != b.data_ptr(), k


@pytest.mark.parametrize("name", sorted(EVAL_CASES))
@pytest.mark.parametrize("in_flight", [1, 2])
def test_eval_epoch_graphed_lanes(g2, name, in_flight):
    """eval_epoch(graphed=True) (round 6): every batch = a copy into device-resident inputs + ONE hipGraph launch per lane (a
    graph per batch shape: the ragged last batch gets its own) - matrix, R@1 and metric strings equal the eager loop's; a
    second epoch reuses the captured graphs; changed weights drop them."""
    from centerclip_amd import eval as ev
    model, sd, cfg = _small_model(g2, cluster_inter=1)
    batches, attrs = eval_case_batches(EVAL_CASES[name], cfg)
    loader = _Loader(batches)
    loader.dataset = Namespace(**attrs)
    sims = []

    class Spy(ev.HipBackend):
        @staticmethod
        def dot_operands(t_op, v_op, n_video, mult):
            sims.append(ev.HipBackend.dot_operands(t_op, v_op, n_video, mult).clone())
            return sims[-1]
    a = Namespace(inference_speed_test=False)
    eager = ev.eval_epoch(model, loader, torch.device(DEV), args=a, backend=Spy, in_flight=in_flight)
    g1 = ev.eval_epoch(model, loader, torch.device(DEV), args=a, backend=Spy, in_flight=in_flight, graphed=True)
    kept = model._eval_graphs
    g2_ = ev.eval_epoch(model, loader, torch.device(DEV), args=a, backend=Spy, in_flight=in_flight, graphed=True)
    assert model._eval_graphs is kept                               # the second epoch replays the first one's graphs
    multi = bool(attrs.get("multi_sentence_per_video", False))
    if not multi:
        assert sum(len(l.graphs) for l in kept[1].values()) >= 1
    assert torch.equal(sims[0], sims[1]) and torch.equal(sims[0], sims[2])
    assert eager[0] == g1[0] == g2_[0] and list(eager[2]) == list(g1[2]) == list(g2_[2])
    with torch.no_grad():
        model.clip.logit_scale.add_(0.0)                            # a tracked in-place write: the version key changes
    ev.eval_epoch(model, loader, torch.device(DEV), args=a, backend=Spy, in_flight=in_flight, graphed=True)
    assert model._eval_graphs is not kept


# ------------------------------------------------------------------------------------------------ N4: loss gradient
@pytest.mark.parametrize("tag,n", [("lg_a", 6), ("lg_b", 33)])
def test_contrastive_loss_gradients_against_reference_autograd(g2, g3, tag, n):
    """losses.contrastive_loss (cc_contrastive_loss_grad_f32) against the reference module's own loss and torch.autograd's
    gradients of it (clip4clip.py:245-262 -> losses.py:8-18) for sequence_output, visual_output and logit_scale; the
    incoming gradient scales them; repeated calls give the same bits (fixed summation orders, no atomics)."""
    from centerclip_amd.losses import contrastive_loss
    from oracle.recipes import loss_grad_case
    cfg = g2["s1_cfg"]
    seq, vis, vmask = loss_grad_case(tag, n, int(cfg[12]), int(cfg[0]))
    scale = float(g3[f"{tag}_scale"])
    seq_t = torch.from_numpy(seq).to(DEV).requires_grad_(True)
    vis_t = torch.from_numpy(vis).to(DEV).requires_grad_(True)
    ls = torch.tensor(scale, device=DEV, requires_grad=True)
    loss, l1, l2 = contrastive_loss(seq_t, vis_t, torch.from_numpy(vmask).to(DEV), ls)
    (3.0 * loss).backward()
    ref3 = g3[f"{tag}_loss3"]
    assert abs(float(l1.detach()) - ref3[0]) <= 2e-5 and abs(float(l2.detach()) - ref3[1]) <= 2e-5
    assert abs(float(loss.detach()) - ref3[2]) <= 2e-5
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    e_seq, e_vis = rel(seq_t.grad.cpu().numpy() / 3.0, g3[f"{tag}_dseq"]), rel(vis_t.grad.cpu().numpy() / 3.0, g3[f"{tag}_dvis"])
    e_ls = abs(float(ls.grad) / 3.0 - float(g3[f"{tag}_dls"])) / max(1.0, abs(float(g3[f"{tag}_dls"])))
    print(f"[loss grad {tag}] rel err d_seq {e_seq:.1e} d_vis {e_vis:.1e} d_logit_scale {e_ls:.1e}")
    assert e_seq <= 1e-4 and e_vis <= 1e-4 and e_ls <= 1e-4
    assert seq_t.grad.shape == seq_t.shape and vis_t.grad.shape == vis_t.shape
    # the oracle agrees as well (pins the test's own reference path)
    o3, odseq, odvis, odls = clo.contrastive_loss_and_grads(torch.from_numpy(seq), torch.from_numpy(vis), torch.from_numpy(vmask), scale)
    assert rel(seq_t.grad.cpu().numpy() / 3.0, odseq.numpy()) <= 1e-4 and rel(vis_t.grad.cpu().numpy() / 3.0, odvis.numpy()) <= 1e-4
    # determinism
    seq2 = torch.from_numpy(seq).to(DEV).requires_grad_(True)
    loss2, _, _ = contrastive_loss(seq2, vis_t.detach(), torch.from_numpy(vmask).to(DEV), ls.detach())
    loss2.backward()
    assert torch.equal(loss2.detach(), loss.detach()) and torch.equal(seq2.grad * 3.0, seq_t.grad)


def test_training_forward_returns_a_differentiable_loss(g2):
    """CLIP4Clip.forward in training mode: 'loss' carries a grad_fn and its backward reaches logit_scale (the towers have no
    backward: the feature gradients stop at sequence_output / visual_output); value = the eval-mode logits' symmetric CrossEn."""
    from centerclip_amd.losses import symmetric_contrastive_loss
    model, sd, cfg = _small_model(g2, cluster_inter=1)
    g = np.load(R2)
    ids, amask = torch.from_numpy(g["s1_ids"]).to(DEV), torch.from_numpy(g["s1_amask"]).to(DEV)
    video = torch.from_numpy(g["s1_video"]).to(DEV)
    vmask = torch.from_numpy(g["s1_train_vmask"]).to(DEV)
    model.train()
    out = model(ids, torch.zeros_like(ids), amask, video, vmask)
    assert out["loss"].requires_grad and out["loss"].grad_fn is not None
    out["loss"].backward()
    assert model.clip.logit_scale.grad is not None and bool(torch.isfinite(model.clip.logit_scale.grad))
    model.eval()
    with torch.no_grad():
        logits, _ = model.get_similarity_logits(out["sequence_output"], out["visual_output"], amask, vmask)
        want, _, _ = symmetric_contrastive_loss(logits)
    assert abs(float(out["sim_loss"]) - float(want)) <= 1e-4 * max(1.0, abs(float(want)))
    assert abs(float(out["loss"]) - float(g["s1_train_loss"])) <= 2e-3          # (own medoids; the fixture test pins it tighter)


# ------------------------------------------------------------------------------------------------ minor branches
def test_linear_patch_3d_against_reference(g2, g3):
    """linear_patch='3d' (clip.py:296-317): Conv3d patch embedding (frame t sees t-1, t, t+1 of its clip, zero padded) as a
    9*p*p-deep im2col GEMM, against the reference's CLIP.encode_image on the same weights; and CLIP4Clip.from_pretrained's
    initialisation trick (conv2 = conv1 in the centre slice) reproduces the '2d' features."""
    from centerclip_amd.clip import CLIP
    from oracle.recipes import PATCH3D_SEED, conv3d_patch_weight, dyadic
    E, RES, P, VW, VL, CTX, VOCAB, TW, TH, TL, B, T, T_new = [int(v) for v in g2["s1_cfg"]]
    sd = {k[6:]: torch.from_numpy(g2[k].astype(np.float32) if g2[k].dtype == np.float16 else g2[k])
          for k in g2.files if k.startswith("s1_sd/")}
    model = CLIP(E, RES, VL, VW, P, CTX, VOCAB, TW, TH, TL, linear_patch='3d', video_frames=T, args=None)
    model.load_state_dict(sd, strict=False)
    w2 = torch.from_numpy(conv3d_patch_weight(PATCH3D_SEED, (VW, 3, 3, P, P)))
    with torch.no_grad():
        model.visual.conv2.weight.copy_(w2)
    model = model.to(DEV).eval()
    video = torch.from_numpy(dyadic(PATCH3D_SEED + 1, (2 * T, 3, RES, RES)))
    feats, _ = model.encode_image(video.to(DEV), video_frame=T)
    ref = torch.from_numpy(g3["p3d_feats"])
    assert feats.shape == ref.shape and float((nrm(feats.cpu()) - nrm(ref)).abs().max()) <= 1e-3
    sd3 = dict(sd)
    sd3["visual.conv2.weight"] = w2
    oref = clo.visual_forward(sd3, video, T, linear_patch='3d')
    assert float((nrm(feats.cpu()) - nrm(oref)).abs().max()) <= 1e-3
    # centre-slice inflation == the 2-d patch embedding
    with torch.no_grad():
        model.visual.conv2.weight.zero_()
        model.visual.conv2.weight[:, :, 1] = model.visual.conv1.weight
    f3, _ = model.encode_image(video.to(DEV), video_frame=T)
    m2 = CLIP(E, RES, VL, VW, P, CTX, VOCAB, TW, TH, TL, linear_patch='2d', video_frames=T, args=None)
    m2.load_state_dict(sd, strict=False)
    f2, _ = m2.to(DEV).eval().encode_image(video.to(DEV), video_frame=T)
    assert float((f3 - f2).abs().max()) <= 1e-5 * float(f2.abs().max())


def test_mean_residual_and_training_sparse_sampling_against_reference(g3):
    """TokenClusterInter(mean_residual=True) -> (x', residual_x) and algorithm='sparse_sampling' in training mode (the
    reference's random ids: same NumPy calls under the same seed) against the reference module's outputs, bit for bit."""
    from centerclip_amd.cluster import TokenClusterInter
    from oracle.recipes import MINOR_SEED
    B, T, T_new, n, W = 2, 4, 2, 16, 32
    x = torch.from_numpy(lattice(MINOR_SEED, (1 + n, B * T, W))).to(DEV)
    mod = TokenClusterInter(algorithm="kmediods++", block_id=3, before_cluster_num=n, cluster_num=n, before_block_frames=T,
                            after_block_frames=T_new, original_frame=T, distance="euclidean", threshold=1e-6, iter_limit=100,
                            split_size=16, norm_p=2.0, mean_residual=True, transformer_width=W).to(DEV).eval()
    y, res = mod(x)
    assert np.array_equal(y.cpu().numpy(), g3["mr_out"]) and np.array_equal(res.cpu().numpy(), g3["mr_residual"])
    K = 5
    mod = TokenClusterInter(algorithm="sparse_sampling", block_id=3, before_cluster_num=n, cluster_num=K, before_block_frames=T,
                            after_block_frames=T_new, original_frame=T, transformer_width=W).to(DEV).train()
    np.random.seed(MINOR_SEED)
    xg = x.clone().requires_grad_(True)
    y, res = mod(xg)
    assert res is None and np.array_equal(y.detach().cpu().numpy(), g3["ss_train_out"])
    y.sum().backward()                                    # gather + CLS mean: every picked token gets 1, every CLS 1 / fd
    g = xg.grad.cpu()
    assert float(g[0].min()) == float(g[0].max()) == 1.0 / (T // T_new)
    assert float(g[1:].sum()) == float(B * T_new * K * W)


# ------------------------------------------------------------------------------------------------ host -> device staging
def test_device_feeder_overlapped_staging_gives_the_resident_results(g2):
    """feeder.DeviceFeeder: batches staged from pinned host memory on a copy stream (two slots) give bit for bit the features
    of the same batches placed on the device by hand - also when the consumer is a hipGraph captured per slot."""
    from centerclip_amd.feeder import DeviceFeeder
    model, sd, cfg = _small_model(g2, cluster_inter=1)
    RES, CTX, VOCAB, T = int(cfg[1]), int(cfg[5]), int(cfg[6]), int(cfg[11])
    gen = torch.Generator().manual_seed(3)
    host = []
    for i in range(5):
        ids = torch.zeros(3, CTX, dtype=torch.long)
        for b in range(3):
            ln = int(torch.randint(4, CTX + 1, (1,), generator=gen))
            ids[b, 0], ids[b, ln - 1] = VOCAB - 2, VOCAB - 1
            ids[b, 1:ln - 1] = torch.randint(1, VOCAB - 2, (ln - 2,), generator=gen)
        u8 = torch.randint(0, 256, (3, 1, T, RES, RES, 3), dtype=torch.uint8, generator=gen)
        host.append((ids.pin_memory(), u8.pin_memory(), torch.ones(3, 1, T, dtype=torch.long).pin_memory()))

    def run(ids, video, vmask):
        out = model(ids, torch.zeros_like(ids), (ids > 0).long(), video, vmask)
        return out["sequence_output"], out["visual_output"]

    with torch.no_grad():
        want = [tuple(t.clone() for t in run(*(h.to(DEV) for h in hb))) for hb in host]
        feeder = DeviceFeeder(DEV, depth=2)
        got = [tuple(t.clone() for t in run(*bufs)) for _, bufs in feeder(host)]
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        # one captured graph per slot
        graphs, got2 = {}, []
        for k, bufs in DeviceFeeder(DEV, depth=2)(host):
            if k not in graphs:
                run(*bufs)
                torch.cuda.synchronize()
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph):
                    outs = run(*bufs)
                graphs[k] = (gph, outs)
            graphs[k][0].replay()
            got2.append(tuple(t.clone() for t in graphs[k][1]))
        for a, b in zip(got2, want):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_out_of_range_token_ids_do_not_fault(g2):
    """An id outside [0, vocab) (uninitialised buffer, corrupt loader output) is clamped by the embedding kernel instead of
    reading out of bounds - the enqueue-only C path cannot raise as nn.Embedding does, and a wild read kills the process."""
    model, sd, cfg = _small_model(g2, cluster_inter=0)
    CTX, VOCAB = int(cfg[5]), int(cfg[6])
    ids = torch.full((2, CTX), 2 ** 40, dtype=torch.long)
    ids[1] = -7
    ids[:, 3] = VOCAB - 1
    feats = model.clip.encode_text(ids.to(DEV))
    torch.cuda.synchronize()
    assert feats.shape == (2, int(cfg[0])) and bool(torch.isfinite(feats).all())
