"""Round-6 parity cases (need a real MI355X, ``-m gpu``).

  * N3 through the strip form of the uint8 patch gather (csrc/transformer.hip im2col_u8_strip_kernel: shifts instead of
    divisions, every CLIP patch size): bit-identical patch matrices / features to the float loader path at ViT-B/32 and
    ViT-B/16 geometry, and `linear_patch='3d'` from uint8 frames (was CC_ERR_UNSUPPORTED), dataloaders/transforms.py:19-34,166
    + modules/clip.py:296-317.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g2():
    return np.load(os.path.join(HERE, "golden", "r2_golden.npz"))


def _clip(g2, linear_patch, patch=None):
    from centerclip_amd.clip import CLIP
    E, RES, P, VW, VL, CTX, VOCAB, TW, TH, TL, B, T, T_new = [int(v) for v in g2["s1_cfg"]]
    sd = {k[6:]: torch.from_numpy(g2[k].astype(np.float32) if g2[k].dtype == np.float16 else g2[k])
          for k in g2.files if k.startswith("s1_sd/")}
    model = CLIP(E, RES, VL, VW, patch or P, CTX, VOCAB, TW, TH, TL, linear_patch=linear_patch, video_frames=T, args=None)
    if patch is None or patch == P:
        model.load_state_dict(sd, strict=False)
    return model, RES, T


@pytest.mark.parametrize("linear_patch", ["2d", "3d"])
@pytest.mark.parametrize("patch", [None, 8])
def test_uint8_frames_strip_gather_bit_identical(g2, linear_patch, patch):
    """uint8 frames (CHW and the decoder's HWC) == the loader's float path, bit for bit, for both patch embeddings; with a
    second patch size (8: 3 shift amounts differ) on randomly initialised weights."""
    from oracle import clip_oracle as clo
    torch.manual_seed(3)
    model, RES, T = _clip(g2, linear_patch, patch)
    if linear_patch == "3d":
        with torch.no_grad():
            model.visual.conv2.weight.normal_(0, 0.02)
    model = model.to(DEV).eval()
    rng = np.random.default_rng(17)
    u_hwc = torch.from_numpy(rng.integers(0, 256, size=(3 * T, RES, RES, 3), dtype=np.uint8))
    u_chw = u_hwc.permute(0, 3, 1, 2).contiguous()
    x = clo.loader_normalize(u_hwc, channels_last=True)
    with torch.no_grad():
        f_ref, _ = model.encode_image(x.to(DEV), video_frame=T)
        f_chw, _ = model.encode_image(u_chw.to(DEV), video_frame=T)
        f_hwc, _ = model.encode_image(u_hwc.to(DEV), video_frame=T)
    assert bool(torch.isfinite(f_ref).all())
    assert torch.equal(f_chw, f_ref) and torch.equal(f_hwc, f_ref)


def test_train_epoch_with_grad_scaler():
    """main.py:309-330 (--fp16): train_epoch(scaler=torch.amp.GradScaler('cuda', )) - the scaled loss passes through the HIP
    backward (per-tensor power-of-two operand scales chosen on the device), so after unscale_ the step equals the unscaled
    one to fp32 rounding; a scale that overflows the gradients skips the step and backs the scale off."""
    from argparse import Namespace
    from centerclip_amd.clip4clip import CLIP4Clip
    from centerclip_amd.train import BertAdam, prep_optim_params_groups, train_epoch
    g = np.load(os.path.join(HERE, "golden", "clip_golden.npz"))
    sd = {k[3:]: torch.from_numpy(g[k].astype(np.float32) if g[k].dtype == np.float16 else g[k]) for k in g.files if k.startswith("sd/")}
    B, T = int(g["cfg"][10]), int(g["cfg"][11])
    cfg = Namespace(cluster_inter=1, cluster_algo='kmediods++', max_frames=T, target_frames_blocks=[4, 2, 2],
                    cluster_num_blocks=[16, 6, 6], cluster_distance='euclidean', cluster_threshold=1e-6, cluster_iter_limit=100,
                    minkowski_norm_p=2.0, pretrained_clip_name='ViT-B/32', aggregation=None, pre_norm=False, loose_type=True,
                    sim_header='meanP', linear_patch='2d')
    video = torch.from_numpy(g["video"]).view(B, 1, T, 3, 64, 64)
    ids = torch.from_numpy(g["t_ids"])[:B]
    batch = (ids, (ids > 0).long(), torch.zeros_like(ids), video, torch.ones(B, 1, T, dtype=torch.long))
    args = Namespace(lr=1e-3, wd=0.2, new_added_modules=["Cross", "cluster_embed"], gradient_accumulation_steps=1, clip_grad_norm=1.0)

    def run(scaler):
        model = CLIP4Clip.from_state_dict(dict(sd), cfg).float().to(DEV)
        opt = BertAdam(prep_optim_params_groups(args, model, coef_lr=1.0), lr=args.lr, warmup=0.1, t_total=40,
                       schedule='warmup_cosine', b1=0.9, b2=0.98, e=1e-6, max_grad_norm=1.0)
        loss, gs = train_epoch(0, args, model, [batch] * 2, DEV, opt, 0, scaler=scaler)
        return loss, gs, {n: p.detach().clone() for n, p in model.named_parameters()}

    l0, g0, p0 = run(None)
    sc = torch.amp.GradScaler('cuda', init_scale=2.0 ** 10, growth_interval=1000)
    l1, g1, p1 = run(sc)
    assert g0 == g1 == 2 and abs(l0 - l1) <= 1e-4 * max(1.0, abs(l0)) and sc.get_scale() == 2.0 ** 10
    worst = max(float((p1[n] - p0[n]).abs().max() / p0[n].abs().max().clamp_min(1e-6)) for n in p0)
    print(f"[scaler] worst relative parameter difference after 2 steps: {worst:.2e}")
    assert worst <= 2e-3            # (observed: 0 - a power-of-two scale passes through the backward exactly)
    init = {n: p.detach().clone() for n, p in CLIP4Clip.from_state_dict(dict(sd), cfg).float().to(DEV).named_parameters()}
    assert max(float((p0[n] - init[n]).abs().max()) for n in p0) > 0            # (the steps did move the parameters)
    # a scale that overflows the gradients (inf): both steps skipped by GradScaler.step, parameters untouched
    big = torch.amp.GradScaler('cuda', init_scale=float("inf"), growth_interval=1000)
    l2, g2, p2 = run(big)
    assert g2 == 2 and abs(l2 - l0) <= 0.5 * max(1.0, abs(l0))              # (the reported loss is the unscaled one)
    assert all(torch.equal(p2[n], init[n]) for n in p2 if n != "clip.logit_scale")


# ------------------------------------------------------------------------------------------------ loose k-medoids thresholds
@pytest.fixture(scope="module")
def g6():
    return np.load(os.path.join(HERE, "golden", "r6_golden.npz"))


def _loose_cases():
    from oracle.recipes import LOOSE_THRESHOLD_CASES
    return sorted(LOOSE_THRESHOLD_CASES)


@pytest.mark.parametrize("tag", _loose_cases())
def test_kmedoids_loose_threshold_against_reference(g6, tag):
    """fast_kmeans.py:85-88 literally (was CC_ERR_UNSUPPORTED): a threshold between two values of the reference's own
    center_shift sequence stops a split chunk in mid-course - the reference's medoids / assignment bit for bit, the number of
    iterations each chunk executed, and NOT the fixed point's answer."""
    from oracle.recipes import LOOSE_THRESHOLD_CASES, loose_threshold_inputs
    from centerclip_amd.cluster.fast_kmeans import _run
    seed, P, N, W, K, split, iters, distance, pre_norm, id_sort, _, _ = LOOSE_THRESHOLD_CASES[tag]
    X = torch.from_numpy(loose_threshold_inputs(tag)).to(DEV)
    thr = float(g6[f"{tag}_threshold"][0])
    a, m, it = _run(X, K, distance, thr, iters, id_sort, 2.0, split if P > split else P, pre_norm, return_iters=True)
    assert np.array_equal(m.cpu().numpy(), g6[f"{tag}_medoids"].astype(np.int64))
    assert np.array_equal(a.cpu().numpy(), g6[f"{tag}_assign"].astype(np.int64))
    steps = g6[f"{tag}_steps"]
    want_it = np.concatenate([np.full(min(split, P - c * split), steps[c]) for c in range(len(steps))])
    assert np.array_equal(it.cpu().numpy(), want_it), (it.cpu().numpy(), want_it)
    _, m_fix = _run(X, K, distance, 1e-6, 100, id_sort, 2.0, split if P > split else P, pre_norm)
    assert not torch.equal(m, m_fix)
    a2, m2, _ = _run(X, K, distance, thr, iters, id_sort, 2.0, split if P > split else P, pre_norm, return_iters=True)
    assert torch.equal(a, a2) and torch.equal(m, m2)


def test_kmedoids_loose_threshold_seeded_sweep_against_oracle():
    """Seeded problems (inputs exact in the fp16 split of the distance kernel: multiples of 1/32), thresholds spread over the
    range of the shifts, with and without id_sort, ragged last chunks, iter_limit 1 / 3 / 60: the HIP path == the literal
    oracle (oracle/cluster_oracle.py literal_batch_kmedoids_with_split), indices bit for bit (euclidean, level P1).  Cosine /
    pre_norm on such inputs are level P3 (the normalisation rounds): the k-medoids objective within 1 % of the
    oracle's at the same loose threshold, and the executed iteration counts equal."""
    from oracle import cluster_oracle as co
    from centerclip_amd.cluster.fast_kmeans import _run
    rng = np.random.RandomState(606)
    exact = 0
    for case in range(16):
        P, N, W = int(rng.randint(1, 8)), int(rng.choice([9, 40, 98, 196, 230])), int(rng.choice([8, 32, 64, 100]))
        K = int(rng.randint(2, max(3, N // 4)))
        split = int(rng.choice([1, 2, 3, 4, 16]))
        distance = "cosine" if case % 5 == 4 else "euclidean"
        pre_norm, id_sort = bool(case % 6 == 5), bool(case % 3 != 0)
        X = torch.from_numpy((rng.randint(-64, 65, size=(P, N, W)) / 32.0).astype(np.float32))
        scale = float(np.sqrt(2 * W) * 2.0 * K) if distance == "euclidean" and not pre_norm else float(1.4 * K)
        thr = float(scale * rng.choice([0.02, 0.1, 0.3, 0.6]))
        iters = int(rng.choice([1, 3, 60]))
        chunk = split if P > split else P
        a, m, it = _run(X.to(DEV), K, distance, thr, iters, id_sort, 2.0, chunk, pre_norm, return_iters=True)
        outs = [co.literal_batch_kmedoids(c, K, distance, thr, iters, id_sort, 2.0, return_steps=True)
                for c in torch.split(X / (X.norm(dim=-1, keepdim=True) + 1e-6) if pre_norm else X, chunk, dim=0)]
        ao, mo = torch.cat([o[0] for o in outs], 0), torch.cat([o[1] for o in outs], 0)
        steps = np.concatenate([np.full(o[0].shape[0], o[2]) for o in outs])
        what = (case, P, N, W, K, split, distance, pre_norm, id_sort, thr, iters)
        if distance == "euclidean" and not pre_norm:
            assert np.array_equal(m.cpu().numpy(), mo.numpy()), what
            assert np.array_equal(a.cpu().numpy(), ao.numpy()), what
            assert np.array_equal(it.cpu().numpy(), steps), what
            exact += 1
        else:
            Xn = (X / (X.norm(dim=-1, keepdim=True) + 1e-6)).double()
            d = torch.cdist(Xn, Xn)
            obj = lambda med: float(sum(d[p][:, med[p]].min(dim=1).values.sum() for p in range(P)))
            assert abs(obj(m.cpu()) - obj(mo)) <= 0.02 * obj(mo) + 1e-9, what
    assert exact >= 8


def test_token_cluster_module_with_loose_threshold():
    """TokenClusterInter (cluster.py:206-352) with a loose threshold: output rows == gathering the oracle's medoids."""
    from oracle import cluster_oracle as co
    from centerclip_amd.cluster import TokenClusterInter
    g = torch.Generator().manual_seed(12)
    B, T, T_new, n, W, K = 2, 4, 2, 16, 64, 6
    x = torch.randint(-64, 65, (1 + n, B * T, W), generator=g).float() / 32.0      # (exact in the distance kernel's fp16 split)
    mod = TokenClusterInter(algorithm='kmediods++', block_id=1, before_cluster_num=n, cluster_num=K, before_block_frames=T,
                            after_block_frames=T_new, original_frame=T, distance='euclidean', threshold=25.0, iter_limit=60,
                            id_sort=True, aggregation=None, split_size=8, norm_p=2.0)
    y = mod(x.to(DEV))
    y = y[0] if isinstance(y, tuple) else y
    ref = co.literal_token_cluster(x, T, T_new, K, "euclidean", 25.0, 60, split_size=8)
    ref = ref[0] if isinstance(ref, tuple) else ref
    assert y.shape == ref.shape and float((y.cpu() - ref).abs().max()) <= 1e-5


# ------------------------------------------------------------------------------------------------ N above 4,095
@pytest.mark.parametrize("tag", ["p1w_4500", "p1w_8191"])
def test_p1_lattice_above_4095_tokens(g6, tag):
    """batch_fast_kmedoids_with_split at N = 4,500 (two chunks) and N = 8,191 (the largest length at which ATen's row sum
    stays within two accumulator levels: 255 passes of 32 terms) - the reference's indices bit for bit (was CC_ERR_UNSUPPORTED
    above 4,095: 13-bit token ids in the member lists, up to 128 mask words per cluster)."""
    from centerclip_amd import cluster as cl
    from oracle.recipes import P1_WIDE_CASES, lattice
    seed, P, N, W, K, split, iters = P1_WIDE_CASES[tag]
    X = torch.from_numpy(lattice(seed, (P, N, W))).to(DEV)
    a, m = cl.batch_fast_kmedoids_with_split(X, K, distance="euclidean", threshold=1e-6, iter_limit=iters,
                                             id_sort=True, norm_p=2.0, split_size=split, pre_norm=False)
    assert np.array_equal(m.cpu().numpy(), g6[f"{tag}_medoids"].astype(np.int64))
    assert np.array_equal(a.cpu().numpy(), g6[f"{tag}_assign"].astype(np.int64))


def test_loose_threshold_inside_the_fused_tower_and_a_hipgraph(g2):
    """The stepped selection (2 * iter_limit + 2 launches, a memset, no host read) inside the fused visual tower: the features
    of a model whose cluster module carries a loose threshold differ from the tight-threshold model's, a captured hipGraph of the
    forward replays to the eager result bit for bit, and the medoids the tower used are the stand-alone op's."""
    from argparse import Namespace
    from centerclip_amd.clip4clip import CLIP4Clip
    sd = {k[6:]: torch.from_numpy(g2[k].astype(np.float32) if g2[k].dtype == np.float16 else g2[k])
          for k in g2.files if k.startswith("s1_sd/")}
    cfg = g2["s1_cfg"]
    RES, T, T_new, B = int(cfg[1]), int(cfg[11]), int(cfg[12]), int(cfg[10])

    def build(threshold):
        a = Namespace(cluster_inter=1, deep_cluster=0, cluster_algo='kmediods++', max_frames=T,
                      target_frames_blocks=[4, T_new, T_new], cluster_num_blocks=[16, 6, 6], cluster_distance='euclidean',
                      cluster_threshold=threshold, cluster_iter_limit=20, minkowski_norm_p=2.0, aggregation=None,
                      pretrained_clip_name='ViT-B/32', pre_norm=False, loose_type=True, sim_header='meanP', linear_patch='2d',
                      pre_visual_pooling=0)
        return CLIP4Clip.from_state_dict(dict(sd), a).to(DEV).eval()
    gen = torch.Generator().manual_seed(21)
    video = torch.randn(B * T, 3, RES, RES, generator=gen).to(DEV)
    tight, loose = build(1e-6), build(30.0)
    with torch.no_grad():
        f_t, _ = tight.clip.visual.encode(video, T, want_medoids=True)
        m_t = tight.clip.visual.last_medoids.clone()
        f_l, _ = loose.clip.visual.encode(video, T, want_medoids=True)
        m_l = loose.clip.visual.last_medoids.clone()
        assert not torch.equal(m_t, m_l) and not torch.equal(f_t, f_l)
        f_l2, _ = loose.clip.visual.encode(video, T)                       # the shipped call (no medoid buffer)
        assert torch.equal(f_l, f_l2)
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            f_g, _ = loose.clip.visual.encode(video, T)
        gph.replay()
        torch.cuda.synchronize()
        assert torch.equal(f_g, f_l)


# ------------------------------------------------------------------------------------------------ full-size properties
@pytest.mark.parametrize("cfg", ["cfg2", "cfg5"])
def test_full_size_batch_composition_independence(cfg):
    """A size-independent property at BASELINE.json's full shapes (12-layer towers, full width, the per-GPU batch): the features
    of a clip / caption do not depend on what else is in the batch beyond rounding.  Given the medoid sets of the full-batch
    run, encoding a sub-batch (one clip; three clips) runs every launch on other tile shapes (pick_tile by M, the in_proj +
    attention tile height by the number of sequences), other workgroup counts and other row offsets (caption compaction).  A
    GEMM row's k order and a sequence's attention are the same arithmetic in every form; what moves is the association of the
    folded LayerNorm's partial row sums (one slot per tile column x wave column) and with it single fp16 roundings of the centred
    copy - the L2-normalised visual embeddings agree to 1e-4, a tenth of the contract's 1e-3 (measured 2.9e-5 / 2.4e-5); the text
    tower, whose launches keep their tiles at every caption count, gives the same bits."""
    import bench
    from centerclip_amd.clip4clip import CLIP4Clip
    c = dict(bench.FORWARD_CFGS[cfg])
    if cfg == "cfg5":
        c["B"] = 8                                                   # (half its per-GPU batch: the property, not the time)
    sd = bench.random_state_dict(c, seed=0)
    model = CLIP4Clip.from_state_dict(dict(sd), bench.task_config(c)).to(DEV).eval()
    ids, amask, video, vmask = bench.synthetic_batch(c, DEV, seed=41)
    B, T, Tn = c["B"], c["T"], c["T_new"]
    frames = video.view(B * T, 3, c["res"], c["res"])
    vis = model.clip.visual
    nrm = lambda x: x / x.norm(dim=-1, keepdim=True)
    worst_v = worst_t = 0.0
    with torch.no_grad():
        full, _ = vis.encode(frames, T, want_medoids=True)
        med = vis.last_medoids.clone()                               # [T_new * B, K], problem p = segment * B + clip
        tfull = model.clip.encode_text(ids)
        assert torch.isfinite(full).all() and torch.isfinite(tfull).all()
        again, _ = vis.encode(frames, T, forced_medoids=med)
        assert torch.equal(again, full)                              # (the same batch: the same bits)
        for sub in ([0], [B - 1], [1, B // 2, B - 2]):
            rows = torch.tensor([s_ * B + b for s_ in range(Tn) for b in sub], device=DEV)
            fr = video[sub].reshape(len(sub) * T, 3, c["res"], c["res"]).contiguous()
            part, _ = vis.encode(fr, T, forced_medoids=med[rows].contiguous())
            want = full.view(B, Tn, -1)[sub].reshape(len(sub) * Tn, -1)
            worst_v = max(worst_v, float((nrm(part) - nrm(want)).abs().max()))
            tpart = model.clip.encode_text(ids[sub].contiguous())
            worst_t = max(worst_t, float((nrm(tpart) - nrm(tfull[sub])).abs().max()))
    print(f"[{cfg}] sub-batch vs full batch, normalised embeddings: visual {worst_v:.1e}, text {worst_t:.1e}")
    assert worst_v <= 1e-4 and worst_t == 0.0
