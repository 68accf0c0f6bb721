"""Round-2 parity tests on the MI355X (``-m gpu``), all through torch.ops.centerclip / the C ABI:

  * the reference's own CLIP4Clip.forward -> get_similarity_logits fixtures (with / without pre_visual_pooling, masks
    with zeros, a fully masked clip), the training branch's loss values, CrossEn                   [S1, S2, N4]
  * block-level API: ResidualAttentionBlock.forward / Transformer.forward on LND activations, return_hidden,
    CLIP.forward                                                                                    [V2, boundary]
  * the eval loop on a toy dataset (the reference-generated matrices are in tests/test_r3_gpu.py)               [S3]
  * similarity at the north-star size 10k x 1k (and a ragged 9,999 x 1,003) vs the oracle, + rank counts   [S3, N1]
  * pre_norm=True: reference fixture (bit-exact) and objective gap vs the oracle on generic floats   [C2]
  * folded LayerNorm on rows with |mean| / sigma = 10, 100 and 100x outlier channels                [V2]
  * full-width forwards at B >= 2 for cfg3 (12 -> 4), cfg4 (64 -> 8) and cfg5 (ViT-B/16)           [V1]
  * plans with two cluster blocks (medoids buffer of the last one), the threshold contract, packed records.

Tolerances as in tests/test_clip_gpu.py (north star: 1e-3 on L2-normalised embeddings / cosine similarities, bit-exact
indices).
"""
import math
import os
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import clip_oracle as clo
from oracle import cluster_oracle as co
from oracle.recipes import PRENORM_CASES, norm32_tokens

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
R2 = os.path.join(HERE, "golden", "r2_golden.npz")
CLIPG = os.path.join(HERE, "golden", "clip_golden.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(R2)


@pytest.fixture(scope="module")
def gc():
    return np.load(CLIPG)


def nrm(x):
    return x / x.norm(dim=-1, keepdim=True)


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def s1_state_dict(g):
    return {k[6:]: torch.from_numpy(g[k].astype(np.float32) if g[k].dtype == np.float16 else g[k])
            for k in g.files if k.startswith("s1_sd/")}


def s1_args(T, T_new, **kw):
    a = Namespace(cluster_inter=1, deep_cluster=0, cluster_algo='kmediods++', max_frames=T,
                  target_frames_blocks=[4, T_new, T_new], cluster_num_blocks=[16, 6, 6], cluster_distance='euclidean',
                  cluster_threshold=1e-6, cluster_iter_limit=100, minkowski_norm_p=2.0, aggregation=None,
                  pretrained_clip_name='ViT-B/32', pre_norm=False, loose_type=True, sim_header='meanP', linear_patch='2d',
                  pre_visual_pooling=0)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def s1_model(g, **kw):
    from centerclip_amd.clip4clip import CLIP4Clip
    T, T_new = int(g["s1_cfg"][11]), int(g["s1_cfg"][12])
    return CLIP4Clip.from_state_dict(s1_state_dict(g), s1_args(T, T_new, **kw)).to(DEV).eval()


# ------------------------------------------------------------------------------------------------ S1 / S2 / N4
@pytest.mark.parametrize("pvp", [0, 1])
def test_s1_full_module_matches_reference(g, pvp):
    """CLIP4Clip.forward (both towers in one enqueue) -> get_similarity_logits against the reference module's own
    outputs, given the reference's medoids; the fully masked clip yields NaN logits exactly where the reference does."""
    model = s1_model(g, pre_visual_pooling=pvp)
    ids, amask = torch.from_numpy(g["s1_ids"]).to(DEV), torch.from_numpy(g["s1_amask"]).to(DEV)
    video, vmask = torch.from_numpy(g["s1_video"]).to(DEV), torch.from_numpy(g["s1_vmask"]).to(DEV)
    seg = torch.zeros_like(ids)
    tag = "s1_pvp%d_" % pvp
    ref_seq, ref_vis, ref_logits = (torch.from_numpy(g[tag + k]) for k in ("seq", "vis", "logits"))
    model.clip.visual.forced_medoids = torch.from_numpy(g["s1_medoids"])
    with torch.no_grad():
        out = model(ids, seg, amask, video, vmask)
        logits, extra = model.get_similarity_logits(out["sequence_output"], out["visual_output"], amask, vmask)
    assert extra == () and out["loss"] is None
    seq, vis = out["sequence_output"].cpu(), out["visual_output"].cpu()
    assert seq.shape == ref_seq.shape and vis.shape == ref_vis.shape
    assert float((nrm(seq) - nrm(ref_seq)).abs().max()) <= 1e-3
    if pvp:            # pooled + normalised [B, E]; the fully masked clip is 0 / 0 = NaN in the reference as well
        assert torch.isnan(vis[2]).all() and torch.isnan(ref_vis[2]).all()
        assert float((vis[:2] - ref_vis[:2]).abs().max()) <= 1e-3
    else:
        assert float((nrm(vis) - nrm(ref_vis)).abs().max()) <= 1e-3
    lg = logits.cpu()
    assert torch.equal(torch.isnan(lg), torch.isnan(ref_logits))
    mult = math.exp(float(s1_state_dict(g)["logit_scale"]))
    assert float((lg[:, :2] - ref_logits[:, :2]).abs().max()) <= 1e-3 * mult
    # text-only and video-only calls (multi-sentence eval, main.py:430,439) give the same features as the paired call
    with torch.no_grad():
        o_t = model(ids, seg, amask)
        o_v = model(video=video, video_mask=vmask)
    assert o_t["visual_output"] is None and o_v["sequence_output"] is None
    assert float((nrm(o_t["sequence_output"].cpu()) - nrm(seq)).abs().max()) <= 2e-4
    model.clip.visual.forced_medoids = None
    # without the hook the module clusters on its own: the medoids it finds are reported next to the reference's
    model.clip.visual.keep_medoids = True
    with torch.no_grad():
        model(ids, seg, amask, video, vmask)
    own = model.clip.visual.last_medoids.cpu().numpy()
    assert own.shape == g["s1_medoids"].shape and (np.diff(own, axis=1) > 0).all()


def test_s1_training_branch_loss_values(g):
    """model.train(): feature all-gather (identity at world size 1) -> logits -> (CrossEn(sim) + CrossEn(sim^T)) / 2,
    against the loss the reference's training branch reports (clip4clip.py:245-262).  Forward values only."""
    from centerclip_amd.losses import CrossEn
    model = s1_model(g).train()
    ids, amask = torch.from_numpy(g["s1_ids"]).to(DEV), torch.from_numpy(g["s1_amask"]).to(DEV)
    video, vmask = torch.from_numpy(g["s1_video"]).to(DEV), torch.from_numpy(g["s1_train_vmask"]).to(DEV)
    model.clip.visual.forced_medoids = torch.from_numpy(g["s1_medoids"])
    out = model(ids, torch.zeros_like(ids), amask, video, vmask)
    assert abs(float(out["loss"].detach()) - float(g["s1_train_loss"])) <= 3e-3
    assert abs(float(out["sim_loss"].detach()) - float(g["s1_train_sim_loss"])) <= 3e-3 and float(out["cluster_loss"]) == 0.0
    sim = torch.from_numpy(g["n4_sim"]).to(DEV)
    got = [float(CrossEn()(sim)), float(CrossEn()(sim.t()))]
    np.testing.assert_allclose(got, g["n4_crossen"], rtol=0, atol=2e-6)


# ------------------------------------------------------------------------------------------------ block-level API
def small_clip(gc, cluster=True):
    from centerclip_amd.clip import build_clip_model
    sd = {k[3:]: torch.from_numpy(gc[k].astype(np.float32) if gc[k].dtype == np.float16 else gc[k])
          for k in gc.files if k.startswith("sd/")}
    T = int(gc["cfg"][11])
    args = Namespace(cluster_inter=1 if cluster else 0, cluster_algo='kmediods++', max_frames=T,
                     target_frames_blocks=[4, 2, 2], cluster_num_blocks=[16, 6, 6], cluster_distance='euclidean',
                     cluster_threshold=1e-6, cluster_iter_limit=100, minkowski_norm_p=2.0, pretrained_clip_name='ViT-B/32',
                     aggregation=None, pre_norm=False)
    model, _ = build_clip_model(dict(sd), args=args)
    return model.to(DEV), sd, T


def test_block_and_transformer_forward_lnd(gc):
    """ResidualAttentionBlock.forward((x, video_frame, cluster_loss)) and Transformer.forward(x, video_frame, visual)
    on the reference's LND activations (clip.py:228-269), vs the oracle block and vs the fused encoder."""
    model, sd, T = small_clip(gc, cluster=False)
    gen = torch.Generator().manual_seed(5)
    W = model.visual.width
    x = torch.randn(17, 8, W, generator=gen)                       # [L, N, W]
    blk = model.visual.transformer.resblocks[0]
    y, vf, closs = blk((x.to(DEV), 4, torch.zeros([], device=DEV)))
    ref = clo.resblock(x.permute(1, 0, 2), sd, "visual.transformer.resblocks.0.", model.visual.heads, causal=False).permute(1, 0, 2)
    assert y.shape == x.shape and vf == 4 and float(closs) == 0.0
    assert relerr(y.cpu(), ref) < 2e-3
    assert torch.equal(x, x.clone())                                # (inputs are never mutated: a fresh tensor comes back)
    # text block: causal
    xt = torch.randn(16, 3, model.transformer.width, generator=gen)
    yt = model.transformer.resblocks[1]((xt.to(DEV), -1, torch.zeros([], device=DEV)))[0]
    reft = clo.resblock(xt.permute(1, 0, 2), sd, "transformer.resblocks.1.", model.transformer.heads, causal=True).permute(1, 0, 2)
    assert relerr(yt.cpu(), reft) < 2e-3
    # the whole Transformer, block by block, equals the oracle chain and (closely) the fused text encoder
    full = model.transformer(xt.to(DEV))
    r = xt.permute(1, 0, 2)
    for i in range(model.transformer.layers):
        r = clo.resblock(r, sd, "transformer.resblocks.%d." % i, model.transformer.heads, causal=True)
    assert relerr(full.cpu(), r.permute(1, 0, 2)) < 3e-3
    tup = model.visual.transformer(x.to(DEV), video_frame=4, visual=True)
    assert isinstance(tup, tuple) and len(tup) == 3 and tup[0].shape == x.shape


def test_transformer_forward_with_cluster_block_matches_fused_encoder(gc):
    """Driving the visual Transformer block by block (cluster module inside block 2, LND in / out) reproduces the fused
    VisualTransformer.forward on lattice-free inputs when both use the same medoids."""
    model, sd, T = small_clip(gc, cluster=True)
    video = torch.from_numpy(gc["video"]).to(DEV)
    # the input of block 1 from the oracle's stem (conv + cls + pos + ln_pre), then block by block on LND activations
    W, p = model.visual.width, model.visual.patch_size
    x = F.conv2d(video.cpu().float(), sd["visual.conv1.weight"].float(), stride=p)
    x = x.reshape(x.shape[0], W, -1).permute(0, 2, 1)
    x = torch.cat([sd["visual.class_embedding"].float().expand(x.shape[0], 1, W), x], 1) + sd["visual.positional_embedding"].float()
    x = clo.layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"]).permute(1, 0, 2).contiguous()     # LND
    out, vf, closs = model.visual.transformer(x.to(DEV), video_frame=T, visual=True)
    own = model.visual.transformer.resblocks[1].tokencluster_inter.last_medoids
    assert own is not None and vf == T and float(closs) == 0.0
    _, hidden = model.visual.encode(video, T, want_hidden=True, forced_medoids=own)        # fused, frame-major [N', L', W]
    out_nld = out.permute(1, 0, 2).cpu()
    assert out_nld.shape == hidden.shape
    assert relerr(out_nld, hidden.cpu()) < 5e-3
    ref, refh = clo.visual_forward(sd, video.cpu(), T, cluster_plan={1: (2, 6)}, forced_medoids={1: own.cpu()}, return_hidden=True)
    assert relerr(out_nld, refh) < 5e-3


def test_return_hidden_and_clip_forward(gc):
    model, sd, T = small_clip(gc, cluster=True)
    video, ids = torch.from_numpy(gc["video"]).to(DEV), torch.from_numpy(gc["t_ids"]).to(DEV)
    model.visual.forced_medoids = None
    x, hidden = model.encode_image(video, return_hidden=True, video_frame=T)
    feat, closs = model.encode_image(video, video_frame=T)
    assert hidden.shape == (feat.shape[0], 7, model.embed_dim) and float(closs) == 0.0
    assert float((x - hidden[:, 0, :]).abs().max()) == 0.0
    assert relerr(x.cpu(), feat.cpu()) < 1e-5                        # the CLS row of the all-token projection
    # all-token projection vs the oracle on the same hidden state
    _, h = model.visual.encode(video, T, want_hidden=True)
    ref = clo.layer_norm(h.cpu(), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) @ sd["visual.proj"].float()
    assert relerr(hidden.cpu(), ref) < 1e-4
    xt, ht = model.encode_text(ids, return_hidden=True)
    assert ht.shape == (ids.shape[0], ids.shape[1], model.embed_dim)
    eot = ids.argmax(-1)
    assert relerr(ht[torch.arange(ids.shape[0]), eot].cpu(), xt.cpu()) < 1e-5
    np.testing.assert_allclose(nrm(xt.cpu()).numpy(), nrm(torch.from_numpy(gc["t_feat"])).numpy(), rtol=0, atol=1e-3)
    # CLIP.forward(image, text) (clip.py:498-512) on a model without clustering
    m2, sd2, _ = small_clip(gc, cluster=False)
    li, lt = m2(video[:3], ids)
    fi = nrm(clo.visual_forward(sd2, video[:3].cpu(), 1))
    ft = nrm(clo.text_forward(sd2, ids.cpu()))
    mult = math.exp(float(sd2["logit_scale"]))
    want = mult * fi @ ft.t()
    assert li.shape == (3, 3) and lt.shape == (3, 3)
    assert float((li.cpu() - want).abs().max()) <= 1e-3 * mult and float((lt.cpu() - want.t()).abs().max()) <= 1e-3 * mult


def test_caption_compaction_is_bit_identical_to_all_rows(gc):
    """The text tower runs on the rows up to each caption's EOT (found on the device); the features equal the all-rows
    run bit for bit - also for a caption without padding, one of length 1, both towers paired, and through the row
    policy that turns the compaction off."""
    model, sd, T = small_clip(gc, cluster=True)
    CTX, VOCAB = int(gc["cfg"][5]), int(gc["cfg"][6])
    gen = torch.Generator().manual_seed(4)
    lens = [CTX, 1, 2, 5, 9, CTX - 1, 3, 7]
    ids = torch.zeros(len(lens), CTX, dtype=torch.long)
    for b, ln in enumerate(lens):
        if ln > 1:
            ids[b, 0] = VOCAB - 2
            ids[b, 1:ln - 1] = torch.randint(1, VOCAB - 2, (max(ln - 2, 0),), generator=gen)
        ids[b, ln - 1] = VOCAB - 1                                   # EOT = largest id, at position ln - 1
    ids = ids.to(DEV)
    # (the last block's row selection is a separate saving with its own test; off here so that both forms run the same GEMMs)
    with model.row_policy(all_last_block_rows=True):
        feat = model.encode_text(ids)
        dense_feat, hidden = model.encode_text(ids, return_hidden=True)   # the all-rows path (hidden state requested)
        assert torch.equal(feat, dense_feat)
        video = torch.from_numpy(gc["video"]).to(DEV)
        with model.row_policy(all_text_rows=True, all_last_block_rows=True):
            off = model.encode_text(ids)
            v_off, t_off = model.encode_pair(video, ids[:3], video_frame=T)
        assert torch.equal(feat, off)
        v_on, t_on = model.encode_pair(video, ids[:3], video_frame=T)
        assert torch.equal(t_on, t_off) and torch.equal(v_on, v_off)
    ref = clo.text_forward(sd, ids.cpu())
    assert float((nrm(feat.cpu()) - nrm(ref)).abs().max()) <= 1e-3
    # hidden rows behind the EOT exist in the all-rows run (the reference computes them too)
    assert hidden.shape == (len(lens), CTX, model.embed_dim) and bool(torch.isfinite(hidden).all())


def test_last_block_runs_on_the_rows_the_heads_read(gc):
    """Without a request for the hidden state the last block of each tower computes out_proj / c_fc / c_proj for the CLS
    (visual) and EOT (text) rows only.  The same rows of the all-rows run differ only through the summation order of the
    GEMMs (another K split, another partition of the LayerNorm partial sums), which can flip the rounding of an fp16
    intermediate (the QuickGELU output, the fp16 copy of the residual row): observed <= 8e-5 relative on a few rows, 5e-6
    typically; asserted at 2e-4, and both forms are equally far from the fp32 oracle (checked below for the text tower)."""
    for cluster in (True, False):
        model, sd, T = small_clip(gc, cluster=cluster)
        video = torch.from_numpy(gc["video"]).to(DEV)
        ids = torch.from_numpy(gc["t_ids"]).to(DEV)
        v_sel, t_sel = model.encode_pair(video, ids, video_frame=T)
        only_v = model.encode_image(video, video_frame=T)[0]
        only_t = model.encode_text(ids)
        with model.row_policy(all_last_block_rows=True):
            v_all, t_all = model.encode_pair(video, ids, video_frame=T)
        full_v, hid_v = model.encode_image(video, video_frame=T, return_hidden=True)
        # the towers in different modes (captions not compacted -> all text rows, CLS rows only on the visual side)
        with model.row_policy(all_text_rows=True):
            v_mix, t_mix = model.encode_pair(video, ids, video_frame=T)
        for a, b in ((v_sel, v_all), (t_sel, t_all), (only_v, v_all), (only_t, t_all), (full_v, v_all), (v_mix, v_all),
                     (t_mix, t_all)):
            assert a.shape == b.shape and relerr(a.cpu(), b.cpu()) < 2e-4
        assert bool(torch.isfinite(hid_v).all())
    # more selected rows than one 64-row chunk of the few-rows kernel (cfg 3 has 256 per rank): 35 clips -> 70 CLS rows, 70 EOT rows
    model, sd, T = small_clip(gc, cluster=True)
    RES, CTX, VOCAB = int(gc["cfg"][1]), int(gc["cfg"][5]), int(gc["cfg"][6])
    gen = torch.Generator().manual_seed(21)
    video = torch.randn(35 * T, 3, RES, RES, generator=gen).to(DEV)
    ids = torch.randint(1, VOCAB - 2, (70, CTX), generator=gen)
    eot = torch.randint(2, CTX, (70,), generator=gen)
    for b in range(70):
        ids[b, eot[b]] = VOCAB - 1
        ids[b, eot[b] + 1:] = 0
    ids = ids.to(DEV)
    v_sel, t_sel = model.encode_pair(video, ids, video_frame=T)
    with model.row_policy(all_last_block_rows=True):
        v_all, t_all = model.encode_pair(video, ids, video_frame=T)
    assert v_sel.shape[0] == 70 and relerr(v_sel.cpu(), v_all.cpu()) < 2e-4 and relerr(t_sel.cpu(), t_all.cpu()) < 2e-4
    ref = nrm(clo.text_forward(sd, ids.cpu()))
    e_sel, e_all = (nrm(t_sel.cpu()) - ref).abs().max(), (nrm(t_all.cpu()) - ref).abs().max()
    assert float(e_sel) <= 1e-3 and float(e_all) <= 1e-3 and abs(float(e_sel) - float(e_all)) <= 1e-4


# ------------------------------------------------------------------------------------------------ eval loop (S3)
# (the block-by-block similarity loop is checked against the reference's own matrix in tests/test_r3_gpu.py)


class _ListLoader(list):
    pass


@pytest.mark.parametrize("multi", [False, True])
def test_eval_epoch_single_and_multi_sentence(g, multi):
    """eval_epoch (main.py:381-499) on a toy dataset: feature caching, the similarity matrix and the on-device metrics
    against metrics computed from the oracle's matrix with the reference's NumPy definition (utils/metrics.py:11-26)."""
    from centerclip_amd.eval import eval_epoch
    model = s1_model(g)
    T, RES, CTX, VOCAB = int(g["s1_cfg"][11]), int(g["s1_cfg"][1]), int(g["s1_cfg"][5]), int(g["s1_cfg"][6])
    gen = torch.Generator().manual_seed(21)
    nvid, per = 6, 2 if multi else 1
    n = nvid * per
    video = torch.randn(nvid, 1, T, 3, RES, RES, generator=gen)
    ids = torch.zeros(n, 1, CTX, dtype=torch.long)
    for b in range(n):
        ln = int(torch.randint(4, CTX + 1, (1,), generator=gen))
        ids[b, 0, 0], ids[b, 0, ln - 1] = VOCAB - 2, VOCAB - 1
        ids[b, 0, 1:ln - 1] = torch.randint(1, VOCAB - 2, (ln - 2,), generator=gen)
    vmask = torch.ones(nvid, 1, T, dtype=torch.long)
    loader = _ListLoader()
    ds = Namespace(multi_sentence_per_video=multi)
    if multi:                                       # sentence s describes video s // per; the loader repeats the video
        ds.cut_off_points = [per * (v + 1) for v in range(nvid)]
        ds.sentence_num, ds.video_num = n, nvid
        vrep, mrep = video.repeat_interleave(per, 0), vmask.repeat_interleave(per, 0)
    else:
        vrep, mrep = video, vmask
    for s in range(0, n, 4):
        loader.append((ids[s:s + 4], (ids[s:s + 4] > 0).long(), torch.zeros_like(ids[s:s + 4]), vrep[s:s + 4], mrep[s:s + 4]))
    loader.dataset = ds
    r1, t_inf, info = eval_epoch(model, loader, torch.device(DEV), args=Namespace(inference_speed_test=False))
    assert 0.0 <= r1 <= 100.0 and t_inf > 0 and len(info) == 4 and info[0] == "Text-to-Video:"
    # the same numbers from the oracle's features
    sd = s1_state_dict(g)
    with torch.no_grad():
        seq = clo.text_forward(sd, ids.view(n, CTX)).view(n, 1, -1)
        m = model.clip.visual
        feats, _ = m.encode(video.view(-1, 3, RES, RES).to(DEV), T)
        vis = feats.view(nvid, -1, feats.shape[-1]).cpu()
        sim = clo.loose_similarity(seq, vis, torch.ones(nvid, vis.shape[1], dtype=torch.long), float(sd["logit_scale"])).numpy()
    if multi:
        ranks = []
        for s in range(n):
            gt = s // per
            ranks.append(int((sim[s] > sim[s, gt]).sum()))
        want_r1 = 100.0 * np.mean(np.array(ranks) == 0)
    else:
        order = np.sort(-sim, axis=1)
        d = np.diag(-sim)[:, None]
        ind = np.where(order - d == 0)[1]
        want_r1 = 100.0 * np.mean(ind == 0)
    assert abs(r1 - want_r1) < 1e-3


# ------------------------------------------------------------------------------------------------ S3 at full size
@pytest.mark.parametrize("Nt,Nv", [(10000, 1000), (9999, 1003)])
def test_similarity_matrix_north_star_size(Nt, Nv):
    """The 10k x 1k matrix the north star names (and a ragged size): pooled + normalised + ONE NT GEMM vs the oracle in
    float64, <= 1e-3 * exp(logit_scale) (it is ~1e-5), and the rank counts at that size vs NumPy."""
    from centerclip_amd import ops
    from centerclip_amd.metrics import rank_counts
    Tn, E, scale = 3, 512, 1.5
    gen = torch.Generator().manual_seed(Nt)
    t = torch.randn(Nt, E, generator=gen)
    v = torch.randn(Nv, Tn, E, generator=gen)
    m = (torch.rand(Nv, Tn, generator=gen) > 0.2).long()
    m[5] = 0                                                        # one fully masked clip
    m[:, 0] |= (m.sum(1) == 0).long() * (torch.arange(Nv) != 5).long()
    logits = ops.loose_similarity(t.to(DEV), v.to(DEV), m.to(DEV), scale)
    vd = v.double() / v.double().norm(dim=-1, keepdim=True)
    md = m.double().unsqueeze(-1)
    den = md.sum(1)
    den[den == 0] = 1
    pooled = (vd * md).sum(1) / den
    pooled = pooled / pooled.norm(dim=-1, keepdim=True)
    td = t.double() / t.double().norm(dim=-1, keepdim=True)
    want = math.exp(scale) * td @ pooled.t()
    got = logits.cpu().double()
    ok = torch.ones(Nv, dtype=torch.bool)
    ok[5] = False
    assert torch.isnan(got[:, 5]).all()                             # 0 / 0, as in the reference
    err = float((got[:, ok] - want[:, ok]).abs().max())
    assert err <= 1e-3 * math.exp(scale), err
    assert err <= 5e-5                                              # what exact-fp32 MFMA actually delivers
    # rank counts on the device at this size (text -> video: ground truth = column i for the first Nv rows)
    sub = logits[:Nv, :].clone()
    sub[:, 5] = -1e30
    counts = rank_counts(sub).cpu().numpy()
    s = sub.cpu().numpy()
    d = np.diag(s)[:, None]
    np.testing.assert_array_equal(counts[:, 0], (s > d).sum(1))
    np.testing.assert_array_equal(counts[:, 1], (s == d).sum(1))
    # the pre-pooled branch (2-D visual_output) gives the same matrix
    pooled_dev = ops.video_pool_normalize(v.to(DEV), m.to(DEV))
    alt = ops.scaled_dot_nt(ops.normalize_rows(t.to(DEV)), pooled_dev, math.exp(scale))
    assert float((alt[:, ok.to(DEV)] - logits[:, ok.to(DEV)]).abs().max()) <= 2e-5


def test_packed_records_similarity_equals_plain():
    """dist.PackedFeatures: features written into the record, gather() at world size 1, logits read in place."""
    from centerclip_amd import ops
    from centerclip_amd.dist import PackedFeatures
    B, Tn, E = 16, 3, 512
    gen = torch.Generator().manual_seed(3)
    pf = PackedFeatures(B, Tn, E, torch.device(DEV), world=1)
    vis, seq = torch.randn(B * Tn, E, generator=gen), torch.randn(B, E, generator=gen)
    mask = (torch.rand(B, Tn, generator=gen) > 0.3).long()
    mask[:, 0] = 1
    pf.vis.copy_(vis)
    pf.seq.copy_(seq)
    pf.mask.copy_(mask)
    rec = pf.gather()
    assert rec.shape == (1, pf.rec) and pf.rec % 16 == 0
    got = pf.logits(pf.seq, 1.25)
    want = ops.loose_similarity(seq.to(DEV), vis.view(B, Tn, E).to(DEV), mask.to(DEV), 1.25)
    assert torch.equal(got, want)
    # two records laid out as a 2-rank gather would leave them: videos of "rank 1" are found through the group stride
    two = torch.cat([pf.send, pf.send]).view(2, pf.rec)
    g2 = ops.loose_similarity_packed(seq.to(DEV), two, B, Tn, E, pf.vis_off, pf.mask_off, 1.25)
    assert g2.shape == (B, 2 * B) and torch.equal(g2[:, :B], want) and torch.equal(g2[:, B:], want)
    assert torch.equal(pf.gathered_text(), seq.to(DEV))


# ------------------------------------------------------------------------------------------------ C2: pre_norm
@pytest.mark.parametrize("tag", sorted(PRENORM_CASES))
def test_pre_norm_reference_fixture_bit_exact(g, tag):
    """pre_norm=True, euclidean: tokens of norm exactly 32 -> X / (|X| + 1e-6) is an exact division by 2^5 in the
    reference, the normalised tokens are a lattice, and the indices are a bit-exact target (level P1)."""
    from centerclip_amd.cluster import batch_fast_kmedoids_with_split
    seed, P, N, W, K, split = PRENORM_CASES[tag]
    X = torch.from_numpy(norm32_tokens(seed, (P, N, W))).to(DEV)
    a, m = batch_fast_kmedoids_with_split(X, K, distance="euclidean", threshold=1e-6, iter_limit=100, id_sort=True,
                                          norm_p=2.0, split_size=split, pre_norm=True)
    assert np.array_equal(m.cpu().numpy(), g[tag + "_medoids"].astype(np.int64))
    assert np.array_equal(a.cpu().numpy(), g[tag + "_assign"].astype(np.int64))


@pytest.mark.parametrize("tag", sorted(PRENORM_CASES))
def test_pre_norm_cosine_objective_vs_reference_fixture(g, tag):
    """pre_norm=True + cosine re-normalises the unit tokens by 1 / (1 + 1e-6), which rounds: level P3 (not a bit-exact
    target) - the k-medoids objective of the HIP medoids is within 1 % of the reference fixture's."""
    from centerclip_amd.cluster import batch_fast_kmedoids_with_split
    seed, P, N, W, K, split = PRENORM_CASES[tag]
    Xc = torch.from_numpy(norm32_tokens(seed, (P, N, W)))
    _, m = batch_fast_kmedoids_with_split(Xc.to(DEV), K, distance="cosine", threshold=1e-6, iter_limit=100, id_sort=True,
                                          norm_p=2.0, split_size=split, pre_norm=True)
    m, mref = m.cpu(), torch.from_numpy(g[tag + "_cos_medoids"].astype(np.int64))
    xn = (Xc / 32.0).double()
    cos = 1.0 - xn @ xn.transpose(1, 2)

    def obj(med):
        return float(sum(cos[p][:, med[p]].min(dim=1).values.sum() for p in range(P)))
    assert (m[:, 1:] > m[:, :-1]).all()
    assert abs(obj(m) - obj(mref)) <= 0.01 * obj(mref)


def kmedoids_objective(X, medoids):
    d = torch.cdist(X.double(), X.double())
    return float(sum(d[p][:, medoids[p]].min(dim=1).values.sum() for p in range(X.shape[0])))


def test_pre_norm_generic_floats_objective_gap_vs_oracle():
    """P3 contract for pre_norm=True on generic floats: structural invariants + k-medoids objective (on the normalised
    tokens) within 1 % of the oracle's literal_batch_kmedoids_with_split(pre_norm=True)."""
    from centerclip_amd.cluster import batch_fast_kmedoids_with_split
    gen = torch.Generator().manual_seed(77)
    P, N, W, K = 6, 196, 768, 49
    X = torch.randn(P, N, W, generator=gen) * (0.5 + 4 * torch.rand(P, N, 1, generator=gen))      # very different norms
    a, m = batch_fast_kmedoids_with_split(X.to(DEV), K, distance="euclidean", threshold=1e-6, iter_limit=100,
                                          split_size=4, pre_norm=True)
    a, m = a.cpu(), m.cpu()
    ao, mo = co.literal_batch_kmedoids_with_split(X, K, "euclidean", 1e-6, 100, True, 2.0, 4, True)
    Xn = X / (X.norm(dim=-1, keepdim=True) + 1e-6)
    assert (m[:, 1:] > m[:, :-1]).all() and int(m.min()) >= 0 and int(m.max()) < N
    for p in range(P):                                               # every medoid belongs to its own cluster
        assert torch.equal(a[p][m[p]], torch.arange(K))
    obj, obj_o = kmedoids_objective(Xn, m), kmedoids_objective(Xn, mo)
    assert abs(obj - obj_o) <= 0.01 * obj_o
    # clustering the raw tokens instead would differ: the normalisation is really applied
    _, m_raw = batch_fast_kmedoids_with_split(X.to(DEV), K, distance="euclidean", threshold=1e-6, iter_limit=100,
                                              split_size=4, pre_norm=False)
    assert not torch.equal(m_raw.cpu(), m)


def test_threshold_contract():
    """threshold <= 1e-5 (the reference's default and the scripts' 1e-6): every problem runs to its fixed point, one launch;
    a looser threshold runs the reference's literal chunk-mean stop test (round 6; fixtures: tests/test_r6_gpu.py) - here: a
    threshold above every possible shift stops after ONE iteration, which is not the fixed point; NaN is refused."""
    from centerclip_amd.cluster import batch_fast_kmedoids_with_split
    from centerclip_amd._lib import CenterClipHipError
    from centerclip_amd.cluster.fast_kmeans import _run
    X = torch.randn(2, 40, 32, generator=torch.Generator().manual_seed(1)).to(DEV)
    _, m1 = batch_fast_kmedoids_with_split(X, 5, threshold=1e-5)
    _, m2 = batch_fast_kmedoids_with_split(X, 5, threshold=1e-6)
    assert torch.equal(m1, m2)
    _, m3, it3 = _run(X, 5, 'euclidean', 1e9, 60, True, 2.0, 4, False, return_iters=True)
    _, m4, it4 = _run(X, 5, 'euclidean', 1e-6, 1, True, 2.0, 4, False, return_iters=True)
    assert torch.equal(m3, m4) and int(it3.max()) == 1 and int(it4.max()) == 1
    with pytest.raises((CenterClipHipError, ValueError)):
        batch_fast_kmedoids_with_split(X, 5, threshold=float("nan"))


# ------------------------------------------------------------------------------------------------ folded LayerNorm stress
@pytest.mark.parametrize("ratio", [0.0, 10.0, 100.0])
@pytest.mark.parametrize("M,W", [(2400, 768), (512, 512)])
def test_folded_layernorm_large_means_and_outlier_channels(M, W, ratio):
    """Rows with |mean| / sigma = 10 / 100 and three channels at 100x magnitude (real CLIP residual streams have such
    massive-activation channels) through linear_resid_stats_f16 -> linear_ln_f16, vs float64 LayerNorm + Linear.
    The fp16 copy is centred on the row mean of the previous sublayer, so the error stays that of LN-then-fp16."""
    from centerclip_amd import ops
    gen = torch.Generator().manual_seed(int(M + W + ratio))
    h0 = torch.randn(M, W, generator=gen)
    h0[:, [3, 100, W - 7]] *= 100.0                                  # outlier channels
    sig = h0.std(dim=1, keepdim=True)
    sign = (torch.rand(M, 1, generator=gen) > 0.5).float() * 2 - 1
    h0 = h0 - h0.mean(dim=1, keepdim=True) + sign * ratio * sig
    a = torch.randn(M, W, generator=gen).half()
    w1 = (torch.randn(W, W, generator=gen) * W ** -0.5).half()
    b1 = torch.randn(W, generator=gen) * 0.1
    gamma, beta = torch.rand(W, generator=gen) + 0.5, torch.randn(W, generator=gen) * 0.2
    w2 = torch.randn(4 * W, W, generator=gen) * W ** -0.5
    b2 = torch.randn(4 * W, generator=gen) * 0.1
    href = h0.double() + a.double() @ w1.double().t() + b1.double()
    pre = F.layer_norm(href, (W,), gamma.double(), beta.double(), 1e-5) @ w2.double().t() + b2.double()
    h = h0.to(DEV).clone()
    _, st0, sh0 = ops.row_stats(h)                                   # what the previous sublayer left behind
    h16, stats, slots, sh1 = ops.linear_resid_stats_f16(a.to(DEV), w1.to(DEV), b1.to(DEV), h, shift_in=sh0,
                                                        stats_in=st0.view(M, 1, 2))
    assert relerr(h.cpu(), href) < 2e-4
    assert torch.equal(h16, (h - sh1[:, None]).half())               # the copy is centred on the reported shift ...
    assert float((sh1.cpu() - h0.mean(1)).abs().max()) <= 1e-3 * float(h0.abs().max())      # ... = last sublayer's row mean
    wf, c1, c2 = ops.fold_layernorm_linear(w2.to(DEV), b2.to(DEV), gamma.to(DEV), beta.to(DEV))
    y = ops.linear_ln_f16(h16, wf, c1, c2, stats.contiguous(), slots, gelu=False).float().cpu()
    assert relerr(y, pre) < 3e-3, relerr(y, pre)
    # the exactly centred one-slot path (row_stats) agrees
    h16b, st1, _ = ops.row_stats(h)
    y1 = ops.linear_ln_f16(h16b, wf, c1, c2, st1, 1, gelu=False).float().cpu()
    assert relerr(y1, pre) < 3e-3
    if ratio >= 100.0:   # without centring the same pipeline is off by ~|mean|/sigma * 2^-11: the reason for the shift
        h2 = h0.to(DEV).clone()
        h16u, statsu, slotsu, _ = ops.linear_resid_stats_f16(a.to(DEV), w1.to(DEV), b1.to(DEV), h2)
        yu = ops.linear_ln_f16(h16u, wf, c1, c2, statsu.contiguous(), slotsu, gelu=False).float().cpu()
        assert relerr(yu, pre) > 2 * max(relerr(y, pre), 1e-4)


def test_fused_forward_with_large_mean_rows_still_meets_the_contract(gc):
    """The small reference-fixture model with ln_pre.bias shifted by +/-40 sigma: every row of the residual stream gets
    |mean| >> sigma.  Embeddings stay within 1e-3 of the oracle on the same weights (given identical medoids)."""
    from centerclip_amd.clip import build_clip_model
    sd = {k[3:]: torch.from_numpy(gc[k].astype(np.float32) if gc[k].dtype == np.float16 else gc[k])
          for k in gc.files if k.startswith("sd/")}
    sd["visual.ln_pre.bias"] = sd["visual.ln_pre.bias"] + 40.0
    sd["positional_embedding"] = sd["positional_embedding"] + 5.0
    T = int(gc["cfg"][11])
    args = Namespace(cluster_inter=1, cluster_algo='kmediods++', max_frames=T, target_frames_blocks=[4, 2, 2],
                     cluster_num_blocks=[16, 6, 6], cluster_distance='euclidean', cluster_threshold=1e-6,
                     cluster_iter_limit=100, minkowski_norm_p=2.0, pretrained_clip_name='ViT-B/32', aggregation=None,
                     pre_norm=False)
    model, _ = build_clip_model(dict(sd), args=args)
    model = model.to(DEV)
    video, ids = torch.from_numpy(gc["video"]), torch.from_numpy(gc["t_ids"])
    feat, _ = model.visual.encode(video.to(DEV), T, want_medoids=True)
    med = model.visual.last_medoids.cpu()
    ref = clo.visual_forward(sd, video, T, cluster_plan={1: (2, 6)}, forced_medoids={1: med})
    assert float((nrm(feat.cpu()) - nrm(ref)).abs().max()) <= 1e-3
    tf = model.encode_text(ids.to(DEV))
    assert float((nrm(tf.cpu()) - nrm(clo.text_forward(sd, ids))).abs().max()) <= 1e-3


# ------------------------------------------------------------------------------------------------ full-width, B >= 2
FULL = {
    # name: (patch, T, T_new, K, cluster block (1-based), B, split-size model name)
    "cfg3_msvd_12to4": (32, 12, 4, 49, 7, 2, 'ViT-B/32'),
    "cfg4_activitynet_64to8": (32, 64, 8, 49, 7, 2, 'ViT-B/32'),
    "cfg5_vitb16_12to4_k100": (16, 12, 4, 100, 7, 2, 'ViT-B/16'),
}


@pytest.mark.parametrize("name", sorted(FULL))
def test_full_width_forward_batch2(name):
    """Full-size ViT-B towers (12 layers, width 768, random CLIP-init weights) at B = 2 for the cfg3 / cfg4 / cfg5 shapes:
    visual embeddings vs the fp32 oracle given the HIP path's own medoids (multi-chunk coupling included: P = B*T_new
    problems in split-size chunks)."""
    from centerclip_amd.clip import CLIP
    patch, T, T_new, K, cb, B, pname = FULL[name]
    torch.manual_seed(11)
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
    video = torch.randn(B * T, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    feat, _ = model.visual.encode(video.to(DEV), T, want_medoids=True)
    med = model.visual.last_medoids.cpu()
    n = (224 // patch) ** 2
    assert med.shape == (B * T_new, K) and (med[:, 1:] > med[:, :-1]).all() and int(med.max()) < (T // T_new) * n
    with torch.no_grad():
        ref = clo.visual_forward(sd, video, T, cluster_plan={cb - 1: (T_new, K)}, forced_medoids={cb - 1: med})
    assert feat.shape == ref.shape == (B * T_new, 512)
    assert float((nrm(feat.cpu()) - nrm(ref)).abs().max()) <= 1e-3


# ------------------------------------------------------------------------------------------------ multi-cluster plans
def test_two_cluster_blocks_medoids_buffer(gc):
    """A plan with two k-medoids blocks (frames 4 -> 2 -> 1): medoids_out belongs to the LAST one only (the
    earlier, larger block used to write into the same buffer); forced_medoids wants the ids of BOTH blocks (round 5,
    tests/test_clip_gpu.py::test_forced_medoids_with_two_cluster_blocks) - the last block's alone are refused before the
    library would read past them."""
    from centerclip_amd.clip import build_clip_model
    sd = {k[3:]: torch.from_numpy(gc[k].astype(np.float32) if gc[k].dtype == np.float16 else gc[k])
          for k in gc.files if k.startswith("sd/")}
    T = int(gc["cfg"][11])
    args = Namespace(cluster_inter=1, cluster_algo='kmediods++', max_frames=T, target_frames_blocks=[4, 2, 1],
                     cluster_num_blocks=[16, 10, 4], cluster_distance='euclidean', cluster_threshold=1e-6,
                     cluster_iter_limit=100, minkowski_norm_p=2.0, pretrained_clip_name='ViT-B/32', aggregation=None,
                     pre_norm=False)
    model, _ = build_clip_model(dict(sd), args=args)
    model = model.to(DEV)
    video = torch.from_numpy(gc["video"]).to(DEV)
    B = video.shape[0] // T
    guard = torch.full((64,), -7, dtype=torch.long, device=DEV)      # allocated right behind: must stay untouched
    feat, hidden = model.visual.encode(video, T, want_hidden=True, want_medoids=True)
    med = model.visual.last_medoids
    assert feat.shape == (B * 1, model.embed_dim) and hidden.shape == (B, 5, model.visual.width)
    assert med.shape == (B * 1, 4) and int(med.min()) >= 0 and int(med.max()) < 2 * 10 and bool((guard == -7).all())
    assert (med[:, 1:] > med[:, :-1]).all()
    with pytest.raises(ValueError, match="forced_medoids"):
        model.visual.encode(video, T, forced_medoids=med)


def test_spectral_forward_pieces_match_reference(g):
    """N4 (forward pieces of cluster_algo 'spectral'): the normalised Laplacian of the heat-kernel graph - plain and with
    the spatial-temporal mask - within 2e-5 of the reference's, the SVD sign flip exact, the k-medoids tail on an
    embedding, and batch_spectral_clustering end to end with the built-in decomposition and with a caller-supplied one (the
    decomposition itself: tests/test_spectral_gpu.py)."""
    from centerclip_amd.cluster.spectral import (batch_sign_flip_rasmus_bro, batch_spectral_clustering, spectral_laplacian,
                                                  spectral_embedding_kmedoids)
    X = torch.from_numpy(g["sp_x"]).to(DEV)
    sigma = float(g["sp_sigma"])
    lap, aff = spectral_laplacian(X, sigma=sigma, return_affinity=True)
    np.testing.assert_allclose(aff.cpu().numpy(), g["sp_w"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(lap.cpu().numpy(), g["sp_lsym"], rtol=0, atol=2e-5)
    lap_g = spectral_laplacian(X, sigma=sigma, spatial_temporal_graph=torch.from_numpy(g["sp_graph"]))
    np.testing.assert_allclose(lap_g.cpu().numpy(), g["sp_lsym_graph"], rtol=0, atol=2e-5)
    U, S, Vh = (torch.from_numpy(g[k]).to(DEV) for k in ("sp_u", "sp_s", "sp_vh"))
    flipped = batch_sign_flip_rasmus_bro(U, S, Vh)
    assert np.array_equal(flipped.cpu().numpy(), g["sp_u_flipped"])
    # tail: row-normalised embedding -> k-medoids == the k-medoids op with pre_norm on the same embedding
    K = 6
    Q = torch.from_numpy(g["sp_u_flipped"][:, :, -K:].copy()).to(DEV)
    a, m = spectral_embedding_kmedoids(Q, K, norm_p=2.0, threshold=1e-6, iter_limit=100)
    ao, mo = co.literal_batch_kmedoids_with_split(Q.cpu(), K, "euclidean", 1e-6, 100, True, 2.0, Q.shape[0], True)
    Qn = (Q.cpu() / (Q.cpu().norm(dim=-1, keepdim=True) + 1e-6))
    assert abs(kmedoids_objective(Qn, m.cpu()) - kmedoids_objective(Qn, mo)) <= 0.01 * kmedoids_objective(Qn, mo)
    a1, m1 = batch_spectral_clustering(X, K, sigma=sigma, correct_sign=True, norm_p=2.0, threshold=1e-6, iter_limit=100)
    assert a1.shape == (X.shape[0], X.shape[1]) and m1.shape == (X.shape[0], K) and bool((m1[:, 1:] > m1[:, :-1]).all())
    a2, m2 = batch_spectral_clustering(X, K, sigma=sigma, correct_sign=True, norm_p=2.0, threshold=1e-6, iter_limit=100,
                                       eigensolver=lambda L_: torch.linalg.svd(L_, full_matrices=False))
    assert a2.shape == (X.shape[0], X.shape[1]) and m2.shape == (X.shape[0], K) and bool((m2[:, 1:] > m2[:, :-1]).all())


def test_custom_ops_are_registered_with_fake_kernels():
    """torch.ops.centerclip.*: every op has a schema and a fake (meta) kernel; opcheck on two of them with real inputs."""
    from centerclip_amd import torch_ops
    for name in torch_ops.OPS:
        assert hasattr(torch.ops.centerclip, name), name
    a = torch.randn(64, 64, device=DEV).half()
    w = torch.randn(128, 64, device=DEV).half()
    torch.library.opcheck(torch.ops.centerclip.linear_f16.default, (a, w, None, "f16", 0),
                          test_utils=("test_schema", "test_faketensor"))
    x = torch.randn(50, 8, 64, device=DEV)
    torch.library.opcheck(torch.ops.centerclip.token_cluster.default,
                          (x, False, 4, 2, 9, 0, 2.0, 1e-6, 100, 16, False, 0, 0, None, None, None, True),
                          test_utils=("test_schema", "test_faketensor"))


def test_nccl_packed_all_gather_and_sharded_similarity():
    """RCCL on the hardware that is there: world = the number of visible GPUs launched as one process per GPU (skipped
    below 2); on a 1-GPU box the same code runs through an RCCL communicator of size 1 (init, all_gather_into_tensor of
    the packed record, gather_rows), so the nccl backend itself is exercised either way."""
    import subprocess
    import sys
    ndev = torch.cuda.device_count()
    world = ndev if ndev >= 2 else 1
    script = os.path.join(HERE, "nccl_worker.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", "29577", script]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "NCCL_WORKER_OK world=%d" % world in r.stdout
    if world == 1:
        # one GPU: two ranks share it, collectives over gloo - the HIP-backed clip-sharded eval loop at world 2
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29578", script, "--share-gpu"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert "NCCL_WORKER_OK world=2" in r.stdout


def test_bench_line_contract_one_and_two_ranks():
    """bench.py end to end on the device, small step counts: the N = 1 line carries the contract's keys with roofline and
    cpu_baseline objects; `--gpus 2` launches its own two ranks (here sharing the one GPU over gloo, CC_BENCH_SHARE_GPU) and
    reports n_gpus = 2 with the all-gather and the row-sharded similarity."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "1", "--min-seconds", "0.1"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(out.stdout) < 4096 and len(out.stdout.strip().splitlines()) == 1      # ONE compact line, nothing else on stdout
    d = json.loads(out.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "token_cluster_mtokens_per_s", "pairs_per_s"):
        assert key in d, key
    detail = json.load(open(os.path.join(root, d["detail"])))
    assert detail["value"] == d["value"] and "by_symbol_in_situ" in detail["roofline"] and "forward_other_configs" in detail
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1 and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"] and abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    env2 = dict(env, CC_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "1", "--min-seconds", "0.1",
                          "--no-cpu-baseline"], cwd=root, env=env2, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d2 = json.loads(out.stdout.strip().splitlines()[-1])
    assert d2["n_gpus"] == 2 and d2["config"]["global_batch"] == 32 and "feature_all_gather" in d2
    assert len(out.stdout) < 4096
    assert d2["similarity_10k_x_1k"]["sharding"] != "single GPU"
    detail2 = json.load(open(os.path.join(root, d2["detail"])))
    assert "mtokens_per_s_all_ranks" in detail2["token_cluster"]["cfg2"]
    assert d2["token_cluster_mtokens_per_s"] == detail2["token_cluster"]["cfg2"]["mtokens_per_s_all_ranks"]
