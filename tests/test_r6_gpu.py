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
