"""BASELINE.json configs[0]: "ViT-B/32 single 224^2 frame + 1 caption, CPU-only forward + medoid clustering on
random weights (plumbing, no GPU)".  The CPU leg runs it through the oracle; the GPU leg checks the HIP
path against that oracle for the same weights (frames stay 1, tokens 49 -> 25 at block 7)."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import clip_oracle as clo


def cfg1_args():
    return Namespace(cluster_inter=1, cluster_algo='kmediods++', max_frames=1, target_frames_blocks=[1] * 12,
                     cluster_num_blocks=[49] * 6 + [25] * 6, cluster_distance='euclidean', cluster_threshold=1e-6,
                     cluster_iter_limit=100, minkowski_norm_p=2.0, pretrained_clip_name='ViT-B/32', aggregation=None,
                     pre_norm=False)


def cfg1_inputs():
    g = torch.Generator().manual_seed(11)
    video = torch.randn(1, 3, 224, 224, generator=g)
    ids = torch.zeros(1, 32, dtype=torch.long)
    ids[0, 0], ids[0, 1:9], ids[0, 9] = 49406, torch.randint(1, 49405, (8,), generator=g), 49407
    return video, ids


def cfg1_state_dict():
    from centerclip_amd.clip import CLIP
    torch.manual_seed(5)
    m = CLIP(512, 224, 12, 768, 32, 77, 49408, 512, 8, 12, video_frames=1, args=None)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.half().float())
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def test_cfg1_oracle_cpu_plumbing():
    sd = cfg1_state_dict()
    video, ids = cfg1_inputs()
    with torch.no_grad():
        v, hidden = clo.visual_forward(sd, video, 1, cluster_plan={6: (1, 25)}, return_hidden=True)
        t = clo.text_forward(sd, ids)
        logits = clo.loose_similarity(t.view(1, 1, -1), v.view(1, 1, -1), torch.ones(1, 1, dtype=torch.long), float(sd["logit_scale"]))
    assert hidden.shape == (1, 26, 768) and v.shape == (1, 512) and t.shape == (1, 512) and logits.shape == (1, 1)
    assert bool(torch.isfinite(logits).all()) and abs(float(logits) / np.e) <= 1.0 + 1e-5


@pytest.mark.gpu
def test_cfg1_hip_matches_oracle():
    from centerclip_amd.clip import build_clip_model
    sd = cfg1_state_dict()
    video, ids = cfg1_inputs()
    model, _ = build_clip_model(dict(sd), args=cfg1_args())
    model = model.to("cuda:0")
    feat, _ = model.visual.encode(video.to("cuda:0"), 1, want_medoids=True)
    med = model.visual.last_medoids.cpu()
    assert med.shape == (1, 25)
    with torch.no_grad():
        ref = clo.visual_forward(sd, video, 1, cluster_plan={6: (1, 25)}, forced_medoids={6: med})
        tref = clo.text_forward(sd, ids)
    tfeat = model.encode_text(ids.to("cuda:0")).cpu()
    n = lambda x: x / x.norm(dim=-1, keepdim=True)
    assert float((n(feat.cpu()) - n(ref)).abs().max()) <= 1e-3
    assert float((n(tfeat) - n(tref)).abs().max()) <= 1e-3


@pytest.mark.gpu
def test_vitb16_cfg5_shape_against_oracle():
    """BASELINE.json configs[4] shape: ViT-B/16 (196 tokens/frame), 12 frames -> 4 segments at block 7, K=100
    (N = 588 tokens per problem: the global-memory selection path; L = 197 / 101 attention), one clip."""
    from centerclip_amd.clip import CLIP
    torch.manual_seed(21)
    args = Namespace(cluster_inter=1, cluster_algo='kmediods++', max_frames=12,
                     target_frames_blocks=[12] * 6 + [4] * 6, cluster_num_blocks=[196] * 6 + [100] * 6,
                     cluster_distance='euclidean', cluster_threshold=1e-6, cluster_iter_limit=100,
                     minkowski_norm_p=2.0, pretrained_clip_name='ViT-B/16', aggregation=None, pre_norm=False)
    model = CLIP(512, 224, 12, 768, 16, 77, 49408, 512, 8, 12, video_frames=12, args=args)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.half().float())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert model.visual.transformer.resblocks[6].tokencluster_inter.split_size == 4
    model = model.to("cuda:0").eval()
    video = torch.randn(12, 3, 224, 224)
    feat, _ = model.visual.encode(video.to("cuda:0"), 12, want_medoids=True)
    med = model.visual.last_medoids.cpu()
    assert feat.shape == (4, 512) and med.shape == (4, 100) and bool((med[:, 1:] > med[:, :-1]).all())
    with torch.no_grad():
        ref = clo.visual_forward(sd, video, 12, cluster_plan={6: (4, 100)}, forced_medoids={6: med})
    n = lambda x: x / x.norm(dim=-1, keepdim=True)
    d = float((n(feat.cpu()) - n(ref)).abs().max())
    print(f"[ViT-B/16 cfg5-shaped] max|delta| normalised embedding = {d:.2e}")
    assert d <= 1e-3
