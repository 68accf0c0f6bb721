"""Pin oracle/clip_oracle.py against the fixtures captured from the imported reference
(tests/golden/clip_golden.npz).  CPU only.  Tolerance: 2e-5 absolute on fp32 tensors - same math,
different op grouping (e.g. explicit softmax(qk^T)v vs nn.MultiheadAttention's fused path)."""
import os

import numpy as np
import pytest
import torch

from oracle import clip_oracle as clo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_golden.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN)


def state_dict(g):
    return {k[3:]: torch.from_numpy(g[k].astype(np.float32) if g[k].dtype == np.float16 else g[k])
            for k in g.files if k.startswith("sd/")}


PLAN = {1: (2, 6)}


def test_visual_forward_with_cluster_matches_reference(g):
    sd = state_dict(g)
    video = torch.from_numpy(g["video"])
    T = int(g["cfg"][11])
    feat, hidden = clo.visual_forward(sd, video, T, cluster_plan=PLAN, return_hidden=True)
    np.testing.assert_allclose(hidden.numpy(), g["v_hidden"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(feat.numpy(), g["v_feat"], rtol=0, atol=2e-5)
    # forcing the reference's medoids gives the same result (the path the GPU tests use)
    forced = {1: torch.from_numpy(g["v_medoids"])}
    feat2 = clo.visual_forward(sd, video, T, cluster_plan=PLAN, forced_medoids=forced)
    np.testing.assert_allclose(feat2.numpy(), g["v_feat"], rtol=0, atol=2e-5)


def test_visual_forward_without_cluster_matches_reference(g):
    feat = clo.visual_forward(state_dict(g), torch.from_numpy(g["video"]), int(g["cfg"][11]))
    np.testing.assert_allclose(feat.numpy(), g["v_feat_nocluster"], rtol=0, atol=2e-5)


def test_text_forward_matches_reference(g):
    feat = clo.text_forward(state_dict(g), torch.from_numpy(g["t_ids"]))
    np.testing.assert_allclose(feat.numpy(), g["t_feat"], rtol=0, atol=2e-5)


def test_mask_after_cluster_and_loose_similarity(g):
    m3 = clo.video_mask_after_cluster(torch.from_numpy(g["s_mask12"]), 12, 3)
    assert np.array_equal(m3.numpy(), g["s_mask3"])
    logits = clo.loose_similarity(torch.from_numpy(g["s_seq"]), torch.from_numpy(g["s_vis"]), m3, float(g["s_logit_scale"]))
    np.testing.assert_allclose(logits.numpy(), g["s_logits"], rtol=0, atol=2e-5)
    # blocked construction (main.py:502-534) == one shot
    seqs = list(torch.from_numpy(g["s_seq"]).split(2)); vis = list(torch.from_numpy(g["s_vis"]).split(3)); ms = list(m3.split(3))
    blocked = clo.similarity_matrix_blocked(seqs, vis, ms, float(g["s_logit_scale"]))
    np.testing.assert_allclose(blocked.numpy(), g["s_logits"], rtol=0, atol=2e-5)


def test_n3_loader_normalize_is_three_ieee_ops():
    """oracle.loader_normalize (the reference loader's u8 -> float, transforms.py:19-34,166) equals the plain IEEE
    fp32 sequence u/255, -mean, /std bit for bit - the sequence the HIP patch gather performs on uint8 input."""
    rng = np.random.default_rng(5)
    u = rng.integers(0, 256, size=(3, 16, 24, 3), dtype=np.uint8)
    u[0, 0, :8, 0] = [0, 1, 2, 127, 128, 254, 255, 77]
    got = clo.loader_normalize(torch.from_numpy(u), channels_last=True).numpy()
    mean = np.asarray(clo.PIXEL_MEAN, dtype=np.float32)[None, :, None, None]
    std = np.asarray(clo.PIXEL_STD, dtype=np.float32)[None, :, None, None]
    x = u.transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255.0)
    want = ((x - mean).astype(np.float32) / std).astype(np.float32)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    got_chw = clo.loader_normalize(torch.from_numpy(np.ascontiguousarray(u.transpose(0, 3, 1, 2)))).numpy()
    assert np.array_equal(got_chw, want)


def test_visual_forward_with_mean_residual_matches_reference(g):
    """mean_residual inside the tower (cluster.py:228-235, clip.py:239-242; fixture oracle/gen_golden_r5.py): the oracle's
    restatement against the reference's features, hidden state and medoids - and against the reference WITHOUT the flag."""
    g5 = np.load(os.path.join(os.path.dirname(GOLDEN), "r5_golden.npz"))
    sd = state_dict(g)
    video = torch.from_numpy(g["video"])
    T, T_new, n = [int(v) for v in g5["mrv_plan"]]
    plan = {1: (T_new, n)}
    feat, hidden = clo.visual_forward(sd, video, T, cluster_plan=plan, return_hidden=True, mean_residual=(1,))
    np.testing.assert_allclose(hidden.numpy(), g5["mrv_hidden"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(feat.numpy(), g5["mrv_feat"], rtol=0, atol=2e-5)
    forced = {1: torch.from_numpy(g5["mrv_medoids"].astype(np.int64))}
    feat2 = clo.visual_forward(sd, video, T, cluster_plan=plan, forced_medoids=forced, mean_residual=(1,))
    np.testing.assert_allclose(feat2.numpy(), g5["mrv_feat"], rtol=0, atol=2e-5)
    plain = clo.visual_forward(sd, video, T, cluster_plan=plan)
    np.testing.assert_allclose(plain.numpy(), g5["mrv_feat_plain"], rtol=0, atol=2e-5)
