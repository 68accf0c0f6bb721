"""CLIP / similarity fixtures captured from the imported reference (dev container only; see
gen_golden.py).  Small random-weight models at reduced size (width 128 = 2 heads of 64), full
tensors stored; weights are stored as the reference model's own state_dict (fp32).
"""
import os
import sys
import types
import warnings
from argparse import Namespace

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = "/root/reference"


def _import_reference():
    """`import modules` pulls boto3 / ftfy (weight download + tokenizer): stub them, nothing of
    theirs is on the forward path."""
    for name in ("boto3", "botocore", "botocore.exceptions", "ftfy"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["botocore.exceptions"].ClientError = Exception
    sys.modules["ftfy"].fix_text = lambda s: s
    sys.dont_write_bytecode = True
    # the repository root first and /root/reference/modules not at all while importing: the cluster generators put that
    # directory on sys.path (they import `cluster.*` top-level), where `utils` would resolve to modules/utils.py instead of
    # the utils/ package (`python oracle/gen_golden.py all` died here)
    mods_dir = os.path.join(REF, "modules")
    saved = list(sys.path)
    sys.path[:] = [REF] + [p_ for p_ in sys.path if p_ not in (REF, mods_dir)]
    if "utils" in sys.modules and not hasattr(sys.modules["utils"], "__path__"):
        del sys.modules["utils"]
    try:
        import modules.clip as rclip
        import modules.clip4clip as rc4c
        import modules.cluster.cluster as rcc
        import utils.metrics as rmetrics
    finally:
        sys.path[:] = saved
    return rclip, rc4c, rcc, rmetrics


def ref_args(max_frames, frames_blocks, tokens_blocks, **kw):
    a = Namespace(cluster_inter=1, deep_cluster=0, cluster_algo='kmediods++', max_frames=max_frames,
                  target_frames_blocks=frames_blocks, cluster_num_blocks=tokens_blocks,
                  cluster_distance='euclidean', cluster_threshold=1e-6, cluster_iter_limit=100,
                  minkowski_norm_p=2.0, spectral_sigma=2.0, spectral_graph='HeatKernel', spectral_knn_k=1,
                  spectral_spg=False, aggregation=None, pretrained_clip_name='ViT-B/32', cluster_embedding=0,
                  cluster_frame_embedding=0, save_feature_path=None, svd_correct_sign=1, pre_norm=False,
                  cluser_embed_from_clip=0, cluster_inter_dim=128)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def gen_clip():
    rclip, rc4c, rcc, rmetrics = _import_reference()
    out = {}
    torch.manual_seed(1234)

    # ---------------------------------------------------------------- V1-V3 / T1 small model
    E, RES, P, VW, VL = 64, 64, 16, 128, 3
    CTX, VOCAB, TW, TH, TL = 16, 200, 128, 2, 2
    B, T = 2, 4
    args = ref_args(T, [4, 2, 2], [16, 6, 6])            # block 2 (index 1): 4 frames -> 2 segments, 32 -> 6 tokens
    model = rclip.CLIP(E, RES, VL, VW, P, CTX, VOCAB, TW, TH, TL, linear_patch='2d', video_frames=T, args=args).float().eval()
    with torch.no_grad():                                  # make LN / bias terms non-trivial
        for n_, p_ in model.named_parameters():
            if n_.endswith("bias") or "ln_" in n_:
                p_.add_(0.05 * torch.randn_like(p_))
            p_.copy_(p_.half().float())                    # fp16-representable weights (as convert_weights yields)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        out["sd/" + k] = v.numpy().astype(np.float16) if v.is_floating_point() else v.numpy()
    out["cfg"] = np.array([E, RES, P, VW, VL, CTX, VOCAB, TW, TH, TL, B, T], dtype=np.int64)

    video = torch.randn(B * T, 3, RES, RES)
    out["video"] = video.numpy()

    captured = {}
    orig = rcc.batch_fast_kmedoids_with_split

    def spy(*a, **k):
        r = orig(*a, **k)
        captured["medoids"] = r[1].clone()
        return r
    rcc.batch_fast_kmedoids_with_split = spy
    try:
        with torch.no_grad():
            feat, closs = model.encode_image(video, video_frame=T)
            hidden, _ = model.visual(video, video_frame=T)
    finally:
        rcc.batch_fast_kmedoids_with_split = orig
    out["v_feat"], out["v_hidden"], out["v_medoids"] = feat.numpy(), hidden.numpy(), captured["medoids"].numpy()

    # same weights, no clustering (plain CLIP4Clip baseline path)
    model_nc = rclip.CLIP(E, RES, VL, VW, P, CTX, VOCAB, TW, TH, TL, linear_patch='2d', video_frames=T,
                          args=ref_args(T, [4, 4, 4], [16, 16, 16], cluster_inter=0)).float().eval()
    model_nc.load_state_dict(sd)
    with torch.no_grad():
        feat_nc, _ = model_nc.encode_image(video, video_frame=T)
    out["v_feat_nocluster"] = feat_nc.numpy()

    ids = torch.zeros(3, CTX, dtype=torch.long)
    for b, ln in enumerate((5, 9, 16)):
        ids[b, 0] = VOCAB - 2
        ids[b, 1:ln - 1] = torch.randint(1, VOCAB - 2, (ln - 2,))
        ids[b, ln - 1] = VOCAB - 1                         # EOT = largest id
    with torch.no_grad():
        tfeat = model.encode_text(ids)
    out["t_ids"], out["t_feat"] = ids.numpy(), tfeat.numpy()

    # ---------------------------------------------------------------- S1 / S2: mask + meanP similarity
    fake = types.SimpleNamespace(sim_header="meanP", training=False, pre_visual_pooling=0,
                                 clip=types.SimpleNamespace(logit_scale=torch.tensor(2.5)),
                                 cluster_algo='kmediods++', f_frame_duration=4, final_frames=3)
    fake._mean_pooling_for_similarity_visual = types.MethodType(
        rc4c.CLIP4Clip._mean_pooling_for_similarity_visual, fake)
    seq = torch.randn(5, 1, 64) * 3
    vis = torch.randn(7, 3, 64) * 2
    vmask12 = torch.ones(7, 12, dtype=torch.long)
    vmask12[1, 8:] = 0
    vmask12[2, 3:] = 0
    vmask12[3, :] = 0                                      # fully masked clip: denominator 0 -> 1
    vmask3 = rc4c.CLIP4Clip.get_video_mask_after_cluster(fake, vmask12)
    amask = torch.ones(5, 16, dtype=torch.long)
    with torch.no_grad():
        logits = rc4c.CLIP4Clip._loose_similarity(fake, seq, vis, amask, vmask3)
    out["s_seq"], out["s_vis"], out["s_mask12"], out["s_mask3"] = seq.numpy(), vis.numpy(), vmask12.numpy(), vmask3.numpy()
    out["s_logits"], out["s_logit_scale"] = logits.numpy(), np.float32(2.5)

    # ---------------------------------------------------------------- N1: retrieval metrics
    sim = np.random.default_rng(5).standard_normal((40, 40)).astype(np.float32)
    sim[np.arange(40), np.arange(40)] += 1.5
    sim[3, 7] = sim[3, 3]                                  # a tie with the diagonal
    m1 = rmetrics.compute_metrics(sim)
    m2 = rmetrics.compute_metrics(sim.T)
    out["n1_sim"] = sim
    out["n1_t2v"] = np.array([m1["R1"], m1["R5"], m1["R10"], m1["MR"], m1["MeanR"]], dtype=np.float64)
    out["n1_v2t"] = np.array([m2["R1"], m2["R5"], m2["R10"], m2["MR"], m2["MeanR"]], dtype=np.float64)
    out["n1_t2v_cols"] = np.array(m1["cols"], dtype=np.int64)

    np.savez_compressed(os.path.join(GOLD, "clip_golden.npz"), **out)
    print("wrote clip_golden.npz", os.path.getsize(os.path.join(GOLD, "clip_golden.npz")), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    gen_clip()
