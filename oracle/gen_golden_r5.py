"""Round-5 fixtures captured from the IMPORTED reference (dev container only; see gen_golden.py for the rules):

  mrv_*    mean_residual INSIDE the visual tower (modules/cluster/cluster.py:228-235, modules/clip.py:239-242): the small CLIP
           of clip_golden.npz (its state dict and video) with a cluster block that halves the frames and keeps the token count
           (the reference's assert, cluster.py:229), mean_residual switched on at the module (get_cluster_inter never passes it,
           cluster.py:15-60 - the attribute is the only way in) -> image features, hidden state, the block's medoid ids

  p1n_*    batch_fast_kmedoids_with_split above N = 1,023 (the summation tree of ATen's row sum folds its accumulators more
           than once there: 16 passes of 32 terms per fold), integer lattices - parity level P1 (indices bit-exact):
             p1n_b16_64f   N = 1,568 = 8 frames x 196 tokens (ViT-B/16, 64 -> 8 frames), K = 49, split_size 4, two chunks
             p1n_ragged    N = 1,103 (a 7-scalar tail, left-over vectors), K = 100, one chunk

    python oracle/gen_golden_r5.py   ->  tests/golden/r5_golden.npz
"""
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, HERE)
from recipes import lattice  # noqa: E402

# name: (seed, P, N, W, K, split, iter_limit)
P1_R5 = {"p1n_b16_64f": (151, 6, 1568, 64, 49, 4, 100), "p1n_ragged": (152, 2, 1103, 32, 100, 16, 100)}


def gen_mean_residual_tower(out):
    from gen_golden_clip import _import_reference, ref_args
    rclip, _rc4c, rcc, _ = _import_reference()
    g = np.load(os.path.join(GOLD, "clip_golden.npz"))
    E, RES, P, VW, VL, CTX, VOCAB, TW, TH, TL, B, T = (int(v) for v in g["cfg"])
    sd = {k[3:]: torch.from_numpy(g[k].astype(np.float32) if g[k].dtype == np.float16 else g[k])
          for k in g.files if k.startswith("sd/")}
    n = (RES // P) ** 2
    args = ref_args(T, [T, T // 2, T // 2], [n, n, n])     # block 2: 4 frames -> 2 segments, 2 x 16 tokens -> 16
    model = rclip.CLIP(E, RES, VL, VW, P, CTX, VOCAB, TW, TH, TL, linear_patch='2d', video_frames=T, args=args).float().eval()
    model.load_state_dict(sd)
    tc = model.visual.transformer.resblocks[1].tokencluster_inter
    assert tc is not None and model.visual.transformer.resblocks[0].tokencluster_inter is None
    assert model.visual.transformer.resblocks[2].tokencluster_inter is None
    tc.mean_residual = True
    video = torch.from_numpy(g["video"])
    captured = {}
    orig = rcc.batch_fast_kmedoids_with_split

    def spy(*a, **k):
        r = orig(*a, **k)
        captured["medoids"] = r[1].clone()
        return r
    rcc.batch_fast_kmedoids_with_split = spy
    try:
        with torch.no_grad():
            feat, _ = model.encode_image(video, video_frame=T)
            hidden, _ = model.visual(video, video_frame=T)
            tc.mean_residual = False
            feat_plain, _ = model.encode_image(video, video_frame=T)
    finally:
        rcc.batch_fast_kmedoids_with_split = orig
    out["mrv_feat"], out["mrv_hidden"], out["mrv_medoids"] = feat.numpy(), hidden.numpy(), captured["medoids"].numpy()
    out["mrv_feat_plain"] = feat_plain.numpy()            # the same plan without mean_residual (the two must differ)
    out["mrv_plan"] = np.array([T, T // 2, n], dtype=np.int64)
    print("mrv", tuple(feat.shape), tuple(hidden.shape), tuple(captured["medoids"].shape),
          float((feat - feat_plain).abs().max()), flush=True)


def gen_big_n(out):
    sys.path.insert(0, os.path.join("/root/reference", "modules"))
    import cluster.fast_kmeans as fk
    for tag, (seed, P, N, W, K, split, iters) in P1_R5.items():
        X = torch.from_numpy(lattice(seed, (P, N, W)))
        a, m = fk.batch_fast_kmedoids_with_split(X, K, distance="euclidean", threshold=1e-6, iter_limit=iters,
                                                 id_sort=True, norm_p=2.0, split_size=split, pre_norm=False)
        out[f"{tag}_cfg"] = np.array([seed, P, N, W, K, split, iters], dtype=np.int64)
        out[f"{tag}_assign"], out[f"{tag}_medoids"] = a.numpy().astype(np.int16), m.numpy().astype(np.int16)
        print(tag, "done", tuple(m.shape), flush=True)


def main():
    out = {}
    gen_mean_residual_tower(out)
    gen_big_n(out)
    path = os.path.join(GOLD, "r5_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
