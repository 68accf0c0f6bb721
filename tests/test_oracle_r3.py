"""CPU tests (no GPU): the oracle against the round-3 fixtures generated from the imported reference
(tests/golden/r3_golden.npz, oracle/gen_golden_r3.py): multi-chunk k-medoids at the per-GPU problem counts, the
similarity matrix of main._run_on_single_gpu on stored features, main.eval_epoch's matrix from the towers."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import clip_oracle as clo
from oracle import cluster_oracle as co
from oracle.recipes import EVAL_CASES, eval_case_batches, lattice, loss_grad_case, s3_case

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g3():
    return np.load(os.path.join(HERE, "golden", "r3_golden.npz"))


@pytest.fixture(scope="module")
def small():
    g2 = np.load(os.path.join(HERE, "golden", "r2_golden.npz"))
    sd = {k[6:]: torch.from_numpy(g2[k].astype(np.float32) if g2[k].dtype == np.float16 else g2[k])
          for k in g2.files if k.startswith("s1_sd/")}
    return sd, g2["s1_cfg"]


@pytest.mark.parametrize("tag,chunks", [("p1m_cfg3", (0, 15)), ("p1m_cfg4", (3,)), ("p1m_cfg5", (1, 3)), ("p1m_ragged", (2,))])
def test_p1_multi_chunk_oracle(g3, tag, chunks):
    """The literal restatement reproduces the reference's indices chunk by chunk (a chunk is a self-contained call of
    batch_fast_kmedoids, fast_kmeans.py:24-34); a sample of chunks keeps the CPU suite short."""
    seed, P, N, W, K, split, iters = [int(v) for v in g3[f"{tag}_cfg"]]
    X = torch.from_numpy(lattice(seed, (P, N, W)))
    for c in chunks:
        s, e = c * split, min((c + 1) * split, P)
        a, m = co.literal_batch_kmedoids_with_split(X[s:e], K, "euclidean", 1e-6, iters, True, 2.0, split, False)
        assert np.array_equal(m.numpy(), g3[f"{tag}_medoids"][s:e].astype(np.int64))
        assert np.array_equal(a.numpy(), g3[f"{tag}_assign"][s:e].astype(np.int64))


def test_s3_similarity_matrix_oracle(g3, small):
    """clip_oracle.similarity_matrix_blocked == the reference's main._run_on_single_gpu on the same stored features."""
    sd, cfg = small
    T, T_new = int(cfg[11]), int(cfg[12])
    seq_list, vis_list, list_t, list_v = s3_case(int(cfg[0]), T, T_new)
    masks = [clo.video_mask_after_cluster(m[0].view(-1, T), T, T_new) for m in list_v]
    sim = clo.similarity_matrix_blocked(seq_list, vis_list, masks, float(sd["logit_scale"])).numpy()
    ref = g3["s3_sim"]
    assert np.array_equal(np.isnan(sim), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert float(np.abs(sim[ok] - ref[ok]).max()) <= 1e-5 * math.exp(float(sd["logit_scale"]))


@pytest.mark.parametrize("name", sorted(EVAL_CASES))
def test_eval_epoch_matrix_oracle(g3, small, name):
    """The oracle's towers + similarity on the eval fixtures' inputs == the matrix the reference's eval_epoch formed."""
    sd, cfg = small
    T = int(cfg[11])
    batches, attrs = eval_case_batches(EVAL_CASES[name], cfg)
    ids = torch.cat([b[0] for b in batches]).view(-1, int(cfg[5]))
    video = torch.cat([b[3] for b in batches])
    vmask = torch.cat([b[4] for b in batches]).view(-1, T)
    if attrs:                                                # the clip's video comes with the item of its last sentence
        pick = [c - 1 for c in attrs["cut_off_points"]]
        video, vmask = video[pick], vmask[pick]
    with torch.no_grad():
        seq = clo.text_forward(sd, ids).view(ids.shape[0], 1, -1)
        vis = clo.visual_forward(sd, video.reshape((-1,) + tuple(video.shape[3:])), T).view(video.shape[0], T, -1)
        sim = clo.loose_similarity(seq, vis, vmask, float(sd["logit_scale"])).numpy()
    ref = g3[f"ev_{name}_sim"]
    assert sim.shape == ref.shape
    assert float(np.abs(sim - ref).max()) <= 2e-5 * math.exp(float(sd["logit_scale"]))


@pytest.mark.parametrize("tag,n", [("lg_a", 6), ("lg_b", 33)])
def test_contrastive_loss_gradients_oracle(g3, small, tag, n):
    """The oracle's loss + gradients == the reference module's own (get_similarity_logits in training mode -> CrossEn both
    ways -> autograd), clip4clip.py:245-262."""
    sd, cfg = small
    seq, vis, vmask = loss_grad_case(tag, n, int(cfg[12]), int(cfg[0]))
    loss3, dseq, dvis, dls = clo.contrastive_loss_and_grads(torch.from_numpy(seq), torch.from_numpy(vis), torch.from_numpy(vmask),
                                                            float(g3[f"{tag}_scale"]))
    assert np.allclose(loss3.numpy(), g3[f"{tag}_loss3"], rtol=1e-6, atol=1e-6)
    assert np.allclose(dseq.numpy(), g3[f"{tag}_dseq"], rtol=1e-5, atol=1e-8)
    assert np.allclose(dvis.numpy(), g3[f"{tag}_dvis"], rtol=1e-5, atol=1e-8)
    assert abs(float(dls) - float(g3[f"{tag}_dls"])) <= 1e-5 * max(1.0, abs(float(g3[f"{tag}_dls"])))
