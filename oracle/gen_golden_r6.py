"""Round-6 fixtures captured from the IMPORTED reference (dev container only; see gen_golden.py for the rules):

  lt_*   batch_fast_kmedoids_with_split with a LOOSE `threshold` (modules/cluster/fast_kmeans.py:85-88: the loop of a split chunk
         ends as soon as the chunk MEAN of sum_k |X[m_k] - X[m_k_prev]|_2 falls below it - earlier than the fixed point, and at a
         step that depends on every problem of the chunk).  Dyadic inputs (sums of four integers / 64: exact in the fp16 split of
         the Gram kernel - parity level P1, indices bit-exact; pre_norm: tokens of norm exactly 32); the thresholds are combinations of values of the reference's own
         center_shift sequences (recorded here by re-running its loop body with its own functions, oracle/recipes.py
         LOOSE_THRESHOLD_CASES), so a case stops one chunk in mid-course while another runs on:

           lt_two_chunks      P = 6, N = 196, W = 64,  K = 49, split 4 (chunks of 4 + 2), id_sort
           lt_unsorted        the same problem, id_sort = False (the assignment of the last executed iteration is returned)
           lt_wide            P = 3, N = 100, W = 768, K = 10, split 2 (24 passes of 32 terms per row sum of the shift: two runs)
           lt_prenorm         P = 4, N = 98,  W = 96,  K = 25, split 2, pre_norm on tokens of norm exactly 32 (the shift is taken on
                              the normalised X); chunk 1 stops at step 2 although steps 3 and 4 would have moved further
           lt_tiny            P = 4, N = 7,   W = 8,   K = 3,  split 4 (ATen's scalar path for the sum over K; N < 8): one step
           lt_iter_limit      P = 4, N = 196, W = 64,  K = 49, split 4, iter_limit 2 with a threshold those two steps do not reach

  p1w_*  batch_fast_kmedoids_with_split above N = 4,095 (the limit of round 5; the selection kernel's member lists now carry 13-bit
         token ids and up to 128 mask words per cluster: N <= 8,191 = the last length at which ATen's row sum stays within two
         accumulator levels), integer lattices - parity level P1:
           p1w_4500   P = 2, N = 4,500, W = 16, K = 6, split_size 1 (two chunks)
           p1w_8191   P = 1, N = 8,191, W = 8,  K = 4 (255 passes of 32 terms: 15 full runs of 16 passes + a partial one)

    python oracle/gen_golden_r6.py   ->  tests/golden/r6_golden.npz
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
from recipes import LOOSE_THRESHOLD_CASES, P1_WIDE_CASES, lattice, loose_threshold_inputs  # noqa: E402


def shift_sequences(fk, cu, X, K, distance, norm_p, split, pre_norm, iter_limit):
    """center_shift of every (chunk, step) of the reference's loop, with the reference's own functions (no early stop)."""
    X = X.float()
    if pre_norm:
        X = X / (X.norm(dim=-1, keepdim=True) + 1e-6)
    out = []
    for xc in torch.split(X, split, dim=0):
        D = cu.pairwise_distance(xc, xc, metric=distance, all_negative=True, self_nearest=True, p=norm_p)
        big = D.unsqueeze(1).repeat(1, K, 1, 1)
        med = cu.KKZ_init(xc, D, K, batch=True)
        bi = torch.arange(xc.shape[0]).unsqueeze(1)
        kid = torch.arange(K).reshape(1, K, 1).repeat(xc.shape[0], 1, 1)
        seq = []
        for _ in range(iter_limit):
            pre = med
            a = torch.min(D[bi, med, :], dim=1)[1]
            mask = a.unsqueeze(1).repeat(1, K, 1) == kid
            med = torch.argmin(torch.sum(big * mask.unsqueeze(-1) * mask.unsqueeze(-2), dim=-1), dim=-1)
            seq.append(float(torch.sum((xc[bi, med, :] - xc[bi, pre, :]) ** 2, dim=-1).sqrt().sum(dim=-1).mean()))
        out.append(seq)
    return out


def main():
    sys.path.insert(0, os.path.join("/root/reference", "modules"))
    import cluster.fast_kmeans as fk
    import cluster.cluster_utils as cu
    out = {}
    for tag, (seed, P, N, W, K, split, iters, distance, pre_norm, id_sort, pick, _recipe) in LOOSE_THRESHOLD_CASES.items():
        X = torch.from_numpy(loose_threshold_inputs(tag))
        seqs = shift_sequences(fk, cu, X, K, distance, 2.0, split, pre_norm, max(iters, 8))
        ca, sa, wa, cb, sb, wb = pick
        thr = np.float32(wa * seqs[ca][sa] + wb * seqs[cb][sb])
        assert thr > 1e-5 and all(abs(np.float32(v) - thr) > 1e-4 * thr for s_ in seqs for v in s_), (tag, thr, seqs)
        a, m = fk.batch_fast_kmedoids_with_split(X, K, distance=distance, threshold=float(thr), iter_limit=iters,
                                                 id_sort=id_sort, norm_p=2.0, split_size=split, pre_norm=pre_norm)
        a_fix, m_fix = fk.batch_fast_kmedoids_with_split(X, K, distance=distance, threshold=1e-6, iter_limit=100,
                                                         id_sort=id_sort, norm_p=2.0, split_size=split, pre_norm=pre_norm)
        stops = [next((i + 1 for i, v in enumerate(s[:iters]) if np.float32(v) < thr), iters) for s in seqs]
        out[f"{tag}_threshold"] = np.array([thr], dtype=np.float32)
        out[f"{tag}_assign"], out[f"{tag}_medoids"] = a.numpy().astype(np.int16), m.numpy().astype(np.int16)
        out[f"{tag}_steps"] = np.array(stops, dtype=np.int32)                       # iterations executed per chunk
        out[f"{tag}_differs_from_fixed_point"] = np.array([int(not torch.equal(m, m_fix))], dtype=np.int8)
        print(tag, "threshold %.6g" % thr, "steps per chunk", stops, "differs from the fixed point:", not torch.equal(m, m_fix),
              "| shifts chunk 0:", ["%.4g" % v for v in seqs[0][:6]], flush=True)
    for tag, (seed, P, N, W, K, split, iters) in P1_WIDE_CASES.items():
        X = torch.from_numpy(lattice(seed, (P, N, W)))
        a, m = fk.batch_fast_kmedoids_with_split(X, K, distance="euclidean", threshold=1e-6, iter_limit=iters,
                                                 id_sort=True, norm_p=2.0, split_size=split, pre_norm=False)
        out[f"{tag}_assign"], out[f"{tag}_medoids"] = a.numpy().astype(np.int16), m.numpy().astype(np.int16)
        print(tag, tuple(a.shape), m.numpy().tolist(), flush=True)
    np.savez_compressed(os.path.join(GOLD, "r6_golden.npz"), **out)
    print("wrote", os.path.join(GOLD, "r6_golden.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
