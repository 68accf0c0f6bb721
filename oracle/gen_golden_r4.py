"""Round-4 fixtures captured from the IMPORTED reference (dev container only; see gen_golden.py for the rules):

  p1m_b16  batch_fast_kmedoids_with_split on an integer lattice at the shipped ActivityNet / LSMDC ViT-B/16 shape
           (scripts/activitynet.sh:104-122: 4 frames of 196 tokens per segment -> N = 784, K = 160, W = 768, split_size 4,
           cluster.py:56), two chunks with a ragged second one - parity level P1 (indices bit-exact)

  bg_*     ResidualAttentionBlock forward + torch.autograd gradients (input and the 12 parameter tensors) on the
           reference module, visual (no mask) and text (causal mask) flavours - the fixture of the block backward (N4)

    python oracle/gen_golden_r4.py   ->  tests/golden/r4_golden.npz
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
P1_R4 = {"p1m_b16": (145, 6, 784, 768, 160, 4, 100)}
from recipes import BLOCK_GRAD_CASES, block_grad_inputs  # noqa: E402


def gen_block_grads(out):
    """bg_*: the reference's ResidualAttentionBlock (modules/clip.py:196-253) forward and torch.autograd's gradients of
    sum(z * dz) with respect to the input and every parameter (N4: the block backward's fixture)."""
    sys.path.insert(0, HERE)
    from gen_golden_clip import _import_reference
    rclip = _import_reference()[0]
    for tag, cfg in BLOCK_GRAD_CASES.items():
        x, dz, sd = block_grad_inputs(cfg)
        mask = None
        if cfg["causal"]:
            mask = torch.empty(cfg["L"], cfg["L"]).fill_(float("-inf")).triu_(1)      # clip.py:448-454
        blk = rclip.ResidualAttentionBlock(cfg["W"], cfg["heads"], attn_mask=mask, block_id=1, args=None).float()
        blk.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        xt = torch.from_numpy(x).requires_grad_(True)
        z = blk((xt, -1, 0.0))[0]
        (z * torch.from_numpy(dz)).sum().backward()
        out[f"{tag}_z"], out[f"{tag}_dx"] = z.detach().numpy(), xt.grad.numpy()
        for k, p_ in blk.named_parameters():
            out[f"{tag}_grad/{k}"] = p_.grad.numpy()
        print(tag, "done", tuple(z.shape), flush=True)


def main():
    sys.path.insert(0, os.path.join("/root/reference", "modules"))
    import cluster.fast_kmeans as fk
    out = {}
    for tag, (seed, P, N, W, K, split, iters) in P1_R4.items():
        X = torch.from_numpy(lattice(seed, (P, N, W)))
        a, m = fk.batch_fast_kmedoids_with_split(X, K, distance="euclidean", threshold=1e-6, iter_limit=iters,
                                                 id_sort=True, norm_p=2.0, split_size=split, pre_norm=False)
        out[f"{tag}_cfg"] = np.array([seed, P, N, W, K, split, iters], dtype=np.int64)
        out[f"{tag}_assign"], out[f"{tag}_medoids"] = a.numpy().astype(np.int16), m.numpy().astype(np.int16)
        print(tag, "done", tuple(m.shape), flush=True)
    gen_block_grads(out)
    path = os.path.join(GOLD, "r4_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
