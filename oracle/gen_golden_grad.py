"""Gradient fixtures captured from the IMPORTED reference (dev container only; see gen_golden.py for the rules):
the reference's own TokenClusterInter run under torch.autograd for the cases of recipes.GRAD_CASES - the shipped
kmediods++ / aggregation=None branch, cluster means, cluster_embedding + adaptive_cls, pooling, eval-mode
sparse_sampling - with the upstream gradient recipes.grad_output(cfg).  Stored: d loss / d x (fp32, sparse for the
gathers), the parameter gradients, and the selection the reference made (so that the backward op can be checked "given
the identical selection" where the selection itself is not an exact target).   [SURVEY §8f N4: autograd for gather/CLS-mean]

    python oracle/gen_golden_grad.py     ->  tests/golden/cluster_grad_golden.npz
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
REF = "/root/reference"
sys.path.insert(0, HERE)
from recipes import GRAD_CASES, variant_input, grad_output  # noqa: E402


def main():
    sys.path.insert(0, os.path.join(REF, "modules"))
    import cluster.cluster as cc
    t = torch.from_numpy
    out = {}
    for tag, cfg in GRAD_CASES.items():
        x, embed, mult = variant_input(cfg)
        mod = cc.TokenClusterInter(algorithm=cfg["algorithm"], block_id=7, before_cluster_num=cfg["n"],
                                   cluster_num=cfg["K"], before_block_frames=cfg["T"], after_block_frames=cfg["T_new"],
                                   original_frame=cfg["T"], distance="euclidean", threshold=1e-6, iter_limit=100,
                                   id_sort=True, aggregation=cfg["aggregation"], split_size=16, norm_p=2.0,
                                   cluster_embedding=bool(cfg.get("embed")), adaptive_cls=bool(cfg.get("adaptive")),
                                   transformer_width=cfg["W"])
        mod.eval()                                    # (sparse_sampling draws random ids in training mode)
        with torch.no_grad():
            if embed is not None:
                mod.cluster_embed.copy_(t(embed))
            if mult is not None:
                mod.cls_multiplier.copy_(t(mult).reshape(1, -1, 1, 1))
        captured = {}
        orig = cc.batch_fast_kmedoids_with_split

        def spy(*a, **k):
            r = orig(*a, **k)
            captured["assign"], captured["medoids"] = r[0].clone(), r[1].clone()
            return r
        cc.batch_fast_kmedoids_with_split = spy
        xt = t(x).clone().requires_grad_(True)
        try:
            y, _ = mod(xt)
        finally:
            cc.batch_fast_kmedoids_with_split = orig
        G = t(grad_output(cfg))
        assert tuple(G.shape) == tuple(y.shape), (tag, G.shape, y.shape)
        y.backward(G)
        out[f"{tag}_gx"] = xt.grad.numpy()
        if embed is not None:
            out[f"{tag}_gembed"] = mod.cluster_embed.grad.numpy()
        if mult is not None:
            out[f"{tag}_gmult"] = mod.cls_multiplier.grad.reshape(-1).numpy()
        if captured:
            out[f"{tag}_assign"] = captured["assign"].numpy().astype(np.int16)
            out[f"{tag}_medoids"] = captured["medoids"].numpy().astype(np.int16)
        print(tag, "done", tuple(y.shape), float(xt.grad.abs().sum()), flush=True)
    path = os.path.join(GOLD, "cluster_grad_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
