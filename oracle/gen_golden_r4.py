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


def gen_train_grads(out):
    """tr_*: one training step's loss and torch.autograd's gradient of every parameter of the reference CLIP (the small model
    of clip_golden.npz: its state dict, video and captions), through encode_image (k-medoids block included), encode_text,
    the meanP similarity of CLIP4Clip._loose_similarity and the symmetric CrossEn of the training branch
    (modules/clip4clip.py:245-262; evaluated with training=False so that the all_gather of :351-355 - the identity on one
    rank - is skipped, the arithmetic is the training branch's)."""
    import types
    sys.path.insert(0, HERE)
    from gen_golden_clip import _import_reference, ref_args
    rclip, rc4c, _, _ = _import_reference()
    import modules.losses as rlosses
    g = np.load(os.path.join(GOLD, "clip_golden.npz"))
    E, RES, P, VW, VL, CTX, VOCAB, TW, TH, TL, B, T = (int(v) for v in g["cfg"])
    model = rclip.CLIP(E, RES, VL, VW, P, CTX, VOCAB, TW, TH, TL, linear_patch='2d', video_frames=T,
                       args=ref_args(T, [4, 2, 2], [16, 6, 6])).float().train()
    model.load_state_dict({k[3:]: torch.from_numpy(g[k].astype(np.float32) if g[k].dtype == np.float16 else g[k])
                           for k in g.files if k.startswith("sd/")})
    video, ids = torch.from_numpy(g["video"]), torch.from_numpy(g["t_ids"])[:B]
    vfeat, closs = model.encode_image(video, video_frame=T)
    tfeat = model.encode_text(ids)
    fake = types.SimpleNamespace(sim_header="meanP", training=False, pre_visual_pooling=0,
                                 clip=types.SimpleNamespace(logit_scale=model.logit_scale))
    fake._mean_pooling_for_similarity_visual = types.MethodType(rc4c.CLIP4Clip._mean_pooling_for_similarity_visual, fake)
    vis, seq = vfeat.view(B, -1, E), tfeat.view(B, 1, E)
    vmask = torch.ones(B, vis.shape[1], dtype=torch.long)
    sim = rc4c.CLIP4Clip._loose_similarity(fake, seq, vis, torch.ones(B, CTX, dtype=torch.long), vmask)
    ce = rlosses.CrossEn()
    loss = (ce(sim) + ce(sim.T)) / 2
    loss.backward()
    out["tr_loss"], out["tr_sim"] = np.float32(loss.item()), sim.detach().numpy()
    out["tr_vfeat"], out["tr_tfeat"] = vfeat.detach().numpy(), tfeat.detach().numpy()
    n = 0
    for k, p_ in model.named_parameters():
        if p_.grad is not None:
            out["tr_grad/" + k] = p_.grad.numpy()
            n += 1
    print("train grads:", n, "tensors, loss", float(loss), flush=True)
    # three BertAdam steps of the reference optimizer (utils/optimization.py) on two tensors with given gradients
    from utils.optimization import BertAdam
    rng = np.random.default_rng(9)
    p0 = [torch.nn.Parameter(torch.from_numpy(rng.standard_normal(s_).astype(np.float32))) for s_ in ((37, 5), (300,))]
    opt = BertAdam([{'params': [p0[0]], 'weight_decay': 0.2}, {'params': [p0[1]], 'weight_decay': 0.0}], lr=1e-2, warmup=0.1,
                   t_total=20, schedule='warmup_linear', b1=0.9, b2=0.98, e=1e-6, max_grad_norm=1.0)
    out["ba_p0"], out["ba_p1"] = p0[0].detach().numpy().copy(), p0[1].detach().numpy().copy()
    for it in range(3):
        gr = [rng.standard_normal(tuple(p_.shape)).astype(np.float32) * (3.0 if it == 1 else 0.05) for p_ in p0]
        for p_, g_ in zip(p0, gr):
            p_.grad = torch.from_numpy(g_.copy())
        out[f"ba_g{it}_0"], out[f"ba_g{it}_1"] = gr
        opt.step()
        out[f"ba_after{it}_0"], out[f"ba_after{it}_1"] = p0[0].detach().numpy().copy(), p0[1].detach().numpy().copy()
    out["ba_m_0"], out["ba_v_0"] = opt.state[p0[0]]['next_m'].numpy().copy(), opt.state[p0[0]]['next_v'].numpy().copy()


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
    gen_train_grads(out)
    path = os.path.join(GOLD, "r4_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
