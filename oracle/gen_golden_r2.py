"""Round-2 fixtures captured from the IMPORTED reference (dev container only; see gen_golden.py for the rules):

  s1_*   the reference's own CLIP4Clip (built by CLIP4Clip.from_pretrained from a small random-weight "ViT-B-32.pt"):
         forward() in eval mode without / with pre_visual_pooling -> get_similarity_logits(), masks with zeros and a fully
         masked clip, and the training branch's loss values (world-size-1 gloo group, no_grad)   [SURVEY §8c row S1/S2]
  pn_*   batch_fast_kmedoids_with_split(pre_norm=True) on tokens whose L2 norm is exactly 32, so that the reference's
         X / (|X| + 1e-6) is exact in fp32 and the medoid indices are a bit-exact target            [§8c row C2]

    python oracle/gen_golden_r2.py
"""
import os
import sys
import tempfile
import warnings
from argparse import Namespace

import numpy as np
import torch

warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, HERE)
from gen_golden_clip import _import_reference, ref_args  # noqa: E402
from recipes import norm32_tokens, PRENORM_CASES  # noqa: E402


def gen_s1(out):
    rclip, rc4c, rcc, rmetrics = _import_reference()
    torch.manual_seed(4321)
    E, RES, P, VW, VL = 64, 64, 16, 128, 3
    CTX, VOCAB, TW, TH, TL = 16, 200, 128, 2, 2
    B, T, T_new = 3, 4, 2
    small = rclip.CLIP(E, RES, VL, VW, P, CTX, VOCAB, TW, TH, TL, linear_patch='2d', video_frames=T, args=None).float()
    with torch.no_grad():
        for n_, p_ in small.named_parameters():
            if n_.endswith("bias") or "ln_" in n_:
                p_.add_(0.05 * torch.randn_like(p_))
            p_.copy_(p_.half().float())
        small.logit_scale.fill_(2.0)
    sd_small = {k: v.detach().clone() for k, v in small.state_dict().items()}
    tmp = tempfile.mkdtemp()
    torch.save(sd_small, os.path.join(tmp, "ViT-B-32.pt"))
    for pvp in (0, 1):
        task = ref_args(T, [4, T_new, T_new], [16, 6, 6], pretrained_dir=tmp, loose_type=True, sim_header='meanP',
                        linear_patch='2d', cross_num_hidden_layers=2, temperature_new=1.0, pre_visual_pooling=pvp,
                        max_words=CTX, local_rank=0, freeze_clip=0, time_embedding=0, new_added_modules=[None],
                        camoe_dsl=False)
        model = rc4c.CLIP4Clip.from_pretrained('cross-base', cache_dir=None, state_dict=None, task_config=task)
        model = model.float().eval()
        if pvp == 0:
            sd = {k: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("clip.")}
            for k, v in sd.items():
                out["s1_sd/" + k[5:]] = v.numpy().astype(np.float16) if v.is_floating_point() and v.dim() > 0 else v.numpy()
            out["s1_cfg"] = np.array([E, RES, P, VW, VL, CTX, VOCAB, TW, TH, TL, B, T, T_new], dtype=np.int64)
            g = torch.Generator().manual_seed(77)
            video = torch.randn(B, 1, T, 3, RES, RES, generator=g)
            vmask = torch.ones(B, 1, T, dtype=torch.long)
            vmask[1, 0, 3:] = 0                             # last segment of clip 1 masked
            vmask[2, 0, :] = 0                              # clip 2 fully masked (denominator 0 -> 1)
            ids = torch.zeros(B, 1, CTX, dtype=torch.long)
            for b, ln in enumerate((6, 11, 16)):
                ids[b, 0, 0] = VOCAB - 2
                ids[b, 0, 1:ln - 1] = torch.randint(1, VOCAB - 2, (ln - 2,), generator=g)
                ids[b, 0, ln - 1] = VOCAB - 1
            amask = (ids > 0).long()
            seg = torch.zeros_like(ids)
            out["s1_video"], out["s1_vmask"], out["s1_ids"], out["s1_amask"] = (video.numpy(), vmask.numpy(), ids.numpy(),
                                                                               amask.numpy())
        captured = {}
        orig = rcc.batch_fast_kmedoids_with_split

        def spy(*a_, **k_):
            r = orig(*a_, **k_)
            captured["medoids"] = r[1].clone()
            return r
        rcc.batch_fast_kmedoids_with_split = spy
        try:
            with torch.no_grad():
                o = model(ids, seg, amask, video, vmask)
        finally:
            rcc.batch_fast_kmedoids_with_split = orig
        out["s1_medoids"] = captured["medoids"].numpy()
        with torch.no_grad():
            logits, *_ = model.get_similarity_logits(o['sequence_output'], o['visual_output'], amask, vmask)
            # text-only / video-only calls (multi-sentence eval path, main.py:430,439)
            o_t = model(ids, seg, amask)
            o_v = model(video=video, video_mask=vmask)
        assert torch.equal(o_t['sequence_output'], o['sequence_output'])
        assert torch.allclose(o_v['visual_output'], o['visual_output'], rtol=0, atol=0, equal_nan=True)
        tag = "s1_pvp%d_" % pvp
        out[tag + "seq"], out[tag + "vis"], out[tag + "logits"] = (o['sequence_output'].numpy(), o['visual_output'].numpy(),
                                                                    logits.numpy())
        print(tag, "visual_output", tuple(o['visual_output'].shape), "logits", tuple(logits.shape), flush=True)
        if pvp == 0:
            # training branch, loss values only (clip4clip.py:245-262); all_gather needs a process group
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
            vmask_t = vmask.clone()
            vmask_t[2, 0, :2] = 1                           # (a fully masked clip makes the reference's loss NaN)
            out["s1_train_vmask"] = vmask_t.numpy()
            model.train()
            with torch.no_grad():
                ot = model(ids, seg, amask, video, vmask_t)
            model.eval()
            out["s1_train_loss"] = np.float32(ot['loss'].item())
            out["s1_train_sim_loss"] = np.float32(ot['sim_loss'].item())
            # CrossEn alone on a stored matrix
            import modules.losses as rl
            sim = torch.randn(9, 9, generator=g) * 3
            out["n4_sim"] = sim.numpy()
            out["n4_crossen"] = np.array([rl.CrossEn()(sim).item(), rl.CrossEn()(sim.T).item()], dtype=np.float32)
            print("train loss", out["s1_train_loss"], "crossen", out["n4_crossen"], flush=True)


def gen_prenorm(out):
    sys.path.insert(0, os.path.join("/root/reference", "modules"))
    import cluster.fast_kmeans as fk
    for tag, (seed, P, N, W, K, split) in PRENORM_CASES.items():
        X = norm32_tokens(seed, (P, N, W))
        Xt = torch.from_numpy(X)
        assert bool((torch.norm(Xt, dim=-1) == 32.0).all()), "token norms are not exactly 32 on this host"
        a, m = fk.batch_fast_kmedoids_with_split(Xt, K, distance="euclidean", threshold=1e-6, iter_limit=100, id_sort=True,
                                                 norm_p=2.0, split_size=split, pre_norm=True)
        out[f"{tag}_assign"], out[f"{tag}_medoids"] = a.numpy().astype(np.int16), m.numpy().astype(np.int16)
        # the same through the cosine metric (pre_norm + cosine is legal in the reference too)
        a, m = fk.batch_fast_kmedoids_with_split(Xt, K, distance="cosine", threshold=1e-6, iter_limit=100, id_sort=True,
                                                 norm_p=2.0, split_size=split, pre_norm=True)
        out[f"{tag}_cos_assign"], out[f"{tag}_cos_medoids"] = a.numpy().astype(np.int16), m.numpy().astype(np.int16)
        print(tag, "done", flush=True)


def gen_spectral(out):
    """N4: the forward pieces of spectral clustering captured from the reference (spectral.py): the normalised Laplacian
    of a heat-kernel graph (plain and with the spatial-temporal mask), the SVD sign flip on stored factors, and the
    k-medoids tail on a stored lattice embedding."""
    import cluster.spectral as sp
    g = torch.Generator().manual_seed(99)
    B, N, W, sigma = 3, 48, 32, 2.5
    X = torch.randn(B, N, W, generator=g) * 0.6
    Wm = sp.constructW(X, X, sigma=sigma, mode='HeatKernel')
    diag_D = Wm.sum(dim=-1)
    inv_D = torch.diag_embed(torch.pow(diag_D, -0.5))
    L_sym = torch.bmm(torch.bmm(inv_D, torch.diag_embed(diag_D) - Wm), inv_D)
    out["sp_x"], out["sp_sigma"], out["sp_w"], out["sp_lsym"] = X.numpy(), np.float32(sigma), Wm.numpy(), L_sym.numpy()
    graph = sp.spatial_temporal_graph(N, 16, s_kernel=3, t_kernel=3)
    Wg = sp.constructW(X, X, sigma=sigma, mode='HeatKernel', spatial_temporal_graph=graph)
    dg = Wg.sum(dim=-1)
    ig = torch.diag_embed(torch.pow(dg, -0.5))
    out["sp_graph"] = graph.numpy().astype(np.uint8)
    out["sp_lsym_graph"] = torch.bmm(torch.bmm(ig, torch.diag_embed(dg) - Wg), ig).numpy()
    U, S, Vh = torch.linalg.svd(L_sym, full_matrices=False)
    out["sp_u"], out["sp_s"], out["sp_vh"] = U.numpy(), S.numpy(), Vh.numpy()
    out["sp_u_flipped"] = sp.batch_sign_flip_rasmus_bro(U, S, Vh, backend="pytorch").numpy()
    print("spectral done", flush=True)


if __name__ == "__main__":
    out = {}
    gen_s1(out)
    gen_prenorm(out)
    gen_spectral(out)
    path = os.path.join(GOLD, "r2_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
