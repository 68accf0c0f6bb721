"""Round-4 / 5 GPU tests: the similarity GEMM on split planes, in_proj + attention in one launch (short and long sequences), the
training step (parameter gradients against torch.autograd on the reference, BertAdam, the graphed step)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def test_similarity_tiny_components():
    """The split-fp16 similarity GEMM on rows whose mass sits in ONE component while all others are tiny (their lo parts,
    and for the smallest also the hi parts, are subnormal fp16 numbers): logits vs float64 <= 5e-5, i.e. nothing is flushed
    (modules/clip4clip.py:357-366 computes this product in fp32)."""
    from centerclip_amd import ops
    gen = torch.Generator().manual_seed(5)
    Bt, Bv, E = 300, 260, 512
    t = torch.randn(Bt, E, generator=gen) * 1e-4
    v = torch.randn(Bv, E, generator=gen) * 1e-4
    t[torch.arange(Bt), torch.randint(0, E, (Bt,), generator=gen)] = 1.0
    v[torch.arange(Bv), torch.randint(0, E, (Bv,), generator=gen)] = 1.0
    t[:, :8] *= 1e-3                                               # a few components around 1e-7: hi subnormal too
    tn, vn = ops.normalize_rows(t.to(DEV)), ops.normalize_rows(v.to(DEV))
    got = ops.scaled_dot_nt(tn, vn, 1.0).double().cpu()
    want = (t.double() / t.double().norm(dim=-1, keepdim=True)) @ (v.double() / v.double().norm(dim=-1, keepdim=True)).t()
    assert float((got - want).abs().max()) <= 5e-5
    # the GEMM alone, on the fp32 rows it was given: a flushed lo part of a 1e-4 component beside a unit component is an
    # absolute error of 5e-8 in an entry of size 1e-4; the split scheme itself stays below 1e-10 there
    exact = tn.double().cpu() @ vn.double().cpu().t()
    small = exact.abs() < 1e-3
    assert int(small.sum()) > 1000
    assert float((got - exact).abs()[small].max()) <= 5e-9
    assert float((got - exact).abs().max()) <= 5e-7


def test_similarity_from_cached_operand_planes():
    """The evaluation loop's form of the matrix (main.py:502-534): operand planes written per batch with the features, ONE
    GEMM at the end == cc_loose_similarity on the same features bit for bit, ragged sizes, masks with zeros; and
    eval._similarity_matrix (S3) through it == the all-at-once op."""
    from centerclip_amd import ops, torch_ops as T, eval as ev
    gen = torch.Generator().manual_seed(3)
    Nt, Nv, Tn, E = 1203, 517, 3, 512
    t = torch.randn(Nt, E, generator=gen).to(DEV)
    v = torch.randn(Nv, Tn, E, generator=gen).to(DEV)
    m = (torch.rand(Nv, Tn, generator=gen) > 0.2).long().to(DEV)
    m[:, 0] = 1
    want = ops.loose_similarity(t, v, m, 1.0)
    tp = torch.cat([torch.ops.centerclip.normalize_rows_planes(t[i:i + 100].contiguous(), False) for i in range(0, Nt, 100)])
    vp = torch.zeros(T.padded_video_rows(Nv), 3 * E, device=DEV, dtype=torch.float16)
    vp[:Nv] = torch.cat([torch.ops.centerclip.video_pool_normalize_planes(v[i:i + 64].contiguous(), m[i:i + 64].contiguous())
                         for i in range(0, Nv, 64)])
    got = torch.ops.centerclip.scaled_dot_planes(tp, vp, Nv, float(np.float32(np.exp(np.float32(1.0)))))
    # (the same planes, the same GEMM: equal up to the last bit of the scale factor exp(logit_scale) the two entries form)
    assert float(((got - want).abs() / want.abs().clamp_min(1e-6)).max()) <= 2.5e-7
    with pytest.raises(RuntimeError):                              # too few (zeroed) padding rows
        torch.ops.centerclip.scaled_dot_planes(tp, vp[:Nv].contiguous(), Nv, 1.0)
    # fewer fp16 products per multiply-add (round 5): 2 = the text side rounded to fp16, 1 = both sides; against float64 on
    # the same unit rows - 3 products 5e-6, 2 products 1e-4, 1 product 2e-4 at the worst entry (the contract: 1e-3)
    tn = t.double() / t.double().norm(dim=-1, keepdim=True)
    vh = v.double() / v.double().norm(dim=-1, keepdim=True)
    md = m.double().unsqueeze(-1)
    vb = (vh * md).sum(1) / md.sum(1).clamp_min(1.0)
    exact = tn @ (vb / vb.norm(dim=-1, keepdim=True)).t()
    errs = {}
    for p_ in (3, 2, 1):
        c_ = torch.ops.centerclip.scaled_dot_planes(tp, vp, Nv, 1.0, p_)
        errs[p_] = float((c_.double() - exact).abs().max())
    assert errs[3] <= 5e-6 and errs[2] <= 1e-4 and errs[1] <= 2e-4 and errs[3] < errs[2] <= errs[1] * 1.5
    with pytest.raises(RuntimeError):
        torch.ops.centerclip.scaled_dot_planes(tp, vp, Nv, 1.0, 4)


def test_p1_lattice_vit_b16_shipped_shape():
    """Parity level P1 at the shape the shipped ViT-B/16 configurations run (scripts/activitynet.sh:104-122, lsmdc.sh:71-73:
    4 frames x 196 tokens per segment -> N = 784, K = 160, W = 768, split_size 4): two chunks, the second ragged; the
    reference's indices (tests/golden/r4_golden.npz, oracle/gen_golden_r4.py) bit for bit."""
    import os
    from centerclip_amd import cluster as cl
    from oracle.recipes import lattice
    g4 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r4_golden.npz"))
    seed, P, N, W, K, split, iters = [int(v) for v in g4["p1m_b16_cfg"]]
    X = torch.from_numpy(lattice(seed, (P, N, W))).to(DEV)
    a, m = cl.batch_fast_kmedoids_with_split(X, K, distance="euclidean", threshold=1e-6, iter_limit=iters,
                                             id_sort=True, norm_p=2.0, split_size=split, pre_norm=False)
    assert np.array_equal(m.cpu().numpy(), g4["p1m_b16_medoids"].astype(np.int64))
    assert np.array_equal(a.cpu().numpy(), g4["p1m_b16_assign"].astype(np.int64))


@pytest.mark.parametrize("tag", ["bg_visual", "bg_text", "bg_visual_b16", "bg_text77"])
def test_block_backward_against_reference_autograd(tag):
    """N4: forward and backward of one ResidualAttentionBlock (centerclip_amd/train.py: dgrad / wgrad on the forward GEMM
    kernel with swapped operand roles + csrc/backward.hip) against the reference block and torch.autograd on it
    (modules/clip.py:196-253; fixture oracle/gen_golden_r4.py): z, dx and the 12 parameter gradients within 1e-3 of each
    tensor's largest entry; through torch.autograd (block_apply) the same gradients land in .grad; identical bits on a
    second run."""
    import os
    from centerclip_amd.clip import ResidualAttentionBlock
    from centerclip_amd import train
    from oracle.recipes import BLOCK_GRAD_CASES, block_grad_inputs
    g4 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r4_golden.npz"))
    cfg = BLOCK_GRAD_CASES[tag]
    x, dz, sd = block_grad_inputs(cfg)
    blk = ResidualAttentionBlock(cfg["W"], cfg["heads"], attn_mask=(lambda n: None) if cfg["causal"] else None, block_id=1, args=None)
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    blk = blk.to(DEV)
    xt, dzt = torch.from_numpy(x).to(DEV), torch.from_numpy(dz).to(DEV)
    rel = lambda got, ref: float(np.abs(got - ref).max() / np.abs(ref).max())
    z, saved = train.block_forward_train(blk, xt)
    assert rel(z.cpu().numpy(), g4[f"{tag}_z"]) < 1e-3
    dx, grads = train.block_backward(blk, saved, dzt)
    errs = {"dx": rel(dx.cpu().numpy(), g4[f"{tag}_dx"])}
    for k, v in grads.items():
        errs[k] = rel(v.cpu().numpy().reshape(g4[f"{tag}_grad/{k}"].shape), g4[f"{tag}_grad/{k}"])
    print(f"[{tag}] relative errors:", {k: "%.1e" % e for k, e in errs.items()})
    assert len(grads) == 12 and max(errs.values()) < 1e-3, errs
    dx2, grads2 = train.block_backward(blk, saved, dzt)
    assert torch.equal(dx, dx2) and all(torch.equal(grads[k], grads2[k]) for k in grads)
    # the same through torch.autograd
    xa = xt.clone().requires_grad_(True)
    (train.block_apply(blk, xa) * dzt).sum().backward()
    assert torch.equal(xa.grad, dx)
    for k, p_ in blk.named_parameters():
        assert p_.grad is not None and torch.equal(p_.grad.reshape(-1), grads[k].reshape(-1)), k
    # frozen weights (main.py's freeze_layer_num): no gradient, no wgrad work for them, everything else the same bits;
    # a parameter changed in place between forward and backward is an error, not a silently stale W^T
    for p_ in blk.parameters():
        p_.grad = None
    blk.mlp["c_fc"].weight.requires_grad_(False)
    blk.attn.in_proj_weight.requires_grad_(False)
    xb = xt.clone().requires_grad_(True)
    (train.block_apply(blk, xb) * dzt).sum().backward()
    assert torch.equal(xb.grad, dx) and blk.mlp["c_fc"].weight.grad is None and blk.attn.in_proj_weight.grad is None
    assert torch.equal(blk.mlp["c_proj"].weight.grad.reshape(-1), grads["mlp.c_proj.weight"].reshape(-1))
    assert torch.equal(blk.mlp["c_fc"].bias.grad.reshape(-1), grads["mlp.c_fc.bias"].reshape(-1))
    zc = train.block_apply(blk, xt.clone().requires_grad_(True))
    with torch.no_grad():
        blk.ln_2.weight.mul_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        (zc * dzt).sum().backward()


# ----------------------------------------------------------------------------- in_proj + attention in one launch
def _inproj_inputs(M, W, seed):
    from centerclip_amd import ops
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.randn(M, W, generator=g) * 1.5 + 0.3).cuda()
    w = (torch.randn(3 * W, W, generator=g) * 0.04).cuda()
    b = (torch.randn(3 * W, generator=g) * 0.1).cuda()
    gamma = (1.0 + 0.1 * torch.randn(W, generator=g)).cuda()
    beta = (0.1 * torch.randn(W, generator=g)).cuda()
    h16, st, _ = ops.row_stats(x)
    wf, c1, c2 = ops.fold_layernorm_linear(w, b, gamma, beta)
    return h16, st, wf, c1, c2


@pytest.mark.parametrize("nseq,L,heads,causal", [(192, 50, 12, False), (48, 50, 12, False), (16, 32, 8, True), (7, 50, 12, False),
                                                  (3, 17, 2, True), (11, 33, 4, False), (5, 56, 2, True), (40, 9, 3, False)])
def test_inproj_attention_one_launch_is_bit_identical(nseq, L, heads, causal):
    """cc_inproj_attention_f16 (q, k, v kept in LDS) against cc_linear_ln_f16 + cc_attention_f16 on the same rows."""
    from centerclip_amd import ops
    W = heads * 64
    h16, st, wf, c1, c2 = _inproj_inputs(nseq * L, W, 100 + nseq)
    qkv = ops.linear_ln_f16(h16, wf, c1, c2, st, 1)
    want = ops.attention_f16(qkv, nseq, L, heads, causal=causal)
    got = ops.inproj_attention_f16(h16, wf, c1, c2, st, 1, nseq, L, heads, causal=causal)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    # ... and both against the plain fp32 formula of modules/clip.py:210-214 on the fp16 q, k, v
    q, k, v = (t.view(nseq, L, heads, 64).permute(0, 2, 1, 3) for t in qkv.float().split(W, dim=1))
    s = q @ k.transpose(-1, -2) / 8.0
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=s.device).triu(1)
    ref = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(nseq * L, W)
    assert relerr(got.float(), ref) < 2e-3


@pytest.mark.parametrize("nseq,L,heads,causal", [(24, 197, 12, False), (16, 101, 12, False), (9, 77, 8, True), (5, 64, 2, False),
                                                  (3, 57, 2, True), (2, 256, 1, False), (7, 130, 3, True), (1, 161, 2, False)])
def test_inproj_attention_one_launch_long_sequences(nseq, L, heads, causal):
    """Round 5: the one-launch form for 56 < L <= 256 (ViT-B/16's 197-token frames and 101 / 161-token clustered blocks, CLIP's
    77-token captions): whole sequences x one head per workgroup, scores and P in registers (the PV contraction takes the 32
    keys of a block in the accumulator's own order), against the two launches - same fp16 q, k, v, another summation order
    inside the matrix cores, so equal to the rounding of the fp16 output - and against the fp32 formula of modules/clip.py:210-214."""
    from centerclip_amd import ops
    W = heads * 64
    h16, st, wf, c1, c2 = _inproj_inputs(nseq * L, W, 300 + L)
    qkv = ops.linear_ln_f16(h16, wf, c1, c2, st, 1)
    want = ops.attention_f16(qkv, nseq, L, heads, causal=causal)
    got = ops.inproj_attention_f16(h16, wf, c1, c2, st, 1, nseq, L, heads, causal=causal)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    assert relerr(got.float(), want.float()) < 2e-3
    q, k, v = (t.view(nseq, L, heads, 64).permute(0, 2, 1, 3) for t in qkv.float().split(W, dim=1))
    s = q @ k.transpose(-1, -2) / 8.0
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=s.device).triu(1)
    ref = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(nseq * L, W)
    assert relerr(got.float(), ref) < 2e-3
    again = ops.inproj_attention_f16(h16, wf, c1, c2, st, 1, nseq, L, heads, causal=causal)
    assert torch.equal(got, again)                                                   # fixed order: the same bits on every run


def test_inproj_attention_seeded_shape_sweep():
    """A seeded sweep over the one-launch form's whole range (every path: 32- and 64-key slots, the long form with run-time and
    compile-time key-tile counts, one to eight sequences per row tile, ragged last tiles, causal and not; uniform and packed
    variable-length sequences) against the fp32 formula of modules/clip.py:210-214 on the fp16 q, k, v of the same launch inputs."""
    from centerclip_amd import ops
    rng = np.random.RandomState(20250929)
    for case in range(24):
        L = int(rng.choice([1, 7, 16, 31, 32, 33, 48, 56, 57, 63, 64, 65, 80, 96, 97, 128, 129, 160, 192, 193, 224, 225, 255, 256]))
        heads = int(rng.randint(1, 4))
        causal = bool(rng.randint(0, 2))
        nseq = int(rng.randint(1, max(2, min(40, 3000 // L))))
        packed = bool(rng.randint(0, 3) == 0) and L > 1
        W = heads * 64
        h16, st, wf, c1, c2 = _inproj_inputs(nseq * L, W, 1000 + case)
        if packed:
            lens = rng.randint(1, L + 1, size=nseq).astype(np.int32)
            lens[rng.randint(0, nseq)] = L
            off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
            got = ops.inproj_attention_f16(h16, wf, c1, c2, st, 1, nseq, L, heads, causal=causal,
                                           seq_off=torch.from_numpy(off).cuda(), seq_len=torch.from_numpy(lens).cuda())
        else:
            lens, off = np.full(nseq, L, dtype=np.int32), (np.arange(nseq) * L).astype(np.int32)
            got = ops.inproj_attention_f16(h16, wf, c1, c2, st, 1, nseq, L, heads, causal=causal)
        torch.cuda.synchronize()
        qkv = ops.linear_ln_f16(h16, wf, c1, c2, st, 1).float()
        scale_ref = 0.0
        for o, n in zip(off.tolist(), lens.tolist()):
            q, k, v = (t.view(n, heads, 64).permute(1, 0, 2) for t in qkv[o:o + n].split(W, dim=1))
            sc = q @ k.transpose(-1, -2) / 8.0
            if causal:
                sc = sc + torch.full((n, n), float("-inf"), device=sc.device).triu(1)
            ref = (sc.softmax(-1) @ v).permute(1, 0, 2).reshape(n, W)
            scale_ref = max(scale_ref, float(ref.abs().max()))
            err = float((got[o:o + n].float() - ref).abs().max())
            assert err <= 2e-3 * max(scale_ref, 1e-3), (case, L, heads, causal, nseq, packed, o, n, err)
        assert torch.isfinite(got.float()).all()
        if packed:
            assert not got[int(lens.sum()):].any()


def test_inproj_attention_packed_long_captions():
    """Packed variable-length captions with the 77-token context (3 sequence slots of 96 keys per row tile): every caption
    against the two-launch result of that caption alone."""
    from centerclip_amd import ops
    heads, W, L = 8, 512, 77
    lens = [77, 5, 40, 1, 76, 20, 9, 65, 33, 2, 77]                                 # 11 captions: 4 row tiles of 3
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    Mv = int(sum(lens))
    h16, st, wf, c1, c2 = _inproj_inputs(len(lens) * L, W, 9)
    seq_off = torch.from_numpy(off).cuda()
    seq_len = torch.tensor(lens, dtype=torch.int32).cuda()
    got = ops.inproj_attention_f16(h16, wf, c1, c2, st, 1, len(lens), L, heads, causal=True, seq_off=seq_off, seq_len=seq_len)
    torch.cuda.synchronize()
    for s, (o, n) in enumerate(zip(off, lens)):
        qkv = ops.linear_ln_f16(h16[o:o + n].contiguous(), wf, c1, c2, st[o:o + n].contiguous(), 1)
        want = ops.attention_f16(qkv, 1, n, heads, causal=True)
        assert relerr(got[o:o + n].float(), want.float()) < 2e-3, (s, n)
    assert not got[Mv:].any()                                                        # rows behind the packed captions: untouched


def test_inproj_attention_packed_captions():
    """Variable-length sequences packed back to back (the compacted captions of the text tower): every caption must equal
    the two-launch result of that caption alone."""
    from centerclip_amd import ops
    heads, W, L = 8, 512, 32
    lens = [32, 5, 17, 1, 32, 20, 9, 31, 16, 2, 27, 13, 8, 32, 3, 19, 11]          # 17 captions: 3 row tiles of 8
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    Mv = int(sum(lens))
    h16, st, wf, c1, c2 = _inproj_inputs(len(lens) * L, W, 7)
    seq_off = torch.from_numpy(off).cuda()
    seq_len = torch.tensor(lens, dtype=torch.int32).cuda()
    got = ops.inproj_attention_f16(h16, wf, c1, c2, st, 1, len(lens), L, heads, causal=True, seq_off=seq_off, seq_len=seq_len)
    torch.cuda.synchronize()
    for s, (o, n) in enumerate(zip(off, lens)):
        qkv = ops.linear_ln_f16(h16[o:o + n].contiguous(), wf, c1, c2, st[o:o + n].contiguous(), 1)
        want = ops.attention_f16(qkv, 1, n, heads, causal=True)
        assert torch.equal(got[o:o + n], want), (s, n)
    assert not got[Mv:].any()                                                        # rows behind the packed captions: untouched


# ----------------------------------------------------------------------------- N4: a whole training step
def _golden_clip():
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_golden.npz"))
    sd = {k[3:]: torch.from_numpy(g[k].astype(np.float32) if g[k].dtype == np.float16 else g[k]) for k in g.files
          if k.startswith("sd/")}
    return g, sd


def _train_cfg(T):
    from argparse import Namespace
    return Namespace(cluster_inter=1, cluster_algo='kmediods++', max_frames=T, target_frames_blocks=[4, 2, 2],
                     cluster_num_blocks=[16, 6, 6], cluster_distance='euclidean', cluster_threshold=1e-6, cluster_iter_limit=100,
                     minkowski_norm_p=2.0, pretrained_clip_name='ViT-B/32', aggregation=None, pre_norm=False, loose_type=True,
                     sim_header='meanP', linear_patch='2d')


def test_training_step_gradients_against_reference_autograd():
    """CLIP4Clip.forward in training mode (the towers of centerclip_amd.train: patch embedding, ln_pre, blocks with the k-medoids
    module in front of block 2, heads; meanP similarity + symmetric CrossEn) and loss.backward(): the loss, the features and
    the gradient of EVERY parameter against torch.autograd on the reference model (fixture tr_* of r4_golden.npz)."""
    from centerclip_amd.clip4clip import CLIP4Clip
    g, sd = _golden_clip()
    r4 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r4_golden.npz"))
    B, T = int(g["cfg"][10]), int(g["cfg"][11])
    model = CLIP4Clip.from_state_dict(sd, _train_cfg(T)).float().to("cuda:0").train()
    video = torch.from_numpy(g["video"]).view(B, 1, T, 3, 64, 64).cuda()
    ids = torch.from_numpy(g["t_ids"])[:B].cuda()
    vmask = torch.ones(B, 1, T, dtype=torch.long, device="cuda:0")
    out = model(ids, torch.zeros_like(ids), (ids > 0).long(), video, vmask)
    out["loss"].backward()
    torch.cuda.synchronize()
    assert abs(float(out["loss"].detach()) - float(r4["tr_loss"])) < 2e-3 * max(1.0, abs(float(r4["tr_loss"])))
    assert relerr(out["visual_output"].detach().reshape(-1, 64).cpu(), torch.from_numpy(r4["tr_vfeat"])) < 2e-3
    assert relerr(out["sequence_output"].detach().reshape(-1, 64).cpu(), torch.from_numpy(r4["tr_tfeat"])) < 2e-3
    named = dict(model.clip.named_parameters())
    keys = [k[len("tr_grad/"):] for k in r4.files if k.startswith("tr_grad/")]
    assert len(keys) == 74
    worst = (0.0, None)
    for k in keys:
        want = torch.from_numpy(r4["tr_grad/" + k])
        p = named[k]
        assert p.grad is not None, k
        e = relerr(p.grad.detach().float().cpu().reshape(want.shape), want)
        worst = max(worst, (e, k))
        assert e < 1e-2, (k, e)                                   # relative to the tensor's largest entry; fp16 operands
    print("worst gradient error", worst)
    # every parameter the reference reaches is reached here too, and nothing else
    assert {k for k, p in named.items() if p.grad is not None} == set(keys)


def test_bertadam_matches_reference_optimizer():
    """centerclip_amd.train.BertAdam (cc_bertadam_step_f32) vs three steps of utils/optimization.BertAdam: clipping engaged in
    step 2 (|g| large), weight decay in group 1 only, warmup_linear schedule."""
    from centerclip_amd.train import BertAdam, warmup_linear
    r4 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r4_golden.npz"))
    p0 = [torch.nn.Parameter(torch.from_numpy(r4["ba_p0"].copy()).cuda()), torch.nn.Parameter(torch.from_numpy(r4["ba_p1"].copy()).cuda())]
    opt = BertAdam([{'params': [p0[0]], 'weight_decay': 0.2}, {'params': [p0[1]], 'weight_decay': 0.0}], lr=1e-2, warmup=0.1,
                   t_total=20, schedule='warmup_linear', b1=0.9, b2=0.98, e=1e-6, max_grad_norm=1.0)
    assert opt.get_lr() == []
    for it in range(3):
        for j in range(2):
            p0[j].grad = torch.from_numpy(r4[f"ba_g{it}_{j}"].copy()).cuda()
        opt.step()
        for j in range(2):
            np.testing.assert_allclose(p0[j].detach().cpu().numpy(), r4[f"ba_after{it}_{j}"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(opt.state[p0[0]]['next_m'].cpu().numpy(), r4["ba_m_0"], rtol=2e-6, atol=1e-8)
    np.testing.assert_allclose(opt.state[p0[0]]['next_v'].cpu().numpy(), r4["ba_v_0"], rtol=2e-6, atol=1e-10)
    assert opt.state[p0[0]]['step'] == 3 and abs(opt.get_lr()[0] - 1e-2 * warmup_linear(3 / 20, 0.1)) < 1e-12


def test_train_epoch_runs_and_lowers_the_loss():
    """main.py:291-378 on this path: a few steps of train_epoch on one synthetic batch (fp32 master weights, BertAdam with
    the reference's parameter groups): the loss goes down and logit_scale stays clamped."""
    from argparse import Namespace
    from centerclip_amd.clip4clip import CLIP4Clip
    from centerclip_amd.train import BertAdam, prep_optim_params_groups, train_epoch
    g, sd = _golden_clip()
    B, T = int(g["cfg"][10]), int(g["cfg"][11])
    model = CLIP4Clip.from_state_dict(sd, _train_cfg(T)).float().to("cuda:0")
    video = torch.from_numpy(g["video"]).view(B, 1, T, 3, 64, 64)
    ids = torch.from_numpy(g["t_ids"])[:B]
    batch = (ids, (ids > 0).long(), torch.zeros_like(ids), video, torch.ones(B, 1, T, dtype=torch.long))
    args = Namespace(lr=1e-3, wd=0.2, new_added_modules=["Cross", "cluster_embed"], gradient_accumulation_steps=1, clip_grad_norm=None)
    opt = BertAdam(prep_optim_params_groups(args, model, coef_lr=1.0), lr=args.lr, warmup=0.1, t_total=40,
                   schedule='warmup_cosine', b1=0.9, b2=0.98, e=1e-6, max_grad_norm=1.0)
    losses = []
    gs = 0
    for ep in range(3):
        l, gs = train_epoch(ep, args, model, [batch] * 4, "cuda:0", opt, gs)
        losses.append(l)
    assert gs == 12 and np.isfinite(losses).all() and losses[-1] < losses[0] - 0.05, losses
    assert 0.1 <= float(model.clip.logit_scale) <= 4.6052


def test_inproj_attention_argument_range():
    """Outside the one-launch form's range the entry point says so (the towers then run the two launches)."""
    from centerclip_amd import ops
    h16, st, wf, c1, c2 = _inproj_inputs(2 * 257, 128, 3)
    with pytest.raises(RuntimeError, match="unsupported"):
        ops.inproj_attention_f16(h16, wf, c1, c2, st, 1, 2, 257, 2)              # L > 256
    h16, st, wf, c1, c2 = _inproj_inputs(8, 64, 4)
    got = ops.inproj_attention_f16(h16, wf, c1, c2, st, 1, 8, 1, 1)                # one token per sequence: softmax of one score
    qkv = ops.linear_ln_f16(h16, wf, c1, c2, st, 1)
    assert torch.equal(got, qkv[:, 128:192])                                        # attention output = v
    with pytest.raises(ValueError, match="seq_off and seq_len"):                   # (the C entry point: CC_ERR_INVALID)
        ops.inproj_attention_f16(h16, wf, c1, c2, st, 1, 8, 1, 1, seq_off=torch.zeros(8, dtype=torch.int32, device="cuda"))


@pytest.mark.parametrize("M,C", [(9600, 768), (323, 512), (50, 64), (64, 3072), (1, 4)])
def test_cast_transpose_operand_copies(M, C):
    """cc_cast_transpose_f16: one read of a matrix -> its fp16 copy and the zero-padded transpose a wgrad GEMM multiplies; scaled
    (gradients: the device-chosen power of two of cc_cast_scaled_f16), unscaled (weights), fp16 input (saved activations)."""
    from centerclip_amd import train as cctrain
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 3e-4).cuda()
    Mp = -(-M // 64) * 64
    out, out_t, scale = cctrain._cast_transpose(x, scaled=True)
    ref16, ref_scale = cctrain._cast_scaled(x)
    torch.cuda.synchronize()
    assert float(scale) == float(ref_scale) and float(scale) >= 1.0 and float(torch.log2(scale)) % 1 == 0
    assert torch.equal(out, ref16) and out_t.shape == (C, Mp)
    assert torch.equal(out_t[:, :M], ref16.t()) and not out_t[:, M:].any()
    out, out_t, scale = cctrain._cast_transpose(x, scaled=False)
    assert scale is None and torch.equal(out, x.half()) and torch.equal(out_t[:, :M], x.half().t()) and not out_t[:, M:].any()
    _, _, _, cs = cctrain._cast_transpose(x, scaled=True, col_sums=True)            # the bias gradient from the same read
    want = x.double().sum(dim=0)
    assert float((cs.double() - want).abs().max()) <= 1e-5 * float(x.abs().sum(dim=0).max()) + 1e-12
    xh = x.half() * 1000
    same, out_t, _ = cctrain._cast_transpose(xh, scaled=False)
    assert same.data_ptr() == xh.data_ptr() and torch.equal(out_t[:, :M], xh.t()) and not out_t[:, M:].any()


@pytest.mark.parametrize("M,N,K", [(768, 3072, 9600), (9600, 768, 2304), (320, 512, 512)])
def test_linear_with_the_operand_scale_undone_in_the_epilogue(M, N, K):
    """cc_linear_unscaled_f16 = cc_linear_f16 ("f32") followed by cc_unscale_f32, bit for bit (the scale is a power of two)."""
    from centerclip_amd import ops, train as cctrain
    g = torch.Generator().manual_seed(M + N + K)
    dy = (torch.randn(M, K, generator=g) * 2e-5).cuda()
    w16 = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().half()
    dy16, scale = cctrain._cast_scaled(dy)
    want = cctrain._unscale(ops.linear_f16(dy16, w16, None, "f32"), scale)
    got = cctrain._linear_unscaled(dy16, w16, scale)
    torch.cuda.synchronize()
    assert float(scale) > 1.0 and torch.equal(got, want)
    ref = dy.double() @ w16.double().t()
    assert float((got.double() - ref).abs().max()) <= 2e-3 * float(ref.abs().max())


@pytest.mark.parametrize("nseq,L,heads,causal", [(6, 50, 2, False), (5, 32, 3, True), (3, 64, 1, False), (4, 17, 2, True),
                                                 (3, 197, 2, False), (2, 161, 1, False), (3, 77, 2, True), (2, 256, 1, False),
                                                 (2, 65, 1, True), (1, 130, 2, True)])
def test_attention_backward_on_the_matrix_cores(nseq, L, heads, causal):
    """cc_attention_backward_f16 (fp16 MFMA operands, per-head power-of-two scales for dO and dS) against torch.autograd in fp64
    on the same fp16 q, k, v: every part of d_qkv within 2e-3 of its largest entry, gradients of tiny magnitude included.
    L <= 64: the one-launch form; 64 < L <= 256 (ViT-B/16's 197 tokens, its clustered 161, the text tower's 77): the
    query-side + key-side pair."""
    from centerclip_amd import _lib as L_, train as cctrain                                 # noqa: F401
    from centerclip_amd.torch_ops import _st
    W = heads * 64
    g = torch.Generator().manual_seed(nseq * 100 + L)
    qkv = torch.randn(nseq * L, 3 * W, generator=g).cuda().half()
    for mag in (1.0, 3e-7):
        d_out = (torch.randn(nseq * L, W, generator=g) * mag).cuda()
        # sequences get different magnitudes: the scale is chosen per head of a sequence
        d_out.view(nseq, L, W)[0] *= 64.0
        got = torch.empty(nseq * L, 3 * W, device="cuda")
        am = torch.zeros(2, device="cuda")
        nb = L_.lib().cc_attention_backward_workspace_bytes(nseq, L, heads)
        assert (nb == 0) == (L <= 64)
        ws = torch.empty(max(nb, 1), dtype=torch.uint8, device="cuda")
        L_.check(L_.lib().cc_attention_backward_f16(L_.ptr(qkv), L_.ptr(d_out), L_.ptr(got), nseq, L, heads, W, int(causal),
                                                    L_.ptr(am), L_.ptr(ws), nb, _st(got)), "cc_attention_backward_f16")
        x = qkv.double().view(nseq, L, 3, heads, 64).permute(2, 0, 3, 1, 4).detach().requires_grad_(True)   # [3, n, h, L, 64]
        sc = x[0] @ x[1].transpose(-1, -2) / 8.0
        if causal:
            sc = sc + torch.full((L, L), float("-inf"), device="cuda", dtype=torch.float64).triu_(1)
        out = (sc.softmax(dim=-1) @ x[2]).permute(0, 2, 1, 3).reshape(nseq * L, W)
        (out * d_out.double()).sum().backward()
        want = x.grad.permute(1, 3, 0, 2, 4).reshape(nseq * L, 3 * W)
        torch.cuda.synchronize()
        for part in range(3):
            a, b = got[:, part * W:(part + 1) * W].double(), want[:, part * W:(part + 1) * W]
            assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()), (part, mag)
        assert float(am[0]) == float(got.abs().max())


def test_gradient_producers_publish_their_largest_magnitude():
    """cc_layernorm_backward_f32 / cc_quick_gelu_backward_f16 / cc_attention_backward_f16 with an amax pointer: the float equals
    the largest |output|, and cc_cast_transpose_f16(scaled = 2) on it gives the copies and the scale of the two-pass form."""
    from centerclip_amd import _lib as L, train as cctrain
    from centerclip_amd.torch_ops import _st
    g = torch.Generator().manual_seed(77)
    rows, W, heads, Lt = 12 * 50, 128, 2, 50
    am = torch.zeros(3, 2, device="cuda")
    x = torch.randn(rows, W, generator=g).cuda()
    dy = (torch.randn(rows, W, generator=g) * 3e-4).cuda()
    dres = (torch.randn(rows, W, generator=g) * 1e-4).cuda()
    dx, _, _ = cctrain._ln_backward(x, torch.ones(W, device="cuda"), dy, dres, amax=am[0])
    u_pre = torch.randn(rows, 4 * W, generator=g).cuda().half()
    du = (torch.randn(rows, 4 * W, generator=g) * 2e-5).cuda()
    du_pre = torch.empty_like(du)
    L.check(L.lib().cc_quick_gelu_backward_f16(L.ptr(u_pre), L.ptr(du), L.ptr(du_pre), du.numel(), L.ptr(am[1]), _st(du)), "gelu bwd")
    qkv = torch.randn(rows, 3 * W, generator=g).cuda().half()
    datt = (torch.randn(rows, W, generator=g) * 1e-3).cuda()
    dqkv = torch.empty(rows, 3 * W, device="cuda")
    L.check(L.lib().cc_attention_backward_f16(L.ptr(qkv), L.ptr(datt), L.ptr(dqkv), rows // Lt, Lt, heads, W, 0, L.ptr(am[2]),
                                              None, 0, _st(datt)), "attention bwd")
    torch.cuda.synchronize()
    for i, t in enumerate((dx, du_pre, dqkv)):
        assert float(am[i, 0]) == float(t.abs().max()) > 0.0
        a16, a16t, scale = cctrain._cast_transpose(t, scaled=True)
        b16, b16t, scale2 = cctrain._cast_transpose(t, scaled=True, amax=am[i])
        torch.cuda.synchronize()
        assert float(scale) == float(scale2) and torch.equal(a16, b16) and torch.equal(a16t, b16t)


def test_graphed_train_step_equals_eager_steps():
    """train.GraphedTrainStep (forward + backward + BertAdam captured into one hipGraph, the schedule's value through a device
    float) against the same steps launched op by op: identical parameters after 3 calls = 3 optimizer steps (the first
    call's eager warm-up runs on a snapshot that is put back before its replay: one call, one step - main.py:300-340)."""
    from argparse import Namespace
    from centerclip_amd.clip4clip import CLIP4Clip
    from centerclip_amd.train import BertAdam, prep_optim_params_groups, train_epoch, GraphedTrainStep
    g, sd = _golden_clip()
    B, T = int(g["cfg"][10]), int(g["cfg"][11])
    video = torch.from_numpy(g["video"]).view(B, 1, T, 3, 64, 64)
    ids = torch.from_numpy(g["t_ids"])[:B]
    batch = (ids, (ids > 0).long(), torch.zeros_like(ids), video, torch.ones(B, 1, T, dtype=torch.long))
    args = Namespace(lr=1e-3, wd=0.2, new_added_modules=["Cross"], gradient_accumulation_steps=1, clip_grad_norm=None)

    def make(capturable):
        m = CLIP4Clip.from_state_dict(sd, _train_cfg(T)).float().to("cuda:0")
        o = BertAdam(prep_optim_params_groups(args, m), lr=args.lr, warmup=0.2, t_total=20, schedule='warmup_linear', b1=0.9, b2=0.98,
                     e=1e-6, max_grad_norm=1.0, capturable=capturable)
        return m, o
    m0, o0 = make(False)
    train_epoch(0, args, m0, [batch] * 3, "cuda:0", o0, 0)
    m1, o1 = make(True)
    stepper = GraphedTrainStep(m1, o1)
    loss = stepper(batch)                                               # first call: warm-up on a snapshot + capture + 1 replay
    torch.cuda.synchronize()
    assert all(st["step"] == 1 for st in o1.state.values())
    for _ in range(2):
        loss = stepper(batch)
    torch.cuda.synchronize()
    assert np.isfinite(float(loss))
    assert all(st["step"] == 3 for st in o1.state.values())
    for (k, p0), (_, p1) in zip(m0.named_parameters(), m1.named_parameters()):
        assert torch.equal(p0, p1), k
