"""CPU oracle for the CLIP forward / similarity rows of the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/cluster_oracle.py for the rules).  Plain PyTorch fp32,
functional, driven by a state dict with the reference's key names.  Pinned against
the imported reference by tests/golden/clip_golden.npz (oracle/gen_golden.py clip).

Citations are relative to /root/reference.
"""
import math

import torch
import torch.nn.functional as F

from . import cluster_oracle as co


# constants of the reference's loader, dataloaders/decode.py:43-48
PIXEL_MEAN = (0.48145466, 0.4578275, 0.40821073)
PIXEL_STD = (0.26862954, 0.26130258, 0.27577711)


def loader_normalize(frames_u8, channels_last=False, mean=PIXEL_MEAN, std=PIXEL_STD):
    """What the reference's evaluation loader makes of decoded uint8 frames:
    HWC -> CHW permute (dataloaders/transforms.py:157), ``img.float().div_(255)`` (:166,
    GroupToTensorBCHW(div=True)), then TensorNormalize (:19-34) = torchvision
    ``functional.normalize``: ``tensor.sub_(mean[:,None,None]).div_(std[:,None,None])`` with the
    constants as fp32 tensors.  torchvision is a third-party dependency that is absent from this
    image (so dataloaders/transforms.py cannot be imported: parity of this row is pinned by
    this restatement of its published two-line algorithm, checked against plain IEEE numpy
    arithmetic in tests/test_oracle_clip.py, not by reference-generated fixtures).
    frames_u8 [N,3,H,W] or [N,H,W,3] uint8 -> [N,3,H,W] fp32."""
    x = torch.as_tensor(frames_u8)
    if channels_last:
        x = x.permute(0, 3, 1, 2)
    x = x.contiguous().float().div_(255.0)
    m = torch.as_tensor(mean, dtype=torch.float32)[None, :, None, None]
    sd = torch.as_tensor(std, dtype=torch.float32)[None, :, None, None]
    return x.sub_(m).div_(sd)


def layer_norm(x, w, b, eps=1e-5):
    """modules/clip.py:183-189: nn.LayerNorm evaluated in fp32."""
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps)


def quick_gelu(x):
    """modules/clip.py:192-194."""
    return x * torch.sigmoid(1.702 * x)


def mha(x, sd, pre, heads, causal):
    """nn.MultiheadAttention forward on [N, L, W] (batch first here; the reference feeds LND,
    clip.py:205,220-226): packed in_proj (rows q,k,v), heads = contiguous W/heads slices,
    softmax(q k^T / sqrt(d) + mask) v, out_proj."""
    N, L, W = x.shape
    d = W // heads
    qkv = x @ sd[pre + "attn.in_proj_weight"].float().t() + sd[pre + "attn.in_proj_bias"].float()
    q, k, v = qkv.split(W, dim=-1)
    q = q.view(N, L, heads, d).transpose(1, 2)
    k = k.view(N, L, heads, d).transpose(1, 2)
    v = v.view(N, L, heads, d).transpose(1, 2)
    s = (q @ k.transpose(-2, -1)) / math.sqrt(d)
    if causal:                                             # clip.py:448-454
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(N, L, W)
    return o @ sd[pre + "attn.out_proj.weight"].float().t() + sd[pre + "attn.out_proj.bias"].float()


def resblock(x, sd, pre, heads, causal, res_x=None):
    """ResidualAttentionBlock without the cluster hook (clip.py:240,251), x [N, L, W]; res_x: the residual the cluster
    module handed back (mean_residual, clip.py:239-242) - the attention branch is added to it instead of to x."""
    x = (x if res_x is None else res_x) + mha(layer_norm(x, sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"]), sd, pre, heads,
                                              causal)
    h = layer_norm(x, sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"])
    h = quick_gelu(h @ sd[pre + "mlp.c_fc.weight"].float().t() + sd[pre + "mlp.c_fc.bias"].float())
    return x + h @ sd[pre + "mlp.c_proj.weight"].float().t() + sd[pre + "mlp.c_proj.bias"].float()


def visual_forward(sd, video, T, cluster_plan=None, cluster_cfg=None, forced_medoids=None, return_hidden=False,
                   linear_patch='2d', mean_residual=()):
    """VisualTransformer.forward + the ln_post/proj tail of CLIP.encode_image
    (clip.py:304-349,460-469).  video [B*T,3,H,W]; cluster_plan {block_index(0-based): (T_new, K)};
    cluster_cfg dict(distance, threshold, iter_limit, norm_p, split_size, pre_norm[, algorithm, aggregation]).
    forced_medoids {block_index: int64 [T_new*B, K]} replaces the k-medoids result (to compare
    embeddings "given identical medoid sets").  mean_residual: block indices whose cluster module has
    mean_residual set (cluster.py:228-235: residual = the mean over each segment's frames of EVERY token, CLS included; the
    token count must not change).  Returns features [B*T_final, E] (and the hidden state [B*T_final, L, W])."""
    W = sd["visual.conv1.weight"].shape[0]
    p = sd["visual.conv1.weight"].shape[-1]
    heads = W // 64
    layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    if linear_patch == '3d':                                                           # clip.py:306-319
        x3 = video.float().reshape(-1, T, video.shape[-3], video.shape[-2], video.shape[-1]).permute(0, 2, 1, 3, 4)
        x3 = F.conv3d(x3, sd["visual.conv2.weight"].float(), stride=(1, p, p), padding=(1, 0, 0)).permute(0, 2, 1, 3, 4)
        x = x3.reshape(-1, x3.shape[-3], x3.shape[-2], x3.shape[-1])
    else:
        x = F.conv2d(video.float(), sd["visual.conv1.weight"].float(), stride=p)      # clip.py:324
    x = x.reshape(x.shape[0], W, -1).permute(0, 2, 1)                                  # [BT, n, W]
    cls = sd["visual.class_embedding"].float().expand(x.shape[0], 1, W)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"].float()         # :334-336
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])            # :338
    frames = T
    cluster_plan = cluster_plan or {}
    for i in range(layers):
        res_x = None
        if i in cluster_plan:                                                           # :236-242
            T_new, K = cluster_plan[i]
            x_lnd = x.permute(1, 0, 2).contiguous()
            if i in mean_residual:                                                      # cluster.py:228-235
                Lt, BT, _ = x_lnd.shape
                assert Lt == K + 1
                r = x_lnd.reshape(Lt, BT // frames, frames, W)
                r = torch.stack([it.mean(dim=2) for it in torch.split(r, frames // T_new, dim=2)], dim=2)
                res_x = r.contiguous().reshape(Lt, (BT // frames) * T_new, W).permute(1, 0, 2).contiguous()
            if forced_medoids is not None and i in forced_medoids:
                x_lnd = gather_with_medoids(x_lnd, frames, T_new, forced_medoids[i])
            elif (cluster_cfg or {}).get("algorithm", "kmediods++") != "kmediods++" or \
                    (cluster_cfg or {}).get("aggregation") not in [None, "None"]:
                c = cluster_cfg                                                         # N2 variants
                x_lnd = co.literal_token_cluster_variant(x_lnd, frames, T_new, K, c.get("algorithm", "kmediods++"),
                                                         c.get("aggregation"), None, None,
                                                         c.get("distance", "euclidean"), c.get("threshold", 1e-6),
                                                         c.get("iter_limit", 100), c.get("norm_p", 2.0),
                                                         c.get("split_size", 16), c.get("pre_norm", False))
            else:
                c = cluster_cfg or {}
                x_lnd = co.literal_token_cluster(x_lnd, frames, T_new, K, c.get("distance", "euclidean"),
                                                 c.get("threshold", 1e-6), c.get("iter_limit", 100),
                                                 c.get("norm_p", 2.0), c.get("split_size", 16),
                                                 c.get("pre_norm", False))
            x = x_lnd.permute(1, 0, 2).contiguous()
            frames = T_new
        x = resblock(x, sd, "visual.transformer.resblocks.%d." % i, heads, causal=False, res_x=res_x)
    feat = layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) @ sd["visual.proj"].float()
    return (feat, x) if return_hidden else feat


def gather_with_medoids(x_lnd, T, T_new, medoids):
    """The gather / CLS-mean half of TokenClusterInter.forward with given medoid ids
    (modules/cluster/cluster.py:287-289,303-310)."""
    tokens, cls = co.regroup_segments(x_lnd, T, T_new)
    P, _, W = tokens.shape
    B, fd, K = P // T_new, T // T_new, medoids.shape[1]
    picked = tokens[torch.arange(P).unsqueeze(-1), medoids]
    picked = picked.reshape(T_new, B, K, W).permute(1, 0, 2, 3).reshape(B * T_new, K, W)
    seg_cls = torch.stack([c.mean(dim=1) for c in torch.split(cls, fd, dim=1)], dim=1).reshape(B * T_new, 1, W)
    return torch.cat([seg_cls, picked], dim=1).permute(1, 0, 2).contiguous()


def text_forward(sd, ids):
    """CLIP.encode_text (clip.py:471-496): embed + positional, 12 causal blocks, ln_final,
    text_projection, row at the first argmax of the ids."""
    W = sd["ln_final.weight"].shape[0]
    heads = W // 64
    layers = len(set(k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")))
    x = sd["token_embedding.weight"].float()[ids] + sd["positional_embedding"].float()[:ids.shape[1]]
    for i in range(layers):
        x = resblock(x, sd, "transformer.resblocks.%d." % i, heads, causal=True)
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"]) @ sd["text_projection"].float()
    return x[torch.arange(x.shape[0]), ids.argmax(dim=-1)]


def video_mask_after_cluster(video_mask, max_frames, final_frames):
    """clip4clip.py:436-447: a segment inherits the mask of its last frame."""
    fd = max_frames // final_frames
    inds = torch.arange(fd - 1, video_mask.shape[-1], video_mask.shape[-1] // final_frames)
    return video_mask[:, inds]


def mean_pool_visual(visual, video_mask):
    """clip4clip.py:305-316 preceded/followed by the L2 normalisations of :357-360."""
    v = visual / visual.norm(dim=-1, keepdim=True)
    m = video_mask.to(torch.float).unsqueeze(-1)
    s = torch.sum(m, dim=1, dtype=torch.float)
    s[s == 0.] = 1.
    v = torch.sum(v * m, dim=1) / s
    return v / v.norm(dim=-1, keepdim=True)


def loose_similarity(sequence_output, visual_output, video_mask, logit_scale):
    """clip4clip.py:357-366 (meanP, eval branch): exp(logit_scale) * t_hat @ v_bar^T."""
    v = mean_pool_visual(visual_output.float(), video_mask)
    t = sequence_output.float().squeeze(1)
    t = t / t.norm(dim=-1, keepdim=True)
    return math.exp(float(logit_scale)) * torch.matmul(t, v.t())


def similarity_matrix_blocked(seq_batches, vis_batches, mask_batches, logit_scale):
    """main.py:502-534 (_run_on_single_gpu): text-batch x video-batch blocks concatenated."""
    rows = []
    for t in seq_batches:
        rows.append(torch.cat([loose_similarity(t, v, m, logit_scale) for v, m in zip(vis_batches, mask_batches)], dim=-1))
    return torch.cat(rows, dim=0)


def cross_en(sim_matrix):
    """modules/losses.py:8-18: mean over rows of -log_softmax(sim, -1)[i, i]."""
    return -torch.diag(F.log_softmax(sim_matrix.float(), dim=-1)).mean()


def clip4clip_forward(sd, ids, video, video_mask, max_frames, final_frames, cluster_plan, logit_scale,
                      pre_visual_pooling=False, forced_medoids=None):
    """CLIP4Clip.forward (eval branch) followed by get_similarity_logits (modules/clip4clip.py:199-243,412-434) for the
    meanP head: ids [B, 1, L] (or [B, L]), video [B, 1, T, 3, H, W], video_mask [B, 1, T] ->
    (sequence_output [B, 1, E], visual_output [B, T_new, E] or pooled [B, E], logits [B, B])."""
    ids = ids.view(-1, ids.shape[-1])
    T = video.shape[2]
    v = video.reshape((-1,) + tuple(video.shape[3:])).float()
    vmask = video_mask.view(-1, video_mask.shape[-1])
    if cluster_plan:
        vmask = video_mask_after_cluster(vmask, max_frames, final_frames)
    seq = text_forward(sd, ids).view(ids.shape[0], 1, -1)
    vis = visual_forward(sd, v, T, cluster_plan=cluster_plan, forced_medoids=forced_medoids).view(vmask.shape[0], -1, seq.shape[-1])
    if pre_visual_pooling:
        pooled = mean_pool_visual(vis, vmask)
        t = seq.squeeze(1)
        t = t / t.norm(dim=-1, keepdim=True)
        return seq, pooled, math.exp(float(logit_scale)) * torch.matmul(t, pooled.t())
    return seq, vis, loose_similarity(seq, vis, vmask, logit_scale)


def clip4clip_train_loss(sd, ids, video, video_mask, max_frames, final_frames, cluster_plan, logit_scale):
    """The loss of the training branch at world size 1 (clip4clip.py:245-262): (CrossEn(sim) + CrossEn(sim^T)) / 2."""
    _, _, sim = clip4clip_forward(sd, ids, video, video_mask, max_frames, final_frames, cluster_plan, logit_scale)
    return (cross_en(sim) + cross_en(sim.t())) / 2


def contrastive_loss_and_grads(sequence_output, visual_output, video_mask, logit_scale):
    """The training branch's loss (clip4clip.py:245-262) and torch.autograd's gradients of it with respect to
    sequence_output, visual_output and logit_scale -> (loss3 [CrossEn(S), CrossEn(S^T), mean], d_seq, d_vis, d_logit_scale)."""
    seq = sequence_output.detach().clone().float().requires_grad_(True)
    vis = visual_output.detach().clone().float().requires_grad_(True)
    ls = torch.tensor(float(logit_scale), requires_grad=True)
    v = vis / vis.norm(dim=-1, keepdim=True)
    m = video_mask.to(torch.float).unsqueeze(-1)
    den = torch.sum(m, dim=1, dtype=torch.float)
    den = torch.where(den == 0., torch.ones_like(den), den)
    v = torch.sum(v * m, dim=1) / den
    v = v / v.norm(dim=-1, keepdim=True)
    t = seq.reshape(seq.shape[0], -1)
    t = t / t.norm(dim=-1, keepdim=True)
    sim = ls.exp() * torch.matmul(t, v.t())
    l1, l2 = cross_en(sim), cross_en(sim.t())
    loss = (l1 + l2) / 2
    loss.backward()
    return torch.stack([l1.detach(), l2.detach(), loss.detach()]), seq.grad, vis.grad, ls.grad
