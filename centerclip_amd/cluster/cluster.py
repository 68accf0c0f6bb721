"""Mirror of modules/cluster/cluster.py: get_cluster_inter (:15-63) and TokenClusterInter (:66-352) for the
algorithms 'kmediods++' and 'spectral' (aggregation None or mean, cluster_embedding, adaptive_cls), 'pooling' and
'sparse_sampling' (eval: fixed ids; training: the reference's random ids), mean_residual.  Differentiable with respect to x, cluster_embed and cls_multiplier
(torch.ops.centerclip.token_cluster_train / token_cluster_backward, cc_token_cluster_backward_f32)."""
import numpy as np
import torch

from .. import _lib as L
from .. import torch_ops  # noqa: F401  (registers torch.ops.centerclip)


def get_cluster_inter(width, block_id, args=None):
    """Decide per transformer block whether a TokenClusterInter is inserted (cluster.py:15-63).
    block_id starts at 1; ``args.target_frames_blocks`` is prefixed with ``args.max_frames``."""
    if args is None or not args.cluster_inter:
        return None
    frames = [args.max_frames] + list(args.target_frames_blocks)
    cluster_num = args.cluster_num_blocks[block_id - 1]
    before_cluster_num = args.cluster_num_blocks[max(block_id - 2, 0)]
    before_frames, after_frames = frames[block_id - 1], frames[block_id]
    fires = (cluster_num is not None and cluster_num > 1) and \
        (before_frames > after_frames or before_cluster_num > cluster_num)
    if not fires:
        return None
    return TokenClusterInter(algorithm=args.cluster_algo, block_id=block_id,
                             before_cluster_num=before_cluster_num, cluster_num=cluster_num,
                             before_block_frames=before_frames, after_block_frames=after_frames,
                             original_frame=args.max_frames, distance=args.cluster_distance,
                             threshold=args.cluster_threshold, iter_limit=args.cluster_iter_limit,
                             id_sort=True, norm_p=args.minkowski_norm_p,
                             aggregation=getattr(args, 'aggregation', None),
                             split_size=4 if args.pretrained_clip_name == 'ViT-B/16' else 16,
                             cluster_embedding=getattr(args, 'cluster_embedding', False),
                             cluster_frame_embedding=getattr(args, 'cluster_frame_embedding', False),
                             adaptive_cls=False,          # hard-coded in the reference too (cluster.py:59)
                             spectral_sigma=getattr(args, 'spectral_sigma', 2.0),
                             spectral_graph=getattr(args, 'spectral_graph', 'HeatKernel'),
                             spectral_knn_k=getattr(args, 'spectral_knn_k', 1),
                             spectral_spatial_temporal_graph=getattr(args, 'spectral_spg', 0),
                             svd_correct_sign=getattr(args, 'svd_correct_sign', 1),
                             transformer_width=width, pre_norm=getattr(args, 'pre_norm', False))


class _GatherGiven(torch.autograd.Function):
    """Medoid-token gather + per-segment CLS mean for GIVEN ids (training-mode sparse_sampling), differentiable in x."""

    @staticmethod
    def forward(ctx, x, medoids, frame_major, T, T_new, K):
        ctx.save_for_backward(x, medoids)
        ctx.cfg = (frame_major, T, T_new, K)
        empty = torch.empty(0, 1, dtype=torch.long, device=x.device)
        return torch.ops.centerclip.token_apply_selection(x, frame_major, T, T_new, K, 0, medoids, empty, None, None)

    @staticmethod
    def backward(ctx, g):
        x, medoids = ctx.saved_tensors
        frame_major, T, T_new, K = ctx.cfg
        empty = torch.empty(0, 1, dtype=torch.long, device=x.device)
        gx, _, _ = torch.ops.centerclip.token_cluster_backward(g.contiguous(), x, frame_major, T, T_new, K, 0, 0, medoids, empty,
                                                               None, None, False, False)
        return gx, None, None, None, None, None


class TokenClusterInter(torch.nn.Module):
    """Token clustering between transformer blocks: T frames -> T_new segments, the fd*n patch
    tokens of a segment -> K medoid tokens (ascending ids), CLS = mean of the segment's CLS.

    Built: 'kmediods++' with aggregation None (medoid tokens, the shipped scripts) or any other value (cluster
    means, cluster.py:291-301), cluster_embedding, adaptive_cls; 'pooling'; 'sparse_sampling' in eval mode.
    cluster_frame_embedding is accepted the way the reference treats it: the parameter exists (checkpoints that carry
    `cluster_frame_embed` load) and the forward does not use it (its use is commented out, :283-285).
    'spectral' (cluster.py:262-272): the selection comes from spectral clustering of the segment's tokens (graph Laplacian,
    batched Jacobi eigensolver, k-medoids on the embedding - all on the device, cluster/spectral.py), the rest is shared.
    'sparse_sampling' in training mode draws the reference's random ids (same NumPy calls, per segment).  mean_residual
    (not reachable from the reference's arguments) is built for the module / block-level forwards.  The shift algorithms
    raise NotImplementedError at construction.
    """

    def __init__(self, algorithm='kmediods++', block_id=1, before_cluster_num=49, cluster_num=49,
                 before_block_frames=12, after_block_frames=12, original_frame=12, distance='euclidean',
                 threshold=1e-6, iter_limit=80, id_sort=True, aggregation=None, split_size=8, norm_p=2.0,
                 spectral_graph='HeatKernel', spectral_sigma=2.0, spectral_knn_k=0,
                 spectral_spatial_temporal_graph=False, cluster_embedding=False, cluster_frame_embedding=False,
                 adaptive_cls=False, mean_residual=False, transformer_width=768, save_feature_path=None,
                 svd_correct_sign=1, pre_norm=False):
        super().__init__()
        assert algorithm in ['kmediods++', 'pooling', 'sparse_sampling', 'spectral', 'temporal_shift', 'token_shift']
        if algorithm not in ('kmediods++', 'pooling', 'sparse_sampling', 'spectral'):
            raise NotImplementedError("centerclip_amd builds cluster_algo 'kmediods++', 'spectral', 'pooling' and "
                                      "'sparse_sampling' (got %r)" % algorithm)
        kmed = algorithm in ('kmediods++', 'spectral')                            # cluster.py:240 (shared branch)
        self.cluster_embedding = bool(cluster_embedding) if kmed else False      # cluster.py:154-156
        self.adaptive_cls = bool(adaptive_cls) if kmed else False
        scale = transformer_width ** -0.5
        if self.cluster_embedding:                                                # cluster.py:161-164
            self.cluster_embed = torch.nn.Parameter(scale * torch.randn(cluster_num, transformer_width))
        if bool(cluster_frame_embedding) and kmed:                                # cluster.py:155,167-169 (unused in forward)
            self.cluster_frame_embed = torch.nn.Parameter(
                scale * torch.randn(before_block_frames // after_block_frames, transformer_width).unsqueeze(1))
        if self.adaptive_cls:                                                     # cluster.py:170-172
            m = [1 / (before_block_frames // after_block_frames) for _ in range(before_block_frames)]
            self.cls_multiplier = torch.nn.Parameter(torch.tensor(m).float().reshape(1, before_block_frames, 1, 1))
        assert id_sort, "the reference hard-codes id_sort=True (cluster.py:49)"
        self.algorithm = algorithm
        self.block_id = block_id
        self.original_frame = original_frame
        self.before_cluster_num = before_cluster_num
        self.cluster_num = cluster_num
        self.before_block_frames = before_block_frames
        self.after_block_frames = after_block_frames
        self.frame_duration = before_block_frames // after_block_frames
        self.distance = distance
        self.threshold = threshold
        self.iter_limit = iter_limit
        self.id_sort = id_sort
        self.aggregation = aggregation
        self.split_size = split_size
        self.norm_p = norm_p
        self.pre_norm = pre_norm
        # cluster.py:228-237: the residual connection of the block becomes the mean over each segment's frames of EVERY
        # token (needs an unchanged token count); module / block-level forwards only - not inside the fused encoder
        self.mean_residual = bool(mean_residual)
        self.last_medoids = None
        # cluster_algo 'spectral' (cluster.py:142-152,174-182)
        self.spectral_graph = spectral_graph
        self.spectral_sigma = spectral_sigma
        fd = before_block_frames // after_block_frames
        if spectral_knn_k < 5:                     # "when K of spectral_knn_k is small, use an adaptive number"
            self.spectral_knn_k = int(5 * fd) if before_cluster_num < 100 else int(5 * fd + 5)
        else:
            self.spectral_knn_k = spectral_knn_k
        self.svd_correct_sign = svd_correct_sign
        self.spectral_spatial_temporal_graph = spectral_spatial_temporal_graph
        if algorithm == 'spectral' and spectral_spatial_temporal_graph:
            from .spectral import spatial_temporal_graph
            spg = spatial_temporal_graph(before_cluster_num * fd, before_cluster_num,
                                         s_kernel=9 if before_cluster_num < 100 else 19, t_kernel=7)
            self.register_buffer("spg", spg.unsqueeze(0).float())
        else:
            self.spg = None

    def cluster_frame_major(self, x_nld, keep_ids=False):
        """Fast path used by the HIP transformer: x [B*T, 1+n, W] (frame-major) ->
        [B*T_new, 1+K, W].  Same arithmetic as forward(), different strides."""
        BT, Lt, W = x_nld.shape
        return self._run(x_nld, tok_stride=W, frame_stride=Lt * W, BT=BT, Lt=Lt, W=W, frame_major=True,
                         keep_ids=keep_ids)

    def forward(self, x):
        """x [1+n, B*T, W] (LND) -> (x' [1+K, B*T_new, W], residual_x or None)   (cluster.py:206,350-352)"""
        Lt, BT, W = x.shape
        residual_x = None
        if self.mean_residual:                       # cluster.py:228-235 = the 'pooling' reduction of the same tensor
            assert Lt == self.cluster_num + 1
            residual_x = self._pool_frames(x)
        return self._run(x, tok_stride=BT * W, frame_stride=W, BT=BT, Lt=Lt, W=W, frame_major=False), residual_x

    def _pool_frames(self, x):
        """[1+n, B*T, W] -> [1+n, B*T_new, W]: every token (CLS included) averaged over its segment's frames."""
        L.require_device(x)
        x = x.float().contiguous()
        n = x.shape[0] - 1
        op = torch.ops.centerclip.token_cluster_train if (torch.is_grad_enabled() and x.requires_grad) else None
        if op is not None:
            out, _, _ = op(x, False, self.before_block_frames, self.after_block_frames, n, 0, 2.0, 1e-6, 0, 16, False, 1, 0,
                           None, None, None, 0.0, 0, 0, False, None)
            return out
        out, _ = torch.ops.centerclip.token_cluster(x, False, self.before_block_frames, self.after_block_frames, n, 0, 2.0,
                                                    1e-6, 0, 16, False, 1, 0, None, None, None, False)
        return out

    def _sparse_ids_random(self, N, segments):
        """token_sparse_sampling(cluster_num, N, random_shift=True) (cluster_utils.py:150-162), drawn once per segment as the
        reference's loop does (cluster.py:332-336), with the same NumPy calls in the same order - the same global NumPy seed
        gives the same ids.  -> int64 [segments, K] (host)."""
        K = self.cluster_num
        rows = []
        for _ in range(segments):
            avg = N // K
            if avg > 0:
                off = np.multiply(list(range(K)), avg) + np.random.randint(avg, size=K)
            elif N > K:
                off = np.sort(np.random.choice(N, K, replace=False))
            else:
                off = np.clip(np.arange(0, K), 0, N)
            rows.append(np.asarray(off, dtype=np.int64))
        return torch.from_numpy(np.stack(rows))

    def _sparse_ids(self, N, device):
        """token_sparse_sampling(cluster_num, N, random_shift=False) (cluster_utils.py:136-170, eval branch):
        centres of cluster_num equal segments of the N tokens; the same ids for every problem."""
        key = (N, str(device))
        if getattr(self, "_sparse_key", None) != key:
            K = self.cluster_num
            if N > K:
                tick = N / float(K)
                offsets = np.array([int(tick / 2.0 + tick * i) for i in range(K)])
            else:
                offsets = np.clip(np.arange(0, K), 0, N)
            self._sparse_key, self._sparse_val = key, torch.from_numpy(offsets).long().to(device)
        return self._sparse_val

    def _spg_mask(self, device):
        """uint8 [N, N] form of the spatial-temporal graph the kernels read, built once per device."""
        key = str(device)
        cache = self.__dict__.setdefault("_spg_u8", {})
        if key not in cache:
            cache[key] = self.spg[0].to(device).ne(0).to(torch.uint8).contiguous()
        return cache[key]

    def variant(self, N, device):
        """-> (cc_cluster_variant for this module, tensors it points to)."""
        var = L.ClusterVariant()
        var.algorithm = {'kmediods++': 0, 'pooling': 1, 'sparse_sampling': 2, 'spectral': 3}[self.algorithm]
        var.aggregation = 0 if self.aggregation in [None, 'None'] else 1
        keep = []
        if self.cluster_embedding:
            keep.append(self.cluster_embed.detach().to(device).float().contiguous())
            var.cluster_embed = keep[-1].data_ptr()
        if self.adaptive_cls:
            keep.append(self.cls_multiplier.detach().to(device).float().reshape(-1).contiguous())
            var.cls_multiplier = keep[-1].data_ptr()
        if self.algorithm == 'sparse_sampling':
            if self.training:
                # the reference draws fresh random ids per segment in training mode (cluster_utils.py:150-162); the fused
                # encoder caches this struct per model, so it would silently keep the eval-mode centre ids: refuse instead
                # (the module / block-level forwards draw the ids per call, see _run)
                raise NotImplementedError("cluster_algo='sparse_sampling' in training mode inside the fused encoder: "
                                          "use the module / block-level forwards (random ids per call)")
            keep.append(self._sparse_ids(N, device))
            var.fixed_ids = keep[-1].data_ptr()
        if self.algorithm == 'spectral':
            from .spectral import GRAPH_MODES
            var.spectral_sigma, var.spectral_graph_mode = float(self.spectral_sigma), GRAPH_MODES[self.spectral_graph]
            var.spectral_knn_k, var.spectral_correct_sign = int(self.spectral_knn_k), int(bool(self.svd_correct_sign))
            if self.spg is not None:
                keep.append(self._spg_mask(device))
                var.spectral_graph = keep[-1].data_ptr()
        var.mean_residual = int(self.mean_residual)                # (read by the fused encoders, clip.py:239-242)
        return var, keep

    @property
    def is_default_variant(self):
        return (self.algorithm == 'kmediods++' and self.aggregation in [None, 'None'] and not self.cluster_embedding
                and not self.adaptive_cls and not self.mean_residual)

    def _run(self, x, tok_stride, frame_stride, BT, Lt, W, frame_major, keep_ids=True):
        L.require_device(x)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        n = Lt - 1
        K = n if self.algorithm == 'pooling' else self.cluster_num
        N = self.frame_duration * n
        if self.algorithm == 'sparse_sampling' and self.training:
            # random ids per segment, shared by the clips (cluster.py:332-336): a gather with given ids - problem p = s*B + b
            B = BT // self.before_block_frames
            ids = self._sparse_ids_random(N, self.after_block_frames)                       # [T_new, K]
            med = ids.to(x.device).repeat_interleave(B, dim=0).contiguous()                 # [T_new*B, K]
            self.last_medoids = med
            return _GatherGiven.apply(x, med, bool(frame_major), self.before_block_frames, self.after_block_frames, K)
        if torch.is_grad_enabled() and (x.requires_grad or (self.training and any(p.requires_grad for p in self.parameters()))):
            # training: the differentiable op (gradient of the gather / cluster means / CLS mean for the selection made in
            # the forward pass, which is a constant of the backward pass as in the reference: fast_kmeans.py:13,44)
            embed = self.cluster_embed.to(x.device).float().contiguous() if self.cluster_embedding else None
            mult = self.cls_multiplier.to(x.device).float().reshape(-1) if self.adaptive_cls else None
            ids = self._sparse_ids(N, x.device) if self.algorithm == 'sparse_sampling' else None
            sp = (0.0, 0, 0, False, None)
            if self.algorithm == 'spectral':
                from .spectral import GRAPH_MODES
                graph = self._spg_mask(x.device) if self.spg is not None else None
                sp = (float(self.spectral_sigma), GRAPH_MODES[self.spectral_graph], int(self.spectral_knn_k),
                      bool(self.svd_correct_sign), graph)
            out, medoids, _ = torch.ops.centerclip.token_cluster_train(
                x, bool(frame_major), self.before_block_frames, self.after_block_frames, K, L.METRIC_IDS[self.distance],
                float(self.norm_p), float(self.threshold), int(self.iter_limit), int(self.split_size), bool(self.pre_norm),
                {'kmediods++': 0, 'pooling': 1, 'sparse_sampling': 2, 'spectral': 3}[self.algorithm],
                0 if self.aggregation in [None, 'None'] else 1, embed, mult, ids, *sp)
            self.last_medoids = medoids if medoids.numel() else None
            return out
        var, keep = self.variant(N, x.device)
        out, medoids = torch.ops.centerclip.token_cluster(
            x, bool(frame_major), self.before_block_frames, self.after_block_frames, K, L.METRIC_IDS[self.distance],
            float(self.norm_p), float(self.threshold), int(self.iter_limit), int(self.split_size), bool(self.pre_norm),
            int(var.algorithm), int(var.aggregation),
            keep[0] if self.cluster_embedding else None,
            keep[1 if self.cluster_embedding else 0] if self.adaptive_cls else None,
            keep[-1] if self.algorithm == 'sparse_sampling' else None,
            bool(keep_ids and self.algorithm in ('kmediods++', 'spectral')),
            float(var.spectral_sigma), int(var.spectral_graph_mode), int(var.spectral_knn_k), bool(var.spectral_correct_sign),
            keep[-1] if (self.algorithm == 'spectral' and self.spg is not None) else None)
        self.last_medoids = medoids if medoids.numel() else None
        return out
