"""Mirror of modules/cluster/cluster.py: get_cluster_inter (:15-63) and the kmediods++ /
aggregation=None branch of TokenClusterInter (:66-352)."""
import torch

from .. import _lib as L


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
                             transformer_width=width, pre_norm=getattr(args, 'pre_norm', False))


class TokenClusterInter(torch.nn.Module):
    """Token clustering between transformer blocks: T frames -> T_new segments, the fd*n patch
    tokens of a segment -> K medoid tokens (ascending ids), CLS = mean of the segment's CLS.

    Only the path the shipped scripts use is built (algorithm 'kmediods++', aggregation None,
    no learnable extras); other options raise NotImplementedError at construction.
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
        if algorithm != 'kmediods++':
            raise NotImplementedError("centerclip_amd builds cluster_algo='kmediods++' only (got %r)" % algorithm)
        if aggregation not in [None, 'None']:
            raise NotImplementedError("aggregation=%r is not built (medoid tokens only)" % aggregation)
        if cluster_embedding or cluster_frame_embedding or adaptive_cls or mean_residual:
            raise NotImplementedError("cluster_embedding / cluster_frame_embedding / adaptive_cls / mean_residual "
                                      "are not built")
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
        self.last_medoids = None

    def cluster_frame_major(self, x_nld, keep_ids=False):
        """Fast path used by the HIP transformer: x [B*T, 1+n, W] (frame-major) ->
        [B*T_new, 1+K, W].  Same arithmetic as forward(), different strides."""
        BT, Lt, W = x_nld.shape
        return self._run(x_nld, tok_stride=W, frame_stride=Lt * W, BT=BT, Lt=Lt, W=W, frame_major=True,
                         keep_ids=keep_ids)

    def forward(self, x):
        """x [1+n, B*T, W] (LND) -> (x' [1+K, B*T_new, W], None)   (cluster.py:206,350-352)"""
        Lt, BT, W = x.shape
        return self._run(x, tok_stride=BT * W, frame_stride=W, BT=BT, Lt=Lt, W=W, frame_major=False), None

    def _run(self, x, tok_stride, frame_stride, BT, Lt, W, frame_major, keep_ids=True):
        L.require_device(x)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        T, T_new, K = self.before_block_frames, self.after_block_frames, self.cluster_num
        B, n = BT // T, Lt - 1
        lib = L.lib()
        if frame_major:
            out = torch.empty(B * T_new, 1 + K, W, dtype=torch.float32, device=x.device)
            o_tok, o_frame = W, (1 + K) * W
        else:
            out = torch.empty(1 + K, B * T_new, W, dtype=torch.float32, device=x.device)
            o_tok, o_frame = B * T_new * W, W
        medoids = torch.empty(B * T_new, K, dtype=torch.long, device=x.device) if keep_ids else None
        N = self.frame_duration * n
        ws = L.workspace(lib.cc_cluster_workspace_bytes(B * T_new, N, W, int(bool(self.pre_norm))), x.device)
        L.check(lib.cc_token_cluster_f32(L.ptr(x), tok_stride, frame_stride, B, T, T_new, n, W, K,
                                         L.METRIC_IDS[self.distance], float(self.norm_p), float(self.threshold),
                                         int(self.iter_limit), int(self.split_size), int(bool(self.pre_norm)),
                                         L.ptr(out), o_tok, o_frame, L.ptr(medoids), None, None,
                                         L.ptr(ws), ws.numel(), L.stream_ptr(x.device)), "cc_token_cluster_f32")
        self.last_medoids = medoids
        return out
