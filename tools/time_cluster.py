"""Quick timing of the token-cluster op at the BASELINE.json shapes (dev tool, GPU only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centerclip_amd.cluster import TokenClusterInter, batch_fast_kmedoids_with_split
from centerclip_amd.cluster.fast_kmeans import _run

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

cfgs = {"cfg2": (16, 12, 3, 49, 49, 16), "cfg3": (64, 12, 4, 49, 49, 16), "cfg4": (8, 64, 8, 49, 49, 16), "cfg5": (16, 12, 4, 196, 100, 4)}
for name, (B, T, Tn, n, K, split) in cfgs.items():
    W = 768
    for dist_kind in ("gauss", "lattice"):
        g = torch.Generator(device="cuda").manual_seed(0)
        if dist_kind == "gauss":
            x = torch.randn(B * T, 1 + n, W, device="cuda", generator=g)
        else:
            x = torch.randint(-3, 4, (B * T, 1 + n, W), device="cuda", generator=g).float()
        for p in (2.0, 1.0):
            mod = TokenClusterInter(before_cluster_num=n, cluster_num=K, before_block_frames=T, after_block_frames=Tn,
                                    original_frame=T, threshold=1e-6, iter_limit=100, split_size=split, norm_p=p)
            ms = timeit(lambda: mod.cluster_frame_major(x))
            P, N = B * Tn, (T // Tn) * n
            X = x[:, 1:, :].reshape(B, Tn, T // Tn, n, W).permute(1, 0, 2, 3, 4).reshape(P, N, W).contiguous()
            a, m, it = _run(X, K, "euclidean", 1e-6, 100, True, p, split, False, return_iters=True)
            tokens = P * N
            alg_bytes = tokens * W * 4 + P * K * W * 4 + P * K * 8
            print(f"{name} {dist_kind:7s} p={p}: {ms*1e3:8.1f} us/call  {tokens/ms/1e3:8.2f} Mtok/s  alg {alg_bytes/ms/1e6:7.1f} GB/s  iters mean {it.float().mean():.1f} max {int(it.max())}")
