"""Tile choice for the similarity GEMM shape (10k x 1k, K = 3*512 concatenated planes) + the end-to-end similarity call (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, bench_side
from centerclip_amd import ops
dev = "cuda"
for (M, N) in ((10000, 1024), (1250, 1024), (10000, 10240)):
    a = torch.randn(M, 1536, device=dev).half(); w = torch.randn(N, 1536, device=dev).half()
    out = torch.zeros(M, N, device=dev)
    for tile in (0, 1, 3, 5, 6):
        try:
            ms = bench.graph_time_ms(lambda: ops.linear_f16(a, w, None, "f32", out=out, tile=tile), launches=10, replays=4)
            print("M=%d N=%d tile %d: %.1f us  %.0f TF issued" % (M, N, tile, ms * 1e3, 2.0 * M * N * 1536 / ms / 1e9), flush=True)
        except Exception as e:
            print("tile", tile, "failed", e)
print(bench_side.similarity_bench(torch.device(dev)))
