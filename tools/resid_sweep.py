"""Sweep the residual + statistics epilogue (out_proj / c_proj shapes) over tiles, timed as hipGraph replays (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from centerclip_amd import ops
dev = "cuda"
SHAPES = [(9600, 768, 768), (9600, 768, 3072), (2400, 768, 768), (2400, 768, 3072), (512, 512, 512), (512, 512, 2048)]
if len(sys.argv) > 1 and sys.argv[1] == "big":          # the other BASELINE towers: cfg3 (B = 64), cfg4 (T = 64, B = 8), cfg5 (ViT-B/16) + their clustered blocks
    SHAPES = [(M, 768, K) for M in (12800, 25600, 37824, 38400, 6464, 3200) for K in (768, 3072)]
TILES = (0, 1, 5, 6, 8) if len(sys.argv) > 1 else (0, 1, 4, 6, 8, 10)
for M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.randn(N, device=dev); h = torch.zeros(M, N, device=dev)
    h16 = torch.empty(M, N, device=dev, dtype=torch.float16); st = torch.empty(M * 64, device=dev)
    _, st_in, sh_in = ops.row_stats(torch.randn(M, N, device=dev)); sh_out = torch.empty(M, device=dev)
    line = "%5d x %4d x %4d" % (M, N, K)
    for t in TILES:
        try:
            fn = lambda: ops.linear_resid_stats_f16(a, w, b, h, tile=t, h16=h16, stats=st, shift_in=sh_in, stats_in=st_in.view(M, 1, 2), shift_out=sh_out)
            ms = bench.graph_time_ms(fn, launches=20, replays=3)
            line += " | t%d %6.1fus %4.0fTF" % (t, ms * 1e3, 2.0 * M * N * K / ms / 1e9)
        except Exception as e:
            line += " | t%d n/a" % t
    print(line, flush=True)
