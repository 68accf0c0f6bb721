"""Stand-alone time of the backward's GEMMs (fp32 output, EPI_F32) per tile id: dW = dY^T X (contraction = the padded row count)
and dX = dY W (dev tool).  usage: python tools/wgrad_sweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from centerclip_amd import ops
dev = "cuda"
shapes = []
for rows in (9600, 2432, 4928):
    for name, N, K in (("in_proj", 2304, 768), ("out_proj", 768, 768), ("c_fc", 3072, 768), ("c_proj", 768, 3072)):
        if rows != 9600 and name in ("out_proj",):
            continue
        shapes.append(("dW " + name, N, K, rows))            # [N, rows] x [K, rows]^T
for name, N, K in (("in_proj", 2304, 768), ("out_proj", 768, 768), ("c_fc", 3072, 768), ("c_proj", 768, 3072)):
    shapes.append(("dX " + name, 9600, K, N))                 # [rows, N] x [K, N]^T
for role, M, N, K in shapes:
    a = torch.randn(M, K, device=dev).half()
    w = torch.randn(N, K, device=dev).half()
    line = "%-12s %5d x %4d x %5d" % (role, M, N, K)
    for t in (0, 1, 2, 3, 4, 8, 6, 5, 10):
        try:
            fn = lambda: ops.linear_f16(a, w, None, "f32", tile=t)
            ms = bench.graph_time_ms(fn, launches=10, replays=3)
            line += " | t%d %6.1fus %4.0fTF" % (t, ms * 1e3, 2.0 * M * N * K / ms / 1e9)
        except Exception as e:
            line += " | t%d n/a" % t
    print(line, flush=True)
