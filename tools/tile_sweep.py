"""Stand-alone time of the LayerNorm-folded GEMMs (in_proj / c_fc roles) and the residual ones per tile id, as hipGraph
replays of back-to-back launches (dev tool).  usage: python tools/tile_sweep.py [M ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from centerclip_amd import ops
dev = "cuda"
Ms = [int(a) for a in sys.argv[1:]] or [2400, 9600]
for M in Ms:
    for role, N, K, gelu in (("in_proj", 2304, 768, False), ("c_fc", 3072, 768, True)):
        hres = torch.randn(M, K, device=dev)
        h16, _, _ = ops.row_stats(hres)
        stats = torch.randn(M, 12, 2, device=dev).abs()
        w = (torch.randn(N, K, device=dev) * K ** -0.5)
        wf, c1, c2 = ops.fold_layernorm_linear(w, torch.randn(N, device=dev), torch.ones(K, device=dev), torch.zeros(K, device=dev))
        line = "%-8s %5d x %4d x %4d" % (role, M, N, K)
        for t in (0, 1, 5, 7, 10):
            try:
                fn = lambda: ops.linear_ln_f16(h16, wf, c1, c2, stats, 12, gelu=gelu, tile=t)
                ms = bench.graph_time_ms(fn, launches=20, replays=3)
                line += " | t%d %6.1fus %4.0fTF" % (t, ms * 1e3, 2.0 * M * N * K / ms / 1e9)
            except Exception as e:
                line += " | t%d n/a" % t
        print(line, flush=True)
