"""Does a residual GEMM slow down when its A operand was NOT left in the Infinity Cache by the previous launch of the same
loop?  c_proj / out_proj shapes with the A operand (and the residual rows) rotating over enough distinct buffers to exceed
the 256 MB MALL, against the usual same-buffers loop (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from centerclip_amd import ops
dev = "cuda"
for M, N, K in [(9600, 768, 3072), (9600, 768, 768), (9600, 3072, 768)]:
    nb = 10
    As = [torch.randn(M, K, device=dev).half() for _ in range(nb)]
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.randn(N, device=dev)
    if N == 768:
        hs = [torch.zeros(M, N, device=dev) for _ in range(nb)]
        h16 = torch.empty(M, N, device=dev, dtype=torch.float16); st = torch.empty(M * 64, device=dev)
        _, st_in, sh_in = ops.row_stats(torch.randn(M, N, device=dev)); sh_out = torch.empty(M, device=dev)
        def call(i):
            ops.linear_resid_stats_f16(As[i], w, b, hs[i], h16=h16, stats=st, shift_in=sh_in, stats_in=st_in.view(M, 1, 2), shift_out=sh_out)
    else:
        outs = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(2)]
        def call(i):
            ops.linear_f16(As[i], w, b, "f16_gelu", out=outs[i % 2])
    state = {"i": 0}
    def hot():
        call(0)
    def cold():
        state["i"] = (state["i"] + 1) % nb
        call(state["i"])
    ms_hot = bench.graph_time_ms(hot, launches=20, replays=3)
    ms_cold = bench.graph_time_ms(cold, launches=20, replays=3)
    print("%5d x %4d x %4d: same buffers %6.1f us | operands rotating over %d buffers (%.0f MB) %6.1f us" %
          (M, N, K, ms_hot * 1e3, nb, nb * M * K * 2 / 1e6, ms_cold * 1e3), flush=True)
