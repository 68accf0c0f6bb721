"""Known-byte-count launches for calibrating the TCC counters on gfx950 (tools/pmc_reconcile.sh):
  A  torch copy of 512 MiB fp32 (plain 16-byte global loads / stores, far beyond the 256 MiB Infinity Cache)
  B  the 256x128 GEMM tile with ONE column tile (N = 128): its A operand (59 MB, LDS-DMA global_load_lds dwordx4) crosses
     the fabric exactly once, W (0.79 MB) once per XCD, the fp16 output 2.5 MB
  C  the dominant symbol itself (c_proj / out_proj with the residual epilogue), 6 column tiles
Each 4 times, so the per-launch averages are of warm launches."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import ops

dev = "cuda"
src = torch.randn(128 * 1024 * 1024, device=dev)
dst = torch.empty_like(src)
for _ in range(4):
    dst.copy_(src)
torch.cuda.synchronize()
del src, dst
M = 9600
a = torch.randn(M, 3072, device=dev).half()
w1 = (torch.randn(128, 3072, device=dev) * 3072 ** -0.5).half()
b1 = torch.randn(128, device=dev)
o1 = torch.empty(M, 128, device=dev, dtype=torch.float16)
for _ in range(4):
    ops.linear_f16(a, w1, b1, "f16", out=o1, tile=6)
torch.cuda.synchronize()
w = (torch.randn(768, 3072, device=dev) * 3072 ** -0.5).half()
b = torch.randn(768, device=dev)
hres = torch.zeros(M, 768, device=dev)
h16b = torch.empty(M, 768, device=dev, dtype=torch.float16)
stb = torch.empty(M * 64, device=dev)
_, st_in, sh_in = ops.row_stats(torch.randn(M, 768, device=dev))
sh_out = torch.empty(M, device=dev)
for _ in range(4):
    ops.linear_resid_stats_f16(a, w, b, hres, h16=h16b, stats=stb, shift_in=sh_in, stats_in=st_in.view(M, 1, 2), shift_out=sh_out)
torch.cuda.synchronize()
a2 = torch.randn(M, 768, device=dev).half()
w2 = (torch.randn(768, 768, device=dev) * 768 ** -0.5).half()
for _ in range(4):
    ops.linear_resid_stats_f16(a2, w2, b, hres, h16=h16b, stats=stb, shift_in=sh_in, stats_in=st_in.view(M, 1, 2), shift_out=sh_out)
torch.cuda.synchronize()
