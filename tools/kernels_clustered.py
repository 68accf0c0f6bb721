"""The launches of one clustered block (M = 2,400 rows: 48 segments x 50 tokens, cfg 2 behind block 7) for counter passes
(tools/pmc_reconcile.sh clustered python tools/kernels_clustered.py): in_proj + attention, out_proj, c_fc, c_proj - each 5 times."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import ops
M = 2400
b = torch.randn(768, device="cuda")
for name, N, K in [("c_fc", 3072, 768), ("in_proj", 2304, 768)]:
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    bq = torch.randn(N, device="cuda")
    h16, _, _ = ops.row_stats(torch.randn(M, K, device="cuda"))
    stats = torch.randn(M, 12, 2, device="cuda").abs()
    wf, c1, c2 = ops.fold_layernorm_linear(w, bq, torch.ones(K, device="cuda"), torch.zeros(K, device="cuda"))
    for _ in range(5):
        if name == "c_fc":
            ops.linear_ln_f16(h16, wf, c1, c2, stats, 12, gelu=True)
        else:
            ops.inproj_attention_f16(h16, wf, c1, c2, stats, 12, M // 50, 50, K // 64)
    torch.cuda.synchronize()
hres = torch.zeros(M, 768, device="cuda")
h16b = torch.empty(M, 768, device="cuda", dtype=torch.float16)
stb = torch.empty(M * 64, device="cuda")
_, st_in, sh_in = ops.row_stats(torch.randn(M, 768, device="cuda"))
sh_out = torch.empty(M, device="cuda")
for K in (768, 3072):
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(768, K, device="cuda") * K ** -0.5).half()
    for _ in range(5):
        ops.linear_resid_stats_f16(a, w, b, hres, h16=h16b, stats=stb, shift_in=sh_in, stats_in=st_in.view(M, 1, 2), shift_out=sh_out)
    torch.cuda.synchronize()
