"""Launch the kernels whose HBM traffic we report (dominant GEMM of the step, token-cluster kernels)
a few times each, for a rocprofv3 --pmc pass (tools/pmc.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import ops
from centerclip_amd.cluster import TokenClusterInter
W = 768
M = 9600
# the three dominant GEMMs of a block, with the epilogues the encoders really use (LayerNorm folded)
for name, N, K in [("c_fc", 3072, 768), ("in_proj", 2304, 768)]:
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    h16, _, _ = ops.row_stats(torch.randn(M, K, device="cuda"))
    stats = torch.randn(M, 12, 2, device="cuda").abs()
    wf, c1, c2 = ops.fold_layernorm_linear(w, b, torch.ones(K, device="cuda"), torch.zeros(K, device="cuda"))
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(5):
        ops.linear_ln_f16(h16, wf, c1, c2, stats, 12, gelu=(name == "c_fc"), out=out)
    if name == "in_proj":                       # the form the encoders run: the frames' attention in the same launch
        for _ in range(5):
            ops.inproj_attention_f16(h16, wf, c1, c2, stats, 12, M // 50, 50, K // 64)
    torch.cuda.synchronize()
a = torch.randn(M, 3072, device="cuda").half(); w = (torch.randn(768, 3072, device="cuda") * 3072 ** -0.5).half()
b = torch.randn(768, device="cuda"); hres = torch.zeros(M, 768, device="cuda")
h16b = torch.empty(M, 768, device="cuda", dtype=torch.float16); stb = torch.empty(M * 64, device="cuda")
_, st_in, sh_in = ops.row_stats(torch.randn(M, 768, device="cuda"))
sh_out = torch.empty(M, device="cuda")
for _ in range(5):
    ops.linear_resid_stats_f16(a, w, b, hres, h16=h16b, stats=stb, shift_in=sh_in, stats_in=st_in.view(M, 1, 2), shift_out=sh_out)
torch.cuda.synchronize()
# out_proj shape on the same residual epilogue
a2 = torch.randn(M, 768, device="cuda").half(); w2 = (torch.randn(768, 768, device="cuda") * 768 ** -0.5).half()
for _ in range(5):
    ops.linear_resid_stats_f16(a2, w2, b, hres, h16=h16b, stats=stb, shift_in=sh_in, stats_in=st_in.view(M, 1, 2), shift_out=sh_out)
torch.cuda.synchronize()
x = torch.randn(16 * 12, 50, W, device="cuda")
mod = TokenClusterInter(before_cluster_num=49, cluster_num=49, before_block_frames=12, after_block_frames=3,
                        original_frame=12, threshold=1e-6, iter_limit=100, split_size=16, norm_p=2.0)
for _ in range(5):
    mod.cluster_frame_major(x)
torch.cuda.synchronize()
# the similarity tail at the north-star size (prepare kernel + the GEMM on the concatenated planes)
t = torch.randn(10000, 512, device="cuda"); v = torch.randn(1000, 3, 512, device="cuda")
m = torch.ones(1000, 3, dtype=torch.long, device="cuda")
for _ in range(5):
    ops.loose_similarity(t, v, m, 1.0)
torch.cuda.synchronize()
