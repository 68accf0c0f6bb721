"""Launch the kernels whose HBM traffic we report (dominant GEMM of the step, token-cluster kernels)
a few times each, for a rocprofv3 --pmc pass (tools/pmc.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import ops
from centerclip_amd.cluster import TokenClusterInter
W = 768
for name, M, N, K, epi in [("c_fc", 9600, 3072, 768, "f16_gelu"), ("in_proj", 9600, 2304, 768, "f16"), ("c_proj", 9600, 768, 3072, "f32_resid")]:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float16 if epi.startswith("f16") else torch.float32)
    for _ in range(5):
        ops.linear_f16(a, w, b, epi, out=out)
    torch.cuda.synchronize()
x = torch.randn(16 * 12, 50, W, device="cuda")
mod = TokenClusterInter(before_cluster_num=49, cluster_num=49, before_block_frames=12, after_block_frames=3,
                        original_frame=12, threshold=1e-6, iter_limit=100, split_size=16, norm_p=2.0)
for _ in range(5):
    mod.cluster_frame_major(x)
torch.cuda.synchronize()
