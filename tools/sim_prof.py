"""Kernel breakdown of one 10k x 1k similarity call (dev tool; run under tools/prof.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import ops
t = torch.randn(10000, 512, device="cuda"); v = torch.randn(1000, 3, 512, device="cuda")
m = torch.ones(1000, 3, dtype=torch.long, device="cuda")
tn = ops.normalize_rows(t); pv = ops.video_pool_normalize(v, m)
for _ in range(30):
    ops.loose_similarity(t, v, m, 1.0)
for _ in range(30):
    ops.scaled_dot_nt(tn, pv, 2.0)
torch.cuda.synchronize()
