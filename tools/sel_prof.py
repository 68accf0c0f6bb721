import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import _lib as L
from centerclip_amd.cluster.fast_kmeans import _run
lib = L.lib()
for (P, N, K, split) in [(48, 196, 49, 16), (64, 147, 49, 16), (8, 392, 49, 16), (4, 588, 100, 4)]:
    X = torch.randn(P, N, 768, device="cuda")
    buf = torch.zeros(P, 16, dtype=torch.long, device="cuda")
    _run(X, K, "euclidean", 1e-6, 100, True, 2.0, split, False)
    lib.cc_debug_set_select_profile(ctypes.c_void_p(buf.data_ptr()))
    _run(X, K, "euclidean", 1e-6, 100, True, 2.0, split, False)
    torch.cuda.synchronize()
    lib.cc_debug_set_select_profile(ctypes.c_void_p(0))
    b = buf.cpu().double()
    d = (b[:, 1:5] - b[:, 0:4])
    print(f"P={P} N={N} K={K}: cycles(100MHz ticks?) load {d[:,0].mean():.0f}  kkz {d[:,1].mean():.0f}  iters {d[:,2].mean():.0f} ({b[:,5].mean():.1f} it)  final {d[:,3].mean():.0f}  total {(b[:,4]-b[:,0]).mean():.0f} max {(b[:,4]-b[:,0]).max():.0f} | per-iter assign+masks {(b[:,6]/b[:,5]).mean():.0f} group {(b[:,7]/b[:,5]).mean():.0f} rowsum+argmin {(b[:,8]/b[:,5]).mean():.0f} | kkz per step: argmax {(b[:,9]/K).mean():.0f} row-update {(b[:,10]/K).mean():.0f}")
