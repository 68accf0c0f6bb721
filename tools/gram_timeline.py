"""Real-time stamps of the Gram / distance kernel K1 (dev tool; library built with -DCC_DEV_KNOBS): per workgroup entry, first
stage in LDS, k loop done, exit - for the cfg-2 cluster call (48 problems x 196 tokens x 768)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import _lib as L
from centerclip_amd.cluster.fast_kmeans import _run
lib = L.lib()
lib.cc_debug_set_gram_profile.argtypes = [ctypes.c_void_p]
for (P, N, K, split) in [(48, 196, 49, 16), (256, 147, 49, 16), (64, 392, 49, 16)]:
    X = torch.randn(P, N, 768, device="cuda")
    buf = torch.zeros(4096, 4, dtype=torch.long, device="cuda")
    for _ in range(3): _run(X, K, "euclidean", 1e-6, 100, True, 2.0, split, False)
    lib.cc_debug_set_gram_profile(ctypes.c_void_p(buf.data_ptr()))
    _run(X, K, "euclidean", 1e-6, 100, True, 2.0, split, False)
    torch.cuda.synchronize()
    lib.cc_debug_set_gram_profile(ctypes.c_void_p(0))
    t = buf.cpu().double()
    t = t[t[:, 3] > 0] / 100.0
    t0 = t[:, 0].min()
    n = len(t)
    print(f"P={P} N={N}: {n} workgroups | entries: median {float((t[:,0]-t0).median()):.1f} last {float((t[:,0]-t0).max()):.1f} | exits: median "
          f"{float((t[:,3]-t0).median()):.1f} last {float((t[:,3]-t0).max()):.1f} | per workgroup: prologue {float((t[:,1]-t[:,0]).mean()):.1f} "
          f"loop {float((t[:,2]-t[:,1]).mean()):.1f} (max {float((t[:,2]-t[:,1]).max()):.1f}) epilogue {float((t[:,3]-t[:,2]).mean()):.1f} us", flush=True)
