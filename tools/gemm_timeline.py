"""Real-time timeline of one GEMM launch (dev tool, GPU only; needs a library built with -DCC_STAMP_WALL, e.g.
CENTERCLIP_HIP_LIB=ab/lib_wall.so): when the workgroups enter and leave, in microseconds from the first entry, beside the
duration of the launch measured with events around it."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import ops, _lib as L
lib = L.lib()
lib.cc_debug_set_gemm_profile.argtypes = [ctypes.c_void_p]
shapes = [(9600, 768, 768, "f32_resid", 6), (9600, 768, 3072, "f32_resid", 6), (9600, 768, 3072, "f32_resid", 10),
          (2400, 768, 768, "f32_resid", 8), (2400, 768, 3072, "f32_resid", 8)]
for M, N, K, epi, tile in shapes:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float16 if epi.startswith("f16") else torch.float32)
    for _ in range(20): ops.linear_f16(a, w, b, epi, out=out, tile=tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ops.linear_f16(a, w, b, epi, out=out, tile=tile)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    buf = torch.zeros(1 << 16, 4, dtype=torch.long, device="cuda")
    for _ in range(5): ops.linear_f16(a, w, b, epi, out=out, tile=tile)
    lib.cc_debug_set_gemm_profile(ctypes.c_void_p(buf.data_ptr()))
    ops.linear_f16(a, w, b, epi, out=out, tile=tile)
    lib.cc_debug_set_gemm_profile(ctypes.c_void_p(0))
    for _ in range(5): ops.linear_f16(a, w, b, epi, out=out, tile=tile)
    torch.cuda.synchronize()
    t = buf.cpu().double()
    t = t[t[:, 3] > 0] / 100.0                                           # microseconds
    t0 = t[:, 0].min()
    ent = (t[:, 0] - t0).sort().values; ext = (t[:, 3] - t0).sort().values
    n = len(t)
    q = lambda v, f: float(v[min(int(f * (n - 1)), n - 1)])
    first = ent[:256] if n > 256 else ent
    print(f"{M}x{N}x{K} {epi} tile {tile}: {us:5.1f} us / launch back to back, {n} workgroups | entries: first 256 within {float(first.max()):.1f} us, "
          f"median {q(ent, .5):.1f}, last {float(ent.max()):.1f} | exits: first {float(ext.min()):.1f}, median {q(ext, .5):.1f}, last {float(ext.max()):.1f} | "
          f"per workgroup: prologue {float((t[:,1]-t[:,0]).mean()):.1f} loop {float((t[:,2]-t[:,1]).mean()):.1f} epilogue {float((t[:,3]-t[:,2]).mean()):.1f} "
          f"total {float((t[:,3]-t[:,0]).mean()):.1f} us", flush=True)
