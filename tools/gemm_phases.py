"""Per-workgroup phase timeline of the fp16 GEMM (dev tool, GPU only): entry / prologue done / loop done / exit."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import ops, _lib as L
lib = L.lib()
lib.cc_debug_set_gemm_profile.argtypes = [ctypes.c_void_p]
shapes = [(9600, 3072, 768, "f16_gelu", 5), (9600, 3072, 768, "f16_gelu", 1), (9600, 3072, 768, "f16", 5),
          (9600, 768, 3072, "f32_resid", 1), (9600, 2304, 768, "f16", 1), (9600, 768, 768, "f32_resid", 1),
          (2400, 3072, 768, "f16_gelu", 1), (2400, 768, 3072, "f32_resid", 4), (8192, 8192, 8192, "f16", 5)]
for M, N, K, epi, tile in shapes:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float16 if epi.startswith("f16") else torch.float32)
    for _ in range(3): ops.linear_f16(a, w, b, epi, out=out, tile=tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear_f16(a, w, b, epi, out=out, tile=tile)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    buf = torch.zeros(1 << 16, 4, dtype=torch.long, device="cuda")
    lib.cc_debug_set_gemm_profile(ctypes.c_void_p(buf.data_ptr()))
    ops.linear_f16(a, w, b, epi, out=out, tile=tile)
    torch.cuda.synchronize()
    lib.cc_debug_set_gemm_profile(ctypes.c_void_p(0))
    t = buf.cpu().double()
    t = t[t[:, 3] > 0]
    t0 = t[:, 0].min()
    span = t[:, 3].max() - t0
    pro, loop, epi_t = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
    starts = (t[:, 0] - t0)
    late = (starts > 0.1 * span).float().mean()
    print(f"{M}x{N}x{K} {epi} tile{tile}: {us:6.1f} us, {2.0*M*N*K/us/1e6:5.0f} TF | WGs {len(t)} span {span:.0f} ticks ({span/us:.0f} ticks/us)"
          f" | prologue {pro.mean():.0f} loop {loop.mean():.0f} ({loop.mean()/(K/64):.0f}/iter) epilogue {epi_t.mean():.0f} total {(t[:,3]-t[:,0]).mean():.0f}"
          "", flush=True)
