"""Main-loop ablation of the 256x256 GEMM tile (dev experiment): per-workgroup phase cycles for ab/lib_abl<k>.so builds
(ABL bits: 1 = no barrier, 2 = no LDS-DMA in the loop, 4 = no fragment reads in the loop; results are numerically wrong)."""
import sys, os, ctypes, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from centerclip_amd import ops, _lib as L
    lib = L.lib()
    lib.cc_debug_set_gemm_profile.argtypes = [ctypes.c_void_p]
    for M, N, K, epi, tile in [(9600, 3072, 768, "f16", 5), (9600, 2304, 768, "f16", 7), (9600, 768, 3072, "f32_resid", 6),
                               (8192, 8192, 4096, "f16", 5),
                               (2400, 2304, 768, "f16", 1), (2400, 3072, 768, "f16_gelu", 1), (9408, 768, 3072, "f32", 1)]:
        a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
        out = torch.zeros(M, N, device="cuda", dtype=torch.float16 if epi.startswith("f16") else torch.float32)
        for _ in range(3): ops.linear_f16(a, w, None, epi, out=out, tile=tile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.linear_f16(a, w, None, epi, out=out, tile=tile)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        buf = torch.zeros(1 << 16, 4, dtype=torch.long, device="cuda")
        lib.cc_debug_set_gemm_profile(ctypes.c_void_p(buf.data_ptr()))
        ops.linear_f16(a, w, None, epi, out=out, tile=tile)
        torch.cuda.synchronize()
        lib.cc_debug_set_gemm_profile(ctypes.c_void_p(0))
        t = buf.cpu().double(); t = t[t[:, 3] > 0]
        loop = (t[:, 2] - t[:, 1]).mean()
        print(f"  {M}x{N}x{K}: {us:7.1f} us {2.0*M*N*K/us/1e6:6.0f} TF | loop {loop/(K/64):.0f} cycles/k-step, prologue {(t[:,1]-t[:,0]).mean():.0f}, epilogue {(t[:,3]-t[:,2]).mean():.0f}", flush=True)
else:
    for v in sys.argv[1:] or ["abl0", "abl1", "abl2", "abl4", "abl6", "abl7"]:
        print("variant", v, flush=True)
        env = dict(os.environ, CENTERCLIP_HIP_LIB=os.path.join(ROOT, "ab", "lib_%s.so" % v))
        subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=env)
