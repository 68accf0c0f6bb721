"""Real-time timeline of the in_proj + attention launch (dev tool, GPU only; library built with -DCC_DEV_KNOBS -DCC_STAMP_WALL,
and -DCC_ATTN_STAMP_MID to move the third stamp behind the q | k | v writes of the epilogue)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import ops, _lib as L
lib = L.lib()
lib.cc_debug_set_gemm_profile.argtypes = [ctypes.c_void_p]
SHAPES = [(192, 50, 12), (48, 50, 12)] if len(sys.argv) < 2 else [(192, 197, 12), (64, 101, 12), (192, 50, 12)]
for nseq, Lt, heads in SHAPES:
    W, M = heads * 64, nseq * Lt
    h16, st, _ = ops.row_stats(torch.randn(M, W, device="cuda"))
    wf, c1, c2 = ops.fold_layernorm_linear(torch.randn(3 * W, W, device="cuda") * W ** -0.5, torch.randn(3 * W, device="cuda"),
                                           torch.ones(W, device="cuda"), torch.zeros(W, device="cuda"))
    fused = lambda: ops.inproj_attention_f16(h16, wf, c1, c2, st, 1, nseq, Lt, heads)
    def two():
        qkv = ops.linear_ln_f16(h16, wf, c1, c2, st, 1)
        return ops.attention_f16(qkv, nseq, Lt, heads)
    for name, fn in (("fused", fused), ("two launches", two)):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{nseq} x {Lt} x {heads} heads, {name}: {e0.elapsed_time(e1) * 10:.1f} us back to back (allocation included)", flush=True)
    buf = torch.zeros(1 << 16, 4, dtype=torch.long, device="cuda")
    for _ in range(5): fused()
    lib.cc_debug_set_gemm_profile(ctypes.c_void_p(buf.data_ptr()))
    fused()
    lib.cc_debug_set_gemm_profile(ctypes.c_void_p(0))
    torch.cuda.synchronize()
    t = buf.cpu().double()
    t = t[t[:, 3] > 0] / 100.0
    t0 = t[:, 0].min()
    ent = (t[:, 0] - t0).sort().values; ext = (t[:, 3] - t0).sort().values
    n = len(t)
    q = lambda v, f: float(v[min(int(f * (n - 1)), n - 1)])
    print(f"  {n} workgroups | entries: median {q(ent, .5):.1f}, last {float(ent.max()):.1f} | exits: first {float(ext.min()):.1f}, median {q(ext, .5):.1f}, "
          f"last {float(ext.max()):.1f} | per workgroup: prologue {float((t[:,1]-t[:,0]).mean()):.1f} stamp1->2 {float((t[:,2]-t[:,1]).mean()):.1f} "
          f"stamp2->exit {float((t[:,3]-t[:,2]).mean()):.1f} total {float((t[:,3]-t[:,0]).mean()):.1f} us", flush=True)
