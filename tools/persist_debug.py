"""Persistent GEMM (tile 11): which tiles differ from the 256x256 tiled kernel (none should), and a per-workgroup
real-time timeline (library built with -DCC_DEV_KNOBS: CENTERCLIP_HIP_LIB=ab/lib_dev.so).  Dev tool, GPU only."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from centerclip_amd import ops, _lib as L
lib = L.lib()
dev = "cuda"
for M, N, K, gelu in [(9600, 3072, 768, True), (9600, 2304, 768, False)]:
    torch.manual_seed(M + N)
    h = torch.randn(M, K, device=dev) * 2 + 0.3
    w = torch.randn(N, K, device=dev) * K ** -0.5
    h16, st1, _ = ops.row_stats(h)
    wf, c1, c2 = ops.fold_layernorm_linear(w, torch.randn(N, device=dev), torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.2)
    y5 = ops.linear_ln_f16(h16, wf, c1, c2, st1, 1, gelu=gelu, tile=5)
    bad_total = 0
    for rep in range(20):
        y = ops.linear_ln_f16(h16, wf, c1, c2, st1, 1, gelu=gelu, tile=11)
        d = (y != y5)
        if bool(d.any()):
            bad_total += 1
            tm = d.view(-1).nonzero()[:, 0] // N // 256
            tn = d.view(-1).nonzero()[:, 0] % N // 256
            tiles = sorted(set(zip(tm.tolist(), tn.tolist())))
            print(f"rep {rep}: {int(d.sum())} elements differ in tiles (row tile, col tile): {tiles[:12]}", flush=True)
    print(f"{M}x{N}x{K}: {bad_total} of 20 calls differ from the tiled kernel", flush=True)
    if hasattr(lib, "cc_debug_set_persist_profile"):
        lib.cc_debug_set_persist_profile.argtypes = [ctypes.c_void_p]
        buf = torch.zeros(256, 16, dtype=torch.long, device=dev)
        for _ in range(3): ops.linear_ln_f16(h16, wf, c1, c2, st1, 1, gelu=gelu, tile=11)
        lib.cc_debug_set_persist_profile(ctypes.c_void_p(buf.data_ptr()))
        ops.linear_ln_f16(h16, wf, c1, c2, st1, 1, gelu=gelu, tile=11)
        torch.cuda.synchronize()
        lib.cc_debug_set_persist_profile(ctypes.c_void_p(0))
        t = buf.cpu().double() / 100.0
        t0 = t[:, 0].min()
        for b in (0, 8, 16, 100, 255):
            row = t[b]; row = row[row > 0] - t0
            print(f"  wg {b}: " + " ".join("%.1f" % x for x in row.tolist()))
        ends = torch.stack([r[r > 0].max() for r in t]) - t0
        print(f"  last stamp per workgroup: min {float(ends.min()):.1f} median {float(ends.median()):.1f} max {float(ends.max()):.1f} us")
    ms = bench.graph_time_ms(lambda: ops.linear_ln_f16(h16, wf, c1, c2, st1, 1, gelu=gelu, tile=11), launches=20, replays=3)
    ms5 = bench.graph_time_ms(lambda: ops.linear_ln_f16(h16, wf, c1, c2, st1, 1, gelu=gelu, tile=5), launches=20, replays=3)
    print(f"  graph time: persistent {ms * 1e3:.1f} us, tile 5 {ms5 * 1e3:.1f} us", flush=True)
