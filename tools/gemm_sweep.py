"""Sweep the fp16 GEMM over shapes x tile configs (dev tool, GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import ops

def timeit(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

shapes = [(4096, 4096, 4096, "f16"), (8192, 8192, 8192, "f16"), (9408, 768, 3072, "f32"), (9600, 2304, 768, "f16"), (9600, 768, 768, "f32_resid"),
          (9600, 3072, 768, "f16_gelu"), (9600, 768, 3072, "f32_resid"), (2400, 2304, 768, "f16"), (2400, 768, 768, "f32_resid"),
          (2400, 3072, 768, "f16_gelu"), (2400, 768, 3072, "f32_resid"), (512, 1536, 512, "f16"), (512, 512, 512, "f32_resid"),
          (512, 2048, 512, "f16_gelu"), (512, 512, 2048, "f32_resid")]
tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3, 4]
for M, N, K, epi in shapes:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float16 if epi.startswith("f16") else torch.float32)
    line = f"{M:5d}x{N:5d}x{K:5d} {epi:9s}"
    for t in tiles:
        try:
            ms = timeit(lambda: ops.linear_f16(a, w, b, epi, out=out, tile=t))
            line += f" | t{t}: {ms*1e3:7.1f}us {2.0*M*N*K/ms/1e9:6.0f}TF"
        except Exception as e:
            line += f" | t{t}: n/a"
    print(line, flush=True)
